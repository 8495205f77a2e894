/* vb_lane_pattern_kernel.h - the voxel loop, one lane per voxel, for white noise with SEVERAL precisions
 * (option noise-pattern, noisemodel_white.cc:166-226: timepoint t belongs to precision phi_index[t], 255 =
 * masked).
 *
 * Same rotated loop and the same building blocks as vb_lane_kernel.h; what changes is that every sum over
 * timepoints exists once per precision:
 *     A_k = sum_{t in k} J_t J_t',   u_k = sum_{t in k} J_t (y_t - g_t),   s_k = sum_{t in k} (y_t - g_t)^2
 * UpdateTheta (noisemodel_white.cc:275-363) needs sum_k E[phi_k] A_k and sum_k E[phi_k] u_k, UpdateNoise
 * (:228-273) and CalcFreeEnergy (:365-454) need k'Q_k k and tr(Sigma A_k) for each k. The class of a
 * timepoint is the same for all voxels, so which set of registers a timepoint's products are added to is a
 * wave-uniform branch, not a select: the cost per timepoint is that of the one-precision kernel.
 *
 * N is the number of moment sets the kernel carries (2 or 4); cfg.n_phis <= N of them are used. The series
 * is read in place (any element type, masked timepoints): these configurations are multi-echo data, rare
 * next to the one-precision case, and get the plain streaming pass rather than the tiled one. */
#pragma once

#include "vb_lane_kernel.h"

namespace fvb
{
template <int P, int N>
constexpr int lane_pattern_save_rows()
{
    return 3 * P + P * (P + 1) / 2 + 2 * N + 2;
}

#if defined(__HIPCC__)

template <int P, int N>
struct PatternMoments
{
    static constexpr int PT = P * (P + 1) / 2;
    double A[N][PT];
    double u[N][P];
    double s[N];
    double ml[P];
    bool precise;
};

template <int N>
struct PatternNoise
{
    double b[N], c[N]; // Gamma(scale b, shape c) per precision
    double count[N];   // timepoints of each class (trace of Q_k)
};

template <int K, int P, int N>
__device__ __forceinline__ void pattern_accumulate(PatternMoments<P, N> &mo, const double (&J)[P], double r)
{
#pragma unroll
    for (int i = 0; i < P; i++)
    {
#pragma unroll
        for (int j = 0; j <= i; j++)
            mo.A[K][tri(i, j)] += J[i] * J[j];
        mo.u[K][i] += J[i] * r;
    }
    mo.s[K] += r * r;
}

// the class of a timepoint is the same for every voxel: choosing the register set its products go to is a wave-uniform
// branch (classes beyond N - masked timepoints, 255 - go nowhere)
template <int P, int N, int K>
__device__ __forceinline__ void pattern_dispatch(PatternMoments<P, N> &mo, const double (&J)[P], double r, int k)
{
    if constexpr (K < N)
    {
        if (k == K)
            pattern_accumulate<K>(mo, J, r);
        else
            pattern_dispatch<P, N, K + 1>(mo, J, r, k);
    }
}

// wave-uniform class of timepoint t: 0..N-1, or -1 for a masked timepoint. The table lives in LDS (filled
// once per wavefront from cfg.phi_index): a read from global memory here would sit behind the prefetched
// samples in the in-order return queue and its wait would drain them.
__device__ __forceinline__ int pattern_class(const uint8_t *classes, int t)
{
    const int k = (int)classes[t];
    return __builtin_amdgcn_readfirstlane(k == 255 ? -1 : k);
}

// two waves per SIMD (256 registers each) while the moments fit comfortably, else one (512)
template <int P, int N>
constexpr int lane_pattern_waves()
{
    return 2 * N * (P * (P + 1) / 2 + P + 1) <= 64 ? 2 : 1;
}

// LinearizedFwdModel::ReCentre (fwdmodel_linear.cc:126-182) fused with the per-precision sums
template <class Model, int P, int N>
__device__ __forceinline__ int recentre_pattern(const KernelArgs &ka, const ModelArgs &ma, int v, const double (&centre)[P],
    PatternMoments<P, N> &mo, bool precise, const uint8_t *classes)
{
    constexpr int PT = P * (P + 1) / 2;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double tp[P], tp2[P], tp3[P], rden[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = centre[i] * 1e-5; // fwdmodel_linear.cc:157-161
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = centre[i] + delta;
        const double c3 = centre[i] - delta;
        tp[i] = to_model(tr, centre[i]); // fwdmodel.cc:375-379
        tp2[i] = to_model(tr, c2);
        tp3[i] = to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        mo.ml[i] = centre[i];
    }
#pragma unroll
    for (int k = 0; k < N; k++)
    {
#pragma unroll
        for (int i = 0; i < PT; i++)
            mo.A[k][i] = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
            mo.u[k][i] = 0;
        mo.s[k] = 0;
    }
    bool bad_offset = false, bad_jac = false;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(precise);
    mo.precise = precise;
    auto step = [&](int t, double y_cur) {
        double g, J[P];
        sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
#pragma unroll
        for (int i = 0; i < P; i++) // fwdmodel_linear.cc:134-140,174-181 (masked timepoints included)
            bad_jac |= !is_finite(J[i]);
        bad_offset |= !is_finite(g);
        const double r = y_cur - g;
        const int k = pattern_class(classes, t);
        pattern_dispatch<P, N, 0>(mo, J, r, k); // (a wave-uniform chain of branches over the N register sets)
    };
    FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, step)
    return bad_offset ? FVB_BAD_OFFSET : (bad_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}

// WhiteNoiseModel::UpdateTheta (noisemodel_white.cc:275-363): J'XJ and J'X(y - g) with X = diag(E[phi] of each
// timepoint's class) are the E[phi_k]-weighted sums of the per-class moments; the rest is update_theta with
// a precision of one.
template <int P, int N>
__device__ __forceinline__ bool update_theta_pattern(VoxelState<P> &st, const PatternMoments<P, N> &mo,
    const PatternNoise<N> &nz, int n_phis, double alpha)
{
    constexpr int PT = P * (P + 1) / 2;
    Moments<P> w;
#pragma unroll
    for (int i = 0; i < PT; i++)
        w.A[i] = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        w.u[i] = 0;
        w.ml[i] = mo.ml[i];
    }
    w.s = 0;
    w.precise = mo.precise;
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        if (k < n_phis)
        {
            const double phibar = nz.b[k] * nz.c[k]; // GammaDist::CalcMean
#pragma unroll
            for (int i = 0; i < PT; i++)
                w.A[i] += phibar * mo.A[k][i];
#pragma unroll
            for (int i = 0; i < P; i++)
                w.u[i] += phibar * mo.u[k][i];
        }
    }
    st.b = 1;
    st.c = 1;
    return update_theta<P>(st, w, alpha);
}

// k'Q_k k = s_k - 2 d'u_k + d'A_k d (see residual_terms) and tr(Sigma A_k), for every class
template <int P, int N>
__device__ __forceinline__ bool residual_terms_pattern(const VoxelState<P> &st, const PatternMoments<P, N> &mo, int n_phis,
    double tol, double (&kk)[N], double (&trSA)[N])
{
    double d[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        d[i] = st.m[i] - mo.ml[i];
    bool lost = false;
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        kk[k] = 0;
        trSA[k] = 0;
        if (k < n_phis)
        {
            double du = 0, dAd = 0, tr = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
            {
                du += d[i] * mo.u[k][i];
#pragma unroll
                for (int j = 0; j < P; j++)
                {
                    dAd += d[i] * mo.A[k][tri(i, j)] * d[j];
                    tr += st.Sig[tri(i, j)] * mo.A[k][tri(i, j)];
                }
            }
            kk[k] = mo.s[k] - 2 * du + dAd;
            trSA[k] = tr;
            const double scale = mo.s[k] + 2 * fabs(du) + fabs(dAd);
            lost |= !(kk[k] > tol * scale) && (scale > 0);
        }
    }
    return lost;
}

template <int P, int N>
__device__ __forceinline__ void trace_pattern(const VoxelState<P> &st, const PatternMoments<P, N> &mo, int n_phis, double (&trSA)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        double tr = 0;
        if (k < n_phis)
        {
#pragma unroll
            for (int i = 0; i < P; i++)
#pragma unroll
                for (int j = 0; j < P; j++)
                    tr += st.Sig[tri(i, j)] * mo.A[k][tri(i, j)];
        }
        trSA[k] = tr;
    }
}

// The reference's k = y - g(ml) + J (ml - m), squared and summed per class (noisemodel_white.cc:235-252): one
// more streaming pass about the old centre, taken by a wavefront in which some voxel's moment form cancelled
template <class Model, int P, int N>
__device__ __forceinline__ void exact_residual_pattern(const KernelArgs &ka, const ModelArgs &ma, int v,
    const PatternMoments<P, N> &mo, const double (&m)[P], double (&kk)[N], const uint8_t *classes)
{
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double tp[P], tp2[P], tp3[P], rden[P], nd[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = mo.ml[i] * 1e-5;
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = mo.ml[i] + delta;
        const double c3 = mo.ml[i] - delta;
        tp[i] = to_model(tr, mo.ml[i]);
        tp2[i] = to_model(tr, c2);
        tp3[i] = to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        nd[i] = mo.ml[i] - m[i];
    }
#pragma unroll
    for (int k = 0; k < N; k++)
        kk[k] = 0;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(mo.precise); // the Jacobian as the re-centre about ml computed it
    auto step = [&](int t, double y_cur) {
        double g, J[P];
        sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
        double Jd = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            FVB_MODEL_FP
            Jd += J[i] * nd[i];
        }
        const double r = y_cur - g + Jd;
        const double r2 = r * r;
        const int k = pattern_class(classes, t);
#pragma unroll
        for (int q = 0; q < N; q++) // (uniform: the class is the same in every lane)
            if (k == q)
                kk[q] += r2;
    };
    FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, step)
}

// WhiteNoiseModel::UpdateNoise (noisemodel_white.cc:228-273)
template <int N>
__device__ __forceinline__ void update_noise_pattern(const KernelArgs &ka, PatternNoise<N> &nz, int n_phis,
    const double (&kk)[N], const double (&trSA)[N])
{
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        if (k < n_phis)
        {
            const double tmp = kk[k] + trSA[k];
            nz.b[k] = 1 / (tmp * 0.5 + 1 / ka.cfg.noise_prior_b[k]);         // :255
            nz.c[k] = (nz.count[k] - 1) * 0.5 + ka.cfg.noise_prior_c[k];     // :263
            if (ka.cfg.locked_noise_stdev > 0)                               // :265-271
                nz.b[k] = 1 / nz.c[k] / ka.cfg.locked_noise_stdev / ka.cfg.locked_noise_stdev;
        }
    }
}

// lgamma / digamma of the noise shapes, which change once (the first UpdateNoise) or on a revert
template <int N>
struct PatternFCache
{
    double c_fn[N], lgamma_c[N], digamma_c[N];
};

// WhiteNoiseModel::CalcFreeEnergy (noisemodel_white.cc:365-454) + the prior's term (inference_vb.cc:310)
template <int P, int N>
__device__ __forceinline__ bool calc_free_energy_pattern(const KernelArgs &ka, VoxelState<P> &st, const PatternNoise<N> &nz,
    int n_phis, const double (&kk)[N], const double (&trSA)[N], double Fprior, PatternFCache<N> &fc, bool &logdet_valid,
    double &F, bool &finite)
{
    bool ok = true;
    if (!logdet_valid) // (see free_energy_partial, vb_lane_kernel.h)
    {
        st.precValid = false;
        ok = ensure_prec<P>(st);
        logdet_valid = true;
    }
    const double nq = (double)ka.n_unmasked;
    const double expectedLogThetaDist = 0.5 * st.logdetLam - 0.5 * P * (LOG_2PI + 1);
    double expectedLogPhiDist = 0, part0 = 0, part9 = 0, part2 = 0;
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        if (k < n_phis)
        {
            const double si = nz.b[k], ci = nz.c[k];
            const double siPrior = ka.cfg.noise_prior_b[k], ciPrior = ka.cfg.noise_prior_c[k];
            if (!(ci == fc.c_fn[k]))
            {
                fc.lgamma_c[k] = gammaln(ci);
                fc.digamma_c[k] = digamma(ci);
                fc.c_fn[k] = ci;
            }
            const double log_b = log(si);
            const double dg = fc.digamma_c[k] + log_b;
            expectedLogPhiDist += -fc.lgamma_c[k] - ci * log_b - ci + (ci - 1) * dg;
            part0 += dg * (nz.count[k] * 0.5 + ciPrior - 1);
            part9 += -gammaln(ciPrior) - ciPrior * log(siPrior) - si * ci / siPrior;
            part2 += -0.5 * si * ci * kk[k] - 0.5 * trSA[k]; // (trace unscaled, :416-417)
        }
    }
    double logdetPrior = 0, quad = 0, trSL0 = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(st.pprec[i]));
        const double dm = st.m[i] - st.pm[i];
        quad += dm * st.pprec[i] * dm;
        trSL0 += st.Sig[tri(i, i)] * st.pprec[i];
    }
    double parts = part0;
    parts += part2;
    parts += 0.5 * logdetPrior - 0.5 * nq * LOG_2PI - 0.5 * P * LOG_2PI;
    parts += -0.5 * quad;
    parts += -0.5 * trSL0;
    parts += part9;
    F = -expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior;
    return ok;
}

template <int P, int N>
__device__ __forceinline__ void save_posterior_pattern(const KernelArgs &ka, int v, const VoxelState<P> &st,
    const PatternNoise<N> &nz, bool logdet_valid)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double *p = ka.save + v;
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.m[i];
#pragma unroll
    for (int i = 0; i < PT; i++)
        p[(size_t)(r++) * V] = st.Sig[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pm[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pprec[i];
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        p[(size_t)(r++) * V] = nz.b[k];
        p[(size_t)(r++) * V] = nz.c[k];
    }
    p[(size_t)(r++) * V] = st.logdetLam;
    p[(size_t)(r++) * V] = logdet_valid ? 1.0 : 0.0;
}

template <int P, int N>
__device__ __forceinline__ void restore_posterior_pattern(const KernelArgs &ka, int v, VoxelState<P> &st, PatternNoise<N> &nz,
    bool &logdet_valid)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const double *p = ka.save + v;
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        st.m[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < PT; i++)
        st.Sig[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pm[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pprec[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        nz.b[k] = p[(size_t)(r++) * V];
        nz.c[k] = p[(size_t)(r++) * V];
    }
    st.logdetLam = p[(size_t)(r++) * V];
    logdet_valid = p[(size_t)(r++) * V] != 0.0;
    st.covValid = true;
    st.precValid = false;
}

// The voxel loop (inference_vb.cc:398-571), rotated as in vb_lane_kernel (one streaming pass in the code,
// one site that evaluates F); F is evaluated when cfg.need_f says so (a run-time switch here).
template <class Model, int P, int N>
__global__ __launch_bounds__(64, (lane_pattern_waves<P, N>())) void vb_lane_pattern_kernel(const KernelArgs ka)
{
    constexpr int PT = P * (P + 1) / 2;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const int n_phis = ka.cfg.n_phis; // <= N
    const bool need_f = ka.cfg.need_f != 0;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    VoxelState<P> st;
    PatternMoments<P, N> mo;
    PatternNoise<N> nz;
    int status = FVB_OK;

    // noisemodel_white.cc:166-226: the class of every timepoint (dynamic LDS: T bytes), and trace(Q_k)
    extern __shared__ uint8_t classes[];
    for (int t = threadIdx.x; t < T; t += 64)
        classes[t] = ka.cfg.phi_index ? ka.cfg.phi_index[t] : (uint8_t)0;
    __syncthreads();
    if (v >= ka.cfg.n_voxels)
        return;
#pragma unroll
    for (int k = 0; k < N; k++)
        nz.count[k] = 0;
    for (int t = 0; t < T; t++)
    {
        const int k = pattern_class(classes, t);
#pragma unroll
        for (int j = 0; j < N; j++)
            nz.count[j] += (j == k) ? 1.0 : 0.0;
    }

    // ---- Vb::SetupPerVoxelDists, per-voxel part (inference_vb.cc:207-247) ----
    const int n = P + n_phis;
    if (ka.cfg.init_mvn)
    {
        // MVNDist::Load + GetSubmatrix + WhiteParams::InputFromMVN
        // (dist_mvn.cc:347-374,136-166; noisemodel_white.cc:70-79)
        const int nCov = n * (n + 1) / 2;
        const double *src = ka.cfg.init_mvn + v;
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = src[(size_t)i * V];
#pragma unroll
        for (int i = 0; i < P; i++)
            st.m[i] = src[(size_t)(nCov + i) * V];
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            nz.b[k] = 1;
            nz.c[k] = 1;
            if (k < n_phis)
            {
                const double nm = src[(size_t)(nCov + P + k) * V];
                const double nv = src[(size_t)tri(P + k, P + k) * V];
                nz.b[k] = nv / nm; // GammaDist::SetMeanVariance, dist_gamma.cc:29-33
                nz.c[k] = nm / nz.b[k];
            }
        }
    }
    else
    {
        // FwdModel::GetInitialPosterior (fwdmodel.cc:284-313)
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            st.m[i] = (ka.cfg.prior_type[i] == FVB_PRIOR_IMAGE) ? ka.cfg.image_prior[i][v] : ka.cfg.post_mean[i];
            st.Sig[tri(i, i)] = ka.cfg.post_var[i];
        }
        if (Model::needs_data_max)
        {
            double data_max = load_data(ka, v);
            for (int t = 1; t < T; t++)
            {
                const double y = load_data(ka, (size_t)t * V + v);
                data_max = (y > data_max) ? y : data_max;
            }
            Model::init_posterior(ma, data_max, st.m);
        }
#pragma unroll
        for (int i = 0; i < P; i++) // FwdModel::ToFabber, fwdmodel.cc:315-324
        {
            const int tr = ka.cfg.transform[i];
            st.m[i] = to_fabber(tr, st.m[i]);
            st.Sig[tri(i, i)] = to_fabber_var(tr, st.Sig[tri(i, i)]);
        }
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            nz.b[k] = (k < n_phis) ? ka.cfg.noise_post_b[k] : 1.0;
            nz.c[k] = (k < n_phis) ? ka.cfg.noise_post_c[k] : 1.0;
        }
    }
    st.covValid = true;
    st.precValid = false;
    st.logdetLam = 0;
    st.b = st.c = 1;
#pragma unroll
    for (int i = 0; i < P; i++) // fwd_prior = MVNDist(P): zero mean, identity (inference_vb.cc:159)
    {
        st.pm[i] = 0;
        st.pprec[i] = 1;
    }

    double F = 1234.5678; // inference_vb.cc:438
    double Fprior = 0;
    int it = 0;
    int hist_len = 0;
    bool setup_failed = false;

    ConvState conv;
    conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
    conv_reset(conv);
    const bool use_save = need_f && (ka.save != nullptr);
    enum
    {
        FIRST,
        ITERATING,
        REVERTED
    };
    int phase = FIRST;
    int n_lin = 0;
    bool logdet_valid = false;
    enum
    {
        LINEARISE,
        PRIORS,
        THETA,
        NOISE
    };
    int stage = LINEARISE;
    double kk[N], trSA[N]; // of the THETA stage, used again by NOISE
    PatternFCache<N> fcache;
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        kk[k] = trSA[k] = 0;
        fcache.c_fn[k] = __builtin_nan("");
        fcache.lgamma_c[k] = fcache.digamma_c[k] = 0;
    }
    for (;;)
    {
        bool want_f = need_f;
        if (stage == LINEARISE)
        {
            status = recentre_pattern<Model, P, N>(ka, ma, v, st.m, mo, n_lin < ka.precise_passes, classes);
            n_lin = (phase == REVERTED) ? n_lin : n_lin + 1;
            if (status != FVB_OK)
            {
                setup_failed = (phase == FIRST);
                break;
            }
            if (phase == FIRST)
            {
                if (use_save)
                    save_posterior_pattern<P, N>(ka, v, st, nz, logdet_valid); // :432-434
                want_f = false;
            }
        }
        else if (stage == PRIORS)
        {
            if (use_save && conv_need_save(conv)) // :451-458
                save_posterior_pattern<P, N>(ka, v, st, nz, logdet_valid);
            bool ok;
            if (need_f)
                ok = apply_priors<P, true>(ka, v, it, st, Fprior);
            else
                ok = apply_priors<P, false>(ka, v, it, st, Fprior);
            if (!ok)
            {
                status = FVB_BAD_RESULT;
                break;
            }
        }
        else if (stage == THETA)
        {
            if (!update_theta_pattern<P, N>(st, mo, nz, n_phis, conv_lm_alpha(conv)) || !ensure_cov<P>(st)) // :470
            {
                status = FVB_BAD_RESULT;
                break;
            }
            st.precValid = false;
            logdet_valid = true;
            const bool lost = residual_terms_pattern<P, N>(st, mo, n_phis, ka.residual_tol, kk, trSA);
            const int mode = ka.residual_mode; // 0 adaptive, 1 always exact, 2 moments only
            const bool want = (mode == 1) || (mode == 0 && lost);
            if (__any(want)) // wave-uniform: the pass is taken by the whole wavefront or not at all
            {
                double exact[N];
                exact_residual_pattern<Model, P, N>(ka, ma, v, mo, st.m, exact, classes);
                if (want) // per-voxel decision: a voxel's result never depends on its wave-mates
                {
#pragma unroll
                    for (int k = 0; k < N; k++)
                        kk[k] = exact[k];
                }
            }
        }
        else
        {
            update_noise_pattern<N>(ka, nz, n_phis, kk, trSA); // :479
        }
        if (want_f)
        {
            double Fn;
            bool fin = true, ok;
            if (stage == LINEARISE || stage == PRIORS) // the centre is the current mean, so k = y - g
            {
                if (!ensure_cov<P>(st))
                {
                    status = FVB_BAD_RESULT;
                    break;
                }
                double tr_now[N];
                trace_pattern<P, N>(st, mo, n_phis, tr_now);
                ok = calc_free_energy_pattern<P, N>(ka, st, nz, n_phis, mo.s, tr_now, Fprior, fcache, logdet_valid, Fn, fin);
            }
            else
            {
                ok = calc_free_energy_pattern<P, N>(ka, st, nz, n_phis, kk, trSA, Fprior, fcache, logdet_valid, Fn, fin);
            }
            if (!ok)
            {
                status = FVB_BAD_RESULT;
                break;
            }
            if (!fin)
            {
                status = FVB_BAD_FREE_ENERGY;
                break;
            }
            F = Fn;
        }
        if (stage == LINEARISE)
        {
            if (phase == REVERTED) // :516-525 done
                break;
            if (phase == ITERATING)
            {
                if (ka.out.f_history && hist_len < ka.cfg.f_history_rows) // :496-497
                    ka.out.f_history[(size_t)hist_len * V + v] = F;
                hist_len++;
                ++it;
                if (conv_test(conv, F)) // :500
                {
                    if (use_save && conv_need_save(conv)) // :506-513
                        save_posterior_pattern<P, N>(ka, v, st, nz, logdet_valid);
                    if (use_save && conv_need_revert(conv)) // :516-525
                    {
                        restore_posterior_pattern<P, N>(ka, v, st, nz, logdet_valid);
                        phase = REVERTED;
                        continue;
                    }
                    break;
                }
            }
            phase = ITERATING;
        }
        stage = (stage + 1) & 3;
    }

    // ---- result MVN: MVNDist(fwd_post, noise.OutputAsMVN()) packed as MVNDist::Save does
    // (inference_vb.cc:549-550; dist_mvn.cc:57-100,410-429; noisemodel_white.cc:55-68) ----
    if (!ensure_cov<P>(st))
    {
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = 0;
        if (status == FVB_OK)
            status = FVB_BAD_RESULT;
    }
    {
        double *dst = ka.out.mvn + v;
        const int nCov = n * (n + 1) / 2;
#pragma unroll
        for (int i = 0; i < PT; i++)
            dst[(size_t)i * V] = st.Sig[i];
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            if (k < n_phis)
            {
                for (int j = 0; j < P + k; j++)
                    dst[(size_t)tri(P + k, j) * V] = 0.0;
                dst[(size_t)tri(P + k, P + k) * V] = nz.b[k] * nz.b[k] * nz.c[k]; // GammaDist::CalcVariance
                dst[(size_t)(nCov + P + k) * V] = nz.b[k] * nz.c[k];               // GammaDist::CalcMean
            }
        }
#pragma unroll
        for (int i = 0; i < P; i++)
            dst[(size_t)(nCov + i) * V] = st.m[i];
        dst[(size_t)(nCov + n) * V] = 1.0;
    }
    if (ka.out.f_history && hist_len < ka.cfg.f_history_rows) // :553-554
        ka.out.f_history[(size_t)hist_len * V + v] = F;
    hist_len++;
    if (ka.out.f_history_len)
        ka.out.f_history_len[v] = hist_len;
    if (ka.out.free_energy)
        ka.out.free_energy[v] = F;
    if (ka.out.status)
        ka.out.status[v] = status | (setup_failed ? 0x100 : 0);
    if (ka.out.iterations)
        ka.out.iterations[v] = it;
}

#endif // __HIPCC__

} // namespace fvb
