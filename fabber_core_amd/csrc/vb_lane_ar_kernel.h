/*
 * vb_lane_ar_kernel.h - voxelwise VB with the AR(1) noise model, one lane per voxel.
 *
 * Reference: Ar1cNoiseModel (noisemodel_ar.cc) in the configuration BASELINE config 4 uses:
 * num-echoes = 1, ar1-cross-terms = "none" (one noise precision phi, two AR coefficients of
 * which only alpha_1 is ever updated, noisemodel_ar.cc:473-474). The reference keeps dense
 * (T x T) "alpha matrices" per voxel; they are three single diagonals (noisemodel_ar.cc:130-181)
 *     M00 = diag(0,1,...,1)   M20 = diag(1,...,1,0)   M10 = -1 on both first off-diagonals
 * and the marginal is Q = M00 + E[a] M10 + E[a^2] M20 (noisemodel_ar.cc:199-222). Here they are
 * never formed: one streaming pass per re-linearisation keeps, besides the white-noise moments
 * A_d = sum J_t J_t', u_d = sum J_t r_t, s_d = sum r_t^2, the lag-1 moments
 *     C = sum_{t<T} (J_t J_{t+1}' + J_{t+1} J_t'),  u_c = sum_{t<T} (J_t r_{t+1} + J_{t+1} r_t),
 *     s_c = sum_{t<T} r_t r_{t+1}
 * and the first / last rows (J_1, r_1), (J_T, r_T). Then for x, y in {J columns, r}
 *     x'M00 y = <x,y> - x_1 y_1,   x'M20 y = <x,y> - x_T y_T,   x'M10 y = -(lag-1 cross sum)
 * and every quantity of UpdateTheta / UpdateAlpha / UpdatePhi / CalcFreeEnergy
 * (noisemodel_ar.cc:447-747) follows with k = r - J d, d = m - ml.
 */
#pragma once

#include "vb_lane_kernel.h"

namespace fvb
{
#if defined(__HIPCC__)

constexpr double AR_ALPHA_PRIOR_PREC = 1e-4; // noisemodel_ar.cc:394-395

template <int P>
struct ArMoments
{
    static constexpr int PT = P * (P + 1) / 2;
    double Ad[PT], C[PT];
    double ud[P], uc[P];
    double sd, sc;
    double J1[P], JT[P];
    double r1, rT;
    double ml[P];
};

// Posterior of the AR coefficients: means, covariance (c11, c12, c22) and the marginal
// coefficients a1 = E[alpha_1], a2 = Var + E^2 that the last Update of the marginals produced
struct ArAlpha
{
    double mean[2];
    double c11, c12, c22;
    double a1, a2;
};

// Ar1cNoiseModel::HardcodedInitialDists (noisemodel_ar.cc:379-403), or the posterior of noise-initial-posterior
__device__ __forceinline__ void ar_initial_alpha(const KernelArgs &ka, ArAlpha &al)
{
    if (ka.cfg.ar_alpha_given & 2)
    {
        al.mean[0] = ka.cfg.ar_alpha_post_mean[0];
        al.mean[1] = ka.cfg.ar_alpha_post_mean[1];
        al.c11 = ka.cfg.ar_alpha_post_cov[0][0];
        al.c12 = ka.cfg.ar_alpha_post_cov[0][1];
        al.c22 = ka.cfg.ar_alpha_post_cov[1][1];
    }
    else
    {
        al.mean[0] = al.mean[1] = 0;
        al.c11 = al.c22 = 1.0 / AR_ALPHA_PRIOR_PREC;
        al.c12 = 0;
    }
}

template <int P>
constexpr int lane_ar_save_rows()
{
    return lane_save_rows<P>() + 7;
}

// J'MJ, J'Mr, r'Mr for M in {M00 (w=0), M10 (w=1), M20 (w=2)}
template <int P>
__device__ __forceinline__ double ar_JMJ(const ArMoments<P> &mo, int w, int i, int j)
{
    if (w == 0)
        return mo.Ad[tri(i, j)] - mo.J1[i] * mo.J1[j];
    if (w == 2)
        return mo.Ad[tri(i, j)] - mo.JT[i] * mo.JT[j];
    return -mo.C[tri(i, j)];
}
template <int P>
__device__ __forceinline__ double ar_JMr(const ArMoments<P> &mo, int w, int i)
{
    if (w == 0)
        return mo.ud[i] - mo.J1[i] * mo.r1;
    if (w == 2)
        return mo.ud[i] - mo.JT[i] * mo.rT;
    return -mo.uc[i];
}
template <int P>
__device__ __forceinline__ double ar_rMr(const ArMoments<P> &mo, int w)
{
    if (w == 0)
        return mo.sd - mo.r1 * mo.r1;
    if (w == 2)
        return mo.sd - mo.rT * mo.rT;
    return -2 * mo.sc;
}

// k'Mk from the moments with its cancellation flag, and tr(Sigma J'MJ)
template <int P>
__device__ __forceinline__ void ar_forms(const VoxelState<P> &st, const ArMoments<P> &mo, int w, double tol, double &kMk,
    double &trSJMJ, bool &lost)
{
    double d[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        d[i] = st.m[i] - mo.ml[i];
    double du = 0, dAd = 0;
    trSJMJ = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        du += d[i] * ar_JMr<P>(mo, w, i);
#pragma unroll
        for (int j = 0; j < P; j++)
        {
            const double a = ar_JMJ<P>(mo, w, i, j);
            dAd += d[i] * a * d[j];
            trSJMJ += st.Sig[tri(i, j)] * a;
        }
    }
    const double rr = ar_rMr<P>(mo, w);
    kMk = rr - 2 * du + dAd;
    const double scale = fabs(rr) + 2 * fabs(du) + fabs(dAd);
    lost = (w != 1) && !(kMk > tol * scale) && (scale > 0);
}

// Where a lane's series comes from: the tiled copy (the voxelwise kernel) or the caller's [t][voxel] image read in
// place (the spatial kernels, vb_spatial_noise.h). run(body) calls body(t, y_t) for t = 0 .. T-1 in order.
template <typename RAW>
struct ArTileFeed
{
    const RAW *lane_tile;
    int T;
    template <class Body>
    __device__ __forceinline__ void run(Body body) const
    {
        for_each_timepoint_tiled<RAW>(lane_tile, T, body);
    }
};
template <int P>
struct ArStridedFeed
{
    const KernelArgs &ka;
    int v;
    template <class Body>
    __device__ __forceinline__ void run(Body body) const
    {
        const int T = ka.cfg.n_times;
        const size_t V = (size_t)ka.cfg.n_voxels;
        FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, body)
    }
};

// The streaming pass: model + finite-difference Jacobian about `centre`, AR moments
template <class Model, int P, class Feed>
__device__ __forceinline__ int recentre_ar_feed(
    const KernelArgs &ka, const ModelArgs &ma, const Feed &feed, const double (&centre)[P], ArMoments<P> &mo, bool precise = false)
{
    constexpr int PT = P * (P + 1) / 2;
    const int T = ka.cfg.n_times;
    double tp[P], tp2[P], tp3[P], rden[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = centre[i] * 1e-5;
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = centre[i] + delta;
        const double c3 = centre[i] - delta;
        tp[i] = to_model(tr, centre[i]);
        tp2[i] = to_model(tr, c2);
        tp3[i] = to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        mo.ml[i] = centre[i];
        mo.ud[i] = 0;
        mo.uc[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < PT; i++)
    {
        mo.Ad[i] = 0;
        mo.C[i] = 0;
    }
    mo.sd = 0;
    mo.sc = 0;
    bool bad_offset = false, bad_jac = false;
    double Jp[P], rp = 0; // previous row
#pragma unroll
    for (int i = 0; i < P; i++)
        Jp[i] = 0;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(precise);
    auto step = [&](int t, double y_cur) {
        double g, J[P];
        sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
#pragma unroll
        for (int i = 0; i < P; i++)
            bad_jac |= !is_finite(J[i]);
        bad_offset |= !is_finite(g);
        const double r = y_cur - g;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
#pragma unroll
            for (int j = 0; j <= i; j++)
            {
                mo.Ad[tri(i, j)] += J[i] * J[j];
                mo.C[tri(i, j)] += Jp[i] * J[j] + J[i] * Jp[j]; // previous row is zero at t = 0
            }
            mo.ud[i] += J[i] * r;
            mo.uc[i] += Jp[i] * r + J[i] * rp;
        }
        mo.sd += r * r;
        mo.sc += rp * r;
        if (t == 0)
        {
#pragma unroll
            for (int i = 0; i < P; i++)
                mo.J1[i] = J[i];
            mo.r1 = r;
        }
#pragma unroll
        for (int i = 0; i < P; i++)
            Jp[i] = J[i];
        rp = r;
    };
    feed.run(step);
#pragma unroll
    for (int i = 0; i < P; i++)
        mo.JT[i] = Jp[i];
    mo.rT = rp;
    return bad_offset ? FVB_BAD_OFFSET : (bad_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}
template <class Model, int P, typename RAW>
__device__ __forceinline__ int recentre_ar(
    const KernelArgs &ka, const ModelArgs &ma, const RAW *lane_tile, const double (&centre)[P], ArMoments<P> &mo, bool precise = false)
{
    return recentre_ar_feed<Model, P>(ka, ma, ArTileFeed<RAW>{ lane_tile, ka.cfg.n_times }, centre, mo, precise);
}

// Direct k = y - g(ml) + J (ml - m): k'M00k, k'M10k, k'M20k summed as the reference does
template <class Model, int P, class Feed>
__device__ __forceinline__ void exact_residual_ar_feed(const KernelArgs &ka, const ModelArgs &ma, const Feed &feed,
    const ArMoments<P> &mo, const double (&m)[P], double &kk00, double &kk10, double &kk20)
{
    const int T = ka.cfg.n_times;
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
    double sum_all = 0, cross = 0, k_first = 0, k_prev = 0;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
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
        const double k = y_cur - g + Jd;
        sum_all += k * k;
        cross += k_prev * k;
        if (t == 0)
            k_first = k;
        k_prev = k;
    };
    feed.run(step);
    kk00 = sum_all - k_first * k_first;
    kk20 = sum_all - k_prev * k_prev;
    kk10 = -2 * cross;
}
template <class Model, int P, typename RAW>
__device__ __forceinline__ void exact_residual_ar(const KernelArgs &ka, const ModelArgs &ma, const RAW *lane_tile,
    const ArMoments<P> &mo, const double (&m)[P], double &kk00, double &kk10, double &kk20)
{
    exact_residual_ar_feed<Model, P>(ka, ma, ArTileFeed<RAW>{ lane_tile, ka.cfg.n_times }, mo, m, kk00, kk10, kk20);
}

// All residual-dependent scalars of one iteration
struct ArForms
{
    double kk[3]; // k'M00k, k'M10k, k'M20k
    double tr[3]; // tr(Sigma J'M J) for the three M
};

template <class Model, int P, class Feed>
__device__ __forceinline__ void ar_residuals_feed(const KernelArgs &ka, const ModelArgs &ma, const Feed &feed, const VoxelState<P> &st,
    const ArMoments<P> &mo, ArForms &f)
{
    bool lost = false;
#pragma unroll
    for (int w = 0; w < 3; w++)
    {
        bool l;
        ar_forms<P>(st, mo, w, ka.residual_tol, f.kk[w], f.tr[w], l);
        lost |= l;
    }
    const int mode = ka.residual_mode;
    const bool want = (mode == 1) || (mode == 0 && lost);
    if (__any(want))
    {
        double e00, e10, e20;
        exact_residual_ar_feed<Model, P>(ka, ma, feed, mo, st.m, e00, e10, e20);
        if (want)
        {
            f.kk[0] = e00;
            f.kk[1] = e10;
            f.kk[2] = e20;
        }
    }
    else if (mode == 2)
    {
        f.kk[0] = (f.kk[0] > 0.0) ? f.kk[0] : ((f.kk[0] <= 0.0) ? 0.0 : f.kk[0]);
        f.kk[2] = (f.kk[2] > 0.0) ? f.kk[2] : ((f.kk[2] <= 0.0) ? 0.0 : f.kk[2]);
    }
}

template <class Model, int P, typename RAW>
__device__ __forceinline__ void ar_residuals(const KernelArgs &ka, const ModelArgs &ma, const RAW *lane_tile, const VoxelState<P> &st,
    const ArMoments<P> &mo, ArForms &f)
{
    ar_residuals_feed<Model, P>(ka, ma, ArTileFeed<RAW>{ lane_tile, ka.cfg.n_times }, st, mo, f);
}

// With the linearisation centred on the current mean (d = 0): k = r
template <int P>
__device__ __forceinline__ void ar_residuals_at_centre(const VoxelState<P> &st, const ArMoments<P> &mo, ArForms &f)
{
#pragma unroll
    for (int w = 0; w < 3; w++)
    {
        f.kk[w] = ar_rMr<P>(mo, w);
        double tr = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
#pragma unroll
            for (int j = 0; j < P; j++)
                tr += st.Sig[tri(i, j)] * ar_JMJ<P>(mo, w, i, j);
        f.tr[w] = tr;
    }
}

// Ar1cNoiseModel::UpdateTheta (noisemodel_ar.cc:558-634); LMalpha is ignored by this model
template <int P>
__device__ __forceinline__ bool update_theta_ar(VoxelState<P> &st, const ArMoments<P> &mo, const ArAlpha &al)
{
    const double phibar = st.b * st.c;
    double JQJ[P * (P + 1) / 2];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
        {
            JQJ[tri(i, j)] = ar_JMJ<P>(mo, 0, i, j) + al.a1 * ar_JMJ<P>(mo, 1, i, j) + al.a2 * ar_JMJ<P>(mo, 2, i, j);
            st.Lam[tri(i, j)] = phibar * JQJ[tri(i, j)] + ((i == j) ? st.pprec[i] : 0.0);
        }
    st.precValid = true;
    st.covValid = false;
    double rhs[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        double s = ar_JMr<P>(mo, 0, i) + al.a1 * ar_JMr<P>(mo, 1, i) + al.a2 * ar_JMr<P>(mo, 2, i);
#pragma unroll
        for (int j = 0; j < P; j++)
            s += JQJ[tri(i, j)] * mo.ml[j];
        rhs[i] = phibar * s + st.pprec[i] * st.pm[i];
    }
    if (!ensure_cov<P>(st))
        return false;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        double s = 0;
#pragma unroll
        for (int j = 0; j < P; j++)
            s += st.Sig[tri(i, j)] * rhs[j];
        st.m[i] = s;
    }
    return true;
}

// Ar1cNoiseModel::UpdateAlpha + UpdatePhi (noisemodel_ar.cc:447-556). Returns a status.
template <int P>
__device__ __forceinline__ int update_noise_ar(const KernelArgs &ka, VoxelState<P> &st, ArAlpha &al, const ArForms &f)
{
    const double phibar = st.b * st.c;
    double var11;
    if (ka.cfg.ar_alpha_given & 1) // (uniform) noise-initial-prior: a full 2 x 2 prior (InputFromMVN, :302-316)
    {
        const double p11 = ka.cfg.ar_alpha_prior_prec[0][0], p12 = ka.cfg.ar_alpha_prior_prec[0][1], p22 = ka.cfg.ar_alpha_prior_prec[1][1];
        const double m01 = ka.cfg.ar_alpha_prior_mean[0], m02 = ka.cfg.ar_alpha_prior_mean[1];
        const double a = p11 + phibar * (f.kk[2] + f.tr[2]); // OpKLJ(M20)
        if (!is_finite(a))
            return FVB_BAD_AR_ALPHA;
        const double det = a * p22 - p12 * p12;
        if (det == 0 || !is_finite(det))
            return FVB_BAD_RESULT;
        const double rdet = 1.0 / det;
        var11 = p22 * rdet;
        al.c11 = var11;
        al.c12 = -p12 * rdet;
        al.c22 = a * rdet;
        if (al.c11 < 0 || al.c22 < 0)
            return FVB_BAD_AR_ALPHA;
        const double t1 = (p11 * m01 + p12 * m02) + -0.5 * phibar * (f.kk[1] + f.tr[1]); // :501-505
        const double t2 = p12 * m01 + p22 * m02;
        al.mean[0] = al.c11 * t1 + al.c12 * t2;
        al.mean[1] = al.c12 * t1 + al.c22 * t2;
    }
    else
    {
        const double prec11 = AR_ALPHA_PRIOR_PREC + phibar * (f.kk[2] + f.tr[2]); // OpKLJ(M20)
        if (!is_finite(prec11))
            return FVB_BAD_AR_ALPHA;
        var11 = 1.0 / prec11;
        const double var22 = 1.0 / AR_ALPHA_PRIOR_PREC;
        if (var11 < 0)
            return FVB_BAD_AR_ALPHA;
        const double tmp1 = -0.5 * phibar * (f.kk[1] + f.tr[1]); // prior means are zero
        al.c11 = var11;
        al.c12 = 0;
        al.c22 = var22;
        al.mean[0] = var11 * tmp1;
        al.mean[1] = 0;
    }
    al.a1 = al.mean[0];
    al.a2 = var11 + al.mean[0] * al.mean[0];
    // UpdatePhi with the new marginal
    const double kQk = f.kk[0] + al.a1 * f.kk[1] + al.a2 * f.kk[2];
    const double trQ = f.tr[0] + al.a1 * f.tr[1] + al.a2 * f.tr[2];
    st.b = 1 / ((kQk + trQ) * 0.5 + 1 / ka.cfg.noise_prior_b[0]);
    st.c = ((double)ka.cfg.n_times - 1) * 0.5 + ka.cfg.noise_prior_c[0];
    return FVB_OK;
}

// Ar1cNoiseModel::CalcFreeEnergy (noisemodel_ar.cc:643-747)
template <int P>
__device__ __forceinline__ bool calc_free_energy_ar(const KernelArgs &ka, VoxelState<P> &st, const ArAlpha &al,
    const ArForms &f, double Fprior, double &F, bool &finite)
{
    bool ok = ensure_prec<P>(st);
    const double T = (double)ka.cfg.n_times;
    const double si = st.b, ci = st.c, siPrior = ka.cfg.noise_prior_b[0], ciPrior = ka.cfg.noise_prior_c[0];
    const double phibar = si * ci;
    const double logdetAlphaPrec = -log(fabs(al.c11 * al.c22 - al.c12 * al.c12));
    const double expectedLogAlphaDist = 0.5 * logdetAlphaPrec - 0.5 * 2 * (LOG_2PI + 1);
    const double expectedLogThetaDist = 0.5 * st.logdetLam - 0.5 * P * (LOG_2PI + 1);
    const double dg = digamma(ci) + log(si);
    const double expectedLogPhiDist = -gammaln(ci) - ci * log(si) - ci + (ci - 1) * dg;
    double parts = dg * ((T - 1) * 0.5 + ciPrior - 1);                                // [0]
    parts += -2 * gammaln(ciPrior) - 2 * ciPrior * log(siPrior) - si * ci / siPrior; // [9]
    parts += -LOG_2PI * (T - 1 + 0.5 * 2 + 0.5 * P);                                 // [1]
    const double kQk = f.kk[0] + al.a1 * f.kk[1] + al.a2 * f.kk[2];
    const double trQ = f.tr[0] + al.a1 * f.tr[1] + al.a2 * f.tr[2];
    parts += -0.5 * phibar * kQk - 0.5 * phibar * trQ;                               // [2]
    double logdetPrior = 0, quad = 0, trSL0 = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(st.pprec[i]));
        const double dm = st.m[i] - st.pm[i];
        quad += dm * st.pprec[i] * dm;
        trSL0 += st.Sig[tri(i, i)] * st.pprec[i];
    }
    parts += 0.5 * logdetPrior;                                                      // [3]
    parts += -0.5 * quad;                                                            // [4]
    parts += -0.5 * trSL0;                                                           // [5]
    if (ka.cfg.ar_alpha_given & 1) // (uniform) the prior of noise-initial-prior (:720-729)
    {
        const double p11 = ka.cfg.ar_alpha_prior_prec[0][0], p12 = ka.cfg.ar_alpha_prior_prec[0][1], p22 = ka.cfg.ar_alpha_prior_prec[1][1];
        const double d1 = al.mean[0] - ka.cfg.ar_alpha_prior_mean[0], d2 = al.mean[1] - ka.cfg.ar_alpha_prior_mean[1];
        parts += 0.5 * log(fabs(p11 * p22 - p12 * p12));                                   // [6]
        parts += -0.5 * (d1 * p11 * d1 + 2 * d1 * p12 * d2 + d2 * p22 * d2);               // [7]
        parts += -0.5 * (al.c11 * p11 + 2 * al.c12 * p12 + al.c22 * p22);                  // [8]
    }
    else
    {
        parts += 0.5 * 2 * log(AR_ALPHA_PRIOR_PREC);                                     // [6]
        parts += -0.5 * AR_ALPHA_PRIOR_PREC * (al.mean[0] * al.mean[0] + al.mean[1] * al.mean[1]); // [7]
        parts += -0.5 * AR_ALPHA_PRIOR_PREC * (al.c11 + al.c22);                         // [8]
    }
    F = -expectedLogAlphaDist - expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior;
    return ok;
}

template <int P>
__device__ __forceinline__ void save_alpha(const KernelArgs &ka, int v, const ArAlpha &al)
{
    const size_t V = (size_t)ka.cfg.n_voxels;
    double *p = ka.save + (size_t)lane_save_rows<P>() * V + v;
    p[0] = al.mean[0];
    p[V] = al.mean[1];
    p[2 * V] = al.c11;
    p[3 * V] = al.c12;
    p[4 * V] = al.c22;
    p[5 * V] = al.a1;
    p[6 * V] = al.a2;
}
template <int P>
__device__ __forceinline__ void restore_alpha(const KernelArgs &ka, int v, ArAlpha &al)
{
    const size_t V = (size_t)ka.cfg.n_voxels;
    const double *p = ka.save + (size_t)lane_save_rows<P>() * V + v;
    al.mean[0] = p[0];
    al.mean[1] = p[V];
    al.c11 = p[2 * V];
    al.c12 = p[3 * V];
    al.c22 = p[4 * V];
    al.a1 = p[5 * V];
    al.a2 = p[6 * V];
}

// FEED: FEED_TILES_F32 / FEED_TILES_F64 (vb_lane_kernel.h) - the series is always read from the tiled copy
// (AR noise rejects masked timepoints, noisemodel_ar.cc:351-355, so there is no other case)
template <class Model, int P, bool NEEDF, int FEED>
__global__ __launch_bounds__(64, lane_waves<P>()) void vb_lane_ar_kernel(const KernelArgs ka)
{
    constexpr int PT = P * (P + 1) / 2;
    typedef typename FeedTraits<FEED>::raw RAW;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    if (v >= ka.cfg.n_voxels)
        return;
    const RAW *lane_tile = (const RAW *)ka.tiles + (size_t)blockIdx.x * Tile<RAW>::block_elems(T) + (size_t)threadIdx.x * Tile<RAW>::G;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    VoxelState<P> st;
    ArMoments<P> mo;
    ArAlpha al;
    int status = FVB_OK;
    constexpr int n = P + 3;
    constexpr int nCov = n * (n + 1) / 2;

    // ---- Vb::SetupPerVoxelDists (inference_vb.cc:207-247) ----
    if (ka.cfg.init_mvn)
    {
        // MVNDist::Load, GetSubmatrix, Ar1cParams::InputFromMVN (noisemodel_ar.cc:302-316)
        const double *src = ka.cfg.init_mvn + v;
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = src[(size_t)i * V];
#pragma unroll
        for (int i = 0; i < P; i++)
            st.m[i] = src[(size_t)(nCov + i) * V];
        al.mean[0] = src[(size_t)(nCov + P) * V];
        al.mean[1] = src[(size_t)(nCov + P + 1) * V];
        al.c11 = src[(size_t)tri(P, P) * V];
        al.c12 = src[(size_t)tri(P + 1, P) * V];
        al.c22 = src[(size_t)tri(P + 1, P + 1) * V];
        const double nm = src[(size_t)(nCov + P + 2) * V];
        const double nv = src[(size_t)tri(P + 2, P + 2) * V];
        st.b = nv / nm;
        st.c = nm / st.b;
    }
    else
    {
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
            constexpr int G = Tile<RAW>::G;
            double data_max = (double)lane_tile[0];
#pragma nounroll
            for (int t = 1; t < T; t++)
            {
                const double y = (double)lane_tile[(size_t)(t / G) * 64 * G + (t % G)];
                data_max = (y > data_max) ? y : data_max;
            }
            Model::init_posterior(ma, data_max, st.m);
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            const int tr = ka.cfg.transform[i];
            st.m[i] = to_fabber(tr, st.m[i]);
            st.Sig[tri(i, i)] = to_fabber_var(tr, st.Sig[tri(i, i)]);
        }
        // Ar1cNoiseModel::HardcodedInitialDists (noisemodel_ar.cc:379-403)
        ar_initial_alpha(ka, al);
        st.b = ka.cfg.noise_post_b[0];
        st.c = ka.cfg.noise_post_c[0];
    }
    st.covValid = true;
    st.precValid = false;
    st.logdetLam = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        st.pm[i] = 0;
        st.pprec[i] = 1;
    }
    // Ar1cNoiseModel::Precalculate (noisemodel_ar.cc:749-769)
    al.a1 = al.mean[0];
    al.a2 = al.c11 + al.mean[0] * al.mean[0];
    st.c = ka.cfg.noise_prior_c[0] + ((double)T - 1) * 0.5;

    double F = 1234.5678;
    double Fprior = 0;
    int it = 0;
    int hist_len = 0;
    bool setup_failed = false;

    status = recentre_ar<Model, P, RAW>(ka, ma, lane_tile, st.m, mo, true);
    if (status != FVB_OK)
        setup_failed = true;

    if (status == FVB_OK)
    {
        ConvState conv;
        conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
        conv_reset(conv);
        if (ka.save)
        {
            save_state<P>(ka, v, st);
            save_alpha<P>(ka, v, al);
        }
        bool stop = false;
#define FVB_EVAL_F_AR(FORMS)                                                                                 \
    {                                                                                                        \
        double Fn_;                                                                                          \
        bool fin_ = true;                                                                                    \
        if (!calc_free_energy_ar<P>(ka, st, al, (FORMS), Fprior, Fn_, fin_))                                 \
        {                                                                                                    \
            status = FVB_BAD_RESULT;                                                                         \
            break;                                                                                           \
        }                                                                                                    \
        if (!fin_)                                                                                           \
        {                                                                                                    \
            status = FVB_BAD_FREE_ENERGY;                                                                    \
            break;                                                                                           \
        }                                                                                                    \
        F = Fn_;                                                                                             \
    }
        do
        {
            if (ka.save && conv_need_save(conv))
            {
                save_state<P>(ka, v, st);
                save_alpha<P>(ka, v, al);
            }
            if (!apply_priors<P, NEEDF>(ka, v, it, st, Fprior))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            ArForms f;
            if (NEEDF) // "before"
            {
                if (!ensure_cov<P>(st))
                {
                    status = FVB_BAD_RESULT;
                    break;
                }
                ar_residuals_at_centre<P>(st, mo, f);
                FVB_EVAL_F_AR(f)
            }
            if (!update_theta_ar<P>(st, mo, al))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            ar_residuals<Model, P, RAW>(ka, ma, lane_tile, st, mo, f);
            if (NEEDF) // "theta"
                FVB_EVAL_F_AR(f)
            status = update_noise_ar<P>(ka, st, al, f);
            if (status != FVB_OK)
                break;
            if (NEEDF) // "phi"
                FVB_EVAL_F_AR(f)
            status = recentre_ar<Model, P, RAW>(ka, ma, lane_tile, st.m, mo);
            if (status != FVB_OK)
                break;
            if (NEEDF) // "lin"
            {
                ar_residuals_at_centre<P>(st, mo, f);
                FVB_EVAL_F_AR(f)
            }
            if (ka.out.f_history && hist_len < ka.cfg.f_history_rows)
                ka.out.f_history[(size_t)hist_len * V + v] = F;
            hist_len++;
            ++it;
            stop = conv_test(conv, F);
        } while (!stop);

        if (status == FVB_OK)
        {
            if (ka.save && conv_need_save(conv))
            {
                save_state<P>(ka, v, st);
                save_alpha<P>(ka, v, al);
            }
            if (ka.save && conv_need_revert(conv))
            {
                restore_state<P>(ka, v, st);
                restore_alpha<P>(ka, v, al);
                status = recentre_ar<Model, P, RAW>(ka, ma, lane_tile, st.m, mo);
                if (status == FVB_OK && NEEDF)
                {
                    do
                    {
                        if (!ensure_cov<P>(st))
                        {
                            status = FVB_BAD_RESULT;
                            break;
                        }
                        ArForms f;
                        ar_residuals_at_centre<P>(st, mo, f);
                        FVB_EVAL_F_AR(f)
                    } while (false);
                }
            }
        }
#undef FVB_EVAL_F_AR
    }

    // ---- result MVN: fwd_post (+) Ar1cParams::OutputAsMVN (alpha (+) phi), packed ----
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
#pragma unroll
        for (int i = 0; i < PT; i++)
            dst[(size_t)i * V] = st.Sig[i];
#pragma unroll
        for (int r = P; r < n; r++)
#pragma unroll
            for (int c = 0; c <= r; c++)
                dst[(size_t)tri(r, c) * V] = 0.0;
        dst[(size_t)tri(P, P) * V] = al.c11;
        dst[(size_t)tri(P + 1, P) * V] = al.c12;
        dst[(size_t)tri(P + 1, P + 1) * V] = al.c22;
        dst[(size_t)tri(P + 2, P + 2) * V] = st.b * st.b * st.c;
#pragma unroll
        for (int i = 0; i < P; i++)
            dst[(size_t)(nCov + i) * V] = st.m[i];
        dst[(size_t)(nCov + P) * V] = al.mean[0];
        dst[(size_t)(nCov + P + 1) * V] = al.mean[1];
        dst[(size_t)(nCov + P + 2) * V] = st.b * st.c;
        dst[(size_t)(nCov + n) * V] = 1.0;
    }
    if (ka.out.f_history && hist_len < ka.cfg.f_history_rows)
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
