/*
 * vb_wave_kernel.h - voxelwise VB, one WAVEFRONT per voxel, Jacobian staged in LDS.
 *
 * The general form of the hot path: any parameter count up to FVB_MAX_PARAMS, any white-noise
 * pattern (several noise precisions phi_i, masked timepoints), any built-in model through
 * eval_model_runtime. It is also the mapping that keeps the chip busy when there are few voxels
 * (a 64 x 64 slice is 64 waves for the lane kernel, 4096 here).
 *
 *   - the 64 lanes split the TIMEPOINTS: lane l owns t = l, l + 64, ... for the 2P + 1 model
 *     evaluations of the central-difference Jacobian (fwdmodel_linear.cc:126-182); J [T][P],
 *     g(ml) and the data live in LDS for the whole voxel, so the residual
 *     k = y - g(ml) + J (ml - m) is formed directly (noisemodel_white.cc:235) - no moment
 *     cancellation, no second pass;
 *   - the contractions over t (J'Q_iJ, J'Q_i r, r'Q_i r, k'Q_i k) put one OUTPUT ENTRY on each
 *     lane, which runs over t in order (from P = 7 up the entries outnumber the lanes); with few
 *     entries the spare lanes take contiguous chunks of the t range and the chunk sums are added
 *     in chunk order - a fixed association, so results are reproducible run to run;
 *   - P x P work (eq 19-20, the inverse, traces) puts one matrix ENTRY on each lane; the inverse
 *     is the symmetric sweep operator in LDS, whose pivots are the D of the lane kernel's LDL^T;
 *   - scalars (noise posterior, free energy, convergence state machine) are computed redundantly
 *     by every lane from the same LDS values, so control flow is wave-uniform by construction.
 *
 * Reference path per voxel: Vb::SetupPerVoxelDists (inference_vb.cc:207-247) and the body of
 * Vb::DoCalculationsVoxelwise (inference_vb.cc:423-571), WhiteNoiseModel (noisemodel_white.cc).
 */
#pragma once

#include "vb_lane_kernel.h"

#include <string>

namespace fvb
{
// the per-parameter entries of the configuration: the fixed arrays, or the table of a problem with more than
// FVB_MAX_PARAMS parameters (fvb_config.params_ext: this kernel is the one that takes those)
#define FVB_KPARAM(ka, field, k) ((ka).cfg.params_ext ? (ka).cfg.params_ext->field[k] : (ka).cfg.field[k])

// LDS layout in doubles (followed by T int32 for the phi index of each timepoint)
struct WaveLayout
{
    int T, P, N, Ps, PT, PP;
    int y, gl, r, k, J, pv, rden, A, u, s, kq, trs, cnt, b, c, m, ml, pm, pprec, rhs, Lam, Sig, W, W2;
    int sv_m, sv_Lam, sv_Sig, sv_pm, sv_pprec, sv_b, sv_c;
    int part; // 64 partial sums of the chunked contractions
    // AR(1) noise (vb_wave_ar_kernel.h): z = y - g + J ml, X J and J Sigma [T][Ps], the band of the
    // current weighting matrix [T][7] (offsets -3..3, noisemodel_ar.cc:23)
    int z, XJ, JS, band;
    int n_doubles;
    size_t bytes;
};

FVB_HD WaveLayout wave_layout(int T, int P, int N, bool ar = false)
{
    WaveLayout L;
    L.T = T;
    L.P = P;
    L.N = N;
    L.Ps = P | 1; // odd row stride: lanes reading one column of consecutive rows hit distinct banks
    L.PT = P * (P + 1) / 2;
    L.PP = P * P;
    int o = 0;
#define FVB_WL(field, n)                                                                                     \
    L.field = o;                                                                                             \
    o += (n);
    FVB_WL(y, T)
    FVB_WL(gl, T)
    FVB_WL(r, T)
    FVB_WL(k, T)
    FVB_WL(J, T * L.Ps)
    FVB_WL(pv, (2 * P + 1) * P)
    FVB_WL(rden, P)
    FVB_WL(A, N * L.PT)
    FVB_WL(u, N * P)
    FVB_WL(s, N)
    FVB_WL(kq, N)
    FVB_WL(trs, N)
    FVB_WL(cnt, N)
    FVB_WL(b, N)
    FVB_WL(c, N)
    FVB_WL(m, P)
    FVB_WL(ml, P)
    FVB_WL(pm, P)
    FVB_WL(pprec, P)
    FVB_WL(rhs, P)
    FVB_WL(Lam, L.PP)
    FVB_WL(Sig, L.PP)
    FVB_WL(W, L.PP)
    FVB_WL(W2, L.PP)
    FVB_WL(sv_m, P)
    FVB_WL(sv_Lam, L.PP)
    FVB_WL(sv_Sig, L.PP)
    FVB_WL(sv_pm, P)
    FVB_WL(sv_pprec, P)
    FVB_WL(sv_b, N)
    FVB_WL(sv_c, N)
    FVB_WL(part, 64)
    FVB_WL(z, ar ? T : 0)
    FVB_WL(XJ, ar ? T * L.Ps : 0)
    FVB_WL(JS, ar ? T * L.Ps : 0)
    FVB_WL(band, ar ? 7 * T : 0)
#undef FVB_WL
    L.n_doubles = o;
    L.bytes = sizeof(double) * (size_t)o + sizeof(int32_t) * (size_t)T;
    return L;
}

#if defined(__HIPCC__)

int launch_wave_kernel(const KernelArgs &ka, hipStream_t stream, std::string &err);

struct WaveCtx
{
    WaveLayout L;
    double *sh;    // LDS doubles
    int32_t *phi;  // LDS [T]: noise index of each timepoint, -1 = masked
    int lane, v;
    size_t V;
    bool precValid, covValid;
    double logdetLam;
    bool sv_prec;
    // Host-evaluated models (vb_hostmodel.h): the linearisation about the current means as the
    // host computed it - g [T] followed by J [T][P] - instead of the device model bodies.
    const double *lin;
};

#define FVB_WAVE_FOR(idx, n) for (int idx = cx.lane; idx < (n); idx += 64)

__device__ __forceinline__ void wave_sync()
{
    __syncthreads(); // one wave per workgroup: orders the LDS traffic of the 64 lanes
}

// Inverse of a symmetric matrix by sweeping every pivot (Goodnight 1979): after sweeping k,
// W_kk = -1/d, W_ik = W_ik / d, W_ij -= W_ik W_kj / d; all swept, W = -A^-1. The pivots d are the
// D of LDL^T, so log|det|, the sign and the "exactly zero pivot = singular" rule agree with
// ldl_inverse (vb_math.h). src and dst are full P x P LDS matrices.
__device__ __forceinline__ bool wave_sweep_inverse(
    WaveCtx &cx, const double *src, double *dst, double jitter, double &logabs, int &sign)
{
    const int P = cx.L.P, PP = cx.L.PP;
    FVB_WAVE_FOR(e, PP)
    dst[e] = src[e] + ((e / P == e % P) ? jitter : 0.0);
    wave_sync();
    bool ok = true;
    logabs = 0;
    sign = 1;
    for (int k = 0; k < P; k++)
    {
        const double d = dst[k * P + k];
        if (d == 0.0)
            ok = false;
        if (d < 0)
            sign = -sign;
        logabs += log(fabs(d));
        const double rd = 1.0 / d;
        FVB_WAVE_FOR(e, PP)
        {
            const int i = e / P, j = e % P;
            if (i != k && j != k)
                dst[e] -= dst[i * P + k] * dst[k * P + j] * rd;
        }
        wave_sync();
        FVB_WAVE_FOR(i, P)
        {
            if (i != k)
            {
                const double t = dst[i * P + k] * rd;
                dst[i * P + k] = t;
                dst[k * P + i] = t;
            }
            else
                dst[k * P + k] = -rd;
        }
        wave_sync();
    }
    FVB_WAVE_FOR(e, PP)
    dst[e] = -dst[e];
    wave_sync();
    return ok;
}

// MVNDist::GetCovariance / GetPrecisions semantics (dist_mvn.cc:197-265), as mvn_invert
__device__ __forceinline__ bool wave_mvn_invert(WaveCtx &cx, const double *src, double *dst, double &logabs)
{
    int sign;
    if (wave_sweep_inverse(cx, src, dst, 0.0, logabs, sign))
        return true;
    double l2;
    return wave_sweep_inverse(cx, src, dst, 1e-10, l2, sign);
}

__device__ __forceinline__ bool wave_ensure_cov(WaveCtx &cx)
{
    if (cx.covValid)
        return true;
    double logabs;
    const bool ok = wave_mvn_invert(cx, cx.sh + cx.L.Lam, cx.sh + cx.L.Sig, logabs);
    cx.logdetLam = logabs;
    cx.covValid = true;
    return ok;
}

__device__ __forceinline__ bool wave_ensure_prec(WaveCtx &cx)
{
    if (cx.precValid)
        return true;
    double logabs;
    const bool ok = wave_mvn_invert(cx, cx.sh + cx.L.Sig, cx.sh + cx.L.Lam, logabs);
    cx.logdetLam = -logabs;
    cx.precValid = true;
    return ok;
}

// LinearizedFwdModel::ReCentre about the current means (fwdmodel_linear.cc:126-182) followed by
// the per-phi moments A_i = J'Q_iJ, u_i = J'Q_i r, s_i = r'Q_i r with r = y - g(ml).
// moments = false: only g, J and r (the AR kernel forms its own contractions)
__device__ __forceinline__ int wave_recentre(const KernelArgs &ka, const ModelArgs &ma, WaveCtx &cx, bool moments = true)
{
    const WaveLayout &L = cx.L;
    const int T = L.T, P = L.P, N = L.N, Ps = L.Ps, PT = L.PT;
    double *sh = cx.sh;
    bool bad_offset = false, bad_jac = false;
    if (cx.lin)
    {
        FVB_WAVE_FOR(i, P)
        sh[L.ml + i] = sh[L.m + i];
        FVB_WAVE_FOR(t, T)
        {
            const double g = cx.lin[t];
            sh[L.gl + t] = g;
            bad_offset |= !is_finite(g);
            for (int i = 0; i < P; i++)
            {
                const double Jti = cx.lin[T + t * P + i];
                sh[L.J + t * Ps + i] = Jti;
                bad_jac |= !is_finite(Jti);
            }
            sh[L.r + t] = sh[L.y + t] - g;
        }
    }
    else
    {
    FVB_WAVE_FOR(i, P)
    {
        const int tr = FVB_KPARAM(ka, transform, i);
        const double centre = sh[L.m + i];
        double delta = centre * 1e-5; // fwdmodel_linear.cc:157-161
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = centre + delta;
        const double c3 = centre - delta;
        const double tp = to_model(tr, centre); // fwdmodel.cc:375-379
        const double tp2 = to_model(tr, c2);
        const double tp3 = to_model(tr, c3);
        sh[L.rden + i] = 1.0 / (c2 - c3);
        sh[L.ml + i] = centre;
        sh[L.pv + i] = tp;
        for (int j = 0; j < P; j++) // parameter vectors of the 2P perturbed evaluations
        {
            sh[L.pv + (1 + 2 * j) * P + i] = (i == j) ? tp2 : tp;
            sh[L.pv + (2 + 2 * j) * P + i] = (i == j) ? tp3 : tp;
        }
    }
    wave_sync();
    FVB_WAVE_FOR(t, T)
    {
        const double g = eval_model_runtime(ka.cfg.model, ma, P, t, sh + L.pv);
        sh[L.gl + t] = g;
        bad_offset |= !is_finite(g);
        for (int i = 0; i < P; i++)
        {
            FVB_MODEL_FP
            const double f2 = eval_model_runtime(ka.cfg.model, ma, P, t, sh + L.pv + (1 + 2 * i) * P);
            const double f3 = eval_model_runtime(ka.cfg.model, ma, P, t, sh + L.pv + (2 + 2 * i) * P);
            const double Jti = (f2 - f3) * sh[L.rden + i];
            sh[L.J + t * Ps + i] = Jti;
            bad_jac |= !is_finite(Jti);
        }
        sh[L.r + t] = sh[L.y + t] - g;
    }
    }
    wave_sync();
    if (!moments)
    {
        const bool any_offset = __any(bad_offset), any_jac = __any(bad_jac);
        return any_offset ? FVB_BAD_OFFSET : (any_jac ? FVB_BAD_JACOBIAN : FVB_OK);
    }
    // One output entry per lane; with fewer than 33 entries the spare lanes split each entry's
    // t range into C chunks whose partial sums are added in chunk order (fixed, so the result is
    // reproducible run to run).
    const int per_phi = PT + P + 1;
    const int E = N * per_phi;
    const int C = (E <= 32) ? 64 / E : 1;
    const int chunk = (T + C - 1) / C;
    for (int slot = cx.lane; slot < E * C; slot += 64)
    {
        const int e = slot / C, t0 = (slot % C) * chunk;
        const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
        const int phi = e / per_phi, q = e % per_phi;
        double acc = 0;
        int dst;
        if (q < PT)
        {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= q)
                a++;
            const int b = q - a * (a + 1) / 2;
#pragma unroll 4
            for (int t = t0; t < t1; t++)
                acc += (cx.phi[t] == phi) ? sh[L.J + t * Ps + a] * sh[L.J + t * Ps + b] : 0.0;
            dst = L.A + phi * PT + q;
        }
        else if (q < PT + P)
        {
            const int a = q - PT;
#pragma unroll 4
            for (int t = t0; t < t1; t++)
                acc += (cx.phi[t] == phi) ? sh[L.J + t * Ps + a] * sh[L.r + t] : 0.0;
            dst = L.u + phi * P + a;
        }
        else
        {
#pragma unroll 4
            for (int t = t0; t < t1; t++)
                acc += (cx.phi[t] == phi) ? sh[L.r + t] * sh[L.r + t] : 0.0;
            dst = L.s + phi;
        }
        if (C == 1)
            sh[dst] = acc;
        else
            sh[L.part + slot] = acc;
    }
    if (C > 1)
    {
        wave_sync();
        FVB_WAVE_FOR(e, E)
        {
            const int phi = e / per_phi, q = e % per_phi;
            double acc = 0;
            for (int c = 0; c < C; c++)
                acc += sh[L.part + e * C + c];
            sh[(q < PT) ? (L.A + phi * PT + q) : ((q < PT + P) ? (L.u + phi * P + q - PT) : (L.s + phi))] = acc;
        }
    }
    wave_sync();
    const bool any_offset = __any(bad_offset), any_jac = __any(bad_jac);
    return any_offset ? FVB_BAD_OFFSET : (any_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}

// Prior::ApplyToMVN for every parameter (inference_vb.cc:460-463; priors.cc:108-181)
template <bool NEEDF>
__device__ __forceinline__ bool wave_apply_priors(const KernelArgs &ka, WaveCtx &cx, int it, double &Fprior)
{
    const WaveLayout &L = cx.L;
    const int P = L.P;
    double *sh = cx.sh;
    bool has_ard = false;
    for (int k = 0; k < P; k++)
        has_ard |= (FVB_KPARAM(ka, prior_type, k) == FVB_PRIOR_ARD);
    bool ok = true;
    if (has_ard)
        ok = wave_ensure_cov(cx);
    double fk = 0;
    FVB_WAVE_FOR(k, P)
    {
        const int type = FVB_KPARAM(ka, prior_type, k);
        if (type == FVB_PRIOR_ARD) // priors.cc:150-181
        {
            const double post_mean = sh[L.m + k];
            const double post_cov = sh[L.Sig + k * P + k];
            const double new_cov = post_mean * post_mean + post_cov;
            if (it == 0)
            {
                sh[L.pprec + k] = 1.0 / FVB_KPARAM(ka, prior_var, k);
                sh[L.pm + k] = FVB_KPARAM(ka, prior_mean, k);
            }
            else
                sh[L.pprec + k] = 1.0 / new_cov;
            if (NEEDF)
            {
                const double bb = 2 / new_cov;
                fk = -1.5 * (log(bb) + digamma(0.5)) - 0.5 - gammaln(0.5) - 0.5 * log(bb);
            }
        }
        else if (type == FVB_PRIOR_IMAGE) // priors.cc:133-142
        {
            sh[L.pm + k] = FVB_KPARAM(ka, image_prior, k)[cx.v];
            sh[L.pprec + k] = FVB_KPARAM(ka, prior_prec, k);
        }
        else // priors.cc:108-117
        {
            sh[L.pm + k] = FVB_KPARAM(ka, prior_mean, k);
            sh[L.pprec + k] = FVB_KPARAM(ka, prior_prec, k);
        }
    }
    Fprior = __shfl(fk, P - 1); // the value of the LAST prior ('=' not '+=' in the reference)
    wave_sync();
    return ok;
}

// WhiteNoiseModel::UpdateTheta (noisemodel_white.cc:275-363)
__device__ __forceinline__ bool wave_update_theta(WaveCtx &cx, double alpha)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, N = L.N, PT = L.PT, PP = L.PP;
    double *sh = cx.sh;
    FVB_WAVE_FOR(e, PP)
    {
        const int i = e / P, j = e % P;
        double acc = 0;
        for (int phi = 0; phi < N; phi++)
            acc += (sh[L.b + phi] * sh[L.c + phi]) * sh[L.A + phi * PT + tri(i, j)];
        sh[L.Lam + e] = acc + ((i == j) ? sh[L.pprec + i] : 0.0); // eq (19)
    }
    cx.precValid = true;
    cx.covValid = false;
    if (alpha <= 0.0)
    {
        FVB_WAVE_FOR(i, P)
        {
            double acc = 0;
            for (int phi = 0; phi < N; phi++)
            {
                double aml = 0;
                for (int j = 0; j < P; j++)
                    aml += sh[L.A + phi * PT + tri(i, j)] * sh[L.ml + j];
                acc += (sh[L.b + phi] * sh[L.c + phi]) * (sh[L.u + phi * P + i] + aml);
            }
            sh[L.rhs + i] = acc + sh[L.pprec + i] * sh[L.pm + i];
        }
        wave_sync();
        if (!wave_ensure_cov(cx))
            return false;
        FVB_WAVE_FOR(i, P)
        {
            double s = 0;
            for (int j = 0; j < P; j++)
                s += sh[L.Sig + i * P + j] * sh[L.rhs + j];
            sh[L.m + i] = s; // eq (20)
        }
        wave_sync();
    }
    else
    {
        // Levenberg-Marquardt form, noisemodel_white.cc:330-350
        FVB_WAVE_FOR(i, P)
        {
            double acc = 0;
            for (int phi = 0; phi < N; phi++)
                acc += (sh[L.b + phi] * sh[L.c + phi]) * sh[L.u + phi * P + i];
            sh[L.rhs + i] = acc + sh[L.pprec + i] * sh[L.pm + i] - sh[L.pprec + i] * sh[L.ml + i];
        }
        wave_sync();
        FVB_WAVE_FOR(e, PP)
        sh[L.W + e] = sh[L.Lam + e] + ((e / P == e % P) ? alpha * sh[L.Lam + e] : 0.0);
        wave_sync();
        double la;
        int sg;
        if (wave_sweep_inverse(cx, sh + L.W, sh + L.W2, 0.0, la, sg))
        {
            FVB_WAVE_FOR(i, P)
            {
                double s = 0;
                for (int j = 0; j < P; j++)
                    s += sh[L.W2 + i * P + j] * sh[L.rhs + j];
                sh[L.m + i] = sh[L.ml + i] + s;
            }
        } // singular: warn and keep the means (:347-350)
        wave_sync();
    }
    return true;
}

// k'Q_i k and tr(Sigma J'Q_i J) per phi. at_centre: the means ARE the linearisation centre
// (straight after a re-centre), so k = r and k'Q_i k = s_i.
__device__ __forceinline__ void wave_residuals(WaveCtx &cx, bool at_centre)
{
    const WaveLayout &L = cx.L;
    const int T = L.T, P = L.P, N = L.N, Ps = L.Ps, PT = L.PT;
    double *sh = cx.sh;
    if (!at_centre)
    {
        FVB_WAVE_FOR(t, T)
        {
            double Jd = 0;
            for (int i = 0; i < P; i++)
                Jd += sh[L.J + t * Ps + i] * (sh[L.ml + i] - sh[L.m + i]);
            sh[L.k + t] = sh[L.r + t] + Jd; // noisemodel_white.cc:235
        }
        wave_sync();
    }
    // entries: (k'Q_i k, tr(Sigma A_i)) per phi, each split into C chunks as in wave_recentre
    const int E = 2 * N; // <= 16
    const int C = 64 / E;
    for (int slot = cx.lane; slot < E * C; slot += 64)
    {
        const int e = slot / C, c = slot % C, phi = e >> 1;
        double acc = 0;
        if ((e & 1) == 0)
        {
            if (at_centre)
                acc = (c == 0) ? sh[L.s + phi] : 0.0;
            else
            {
                const int chunk = (T + C - 1) / C, t0 = c * chunk;
                const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
#pragma unroll 4
                for (int t = t0; t < t1; t++)
                    acc += (cx.phi[t] == phi) ? sh[L.k + t] * sh[L.k + t] : 0.0;
            }
        }
        else
        {
            const int PP = L.PP, chunk = (PP + C - 1) / C, q0 = c * chunk;
            const int q1 = (q0 + chunk < PP) ? q0 + chunk : PP;
            for (int q = q0; q < q1; q++)
                acc += sh[L.Sig + q] * sh[L.A + phi * PT + tri(q / P, q % P)];
        }
        sh[L.part + slot] = acc;
    }
    wave_sync();
    FVB_WAVE_FOR(e, E)
    {
        double acc = 0;
        for (int c = 0; c < C; c++)
            acc += sh[L.part + e * C + c];
        sh[((e & 1) == 0 ? L.kq : L.trs) + (e >> 1)] = acc;
    }
    wave_sync();
}

// WhiteNoiseModel::UpdateNoise (noisemodel_white.cc:228-273)
__device__ __forceinline__ void wave_update_noise(const KernelArgs &ka, WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    double *sh = cx.sh;
    FVB_WAVE_FOR(phi, L.N)
    {
        const double tmp = sh[L.kq + phi] + sh[L.trs + phi];
        double b = 1 / (tmp * 0.5 + 1 / ka.cfg.noise_prior_b[phi]);              // eq (22)
        const double c = (sh[L.cnt + phi] - 1) * 0.5 + ka.cfg.noise_prior_c[phi]; // eq (21)
        if (ka.cfg.locked_noise_stdev > 0)
            b = 1 / c / ka.cfg.locked_noise_stdev / ka.cfg.locked_noise_stdev;
        sh[L.b + phi] = b;
        sh[L.c + phi] = c;
    }
    wave_sync();
}

// WhiteNoiseModel::CalcFreeEnergy (noisemodel_white.cc:365-454) from kq / trs; every lane
// computes the same scalar.
__device__ __forceinline__ bool wave_free_energy(
    const KernelArgs &ka, WaveCtx &cx, double Fprior, double &F, bool &finite)
{
    const bool ok = wave_ensure_prec(cx);
    const WaveLayout &L = cx.L;
    const int P = L.P, N = L.N;
    const double *sh = cx.sh;
    const double nq = (double)ka.n_unmasked;
    const double expectedLogThetaDist = 0.5 * cx.logdetLam - 0.5 * P * (LOG_2PI + 1);
    double expectedLogPhiDist = 0, p0 = 0, p9 = 0, p2 = 0;
    for (int phi = 0; phi < N; phi++)
    {
        const double si = sh[L.b + phi], ci = sh[L.c + phi];
        const double siPrior = ka.cfg.noise_prior_b[phi], ciPrior = ka.cfg.noise_prior_c[phi];
        const double dg = digamma(ci) + log(si);
        expectedLogPhiDist += -gammaln(ci) - ci * log(si) - ci + (ci - 1) * dg;
        p0 += dg * (sh[L.cnt + phi] * 0.5 + ciPrior - 1);
        p9 += -gammaln(ciPrior) - ciPrior * log(siPrior) - si * ci / siPrior;
        p2 += -0.5 * si * ci * sh[L.kq + phi] - 0.5 * sh[L.trs + phi];
    }
    double parts = p0;
    parts += p9;
    parts += p2;
    double logdetPrior = 0, quad = 0, trSL0 = 0;
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(sh[L.pprec + i]));
        const double dm = sh[L.m + i] - sh[L.pm + i];
        quad += dm * sh[L.pprec + i] * dm;
        trSL0 += sh[L.Sig + i * P + i] * sh[L.pprec + i];
    }
    parts += 0.5 * logdetPrior - 0.5 * nq * LOG_2PI - 0.5 * P * LOG_2PI;
    parts += -0.5 * quad;
    parts += -0.5 * trSL0;
    F = -expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior; // Vb::CalculateF, inference_vb.cc:310
    return ok;
}

__device__ __forceinline__ void wave_copy(WaveCtx &cx, int dst, int src, int n)
{
    FVB_WAVE_FOR(i, n)
    cx.sh[dst + i] = cx.sh[src + i];
}

__device__ __forceinline__ void wave_save_state(WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    wave_copy(cx, L.sv_m, L.m, L.P);
    wave_copy(cx, L.sv_Lam, L.Lam, L.PP);
    wave_copy(cx, L.sv_Sig, L.Sig, L.PP);
    wave_copy(cx, L.sv_pm, L.pm, L.P);
    wave_copy(cx, L.sv_pprec, L.pprec, L.P);
    wave_copy(cx, L.sv_b, L.b, L.N);
    wave_copy(cx, L.sv_c, L.c, L.N);
    cx.sv_prec = cx.precValid;
    wave_sync();
}

__device__ __forceinline__ void wave_restore_state(WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    wave_copy(cx, L.m, L.sv_m, L.P);
    wave_copy(cx, L.Lam, L.sv_Lam, L.PP);
    wave_copy(cx, L.Sig, L.sv_Sig, L.PP);
    wave_copy(cx, L.pm, L.sv_pm, L.P);
    wave_copy(cx, L.pprec, L.sv_pprec, L.P);
    wave_copy(cx, L.b, L.sv_b, L.N);
    wave_copy(cx, L.c, L.sv_c, L.N);
    // whichever representation was valid is kept, the other is re-derived on demand (as MVNDist)
    cx.precValid = cx.sv_prec;
    cx.covValid = !cx.sv_prec;
    wave_sync();
}

template <bool NEEDF>
__global__ __launch_bounds__(64) void vb_wave_kernel(const KernelArgs ka, const WaveLayout L)
{
    extern __shared__ double wave_lds[];
    WaveCtx cx;
    cx.L = L;
    cx.sh = wave_lds;
    cx.phi = (int32_t *)(wave_lds + L.n_doubles);
    cx.lane = threadIdx.x;
    cx.v = blockIdx.x;
    cx.V = (size_t)ka.cfg.n_voxels;
    cx.lin = nullptr;
    const int v = cx.v, T = L.T, P = L.P, N = L.N, PP = L.PP;
    const size_t V = cx.V;
    double *sh = cx.sh;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    // ---- stage the voxel's time series and the noise pattern ----
    FVB_WAVE_FOR(t, T)
    {
        sh[L.y + t] = load_data(ka, (size_t)t * V + v);
        const int idx = ka.cfg.phi_index ? (int)ka.cfg.phi_index[t] : 0;
        cx.phi[t] = (idx == 255) ? -1 : idx;
    }
    FVB_WAVE_FOR(e, PP)
    {
        sh[L.Sig + e] = 0;
        sh[L.Lam + e] = 0;
    }
    wave_sync();
    FVB_WAVE_FOR(phi, N)
    {
        int n = 0;
        for (int t = 0; t < T; t++)
            n += (cx.phi[t] == phi);
        sh[L.cnt + phi] = (double)n;
    }

    // ---- Vb::SetupPerVoxelDists, per-voxel part (inference_vb.cc:207-247) ----
    const int n = P + N;
    const int nCov = n * (n + 1) / 2;
    if (ka.cfg.init_mvn)
    {
        const double *src = ka.cfg.init_mvn + v;
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = src[(size_t)tri(e / P, e % P) * V];
        FVB_WAVE_FOR(i, P)
        sh[L.m + i] = src[(size_t)(nCov + i) * V];
        FVB_WAVE_FOR(phi, N)
        {
            const double nm = src[(size_t)(nCov + P + phi) * V];
            const double nv = src[(size_t)tri(P + phi, P + phi) * V];
            const double b = nv / nm; // GammaDist::SetMeanVariance, dist_gamma.cc:29-33
            sh[L.b + phi] = b;
            sh[L.c + phi] = nm / b;
        }
    }
    else
    {
        double data_max = 0;
        if (ka.cfg.model == FVB_MODEL_EXP) // examples/fwdmodel_exp.cc InitVoxelPosterior
        {
            data_max = sh[L.y];
            for (int t = 1; t < T; t++)
                data_max = (sh[L.y + t] > data_max) ? sh[L.y + t] : data_max;
        }
        FVB_WAVE_FOR(i, P)
        {
            double mean = (FVB_KPARAM(ka, prior_type, i) == FVB_PRIOR_IMAGE) ? FVB_KPARAM(ka, image_prior, i)[v] : FVB_KPARAM(ka, post_mean, i);
            if (ka.cfg.model == FVB_MODEL_EXP && (i % 2) == 0)
                mean = data_max / (P / 2 + i / 2);
            const int tr = FVB_KPARAM(ka, transform, i);
            sh[L.m + i] = to_fabber(tr, mean); // FwdModel::ToFabber, fwdmodel.cc:315-324
            sh[L.Sig + i * P + i] = to_fabber_var(tr, FVB_KPARAM(ka, post_var, i));
        }
        FVB_WAVE_FOR(phi, N)
        {
            sh[L.b + phi] = ka.cfg.noise_post_b[phi];
            sh[L.c + phi] = ka.cfg.noise_post_c[phi];
        }
    }
    FVB_WAVE_FOR(i, P) // fwd_prior = MVNDist(P): zero mean, identity (inference_vb.cc:159)
    {
        sh[L.pm + i] = 0;
        sh[L.pprec + i] = 1;
    }
    cx.covValid = true;
    cx.precValid = false;
    cx.logdetLam = 0;
    cx.sv_prec = false;
    wave_sync();

    double F = 1234.5678; // inference_vb.cc:438
    double Fprior = 0;
    int it = 0, hist_len = 0;
    bool setup_failed = false;
    const bool use_save = ka.cfg.convergence == FVB_CONV_FREDUCE || ka.cfg.convergence == FVB_CONV_TRIALMODE
        || ka.cfg.convergence == FVB_CONV_LM;

    int status = wave_recentre(ka, ma, cx); // inference_vb.cc:235 and :443 share one pass
    if (status != FVB_OK)
        setup_failed = true;

    if (status == FVB_OK)
    {
        ConvState conv;
        conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
        conv_reset(conv);
        if (use_save)
            wave_save_state(cx); // :432-434
        bool stop = false;
#define FVB_WAVE_EVAL_F()                                                                                    \
    {                                                                                                        \
        double Fn_;                                                                                          \
        bool fin_ = true;                                                                                    \
        if (!wave_free_energy(ka, cx, Fprior, Fn_, fin_))                                                    \
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
            if (use_save && conv_need_save(conv)) // :451-458
                wave_save_state(cx);
            if (!wave_apply_priors<NEEDF>(ka, cx, it, Fprior))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            if (NEEDF) // "before" :468
            {
                if (!wave_ensure_cov(cx))
                {
                    status = FVB_BAD_RESULT;
                    break;
                }
                wave_residuals(cx, true);
                FVB_WAVE_EVAL_F()
            }
            if (!wave_update_theta(cx, conv_lm_alpha(conv)) || !wave_ensure_cov(cx)) // :470
            {
                status = FVB_BAD_RESULT;
                break;
            }
            wave_residuals(cx, false);
            if (NEEDF) // "theta" :477
                FVB_WAVE_EVAL_F()
            wave_update_noise(ka, cx); // :479
            if (NEEDF) // "phi" :485
                FVB_WAVE_EVAL_F()
            status = wave_recentre(ka, ma, cx); // :490
            if (status != FVB_OK)
                break;
            if (NEEDF) // "lin" :495
            {
                wave_residuals(cx, true);
                FVB_WAVE_EVAL_F()
            }
            if (cx.lane == 0 && ka.out.f_history && hist_len < ka.cfg.f_history_rows) // :496-497
                ka.out.f_history[(size_t)hist_len * V + v] = F;
            hist_len++;
            ++it;
            stop = conv_test(conv, F);
        } while (!stop);

        if (status == FVB_OK)
        {
            if (use_save && conv_need_save(conv)) // :506-513
                wave_save_state(cx);
            if (use_save && conv_need_revert(conv)) // :516-525
            {
                wave_restore_state(cx);
                status = wave_recentre(ka, ma, cx);
                if (status == FVB_OK && NEEDF)
                {
                    do
                    {
                        if (!wave_ensure_cov(cx))
                        {
                            status = FVB_BAD_RESULT;
                            break;
                        }
                        wave_residuals(cx, true);
                        FVB_WAVE_EVAL_F()
                    } while (false);
                }
            }
        }
#undef FVB_WAVE_EVAL_F
    }

    // ---- result MVN: MVNDist(fwd_post, noise.OutputAsMVN()) packed as MVNDist::Save does
    // (inference_vb.cc:549-550; dist_mvn.cc:57-100,410-429; noisemodel_white.cc:55-68) ----
    if (!wave_ensure_cov(cx))
    {
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = 0;
        if (status == FVB_OK)
            status = FVB_BAD_RESULT;
        wave_sync();
    }
    {
        double *dst = ka.out.mvn + v;
        FVB_WAVE_FOR(e, nCov)
        {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= e)
                i++;
            const int j = e - i * (i + 1) / 2;
            double val = 0;
            if (i < P)
                val = sh[L.Sig + i * P + j];
            else if (i == j)
            {
                const double b = sh[L.b + (i - P)], c = sh[L.c + (i - P)];
                val = b * b * c; // GammaDist::CalcVariance
            }
            dst[(size_t)e * V] = val;
        }
        FVB_WAVE_FOR(i, n)
        dst[(size_t)(nCov + i) * V] = (i < P) ? sh[L.m + i] : sh[L.b + (i - P)] * sh[L.c + (i - P)];
        if (cx.lane == 0)
            dst[(size_t)(nCov + n) * V] = 1.0;
    }
    if (cx.lane == 0)
    {
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
}

#endif // __HIPCC__

} // namespace fvb
