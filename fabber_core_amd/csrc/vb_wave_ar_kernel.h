/*
 * vb_wave_ar_kernel.h - voxelwise VB with the AR(1) noise model in its general form, one
 * WAVEFRONT per voxel: one or two interleaved echoes (num-echoes) and the cross-term variants
 * none / same / dual (2 / 3 / 4 alphas), any parameter count, any built-in model.
 *
 * Reference: Ar1cNoiseModel / Ar1cMatrixCache (noisemodel_ar.cc:83-769) inside the loop of
 * Vb::DoCalculationsVoxelwise (inference_vb.cc:415-576). The lane kernel (vb_lane_ar_kernel.h)
 * covers the one-echo case of the headline configuration at full throughput; this kernel covers
 * everything and is what one echo falls back to for parameter counts without a lane instantiation.
 *
 * The reference keeps dense T x T "alpha matrices" M(n, a12pow, a34pow). Each is ONE line of
 * nTimes - 1 equal entries (+-1) starting at (row, col), stepping by nPhis along the diagonal and
 * reflected (noisemodel_ar.cc:112-181), so
 *   - a weighted sum of them (the marginals Q_n, X = sum_n b_n c_n Q_n) is a band of half-width 3:
 *     it is built in LDS as band[t][-3..3] by running along the lines, and applied to J and to the
 *     residual with 7 multiply-adds per entry;
 *   - OperatorKLJ on a single matrix, k'Mk + tr(Lambda^-1 J'MJ), is a sum along its line of
 *     k_r k_c + (J Sigma)_r . J_c.
 * J, J Sigma and X J live in LDS ([T][P]); lanes split the timepoints; sums over t are formed as
 * 64 lane partials added in lane order (fixed association). The alpha posterior (at most 4 x 4),
 * the phi posteriors and all scalars are computed redundantly by every lane, which keeps the
 * control flow wave-uniform.
 */
#pragma once

#include "vb_wave_kernel.h"

namespace fvb
{
#if defined(__HIPCC__)

// the six (a12pow, a34pow) combinations in a fixed order
__device__ __forceinline__ void ar_combo(int idx, int &a12, int &a34)
{
    const int A12[6] = { 0, 1, 2, 0, 1, 0 }, A34[6] = { 0, 0, 0, 1, 1, 2 };
    a12 = A12[idx];
    a34 = A34[idx];
}

// Ar1cMatrixCache::Update, first half (noisemodel_ar.cc:112-181): the line of M(n, a12pow, a34pow),
// 0-based start, and its value
__device__ __forceinline__ void ar_line(int nPhis, int n /*1-based*/, int a12, int a34, int &row0, int &col0, double &value)
{
    int row, col;
    switch (a12 * 10 + a34)
    {
    case 0:
        row = col = 1 + nPhis;
        break;
    case 10:
        row = 1;
        col = 1 + nPhis;
        break;
    case 20:
        row = col = 1;
        break;
    case 1:
        row = 4;
        col = 3;
        break;
    case 11:
        row = 4;
        col = 1;
        break;
    default: // 02
        row = col = 4;
        break;
    }
    value = (a12 + a34 == 1) ? -1.0 : 1.0;
    if (n == 2)
    {
        row = row - 1 + 2 * (row % 2); // 2n->2n-1, 2n-1->2n
        col = col - 1 + 2 * (col % 2);
    }
    row0 = row - 1;
    col0 = col - 1;
}

// inverse, log|det| and sign of a small symmetric matrix held by every lane (the alpha posterior)
template <int N>
__device__ __forceinline__ bool small_sym_inverse(const double (&a)[N][N], double (&inv)[N][N], double &logabs, int &sign)
{
    bool ok = true;
    logabs = 0;
    sign = 1;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++)
            inv[i][j] = a[i][j];
#pragma unroll
    for (int k = 0; k < N; k++)
    {
        const double d = inv[k][k];
        if (d == 0.0)
            ok = false;
        if (d < 0)
            sign = -sign;
        logabs += log(fabs(d));
        const double rd = 1.0 / d;
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = 0; j < N; j++)
                if (i != k && j != k)
                    inv[i][j] -= inv[i][k] * inv[k][j] * rd;
#pragma unroll
        for (int i = 0; i < N; i++)
            if (i != k)
            {
                inv[i][k] *= rd;
                inv[k][i] = inv[i][k];
            }
        inv[k][k] = -rd;
    }
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++)
            inv[i][j] = -inv[i][j];
    return ok;
}

// Ar1cParams: alpha MVN, marginal weights; the phi posteriors live in LDS (L.b, L.c)
template <int NPHI, int NA>
struct ArState
{
    double am[NA], aprec[NA][NA], acov[NA][NA];
    double w[NPHI][6]; // Q_n = sum_combo w[n][combo] M(n, combo)   (noisemodel_ar.cc:199-222)
};

template <int NPHI, int NA>
__device__ __forceinline__ void ar_update_marginals(ArState<NPHI, NA> &st)
{
#pragma unroll
    for (int n = 1; n <= NPHI; n++)
    {
        double(&wn)[6] = st.w[n - 1];
        wn[0] = 1;
        wn[1] = st.am[n - 1];
        wn[2] = st.acov[n - 1][n - 1] + st.am[n - 1] * st.am[n - 1];
        wn[3] = wn[4] = wn[5] = 0;
        if (NA >= 3)
        {
            const int T = ((NA == 4) ? 2 + n : 3) - 1;
            wn[3] = st.am[T];
            wn[4] = st.acov[n - 1][T] + st.am[n - 1] * st.am[T];
            wn[5] = st.acov[T][T] + st.am[T] * st.am[T];
        }
    }
}

// sum over the 64 lanes, added in lane order; every lane gets the result
__device__ __forceinline__ double wave_sum(WaveCtx &cx, double x)
{
    double *part = cx.sh + cx.L.part;
    part[cx.lane] = x;
    wave_sync();
    double s = 0;
    for (int i = 0; i < 64; i++)
        s += part[i];
    wave_sync();
    return s;
}

// band[t][d + 3] = sum over n, combo of scale[n] w[n][combo] M(n, combo)[t][t + d]
template <int NPHI, int NA>
__device__ __forceinline__ void ar_build_band(WaveCtx &cx, const ArState<NPHI, NA> &st, const double (&scale)[NPHI])
{
    const WaveLayout &L = cx.L;
    const int T = L.T, nT = T / NPHI;
    double *band = cx.sh + L.band;
    FVB_WAVE_FOR(e, 7 * T)
    band[e] = 0;
    wave_sync();
    for (int n = 1; n <= NPHI; n++)
        for (int combo = 0; combo < ((NA >= 3) ? 6 : 3); combo++)
        {
            const double wt = scale[n - 1] * st.w[n - 1][combo];
            if (wt == 0.0) // wave-uniform
                continue;
            int a12, a34, row0, col0;
            double value;
            ar_combo(combo, a12, a34);
            ar_line(NPHI, n, a12, a34, row0, col0, value);
            // within one line every count touches its own band entries
            FVB_WAVE_FOR(count, nT - 1)
            {
                const int r = row0 + count * NPHI, c = col0 + count * NPHI;
                band[r * 7 + (c - r) + 3] += wt * value;
                if (r != c)
                    band[c * 7 + (r - c) + 3] += wt * value;
            }
            wave_sync();
        }
}

// dst[t][i] = sum_d band[t][d] src[t + d][i]  for a [T][Ps] matrix
__device__ __forceinline__ void ar_band_times_matrix(WaveCtx &cx, int src, int dst)
{
    const WaveLayout &L = cx.L;
    const int T = L.T, P = L.P, Ps = L.Ps;
    double *sh = cx.sh;
    FVB_WAVE_FOR(t, T)
    {
        for (int i = 0; i < P; i++)
        {
            double acc = 0;
#pragma unroll
            for (int d = -3; d <= 3; d++)
            {
                const int t2 = t + d;
                if (t2 >= 0 && t2 < T)
                    acc += sh[L.band + t * 7 + d + 3] * sh[src + t2 * Ps + i];
            }
            sh[dst + t * Ps + i] = acc;
        }
    }
    wave_sync();
}

// x' B y for the current band and two LDS vectors
__device__ __forceinline__ double ar_band_form(WaveCtx &cx, int x, int y)
{
    const WaveLayout &L = cx.L;
    const int T = L.T;
    const double *sh = cx.sh;
    double acc = 0;
    FVB_WAVE_FOR(t, T)
    {
        double q = 0;
#pragma unroll
        for (int d = -3; d <= 3; d++)
        {
            const int t2 = t + d;
            if (t2 >= 0 && t2 < T)
                q += sh[L.band + t * 7 + d + 3] * sh[y + t2];
        }
        acc += sh[x + t] * q;
    }
    return wave_sum(cx, acc);
}

// sum_t sum_i A[t][i] B[t][i] for two [T][Ps] matrices: tr(Sigma J'BJ) with A = B J, B = J Sigma
__device__ __forceinline__ double ar_matrix_dot(WaveCtx &cx, int a, int b)
{
    const WaveLayout &L = cx.L;
    const double *sh = cx.sh;
    double acc = 0;
    FVB_WAVE_FOR(t, L.T)
    {
        for (int i = 0; i < L.P; i++)
            acc += sh[a + t * L.Ps + i] * sh[b + t * L.Ps + i];
    }
    return wave_sum(cx, acc);
}

// J Sigma -> L.JS (Sigma must be valid)
__device__ __forceinline__ void ar_j_sigma(WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, Ps = L.Ps;
    double *sh = cx.sh;
    FVB_WAVE_FOR(t, L.T)
    {
        for (int i = 0; i < P; i++)
        {
            double acc = 0;
            for (int j = 0; j < P; j++)
                acc += sh[L.J + t * Ps + j] * sh[L.Sig + j * P + i];
            sh[L.JS + t * Ps + i] = acc;
        }
    }
    wave_sync();
}

// k = y - g(ml) + J (ml - m)  (noisemodel_ar.cc:459)
__device__ __forceinline__ void ar_residual(WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    double *sh = cx.sh;
    FVB_WAVE_FOR(t, L.T)
    {
        double Jd = 0;
        for (int i = 0; i < L.P; i++)
            Jd += sh[L.J + t * L.Ps + i] * (sh[L.ml + i] - sh[L.m + i]);
        sh[L.k + t] = sh[L.r + t] + Jd;
    }
    wave_sync();
}

// OperatorKLJ on one alpha matrix (noisemodel_ar.cc:433-445): k'Mk + tr(Lambda^-1 J'MJ); needs
// L.k and L.JS
__device__ __forceinline__ double ar_op_klj(WaveCtx &cx, int nPhis, int n, int a12, int a34)
{
    const WaveLayout &L = cx.L;
    const int nT = L.T / nPhis, P = L.P, Ps = L.Ps;
    const double *sh = cx.sh;
    int row0, col0;
    double value;
    ar_line(nPhis, n, a12, a34, row0, col0, value);
    double acc = 0;
    FVB_WAVE_FOR(count, nT - 1)
    {
        const int r = row0 + count * nPhis, c = col0 + count * nPhis;
        double g = sh[L.k + r] * sh[L.k + c];
        for (int i = 0; i < P; i++)
            g += sh[L.JS + r * Ps + i] * sh[L.J + c * Ps + i];
        acc += g;
    }
    return value * ((row0 != col0) ? 2.0 : 1.0) * wave_sum(cx, acc);
}

// Ar1cNoiseModel::UpdateTheta (noisemodel_ar.cc:558-634)
template <int NPHI, int NA>
__device__ __forceinline__ bool ar_update_theta(WaveCtx &cx, const ArState<NPHI, NA> &st)
{
    const WaveLayout &L = cx.L;
    const int T = L.T, P = L.P, Ps = L.Ps, PT = L.PT, PP = L.PP;
    double *sh = cx.sh;
    double sc[NPHI];
#pragma unroll
    for (int i = 0; i < NPHI; i++)
        sc[i] = sh[L.b + i] * sh[L.c + i];
    ar_build_band<NPHI, NA>(cx, st, sc); // X = sum_n b_n c_n Q_n
    ar_band_times_matrix(cx, L.J, L.XJ);
    FVB_WAVE_FOR(t, T) // z = data - g(ml) + J ml
    {
        double Jm = 0;
        for (int i = 0; i < P; i++)
            Jm += sh[L.J + t * Ps + i] * sh[L.ml + i];
        sh[L.z + t] = sh[L.r + t] + Jm;
    }
    wave_sync();
    // J'XJ (lower triangle) and J'Xz: one output entry per lane, chunks of t for the spare lanes,
    // chunk sums added in chunk order - as wave_recentre
    const int E = PT + P;
    const int C = (E <= 32) ? 64 / E : 1;
    const int chunk = (T + C - 1) / C;
    for (int base = 0; base < E * C; base += 64)
    {
        const int slot = base + cx.lane;
        if (slot < E * C)
        {
            const int e = slot / C, t0 = (slot % C) * chunk;
            const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
            double acc = 0;
            if (e < PT)
            {
                int a = 0;
                while ((a + 1) * (a + 2) / 2 <= e)
                    a++;
                const int b = e - a * (a + 1) / 2;
                for (int t = t0; t < t1; t++)
                    acc += sh[L.J + t * Ps + a] * sh[L.XJ + t * Ps + b];
            }
            else
            {
                const int a = e - PT;
                for (int t = t0; t < t1; t++)
                    acc += sh[L.XJ + t * Ps + a] * sh[L.z + t];
            }
            if (C == 1)
                sh[((e < PT) ? L.A : L.u) + ((e < PT) ? e : e - PT)] = acc;
            else
                sh[L.part + slot] = acc;
        }
    }
    if (C > 1)
    {
        wave_sync();
        FVB_WAVE_FOR(e, E)
        {
            double acc = 0;
            for (int c = 0; c < C; c++)
                acc += sh[L.part + e * C + c];
            sh[((e < PT) ? L.A : L.u) + ((e < PT) ? e : e - PT)] = acc;
        }
    }
    wave_sync();
    FVB_WAVE_FOR(e, PP)
    {
        const int i = e / P, j = e % P;
        sh[L.Lam + e] = sh[L.A + tri(i, j)] + ((i == j) ? sh[L.pprec + i] : 0.0);
    }
    FVB_WAVE_FOR(i, P)
    sh[L.rhs + i] = sh[L.u + i] + sh[L.pprec + i] * sh[L.pm + i];
    cx.precValid = true;
    cx.covValid = false;
    wave_sync();
    if (!wave_ensure_cov(cx))
        return false;
    FVB_WAVE_FOR(i, P)
    {
        double s = 0;
        for (int j = 0; j < P; j++)
            s += sh[L.Sig + i * P + j] * sh[L.rhs + j];
        sh[L.m + i] = s;
    }
    wave_sync();
    return true;
}

// Ar1cNoiseModel::UpdateAlpha (noisemodel_ar.cc:447-528). Needs L.k and L.JS. Returns a status.
template <int NPHI, int NA>
__device__ __forceinline__ int ar_update_alpha(const KernelArgs &ka, WaveCtx &cx, ArState<NPHI, NA> &st)
{
    const WaveLayout &L = cx.L;
    const double *sh = cx.sh;
    double sc[NPHI];
#pragma unroll
    for (int i = 0; i < NPHI; i++)
        sc[i] = sh[L.b + i] * sh[L.c + i];
    double prec[NA][NA], tmp[NA];
    const bool prior_given = (ka.cfg.ar_alpha_given & 1) != 0; // noise-initial-prior (InputFromMVN, :302-316)
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        tmp[i] = 0; // prior precisions * prior means (:501-502)
#pragma unroll
        for (int j = 0; j < NA; j++)
        {
            // hard-coded prior: zero mean, precision 1e-4 I (:391-394)
            prec[i][j] = prior_given ? ka.cfg.ar_alpha_prior_prec[i][j] : ((i == j) ? 1e-4 : 0.0);
            if (prior_given)
                tmp[i] += ka.cfg.ar_alpha_prior_prec[i][j] * ka.cfg.ar_alpha_prior_mean[j];
        }
    }
#pragma unroll
    for (int i = 1; i <= NPHI; i++)
    {
        prec[i - 1][i - 1] += sc[i - 1] * ar_op_klj(cx, NPHI, i, 2, 0);
        tmp[i - 1] += -0.5 * sc[i - 1] * ar_op_klj(cx, NPHI, i, 1, 0);
    }
    if (NA > 2) // cross terms (:476-485, :505-509); two echoes
    {
        constexpr int Tn = NA - 1, S = (NPHI > 1) ? 1 : 0;
        prec[2][0] += 0.5 * sc[0] * ar_op_klj(cx, NPHI, 1, 1, 1);
        prec[0][2] = prec[2][0];
        prec[Tn][1] += 0.5 * sc[S] * ar_op_klj(cx, NPHI, 2, 1, 1);
        prec[1][Tn] = prec[Tn][1];
        prec[2][2] += sc[0] * ar_op_klj(cx, NPHI, 1, 0, 2);
        prec[Tn][Tn] += sc[S] * ar_op_klj(cx, NPHI, 2, 0, 2);
        tmp[2] += -0.5 * sc[0] * ar_op_klj(cx, NPHI, 1, 0, 1);
        tmp[Tn] += -0.5 * sc[S] * ar_op_klj(cx, NPHI, 2, 0, 1);
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j < NA; j++)
        {
            st.aprec[i][j] = prec[i][j];
            finite = finite && is_finite(prec[i][j]);
        }
    if (!finite)
        return FVB_BAD_AR_ALPHA; // :488-490
    double la;
    int sg;
    if (!small_sym_inverse<NA>(st.aprec, st.acov, la, sg))
        return FVB_BAD_RESULT;
#pragma unroll
    for (int i = 0; i < NA; i++)
        if (st.acov[i][i] < 0)
            return FVB_BAD_AR_ALPHA; // :491-499 negative variance
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        double s = 0;
#pragma unroll
        for (int j = 0; j < NA; j++)
            s += st.acov[i][j] * tmp[j];
        st.am[i] = s;
    }
    ar_update_marginals<NPHI, NA>(st);
    return FVB_OK;
}

// k'Q_n k + tr(Sigma J'Q_n J) for every n (needs L.k, L.JS); uses L.XJ as scratch
template <int NPHI, int NA>
__device__ __forceinline__ void ar_q_forms(WaveCtx &cx, const ArState<NPHI, NA> &st, double (&kqk)[NPHI], double (&trq)[NPHI])
{
    const WaveLayout &L = cx.L;
#pragma unroll
    for (int n = 0; n < NPHI; n++)
    {
        double only[NPHI];
#pragma unroll
        for (int i = 0; i < NPHI; i++)
            only[i] = (i == n) ? 1.0 : 0.0;
        ar_build_band<NPHI, NA>(cx, st, only);
        kqk[n] = ar_band_form(cx, L.k, L.k);
        ar_band_times_matrix(cx, L.J, L.XJ);
        trq[n] = ar_matrix_dot(cx, L.XJ, L.JS);
    }
}

// Ar1cNoiseModel::UpdatePhi (noisemodel_ar.cc:530-556)
template <int NPHI, int NA>
__device__ __forceinline__ void ar_update_phi(const KernelArgs &ka, WaveCtx &cx, const ArState<NPHI, NA> &st)
{
    const WaveLayout &L = cx.L;
    double kqk[NPHI], trq[NPHI];
    ar_q_forms<NPHI, NA>(cx, st, kqk, trq);
    const int nT = L.T / NPHI;
    if (cx.lane == 0)
    {
#pragma unroll
        for (int i = 0; i < NPHI; i++)
        {
            const double tmp = kqk[i] + trq[i];
            cx.sh[L.b + i] = 1 / (tmp * 0.5 + 1 / ka.cfg.noise_prior_b[i]);
            cx.sh[L.c + i] = (nT - 1) * 0.5 + ka.cfg.noise_prior_c[i];
        }
    }
    wave_sync();
}

// Ar1cNoiseModel::CalcFreeEnergy (noisemodel_ar.cc:643-747) + the prior's term (inference_vb.cc:310)
template <int NPHI, int NA>
__device__ __forceinline__ bool ar_free_energy(const KernelArgs &ka, WaveCtx &cx, const ArState<NPHI, NA> &st,
    double Fprior, double &F, bool &finite)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, nT = L.T / NPHI;
    bool ok = wave_ensure_prec(cx);
    ok = wave_ensure_cov(cx) && ok;
    ar_residual(cx);
    ar_j_sigma(cx);
    double kqk[NPHI], trq[NPHI];
    ar_q_forms<NPHI, NA>(cx, st, kqk, trq);
    const double *sh = cx.sh;
    double la;
    int sg;
    double tmpinv[NA][NA];
    small_sym_inverse<NA>(st.aprec, tmpinv, la, sg);
    const double expectedLogAlphaDist = 0.5 * la - 0.5 * NA * (LOG_2PI + 1);
    const double expectedLogThetaDist = 0.5 * cx.logdetLam - 0.5 * P * (LOG_2PI + 1);
    double expectedLogPhiDist = 0, p0 = 0, p9 = 0, p2a = 0, p2b = 0;
#pragma unroll
    for (int i = 0; i < NPHI; i++)
    {
        const double si = sh[L.b + i], ci = sh[L.c + i];
        const double siPrior = ka.cfg.noise_prior_b[i], ciPrior = ka.cfg.noise_prior_c[i];
        const double dg = digamma(ci) + log(si);
        expectedLogPhiDist += -gammaln(ci) - ci * log(si) - ci + (ci - 1) * dg;
        p0 += dg * ((nT - 1) * 0.5 + ciPrior - 1);
        p9 += -2 * gammaln(ciPrior) - 2 * ciPrior * log(siPrior) - si * ci / siPrior;
        p2a += (si * ci) * kqk[i];
        p2b += (si * ci) * trq[i];
    }
    double parts = p0;
    parts += -LOG_2PI * (nT - 1 + 0.5 * NA + 0.5 * P);
    parts += -0.5 * p2a - 0.5 * p2b;
    double logdetPrior = 0, quad = 0, trSL0 = 0;
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(sh[L.pprec + i]));
        const double dm = sh[L.m + i] - sh[L.pm + i];
        quad += dm * sh[L.pprec + i] * dm;
        trSL0 += sh[L.Sig + i * P + i] * sh[L.pprec + i];
    }
    parts += 0.5 * logdetPrior;
    parts += -0.5 * quad;
    parts += -0.5 * trSL0;
    double qa = 0, tra = 0;
    if (ka.cfg.ar_alpha_given & 1) // the prior of noise-initial-prior: parts [6] - [8] with its mean and precision matrix (:720-729)
    {
        double p0m[NA][NA], p0i[NA][NA], lp;
        int sp;
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NA; j++)
                p0m[i][j] = ka.cfg.ar_alpha_prior_prec[i][j];
        small_sym_inverse<NA>(p0m, p0i, lp, sp);
        parts += 0.5 * lp;
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NA; j++)
            {
                qa += (st.am[i] - ka.cfg.ar_alpha_prior_mean[i]) * p0m[i][j] * (st.am[j] - ka.cfg.ar_alpha_prior_mean[j]);
                tra += st.acov[i][j] * p0m[j][i];
            }
    }
    else
    {
        parts += 0.5 * NA * log(1e-4); // log det of the alpha prior precision 1e-4 I
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            qa += st.am[i] * 1e-4 * st.am[i];
            tra += st.acov[i][i] * 1e-4;
        }
    }
    parts += -0.5 * qa;
    parts += -0.5 * tra;
    parts += p9;
    F = -expectedLogAlphaDist - expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior;
    return ok;
}

// result MVN: MVNDist(fwd_post, Ar1cParams::OutputAsMVN()) (noisemodel_ar.cc:287-300), and the per-voxel outputs
template <int NPHI, int NA>
__device__ __forceinline__ void ar_write_outputs(const KernelArgs &ka, WaveCtx &cx, const ArState<NPHI, NA> &st, int status,
    bool setup_failed, double F, int it, int hist_len)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, PP = L.PP, v = cx.v;
    const size_t V = cx.V;
    double *sh = cx.sh;
    const int n = P + NA + NPHI;
    const int nCov = n * (n + 1) / 2;
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
            else if (i < P + NA)
            {
                if (j >= P)
                {
#pragma unroll
                    for (int a = 0; a < NA; a++)
#pragma unroll
                        for (int b = 0; b < NA; b++)
                            if (a == i - P && b == j - P)
                                val = st.acov[a][b];
                }
            }
            else if (i == j)
            {
                const double b = sh[L.b + (i - P - NA)], c = sh[L.c + (i - P - NA)];
                val = b * b * c;
            }
            dst[(size_t)e * V] = val;
        }
        FVB_WAVE_FOR(i, n)
        {
            double val = 0;
            if (i < P)
                val = sh[L.m + i];
            else if (i < P + NA)
            {
#pragma unroll
                for (int a = 0; a < NA; a++)
                    if (a == i - P)
                        val = st.am[a];
            }
            else
                val = sh[L.b + (i - P - NA)] * sh[L.c + (i - P - NA)];
            dst[(size_t)(nCov + i) * V] = val;
        }
        if (cx.lane == 0)
            dst[(size_t)(nCov + n) * V] = 1.0;
    }
    if (cx.lane == 0)
    {
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
}

template <int NPHI, int NA, bool NEEDF>
__global__ __launch_bounds__(64) void vb_wave_ar_kernel(const KernelArgs ka, const WaveLayout L)
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
    const int v = cx.v, T = L.T, P = L.P, PP = L.PP;
    const size_t V = cx.V;
    double *sh = cx.sh;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    FVB_WAVE_FOR(t, T)
    {
        sh[L.y + t] = load_data(ka, (size_t)t * V + v);
        cx.phi[t] = 0;
    }
    FVB_WAVE_FOR(e, PP)
    {
        sh[L.Sig + e] = 0;
        sh[L.Lam + e] = 0;
    }
    wave_sync();

    // ---- Vb::SetupPerVoxelDists (inference_vb.cc:207-247) with Ar1cNoiseModel's initial
    // distributions (noisemodel_ar.cc:379-403) ----
    ArState<NPHI, NA> st, st_saved;
    constexpr int NN = NA + NPHI;
    const int n = P + NN;
    const int nCov = n * (n + 1) / 2;
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        st.am[i] = 0;
#pragma unroll
        for (int j = 0; j < NA; j++)
        {
            st.aprec[i][j] = (i == j) ? 1e-4 : 0.0;
            st.acov[i][j] = (i == j) ? 1e4 : 0.0;
        }
    }
    if (ka.cfg.ar_alpha_given & 2) // noise-initial-posterior (inference_vb.cc:205; continue-from-mvn below overrides it, :214)
    {
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            st.am[i] = ka.cfg.ar_alpha_post_mean[i];
#pragma unroll
            for (int j = 0; j < NA; j++)
                st.acov[i][j] = ka.cfg.ar_alpha_post_cov[i][j];
        }
        double la0;
        int sg0;
        small_sym_inverse<NA>(st.acov, st.aprec, la0, sg0);
    }
    if (ka.cfg.init_mvn)
    {
        // MVNDist::Load + Ar1cParams::InputFromMVN (dist_mvn.cc:347-374; noisemodel_ar.cc:302-316)
        const double *src = ka.cfg.init_mvn + v;
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = src[(size_t)tri(e / P, e % P) * V];
        FVB_WAVE_FOR(i, P)
        sh[L.m + i] = src[(size_t)(nCov + i) * V];
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            st.am[i] = src[(size_t)(nCov + P + i) * V];
#pragma unroll
            for (int j = 0; j < NA; j++)
                st.acov[i][j] = src[(size_t)tri(P + i, P + j) * V];
        }
        double la;
        int sg;
        small_sym_inverse<NA>(st.acov, st.aprec, la, sg);
        FVB_WAVE_FOR(i, NPHI)
        {
            const double nm = src[(size_t)(nCov + P + NA + i) * V];
            const double nv = src[(size_t)tri(P + NA + i, P + NA + i) * V];
            const double b = nv / nm;
            sh[L.b + i] = b;
            sh[L.c + i] = nm / b;
        }
    }
    else
    {
        double data_max = 0;
        if (ka.cfg.model == FVB_MODEL_EXP)
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
            sh[L.m + i] = to_fabber(tr, mean);
            sh[L.Sig + i * P + i] = to_fabber_var(tr, FVB_KPARAM(ka, post_var, i));
        }
        FVB_WAVE_FOR(i, NPHI)
        {
            sh[L.b + i] = ka.cfg.noise_post_b[i];
            sh[L.c + i] = ka.cfg.noise_post_c[i];
        }
    }
    FVB_WAVE_FOR(i, P)
    {
        sh[L.pm + i] = 0;
        sh[L.pprec + i] = 1;
    }
    cx.covValid = true;
    cx.precValid = false;
    cx.logdetLam = 0;
    cx.sv_prec = false;
    wave_sync();

    double F = 1234.5678, Fprior = 0;
    int it = 0, hist_len = 0;
    bool setup_failed = false;
    const bool use_save = ka.cfg.convergence == FVB_CONV_FREDUCE || ka.cfg.convergence == FVB_CONV_TRIALMODE
        || ka.cfg.convergence == FVB_CONV_LM;

    int status = wave_recentre(ka, ma, cx, false); // inference_vb.cc:235 and :443
    if (status != FVB_OK)
        setup_failed = true;
    // Precalculate (noisemodel_ar.cc:749-769)
    ar_update_marginals<NPHI, NA>(st);
    FVB_WAVE_FOR(i, NPHI)
    sh[L.c + i] = ka.cfg.noise_prior_c[i] + (T / NPHI - 1) * 0.5;
    wave_sync();

    if (status == FVB_OK)
    {
        ConvState conv;
        conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
        conv_reset(conv);
        if (use_save)
        {
            wave_save_state(cx);
            st_saved = st;
        }
        bool stop = false;
#define FVB_AR_EVAL_F()                                                                                      \
    {                                                                                                        \
        double Fn_;                                                                                          \
        bool fin_ = true;                                                                                    \
        if (!ar_free_energy<NPHI, NA>(ka, cx, st, Fprior, Fn_, fin_))                                        \
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
            if (use_save && conv_need_save(conv))
            {
                wave_save_state(cx);
                st_saved = st;
            }
            if (!wave_apply_priors<NEEDF>(ka, cx, it, Fprior))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            if (NEEDF)
                FVB_AR_EVAL_F()
            if (!ar_update_theta<NPHI, NA>(cx, st)) // the AR model ignores the LM damping (:558-634)
            {
                status = FVB_BAD_RESULT;
                break;
            }
            if (NEEDF)
                FVB_AR_EVAL_F()
            // UpdateNoise = UpdateAlpha, then UpdatePhi (:405-410)
            if (!wave_ensure_prec(cx) || !wave_ensure_cov(cx))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            ar_residual(cx);
            ar_j_sigma(cx);
            status = ar_update_alpha<NPHI, NA>(ka, cx, st);
            if (status != FVB_OK)
                break;
            ar_update_phi<NPHI, NA>(ka, cx, st);
            if (NEEDF)
                FVB_AR_EVAL_F()
            status = wave_recentre(ka, ma, cx, false);
            if (status != FVB_OK)
                break;
            if (NEEDF)
                FVB_AR_EVAL_F()
            if (cx.lane == 0 && ka.out.f_history && hist_len < ka.cfg.f_history_rows)
                ka.out.f_history[(size_t)hist_len * V + v] = F;
            hist_len++;
            ++it;
            stop = conv_test(conv, F);
        } while (!stop);

        if (status == FVB_OK)
        {
            if (use_save && conv_need_save(conv))
            {
                wave_save_state(cx);
                st_saved = st;
            }
            if (use_save && conv_need_revert(conv))
            {
                wave_restore_state(cx);
                st = st_saved;
                status = wave_recentre(ka, ma, cx, false);
                if (status == FVB_OK && NEEDF)
                {
                    do
                    {
                        FVB_AR_EVAL_F()
                    } while (false);
                }
            }
        }
#undef FVB_AR_EVAL_F
    }

    ar_write_outputs<NPHI, NA>(ka, cx, st, status, setup_failed, F, it, hist_len);
}

#endif // __HIPCC__
} // namespace fvb
