/*
 * vb_lane_arn_kernel.h - voxelwise VB with the AR(1) noise model in its general form (two interleaved echoes,
 * ar1-cross-terms none / same / dual: 2 / 3 / 4 AR coefficients), one lane per voxel.
 *
 * Reference: Ar1cNoiseModel / Ar1cMatrixCache (noisemodel_ar.cc:83-769). Its dense "alpha matrices"
 * M(n, a12pow, a34pow) are single lines of nTimes - 1 equal entries (+1, or -1 where a12pow + a34pow == 1) that start
 * at (row, col) and step by the number of echoes along the diagonal, reflected to stay symmetric (:112-181); the
 * marginal precision of echo n is Q_n = sum_ab w_n[a][b] M(n, a, b) with weights from the alpha posterior (:199-222).
 * For two echoes that is 12 lines with lags 0, 1, 2, 3 in the interleaved series.
 *
 * The one-echo kernel (vb_lane_ar_kernel.h) keeps the moments of its 3 lines in registers and streams the series
 * once per iteration. Twelve lines' moments (P x P + P + 1 entries each) do not fit a lane - but they need not be
 * kept, because every use of them is a WEIGHTED sum or a contraction whose other factor is known when the pass
 * starts:
 *   pass 1 (re-centre): with the present alpha / phi posterior the EFFECTIVE moments J'XJ, J'X(y - g),
 *       X = sum_n phibar_n Q_n, are accumulated directly - per timepoint one coefficient per lag (the sum of the
 *       weights of the lines that have an entry there) times the outer product with the row 0, 1, 2 or 3 back;
 *       UpdateTheta (:558-634) needs nothing else;
 *   pass 2 (after UpdateTheta): for each line l the scalar S_l = k'M_l k + tr(Sigma J'M_l J),
 *       k = y - g - J (m - centre), summed along the line as the reference does, with the rows of J and the k of the
 *       last three timepoints in registers: UpdateAlpha (OperatorKLJ, :433-528), UpdatePhi (:530-556) and the
 *       free energy's data term (:643-747) are linear in these 12 numbers.
 * Two streaming passes per iteration instead of one; with F the at-centre forms the "lin" / "before" evaluations need
 * come out of pass 1. No moment form, hence no cancellation and no rescue path.
 * Restated from oracle/vb_oracle_arn.inc's reading of the reference; held to it by tests/test_ar_general.py.
 */
#pragma once

#include "vb_lane_ar_kernel.h"

namespace fvb
{
#if defined(__HIPCC__)

// the line types in the order (a12pow, a34pow) = (0,0) (1,0) (2,0) (0,1) (1,1) (0,2)
enum
{
    ARN_00 = 0,
    ARN_10 = 1,
    ARN_20 = 2,
    ARN_01 = 3,
    ARN_11 = 4,
    ARN_02 = 5
};

// Ar1cMatrixCache::Update for two echoes (noisemodel_ar.cc:93-181), 0-based: the larger of (row, col) of a line's
// first entry, the distance between them, and the entries' value
__device__ __forceinline__ constexpr int arn_hi(int n, int k)
{
    // echo 1: (2,2) (0,2) (0,0) (3,2) (3,0) (3,3); echo 2 (rows and columns 2i <-> 2i+1): (3,3) (1,3) (1,1) (2,3) (2,1) (2,2)
    return n == 0 ? (k == ARN_00 ? 2 : k == ARN_10 ? 2 : k == ARN_20 ? 0 : 3)
                  : (k == ARN_00 ? 3 : k == ARN_10 ? 3 : k == ARN_20 ? 1 : k == ARN_01 ? 3 : 2);
}
__device__ __forceinline__ constexpr int arn_lag(int n, int k)
{
    return n == 0 ? (k == ARN_10 ? 2 : k == ARN_01 ? 1 : k == ARN_11 ? 3 : 0)
                  : (k == ARN_10 ? 2 : k == ARN_01 ? 1 : k == ARN_11 ? 1 : 0);
}
__device__ __forceinline__ constexpr double arn_value(int k)
{
    return (k == ARN_10 || k == ARN_01) ? -1.0 : 1.0;
}
// does line (n, k) have an entry whose later index is t? (nT = samples per echo)
__device__ __forceinline__ bool arn_active(int n, int k, int t, int nT)
{
    const int e = t - arn_hi(n, k);
    return e >= 0 && (e & 1) == 0 && (e >> 1) <= nT - 2;
}

template <int NA>
struct ArnAlpha
{
    static constexpr int NT = NA * (NA + 1) / 2;
    double mean[NA];
    double cov[NT];
    double logdetPrec; // log|det| of the precision the covariance came from
    double w[2][6];    // Ar1cParams marginal weights (:199-222), by echo and line type
    // Ar1cParams::Update of the marginals
    __device__ __forceinline__ void update_marginal()
    {
#pragma unroll
        for (int n = 0; n < 2; n++)
        {
#pragma unroll
            for (int k = 0; k < 6; k++)
                w[n][k] = 0;
            w[n][ARN_00] = 1;
            w[n][ARN_10] = mean[n];
            w[n][ARN_20] = cov[tri(n, n)] + mean[n] * mean[n];
            if (NA >= 3)
            {
                const int x = (NA == 4) ? 2 + n : 2; // the cross term's coefficient
                w[n][ARN_01] = mean[x];
                w[n][ARN_11] = cov[tri(x > n ? x : n, x > n ? n : x)] + mean[n] * mean[x];
                w[n][ARN_02] = cov[tri(x, x)] + mean[x] * mean[x];
            }
        }
    }
};

template <int P>
struct ArnMoments
{
    static constexpr int PT = P * (P + 1) / 2;
    double A[PT]; // J'XJ
    double u[P];  // J'X(y - g)
    double ml[P]; // the linearisation's centre
};

// One streaming pass about `centre`. MOMENTS: the effective moments with the line coefficients cw (weight x phibar x
// value). FORMS: S[n][k] = value (k'M k + tr(Sigma J'M J)) for every line, k = y - g + J nd (nd = centre - m; all
// zero at the centre itself).
template <class Model, int P, int NA, bool MOMENTS, bool FORMS, class Feed>
__device__ __forceinline__ int arn_pass(const KernelArgs &ka, const ModelArgs &ma, const Feed &feed, const double (&centre)[P],
    const double (&cw)[2][6], ArnMoments<P> &mo, const double *Sig, const double (&nd)[P], double (&S)[2][6], bool precise)
{
    constexpr int PT = P * (P + 1) / 2;
    constexpr int NK = (NA >= 3) ? 6 : 3; // (without cross terms the lines 01, 11, 02 carry no weight)
    const int nT = ka.cfg.n_times / 2;
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
        if (MOMENTS)
        {
            mo.ml[i] = centre[i];
            mo.u[i] = 0;
        }
    }
    if (MOMENTS)
    {
#pragma unroll
        for (int i = 0; i < PT; i++)
            mo.A[i] = 0;
    }
    if (FORMS)
    {
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int k = 0; k < 6; k++)
                S[n][k] = 0;
    }
    bool bad_offset = false, bad_jac = false;
    double Jw[3][P], kw[3]; // the rows 1, 2, 3 timepoints back (zero before the series starts: never used there)
#pragma unroll
    for (int b = 0; b < 3; b++)
    {
        kw[b] = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
            Jw[b][i] = 0;
    }
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
        double k = y_cur - g;
        if (FORMS)
        {
            double Jd = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
                Jd += J[i] * nd[i];
            k += Jd;
        }
        if (MOMENTS)
        {
            // one coefficient per lag: the lines that have an entry ending at t
            double c[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int q = 0; q < NK; q++)
                    if (arn_active(n, q, t, nT)) // (uniform)
                        c[arn_lag(n, q)] += cw[n][q];
            // (at the centre k = y - g = the residual the moments want)
            const double r = y_cur - g;
#pragma unroll
            for (int i = 0; i < P; i++)
            {
#pragma unroll
                for (int j = 0; j <= i; j++)
                {
                    double a = c[0] * (J[i] * J[j]);
                    a += c[1] * (J[i] * Jw[0][j] + Jw[0][i] * J[j]);
                    a += c[2] * (J[i] * Jw[1][j] + Jw[1][i] * J[j]);
                    if (NA >= 3)
                        a += c[3] * (J[i] * Jw[2][j] + Jw[2][i] * J[j]);
                    mo.A[tri(i, j)] += a;
                }
                // (kw holds y - g of the earlier rows in a MOMENTS pass: nd = 0 there)
                double b = c[0] * (J[i] * r);
                b += c[1] * (J[i] * kw[0] + Jw[0][i] * r);
                b += c[2] * (J[i] * kw[1] + Jw[1][i] * r);
                if (NA >= 3)
                    b += c[3] * (J[i] * kw[2] + Jw[2][i] * r);
                mo.u[i] += b;
            }
        }
        if (FORMS)
        {
            double v[P]; // Sigma J_t'
#pragma unroll
            for (int i = 0; i < P; i++)
            {
                double s = 0;
#pragma unroll
                for (int j = 0; j < P; j++)
                    s += Sig[tri(i, j)] * J[j];
                v[i] = s;
            }
            // per lag: k_t k_p and J_p Sigma J_t' with the partner row p = t - lag
            double kk[4], jj[4];
            kk[0] = k * k;
            jj[0] = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
                jj[0] += J[i] * v[i];
#pragma unroll
            for (int b = 0; b < 3; b++)
            {
                kk[b + 1] = 2 * (k * kw[b]);
                double s = 0;
#pragma unroll
                for (int i = 0; i < P; i++)
                    s += Jw[b][i] * v[i];
                jj[b + 1] = 2 * s;
            }
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int q = 0; q < NK; q++)
                    if (arn_active(n, q, t, nT)) // (uniform)
                        S[n][q] += arn_value(q) * (kk[arn_lag(n, q)] + jj[arn_lag(n, q)]);
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            Jw[2][i] = Jw[1][i];
            Jw[1][i] = Jw[0][i];
            Jw[0][i] = J[i];
        }
        kw[2] = kw[1];
        kw[1] = kw[0];
        kw[0] = k;
    };
    feed.run(step);
    return bad_offset ? FVB_BAD_OFFSET : (bad_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}

// line coefficients of the effective moments: phibar_n x weight x value
template <int NA>
__device__ __forceinline__ void arn_coefficients(const ArnAlpha<NA> &al, const double (&pb)[2], const double (&pc)[2], double (&cw)[2][6])
{
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int k = 0; k < 6; k++)
            cw[n][k] = pb[n] * pc[n] * al.w[n][k] * arn_value(k);
}

// Ar1cNoiseModel::UpdateTheta (noisemodel_ar.cc:558-634)
template <int P>
__device__ __forceinline__ bool update_theta_arn(VoxelState<P> &st, const ArnMoments<P> &mo)
{
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            st.Lam[tri(i, j)] = mo.A[tri(i, j)] + ((i == j) ? st.pprec[i] : 0.0);
    st.precValid = true;
    st.covValid = false;
    double rhs[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        double s = mo.u[i];
#pragma unroll
        for (int j = 0; j < P; j++)
            s += mo.A[tri(i, j)] * mo.ml[j];
        rhs[i] = s + st.pprec[i] * st.pm[i];
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

// Ar1cNoiseModel::HardcodedInitialDists (noisemodel_ar.cc:379-403), or the posterior of noise-initial-posterior
template <int NA>
__device__ __forceinline__ void arn_initial_alpha(const KernelArgs &ka, ArnAlpha<NA> &al)
{
    if (ka.cfg.ar_alpha_given & 2)
    {
        constexpr int NT = NA * (NA + 1) / 2;
        double inv[NT], la;
        int sg;
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            al.mean[i] = ka.cfg.ar_alpha_post_mean[i];
#pragma unroll
            for (int j = 0; j <= i; j++)
                al.cov[tri(i, j)] = ka.cfg.ar_alpha_post_cov[i][j];
        }
        ldl_inverse<NA>(al.cov, inv, la, sg);
        al.logdetPrec = -la;
        return;
    }
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        al.mean[i] = 0;
#pragma unroll
        for (int j = 0; j <= i; j++)
            al.cov[tri(i, j)] = (i == j) ? 1.0 / AR_ALPHA_PRIOR_PREC : 0.0;
    }
    al.logdetPrec = NA * log(AR_ALPHA_PRIOR_PREC);
}

// Ar1cNoiseModel::UpdateAlpha + UpdatePhi (noisemodel_ar.cc:447-556) from the line scalars
template <int NA>
__device__ __forceinline__ int update_noise_arn(const KernelArgs &ka, ArnAlpha<NA> &al, double (&pb)[2], double (&pc)[2],
    const double (&S)[2][6])
{
    constexpr int NT = NA * (NA + 1) / 2;
    const double nT = (double)(ka.cfg.n_times / 2);
    const double sc[2] = { pb[0] * pc[0], pb[1] * pc[1] };
    double prec[NT];
    const bool prior_given = (ka.cfg.ar_alpha_given & 1) != 0; // (uniform) noise-initial-prior (InputFromMVN, :302-316)
#pragma unroll
    for (int i = 0; i < NA; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            prec[tri(i, j)] = prior_given ? ka.cfg.ar_alpha_prior_prec[i][j] : ((i == j) ? AR_ALPHA_PRIOR_PREC : 0.0);
    prec[tri(0, 0)] += sc[0] * S[0][ARN_20];
    prec[tri(1, 1)] += sc[1] * S[1][ARN_20];
    if (NA >= 3)
    {
        constexpr int X = NA - 1; // the last alpha (:470)
        prec[tri(2, 0)] += 0.5 * sc[0] * S[0][ARN_11];
        prec[tri(X, 1)] += 0.5 * sc[1] * S[1][ARN_11];
        prec[tri(2, 2)] += sc[0] * S[0][ARN_02];
        prec[tri(X, X)] += sc[1] * S[1][ARN_02];
    }
    bool finite = true;
#pragma unroll
    for (int i = 0; i < NT; i++)
        finite = finite && is_finite(prec[i]);
    if (!finite)
        return FVB_BAD_AR_ALPHA;
    double cov[NT], logabs;
    int sign;
    if (!ldl_inverse<NA>(prec, cov, logabs, sign))
        return FVB_BAD_RESULT;
#pragma unroll
    for (int i = 0; i < NA; i++)
        if (cov[tri(i, i)] < 0)
            return FVB_BAD_AR_ALPHA;
    double tmp[NA];
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        tmp[i] = 0; // (the hard-coded prior's means are zero); else prior precisions x prior means (:501-502)
        if (prior_given)
#pragma unroll
            for (int j = 0; j < NA; j++)
                tmp[i] += ka.cfg.ar_alpha_prior_prec[i][j] * ka.cfg.ar_alpha_prior_mean[j];
    }
    tmp[0] += -0.5 * sc[0] * S[0][ARN_10];
    tmp[1] += -0.5 * sc[1] * S[1][ARN_10];
    if (NA >= 3)
    {
        constexpr int X = NA - 1;
        tmp[2] += -0.5 * sc[0] * S[0][ARN_01];
        tmp[X] += -0.5 * sc[1] * S[1][ARN_01];
    }
#pragma unroll
    for (int i = 0; i < NT; i++)
        al.cov[i] = cov[i];
    al.logdetPrec = logabs;
#pragma unroll
    for (int i = 0; i < NA; i++)
    {
        double s = 0;
#pragma unroll
        for (int j = 0; j < NA; j++)
            s += cov[tri(i > j ? i : j, i > j ? j : i)] * tmp[j];
        al.mean[i] = s;
    }
    al.update_marginal();
    // UpdatePhi with the new marginal
#pragma unroll
    for (int n = 0; n < 2; n++)
    {
        double t = 0;
#pragma unroll
        for (int k = 0; k < 6; k++)
            t += al.w[n][k] * S[n][k];
        pb[n] = 1 / (t * 0.5 + 1 / ka.cfg.noise_prior_b[n]);
        pc[n] = (nT - 1) * 0.5 + ka.cfg.noise_prior_c[n];
    }
    return FVB_OK;
}

// Ar1cNoiseModel::CalcFreeEnergy (noisemodel_ar.cc:643-747)
template <int P, int NA>
__device__ __forceinline__ bool calc_free_energy_arn(const KernelArgs &ka, VoxelState<P> &st, const ArnAlpha<NA> &al,
    const double (&pb)[2], const double (&pc)[2], const double (&S)[2][6], double Fprior, double &F, bool &finite)
{
    bool ok = ensure_prec<P>(st);
    const double nT = (double)(ka.cfg.n_times / 2);
    const double expectedLogAlphaDist = 0.5 * al.logdetPrec - 0.5 * NA * (LOG_2PI + 1);
    const double expectedLogThetaDist = 0.5 * st.logdetLam - 0.5 * P * (LOG_2PI + 1);
    double expectedLogPhiDist = 0, parts = 0;
#pragma unroll
    for (int n = 0; n < 2; n++)
    {
        const double si = pb[n], ci = pc[n], siPrior = ka.cfg.noise_prior_b[n], ciPrior = ka.cfg.noise_prior_c[n];
        const double dg = digamma(ci) + log(si);
        expectedLogPhiDist += -gammaln(ci) - ci * log(si) - ci + (ci - 1) * dg;
        parts += dg * ((nT - 1) * 0.5 + ciPrior - 1);                                     // [0]
        parts += -2 * gammaln(ciPrior) - 2 * ciPrior * log(siPrior) - si * ci / siPrior; // [9]
        double t = 0;
#pragma unroll
        for (int k = 0; k < 6; k++)
            t += al.w[n][k] * S[n][k];
        parts += -0.5 * si * ci * t;                                                      // [2]
    }
    parts += -LOG_2PI * (nT - 1 + 0.5 * NA + 0.5 * P);                                    // [1]
    double logdetPrior = 0, quad = 0, trSL0 = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(st.pprec[i]));
        const double dm = st.m[i] - st.pm[i];
        quad += dm * st.pprec[i] * dm;
        trSL0 += st.Sig[tri(i, i)] * st.pprec[i];
    }
    parts += 0.5 * logdetPrior; // [3]
    parts += -0.5 * quad;       // [4]
    parts += -0.5 * trSL0;      // [5]
    if (ka.cfg.ar_alpha_given & 1) // (uniform) the prior of noise-initial-prior: its log-determinant, quadratic form, trace (:720-729)
    {
        constexpr int NT = NA * (NA + 1) / 2;
        double p0[NT], p0i[NT], lp;
        int sp;
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j <= i; j++)
                p0[tri(i, j)] = ka.cfg.ar_alpha_prior_prec[i][j];
        ldl_inverse<NA>(p0, p0i, lp, sp);
        double qa = 0, tra = 0;
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j < NA; j++)
            {
                const double pij = p0[tri(i > j ? i : j, i > j ? j : i)];
                qa += (al.mean[i] - ka.cfg.ar_alpha_prior_mean[i]) * pij * (al.mean[j] - ka.cfg.ar_alpha_prior_mean[j]);
                tra += al.cov[tri(i > j ? i : j, i > j ? j : i)] * pij;
            }
        parts += 0.5 * lp;   // [6]
        parts += -0.5 * qa;  // [7]
        parts += -0.5 * tra; // [8]
    }
    else
    {
        double mm = 0, trc = 0;
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            mm += al.mean[i] * al.mean[i];
            trc += al.cov[tri(i, i)];
        }
        parts += 0.5 * NA * log(AR_ALPHA_PRIOR_PREC); // [6]
        parts += -0.5 * AR_ALPHA_PRIOR_PREC * mm;     // [7]
        parts += -0.5 * AR_ALPHA_PRIOR_PREC * trc;    // [8]
    }
    F = -expectedLogAlphaDist - expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior;
    return ok;
}

template <int P, int NA>
constexpr int lane_arn_save_rows()
{
    return lane_save_rows<P>() + 2 + NA + NA * (NA + 1) / 2 + 1;
}

template <int P, int NA>
__device__ __forceinline__ void save_arn(const KernelArgs &ka, int v, const ArnAlpha<NA> &al, const double (&pb)[2], const double (&pc)[2])
{
    const size_t V = (size_t)ka.cfg.n_voxels;
    double *p = ka.save + (size_t)lane_save_rows<P>() * V + v;
    int r = 0;
    p[(size_t)(r++) * V] = pb[1];
    p[(size_t)(r++) * V] = pc[1];
#pragma unroll
    for (int i = 0; i < NA; i++)
        p[(size_t)(r++) * V] = al.mean[i];
#pragma unroll
    for (int i = 0; i < NA * (NA + 1) / 2; i++)
        p[(size_t)(r++) * V] = al.cov[i];
    p[(size_t)(r++) * V] = al.logdetPrec;
}
template <int P, int NA>
__device__ __forceinline__ void restore_arn(const KernelArgs &ka, int v, ArnAlpha<NA> &al, double (&pb)[2], double (&pc)[2])
{
    const size_t V = (size_t)ka.cfg.n_voxels;
    const double *p = ka.save + (size_t)lane_save_rows<P>() * V + v;
    int r = 0;
    pb[1] = p[(size_t)(r++) * V];
    pc[1] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < NA; i++)
        al.mean[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < NA * (NA + 1) / 2; i++)
        al.cov[i] = p[(size_t)(r++) * V];
    al.logdetPrec = p[(size_t)(r++) * V];
    al.update_marginal();
}

// FEED: FEED_TILES_F32 / FEED_TILES_F64 - the series is always read from the tiled copy (AR noise rejects masked
// timepoints, noisemodel_ar.cc:351-355)
// Registers: the two passes' accumulators, the three-row window and the alpha posterior next to the voxel state do
// not fit 256 VGPRs (363 - 2164 spilled with two waves per SIMD). With ONE wave per SIMD the spills land in the
// other half of the register file (AGPRs) instead of scratch: measured at T = 200, P = 4, 262 144 voxels, 10
// iterations (tools/measure/ar_wave_rate.py) 36.9 against 24.9 M voxels/s with four AR coefficients, 19.1 against
// 9.3 with F; only the smallest variant (no cross terms, no F) is faster with two waves (54.5 against 43.7).
template <class Model, int P, int NA, bool NEEDF, int FEED>
__global__ __launch_bounds__(64, (NA == 2 && !NEEDF) ? 2 : 1) void vb_lane_arn_kernel(const KernelArgs ka)
{
    constexpr int PT = P * (P + 1) / 2;
    constexpr int NT = NA * (NA + 1) / 2;
    typedef typename FeedTraits<FEED>::raw RAW;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    if (v >= ka.cfg.n_voxels)
        return;
    const RAW *lane_tile = (const RAW *)ka.tiles + (size_t)blockIdx.x * Tile<RAW>::block_elems(T) + (size_t)threadIdx.x * Tile<RAW>::G;
    const ArTileFeed<RAW> feed{ lane_tile, T };

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    VoxelState<P> st;
    ArnMoments<P> mo;
    ArnAlpha<NA> al;
    double pb[2], pc[2];     // the two echoes' precision posteriors Gamma(scale b, shape c)
    double S[2][6], cw[2][6]; // line scalars of the last forms pass; line coefficients of the last moments pass
    double zero[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        zero[i] = 0;
    int status = FVB_OK;
    constexpr int n = P + NA + 2;
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
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            al.mean[i] = src[(size_t)(nCov + P + i) * V];
#pragma unroll
            for (int j = 0; j <= i; j++)
                al.cov[tri(i, j)] = src[(size_t)tri(P + i, P + j) * V];
        }
        {
            double inv[NT], la;
            int sg;
            (void)ldl_inverse<NA>(al.cov, inv, la, sg);
            al.logdetPrec = -la;
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            const double nm = src[(size_t)(nCov + P + NA + e) * V];
            const double nv = src[(size_t)tri(P + NA + e, P + NA + e) * V];
            pb[e] = nv / nm;
            pc[e] = nm / pb[e];
        }
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
        arn_initial_alpha<NA>(ka, al);
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            pb[e] = ka.cfg.noise_post_b[e];
            pc[e] = ka.cfg.noise_post_c[e];
        }
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
    al.update_marginal();
#pragma unroll
    for (int e = 0; e < 2; e++)
        pc[e] = ka.cfg.noise_prior_c[e] + ((double)(T / 2) - 1) * 0.5;
    st.b = pb[0];
    st.c = pc[0];

    double F = 1234.5678;
    double Fprior = 0;
    int it = 0;
    int hist_len = 0;
    bool setup_failed = false;

    // the linearisation about the initial means: effective moments for the first UpdateTheta and, with F, the forms
    // at the centre for its "before" evaluation
    arn_coefficients<NA>(al, pb, pc, cw);
    status = arn_pass<Model, P, NA, true, NEEDF>(ka, ma, feed, st.m, cw, mo, st.Sig, zero, S, true);
    if (status != FVB_OK)
        setup_failed = true;

    if (status == FVB_OK)
    {
        ConvState conv;
        conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
        conv_reset(conv);
        if (ka.save)
        {
            st.b = pb[0];
            st.c = pc[0];
            save_state<P>(ka, v, st);
            save_arn<P, NA>(ka, v, al, pb, pc);
        }
        bool stop = false;
#define FVB_EVAL_F_ARN()                                                                                     \
    {                                                                                                        \
        double Fn_;                                                                                          \
        bool fin_ = true;                                                                                    \
        if (!calc_free_energy_arn<P, NA>(ka, st, al, pb, pc, S, Fprior, Fn_, fin_))                          \
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
                st.b = pb[0];
                st.c = pc[0];
                save_state<P>(ka, v, st);
                save_arn<P, NA>(ka, v, al, pb, pc);
            }
            if (!apply_priors<P, NEEDF>(ka, v, it, st, Fprior))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            if (NEEDF) // "before": S holds the forms at the centre (the pass that ended the last iteration)
            {
                if (!ensure_cov<P>(st))
                {
                    status = FVB_BAD_RESULT;
                    break;
                }
                FVB_EVAL_F_ARN()
            }
            if (!update_theta_arn<P>(st, mo))
            {
                status = FVB_BAD_RESULT;
                break;
            }
            {
                double nd[P];
#pragma unroll
                for (int i = 0; i < P; i++)
                    nd[i] = mo.ml[i] - st.m[i];
                (void)arn_pass<Model, P, NA, false, true>(ka, ma, feed, mo.ml, cw, mo, st.Sig, nd, S, false);
            }
            if (NEEDF) // "theta"
                FVB_EVAL_F_ARN()
            status = update_noise_arn<NA>(ka, al, pb, pc, S);
            if (status != FVB_OK)
                break;
            if (NEEDF) // "phi"
                FVB_EVAL_F_ARN()
            arn_coefficients<NA>(al, pb, pc, cw);
            status = arn_pass<Model, P, NA, true, NEEDF>(ka, ma, feed, st.m, cw, mo, st.Sig, zero, S, false);
            if (status != FVB_OK)
                break;
            if (NEEDF) // "lin"
                FVB_EVAL_F_ARN()
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
                st.b = pb[0];
                st.c = pc[0];
                save_state<P>(ka, v, st);
                save_arn<P, NA>(ka, v, al, pb, pc);
            }
            if (ka.save && conv_need_revert(conv))
            {
                restore_state<P>(ka, v, st);
                pb[0] = st.b;
                pc[0] = st.c;
                restore_arn<P, NA>(ka, v, al, pb, pc);
                if (NEEDF)
                {
                    do
                    {
                        if (!ensure_cov<P>(st))
                        {
                            status = FVB_BAD_RESULT;
                            break;
                        }
                        arn_coefficients<NA>(al, pb, pc, cw);
                        status = arn_pass<Model, P, NA, true, true>(ka, ma, feed, st.m, cw, mo, st.Sig, zero, S, false);
                        if (status != FVB_OK)
                            break;
                        FVB_EVAL_F_ARN()
                    } while (false);
                }
                else
                {
                    arn_coefficients<NA>(al, pb, pc, cw);
                    status = arn_pass<Model, P, NA, true, false>(ka, ma, feed, st.m, cw, mo, st.Sig, zero, S, false);
                }
            }
        }
#undef FVB_EVAL_F_ARN
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
#pragma unroll
        for (int i = 0; i < NA; i++)
#pragma unroll
            for (int j = 0; j <= i; j++)
                dst[(size_t)tri(P + i, P + j) * V] = al.cov[tri(i, j)];
#pragma unroll
        for (int e = 0; e < 2; e++)
            dst[(size_t)tri(P + NA + e, P + NA + e) * V] = pb[e] * pb[e] * pc[e];
#pragma unroll
        for (int i = 0; i < P; i++)
            dst[(size_t)(nCov + i) * V] = st.m[i];
#pragma unroll
        for (int i = 0; i < NA; i++)
            dst[(size_t)(nCov + P + i) * V] = al.mean[i];
#pragma unroll
        for (int e = 0; e < 2; e++)
            dst[(size_t)(nCov + P + NA + e) * V] = pb[e] * pc[e];
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
