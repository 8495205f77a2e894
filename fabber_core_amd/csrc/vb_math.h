/*
 * vb_math.h - scalar and small-matrix building blocks of the voxelwise VB kernels.
 *
 * Everything here is FVB_HD (__host__ __device__) so that the same source is compiled into the
 * gfx950 kernels and into the host library (where the unit tests of the convergence state
 * machine and of the special functions exercise it without a GPU).
 */
#pragma once

#include "../../include/fabber_vb.h"

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FVB_HD __host__ __device__ __forceinline__
#else
#define FVB_HD inline
#endif

namespace fvb
{
constexpr double LOG_2PI = 1.8378770664093454835606594728112;

FVB_HD bool is_finite(double x)
{
    // the reference's test "0 * x == 0 * x" (fwdmodel_linear.cc:134,174): false for NaN and the infinities. Asked of
    // the value's class, not computed as x - x == 0: where x is a product and contraction is allowed the compiler
    // turns x - x into fma(a, b, -x), the product's rounding error, which is not zero (gfx950 fuses across several
    // uses of the product) - every Jacobian entry of the wave kernel read "not finite" when its quotient was first
    // compiled with contraction (round 3).
    return __builtin_isfinite(x);
}

// ---- exp to half an ulp ------------------------------------------------------------------------------------------
// The reference's Jacobian is a central difference with a step of 1e-5 |theta| - 1e-10 where a parameter is exactly 0,
// which is where the exponential model's rates start - so f(theta + d) - f(theta - d) keeps ~5 (there: ~0) digits of
// f, and the rounding of every exp() inside f is what the first linearisations amplify into the trajectory of the
// (chaotic) bi-exponential fit. glibc's exp is good to 0.51 ulp, the device library's to 1 ulp: measured against the
// binary128 ground truth (tests/golden/c*_truth_binary128.npz, tools/measure/c5_truth.py) the kernels' error after
// 10 - 20 iterations was 1.25 - 1.3 x a CPU build's, and the CPU oracle with its exp degraded to 1 ulp lands 1.45 x
// above itself (profiles/r3_exp_accuracy.md). So where the difference quotient is formed from exponentials in the
// FIRST linearisations of a run - the LOG transform and the exponential model in its pointwise passes, which is where
// the amplification is largest - exp is this one:
//   x = (32 m + j) ln2 / 32 + r, |r| <= ln2 / 64;  exp(x) = 2^m 2^(j/32) (1 + p(r)),  2^(j/32) = hi + lo from a table,
// p = expm1 to 2^-67 (degree 7), the reduction's rounding error carried into p, one rounding at the end: 0.52 ulp
// (tests/test_math_host.py against long double). ~20 flops and two table loads: as much as the library's.
#if defined(__HIP_DEVICE_COMPILE__)
__device__
#endif
    static const double FVB_EXP_TABLE[64]
    = {
    0x1.0000000000000p+0, 0x0.0p+0,
    0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55,
    0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54,
    0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54,
    0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55,
    0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54,
    0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54,
    0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55,
    0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55,
    0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54,
    0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55,
    0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59,
    0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56,
    0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55,
    0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54,
    0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54,
    0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54,
    0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55,
    0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55,
    0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54,
    0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54,
    0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57,
    0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56,
    0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54,
    0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54,
    0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56,
    0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55,
    0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56,
    0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55,
    0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54,
    0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54,
    0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54,
      };
// (Inlined into the lane kernels at its 30 sites - the transforms of a re-centre's prologue, the pointwise pass - its
// constants and temporaries stayed live across the streaming loop, which has no register to spare: lane<exp,4> went
// from 6 to 32 spilled registers and from 15.3 to 18.4 ms per launch; as a function called per exponential, 17.3 ms;
// as one out-of-line pointwise pass, 17.8. The kernels therefore use it only in instances that exist for pointwise
// linearisations alone: recentre<Model, P, ACC = true> in the spatial set-up kernel and in the second-sweep kernel
// of a spatial run's first iteration; profiles/r3_exp_accuracy.md.)
// table: where the kernel keeps a copy of FVB_EXP_TABLE (LDS: a read from global memory per exponential would queue
// behind the prefetched samples of a streaming pass and its wait would drain them), or NULL = the table itself
FVB_HD double exp_acc(double x, const double *table = nullptr)
{
    if (!(fabs(x) < 690.0)) // overflow / underflow / NaN: the library's answer
        return exp(x);
    const double *tbl = table ? table : FVB_EXP_TABLE;
    const double INV = 0x1.71547652b82fep+5, C_HI = 0x1.62e42fefa0000p-6, C_LO = 0x1.cf79abc9e3b3ap-45;
    const double K7 = 1.0 / 5040, K6 = 1.0 / 720, K5 = 1.0 / 120, K4 = 1.0 / 24, K3 = 1.0 / 6, K2 = 0.5;
    const double kd = rint(x * INV);
    const int n = (int)kd;
    const double r_hi = __builtin_fma(-kd, C_HI, x); // exact: C_HI has 36 bits, |k| < 2^15
    const double r_lo = kd * C_LO;
    const double r = r_hi - r_lo;
    const double c = (r_hi - r) - r_lo; // what the subtraction rounded away
    const int j = n & 31, m = n >> 5;
    const double t_hi = tbl[2 * j], t_lo = tbl[2 * j + 1];
    // p = expm1(r): r + r^2 (1/2 + r (1/6 + r (1/24 + r (1/120 + r (1/720 + r / 5040))))); remainder r^8 / 40320 < 2^-67
    const double r2 = r * r;
    double q = __builtin_fma(r, K7, K6);
    q = __builtin_fma(r, q, K5);
    q = __builtin_fma(r, q, K4);
    q = __builtin_fma(r, q, K3);
    q = __builtin_fma(r, q, K2);
    const double p = r + __builtin_fma(r2, q, c);
    const double y = t_hi + __builtin_fma(t_hi, p, t_lo);
    return ldexp(y, m);
}

// ---- transforms.h:114-242, transforms.cc:17-25 ----------------------------------------------
FVB_HD double to_model(int tr, double val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_LOG:
        return exp(val);
    case FVB_TRANSFORM_SOFTPLUS:
        return (val < 10) ? log(1 + exp(val)) : val;
    case FVB_TRANSFORM_FRACTIONAL:
        return 1 / (1 + exp(val));
    case FVB_TRANSFORM_ABS:
        return fabs(val);
    default:
        return val;
    }
}
// the same with the half-ulp exp where the transform is the exponential (the pointwise passes, see exp_acc)
FVB_HD double to_model_acc(int tr, double val, const double *table = nullptr)
{
    return tr == FVB_TRANSFORM_LOG ? exp_acc(val, table) : to_model(tr, val);
}
FVB_HD double to_fabber(int tr, double val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_LOG:
        return log(val);
    case FVB_TRANSFORM_SOFTPLUS:
        return (val < 10) ? log(exp(val) - 1) : val;
    case FVB_TRANSFORM_FRACTIONAL:
        return log(1 / val - 1);
    default:
        return val;
    }
}
FVB_HD double to_model_var(int tr, double val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_IDENTITY:
    case FVB_TRANSFORM_FRACTIONAL:
        return val;
    case FVB_TRANSFORM_LOG:
        return exp(val);
    default:
    {
        double d = to_model(tr, sqrt(val)) - to_model(tr, 0.0);
        return d * d;
    }
    }
}
FVB_HD double to_fabber_var(int tr, double val)
{
    switch (tr)
    {
    case FVB_TRANSFORM_IDENTITY:
    case FVB_TRANSFORM_FRACTIONAL:
        return val;
    case FVB_TRANSFORM_LOG:
        return log(val);
    default:
    {
        double d = to_fabber(tr, to_model(tr, 0.0) + sqrt(val));
        return d * d;
    }
    }
}

// ---- tools.cc:87-98: 6-term Lanczos log-gamma -------------------------------------------------
FVB_HD double gammaln(double x)
{
    double total = 1.000000000190015;
    total += 76.18009172947146 / (x + 1);
    total += -86.50532032941677 / (x + 2);
    total += 24.01409824083091 / (x + 3);
    total += -1.231739572450155 / (x + 4);
    total += 0.1208650973866179e-2 / (x + 5);
    total += -0.5395239384953e-5 / (x + 6);
    return log(2.5066282746310005 * total / x) + (x + 0.5) * log(x + 5.5) - x - 5.5;
}

// ---- digamma (MISCMATHS::digamma is outside the reference tree): recurrence to x >= 10, then
// the asymptotic series. fp64 throughout. -------------------------------------------------------
FVB_HD double digamma(double x)
{
    double r = 0;
    // at most 10 steps for the arguments VB produces (c >= 1e-6)
    for (int i = 0; i < 10; i++)
    {
        if (x < 10.0)
        {
            r -= 1.0 / x;
            x += 1.0;
        }
    }
    const double f = 1.0 / (x * x);
    const double t = f
        * (-1.0 / 12.0
              + f * (1.0 / 120.0
                        + f * (-1.0 / 252.0
                                  + f * (1.0 / 240.0
                                            + f * (-1.0 / 132.0
                                                      + f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}

// ---- packed symmetric matrices ----------------------------------------------------------------
// Lower triangle, row-major: element (i,j), j <= i, lives at i(i+1)/2 + j - the same order as the
// reference's MVN image rows (dist_mvn.cc:419-425), so results are stored without reshuffling.
FVB_HD constexpr int tri(int i, int j)
{
    return (i >= j) ? (i * (i + 1) / 2 + j) : (j * (j + 1) / 2 + i);
}

// LDL^T factorisation + inverse + log|det| of a packed symmetric PxP matrix, fully unrolled for
// compile-time P so that every element stays in a register. Plays the role of NEWMAT's .i() and
// LogDeterminant() (dist_mvn.cc:213,248; noisemodel_white.cc:308,390,422): like the LU behind
// those, only an exactly zero pivot counts as singular; non-finite values propagate.
// Returns false if singular. logabs/sign may be ignored by the caller.
template <int P>
FVB_HD bool ldl_inverse(const double (&a)[P * (P + 1) / 2], double (&inv)[P * (P + 1) / 2], double &logabs, int &sign)
{
    // In place, by sweeping every pivot (Goodnight 1979): sweeping k maps W_kk -> -1/d,
    // W_ik -> W_ik / d, W_ij -> W_ij - W_ik W_jk / d (d = W_kk); with all pivots swept W = -A^-1.
    // The pivots are the D of the LDL^T factorisation. Needs no storage beyond the result itself,
    // which matters more than the flop count here: the kernels hold the whole voxel state in
    // registers while they invert.
    bool ok = true;
    sign = 1;
    // log|det| = sum of log|pivot| as ONE logarithm: the pivots' mantissas are multiplied ([0.5, 1) each: no
    // under- or overflow for P <= 16), their exponents added (round 2: a logarithm per pivot, ~60 dependent
    // instructions each, in the kernels that evaluate F). A zero, infinite or NaN pivot comes out as log() has it.
    double mant = 1.0;
    int expo = 0;
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; i++)
        inv[i] = a[i];
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        const double d = inv[tri(k, k)];
        if (d == 0.0)
            ok = false;
        if (d < 0)
            sign = -sign;
        int e;
        mant *= frexp(fabs(d), &e);
        expo += e;
        const double rd = 1.0 / d;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            if (i == k)
                continue;
            const double cik = inv[tri(i, k)] * rd;
#pragma unroll
            for (int j = 0; j <= i; j++)
            {
                if (j == k)
                    continue;
                inv[tri(i, j)] -= cik * inv[tri(j, k)];
            }
        }
#pragma unroll
        for (int i = 0; i < P; i++)
            if (i != k)
                inv[tri(i, k)] *= rd;
        inv[tri(k, k)] = -rd;
    }
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; i++)
        inv[i] = -inv[i];
    logabs = log(mant) + expo * 0.6931471805599453;
    return ok;
}

// MVNDist::GetCovariance / GetPrecisions semantics (dist_mvn.cc:197-265): invert; if singular add
// 1e-10 to the diagonal and retry; a second failure is the NEWMAT exception the voxel loop
// catches (inference_vb.cc:537) -> returns false.
template <int P>
FVB_HD bool mvn_invert(const double (&a)[P * (P + 1) / 2], double (&inv)[P * (P + 1) / 2], double &logabs, int &sign)
{
    if (ldl_inverse<P>(a, inv, logabs, sign))
        return true;
    double tmp[P * (P + 1) / 2];
#pragma unroll
    for (int i = 0; i < P * (P + 1) / 2; i++)
        tmp[i] = a[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        tmp[tri(i, i)] += 1e-10;
    double l2;
    int s2;
    return ldl_inverse<P>(tmp, inv, l2, s2);
}

// ---- convergence detectors (convergence.cc:34-378) as one POD state machine --------------------
struct ConvState
{
    int32_t type;
    int32_t its, max_its;
    int32_t trials, max_trials;
    int32_t save, revert, trialmode, lm;
    double prev_f, min_fchange;
    double alpha, alphastart, alphamax;
};

FVB_HD void conv_init(ConvState &c, int type, int max_iterations, int max_trials, double min_fchange)
{
    c.type = type;
    c.max_its = max_iterations + (type == FVB_CONV_TRIALMODE ? 1 : 0); // convergence.cc:145
    c.max_trials = max_trials;
    c.min_fchange = min_fchange;
}

// Reset(F = -99e99): convergence.cc:63-67, :80-86, :152-160, :262-276
FVB_HD void conv_reset(ConvState &c)
{
    c.its = 0;
    c.prev_f = -99e99;
    c.save = (c.type == FVB_CONV_TRIALMODE || c.type == FVB_CONV_LM) ? 1 : 0;
    c.revert = 0;
    c.trials = 0;
    c.trialmode = 0;
    c.lm = 0;
    c.alphastart = 1e-6;
    c.alpha = 0.0;
    c.alphamax = 1e6;
}

FVB_HD bool conv_need_save(const ConvState &c)
{
    return c.save != 0; // maxits: never true (ConvergenceDetector::NeedSave)
}
FVB_HD bool conv_need_revert(const ConvState &c)
{
    return c.revert != 0;
}
// LMalpha() returns float (convergence.h:73,236)
FVB_HD double conv_lm_alpha(const ConvState &c)
{
    return (c.type == FVB_CONV_LM) ? (double)(float)c.alpha : 0.0;
}

FVB_HD bool conv_test_counting(ConvState &c)
{
    ++c.its;
    return c.its >= c.max_its;
}
FVB_HD bool conv_test_fchange(ConvState &c, double F)
{
    double diff = F - c.prev_f;
    c.prev_f = F;
    diff = diff > 0 ? diff : -diff;
    if (diff < c.min_fchange)
        return true;
    return conv_test_counting(c);
}

FVB_HD bool conv_test(ConvState &c, double F)
{
    switch (c.type)
    {
    case FVB_CONV_MAXITS:
        return conv_test_counting(c);
    case FVB_CONV_FCHANGE:
        return conv_test_fchange(c, F);
    case FVB_CONV_FREDUCE: // convergence.cc:117-131
    {
        double diff = F - c.prev_f;
        if (diff < 0)
        {
            c.revert = 1;
            return true;
        }
        return conv_test_fchange(c, F);
    }
    case FVB_CONV_TRIALMODE: // convergence.cc:162-243
    {
        double diff = F - c.prev_f;
        if (!c.trialmode)
        {
            if (diff < 0)
            {
                c.its = 1;
                c.trials = 1;
                c.trialmode = 1;
                c.revert = 1;
                c.save = 0;
                return false;
            }
            double absdiff = diff > 0 ? diff : -diff;
            if (absdiff < c.min_fchange)
            {
                c.revert = 0;
                c.save = 0;
                return true;
            }
            c.save = 1;
            c.revert = 0;
            c.prev_f = F;
            ++c.its;
            return c.its >= c.max_its;
        }
        ++c.trials;
        if (diff > 0)
        {
            if (diff < c.min_fchange)
            {
                c.revert = 0;
                c.save = 0;
                return true;
            }
            c.trialmode = 0;
            c.trials = 0;
            c.save = 1;
            c.revert = 0;
            c.prev_f = F;
            return false;
        }
        if (c.trials >= c.max_trials)
        {
            c.save = 0;
            c.revert = 1;
            return true;
        }
        c.save = 0;
        c.revert = 0;
        return false;
    }
    default: // FVB_CONV_LM, convergence.cc:278-378
    {
        double diff = F - c.prev_f;
        double absdiff = diff < 0 ? -diff : diff;
        if (!c.lm)
        {
            if (diff < 0)
            {
                c.lm = 1;
                c.revert = 1;
                c.alpha = c.alphastart;
                return false;
            }
            if (absdiff < c.min_fchange)
            {
                c.revert = 0;
                return true;
            }
            if (c.its >= c.max_its)
            {
                c.revert = 0;
                return true;
            }
            c.prev_f = F;
            ++c.its;
            return false;
        }
        if (diff > 0)
        {
            if (c.alpha == c.alphastart)
                c.lm = 0;
            else
                c.alpha /= 10;
            c.revert = 0;
            c.prev_f = F;
            ++c.its;
            return false;
        }
        if (c.alpha >= c.alphamax)
        {
            c.revert = 1;
            return true;
        }
        if (c.its >= c.max_its)
        {
            c.revert = 0;
            return true;
        }
        c.alpha *= 10;
        c.revert = 1;
        return false;
    }
    }
}

} // namespace fvb
