/*
 * vb_models.h - device bodies of the forward models that can be evaluated inside a kernel.
 *
 * A model body is a stateless struct with
 *     static double eval(const ModelArgs &a, int t, const double (&p)[P])
 * returning the model prediction at timepoint index t (0-based) for MODEL-space parameters p
 * (i.e. after FwdModel::EvaluateFabber has applied the parameter transforms, fwdmodel.cc:365-382),
 * and
 *     static void init_posterior(const ModelArgs &a, double data_max, double (&means)[P])
 * mirroring FwdModel::InitVoxelPosterior (model-space means).
 *
 * The kernels call eval() 2P+1 times per timepoint with parameter vectors that differ in one
 * entry; because P is a template constant and everything is inlined, the compiler shares the
 * sub-expressions that do not depend on the perturbed entry (for the exponential model: 3 N
 * instead of (4N+1) N exp() per timepoint) without changing any value that is computed.
 */
#pragma once

#include "vb_math.h"

// The model prediction feeds a central difference with a step of 1e-5 |theta| (floor 1e-10): a last-bit change of f
// is amplified by |f| / (2 delta) in the Jacobian (1e-6 relative and more when a parameter is small). Rounds 1 and 2
// therefore compiled the model bodies and the differencing WITHOUT fused multiply-add contraction - the operation
// sequence of the reference's C++ on a CPU without FMA. Round 3 measured both builds against the binary128 ground
// truth of the bi-exponential fit (tools/measure/c3_truth.py; profiles/r3_c3_truth_contract.json, r3_fma_contraction.md):
// with contraction the lane kernel's error is lower at every iteration (65 536 voxels, median after 3 iterations
// 1.01e-4 against 1.30e-4; 75.7 % of the final posteriors within 1e-4 of the truth against 75.4 %; the two CPU builds:
// 73.8 % without FMA, 74.6 % with) and the C3 run takes 14.4 ms instead of 15.1. An FMA rounds once where the plain
// sequence rounds twice: it is the more accurate evaluation of the same expression, and the reference itself is built
// with whatever its compiler does (-ffp-contract=fast is gcc's default). So: contraction is left to the compiler.
// FVB_STRICT_MODELS restores the old build for comparisons.
#ifdef FVB_STRICT_MODELS
#define FVB_MODEL_FP _Pragma("clang fp contract(off)")
#else
#define FVB_MODEL_FP _Pragma("clang fp contract(fast)")
#endif

namespace fvb
{
struct ModelArgs
{
    int32_t iopt0;        // poly: degree; exp: number of exponentials
    double dopt0;         // exp: dt
    const double *design; // linear: [T][P] row-major
    // host-evaluated models (HostLinModel below): THIS voxel's linearisation, g [lin_T] then J [lin_T][P], as
    // the host's model code computed it about the centre the caller is working with
    const double *lin;
    int32_t lin_T;
    // the kernel's copy of the table of exp_acc (vb_math.h) in LDS, or NULL (kernels built with recentre<..., ACC>)
    const double *exp_table;
};

// A "sweep" produces, for t = 0, 1, 2, ... in order, the 2P + 1 predictions the central
// difference needs: g = f(tp) and f2[i] / f3[i] = f(tp with entry i replaced by tp2[i] / tp3[i]).
// The default sweep just calls Model::eval; a model may provide its own (ExpModel below) when
// walking along t lets it share work between timepoints.
template <class Model, int P>
struct PointwiseSweep
{
    FVB_HD void init(const ModelArgs &, const double (&)[P], const double (&)[P], const double (&)[P])
    {
    }
    // set_precise(true) asks for the most precise evaluation a sweep has (wave-uniform); the
    // pointwise sweep has only one.
    FVB_HD void set_precise(bool)
    {
    }
    FVB_HD void eval(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
        double &g, double (&f2)[P], double (&f3)[P])
    {
        FVB_MODEL_FP
        g = Model::eval(ma, t, tp);
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            double q[P];
#pragma unroll
            for (int j = 0; j < P; j++)
                q[j] = tp[j];
            q[i] = tp2[i];
            f2[i] = Model::eval(ma, t, q);
            q[i] = tp3[i];
            f3[i] = Model::eval(ma, t, q);
        }
    }
    // g and the central-difference Jacobian row J_i = (f2_i - f3_i) * rden_i, rden_i = 1 / (c2_i - c3_i)
    // (fwdmodel_linear.cc:170). Sweeps of models with known structure evaluate the SAME difference
    // quotient - same perturbed parameter values - without forming the two nearly equal sums first.
    FVB_HD void eval_jac(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
        const double (&rden)[P], double &g, double (&J)[P])
    {
        FVB_MODEL_FP
        double f2[P], f3[P];
        eval(ma, t, tp, tp2, tp3, g, f2, f3);
#pragma unroll
        for (int i = 0; i < P; i++)
            J[i] = (f2[i] - f3[i]) * rden[i];
    }
    // The two entry points of the lane kernels' streaming pass (vb_lane_kernel.h, recentre_tiles):
    // step_fast<EXACT> inside the unrolled main loop, where the position of a timepoint within its
    // block of FVB_EXP_RESYNC is a compile-time constant (EXACT: first of the block) and the sweep is
    // known not to be in its precise mode; step_any wherever that is only known at run time.
    template <bool EXACT>
    FVB_HD void step_fast(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P],
        const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P])
    {
        eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
    }
    FVB_HD void step_any(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P],
        const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P], bool)
    {
        eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
    }
};

// Sweep for a model that is linear in each of its parameters, f = sum_n basis_n(t) p_n (the
// polynomial and the design-matrix model): the perturbed predictions are f(tp) plus the basis
// function times the perturbation, one multiply-add each instead of a full evaluation -
// f2_i = g + b_i (tp2_i - tp_i). In exact arithmetic that IS f(tp with tp2_i); in floating point
// it differs from the full sum by its rounding, ~1e-16 |g|, i.e. by as much as two evaluations of
// the reference's own sum differ under re-association. Pointwise when set_precise(true).
template <class Model, int P>
struct LinearInParameterSweep
{
    double d2[P], d3[P];
    bool precise;
    FVB_HD void init(const ModelArgs &, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P])
    {
        precise = false;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            d2[i] = tp2[i] - tp[i];
            d3[i] = tp3[i] - tp[i];
        }
    }
    FVB_HD void set_precise(bool p)
    {
        precise = p;
    }
    FVB_HD void eval(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
        double &g, double (&f2)[P], double (&f3)[P])
    {
        FVB_MODEL_FP
        if (precise) // wave-uniform
        {
            PointwiseSweep<Model, P> pw;
            pw.eval(ma, t, tp, tp2, tp3, g, f2, f3);
            return;
        }
        g = Model::eval(ma, t, tp);
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            const double b = Model::basis(ma, t, i);
            f2[i] = g + b * d2[i];
            f3[i] = g + b * d3[i];
        }
    }
    // f2_i - f3_i = b_i (tp2_i - tp3_i) exactly: the difference quotient is the basis function
    // times a per-parameter constant (1 up to rounding for an untransformed parameter). The two
    // sums g + b d2 and g + b d3 would only add their own rounding, ~1e-16 |g| / (c2 - c3), to it.
    FVB_HD void eval_jac(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
        const double (&rden)[P], double &g, double (&J)[P])
    {
        FVB_MODEL_FP
        if (precise) // wave-uniform
        {
            PointwiseSweep<Model, P> pw;
            pw.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
            return;
        }
        g = Model::eval(ma, t, tp);
#pragma unroll
        for (int i = 0; i < P; i++)
            J[i] = Model::basis(ma, t, i) * ((tp2[i] - tp3[i]) * rden[i]); // (the factor does not depend on t)
    }
    template <bool EXACT>
    FVB_HD void step_fast(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P],
        const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P])
    {
        FVB_MODEL_FP
        g = Model::eval(ma, t, tp);
#pragma unroll
        for (int i = 0; i < P; i++)
            J[i] = Model::basis(ma, t, i) * ((tp2[i] - tp3[i]) * rden[i]);
    }
    FVB_HD void step_any(const ModelArgs &ma, int t, const double (&tp)[P], const double (&tp2)[P],
        const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P], bool prec)
    {
        precise = prec;
        eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
    }
};

// fwdmodel_poly.cc:62-80. The reference accumulates i^n in an int.
template <int P>
struct PolyModel
{
    static constexpr bool host_evaluated = false;
    static constexpr int model_id = FVB_MODEL_POLY;
    typedef LinearInParameterSweep<PolyModel<P>, P> Sweep;
    // d f / d p_n at timepoint t (the model is linear in every parameter)
    static FVB_HD double basis(const ModelArgs &, int t, int n)
    {
        int pw = 1;
        for (int k = 0; k < n; k++)
            pw *= t + 1;
        return (double)pw;
    }
    static FVB_HD double eval(const ModelArgs &, int t, const double (&p)[P])
    {
        FVB_MODEL_FP
        const int i = t + 1;
        double res = 0;
        int pw = 1;
#pragma unroll
        for (int n = 0; n < P; n++)
        {
            res += p[n] * pw;
            pw *= i;
        }
        return res;
    }
    static FVB_HD void init_posterior(const ModelArgs &, double, double (&)[P])
    {
    }
    static constexpr bool needs_data_max = false;
};

// fwdmodel_linear.cc:92-96 with m_centre = 0, m_offset = 0 (fwdmodel_linear.cc:66-69)
template <int P>
struct LinearModel
{
    static constexpr bool host_evaluated = false;
    static constexpr int model_id = FVB_MODEL_LINEAR;
    typedef LinearInParameterSweep<LinearModel<P>, P> Sweep;
    static FVB_HD double basis(const ModelArgs &a, int t, int n)
    {
        return a.design[(size_t)t * P + n];
    }
    static FVB_HD double eval(const ModelArgs &a, int t, const double (&p)[P])
    {
        FVB_MODEL_FP
        const double *row = a.design + (size_t)t * P; // wave-uniform address: scalar loads
        double s = 0;
#pragma unroll
        for (int j = 0; j < P; j++)
            s += row[j] * (p[j] - 0.0);
        return s + 0.0;
    }
    static FVB_HD void init_posterior(const ModelArgs &, double, double (&)[P])
    {
    }
    static constexpr bool needs_data_max = false;
};

// examples/fwdmodel_exp.cc:65-91: sum_i amp_i exp(-r_i k dt), parameters (amp1, r1, amp2, r2...)
template <int P>
struct ExpModel
{
    static constexpr bool host_evaluated = false;
    static_assert(P % 2 == 0, "exp model has 2 parameters per exponential");
    static constexpr int model_id = FVB_MODEL_EXP;
    static FVB_HD double eval(const ModelArgs &a, int t, const double (&p)[P])
    {
        FVB_MODEL_FP
        const double tt = double(t) * a.dopt0;
        double res = 0;
#pragma unroll
        for (int i = 0; i < P / 2; i++)
        {
            double val = p[2 * i] * exp(-p[2 * i + 1] * tt);
            res += val;
        }
        return res;
    }
    static FVB_HD void init_posterior(const ModelArgs &, double data_max, double (&means)[P])
    {
#pragma unroll
        for (int i = 0; i < P / 2; i++)
            means[2 * i] = data_max / (P / 2 + i);
    }
    static constexpr bool needs_data_max = true;

    // The 2P + 1 parameter vectors of a central difference contain only 3 distinct rates per
    // exponential (r, r + d, r - d), i.e. 3 P/2 distinct exp(-r t dt) per timepoint - and along
    // t each of them is a geometric sequence. The sweep evaluates them exactly (the expression
    // of eval above) every FVB_EXP_RESYNC timepoints and advances them by one multiplication
    // with exp(-r dt) in between: at most FVB_EXP_RESYNC - 1 extra roundings (~1e-16 each) on a
    // value whose own evaluation carries one. fp64 exp is ~25 of the ~30 instructions per
    // exponential and the kernels are VALU-bound, see DESIGN.md. FVB_EXP_RESYNC = 1 is the
    // pointwise evaluation.
#ifndef FVB_EXP_RESYNC
#define FVB_EXP_RESYNC 8
#endif
    struct Sweep
    {
        static constexpr int N = P / 2;
        double e0[N], e2[N], e3[N], s0[N], s2[N], s3[N];
        bool precise; // pointwise evaluation at every t (wave-uniform)
        FVB_HD void init(const ModelArgs &a, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P])
        {
            FVB_MODEL_FP
            precise = false;
            if (FVB_EXP_RESYNC > 1)
            {
#pragma unroll
                for (int i = 0; i < N; i++)
                {
                    s0[i] = exp(-tp[2 * i + 1] * a.dopt0);
                    s2[i] = exp(-tp2[2 * i + 1] * a.dopt0);
                    s3[i] = exp(-tp3[2 * i + 1] * a.dopt0);
                }
            }
        }
        FVB_HD void set_precise(bool p)
        {
            precise = p;
        }
        // exp(-rate t dt) for the three rates of every exponential at timepoint t
        FVB_HD void advance(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P])
        {
            FVB_MODEL_FP
            if (FVB_EXP_RESYNC <= 1 || precise || (t % FVB_EXP_RESYNC) == 0) // wave-uniform
            {
                const double tt = double(t) * a.dopt0;
#pragma unroll
                for (int i = 0; i < N; i++)
                {
                    e0[i] = exp(-tp[2 * i + 1] * tt);
                    e2[i] = exp(-tp2[2 * i + 1] * tt);
                    e3[i] = exp(-tp3[2 * i + 1] * tt);
                }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < N; i++)
                {
                    e0[i] *= s0[i];
                    e2[i] *= s2[i];
                    e3[i] *= s3[i];
                }
            }
        }
        FVB_HD void eval(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
            double &g, double (&f2)[P], double (&f3)[P])
        {
            FVB_MODEL_FP
            advance(a, t, tp, tp2, tp3);
            combine(tp, tp2, tp3, g, f2, f3);
        }
        // exp(-rate t dt) evaluated as eval() does, for the three rates of every exponential
        FVB_HD void resync(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P])
        {
            FVB_MODEL_FP
            const double tt = double(t) * a.dopt0;
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                e0[i] = exp(-tp[2 * i + 1] * tt);
                e2[i] = exp(-tp2[2 * i + 1] * tt);
                e3[i] = exp(-tp3[2 * i + 1] * tt);
            }
        }
        // the same with the half-ulp exp (vb_math.h, exp_acc): the pointwise passes - the first linearisations of a
        // run, whose rounding the fit amplifies most
        FVB_HD void resync_acc(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P])
        {
            FVB_MODEL_FP
            const double tt = double(t) * a.dopt0;
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                e0[i] = exp_acc(-tp[2 * i + 1] * tt, a.exp_table);
                e2[i] = exp_acc(-tp2[2 * i + 1] * tt, a.exp_table);
                e3[i] = exp_acc(-tp3[2 * i + 1] * tt, a.exp_table);
            }
        }
        FVB_HD void multiply()
        {
            FVB_MODEL_FP
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                e0[i] *= s0[i];
                e2[i] *= s2[i];
                e3[i] *= s3[i];
            }
        }
        // the 2P + 1 predictions from the current exponentials
        FVB_HD void combine(const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P], double &g,
            double (&f2)[P], double (&f3)[P])
        {
            FVB_MODEL_FP
            // the sums below add the terms in the order of eval(): res = 0; res += amp_j * exp_j
            double val[N];
#pragma unroll
            for (int j = 0; j < N; j++)
                val[j] = tp[2 * j] * e0[j];
            // (eval() starts from res = 0: 0 + x is x, bit for bit, except that a -0 term would
            // become +0 - the sums here start from their first term and save the additions)
            double res = val[0];
#pragma unroll
            for (int j = 1; j < N; j++)
                res += val[j];
            g = res;
#pragma unroll
            for (int i = 0; i < N; i++)
            {
                double a2 = (i == 0) ? tp2[0] * e0[0] : val[0];
                double a3 = (i == 0) ? tp3[0] * e0[0] : val[0];
                double r2 = (i == 0) ? tp[0] * e2[0] : val[0];
                double r3 = (i == 0) ? tp[0] * e3[0] : val[0];
#pragma unroll
                for (int j = 1; j < N; j++)
                {
                    a2 += (j == i) ? tp2[2 * i] * e0[i] : val[j];
                    a3 += (j == i) ? tp3[2 * i] * e0[i] : val[j];
                    r2 += (j == i) ? tp[2 * i] * e2[i] : val[j];
                    r3 += (j == i) ? tp[2 * i] * e3[i] : val[j];
                }
                f2[2 * i] = a2;
                f3[2 * i] = a3;
                f2[2 * i + 1] = r2;
                f3[2 * i + 1] = r3;
            }
        }
        // The difference quotient is formed from the two perturbed SUMS, as the reference forms it,
        // not from the one term that differs (f2 - f3 = (amp2 - amp3) e0 for an amplitude,
        // amp (e2 - e3) for a rate): the rounding of those sums, ~1e-16 |g| / (c2 - c3) of noise in
        // J, is part of the reference's behaviour on this model. Measured on the bi-exponential C3
        // problem with the CPU oracle (tests/measure/structured_j.py): with the noise-free quotient
        // 0.60 % of the voxels end in a non-finite prediction (most of them in iterations 8-9)
        // against 0.14 % with the reference's arithmetic - the kernel saved 20 % of its time and
        // failed twice as many voxels as the reference, so it is not done.
        FVB_HD void eval_jac(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
            const double (&rden)[P], double &g, double (&J)[P])
        {
            FVB_MODEL_FP
            double f2[P], f3[P];
            eval(a, t, tp, tp2, tp3, g, f2, f3);
#pragma unroll
            for (int i = 0; i < P; i++)
                J[i] = (f2[i] - f3[i]) * rden[i];
        }
        // (see PointwiseSweep::step_fast) EXACT = t is a multiple of FVB_EXP_RESYNC
        template <bool EXACT>
        FVB_HD void step_fast(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P],
            const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P])
        {
            FVB_MODEL_FP
            if (EXACT || FVB_EXP_RESYNC <= 1)
                resync(a, t, tp, tp2, tp3);
            else
                multiply();
            double f2[P], f3[P];
            combine(tp, tp2, tp3, g, f2, f3);
#pragma unroll
            for (int i = 0; i < P; i++)
                J[i] = (f2[i] - f3[i]) * rden[i];
        }
        // pointwise with the half-ulp exp: recentre<Model, P, ACC = true> (vb_lane_kernel.h), i.e. only in kernels that
        // were built for the pointwise passes
        FVB_HD void step_acc(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P], const double (&tp3)[P],
            const double (&rden)[P], double &g, double (&J)[P])
        {
            FVB_MODEL_FP
            resync_acc(a, t, tp, tp2, tp3);
            double f2[P], f3[P];
            combine(tp, tp2, tp3, g, f2, f3);
#pragma unroll
            for (int i = 0; i < P; i++)
                J[i] = (f2[i] - f3[i]) * rden[i];
        }
        FVB_HD void step_any(const ModelArgs &a, int t, const double (&tp)[P], const double (&tp2)[P],
            const double (&tp3)[P], const double (&rden)[P], double &g, double (&J)[P], bool prec)
        {
            FVB_MODEL_FP
            if (FVB_EXP_RESYNC <= 1 || prec || (t % FVB_EXP_RESYNC) == 0) // wave-uniform
                resync(a, t, tp, tp2, tp3);
            else
                multiply();
            double f2[P], f3[P];
            combine(tp, tp2, tp3, g, f2, f3);
#pragma unroll
            for (int i = 0; i < P; i++)
                J[i] = (f2[i] - f3[i]) * rden[i];
        }
    };
};

// A model that exists only as host code (a FwdModel subclass of a model library written for the reference,
// cfg.model = FVB_MODEL_HOSTJAC): the host evaluates it - the 2P + 1 predictions of
// LinearizedFwdModel::ReCentre (fwdmodel_linear.cc:126-182) - and hands over g and J; the "sweep" of such
// a model reads them back, so that everything built on sweeps (the streaming moments, the directly summed
// residual) works unchanged. Which linearisation ma.lin points to (the one about the centre being asked for)
// is the calling kernel's business.
template <int P>
struct HostLinModel
{
    static constexpr bool host_evaluated = true;
    static constexpr int model_id = FVB_MODEL_HOSTJAC;
    static constexpr bool needs_data_max = false;
    static FVB_HD void init_posterior(const ModelArgs &, double, double (&)[P])
    {
    }
    // (pointwise evaluation at arbitrary parameters is what such a model cannot do on the device; only the
    // wave-cooperative rescue of the tile-fed lane kernels asks for it, and no host-model route uses those)
    static FVB_HD double eval(const ModelArgs &, int, const double (&)[P])
    {
        return __builtin_nan("");
    }
    struct Sweep
    {
        const double *g_row, *j_rows;
        FVB_HD void init(const ModelArgs &a, const double (&)[P], const double (&)[P], const double (&)[P])
        {
            g_row = a.lin;
            j_rows = a.lin + a.lin_T;
        }
        FVB_HD void set_precise(bool)
        {
        }
        FVB_HD void eval_jac(const ModelArgs &, int t, const double (&)[P], const double (&)[P], const double (&)[P],
            const double (&)[P], double &g, double (&J)[P])
        {
            g = g_row[t];
#pragma unroll
            for (int i = 0; i < P; i++)
                J[i] = j_rows[(size_t)t * P + i];
        }
    };
};

// Runtime-P evaluation used by the generic (wave-per-voxel, post-processing) paths.
FVB_HD double eval_model_runtime(int model, const ModelArgs &a, int P, int t, const double *p)
{
    FVB_MODEL_FP
    switch (model)
    {
    case FVB_MODEL_POLY:
    {
        const int i = t + 1;
        double res = 0;
        int pw = 1;
        for (int n = 0; n < P; n++)
        {
            res += p[n] * pw;
            pw *= i;
        }
        return res;
    }
    case FVB_MODEL_LINEAR:
    {
        const double *row = a.design + (size_t)t * P;
        double s = 0;
        for (int j = 0; j < P; j++)
            s += row[j] * (p[j] - 0.0);
        return s + 0.0;
    }
    case FVB_MODEL_EXP:
    {
        const double tt = double(t) * a.dopt0;
        double res = 0;
        for (int i = 0; i < P / 2; i++)
            res += p[2 * i] * exp(-p[2 * i + 1] * tt);
        return res;
    }
    default:
        return NAN;
    }
}

} // namespace fvb
