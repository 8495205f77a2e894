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

// The model prediction feeds a central difference with a step of 1e-5 |theta| (floor 1e-10):
// a last-bit change of f is amplified by |f| / (2 delta) in the Jacobian (1e-6 relative and more
// when a parameter is small). To keep J as close as possible to what the reference's CPU code
// computes, model bodies and the differencing are compiled WITHOUT fused multiply-add
// contraction, i.e. with the same operation sequence as the reference's C++.
#define FVB_NO_CONTRACT _Pragma("clang fp contract(off)")

namespace fvb
{
struct ModelArgs
{
    int32_t iopt0;        // poly: degree; exp: number of exponentials
    double dopt0;         // exp: dt
    const double *design; // linear: [T][P] row-major
};

// fwdmodel_poly.cc:62-80. The reference accumulates i^n in an int.
template <int P>
struct PolyModel
{
    static constexpr int model_id = FVB_MODEL_POLY;
    static FVB_HD double eval(const ModelArgs &, int t, const double (&p)[P])
    {
        FVB_NO_CONTRACT
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
    static constexpr int model_id = FVB_MODEL_LINEAR;
    static FVB_HD double eval(const ModelArgs &a, int t, const double (&p)[P])
    {
        FVB_NO_CONTRACT
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
    static_assert(P % 2 == 0, "exp model has 2 parameters per exponential");
    static constexpr int model_id = FVB_MODEL_EXP;
    static FVB_HD double eval(const ModelArgs &a, int t, const double (&p)[P])
    {
        FVB_NO_CONTRACT
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
};

// Runtime-P evaluation used by the generic (wave-per-voxel, post-processing) paths.
FVB_HD double eval_model_runtime(int model, const ModelArgs &a, int P, int t, const double *p)
{
    FVB_NO_CONTRACT
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
