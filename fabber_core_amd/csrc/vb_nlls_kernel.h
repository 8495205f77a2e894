/*
 * vb_nlls_kernel.h - non-linear least squares (method=nlls), one lane per voxel.
 *
 * NLLSInferenceTechnique::DoCalculations (inference_nlls.cc:94-214) minimises, per voxel,
 * cf(p) = |y - f(p)|^2 (NLLSCF::cf, :223-235) with gradient -2 J'(y - f) (:237-256) and the
 * Gauss-Newton Hessian 2 J'J (:258-290), J from LinearizedFwdModel::ReCentre, masked timepoints
 * dropped from y, f and J. The minimiser itself is FSL MISCMATHS `nonlin` (NL_LM), which is not
 * part of the reference's tree; it is restated here from its published algorithm:
 *
 *   cf = cf(p); lambda = 0.1
 *   up to 200 times:
 *       H = 2 J'J with the diagonal damped - Levenberg (default): H_ii + lambda;
 *                                            Levenberg-Marquardt (option lm): H_ii (1 + lambda)
 *       step = -H^-1 grad ; ncf = cf(p + step)
 *       if the solve worked and ncf < cf: p += step, lambda /= 10,
 *            stop if 2 |cf - ncf| <= 1e-8 (|cf| + |ncf| + eps); cf = ncf; new grad / H
 *       else: lambda *= 10, stop if lambda > 1e20 (grad / H kept)
 *
 * The moments pass of the VB lane kernel (recentre(), vb_lane_kernel.h) delivers everything one
 * iteration needs in ONE sweep over the series: at the trial point p + step it returns
 * s = |y - f|^2 = ncf together with J'J and J'r, which become the next iteration's H and grad
 * when the step is accepted (the reference evaluates the model once for cf and 2P + 1 times more
 * for grad/hess: the same 2P + 1 evaluations, so no work is wasted on accepted steps).
 *
 * Result (inference_nlls.cc:150-207): means = p, precisions = J'J / (cf / (N_samples - P)) with
 * diagonal entries below 1e-6 raised to 1e-6, covariance = inverse; where the model or the
 * inverse fails: precisions 1e-12 I.
 */
#pragma once

#include "vb_lane_kernel.h"
#include "vb_wave_kernel.h"

#include <cfloat>

namespace fvb
{
struct NllsArgs
{
    KernelArgs ka;
    fvb_nlls nl;
};

#if defined(__HIPCC__)

template <class Model, int P>
__global__ __launch_bounds__(64, 2) void nlls_lane_kernel(const NllsArgs na)
{
    constexpr int PT = P * (P + 1) / 2;
    const KernelArgs &ka = na.ka;
    const int v = blockIdx.x * 64 + threadIdx.x;
    const size_t V = (size_t)ka.cfg.n_voxels;
    if (v >= ka.cfg.n_voxels)
        return;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    // starting estimate, Fabber space (inference_nlls.cc:131-132: the means of the 'posterior'
    // from HardcodedInitialDists or fwd-inital-posterior are used as they are)
    double par[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        par[i] = ka.cfg.post_mean[i];

    Moments<P> cur;
    int status = recentre<Model, P>(ka, ma, v, par, cur, true);
    double cf = cur.s;
    double lambda = na.nl.lambda0;
    int niter = 0;
    bool running = (status == FVB_OK);
    while (running && niter < na.nl.max_iterations)
    {
        niter++;
        double H[PT], Hinv[PT];
#pragma unroll
        for (int i = 0; i < PT; i++)
            H[i] = 2.0 * cur.A[i];
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            if (na.nl.lm)
                H[tri(i, i)] *= (1.0 + lambda);
            else
                H[tri(i, i)] += lambda;
        }
        double logabs;
        int sign;
        bool solved = ldl_inverse<P>(H, Hinv, logabs, sign);
        double trial[P];
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            double step = 0;
#pragma unroll
            for (int j = 0; j < P; j++)
                step += Hinv[tri(i, j)] * (2.0 * cur.u[j]); // -H^-1 grad, grad = -2 J'r
            solved = solved && is_finite(step);
            trial[i] = par[i] + step;
        }
        if (!solved) // keep the model evaluation below away from non-finite parameters
        {
#pragma unroll
            for (int i = 0; i < P; i++)
                trial[i] = par[i];
        }
        Moments<P> tr;
        const int st = recentre<Model, P>(ka, ma, v, trial, tr, true);
        const double ncf = tr.s;
        if (solved && ncf < cf) // (a non-finite ncf compares false, as in the reference)
        {
#pragma unroll
            for (int i = 0; i < P; i++)
                par[i] = trial[i];
            cur = tr;
            lambda *= 0.1;
            const bool converged = 2.0 * fabs(cf - ncf) <= na.nl.cf_tolerance * (fabs(cf) + fabs(ncf) + DBL_EPSILON);
            cf = ncf;
            if (st != FVB_OK) // the next gradient's ReCentre throws (fwdmodel_linear.cc:134-181)
            {
                status = st;
                running = false;
            }
            if (converged)
                running = false;
        }
        else
        {
            lambda *= 10.0;
            if (lambda > na.nl.lambda_max)
                running = false;
        }
    }

    // ---- the NLLS precision (inference_nlls.cc:160-184) ----
    double cov[PT];
    bool fallback = (status != FVB_OK);
    if (!fallback)
    {
        const double mse = cf / (double)(ka.n_unmasked - P);
        double prec[PT];
        bool finite = true;
#pragma unroll
        for (int i = 0; i < PT; i++)
        {
            prec[i] = cur.A[i] / mse;
            finite = finite && is_finite(prec[i]);
        }
#pragma unroll
        for (int i = 0; i < P; i++)
            if (prec[tri(i, i)] < 1e-6)
                prec[tri(i, i)] = 1e-6;
        double logabs;
        int sign;
        bool ok = finite && mvn_invert<P>(prec, cov, logabs, sign);
#pragma unroll
        for (int i = 0; i < PT; i++)
            ok = ok && is_finite(cov[i]);
        if (!ok)
        {
            fallback = true;
            // a perfect fit (mse = 0) or T = P leaves no finite precision: that is not a failure
            // of the voxel, it just gets the uninformative precision of the catch branch
            if (finite)
                status = FVB_BAD_RESULT;
        }
    }
    if (fallback) // inference_nlls.cc:186-207: precisions 1e-12 I
    {
#pragma unroll
        for (int i = 0; i < P; i++)
#pragma unroll
            for (int j = 0; j <= i; j++)
                cov[tri(i, j)] = (i == j) ? 1e12 : 0.0;
    }
    double *dst = ka.out.mvn + v;
#pragma unroll
    for (int i = 0; i < PT; i++)
        dst[(size_t)i * V] = cov[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        dst[(size_t)(PT + i) * V] = par[i];
    dst[(size_t)(PT + P) * V] = 1.0;
    if (ka.out.status)
        ka.out.status[v] = status;
    if (ka.out.iterations)
        ka.out.iterations[v] = niter;
    if (ka.out.free_energy) // no free energy in NLLS: the final sum of squares is reported here
        ka.out.free_energy[v] = cf;
}

#endif // __HIPCC__

typedef void (*NllsKernelFn)(const NllsArgs);
struct NllsKernelInfo
{
    NllsKernelFn fn;
    const char *name;
};
NllsKernelInfo get_nlls_kernel(int model, int P);
} // namespace fvb

// ---------------------------------------------------------------------------------------------
// The same algorithm, one WAVEFRONT per voxel (any parameter count up to FVB_MAX_PARAMS, any
// built-in model; also what small volumes use): the linearisation, the moments J'J, J'r, r'r and
// the symmetric sweep inverse are the wave kernel's (vb_wave_kernel.h); every lane holds the same
// scalars, so the minimiser's control flow is wave-uniform.
//
// The loop is written as three pieces around its ONE model evaluation per iteration (the re-centre at
// the trial point) - propose a step, judge the trial, finish - so that a forward model that exists
// only as host code (FVB_MODEL_HOSTJAC, a FwdModel of a model library) runs it too: nlls_wave_step_kernel
// below does one piece sequence per launch, the host evaluates the model in between
// (fabber_nlls_run_hostmodel_host; inference_nlls.cc:94-214 works with any FwdModel).
// ---------------------------------------------------------------------------------------------
namespace fvb
{
// what the minimiser carries from one iteration to the next besides the accepted point in LDS
// (sv_pm = parameters, sv_Lam = J'J packed, sv_m = J'r)
struct NllsWaveState
{
    double cf, lambda;
    int32_t niter, status, running, solved;
};

// scalars of a voxel between the launches of the step kernel
struct NllsHmScalars
{
    NllsWaveState st;
    int32_t phase, pad;
};

struct NllsHmArgs
{
    NllsArgs na;
    WaveLayout L;
    double *persist;          // [V][persist_doubles]: LDS block [L.b, L.part) of the voxel
    NllsHmScalars *scalars;   // [V]
    const double *lin;        // [batch][T (P + 1)]: g then J of the batch's voxels about the trial point
    const int32_t *batch_ids; // [batch]
    double *means_out;        // [V][P] the next trial point
    int32_t *phase_out;       // [V] 0 = new, 1 = running, 3 = done (HmPhase of vb_hostmodel.h)
    int32_t persist_doubles;
};

#if defined(__HIPCC__)

// the accepted point <- the point the moments in LDS belong to
__device__ __forceinline__ void nlls_wave_accept(WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    double *sh = cx.sh;
    FVB_WAVE_FOR(i, L.P)
    {
        sh[L.sv_pm + i] = sh[L.m + i];
        sh[L.sv_m + i] = sh[L.u + i];
    }
    FVB_WAVE_FOR(e, L.PT)
    sh[L.sv_Lam + e] = sh[L.A + e];
    wave_sync();
}

// H = 2 J'J damped, step = -H^-1 grad, trial point into sh[L.m]
__device__ __forceinline__ void nlls_wave_propose(const fvb_nlls &nl, WaveCtx &cx, NllsWaveState &s)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, PP = L.PP;
    double *sh = cx.sh;
    s.niter++;
    FVB_WAVE_FOR(e, PP)
    {
        const int i = e / P, j = e % P;
        double h = 2.0 * sh[L.sv_Lam + tri(i, j)];
        if (i == j)
            h = nl.lm ? h * (1.0 + s.lambda) : h + s.lambda;
        sh[L.Lam + e] = h;
    }
    wave_sync();
    double la;
    int sg;
    bool solved = wave_sweep_inverse(cx, sh + L.Lam, sh + L.Sig, 0.0, la, sg);
    FVB_WAVE_FOR(i, P)
    {
        double step = 0;
        for (int j = 0; j < P; j++)
            step += sh[L.Sig + i * P + j] * (2.0 * sh[L.sv_m + j]);
        sh[L.rhs + i] = step;
    }
    wave_sync();
    for (int i = 0; i < P; i++)
        solved = solved && is_finite(sh[L.rhs + i]);
    FVB_WAVE_FOR(i, P)
    sh[L.m + i] = sh[L.sv_pm + i] + (solved ? sh[L.rhs + i] : 0.0);
    wave_sync();
    s.solved = solved ? 1 : 0;
}

// the trial point's re-centre is in LDS (its status: st): accept or reject it
__device__ __forceinline__ void nlls_wave_judge(const fvb_nlls &nl, WaveCtx &cx, NllsWaveState &s, int st)
{
    const double ncf = cx.sh[cx.L.s];
    if (s.solved && ncf < s.cf)
    {
        nlls_wave_accept(cx);
        s.lambda *= 0.1;
        const bool converged = 2.0 * fabs(s.cf - ncf) <= nl.cf_tolerance * (fabs(s.cf) + fabs(ncf) + DBL_EPSILON);
        s.cf = ncf;
        if (st != FVB_OK)
        {
            s.status = st;
            s.running = 0;
        }
        if (converged)
            s.running = 0;
    }
    else
    {
        s.lambda *= 10.0;
        if (s.lambda > nl.lambda_max)
            s.running = 0;
    }
}

// the NLLS precision (inference_nlls.cc:160-184) and the voxel's outputs
__device__ __forceinline__ void nlls_wave_finish(const KernelArgs &ka, WaveCtx &cx, NllsWaveState &s)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, PT = L.PT, PP = L.PP, v = cx.v;
    const size_t V = cx.V;
    double *sh = cx.sh;
    bool fallback = (s.status != FVB_OK);
    if (!fallback)
    {
        const double mse = s.cf / (double)(ka.n_unmasked - P);
        FVB_WAVE_FOR(e, PP)
        {
            const int i = e / P, j = e % P;
            double p = sh[L.sv_Lam + tri(i, j)] / mse;
            if (i == j && p < 1e-6)
                p = 1e-6;
            sh[L.Lam + e] = p;
        }
        wave_sync();
        bool finite = true;
        for (int e = 0; e < PT; e++)
            finite = finite && is_finite(sh[L.sv_Lam + e] / mse);
        double la;
        bool ok = finite && wave_mvn_invert(cx, sh + L.Lam, sh + L.Sig, la);
        if (ok)
            for (int e = 0; e < PP; e++)
                ok = ok && is_finite(sh[L.Sig + e]);
        if (!ok)
        {
            fallback = true;
            if (finite)
                s.status = FVB_BAD_RESULT;
        }
    }
    double *dst = ka.out.mvn + v;
    FVB_WAVE_FOR(e, PT)
    {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= e)
            i++;
        const int j = e - i * (i + 1) / 2;
        dst[(size_t)e * V] = fallback ? ((i == j) ? 1e12 : 0.0) : sh[L.Sig + i * P + j];
    }
    FVB_WAVE_FOR(i, P)
    dst[(size_t)(PT + i) * V] = sh[L.sv_pm + i];
    if (cx.lane == 0)
    {
        dst[(size_t)(PT + P) * V] = 1.0;
        if (ka.out.status)
            ka.out.status[v] = s.status;
        if (ka.out.iterations)
            ka.out.iterations[v] = s.niter;
        if (ka.out.free_energy)
            ka.out.free_energy[v] = s.cf;
    }
}

// series and masked timepoints of the voxel into LDS
__device__ __forceinline__ void nlls_wave_stage(const KernelArgs &ka, WaveCtx &cx)
{
    const WaveLayout &L = cx.L;
    FVB_WAVE_FOR(t, L.T)
    {
        cx.sh[L.y + t] = load_data(ka, (size_t)t * cx.V + cx.v);
        const int idx = ka.cfg.phi_index ? (int)ka.cfg.phi_index[t] : 0;
        cx.phi[t] = (idx == 255) ? -1 : 0; // masked timepoints drop out of every sum
    }
}

__global__ __launch_bounds__(64) void nlls_wave_kernel(const NllsArgs na, const WaveLayout L)
{
    extern __shared__ double wave_lds[];
    const KernelArgs &ka = na.ka;
    WaveCtx cx;
    cx.L = L;
    cx.sh = wave_lds;
    cx.phi = (int32_t *)(wave_lds + L.n_doubles);
    cx.lane = threadIdx.x;
    cx.v = blockIdx.x;
    cx.V = (size_t)ka.cfg.n_voxels;
    cx.lin = nullptr;
    cx.precValid = cx.covValid = false;
    cx.logdetLam = 0;
    cx.sv_prec = false;
    double *sh = cx.sh;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;

    nlls_wave_stage(ka, cx);
    FVB_WAVE_FOR(i, L.P)
    sh[L.m + i] = FVB_KPARAM(ka, post_mean, i); // starting estimate, Fabber space
    wave_sync();

    NllsWaveState s;
    s.status = wave_recentre(ka, ma, cx);
    s.cf = sh[L.s];
    nlls_wave_accept(cx);
    s.lambda = na.nl.lambda0;
    s.niter = 0;
    s.solved = 0;
    s.running = (s.status == FVB_OK);
    while (s.running && s.niter < na.nl.max_iterations)
    {
        nlls_wave_propose(na.nl, cx, s);
        const int st = wave_recentre(ka, ma, cx);
        nlls_wave_judge(na.nl, cx, s, st);
    }
    nlls_wave_finish(ka, cx, s);
}

// One launch = for every voxel of the batch: take the host's linearisation about the point asked for, judge it
// (or, the first time, make it the starting point), then either ask for the next trial point or finish.
__global__ __launch_bounds__(64) void nlls_wave_step_kernel(const NllsHmArgs ha)
{
    extern __shared__ double wave_lds[];
    const KernelArgs &ka = ha.na.ka;
    const WaveLayout &L = ha.L;
    const int slot = blockIdx.x;
    const int v = ha.batch_ids[slot];
    WaveCtx cx;
    cx.L = L;
    cx.sh = wave_lds;
    cx.phi = (int32_t *)(wave_lds + L.n_doubles);
    cx.lane = threadIdx.x;
    cx.v = v;
    cx.V = (size_t)ka.cfg.n_voxels;
    cx.lin = ha.lin + (size_t)slot * L.T * (L.P + 1);
    cx.precValid = cx.covValid = false;
    cx.logdetLam = 0;
    cx.sv_prec = false;
    double *sh = cx.sh;
    ModelArgs ma;
    ma.iopt0 = 0;
    ma.dopt0 = 0;
    ma.design = nullptr;

    nlls_wave_stage(ka, cx);
    NllsHmScalars sc = ha.scalars[v];
    double *persist = ha.persist + (size_t)v * ha.persist_doubles;
    const bool first = sc.phase == 0;
    if (first)
    {
        FVB_WAVE_FOR(e, L.part - L.b)
        sh[L.b + e] = 0;
        wave_sync();
        FVB_WAVE_FOR(i, L.P)
        sh[L.m + i] = FVB_KPARAM(ka, post_mean, i);
    }
    else
    {
        FVB_WAVE_FOR(e, L.part - L.b)
        sh[L.b + e] = persist[e];
    }
    wave_sync();
    NllsWaveState &s = sc.st;
    const int st = wave_recentre(ka, ma, cx); // the host's g and J; moments
    if (first)
    {
        s.status = st;
        s.cf = sh[L.s];
        nlls_wave_accept(cx);
        s.lambda = ha.na.nl.lambda0;
        s.niter = 0;
        s.solved = 0;
        s.running = (s.status == FVB_OK);
    }
    else
        nlls_wave_judge(ha.na.nl, cx, s, st);
    if (s.running && s.niter < ha.na.nl.max_iterations)
    {
        nlls_wave_propose(ha.na.nl, cx, s);
        sc.phase = 1;
        FVB_WAVE_FOR(e, L.part - L.b)
        persist[e] = sh[L.b + e];
        FVB_WAVE_FOR(i, L.P)
        ha.means_out[(size_t)v * L.P + i] = sh[L.m + i];
    }
    else
    {
        nlls_wave_finish(ka, cx, s);
        sc.phase = 3;
    }
    if (cx.lane == 0)
    {
        ha.scalars[v] = sc;
        ha.phase_out[v] = sc.phase;
    }
}

#endif // __HIPCC__
} // namespace fvb
