/*
 * vb_hostmodel.h - voxelwise VB for forward models that can only be evaluated on the host.
 *
 * A model library written for the reference (a FwdModel subclass behind fabber_load_models) has
 * no device body. Its Evaluate() stays where it is - on the host, 2P + 1 calls per voxel and
 * iteration, exactly the calls LinearizedFwdModel::ReCentre makes (fwdmodel_linear.cc:126-182) -
 * and everything else of the loop runs here: the host hands over, per voxel, the linearisation
 * about the current means (g [T], J [T][P]); one launch of vb_wave_step_kernel carries every
 * voxel from that re-centre to the next one (one wavefront per voxel, J in LDS, the device
 * functions of vb_wave_kernel.h); the new means go back to the host. The per-voxel state
 * (posterior, prior, noise, convergence detector, saved copy for revert) persists in HBM between
 * launches.
 *
 * The loop of Vb::DoCalculationsVoxelwise (inference_vb.cc:423-571) cut at its re-centres:
 *
 *   NEW     [setup re-centre :235]  reset detector, save          -> body
 *   LOOP    [re-centre :490]        F "lin", history, ++it, test  -> finished | body
 *   body    save?, priors, F "before", UpdateTheta, F "theta", UpdateNoise, F "phi"   (ask for a re-centre)
 *   finished save?, revert? -> REVERT (ask for a re-centre about the restored means) | DONE
 *   REVERT  [re-centre :521]        F                             -> DONE
 */
#pragma once

#include "vb_wave_kernel.h"

namespace fvb
{
enum HmPhase
{
    HM_NEW = 0,
    HM_LOOP = 1,
    HM_REVERT = 2,
    HM_DONE = 3
};

// scalars of a voxel that live in registers during a step and in HBM between steps
struct HmScalars
{
    ConvState conv;
    double F, Fprior, logdetLam;
    int32_t it, hist_len, status, phase;
    int32_t precValid, covValid, sv_prec, setup_failed;
};

struct HmArgs
{
    KernelArgs ka;
    WaveLayout L;
    double *persist;          // [V][persist_doubles]: LDS block [L.b, L.part) of the voxel
    HmScalars *scalars;       // [V]
    const double *lin;        // [batch][T (P + 1)]: g then J of the batch's voxels
    const int32_t *batch_ids; // [batch] the voxels this launch works on (one workgroup each)
    void *ar_scalars;         // AR(1) noise: [V] HmArScalars<NPHI, NA> instead (vb_hostmodel_ar.h)
    double *means_out;        // [V][P] means the NEXT linearisation is wanted about
    int32_t *phase_out;       // [V]
    int32_t persist_doubles;
};

#if defined(__HIPCC__)

#define FVB_WAVE_FOR(idx, n) for (int idx = cx.lane; idx < (n); idx += 64)

// result MVN and the per-voxel outputs, as at the end of vb_wave_kernel
__device__ __forceinline__ void hm_write_outputs(const KernelArgs &ka, WaveCtx &cx, HmScalars &sc)
{
    const WaveLayout &L = cx.L;
    const int P = L.P, N = L.N, PP = L.PP, v = cx.v;
    const size_t V = cx.V;
    double *sh = cx.sh;
    const int n = P + N, nCov = n * (n + 1) / 2;
    if (!wave_ensure_cov(cx))
    {
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = 0;
        if (sc.status == FVB_OK)
            sc.status = FVB_BAD_RESULT;
        wave_sync();
    }
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
            val = b * b * c;
        }
        dst[(size_t)e * V] = val;
    }
    FVB_WAVE_FOR(i, n)
    dst[(size_t)(nCov + i) * V] = (i < P) ? sh[L.m + i] : sh[L.b + (i - P)] * sh[L.c + (i - P)];
    if (cx.lane == 0)
    {
        dst[(size_t)(nCov + n) * V] = 1.0;
        if (ka.out.f_history && sc.hist_len < ka.cfg.f_history_rows) // :553-554
            ka.out.f_history[(size_t)sc.hist_len * V + v] = sc.F;
        sc.hist_len++;
        if (ka.out.f_history_len)
            ka.out.f_history_len[v] = sc.hist_len;
        if (ka.out.free_energy)
            ka.out.free_energy[v] = sc.F;
        if (ka.out.status)
            ka.out.status[v] = sc.status | (sc.setup_failed ? 0x100 : 0);
        if (ka.out.iterations)
            ka.out.iterations[v] = sc.it;
    }
}

template <bool NEEDF>
__global__ __launch_bounds__(64) void vb_wave_step_kernel(const HmArgs ha)
{
    extern __shared__ double wave_lds[];
    const KernelArgs &ka = ha.ka;
    const WaveLayout &L = ha.L;
    const int slot = blockIdx.x; // this launch's batch: the voxel and its place in the batch's linearisations
    const int v = ha.batch_ids[slot];
    WaveCtx cx;
    cx.L = L;
    cx.sh = wave_lds;
    cx.phi = (int32_t *)(wave_lds + L.n_doubles);
    cx.lane = threadIdx.x;
    cx.v = v;
    cx.V = (size_t)ka.cfg.n_voxels;
    const int T = L.T, P = L.P, N = L.N, PP = L.PP;
    const size_t V = cx.V;
    cx.lin = ha.lin + (size_t)slot * T * (P + 1);
    double *sh = cx.sh;
    ModelArgs ma;
    ma.iopt0 = 0;
    ma.dopt0 = 0;
    ma.design = nullptr;

    // ---- the voxel's series and noise pattern (re-staged every step; T floats) ----
    FVB_WAVE_FOR(t, T)
    {
        sh[L.y + t] = load_data(ka, (size_t)t * V + v);
        const int idx = ka.cfg.phi_index ? (int)ka.cfg.phi_index[t] : 0;
        cx.phi[t] = (idx == 255) ? -1 : idx;
    }
    wave_sync();
    FVB_WAVE_FOR(phi, N)
    {
        int n = 0;
        for (int t = 0; t < T; t++)
            n += (cx.phi[t] == phi);
        sh[L.cnt + phi] = (double)n;
    }

    // ---- state: from the initial MVN on the first step, from HBM afterwards ----
    HmScalars sc = ha.scalars[v];
    double *persist = ha.persist + (size_t)v * ha.persist_doubles;
    if (sc.phase == HM_NEW)
    {
        const int n = P + N, nCov = n * (n + 1) / 2;
        const double *src = ka.cfg.init_mvn + v;
        FVB_WAVE_FOR(e, L.part - L.b)
        sh[L.b + e] = 0;
        wave_sync();
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = src[(size_t)tri(e / P, e % P) * V];
        FVB_WAVE_FOR(i, P)
        {
            sh[L.m + i] = src[(size_t)(nCov + i) * V];
            sh[L.pm + i] = 0; // fwd_prior = MVNDist(P): zero mean, identity (inference_vb.cc:159)
            sh[L.pprec + i] = 1;
        }
        FVB_WAVE_FOR(phi, N)
        {
            const double nm = src[(size_t)(nCov + P + phi) * V];
            const double nv = src[(size_t)tri(P + phi, P + phi) * V];
            const double b = nv / nm; // GammaDist::SetMeanVariance, dist_gamma.cc:29-33
            sh[L.b + phi] = b;
            sh[L.c + phi] = nm / b;
        }
        cx.covValid = true;
        cx.precValid = false;
        cx.logdetLam = 0;
        cx.sv_prec = false;
        sc.F = 1234.5678; // inference_vb.cc:438
        sc.Fprior = 0;
        sc.it = 0;
        sc.hist_len = 0;
        sc.status = FVB_OK;
        sc.setup_failed = 0;
    }
    else
    {
        FVB_WAVE_FOR(e, L.part - L.b)
        sh[L.b + e] = persist[e];
        cx.covValid = sc.covValid != 0;
        cx.precValid = sc.precValid != 0;
        cx.sv_prec = sc.sv_prec != 0;
        cx.logdetLam = sc.logdetLam;
    }
    wave_sync();

    const bool use_save = ka.cfg.convergence == FVB_CONV_FREDUCE || ka.cfg.convergence == FVB_CONV_TRIALMODE
        || ka.cfg.convergence == FVB_CONV_LM;
    double F = sc.F, Fprior = sc.Fprior;
    int status = sc.status;
    bool done = false;

#define FVB_HM_EVAL_F()                                                                                      \
    {                                                                                                        \
        double Fn_;                                                                                          \
        bool fin_ = true;                                                                                    \
        if (!wave_free_energy(ka, cx, Fprior, Fn_, fin_))                                                    \
        {                                                                                                    \
            status = FVB_BAD_RESULT;                                                                         \
            done = true;                                                                                     \
            break;                                                                                           \
        }                                                                                                    \
        if (!fin_)                                                                                           \
        {                                                                                                    \
            status = FVB_BAD_FREE_ENERGY;                                                                    \
            done = true;                                                                                     \
            break;                                                                                           \
        }                                                                                                    \
        F = Fn_;                                                                                             \
    }

    do // (one pass; `break` leaves with `done` saying whether the voxel is finished)
    {
        const int lin_status = wave_recentre(ka, ma, cx); // loads the host's g, J; moments
        if (sc.phase == HM_REVERT) // :516-525, after the re-centre about the restored means
        {
            status = lin_status;
            done = true;
            if (status == FVB_OK && NEEDF)
            {
                if (!wave_ensure_cov(cx))
                {
                    status = FVB_BAD_RESULT;
                    break;
                }
                wave_residuals(cx, true);
                FVB_HM_EVAL_F()
            }
            break;
        }
        if (sc.phase == HM_NEW)
        {
            if (lin_status != FVB_OK) // the first re-centre is outside the reference's try block
            {
                status = lin_status;
                sc.setup_failed = 1;
                done = true;
                break;
            }
            conv_init(sc.conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
            conv_reset(sc.conv);
            if (use_save)
                wave_save_state(cx); // :432-434
        }
        else
        {
            status = lin_status; // :490
            if (status != FVB_OK)
            {
                done = true;
                break;
            }
            if (NEEDF) // "lin" :495
            {
                wave_residuals(cx, true);
                FVB_HM_EVAL_F()
            }
            if (cx.lane == 0 && ka.out.f_history && sc.hist_len < ka.cfg.f_history_rows) // :496-497
                ka.out.f_history[(size_t)sc.hist_len * V + v] = F;
            sc.hist_len++;
            ++sc.it;
            if (conv_test(sc.conv, F))
            {
                if (use_save && conv_need_save(sc.conv)) // :506-513
                    wave_save_state(cx);
                if (use_save && conv_need_revert(sc.conv)) // :516-525: needs a re-centre first
                {
                    wave_restore_state(cx);
                    sc.phase = HM_REVERT;
                    break;
                }
                done = true;
                break;
            }
        }
        // ---- the loop body up to its re-centre (:451-490) ----
        if (use_save && conv_need_save(sc.conv)) // :451-458
            wave_save_state(cx);
        if (!wave_apply_priors<NEEDF>(ka, cx, sc.it, Fprior))
        {
            status = FVB_BAD_RESULT;
            done = true;
            break;
        }
        if (NEEDF) // "before" :468
        {
            if (!wave_ensure_cov(cx))
            {
                status = FVB_BAD_RESULT;
                done = true;
                break;
            }
            wave_residuals(cx, true);
            FVB_HM_EVAL_F()
        }
        if (!wave_update_theta(cx, conv_lm_alpha(sc.conv)) || !wave_ensure_cov(cx)) // :470
        {
            status = FVB_BAD_RESULT;
            done = true;
            break;
        }
        wave_residuals(cx, false);
        if (NEEDF) // "theta" :477
            FVB_HM_EVAL_F()
        wave_update_noise(ka, cx); // :479
        if (NEEDF) // "phi" :485
            FVB_HM_EVAL_F()
        sc.phase = HM_LOOP;
    } while (false);
#undef FVB_HM_EVAL_F

    sc.F = F;
    sc.Fprior = Fprior;
    sc.status = status;
    if (done)
    {
        sc.phase = HM_DONE;
        hm_write_outputs(ka, cx, sc);
    }
    else
    {
        FVB_WAVE_FOR(e, L.part - L.b)
        persist[e] = sh[L.b + e];
        FVB_WAVE_FOR(i, P)
        ha.means_out[(size_t)v * P + i] = sh[L.m + i];
    }
    if (cx.lane == 0)
    {
        sc.covValid = cx.covValid;
        sc.precValid = cx.precValid;
        sc.sv_prec = cx.sv_prec;
        sc.logdetLam = cx.logdetLam;
        ha.scalars[v] = sc;
        ha.phase_out[v] = sc.phase;
    }
}

#undef FVB_WAVE_FOR

#endif // __HIPCC__

} // namespace fvb
