/*
 * vb_hostmodel_ar.h - voxelwise VB with AR(1) noise (Ar1cNoiseModel, noisemodel_ar.cc) for forward models that
 * can only be evaluated on the host: vb_hostmodel.h's cut of the loop at its re-centres, with the noise model's
 * steps of vb_wave_ar_kernel.h between them. One launch of vb_wave_ar_step_kernel carries every voxel from one
 * re-centre to the next; the alpha posterior (a few doubles, registers during a step) persists next to the
 * detector state, the LDS block [L.b, L.part) as for white noise.
 */
#pragma once

#include "vb_wave_ar_kernel.h"
#include "vb_hostmodel.h"

namespace fvb
{
template <int NPHI, int NA>
struct HmArScalars
{
    HmScalars base;
    ArState<NPHI, NA> st, st_saved;
};

#if defined(__HIPCC__)

#define FVB_WAVE_FOR(idx, n) for (int idx = cx.lane; idx < (n); idx += 64)

template <int NPHI, int NA, bool NEEDF>
__global__ __launch_bounds__(64) void vb_wave_ar_step_kernel(const HmArgs ha)
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
    const int T = L.T, P = L.P, PP = L.PP;
    const size_t V = cx.V;
    cx.lin = ha.lin + (size_t)slot * T * (P + 1);
    double *sh = cx.sh;
    ModelArgs ma;
    ma.iopt0 = 0;
    ma.dopt0 = 0;
    ma.design = nullptr;

    FVB_WAVE_FOR(t, T)
    {
        sh[L.y + t] = load_data(ka, (size_t)t * V + v);
        cx.phi[t] = 0;
    }
    wave_sync();

    typedef HmArScalars<NPHI, NA> Scalars;
    Scalars *all = (Scalars *)ha.ar_scalars;
    HmScalars sc = all[v].base;
    ArState<NPHI, NA> st = all[v].st, st_saved = all[v].st_saved;
    double *persist = ha.persist + (size_t)v * ha.persist_doubles;
    if (sc.phase == HM_NEW)
    {
        // MVNDist::Load + Ar1cParams::InputFromMVN (dist_mvn.cc:347-374; noisemodel_ar.cc:302-316)
        constexpr int NN = NA + NPHI;
        const int n = P + NN, nCov = n * (n + 1) / 2;
        const double *src = ka.cfg.init_mvn + v;
        FVB_WAVE_FOR(e, L.part - L.b)
        sh[L.b + e] = 0;
        wave_sync();
        FVB_WAVE_FOR(e, PP)
        sh[L.Sig + e] = src[(size_t)tri(e / P, e % P) * V];
        FVB_WAVE_FOR(i, P)
        {
            sh[L.m + i] = src[(size_t)(nCov + i) * V];
            sh[L.pm + i] = 0;
            sh[L.pprec + i] = 1;
        }
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
            st.am[i] = src[(size_t)(nCov + P + i) * V];
#pragma unroll
            for (int j = 0; j < NA; j++)
                st.acov[i][j] = src[(size_t)tri(P + (i > j ? i : j), P + (i > j ? j : i)) * V];
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
        st_saved = st;
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
        if (!ar_free_energy<NPHI, NA>(ka, cx, st, Fprior, Fn_, fin_))                                        \
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
        const int lin_status = wave_recentre(ka, ma, cx, false); // loads the host's g, J
        if (sc.phase == HM_REVERT) // inference_vb.cc:516-525, after the re-centre about the restored means
        {
            status = lin_status;
            done = true;
            if (status == FVB_OK && NEEDF)
                FVB_HM_EVAL_F()
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
            // Precalculate (noisemodel_ar.cc:749-769)
            ar_update_marginals<NPHI, NA>(st);
            FVB_WAVE_FOR(i, NPHI)
            sh[L.c + i] = ka.cfg.noise_prior_c[i] + (T / NPHI - 1) * 0.5;
            wave_sync();
            conv_init(sc.conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
            conv_reset(sc.conv);
            if (use_save)
            {
                wave_save_state(cx);
                st_saved = st;
            }
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
                FVB_HM_EVAL_F()
            if (cx.lane == 0 && ka.out.f_history && sc.hist_len < ka.cfg.f_history_rows) // :496-497
                ka.out.f_history[(size_t)sc.hist_len * V + v] = F;
            sc.hist_len++;
            ++sc.it;
            if (conv_test(sc.conv, F))
            {
                if (use_save && conv_need_save(sc.conv)) // :506-513
                {
                    wave_save_state(cx);
                    st_saved = st;
                }
                if (use_save && conv_need_revert(sc.conv)) // :516-525: needs a re-centre first
                {
                    wave_restore_state(cx);
                    st = st_saved;
                    sc.phase = HM_REVERT;
                    break;
                }
                done = true;
                break;
            }
        }
        // ---- the loop body up to its re-centre, as vb_wave_ar_kernel runs it ----
        if (use_save && conv_need_save(sc.conv))
        {
            wave_save_state(cx);
            st_saved = st;
        }
        if (!wave_apply_priors<NEEDF>(ka, cx, sc.it, Fprior))
        {
            status = FVB_BAD_RESULT;
            done = true;
            break;
        }
        if (NEEDF)
            FVB_HM_EVAL_F()
        if (!ar_update_theta<NPHI, NA>(cx, st)) // the AR model ignores the LM damping (noisemodel_ar.cc:558-634)
        {
            status = FVB_BAD_RESULT;
            done = true;
            break;
        }
        if (NEEDF)
            FVB_HM_EVAL_F()
        // UpdateNoise = UpdateAlpha, then UpdatePhi (:405-410)
        if (!wave_ensure_prec(cx) || !wave_ensure_cov(cx))
        {
            status = FVB_BAD_RESULT;
            done = true;
            break;
        }
        ar_residual(cx);
        ar_j_sigma(cx);
        status = ar_update_alpha<NPHI, NA>(ka, cx, st);
        if (status != FVB_OK)
        {
            done = true;
            break;
        }
        ar_update_phi<NPHI, NA>(ka, cx, st);
        if (NEEDF)
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
        ar_write_outputs<NPHI, NA>(ka, cx, st, status, sc.setup_failed != 0, F, sc.it, sc.hist_len);
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
        all[v].base = sc;
        all[v].st = st;
        all[v].st_saved = st_saved;
        ha.phase_out[v] = sc.phase;
    }
}

#undef FVB_WAVE_FOR

#endif // __HIPCC__

} // namespace fvb
