/*
 * vb_spatial_noise.h - spatial VB (vb_spatial.h) under the noise models beyond "white noise, one precision".
 *
 * Vb::DoCalculationsSpatial calls the noise model through its virtuals (inference_vb.cc:645 UpdateTheta, :688
 * UpdateNoise, :643-700 CalcFreeEnergy): the loop is the same for WhiteNoiseModel with a noise-pattern
 * (noisemodel_white.cc:166-273, several precisions) and for Ar1cNoiseModel (noisemodel_ar.cc); the reference's own
 * test-suite runs the AR fit under method=spatialvb (test/test_vb.cc:617-694, instantiated at :847).
 *
 * What the FIRST sweep needs of any of these models is J'XJ and J'X(y - g) (eq 19, 20) - fixed while the sweep runs,
 * because the noise posterior and the linearisation only change in the second sweep. The second-sweep kernel of this
 * file therefore leaves them in the state image's A / U / S rows as EFFECTIVE moments (and E[phi], or 1, in rows
 * B x C), and the per-level kernel, the prep kernel and both forms of the ordered sweep (vb_spatial.h) run unchanged:
 *     several precisions   A_eff = sum_k E[phi_k] A_k,  u_eff = sum_k E[phi_k] u_k,  B = C = 1
 *     AR(1), one echo      A_eff = J'QJ,  u_eff = J'Q(y - g),  s_eff = (y - g)'Q(y - g),  B, C = phi's posterior,
 *                          Q = M00 + E[a] M10 + E[a^2] M20 from the lag-1 moments (vb_lane_ar_kernel.h)
 * The model's own sufficient statistics (per-class moments; diagonal and lag-1 moments with the first / last rows,
 * the alpha posterior) live in extra rows after SpLayout<P>::ROWS and are read by the second sweep, by the set-up
 * and result kernels, and by the first sweep's free-energy checks (policy::sweep_F, only when F is evaluated).
 *
 * A policy NZ provides: EXTRA_ROWS, WAVES (launch bound), LDS_CLASSES (the kernel keeps cfg.phi_index in LDS),
 * Full (moments + noise posterior), load_full / store_full, init_noise, recentre, update_noise, F_full, effective,
 * sweep_F, n_noise_out, pack_noise.
 */
#pragma once

#include "vb_lane_ar_kernel.h"
#include "vb_lane_arn_kernel.h"
#include "vb_lane_pattern_kernel.h"
#include "vb_spatial.h"

namespace fvb
{
#if defined(__HIPCC__)

// ---- white noise, N precisions (noise-pattern) ---------------------------------------------------------
template <int P, int N>
struct SpPattern
{
    typedef SpLayout<P> L;
    static constexpr int PT = L::PT;
    static constexpr int NB = L::ROWS, NC = NB + N, AK = NC + N, UK = AK + N * PT, SK = UK + N * P;
    static constexpr int EXTRA_ROWS = 3 * N + N * PT + N * P;
    static constexpr int WAVES = lane_pattern_waves<P, N>();
    static constexpr bool LDS_CLASSES = true;
    struct Full
    {
        PatternMoments<P, N> mo;
        PatternNoise<N> nz;
        double kk[N], tr[N]; // of the last update_noise (about the centre the moments belong to)
    };
    static __device__ __forceinline__ int n_noise_out(const KernelArgs &ka)
    {
        return ka.cfg.n_phis;
    }
    static __device__ __forceinline__ void load_full(const SpatialArgs &sa, int v, Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        const double *p = sa.state + v;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            f.nz.b[k] = p[(size_t)(NB + k) * V];
            f.nz.c[k] = p[(size_t)(NC + k) * V];
            f.nz.count[k] = sa.nz_count[k];
#pragma unroll
            for (int i = 0; i < PT; i++)
                f.mo.A[k][i] = p[(size_t)(AK + k * PT + i) * V];
#pragma unroll
            for (int i = 0; i < P; i++)
                f.mo.u[k][i] = p[(size_t)(UK + k * P + i) * V];
            f.mo.s[k] = p[(size_t)(SK + k) * V];
            f.kk[k] = f.tr[k] = 0;
        }
#pragma unroll
        for (int i = 0; i < P; i++)
            f.mo.ml[i] = p[(size_t)(L::ML + i) * V];
        f.mo.precise = sa.it < sa.ka.precise_passes; // (as sp_load)
    }
    static __device__ __forceinline__ void store_full(const SpatialArgs &sa, int v, const Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        double *p = sa.state + v;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            p[(size_t)(NB + k) * V] = f.nz.b[k];
            p[(size_t)(NC + k) * V] = f.nz.c[k];
#pragma unroll
            for (int i = 0; i < PT; i++)
                p[(size_t)(AK + k * PT + i) * V] = f.mo.A[k][i];
#pragma unroll
            for (int i = 0; i < P; i++)
                p[(size_t)(UK + k * P + i) * V] = f.mo.u[k][i];
            p[(size_t)(SK + k) * V] = f.mo.s[k];
        }
    }
    // initial noise posterior: hardcoded (noisemodel_white.cc:127-164, resolved by the host into cfg) or from the MVN
    // (WhiteParams::InputFromMVN, noisemodel_white.cc:70-79); src = this voxel's column of init_mvn or NULL
    static __device__ __forceinline__ void init_noise(const SpatialArgs &sa, const double *src, VoxelState<P> &st, Full &f)
    {
        const KernelArgs &ka = sa.ka;
        const size_t V = (size_t)ka.cfg.n_voxels;
        const int n_phis = ka.cfg.n_phis, n = P + n_phis, nCov = n * (n + 1) / 2;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            f.nz.b[k] = f.nz.c[k] = 1;
            f.nz.count[k] = sa.nz_count[k];
            f.kk[k] = f.tr[k] = 0;
            if (k < n_phis)
            {
                if (src)
                {
                    const double nm = src[(size_t)(nCov + P + k) * V];
                    const double nv = src[(size_t)tri(P + k, P + k) * V];
                    f.nz.b[k] = nv / nm; // GammaDist::SetMeanVariance, dist_gamma.cc:29-33
                    f.nz.c[k] = nm / f.nz.b[k];
                }
                else
                {
                    f.nz.b[k] = ka.cfg.noise_post_b[k];
                    f.nz.c[k] = ka.cfg.noise_post_c[k];
                }
            }
        }
    }
    template <class Model>
    static __device__ __forceinline__ int recentre(const SpatialArgs &sa, const ModelArgs &ma, int v, const double (&centre)[P],
        Full &f, bool precise, const uint8_t *classes)
    {
        return recentre_pattern<Model, P, N>(sa.ka, ma, v, centre, f.mo, precise, classes);
    }
    // WhiteNoiseModel::UpdateNoise (noisemodel_white.cc:228-273)
    template <class Model>
    static __device__ __forceinline__ int update_noise(const SpatialArgs &sa, const ModelArgs &ma, int v, VoxelState<P> &st, Full &f,
        const uint8_t *classes)
    {
        const KernelArgs &ka = sa.ka;
        const int n_phis = ka.cfg.n_phis;
        const bool lost = residual_terms_pattern<P, N>(st, f.mo, n_phis, ka.residual_tol, f.kk, f.tr);
        const int mode = ka.residual_mode; // 0 adaptive, 1 always exact, 2 moments only
        const bool want = (mode == 1) || (mode == 0 && lost);
        if (__any(want)) // wave-uniform: the pass is taken by the whole wavefront or not at all
        {
            double exact[N];
            exact_residual_pattern<Model, P, N>(ka, ma, v, f.mo, st.m, exact, classes);
            if (want)
            {
#pragma unroll
                for (int k = 0; k < N; k++)
                    f.kk[k] = exact[k];
            }
        }
        update_noise_pattern<N>(ka, f.nz, n_phis, f.kk, f.tr);
        return FVB_OK;
    }
    // CalcFreeEnergy with the complete statistics in registers. at_centre: the moments were taken about the current
    // means (k = y - g); else the forms of the last update_noise are used (same means, same centre).
    static __device__ __forceinline__ bool F_full(const SpatialArgs &sa, VoxelState<P> &st, Full &f, bool at_centre, double Fprior,
        double &F, bool &finite)
    {
        const KernelArgs &ka = sa.ka;
        const int n_phis = ka.cfg.n_phis;
        PatternFCache<N> fc;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            fc.c_fn[k] = __builtin_nan("");
            fc.lgamma_c[k] = fc.digamma_c[k] = 0;
        }
        bool logdet_valid = st.precValid;
        if (at_centre)
        {
            double tr_now[N];
            trace_pattern<P, N>(st, f.mo, n_phis, tr_now);
            return calc_free_energy_pattern<P, N>(ka, st, f.nz, n_phis, f.mo.s, tr_now, Fprior, fc, logdet_valid, F, finite);
        }
        return calc_free_energy_pattern<P, N>(ka, st, f.nz, n_phis, f.kk, f.tr, Fprior, fc, logdet_valid, F, finite);
    }
    // J'XJ, J'X(y - g) in update_theta_pattern's order of operations; E[phi] is inside them
    static __device__ __forceinline__ void effective(const SpatialArgs &sa, const Full &f, VoxelState<P> &st, Moments<P> &w)
    {
        const int n_phis = sa.ka.cfg.n_phis;
#pragma unroll
        for (int i = 0; i < PT; i++)
            w.A[i] = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            w.u[i] = 0;
            w.ml[i] = f.mo.ml[i];
        }
        w.s = 0;
        w.precise = f.mo.precise;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            if (k < n_phis)
            {
                const double phibar = f.nz.b[k] * f.nz.c[k]; // GammaDist::CalcMean
#pragma unroll
                for (int i = 0; i < PT; i++)
                    w.A[i] += phibar * f.mo.A[k][i];
#pragma unroll
                for (int i = 0; i < P; i++)
                    w.u[i] += phibar * f.mo.u[k][i];
                w.s += phibar * f.mo.s[k];
            }
        }
        st.b = 1;
        st.c = 1;
    }
    // the first sweep's free energy (see SpWhite::sweep_F): the statistics come from the state image
    static __device__ __forceinline__ bool sweep_F(const SpatialArgs &sa, int v, VoxelState<P> &st, const Moments<P> &, bool at_centre,
        double Fprior, double &F, bool &finite)
    {
        Full f;
        load_full(sa, v, f);
        if (!at_centre)
            (void)residual_terms_pattern<P, N>(st, f.mo, sa.ka.cfg.n_phis, 0.0, f.kk, f.tr);
        return F_full(sa, st, f, at_centre, Fprior, F, finite);
    }
    // noise block of the result MVN (noisemodel_white.cc:55-68); dst = this voxel's column, n = P + n_phis
    static __device__ __forceinline__ void pack_noise(const SpatialArgs &sa, int v, double *dst)
    {
        const KernelArgs &ka = sa.ka;
        const size_t V = (size_t)ka.cfg.n_voxels;
        const int n_phis = ka.cfg.n_phis, n = P + n_phis, nCov = n * (n + 1) / 2;
        const double *p = sa.state + v;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            if (k < n_phis)
            {
                const double b = p[(size_t)(NB + k) * V], c = p[(size_t)(NC + k) * V];
                for (int j = 0; j < P + k; j++)
                    dst[(size_t)tri(P + k, j) * V] = 0.0;
                dst[(size_t)tri(P + k, P + k) * V] = b * b * c;
                dst[(size_t)(nCov + P + k) * V] = b * c;
            }
        }
    }
};

// ---- AR(1) noise, one echo, no cross terms (2 alphas of which alpha_1 is updated, 1 phi) -------------------
template <int P>
struct SpAr1
{
    typedef SpLayout<P> L;
    static constexpr int PT = L::PT;
    static constexpr int AL = L::ROWS, AD = AL + 7, CC = AD + PT, UD = CC + PT, UC = UD + P, SD = UC + P, SC = SD + 1, J1 = SC + 1,
                         JT = J1 + P, R1 = JT + P, RT = R1 + 1;
    static constexpr int EXTRA_ROWS = 7 + 2 * PT + 4 * P + 4;
    static constexpr int WAVES = lane_waves<P>();
    static constexpr bool LDS_CLASSES = false;
    struct Full
    {
        ArMoments<P> mo;
        ArAlpha al;
        ArForms f; // of the last update_noise
    };
    static __device__ __forceinline__ int n_noise_out(const KernelArgs &)
    {
        return 3;
    }
    static __device__ __forceinline__ void load_alpha(const SpatialArgs &sa, int v, ArAlpha &al)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        const double *p = sa.state + (size_t)AL * V + v;
        al.mean[0] = p[0];
        al.mean[1] = p[V];
        al.c11 = p[2 * V];
        al.c12 = p[3 * V];
        al.c22 = p[4 * V];
        al.a1 = p[5 * V];
        al.a2 = p[6 * V];
    }
    static __device__ __forceinline__ void load_full(const SpatialArgs &sa, int v, Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        const double *p = sa.state + v;
        load_alpha(sa, v, f.al);
#pragma unroll
        for (int i = 0; i < PT; i++)
        {
            f.mo.Ad[i] = p[(size_t)(AD + i) * V];
            f.mo.C[i] = p[(size_t)(CC + i) * V];
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            f.mo.ud[i] = p[(size_t)(UD + i) * V];
            f.mo.uc[i] = p[(size_t)(UC + i) * V];
            f.mo.J1[i] = p[(size_t)(J1 + i) * V];
            f.mo.JT[i] = p[(size_t)(JT + i) * V];
            f.mo.ml[i] = p[(size_t)(L::ML + i) * V];
        }
        f.mo.sd = p[(size_t)SD * V];
        f.mo.sc = p[(size_t)SC * V];
        f.mo.r1 = p[(size_t)R1 * V];
        f.mo.rT = p[(size_t)RT * V];
#pragma unroll
        for (int w = 0; w < 3; w++)
            f.f.kk[w] = f.f.tr[w] = 0;
    }
    static __device__ __forceinline__ void store_full(const SpatialArgs &sa, int v, const Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        double *p = sa.state + v;
        double *a = p + (size_t)AL * V;
        a[0] = f.al.mean[0];
        a[V] = f.al.mean[1];
        a[2 * V] = f.al.c11;
        a[3 * V] = f.al.c12;
        a[4 * V] = f.al.c22;
        a[5 * V] = f.al.a1;
        a[6 * V] = f.al.a2;
#pragma unroll
        for (int i = 0; i < PT; i++)
        {
            p[(size_t)(AD + i) * V] = f.mo.Ad[i];
            p[(size_t)(CC + i) * V] = f.mo.C[i];
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            p[(size_t)(UD + i) * V] = f.mo.ud[i];
            p[(size_t)(UC + i) * V] = f.mo.uc[i];
            p[(size_t)(J1 + i) * V] = f.mo.J1[i];
            p[(size_t)(JT + i) * V] = f.mo.JT[i];
        }
        p[(size_t)SD * V] = f.mo.sd;
        p[(size_t)SC * V] = f.mo.sc;
        p[(size_t)R1 * V] = f.mo.r1;
        p[(size_t)RT * V] = f.mo.rT;
    }
    // Ar1cNoiseModel::HardcodedInitialDists (noisemodel_ar.cc:379-403) or Ar1cParams::InputFromMVN (:302-316), then
    // Ar1cNoiseModel::Precalculate (:749-769)
    static __device__ __forceinline__ void init_noise(const SpatialArgs &sa, const double *src, VoxelState<P> &st, Full &f)
    {
        const KernelArgs &ka = sa.ka;
        const size_t V = (size_t)ka.cfg.n_voxels;
        constexpr int n = P + 3, nCov = n * (n + 1) / 2;
        if (src)
        {
            f.al.mean[0] = src[(size_t)(nCov + P) * V];
            f.al.mean[1] = src[(size_t)(nCov + P + 1) * V];
            f.al.c11 = src[(size_t)tri(P, P) * V];
            f.al.c12 = src[(size_t)tri(P + 1, P) * V];
            f.al.c22 = src[(size_t)tri(P + 1, P + 1) * V];
            const double nm = src[(size_t)(nCov + P + 2) * V];
            const double nv = src[(size_t)tri(P + 2, P + 2) * V];
            st.b = nv / nm;
            st.c = nm / st.b;
        }
        else
        {
            ar_initial_alpha(ka, f.al);
            st.b = ka.cfg.noise_post_b[0];
            st.c = ka.cfg.noise_post_c[0];
        }
        f.al.a1 = f.al.mean[0];
        f.al.a2 = f.al.c11 + f.al.mean[0] * f.al.mean[0];
        st.c = ka.cfg.noise_prior_c[0] + ((double)ka.cfg.n_times - 1) * 0.5;
#pragma unroll
        for (int w = 0; w < 3; w++)
            f.f.kk[w] = f.f.tr[w] = 0;
    }
    template <class Model>
    static __device__ __forceinline__ int recentre(const SpatialArgs &sa, const ModelArgs &ma, int v, const double (&centre)[P],
        Full &f, bool precise, const uint8_t *)
    {
        return recentre_ar_feed<Model, P>(sa.ka, ma, ArStridedFeed<P>{ sa.ka, v }, centre, f.mo, precise);
    }
    // Ar1cNoiseModel::UpdateNoise = UpdateAlpha + UpdatePhi (noisemodel_ar.cc:405-417, 447-556)
    template <class Model>
    static __device__ __forceinline__ int update_noise(const SpatialArgs &sa, const ModelArgs &ma, int v, VoxelState<P> &st, Full &f,
        const uint8_t *)
    {
        ar_residuals_feed<Model, P>(sa.ka, ma, ArStridedFeed<P>{ sa.ka, v }, st, f.mo, f.f);
        return update_noise_ar<P>(sa.ka, st, f.al, f.f);
    }
    static __device__ __forceinline__ bool F_full(const SpatialArgs &sa, VoxelState<P> &st, Full &f, bool at_centre, double Fprior,
        double &F, bool &finite)
    {
        if (at_centre)
        {
            ArForms c;
            ar_residuals_at_centre<P>(st, f.mo, c);
            return calc_free_energy_ar<P>(sa.ka, st, f.al, c, Fprior, F, finite);
        }
        return calc_free_energy_ar<P>(sa.ka, st, f.al, f.f, Fprior, F, finite);
    }
    // J'QJ, J'Q(y - g), (y - g)'Q(y - g) with the marginal Q of the current alpha posterior; E[phi] stays in (b, c)
    static __device__ __forceinline__ void effective(const SpatialArgs &, const Full &f, VoxelState<P> &, Moments<P> &w)
    {
        const ArMoments<P> &mo = f.mo;
        const ArAlpha &al = f.al;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
#pragma unroll
            for (int j = 0; j <= i; j++)
                w.A[tri(i, j)] = ar_JMJ<P>(mo, 0, i, j) + al.a1 * ar_JMJ<P>(mo, 1, i, j) + al.a2 * ar_JMJ<P>(mo, 2, i, j);
            w.u[i] = ar_JMr<P>(mo, 0, i) + al.a1 * ar_JMr<P>(mo, 1, i) + al.a2 * ar_JMr<P>(mo, 2, i);
            w.ml[i] = mo.ml[i];
        }
        w.s = ar_rMr<P>(mo, 0) + al.a1 * ar_rMr<P>(mo, 1) + al.a2 * ar_rMr<P>(mo, 2);
        w.precise = false;
    }
    static __device__ __forceinline__ bool sweep_F(const SpatialArgs &sa, int v, VoxelState<P> &st, const Moments<P> &, bool at_centre,
        double Fprior, double &F, bool &finite)
    {
        Full f;
        load_full(sa, v, f);
        if (!at_centre)
        {
#pragma unroll
            for (int w = 0; w < 3; w++)
            {
                bool lost;
                ar_forms<P>(st, f.mo, w, 0.0, f.f.kk[w], f.f.tr[w], lost);
            }
        }
        return F_full(sa, st, f, at_centre, Fprior, F, finite);
    }
    // Ar1cParams::OutputAsMVN (noisemodel_ar.cc:287-300): (alpha_1, alpha_2, phi)
    static __device__ __forceinline__ void pack_noise(const SpatialArgs &sa, int v, double *dst)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        constexpr int n = P + 3, nCov = n * (n + 1) / 2;
        ArAlpha al;
        load_alpha(sa, v, al);
        const double b = sa.state[(size_t)L::B * V + v], c = sa.state[(size_t)L::C * V + v];
#pragma unroll
        for (int r = P; r < n; r++)
#pragma unroll
            for (int cc = 0; cc <= r; cc++)
                dst[(size_t)tri(r, cc) * V] = 0.0;
        dst[(size_t)tri(P, P) * V] = al.c11;
        dst[(size_t)tri(P + 1, P) * V] = al.c12;
        dst[(size_t)tri(P + 1, P + 1) * V] = al.c22;
        dst[(size_t)tri(P + 2, P + 2) * V] = b * b * c;
        dst[(size_t)(nCov + P) * V] = al.mean[0];
        dst[(size_t)(nCov + P + 1) * V] = al.mean[1];
        dst[(size_t)(nCov + P + 2) * V] = b * c;
    }
};

// ---- AR(1) in its general form: two interleaved echoes, 2 / 3 / 4 AR coefficients (ar1-cross-terms none / same / dual) ----
// The two-echo lane kernel's scheme (vb_lane_arn_kernel.h) inside the spatial loop: the twelve lines' moments are not
// kept; the state image holds the EFFECTIVE moments J'XJ, J'X(y - g) (X = sum_n E[phi_n] Q_n with the alpha posterior's
// marginal weights; B = C = 1) that the first sweep needs, the alpha posterior, the two precisions' posteriors and the
// twelve line scalars S_l = k'M_l k + tr(Sigma J'M_l J) of the last pass. Per iteration the second-sweep kernel streams
// the series twice, as the voxelwise kernel does: pass 2 about the OLD centre with the first sweep's new means and
// covariance (S_l -> UpdateAlpha, UpdatePhi, noisemodel_ar.cc:447-556, and F "phi"), then the re-centre's pass 1 with
// the new alpha / phi posterior (effective moments, the at-centre S_l for F "lin" and the next iteration's "before").
// The first sweep's F "theta" (new means, old centre) is a pass of its own - the policy carries the MODEL for it; it
// runs only where F is evaluated.
template <class Model, int P, int NA>
struct SpArN
{
    typedef SpLayout<P> L;
    static constexpr int PT = L::PT, NT = NA * (NA + 1) / 2;
    static constexpr int AM = L::ROWS, AC = AM + NA, ALD = AC + NT, PB = ALD + 1, PC = PB + 2, SS = PC + 2;
    static constexpr int EXTRA_ROWS = NA + NT + 1 + 4 + 12;
    static constexpr int WAVES = 1; // (as vb_lane_arn_kernel: what does not fit 256 registers lands in the other half of the file)
    static constexpr bool LDS_CLASSES = false;
    struct Full
    {
        ArnAlpha<NA> al;
        double pb[2], pc[2];
        ArnMoments<P> mo;
        double S[2][6];
        double Sig[PT]; // the posterior covariance the next forms pass belongs to
    };
    static __device__ __forceinline__ int n_noise_out(const KernelArgs &)
    {
        return NA + 2;
    }
    static __device__ __forceinline__ ModelArgs model_args(const KernelArgs &ka)
    {
        ModelArgs ma;
        ma.iopt0 = ka.cfg.model_iopt[0];
        ma.dopt0 = ka.cfg.model_dopt[0];
        ma.design = ka.cfg.design;
        return ma;
    }
    static __device__ __forceinline__ void load_full(const SpatialArgs &sa, int v, Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        const double *p = sa.state + v;
#pragma unroll
        for (int i = 0; i < NA; i++)
            f.al.mean[i] = p[(size_t)(AM + i) * V];
#pragma unroll
        for (int i = 0; i < NT; i++)
            f.al.cov[i] = p[(size_t)(AC + i) * V];
        f.al.logdetPrec = p[(size_t)ALD * V];
        f.al.update_marginal();
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            f.pb[e] = p[(size_t)(PB + e) * V];
            f.pc[e] = p[(size_t)(PC + e) * V];
        }
#pragma unroll
        for (int i = 0; i < PT; i++)
        {
            f.mo.A[i] = p[(size_t)(L::A + i) * V];
            f.Sig[i] = p[(size_t)(L::SIG + i) * V];
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            f.mo.u[i] = p[(size_t)(L::U + i) * V];
            f.mo.ml[i] = p[(size_t)(L::ML + i) * V];
        }
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int k = 0; k < 6; k++)
                f.S[n][k] = p[(size_t)(SS + n * 6 + k) * V];
    }
    static __device__ __forceinline__ void store_full(const SpatialArgs &sa, int v, const Full &f)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        double *p = sa.state + v;
#pragma unroll
        for (int i = 0; i < NA; i++)
            p[(size_t)(AM + i) * V] = f.al.mean[i];
#pragma unroll
        for (int i = 0; i < NT; i++)
            p[(size_t)(AC + i) * V] = f.al.cov[i];
        p[(size_t)ALD * V] = f.al.logdetPrec;
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            p[(size_t)(PB + e) * V] = f.pb[e];
            p[(size_t)(PC + e) * V] = f.pc[e];
        }
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int k = 0; k < 6; k++)
                p[(size_t)(SS + n * 6 + k) * V] = f.S[n][k];
    }
    // Ar1cNoiseModel::HardcodedInitialDists (noisemodel_ar.cc:379-403) or Ar1cParams::InputFromMVN (:302-316), then
    // Ar1cNoiseModel::Precalculate (:749-769)
    static __device__ __forceinline__ void init_noise(const SpatialArgs &sa, const double *src, VoxelState<P> &st, Full &f)
    {
        const KernelArgs &ka = sa.ka;
        const size_t V = (size_t)ka.cfg.n_voxels;
        constexpr int n = P + NA + 2, nCov = n * (n + 1) / 2;
        if (src)
        {
#pragma unroll
            for (int i = 0; i < NA; i++)
            {
                f.al.mean[i] = src[(size_t)(nCov + P + i) * V];
#pragma unroll
                for (int j = 0; j <= i; j++)
                    f.al.cov[tri(i, j)] = src[(size_t)tri(P + i, P + j) * V];
            }
            double inv[NT], la;
            int sg;
            (void)ldl_inverse<NA>(f.al.cov, inv, la, sg);
            f.al.logdetPrec = -la;
#pragma unroll
            for (int e = 0; e < 2; e++)
            {
                const double nm = src[(size_t)(nCov + P + NA + e) * V];
                const double nv = src[(size_t)tri(P + NA + e, P + NA + e) * V];
                f.pb[e] = nv / nm;
                f.pc[e] = nm / f.pb[e];
            }
        }
        else
        {
            arn_initial_alpha<NA>(ka, f.al);
#pragma unroll
            for (int e = 0; e < 2; e++)
                f.pb[e] = ka.cfg.noise_post_b[e];
        }
        f.al.update_marginal();
#pragma unroll
        for (int e = 0; e < 2; e++)
            f.pc[e] = ka.cfg.noise_prior_c[e] + ((double)(ka.cfg.n_times / 2) - 1) * 0.5;
#pragma unroll
        for (int i = 0; i < PT; i++)
            f.Sig[i] = st.Sig[i];
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int k = 0; k < 6; k++)
                f.S[n][k] = 0;
    }
    // the re-centre: effective moments with the present alpha / phi posterior and the line scalars at the centre
    template <class M2>
    static __device__ __forceinline__ int recentre(const SpatialArgs &sa, const ModelArgs &ma, int v, const double (&centre)[P], Full &f,
        bool precise, const uint8_t *)
    {
        double cw[2][6], zero[P];
#pragma unroll
        for (int i = 0; i < P; i++)
            zero[i] = 0;
        arn_coefficients<NA>(f.al, f.pb, f.pc, cw);
        return arn_pass<Model, P, NA, true, true>(sa.ka, ma, ArStridedFeed<P>{ sa.ka, v }, centre, cw, f.mo, f.Sig, zero, f.S, precise);
    }
    // Ar1cNoiseModel::UpdateNoise = UpdateAlpha + UpdatePhi (noisemodel_ar.cc:405-417, 447-556) from the line scalars of a
    // pass about the centre the moments belong to, with the first sweep's means and covariance
    template <class M2>
    static __device__ __forceinline__ int update_noise(const SpatialArgs &sa, const ModelArgs &ma, int v, VoxelState<P> &st, Full &f,
        const uint8_t *)
    {
        double cw[2][6], nd[P];
#pragma unroll
        for (int i = 0; i < P; i++)
            nd[i] = f.mo.ml[i] - st.m[i];
#pragma unroll
        for (int i = 0; i < PT; i++)
            f.Sig[i] = st.Sig[i];
        arn_coefficients<NA>(f.al, f.pb, f.pc, cw);
        // (the Jacobian as the linearisation about ml computed it: sp_load's rule for `precise`)
        (void)arn_pass<Model, P, NA, false, true>(sa.ka, ma, ArStridedFeed<P>{ sa.ka, v }, f.mo.ml, cw, f.mo, st.Sig, nd, f.S,
            sa.it < sa.ka.precise_passes);
        int status = update_noise_arn<NA>(sa.ka, f.al, f.pb, f.pc, f.S);
        if (status == FVB_OK && sa.locked_linear)
        {
            // no re-centre follows (inference_vb.cc:695-696): the effective moments about the same centre with the new
            // alpha / phi posterior (the line scalars stay those of the pass above: F "lin" is F "phi" then)
            double keep[2][6], zero[P];
#pragma unroll
            for (int i = 0; i < P; i++)
                zero[i] = 0;
            arn_coefficients<NA>(f.al, f.pb, f.pc, cw);
            double centre[P];
#pragma unroll
            for (int i = 0; i < P; i++)
                centre[i] = f.mo.ml[i];
            status = arn_pass<Model, P, NA, true, false>(sa.ka, ma, ArStridedFeed<P>{ sa.ka, v }, centre, cw, f.mo, st.Sig, zero, keep,
                sa.it < sa.ka.precise_passes);
        }
        return status;
    }
    static __device__ __forceinline__ bool F_full(const SpatialArgs &sa, VoxelState<P> &st, Full &f, bool, double Fprior, double &F,
        bool &finite)
    {
        return calc_free_energy_arn<P, NA>(sa.ka, st, f.al, f.pb, f.pc, f.S, Fprior, F, finite);
    }
    static __device__ __forceinline__ void effective(const SpatialArgs &, const Full &f, VoxelState<P> &st, Moments<P> &w)
    {
#pragma unroll
        for (int i = 0; i < PT; i++)
            w.A[i] = f.mo.A[i];
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            w.u[i] = f.mo.u[i];
            w.ml[i] = f.mo.ml[i];
        }
        w.s = 0;
        w.precise = false;
        st.b = st.c = 1; // (E[phi] is inside the effective moments)
    }
    // the first sweep's free energy: "before" reads the line scalars the last re-centre left (same centre, same
    // covariance); "theta" - new means and covariance about the old centre - streams the series once
    static __device__ __forceinline__ bool sweep_F(const SpatialArgs &sa, int v, VoxelState<P> &st, const Moments<P> &, bool at_centre,
        double Fprior, double &F, bool &finite)
    {
        Full f;
        load_full(sa, v, f);
        if (!at_centre)
        {
            double cw[2][6], nd[P];
#pragma unroll
            for (int i = 0; i < P; i++)
                nd[i] = f.mo.ml[i] - st.m[i];
            arn_coefficients<NA>(f.al, f.pb, f.pc, cw);
            (void)arn_pass<Model, P, NA, false, true>(sa.ka, model_args(sa.ka), ArStridedFeed<P>{ sa.ka, v }, f.mo.ml, cw, f.mo, st.Sig, nd, f.S,
                sa.it < sa.ka.precise_passes);
        }
        return calc_free_energy_arn<P, NA>(sa.ka, st, f.al, f.pb, f.pc, f.S, Fprior, F, finite);
    }
    // Ar1cParams::OutputAsMVN (noisemodel_ar.cc:287-300): (alphas, phi_1, phi_2)
    static __device__ __forceinline__ void pack_noise(const SpatialArgs &sa, int v, double *dst)
    {
        const size_t V = (size_t)sa.ka.cfg.n_voxels;
        constexpr int n = P + NA + 2, nCov = n * (n + 1) / 2;
        const double *p = sa.state + v;
#pragma unroll
        for (int r = P; r < n; r++)
#pragma unroll
            for (int c = 0; c <= r; c++)
                dst[(size_t)tri(r, c) * V] = 0.0;
#pragma unroll
        for (int i = 0; i < NA; i++)
        {
#pragma unroll
            for (int j = 0; j <= i; j++)
                dst[(size_t)tri(P + i, P + j) * V] = p[(size_t)(AC + tri(i, j)) * V];
            dst[(size_t)(nCov + P + i) * V] = p[(size_t)(AM + i) * V];
        }
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            const double b = p[(size_t)(PB + e) * V], c = p[(size_t)(PC + e) * V];
            dst[(size_t)tri(P + NA + e, P + NA + e) * V] = b * b * c;
            dst[(size_t)(nCov + P + NA + e) * V] = b * c;
        }
    }
};

// cfg.phi_index into LDS (dynamic, T bytes) for the policies that read it per timepoint; every lane takes part
template <class NZ>
__device__ __forceinline__ const uint8_t *sp_classes(const KernelArgs &ka)
{
    extern __shared__ uint8_t sp_class_table[];
    if (!NZ::LDS_CLASSES)
        return nullptr;
    const int T = ka.cfg.n_times;
    for (int t = threadIdx.x; t < T; t += 64)
        sp_class_table[t] = ka.cfg.phi_index ? ka.cfg.phi_index[t] : (uint8_t)0;
    __syncthreads();
    return sp_class_table;
}

// ---- setup: Vb::SetupPerVoxelDists (inference_vb.cc:207-247) -----------------------------------------------
template <class Model, int P, class NZ>
__global__ __launch_bounds__(64, NZ::WAVES) void vb_spatial_setup_nz_kernel(const SpatialArgs sa)
{
    const KernelArgs &ka = sa.ka;
    constexpr int PT = P * (P + 1) / 2;
    const uint8_t *classes = sp_classes<NZ>(ka);
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    if (v >= ka.cfg.n_voxels)
        return;
    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;
    VoxelState<P> st;
    Moments<P> mo;
    typename NZ::Full full;
    st.b = st.c = 1;
    if (ka.cfg.init_mvn)
    {
        const int n = P + NZ::n_noise_out(ka);
        const int nCov = n * (n + 1) / 2;
        const double *src = ka.cfg.init_mvn + v;
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = src[(size_t)i * V];
#pragma unroll
        for (int i = 0; i < P; i++)
            st.m[i] = src[(size_t)(nCov + i) * V];
        NZ::init_noise(sa, src, st, full);
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
            double data_max = load_data(ka, v);
            for (int t = 1; t < T; t++)
            {
                const double y = load_data(ka, (size_t)t * V + v);
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
        NZ::init_noise(sa, nullptr, st, full);
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
    if (Model::host_evaluated)
    {
        ma.lin_T = T;
        ma.lin = sa.lin_next + (size_t)v * T * (P + 1);
    }
    double centre[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        centre[i] = sa.locked_centres ? sa.locked_centres[(size_t)i * V + v] : st.m[i];
    const int status = NZ::template recentre<Model>(sa, ma, v, centre, full, true, classes);
    sa.status[v] = status ? (status | 0x100) : 0;
    NZ::effective(sa, full, st, mo);
    sp_store_theta<P>(sa, v, st);
    sp_store_noise<P>(sa, v, st, mo);
    NZ::store_full(sa, v, full);
}

// ---- second sweep: UpdateNoise, ReCentre, F (inference_vb.cc:674-722), all voxels ------------------------------
template <class Model, int P, bool NEEDF, bool FAST, class NZ>
__global__ __launch_bounds__(64, NZ::WAVES) void vb_spatial_noise_nz_kernel(const SpatialArgs sa)
{
    const KernelArgs &ka = sa.ka;
    const uint8_t *classes = sp_classes<NZ>(ka);
    const int v = sa.owned_begin + blockIdx.x * 64 + threadIdx.x;
    if (v >= sa.owned_end)
        return;
    if (sa.status[v] != 0)
        return;
    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;
    VoxelState<P> st;
    Moments<P> mo;
    sp_load<P>(sa, v, st, mo); // (the effective moments)
    if (FAST)
    {
        if (!sp_complete_theta<P, NEEDF, NZ>(sa, v, st, mo))
            return;
    }
    typename NZ::Full full;
    NZ::load_full(sa, v, full);
    if (Model::host_evaluated) // the residual about the centre the moments belong to ...
    {
        ma.lin_T = ka.cfg.n_times;
        ma.lin = sa.lin_cur + (size_t)v * ka.cfg.n_times * (P + 1);
    }
    int status = NZ::template update_noise<Model>(sa, ma, v, st, full, classes);
    if (Model::host_evaluated) // ... and the re-centre about the means of this iteration's first sweep
        ma.lin = sa.lin_next + (size_t)v * ka.cfg.n_times * (P + 1);
    if (status == FVB_OK && !sa.locked_linear) // inference_vb.cc:695-696
        status = NZ::template recentre<Model>(sa, ma, v, st.m, full, sp_precise(sa), classes);
    if (status == FVB_OK && NEEDF)
    {
        // only the last of the reference's four F evaluations per iteration is observable; it uses the prior
        // term of the LAST voxel of the first sweep (inference_vb.cc:612,689,702)
        st.covValid = true;
        st.precValid = false;
        double F;
        bool finite = true;
        if (!NZ::F_full(sa, st, full, !sa.locked_linear, *sa.fprior_last, F, finite))
            status = FVB_BAD_RESULT;
        else if (!finite)
            status = FVB_BAD_FREE_ENERGY;
        else if (ka.out.free_energy)
            ka.out.free_energy[v] = F;
    }
    if (status != FVB_OK)
        sa.status[v] = status;
    NZ::effective(sa, full, st, mo);
    sp_store_noise<P>(sa, v, st, mo);
    NZ::store_full(sa, v, full);
}

// ---- result image (inference_vb.cc:757-762) --------------------------------------------------------------------
template <int P, class NZ>
__global__ __launch_bounds__(256) void vb_spatial_pack_nz_kernel(const SpatialArgs sa)
{
    typedef SpLayout<P> L;
    const KernelArgs &ka = sa.ka;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= ka.cfg.n_voxels)
        return;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const int n = P + NZ::n_noise_out(ka);
    const int nCov = n * (n + 1) / 2;
    const double *p = sa.state + v;
    double *dst = ka.out.mvn + v;
#pragma unroll
    for (int i = 0; i < L::PT; i++)
        dst[(size_t)i * V] = p[(size_t)(L::SIG + i) * V];
    NZ::pack_noise(sa, v, dst);
#pragma unroll
    for (int i = 0; i < P; i++)
        dst[(size_t)(nCov + i) * V] = p[(size_t)(L::M + i) * V];
    dst[(size_t)(nCov + n) * V] = 1.0;
    if (ka.out.status)
        ka.out.status[v] = sa.status[v];
    if (ka.out.iterations)
        ka.out.iterations[v] = sa.it;
    if (ka.out.free_energy && !ka.cfg.need_f)
        ka.out.free_energy[v] = 1234.5678;
}

#endif // __HIPCC__

// which statistics the state image carries beyond SpLayout
enum
{
    FVB_SPNZ_WHITE = 0,    // white noise, one precision: the kernels of vb_spatial.h
    FVB_SPNZ_PATTERN2 = 1, // white noise, 2 precisions
    FVB_SPNZ_PATTERN4 = 2, // white noise, 3 - 4 precisions
    FVB_SPNZ_AR1 = 3,      // AR(1), one echo
    FVB_SPNZ_ARN2 = 4,     // AR(1), two echoes, ar1-cross-terms none (2 AR coefficients)
    FVB_SPNZ_ARN3 = 5,     // ... same (3)
    FVB_SPNZ_ARN4 = 6,     // ... dual (4)
    FVB_SPNZ_PATTERN8 = 7  // white noise, 5 - 8 precisions
};
SpatialKernels get_spatial_kernels_nz_pattern8(int model, int P, bool need_f); // vb_spatial_nz_p8_*.hip
SpatialKernels get_spatial_kernels_nz_arn(int model, int P, bool need_f, int kind); // vb_spatial_nz_arn_*.hip
SpatialKernels get_spatial_kernels_nz_poly(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_linear(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_exp(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_host(int P, bool need_f, int kind);

#if defined(__HIPCC__)
// one (model, P, noise policy): the white-noise table with the policy's set-up / second-sweep / result kernels and, when
// F is evaluated, its first-sweep kernels (without F the first sweep does not depend on the noise model)
template <template <int> class MODEL, int PP, class NZ>
SpatialKernels spatial_nz_table(bool need_f, const char *name)
{
    SpatialKernels k{};
    k.setup = vb_spatial_setup_nz_kernel<MODEL<PP>, PP, NZ>;
    k.ak_partial = vb_spatial_ak_partial_kernel<PP>;
    k.ak_reduce = vb_spatial_ak_reduce_kernel<PP>;
    k.ak_final = vb_spatial_ak_final_kernel<PP>;
    k.theta = need_f ? (SpatialThetaFn)vb_spatial_theta_kernel<PP, true, NZ> : (SpatialThetaFn)vb_spatial_theta_kernel<PP, false>;
    k.noise = need_f ? (SpatialKernelFn)vb_spatial_noise_nz_kernel<MODEL<PP>, PP, true, false, NZ>
                     : (SpatialKernelFn)vb_spatial_noise_nz_kernel<MODEL<PP>, PP, false, false, NZ>;
    k.pack = vb_spatial_pack_nz_kernel<PP, NZ>;
    k.state_rows = SpLayout<PP>::ROWS + NZ::EXTRA_ROWS;
    k.name = name;
    k.prep = need_f ? (SpatialPrepFn)vb_spatial_prep_kernel<PP, true, NZ> : (SpatialPrepFn)vb_spatial_prep_kernel<PP, false>;
    k.noise_fast = need_f ? (SpatialKernelFn)vb_spatial_noise_nz_kernel<MODEL<PP>, PP, true, true, NZ>
                          : (SpatialKernelFn)vb_spatial_noise_nz_kernel<MODEL<PP>, PP, false, true, NZ>;
    k.slab_sweep[0] = vb_spatial_slab_sweep_kernel<1>;
    k.slab_sweep[1] = vb_spatial_slab_sweep_kernel<2>;
    k.slab_sweep[2] = vb_spatial_slab_sweep_kernel<(PP <= 4 ? 4 : 8)>;
    k.lds_classes = NZ::LDS_CLASSES ? 1 : 0;
    return k;
}
#define FVB_SPATIAL_ARN_CASE(MODEL, TAG, PP)                                                                 \
    case PP:                                                                                                 \
        if (kind == FVB_SPNZ_ARN2)                                                                           \
            return spatial_nz_table<MODEL, PP, SpArN<MODEL<PP>, PP, 2> >(need_f, "spatial<" TAG "," #PP ",ar2:2>"); \
        if (kind == FVB_SPNZ_ARN3)                                                                           \
            return spatial_nz_table<MODEL, PP, SpArN<MODEL<PP>, PP, 3> >(need_f, "spatial<" TAG "," #PP ",ar2:3>"); \
        if (kind == FVB_SPNZ_ARN4)                                                                           \
            return spatial_nz_table<MODEL, PP, SpArN<MODEL<PP>, PP, 4> >(need_f, "spatial<" TAG "," #PP ",ar2:4>"); \
        return SpatialKernels{};
#define FVB_SPATIAL_P8_CASE(MODEL, TAG, PP)                                                                  \
    case PP:                                                                                                 \
        return spatial_nz_table<MODEL, PP, SpPattern<PP, 8> >(need_f, "spatial<" TAG "," #PP ",pattern8>");
#define FVB_SPATIAL_NZ_CASE(MODEL, TAG, PP)                                                                  \
    case PP:                                                                                                 \
        if (kind == FVB_SPNZ_PATTERN2)                                                                       \
            return spatial_nz_table<MODEL, PP, SpPattern<PP, 2> >(need_f, "spatial<" TAG "," #PP ",pattern2>"); \
        if (kind == FVB_SPNZ_PATTERN4)                                                                       \
            return spatial_nz_table<MODEL, PP, SpPattern<PP, 4> >(need_f, "spatial<" TAG "," #PP ",pattern4>"); \
        if (kind == FVB_SPNZ_AR1)                                                                            \
            return spatial_nz_table<MODEL, PP, SpAr1<PP> >(need_f, "spatial<" TAG "," #PP ",ar1>");          \
        return SpatialKernels{};
#endif

} // namespace fvb
