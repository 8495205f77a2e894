/*
 * vb_spatial.h - spatial VB (Vb::DoCalculationsSpatial, inference_vb.cc:578-767) on MI355X.
 *
 * The reference sweeps the voxels in index order twice per iteration. In the first sweep voxel
 * v's spatial prior (SpatialPrior::ApplyToMVN, priors.cc:346-488) reads its neighbours' CURRENT
 * posterior means - already updated for u < v, previous iteration for u > v (Gauss-Seidel) -
 * and then UpdateTheta changes v's own mean. That order is kept EXACTLY: with
 * level(v) = x + y + z (first neighbours) or x + 2y + 3z (second neighbours too) every listed
 * neighbour of v with a smaller index has a strictly smaller level and every one with a larger
 * index a strictly larger level, so voxels of one level are independent and the levels are
 * processed in ascending order - one launch each (vb_spatial_theta_kernel, the exact form that
 * takes everything), or the split form further down. The second sweep (UpdateNoise, ReCentre, F) has no coupling and is
 * one lane-per-voxel launch over all voxels (vb_spatial_noise_kernel) sharing its device
 * functions with the voxelwise kernel. The global smoothing precisions aK (CalculateaK,
 * priors.cc:221-344) are two deterministic tree reductions per spatial parameter per iteration.
 *
 * Per-voxel state lives in HBM between launches as a [rows][V] image (SpLayout).
 */
#pragma once

#include "vb_lane_kernel.h"

namespace fvb
{
// Row offsets of the persistent per-voxel state image
template <int P>
struct SpLayout
{
    static constexpr int PT = P * (P + 1) / 2;
    static constexpr int M = 0;            // posterior means [P]
    static constexpr int SIG = M + P;      // posterior covariance [PT]
    static constexpr int LOGDET = SIG + PT; // log|det Lambda|
    static constexpr int PM = LOGDET + 1;  // prior means [P]
    static constexpr int PPREC = PM + P;   // prior precisions (diagonal) [P]
    static constexpr int B = PPREC + P;    // noise scale
    static constexpr int C = B + 1;        // noise shape
    static constexpr int A = C + 1;        // J'J [PT]
    static constexpr int U = A + PT;       // J'r [P]
    static constexpr int S = U + P;        // r'r
    static constexpr int ML = S + 1;       // linearisation centre [P]
    static constexpr int ROWS = ML + P;
};

struct SpatialArgs
{
    KernelArgs ka;
    double *state;          // [SpLayout<P>::ROWS][V]
    const int32_t *nn;      // [V][6] first neighbours (0-based, -1 = none), reference order
    const int32_t *nn_dir;  // [V] which offset (+x -x +y -y +z -z = 0 .. 5) each list slot was found with, 3 bits per slot
    const int32_t *order;   // voxel ids sorted by level
    double *aK;             // [P] smoothing precision per (spatial) parameter
    double *partials;       // [n_blocks][P][2] reduction scratch: one entry per SEGMENT of the voxel list
    const int32_t *seg_start; // [n_blocks + 1] first voxel of each segment: a z-plane, cut every 4096 voxels
    double *fprior_last;    // F contribution of the priors of the LAST voxel of the first sweep
    int32_t *status;        // [V]
    int32_t spatial_dims;
    int32_t update_first_iter;
    double spatial_speed, q1, q2;
    int32_t it;
    int32_t level_begin, level_count;
    int32_t n_blocks;
    // Multi-GPU slabs: this process updates voxels [owned_begin, owned_end) of its local list; the
    // rest are ghost copies of neighbouring slabs' boundary planes (read as neighbours, never
    // updated here, refreshed by the halo exchange). One process: the whole list.
    int32_t owned_begin, owned_end;
    int32_t n_voxels_global; // the V of h_K = V/2 + q2 (priors.cc:321)
    double *ak_sums;         // [P][2] (trace_term, term2): this slab's, then the all-reduced ones
    // host-evaluated models (HostLinModel): [V][n_times * (P + 1)] linearisations, g then J per voxel - about the
    // centre the moments in the state belong to (lin_cur) and about the means the second sweep re-centres on (lin_next)
    const double *lin_cur, *lin_next;
    // ---- the split first sweep (below); NULL / 0 when the per-level launches are used. Only the parameters with a
    // first-neighbour prior (types M, m) take part in it: types P and p read no neighbour's value as the reference
    // codes them (priors.cc:455), their prior mean is complete in the prep kernel ----
    const int32_t *pos_of;   // [V] position of a voxel in the slab-major numbering
    int32_t n_pos;           // positions (padded to a multiple of 16)
    int32_t n_spatial;       // parameters of types M, m ...
    int32_t spatial_param[FVB_MAX_PARAMS]; // ... and which they are
    double *sw_x;            // [n_spatial][n_pos] OUT: the new means of those parameters
    double *sw_pm;           // [n_spatial][n_pos] OUT: their prior means
    // the record of a voxel, eq (20) for a swept parameter k with the terms of the OTHER parameters summed already:
    //   m_k = fma(Sig_kk', rhs_k', ...fma(Sig_kk, rhs_k, pre_k)),  rhs_k = fma(pprec_k, mu0_k, base_k)  (k, k': swept)
    double *sw_pre;          // [n_spatial][n_pos] sum over the parameters that are not swept of Sig_kj rhs_j (index order)
    double *sw_rhsk;         // [n_spatial][n_pos] base_k
    double *sw_pprec;        // [n_spatial][n_pos] prior precision of the swept parameter
    double *sw_q;            // [n_spatial][n_pos] (1 / prior precision) x spatial precision
    double *sw_sigk;         // [n_spatial][n_spatial][n_pos] Sig_kk'
    double *sw_nbr;          // [n_spatial][3][n_pos] the means the +x, +y, +z neighbours had BEFORE the sweep (+0.0: none)
    int32_t *sw_npos;        // [4][n_pos] positions of the -x, -y, -z and +z neighbours, -1 = none
    int32_t *sw_alive;       // [n_pos] 0 = the voxel takes no part in the sweep, else 1 + its number of live neighbours
    // positions are slab-major, a slab = sl_dz z-planes, inside a slab level-major (level = x + y + z): the voxels of
    // one (slab, level) are a contiguous RUN; sl_first_run[s] .. sl_first_run[s + 1] are slab s's runs
    const int32_t *sw_level_pos;   // [n_runs] first position of a run
    const int32_t *sw_level_count; // [n_runs] voxels in it
    int32_t n_levels;              // n_runs
    unsigned long long *sw_gran; // [n_spatial][n_pos][2] the INBOX of a voxel: the mean of its z-1 neighbour when that lives
                                 // in the slab below, as two self-validating 8-byte granules {low / high half of the
                                 // double, serial number of the sweep that wrote it}
    uint32_t sw_serial;      // this sweep's serial number (1, 2, ...; the buffer starts zeroed)
    int32_t n_slabs;
    const int32_t *sl_first_run; // [n_slabs + 1]
    int32_t sl_width;        // lanes that work on one run (a multiple of 64 that divides 1024)
    int32_t sl_max_run;      // longest run (the LDS buffers hold two of them per spatial parameter)
    // the prep kernel walks the volume in tiles of 8 x 8 voxels of a plane (tile_nx x tile_ny tiles per plane, planes
    // tile_z0 ...): the voxel at (x, y, z) is dense[z xsize ysize + y xsize + x - dense_base], -1 = none. 0 tiles: in
    // index order, 64 consecutive voxels per wavefront
    const int32_t *dense;
    long long dense_base, dense_span;
    int32_t xsize, ysize, tile_nx, tile_ny, tile_z0, n_tiles;
    int32_t *sw_flags;       // [0] != 0: the split sweep met a case it does not handle (a voxel failed during the sweep, a
                             // non-finite mean next to a type P / p prior, an inbox that never arrived): the run is
                             // repeated with the per-level launches
    // ---- the slab sweep across DEVICES (fabber_vb_run_spatial_host_multi): a device owns a range of z-planes of the
    // volume and holds ghost copies of the planes next to it; in sw_npos a neighbour that is a ghost BELOW the owned
    // range reads FVB_NP_BELOW (its mean of this sweep arrives in the voxel's inbox, written by the device below), one
    // that is a ghost ABOVE reads FVB_NP_ABOVE (it contributes the mean the last halo exchange left, and this
    // voxel's new mean goes into ITS inbox on the device above: sw_gran_up at position sw_up_pos[pos]) ----
    unsigned long long *sw_gran_up; // the granules of the slab above (peer-mapped memory), or NULL
    const int32_t *sw_up_pos;       // [n_pos] position of the z+1 neighbour in the numbering of the slab above, -1 = none
    int32_t up_n_pos;               // n_pos of the slab above
    int32_t sl_remote;              // 1: inboxes are written by another device: system-scope accesses
    // ---- noise models other than white noise with one precision (vb_spatial_noise.h) ----
    double nz_count[8];      // noise-pattern: timepoints of each class (trace of Q_k, noisemodel_white.cc:207-225)
    int32_t locked_linear;   // locked-linear-from-mvn: the second sweep does not re-centre (inference_vb.cc:695-696)
    const double *locked_centres; // [P][V] the fixed centres (set-up re-centre, inference_vb.cc:225-232), or NULL
};

enum
{
    FVB_NP_NONE = -1,
    FVB_NP_BELOW = -2,
    FVB_NP_ABOVE = -3
};

#if defined(__HIPCC__)

template <int P>
__device__ __forceinline__ void sp_load(const SpatialArgs &sa, int v, VoxelState<P> &st, Moments<P> &mo)
{
    typedef SpLayout<P> L;
    const size_t V = (size_t)sa.ka.cfg.n_voxels;
    const double *p = sa.state + v;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        st.m[i] = p[(size_t)(L::M + i) * V];
        st.pm[i] = p[(size_t)(L::PM + i) * V];
        st.pprec[i] = p[(size_t)(L::PPREC + i) * V];
        mo.u[i] = p[(size_t)(L::U + i) * V];
        mo.ml[i] = p[(size_t)(L::ML + i) * V];
    }
#pragma unroll
    for (int i = 0; i < L::PT; i++)
    {
        st.Sig[i] = p[(size_t)(L::SIG + i) * V];
        mo.A[i] = p[(size_t)(L::A + i) * V];
    }
    st.logdetLam = p[(size_t)L::LOGDET * V];
    st.b = p[(size_t)L::B * V];
    st.c = p[(size_t)L::C * V];
    mo.s = p[(size_t)L::S * V];
    // how the linearisation these moments belong to evaluated the model: the set-up re-centre is linearisation 0, the
    // one that ends iteration it - 1 is linearisation it; the first ka.precise_passes of a run are pointwise (see
    // sp_precise below)
    mo.precise = sa.it < sa.ka.precise_passes;
    st.covValid = true;
    st.precValid = false;
}

// What the prep kernel of the split first sweep reads of a voxel's state: the means, the effective moments and the
// noise posterior - 2 P + PT + 2 of the image's rows (the covariance's diagonal too where an ARD prior follows the
// posterior). The prior, the covariance and log|det Lambda| are what it WRITES. With F evaluated it checks the free
// energy "before" (sweep_F) and needs all of it.
template <int P, bool NEEDF>
__device__ __forceinline__ void sp_load_prep(const SpatialArgs &sa, int v, VoxelState<P> &st, Moments<P> &mo)
{
    if (NEEDF)
        return sp_load<P>(sa, v, st, mo);
    typedef SpLayout<P> L;
    const size_t V = (size_t)sa.ka.cfg.n_voxels;
    const double *p = sa.state + v;
    bool any_ard = false;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        st.m[i] = p[(size_t)(L::M + i) * V];
        mo.u[i] = p[(size_t)(L::U + i) * V];
        mo.ml[i] = p[(size_t)(L::ML + i) * V];
        st.pm[i] = 0;
        st.pprec[i] = 0;
        any_ard |= (sa.ka.cfg.prior_type[i] == FVB_PRIOR_ARD);
    }
#pragma unroll
    for (int i = 0; i < L::PT; i++)
    {
        mo.A[i] = p[(size_t)(L::A + i) * V];
        st.Sig[i] = 0;
    }
    if (any_ard) // (uniform)
    {
#pragma unroll
        for (int i = 0; i < P; i++)
            st.Sig[tri(i, i)] = p[(size_t)(L::SIG + tri(i, i)) * V];
    }
    st.logdetLam = 0;
    st.b = p[(size_t)L::B * V];
    st.c = p[(size_t)L::C * V];
    mo.s = 0;
    mo.precise = sa.it < sa.ka.precise_passes;
    st.covValid = true;
    st.precValid = false;
}

template <int P>
__device__ __forceinline__ void sp_store_theta(const SpatialArgs &sa, int v, const VoxelState<P> &st)
{
    typedef SpLayout<P> L;
    const size_t V = (size_t)sa.ka.cfg.n_voxels;
    double *p = sa.state + v;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        p[(size_t)(L::M + i) * V] = st.m[i];
        p[(size_t)(L::PM + i) * V] = st.pm[i];
        p[(size_t)(L::PPREC + i) * V] = st.pprec[i];
    }
#pragma unroll
    for (int i = 0; i < L::PT; i++)
        p[(size_t)(L::SIG + i) * V] = st.Sig[i];
    p[(size_t)L::LOGDET * V] = st.logdetLam;
}

template <int P>
__device__ __forceinline__ void sp_store_noise(const SpatialArgs &sa, int v, const VoxelState<P> &st, const Moments<P> &mo)
{
    typedef SpLayout<P> L;
    const size_t V = (size_t)sa.ka.cfg.n_voxels;
    double *p = sa.state + v;
    p[(size_t)L::B * V] = st.b;
    p[(size_t)L::C * V] = st.c;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        p[(size_t)(L::U + i) * V] = mo.u[i];
        p[(size_t)(L::ML + i) * V] = mo.ml[i];
    }
#pragma unroll
    for (int i = 0; i < L::PT; i++)
        p[(size_t)(L::A + i) * V] = mo.A[i];
    p[(size_t)L::S * V] = mo.s;
}

// SpatialPrior::ApplyToMVN with second neighbours (types P, p; priors.cc:441-482) AS CODED:
//   spatial_mean = (8 sum(first neighbours) - sum(second neighbours)) * rec,   rec = 1 / (8 nn - nn2) in `int`
//   prior mean   = (1 / prior precision) (spatial precision x spatial_mean + prec0 mean0)
// :455 divides two ints; nn2 <= 5 nn, so the divisor is >= 3 and rec is 0 for every reachable neighbourhood:
// spatial_mean is +-0 - or NaN where a listed neighbour's mean is not finite - and the prior mean reads no
// neighbour's VALUE. The per-level kernel evaluates the expression as it stands (second_order_rec), NaN included;
// the split form takes the prior mean as pcov x (prec0 mean0) and leaves every run in which a mean next to such a
// prior is not finite to the per-level kernel (sw_flags).
__device__ __forceinline__ double second_order_rec(int nn, int nn2)
{
    const int d = 8 * nn - nn2;
    return (d == 0) ? 0.0 : (double)(1 / d); // (d == 0 cannot happen: the reference would divide by zero there)
}
__device__ __forceinline__ double second_order_mean(double contrib_nn, double contrib_nn2, double rec)
{
    return __builtin_fma(8.0, contrib_nn, contrib_nn2) * rec;
}
__device__ __forceinline__ double second_order_pm(double pcov, double spatial_prec, double spatial_mean, double prec0, double mean0)
{
    return pcov * __builtin_fma(spatial_prec, spatial_mean, prec0 * mean0);
}

__device__ __forceinline__ bool is_spatial_type(int t)
{
    return t >= FVB_PRIOR_SPATIAL_M;
}
// ... and the ones whose prior mean waits for the neighbours' current means (the ordered sweep of the split form)
__device__ __forceinline__ bool is_swept_type(int t)
{
    return t == FVB_PRIOR_SPATIAL_M || t == FVB_PRIOR_SPATIAL_m;
}

// The re-centre that ends iteration sa.it is the run's linearisation number sa.it + 1 (the set-up's is number 0). Like
// the voxelwise kernels, the first ka.precise_passes = 2 linearisations of a run evaluate the model pointwise
// (vb_lane_kernel.h, recentre): round 2 did so for the set-up only, and against the binary128 ground truth of a
// spatial block (tests/golden/c5_truth_binary128.npz) the error after 10 iterations was twice a CPU build's.
__device__ __forceinline__ bool sp_precise(const SpatialArgs &sa)
{
    return sa.it + 1 < sa.ka.precise_passes;
}

// eq (20) for a parameter whose prior mean waits for its neighbours (types M, m): the terms of the parameters that do
// NOT wait first (index order), then those that do (index order) - the split form's prep kernel sums the first group
// ahead of the ordered sweep (sw_pre), and every form of the first sweep must round alike.
template <int P>
__device__ __forceinline__ double theta_mean_swept_last(const double (&Sig)[P * (P + 1) / 2], const double (&rhs)[P], int i, const int32_t *types)
{
    double s = 0;
#pragma unroll
    for (int j = 0; j < P; j++)
        s = is_swept_type(types[j]) ? s : __builtin_fma(Sig[tri(i, j)], rhs[j], s);
#pragma unroll
    for (int j = 0; j < P; j++)
        s = is_swept_type(types[j]) ? __builtin_fma(Sig[tri(i, j)], rhs[j], s) : s;
    return s;
}

// update_theta (vb_lane_kernel.h) without the LM form (LMalpha = 0 in the spatial loop), the swept parameters' means in
// the order above
template <int P>
__device__ __forceinline__ bool sp_update_theta(const KernelArgs &ka, VoxelState<P> &st, const Moments<P> &mo)
{
    const double phibar = st.b * st.c;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            st.Lam[tri(i, j)] = phibar * mo.A[tri(i, j)] + ((i == j) ? st.pprec[i] : 0.0); // eq (19)
    st.precValid = true;
    st.covValid = false;
    double rhs[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        rhs[i] = theta_rhs(theta_rhs_base<P>(phibar, mo, i), st.pprec[i], st.pm[i]);
    if (!ensure_cov<P>(st))
        return false;
#pragma unroll
    for (int i = 0; i < P; i++)
        st.m[i] = is_swept_type(ka.cfg.prior_type[i]) ? theta_mean_swept_last<P>(st.Sig, rhs, i, ka.cfg.prior_type)
                                                       : theta_mean<P>(st.Sig, rhs, i); // eq (20)
    return true;
}

// The noise model seen from the FIRST sweep (inference_vb.cc:643-651). UpdateTheta needs J'XJ and J'X(y - g) only, and
// whatever the noise model is, those sit in the A / U rows of the state as "effective" moments with the factor
// E[phi] in rows B x C (vb_spatial_noise.h writes them so for several precisions and for AR(1) noise): priors, eq
// (19), (20) and the whole split sweep never look at the noise model. What does is the free energy "before" and
// "theta", whose value nobody reads but whose failure stops the voxel - this policy evaluates it.
// at_centre: the means are still the linearisation centre (k = y - g); Fprior as Vb::CalculateF adds it.
template <int P>
struct SpWhite
{
    static constexpr int EXTRA_ROWS = 0;
    static __device__ __forceinline__ bool sweep_F(const SpatialArgs &sa, int v, VoxelState<P> &st, const Moments<P> &mo,
        bool at_centre, double Fprior, double &F, bool &finite)
    {
        double kk, trSA;
        if (at_centre)
        {
            kk = mo.s;
            trSA = trace_SA<P>(st, mo);
        }
        else
        {
            bool lost;
            residual_terms<P>(st, mo, 0.0, kk, trSA, lost);
        }
        return calc_free_energy<P>(sa.ka, st, kk, trSA, Fprior, F, finite);
    }
};

// ---- setup: Vb::SetupPerVoxelDists (inference_vb.cc:207-247), one lane per voxel --------------
template <class Model, int P>
__global__ __launch_bounds__(64, lane_waves<P>()) void vb_spatial_setup_kernel(const SpatialArgs sa)
{
    const KernelArgs &ka = sa.ka;
    constexpr int PT = P * (P + 1) / 2;
    // (the table of exp_acc in LDS, see vb_math.h: every lane of the workgroup takes part, before any leaves)
    constexpr bool ACC = Model::model_id == FVB_MODEL_EXP;
    __shared__ double exp_tab[ACC ? 64 : 1];
    if (ACC)
    {
        exp_tab[threadIdx.x] = FVB_EXP_TABLE[threadIdx.x];
        __syncthreads();
    }
    const int v = blockIdx.x * 64 + threadIdx.x;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    if (v >= ka.cfg.n_voxels)
        return;
    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;
    ma.exp_table = ACC ? exp_tab : nullptr;
    VoxelState<P> st;
    Moments<P> mo;
    if (ka.cfg.init_mvn)
    {
        constexpr int n = P + 1;
        constexpr int nCov = n * (n + 1) / 2;
        const double *src = ka.cfg.init_mvn + v;
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = src[(size_t)i * V];
#pragma unroll
        for (int i = 0; i < P; i++)
            st.m[i] = src[(size_t)(nCov + i) * V];
        const double nm = src[(size_t)(nCov + P) * V];
        const double nv = src[(size_t)tri(P, P) * V];
        st.b = nv / nm;
        st.c = nm / st.b;
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
            // (through the prefetching loop of the streaming passes: one load per trip and a wait for it made this
            // scan - 100 dependent memory round trips per wave - 0.9 of the kernel's 2.9 ms at 128^3)
            double data_max = 0;
            auto step = [&](int t, double y) { data_max = (t == 0 || y > data_max) ? y : data_max; };
            FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, step)
            Model::init_posterior(ma, data_max, st.m);
        }
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            const int tr = ka.cfg.transform[i];
            st.m[i] = to_fabber(tr, st.m[i]);
            st.Sig[tri(i, i)] = to_fabber_var(tr, st.Sig[tri(i, i)]);
        }
        st.b = ka.cfg.noise_post_b[0];
        st.c = ka.cfg.noise_post_c[0];
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
    double centre[P]; // the posterior means, or the locked centres (inference_vb.cc:225-232)
#pragma unroll
    for (int i = 0; i < P; i++)
        centre[i] = sa.locked_centres ? sa.locked_centres[(size_t)i * V + v] : st.m[i];
    // (this kernel exists for the first linearisation alone: with the half-ulp exp where the model is exponential)
    const int status = recentre<Model, P, ACC>(ka, ma, v, centre, mo, true);
    sa.status[v] = status ? (status | 0x100) : 0;
    sp_store_theta<P>(sa, v, st);
    sp_store_noise<P>(sa, v, st, mo);
}

// ---- CalculateaK (priors.cc:221-344): per-block partial sums of (trace_term, term2) ------------
template <int P>
__global__ __launch_bounds__(256) void vb_spatial_ak_partial_kernel(const SpatialArgs sa)
{
    typedef SpLayout<P> L;
    const KernelArgs &ka = sa.ka;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const int dims = sa.spatial_dims;
    __shared__ double red[256];
    for (int k = 0; k < P; k++)
    {
        const int type = ka.cfg.prior_type[k];
        if (!is_spatial_type(type))
            continue;
        double trace_term = 0, term2 = 0;
        // one block per segment (a z-plane of the owned voxels, cut every 4096 voxels from the plane's first):
        // the partial sum of a segment does not depend on how the volume is cut into slabs (cuts fall between
        // planes), so a_K - which every voxel's prior depends on - is the same bits for any number of slabs
        for (int v = sa.seg_start[blockIdx.x] + threadIdx.x; v < sa.seg_start[blockIdx.x + 1]; v += 256)
        {
            if (sa.status[v] != 0) // ignored voxels (priors.cc:237-240) ...
                continue;
            const double sigmaK = sa.state[(size_t)(L::SIG + tri(k, k)) * V + v];
            const double wK = sa.state[(size_t)(L::M + k) * V + v];
            int nn = 0;
            double SwK = 0;
            for (int i = 0; i < 6; i++)
            {
                const int u = sa.nn[(size_t)v * 6 + i];
                if (u >= 0 && sa.status[u] == 0) // ... are also gone from every neighbour list (IgnoreVoxel)
                {
                    nn++;
                    SwK += wK - sa.state[(size_t)(L::M + k) * V + u];
                }
            }
            if (type == FVB_PRIOR_SPATIAL_m)
                trace_term += sigmaK * dims * 2;
            else if (type == FVB_PRIOR_SPATIAL_M)
                trace_term += sigmaK * (nn + 1e-8);
            else if (type == FVB_PRIOR_SPATIAL_p)
                trace_term += sigmaK * (4 * dims * dims + 2 * dims);
            else
                trace_term += sigmaK * (nn * nn + nn);
            if (type == FVB_PRIOR_SPATIAL_p || type == FVB_PRIOR_SPATIAL_m)
                SwK += wK * (dims * 2 - nn);
            if (type == FVB_PRIOR_SPATIAL_m || type == FVB_PRIOR_SPATIAL_M)
                term2 += SwK * wK;
            else
                term2 += SwK * SwK;
        }
        for (int which = 0; which < 2; which++)
        {
            red[threadIdx.x] = which ? term2 : trace_term;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1)
            {
                if ((int)threadIdx.x < s)
                    red[threadIdx.x] += red[threadIdx.x + s];
                __syncthreads();
            }
            if (threadIdx.x == 0)
                sa.partials[((size_t)blockIdx.x * P + k) * 2 + which] = red[0];
            __syncthreads();
        }
    }
}

// this slab's sums: the segments' partial sums added in segment order (the order a single device and any
// slab decomposition share: fabber_core_amd/spatial_mgpu.py adds the slabs' segments the same way)
template <int P>
__global__ __launch_bounds__(64) void vb_spatial_ak_reduce_kernel(const SpatialArgs sa)
{
    // lane 2 k + which adds column (k, which) of partials [n_blocks][P][2] from the first segment to the last; the
    // 64 lanes fetch the rows together, 512 doubles at a time through LDS (a lane reading its own column from
    // memory pays a memory round trip every few segments: 77 us for the 512 segments of a 128^3 volume)
    constexpr int CHUNK = 512;
    __shared__ double rows[CHUNK];
    const int k = threadIdx.x >> 1, which = threadIdx.x & 1;
    const bool mine = k < P && is_spatial_type(sa.ka.cfg.prior_type[k < P ? k : 0]);
    const int per_block = 2 * P;
    const int blocks_per_chunk = CHUNK / per_block;
    double acc = 0;
    for (int b0 = 0; b0 < sa.n_blocks; b0 += blocks_per_chunk)
    {
        const int nb = (sa.n_blocks - b0 < blocks_per_chunk) ? sa.n_blocks - b0 : blocks_per_chunk;
        for (int i = threadIdx.x; i < nb * per_block; i += 64)
            rows[i] = sa.partials[(size_t)b0 * per_block + i];
        __syncthreads();
        if (mine)
            for (int b = 0; b < nb; b++)
                acc += rows[b * per_block + 2 * k + which];
        __syncthreads();
    }
    if (k < P)
        sa.ak_sums[2 * k + which] = mine ? acc : 0.0;
}

template <int P>
__global__ void vb_spatial_ak_final_kernel(const SpatialArgs sa)
{
    const KernelArgs &ka = sa.ka;
    const int k = threadIdx.x;
    if (k >= P || !is_spatial_type(ka.cfg.prior_type[k]))
        return;
    // with several slabs ak_sums holds the all-reduced sums by now (set by the host in between)
    const double trace_term = sa.ak_sums[2 * k + 0], term2 = sa.ak_sums[2 * k + 1];
    const double gk = 1 / (0.5 * trace_term + 0.5 * term2 + 1 / sa.q1);
    const double hK = sa.n_voxels_global * 0.5 + sa.q2;
    double aK = gk * hK;
    if (aK < 1e-50)
        aK = 1e-50;
    double aKMax = aK * sa.spatial_speed;
    if (aKMax < 0.5)
        aKMax = 0.5;
    if ((sa.spatial_speed > 0) && (aK > aKMax))
        aK = aKMax;
    sa.aK[k] = aK;
}

// ---- first sweep, one level: priors + UpdateTheta (inference_vb.cc:614-672) -------------------
// There are hundreds of these launches per iteration, so the (large) argument block stays in
// device memory and only the level range travels with the launch.
template <int P, bool NEEDF, class NZ = SpWhite<P> >
__global__ __launch_bounds__(64) void vb_spatial_theta_kernel(const SpatialArgs *__restrict__ sap, int level_begin,
    int level_count, int it)
{
    typedef SpLayout<P> L;
    const SpatialArgs &sa = *sap;
    const KernelArgs &ka = sa.ka;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= level_count)
        return;
    const int v = sa.order[level_begin + i];
    const size_t V = (size_t)ka.cfg.n_voxels;
    // An ignored voxel still has its priors applied before the reference skips it
    // (inference_vb.cc:626-641); only the last voxel's are observable (its F term is reused).
    const bool ignored = sa.status[v] != 0;
    if (ignored && v != sa.owned_end - 1)
        return;
    VoxelState<P> st;
    Moments<P> mo;
    sp_load<P>(sa, v, st, mo);
    const int dims = sa.spatial_dims;
    double Fprior = 0;
    // Neighbour ids and their status do not depend on the parameter: fetched once, with every load
    // unconditional (absent neighbours read the voxel itself and are masked afterwards), so that
    // the six chains id -> status / mean run side by side instead of one after the other - this
    // kernel is launched once per level and its time is the latency of its dependent loads.
    // Vb::IgnoreVoxel (inference_vb.cc:266-297) deletes a failed voxel from the first-neighbour
    // lists of its neighbours and from the second-neighbour lists of its second neighbours; the
    // second-neighbour list itself was fixed when it was built, so it still reaches across a
    // failed intermediate voxel. With fixed tables: test the status of every END point, never of
    // the intermediate one.
    bool any_second = false;
#pragma unroll
    for (int k = 0; k < P; k++)
        any_second |= (ka.cfg.prior_type[k] == FVB_PRIOR_SPATIAL_P || ka.cfg.prior_type[k] == FVB_PRIOR_SPATIAL_p);
    int n1[6];
    bool ok1[6];
#pragma unroll
    for (int a = 0; a < 6; a++)
    {
        const int u = sa.nn[(size_t)v * 6 + a];
        n1[a] = (u < 0) ? v : u;
        ok1[a] = (u >= 0);
    }
    bool live1[6];
#pragma unroll
    for (int a = 0; a < 6; a++)
        live1[a] = ok1[a] && (sa.status[n1[a]] == 0);
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        const int type = ka.cfg.prior_type[k];
        if (is_spatial_type(type))
        {
            // SpatialPrior::ApplyToMVN (priors.cc:362-482)
            const double *mk = sa.state + (size_t)(L::M + k) * V;
            int nn = 0, nn2 = 0;
            double contrib_nn = 0, contrib_nn2 = 0;
            double m1[6];
#pragma unroll
            for (int a = 0; a < 6; a++)
                m1[a] = mk[n1[a]];
#pragma unroll
            for (int a = 0; a < 6; a++)
                if (live1[a])
                {
                    nn++;
                    contrib_nn += m1[a];
                }
            if (any_second && (type == FVB_PRIOR_SPATIAL_P || type == FVB_PRIOR_SPATIAL_p))
            {
                for (int a = 0; a < 6; a++)
                {
                    if (!ok1[a])
                        continue;
                    int n2[6];
#pragma unroll
                    for (int b = 0; b < 6; b++)
                        n2[b] = sa.nn[(size_t)n1[a] * 6 + b];
                    bool live2[6];
                    double m2[6];
#pragma unroll
                    for (int b = 0; b < 6; b++)
                    {
                        const int w = (n2[b] < 0) ? v : n2[b];
                        live2[b] = (n2[b] >= 0) && (n2[b] != v) && (sa.status[w] == 0);
                        m2[b] = mk[w];
                    }
#pragma unroll
                    for (int b = 0; b < 6; b++)
                        if (live2[b])
                        {
                            nn2++;
                            contrib_nn2 += -m2[b];
                        }
                }
            }
            if (type == FVB_PRIOR_SPATIAL_p || type == FVB_PRIOR_SPATIAL_m)
            {
                nn = 2 * dims;
                nn2 = 4 * dims * dims - nn;
            }
            const double aK = sa.aK[k];
            double spatial_prec;
            if (type == FVB_PRIOR_SPATIAL_M)
                spatial_prec = aK * (nn + 1e-8);
            else if (type == FVB_PRIOR_SPATIAL_m)
                spatial_prec = aK * nn;
            else
                spatial_prec = aK * (nn * nn + nn);
            const double prec0 = ka.cfg.prior_prec[k], mean0 = ka.cfg.prior_mean[k];
            if (type == FVB_PRIOR_SPATIAL_p || type == FVB_PRIOR_SPATIAL_m)
                st.pprec[k] = spatial_prec;
            else
                st.pprec[k] = prec0 + spatial_prec;
            double spatial_mean;
            if (type == FVB_PRIOR_SPATIAL_m || type == FVB_PRIOR_SPATIAL_M)
            {
                const double rec = 1 / double(nn);
                spatial_mean = contrib_nn * rec;
            }
            else if (nn != 0)
            {
                const double rec = second_order_rec(nn, nn2); // (integer division, priors.cc:455)
                spatial_mean = second_order_mean(contrib_nn, contrib_nn2, rec);
            }
            else
                spatial_mean = 0;
            const double pcov = 1.0 / st.pprec[k]; // (prior precisions are diagonal)
            if (type == FVB_PRIOR_SPATIAL_m || type == FVB_PRIOR_SPATIAL_M)
                st.pm[k] = pcov * spatial_prec * spatial_mean;
            else
                st.pm[k] = second_order_pm(pcov, spatial_prec, spatial_mean, prec0, mean0);
        }
        else if (type == FVB_PRIOR_ARD) // priors.cc:150-181
        {
            const double new_cov = st.m[k] * st.m[k] + st.Sig[tri(k, k)];
            // (selects, not two branches that store: merged into one store through a pointer phi, the two stores keep
            // st.pm / st.pprec from being promoted to registers)
            st.pprec[k] = 1.0 / ((it == 0) ? ka.cfg.prior_var[k] : new_cov);
            st.pm[k] = ka.cfg.prior_mean[k]; // (set in iteration 0 and never changed: apply_priors, vb_lane_kernel.h)
            const double bb = 2 / new_cov;
            Fprior += -1.5 * (log(bb) + digamma(0.5)) - 0.5 - gammaln(0.5) - 0.5 * log(bb);
        }
        else if (type == FVB_PRIOR_IMAGE)
        {
            st.pm[k] = ka.cfg.image_prior[k][v];
            st.pprec[k] = ka.cfg.prior_prec[k];
        }
        else
        {
            st.pm[k] = ka.cfg.prior_mean[k];
            st.pprec[k] = ka.cfg.prior_prec[k];
        }
    }
    if (v == sa.owned_end - 1) // (with several slabs: the last voxel of THIS slab)
        *sa.fprior_last = Fprior;
    if (ignored)
        return;
    // CalculateF "before" and "theta" (:643, :651). Their values are overwritten before anyone can
    // read them, but their failures are observable: a non-finite F (e.g. a non-finite sample) stops
    // the voxel HERE, before / straight after the update of its means, and makes it an ignored voxel
    // for every voxel that follows in the sweep.
    if (NEEDF)
    {
        double F;
        bool finite = true;
        // the means are still the linearisation centre: k = y - g, k'k = s
        const bool ok = NZ::sweep_F(sa, v, st, mo, true, Fprior, F, finite);
        if (!ok || !finite)
        {
            sa.status[v] = ok ? FVB_BAD_FREE_ENERGY : FVB_BAD_RESULT;
            return;
        }
    }
    int status = FVB_OK;
    if (!sp_update_theta<P>(ka, st, mo)) // LMalpha = 0 in the spatial loop (:649)
        status = FVB_BAD_RESULT;
    if (NEEDF && status == FVB_OK)
    {
        double F;
        bool finite = true;
        const bool ok = NZ::sweep_F(sa, v, st, mo, false, Fprior, F, finite);
        if (!ok || !finite)
            status = ok ? FVB_BAD_FREE_ENERGY : FVB_BAD_RESULT;
    }
    if (status != FVB_OK)
        sa.status[v] = status;
    sp_store_theta<P>(sa, v, st);
}

// =====================================================================================================
// The first sweep, split. Per level the per-level launches above pay one kernel boundary and one voxel's whole
// dependent chain - state, neighbour ids, neighbour means, the P x P inversion, the stores - 382 times per
// iteration at 128^3 (5.2 of 6.2 ms). But of all that only ONE number per parameter with a first-neighbour
// prior (types M, m) really waits for the neighbours: the prior mean
//        mu0_k = (1 / prec0_k) s_k mean(neighbours' current means of parameter k)       (priors.cc:441-482)
// The prior PRECISION s_k = a_K (nn + 1e-8) depends on the neighbour COUNT only, so Lambda (eq 19), its
// inverse Sigma and every entry of eq (20)'s right-hand side except mu0_k's are known for all voxels at
// once. The sweep that has to respect the reference's voxel order shrinks to
//        m_k(v) = sum_j Sigma_kj rhs_j,   rhs_k = fma(prec0_k, mu0_k, base_k)
// a dozen multiply-adds per voxel on records laid out in the order of the sweep. Types P and p, as the reference
// codes them (second_order_rec above), wait for nothing: their prior mean is pcov x prec0 mean0, complete in the
// prep kernel like an ordinary prior's. Three steps per iteration:
//   vb_spatial_prep_kernel        all voxels in parallel: priors, Lambda, Sigma, right-hand sides -> sweep records
//   vb_spatial_slab_sweep_kernel  the ordered part, ONE launch (none without a type M / m prior): a workgroup per
//                                 z-slab, the previous level's means in LDS, an inbox hand-over between slabs
//   vb_spatial_noise_kernel       (FAST) first completes the other means from the sweep's result - the same
//                                 theta_rhs / theta_mean sequence update_theta runs, so the posterior is the
//                                 per-level kernels' bit for bit - then carries on with the second sweep
// What the split form does not model raises sw_flags[0], and the driver repeats the whole run with the per-level
// launches (tests/test_spatial.py forces each): a voxel that FAILS during the first sweep (singular Lambda,
// non-finite F) changes its later neighbours' lists in the reference (Vb::IgnoreVoxel); a mean that is not
// finite next to a type P / p prior turns the prior means of its first and second neighbours into NaN
// (0 x inf) in the reference's voxel order.
// =====================================================================================================

// A voxel that drops out in prep (the run is going to be repeated): give every later neighbour's inbox this
// voxel's old mean with THIS sweep's serial number, so that no neighbour waits for it.
__device__ __forceinline__ void sweep_release(const SpatialArgs &sa, int pos, uint32_t serial, int v)
{
    const size_t NP = (size_t)sa.n_pos;
    const size_t V = (size_t)sa.ka.cfg.n_voxels;
    for (int a = 0; a < 6; a++)
    {
        const int u = sa.nn[(size_t)v * 6 + a];
        if (u < 0 || sa.pos_of[u] < pos)
            continue;
        for (int s = 0; s < sa.n_spatial; s++)
        {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(sa.state[(size_t)sa.spatial_param[s] * V + v]);
            unsigned long long *g = sa.sw_gran + ((size_t)s * NP + sa.pos_of[u]) * 2;
            const unsigned long long now = (unsigned long long)serial << 32;
            __hip_atomic_store(g, now | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, now | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// a type P / p prior's parameter must stay finite (and far from overflow: the reference sums up to 36 of them)
__device__ __forceinline__ bool second_order_safe(double m)
{
    return fabs(m) < 1e300; // (false for NaN)
}

template <int P, bool NEEDF, class NZ = SpWhite<P> >
__global__ __launch_bounds__(64) void vb_spatial_prep_kernel(const SpatialArgs *__restrict__ sap, int it, uint32_t serial)
{
    typedef SpLayout<P> L;
    const SpatialArgs &sa = *sap;
    const KernelArgs &ka = sa.ka;
    // Workgroups b, b + 8, ... share an XCD and its L2 (observed placement; speed only): give each XCD one
    // contiguous eighth of the voxel list. A voxel's record fields go to its (slab-major) position, 8 bytes per
    // field, and neighbouring positions belong to voxels a row or a plane away - written through ONE L2 the
    // pieces of a 128-byte line merge there, through eight they leave as eight partial lines.
    const int per_xcd = (gridDim.x + 7) / 8;
    const int block = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    // A wavefront takes a TILE of 8 x 8 voxels of one plane where the volume's dense map is at hand: the voxels of a
    // tile's diagonals x + y = const are neighbours in a run, so the 8-byte record stores of up to eight lanes fall
    // into one or two 32-byte sectors (64 consecutive voxels of a row are 64 different runs, one sector each:
    // 1.74 GB written for 0.9 GB of records at 128^3).
    int v;
    if (sa.n_tiles > 0)
    {
        if (block >= sa.n_tiles)
            return;
        const int per_plane = sa.tile_nx * sa.tile_ny;
        const int z = sa.tile_z0 + block / per_plane, ty = (block % per_plane) / sa.tile_nx, tx = block % sa.tile_nx;
        const int x = tx * 8 + (threadIdx.x & 7), y = ty * 8 + (threadIdx.x >> 3);
        const long long off = (long long)z * sa.xsize * sa.ysize + (long long)y * sa.xsize + x - sa.dense_base;
        if (x >= sa.xsize || y >= sa.ysize || off < 0 || off >= sa.dense_span)
            return;
        v = sa.dense[off];
        if (v < sa.owned_begin || v >= sa.owned_end)
            return;
    }
    else
    {
        v = sa.owned_begin + block * 64 + threadIdx.x;
        if (v >= sa.owned_end)
            return;
    }
    const size_t V = (size_t)ka.cfg.n_voxels;
    const size_t NP = (size_t)sa.n_pos;
    const int pos = sa.pos_of[v];
    const bool ignored = sa.status[v] != 0;
    if (ignored)
        sa.sw_alive[pos] = 0;
    if (ignored && v != sa.owned_end - 1) // (the last voxel's priors are observable through its F term)
        return;
    VoxelState<P> st;
    Moments<P> mo;
    sp_load_prep<P, NEEDF>(sa, v, st, mo);
    const int dims = sa.spatial_dims;
    // the listed neighbours BY DIRECTION (+x -x +y -y +z -z, the order the reference finds them in): slot d holds the
    // neighbour found with offset d, or the voxel itself marked dead; a sum over d = 0 .. 5 that adds +0.0 for a dead
    // slot is the sum over the list in list order, bit for bit
    int n1[6];
    bool live1[6];
    {
        const int dir = sa.nn_dir[v];
        int listed[6];
#pragma unroll
        for (int a = 0; a < 6; a++)
            listed[a] = sa.nn[(size_t)v * 6 + a];
#pragma unroll
        for (int d = 0; d < 6; d++)
        {
            int u = -1;
#pragma unroll
            for (int a = 0; a < 6; a++)
                u = (((dir >> (3 * a)) & 7) == d) ? listed[a] : u;
            n1[d] = (u < 0) ? v : u;
            live1[d] = (u >= 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++)
        live1[a] = live1[a] && (sa.status[n1[a]] == 0);
    int nn_live = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
        nn_live += live1[a] ? 1 : 0;
    double Fprior = 0;
    int si = 0; // index among the parameters the sweep updates
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        const int type = ka.cfg.prior_type[k];
        if (is_spatial_type(type)) // SpatialPrior::ApplyToMVN (priors.cc:362-482)
        {
            const bool second = (type == FVB_PRIOR_SPATIAL_P || type == FVB_PRIOR_SPATIAL_p);
            const bool dirichlet = (type == FVB_PRIOR_SPATIAL_m || type == FVB_PRIOR_SPATIAL_p);
            const int nn = dirichlet ? 2 * dims : nn_live;
            const double aK = sa.aK[k];
            const double spatial_prec = second ? aK * (nn * nn + nn) : ((type == FVB_PRIOR_SPATIAL_M) ? aK * (nn + 1e-8) : aK * nn);
            st.pprec[k] = dirichlet ? spatial_prec : ka.cfg.prior_prec[k] + spatial_prec;
            const double pcov = 1.0 / st.pprec[k];
            if (second)
            {
                // as coded the MRF mean is 0 x (a finite sum) (second_order_rec): what is left is the parameter's own prior
                st.pm[k] = pcov * (ka.cfg.prior_prec[k] * ka.cfg.prior_mean[k]);
                if (!ignored && !second_order_safe(st.m[k]))
                    sa.sw_flags[0] = 1;
                continue;
            }
            if (!ignored)
            {
                sa.sw_pprec[si * NP + pos] = st.pprec[k];
                sa.sw_q[si * NP + pos] = pcov * spatial_prec;
                // what the +x, +y, +z neighbours contribute - the sweep reaches them AFTER this voxel -: their means now
#pragma unroll
                for (int a = 0; a < 3; a++)
                    sa.sw_nbr[((size_t)si * 3 + a) * NP + pos] = live1[2 * a] ? sa.state[(size_t)(L::M + k) * V + n1[2 * a]] : 0.0;
            }
            si++;
        }
        else if (type == FVB_PRIOR_ARD) // priors.cc:150-181
        {
            const double new_cov = st.m[k] * st.m[k] + st.Sig[tri(k, k)];
            // (selects, not two branches that store: merged into one store through a pointer phi, the two stores keep
            // st.pm / st.pprec from being promoted to registers)
            st.pprec[k] = 1.0 / ((it == 0) ? ka.cfg.prior_var[k] : new_cov);
            st.pm[k] = ka.cfg.prior_mean[k]; // (set in iteration 0 and never changed: apply_priors, vb_lane_kernel.h)
            const double bb = 2 / new_cov;
            Fprior += -1.5 * (log(bb) + digamma(0.5)) - 0.5 - gammaln(0.5) - 0.5 * log(bb);
        }
        else if (type == FVB_PRIOR_IMAGE)
        {
            st.pm[k] = ka.cfg.image_prior[k][v];
            st.pprec[k] = ka.cfg.prior_prec[k];
        }
        else
        {
            st.pm[k] = ka.cfg.prior_mean[k];
            st.pprec[k] = ka.cfg.prior_prec[k];
        }
    }
    if (v == sa.owned_end - 1)
        *sa.fprior_last = Fprior;
    if (ignored)
        return;
    if (NEEDF)
    {
        // CalculateF "before" (:643) with the old posterior and the new priors. Only its FAILURE is observable
        // (a non-finite F stops the voxel before its means are updated); the one term that needs the swept
        // parameters' prior means, (m - mu0)' Lambda0 (m - mu0), is checked where they are known (noise kernel).
        VoxelState<P> tmp = st;
#pragma unroll
        for (int k = 0; k < P; k++)
            if (is_swept_type(ka.cfg.prior_type[k]))
                tmp.pm[k] = tmp.m[k];
        double F0;
        bool finite0 = true;
        if (!NZ::sweep_F(sa, v, tmp, mo, true, Fprior, F0, finite0) || !finite0)
        {
            sa.sw_flags[0] = 1;
            sa.sw_alive[pos] = 0;
            sweep_release(sa, pos, serial, v);
            return;
        }
    }
    // eq (19) and its inverse, exactly update_theta's; the means follow in the sweep and in the noise kernel
    const double phibar = st.b * st.c;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            st.Lam[tri(i, j)] = phibar * mo.A[tri(i, j)] + ((i == j) ? st.pprec[i] : 0.0);
    st.precValid = true;
    st.covValid = false;
    if (!ensure_cov<P>(st))
    {
        sa.sw_flags[0] = 1; // this voxel fails in the first sweep: not modelled here
        sa.sw_alive[pos] = 0;
        sweep_release(sa, pos, serial, v);
        return;
    }
    if (sa.n_spatial > 0) // (uniform) the records of the ordered sweep
    {
        const size_t ns = (size_t)sa.n_spatial;
        double rhs[P];
#pragma unroll
        for (int k = 0; k < P; k++)
            rhs[k] = theta_rhs(theta_rhs_base<P>(phibar, mo, k), st.pprec[k], st.pm[k]); // (complete where k is not swept)
        si = 0;
#pragma unroll
        for (int k = 0; k < P; k++)
            if (is_swept_type(ka.cfg.prior_type[k]))
            {
                double pre = 0; // theta_mean_swept_last's first group
#pragma unroll
                for (int j = 0; j < P; j++)
                    pre = is_swept_type(ka.cfg.prior_type[j]) ? pre : __builtin_fma(st.Sig[tri(k, j)], rhs[j], pre);
                sa.sw_pre[si * NP + pos] = pre;
                sa.sw_rhsk[si * NP + pos] = theta_rhs_base<P>(phibar, mo, k);
                int sj = 0;
#pragma unroll
                for (int j = 0; j < P; j++)
                    if (is_swept_type(ka.cfg.prior_type[j]))
                    {
                        sa.sw_sigk[((size_t)si * ns + sj) * NP + pos] = st.Sig[tri(k, j)];
                        sj++;
                    }
                si++;
            }
        // where the sweep finds the -x, -y, -z neighbours' NEW means (positions), and whose inbox the +z neighbour's is
        sa.sw_npos[0 * NP + pos] = live1[1] ? sa.pos_of[n1[1]] : -1;
        sa.sw_npos[1 * NP + pos] = live1[3] ? sa.pos_of[n1[3]] : -1;
        sa.sw_npos[2 * NP + pos] = live1[5] ? sa.pos_of[n1[5]] : -1;
        sa.sw_npos[3 * NP + pos] = live1[4] ? sa.pos_of[n1[4]] : -1;
    }
    sa.sw_alive[pos] = 1 + nn_live;
    // the posterior's new covariance and the priors are final here (the state's means stay the OLD ones
    // until the noise kernel has seen them: F "before" is evaluated with them)
    {
        double *p = sa.state + v;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            p[(size_t)(L::PM + i) * V] = st.pm[i];
            p[(size_t)(L::PPREC + i) * V] = st.pprec[i];
        }
#pragma unroll
        for (int i = 0; i < L::PT; i++)
            p[(size_t)(L::SIG + i) * V] = st.Sig[i];
        p[(size_t)L::LOGDET * V] = st.logdetLam;
    }
}

// ---- the ordered part, slab form ------------------------------------------------------------------------
// The data-flow sweep above hands EVERY mean over through device-scope memory: 3.8 us per level, 382 levels
// at 128^3. Here a workgroup owns a slab of sl_dz z-planes and walks the slab's levels in order; the means of
// the previous level - the x-1, y-1 and (inside the slab) z-1 neighbours - are in LDS, a level costs a
// workgroup barrier. Only the z-1 neighbours of a slab's lowest plane come from another workgroup: the slab
// below writes them into the INBOX granule of the voxel that needs them (sc1 stores, self-validating as above)
// and, since a voxel of level l needs its z-1 neighbour of level l-1, runs sl_dz - 1 levels ahead of what its
// upper neighbour needs: the hand-over is off the critical path. The later neighbours (x+1, y+1, z+1)
// contribute the means they had BEFORE the sweep; the prep kernel has put those next to the record (sw_nbr).
// 1024 lanes = G groups of sl_width lanes; group g takes the runs g, g + G, ... of the slab and requests its
// next record (and its inbox) right after finishing one: G levels of time to arrive.
template <int NS>
struct SlabRecord
{
    int alive;  // 0, or 1 + the number of live neighbours
    int np[4];  // positions of the -x, -y, -z, +z neighbours (-1 none, FVB_NP_BELOW / FVB_NP_ABOVE: on another device)
    double pre[NS], rhsk[NS], pprec[NS], q[NS];
    double sigk[NS][NS];
    double nbr[NS][3];
    unsigned long long in_lo[NS], in_hi[NS];
    __device__ __forceinline__ void load(const SpatialArgs &sa, int pos, int ns)
    {
        // every load unconditional (a lane without a voxel reads position 0 and ignores it): a load that had to
        // wait for the `alive` word first would cost the step a second memory round trip
        const size_t NP = (size_t)sa.n_pos;
        const bool have = pos >= 0;
        pos = have ? pos : 0;
        const int alive_word = sa.sw_alive[pos];
#pragma unroll
        for (int a = 0; a < 4; a++)
            np[a] = sa.sw_npos[a * NP + pos];
#pragma unroll
        for (int s = 0; s < NS; s++)
            if (s < ns)
            {
                pre[s] = sa.sw_pre[s * NP + pos];
                rhsk[s] = sa.sw_rhsk[s * NP + pos];
                pprec[s] = sa.sw_pprec[s * NP + pos];
                q[s] = sa.sw_q[s * NP + pos];
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++)
                    if (s2 < ns)
                        sigk[s][s2] = sa.sw_sigk[((size_t)s * ns + s2) * NP + pos];
#pragma unroll
                for (int a = 0; a < 3; a++)
                    nbr[s][a] = sa.sw_nbr[((size_t)s * 3 + a) * NP + pos];
                const unsigned long long *g = sa.sw_gran + ((size_t)s * NP + pos) * 2;
                if (sa.sl_remote) // (wave-uniform)
                {
                    in_lo[s] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    in_hi[s] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                else
                {
                    in_lo[s] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    in_hi[s] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        alive = have ? alive_word : 0;
    }
    // the -z neighbour lives in the slab below (another workgroup, or another device): its new mean arrives in the inbox
    __device__ __forceinline__ bool from_below(int slab_begin) const
    {
        return ((np[2] >= 0) && (np[2] < slab_begin)) || (np[2] == FVB_NP_BELOW);
    }
};

// A voxel whose z-1 neighbour lives in the slab below waits here until its inbox holds THIS sweep's value. The
// inbox was requested with the record; a workgroup's groups reach their turn several levels early, so a value that
// was not there yet is polled for while the levels before this one are still being worked on.
template <int NS>
__device__ __forceinline__ void slab_wait_inbox(const SpatialArgs &sa, SlabRecord<NS> &r, int pos, int ns, int slab_begin)
{
    if (!r.alive || !r.from_below(slab_begin))
        return;
    const size_t NP = (size_t)sa.n_pos;
    const unsigned long long serial = sa.sw_serial;
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (s < ns)
        {
            const unsigned long long *g = sa.sw_gran + ((size_t)s * NP + pos) * 2;
            int spins = 0;
#pragma nounroll
            while ((r.in_lo[s] >> 32) != serial || (r.in_hi[s] >> 32) != serial)
            {
                // never (slabs start in order and every slab only waits for the one below) - but if a hand-over is missed
                // all the same, ONE wait runs to its limit: it raises the flag every other wait looks at, so the
                // kernel drains at once and the host repeats the run with the per-level launches
                if (++spins > (1 << 20) || ((spins & 63) == 0 && __hip_atomic_load(sa.sw_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                {
                    __hip_atomic_store(sa.sw_flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (sa.sl_remote)
                {
                    r.in_lo[s] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    r.in_hi[s] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                else
                {
                    r.in_lo[s] = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    r.in_hi[s] = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
}

// one voxel of the run [begin, begin + count): slot = its index in the run. prev_* = the slab's previous run
// (its means are in lds_prev), slab_begin = the slab's first position. Straight-line code: a workgroup has
// two to eight waves at work on a level, so every instruction's latency is on the chain; the six directions are
// selected, not branched on (a direction without a live neighbour adds +0.0: x + 0.0 is x, and the sum starts from
// +0.0, so it is the sum over the listed neighbours in list order bit for bit).
// rec_tab: 1 / n for n = 0 .. 6 (LDS). This is the part of a step that the NEXT level waits for: neighbours' means ->
// this voxel's mean in LDS; slab_store writes the means and prior means to memory after the level's counter.
template <int NS>
__device__ __forceinline__ void slab_step(const SpatialArgs &sa, SlabRecord<NS> &r, int slot, int ns, int prev_begin, int prev_count,
    int slab_begin, const double *rec_tab, const double *lds_prev, double *lds_cur, int lds_stride, double (&m_out)[NS], double (&pm_out)[NS])
{
    double rhs[NS];
    bool in_prev[3];
    int off[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
    {
        const unsigned o = (unsigned)(r.np[a] - prev_begin);
        in_prev[a] = o < (unsigned)prev_count;
        off[a] = in_prev[a] ? (int)o : 0;
    }
    const bool below = r.from_below(slab_begin);
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (s < ns)
        {
            const int k = sa.spatial_param[s];
            const bool dirichlet = sa.ka.cfg.prior_type[k] == FVB_PRIOR_SPATIAL_m; // (uniform)
            const double rec = rec_tab[dirichlet ? 2 * sa.spatial_dims : r.alive - 1];
            const double val_in = __longlong_as_double((long long)((r.in_hi[s] << 32) | (r.in_lo[s] & 0xffffffffull)));
            double earlier[3];
#pragma unroll
            for (int a = 0; a < 3; a++)
                earlier[a] = in_prev[a] ? lds_prev[s * lds_stride + off[a]] : 0.0;
            earlier[2] = below ? val_in : earlier[2];
            double contrib = 0; // +x -x +y -y +z -z
#pragma unroll
            for (int a = 0; a < 3; a++)
            {
                contrib += r.nbr[s][a];
                contrib += earlier[a];
            }
            const double spatial_mean = contrib * rec;
            const double pm = r.q[s] * spatial_mean;
            pm_out[s] = pm;
            rhs[s] = theta_rhs(r.rhsk[s], r.pprec[s], pm);
        }
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (s < ns)
        {
            double m = r.pre[s];
#pragma unroll
            for (int s2 = 0; s2 < NS; s2++)
                if (s2 < ns)
                    m = __builtin_fma(r.sigk[s][s2], rhs[s2], m);
            lds_cur[s * lds_stride + slot] = m;
            m_out[s] = m;
        }
}

// ... and the part nobody in the slab waits for: the mean into the inbox of the z+1 neighbour if that lives in the
// slab above, mean and prior mean to memory for the kernels after this one
template <int NS>
__device__ __forceinline__ void slab_store(const SpatialArgs &sa, const SlabRecord<NS> &r, int pos, int ns, int slab_end,
    const double (&m)[NS], const double (&pm)[NS])
{
    const size_t NP = (size_t)sa.n_pos;
    const unsigned long long serial = sa.sw_serial;
    const int above = (r.np[3] >= slab_end) ? r.np[3] : -1;
    const int up = (r.np[3] == FVB_NP_ABOVE && sa.sw_gran_up) ? sa.sw_up_pos[pos] : -1;
#pragma unroll
    for (int s = 0; s < NS; s++)
        if (s < ns)
        {
            if (up >= 0) // the z+1 neighbour lives on the device above: its inbox there
            {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(m[s]);
                unsigned long long *g = sa.sw_gran_up + ((size_t)s * (size_t)sa.up_n_pos + up) * 2;
                __hip_atomic_store(g, (serial << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(g + 1, (serial << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (above >= 0)
            {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(m[s]);
                unsigned long long *g = sa.sw_gran + ((size_t)s * NP + above) * 2;
                __hip_atomic_store(g, (serial << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 1, (serial << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sa.sw_x[s * NP + pos] = m[s];
            sa.sw_pm[s * NP + pos] = pm[s];
        }
}

template <int NS>
__global__ __launch_bounds__(1024) void vb_spatial_slab_sweep_kernel(const SpatialArgs sa)
{
    extern __shared__ double s_mem[]; // 1 / n for n = 0 .. 7, [2][ns][sl_max_run] means of the last two runs, the slab's run table
    const int ns = sa.n_spatial;
    const int W = sa.sl_width, G = 1024 / W;
    const int g = threadIdx.x / W, lane = threadIdx.x % W; // (W is a multiple of 64: g is wave-uniform)
    const int stride = sa.sl_max_run;
    const int run0 = sa.sl_first_run[blockIdx.x], n_runs = sa.sl_first_run[blockIdx.x + 1] - run0;
    double *rec_tab = s_mem;
    double *means = s_mem + 8;
    int *tab = (int *)(means + (size_t)2 * ns * stride); // [2][n_runs]: first position, count; then one counter
    if (threadIdx.x < 8)
        rec_tab[threadIdx.x] = 1 / double(threadIdx.x); // (the prior's 1 / double(nn), priors.cc:445,450: inf for nn = 0)
    for (int i = threadIdx.x; i < n_runs; i += 1024)
    {
        tab[i] = sa.sw_level_pos[run0 + i];
        tab[n_runs + i] = sa.sw_level_count[run0 + i];
    }
    if (threadIdx.x == 0)
        tab[2 * n_runs] = 0; // `progress`, see below
    __syncthreads();
    if (n_runs == 0)
        return;
    const int slab_begin = tab[0], slab_end = tab[n_runs - 1] + tab[2 * n_runs - 1];
    // No workgroup barrier per level: every wave would have to arrive, also the ones that are busy storing the last
    // level's results and requesting their next records. A level is complete when each of the W / 64 waves of its
    // group has added 1 to `progress` (LDS) after its means are in LDS; the group of the next level polls that
    // counter - the only thing a level waits for is the level before it.
    int *progress = tab + 2 * n_runs;
    const int waves_per_group = W / 64;
    SlabRecord<NS> r;
    if (g < n_runs)
        r.load(sa, lane < tab[n_runs + g] ? tab[g] + lane : -1, ns);
    for (int li = g; li < n_runs; li += G)
    {
        const int begin = tab[li], count = tab[n_runs + li];
        const int prev_begin = li > 0 ? tab[li - 1] : 0, prev_count = li > 0 ? tab[n_runs + li - 1] : 0;
        const double *lds_prev = means + (size_t)((li + 1) & 1) * ns * stride;
        double *lds_cur = means + (size_t)(li & 1) * ns * stride;
        if (lane < count)
            slab_wait_inbox<NS>(sa, r, begin + lane, ns, slab_begin);
        // level li - 1 complete? (then level li - 2's means, which this level overwrites, have been read too)
        const int need = li * waves_per_group;
        while (__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
            __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        double m[NS], pm[NS];
        const bool work = lane < count && r.alive;
        if (work)
            slab_step<NS>(sa, r, lane, ns, prev_begin, prev_count, slab_begin, rec_tab, lds_prev, lds_cur, stride, m, pm);
        for (int i = lane + W; i < count; i += W) // (runs longer than the group is wide: 1024 lanes, one group)
        {
            SlabRecord<NS> one;
            one.load(sa, begin + i, ns);
            if (!one.alive)
                continue;
            slab_wait_inbox<NS>(sa, one, begin + i, ns, slab_begin);
            double m1[NS], pm1[NS];
            slab_step<NS>(sa, one, i, ns, prev_begin, prev_count, slab_begin, rec_tab, lds_prev, lds_cur, stride, m1, pm1);
            slab_store<NS>(sa, one, begin + i, ns, slab_end, m1, pm1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0)
            __hip_atomic_fetch_add(progress, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // while the next group works: results to memory, then this group's next record
        if (work)
            slab_store<NS>(sa, r, begin + lane, ns, slab_end, m, pm);
        const int nx = li + G;
        if (nx < n_runs)
            r.load(sa, lane < tab[n_runs + nx] ? tab[nx] + lane : -1, ns);
    }
}

// The rest of UpdateTheta after the split first sweep: the state holds this iteration's priors, Sigma and
// log|det Lambda| (prep) and still the OLD means; the sweep left the spatial parameters' new means and prior means at
// the voxel's level-major position. Returns false for a voxel that failed in prep (the run is being repeated anyway).
template <int P, bool NEEDF, class NZ>
__device__ __forceinline__ bool sp_complete_theta(const SpatialArgs &sa, int v, VoxelState<P> &st, const Moments<P> &mo)
{
    typedef SpLayout<P> L;
    const KernelArgs &ka = sa.ka;
    const size_t V = (size_t)ka.cfg.n_voxels, NP = (size_t)sa.n_pos;
    const int pos = sa.pos_of[v];
    if (!sa.sw_alive[pos])
        return false;
    const double phibar = st.b * st.c;
    double m_new[P], rhs[P];
    int si = 0;
#pragma unroll
    for (int k = 0; k < P; k++)
        if (is_swept_type(ka.cfg.prior_type[k]))
        {
            st.pm[k] = sa.sw_pm[si * NP + pos];
            m_new[k] = sa.sw_x[si * NP + pos];
            if (NEEDF) // F "before" (:643) would see this prior mean next to the old posterior mean
            {
                const double dm = st.m[k] - st.pm[k];
                if (!is_finite(dm * st.pprec[k] * dm))
                    sa.sw_flags[0] = 1;
            }
            si++;
        }
#pragma unroll
    for (int k = 0; k < P; k++)
        rhs[k] = theta_rhs(theta_rhs_base<P>(phibar, mo, k), st.pprec[k], st.pm[k]);
#pragma unroll
    for (int k = 0; k < P; k++)
        if (!is_swept_type(ka.cfg.prior_type[k]))
            m_new[k] = theta_mean<P>(st.Sig, rhs, k);
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        st.m[k] = m_new[k];
        // (types P, p: a mean that is not finite would reach the prior means of the voxels after this one)
        if ((ka.cfg.prior_type[k] == FVB_PRIOR_SPATIAL_P || ka.cfg.prior_type[k] == FVB_PRIOR_SPATIAL_p) && !second_order_safe(m_new[k]))
            sa.sw_flags[0] = 1;
    }
    if (NEEDF) // F "theta" (:651): its value is overwritten, its failure would have stopped the voxel
    {
        double F0;
        bool finite0 = true;
        st.precValid = true; // log|det Lambda| came with Sigma
        if (!NZ::sweep_F(sa, v, st, mo, false, 0.0, F0, finite0) || !finite0)
            sa.sw_flags[0] = 1;
        st.precValid = false;
    }
    double *p = sa.state + v;
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        p[(size_t)(L::M + k) * V] = st.m[k];
        p[(size_t)(L::PM + k) * V] = st.pm[k];
    }
    return true;
}

// ---- second sweep: UpdateNoise, ReCentre, F (inference_vb.cc:674-722), all voxels -------------
// ACC: the instance for the iterations whose re-centre is one of the run's first (pointwise) linearisations
// (sp_precise): with the half-ulp exp (recentre<..., ACC>). Built for the exponential model only.
template <class Model, int P, bool NEEDF, bool FAST = false, bool ACC = false>
__global__ __launch_bounds__(64, lane_waves<P>()) void vb_spatial_noise_kernel(const SpatialArgs sa)
{
    const KernelArgs &ka = sa.ka;
    __shared__ double exp_tab[ACC ? 64 : 1]; // (the table of exp_acc in LDS, see vb_math.h)
    if (ACC)
    {
        exp_tab[threadIdx.x] = FVB_EXP_TABLE[threadIdx.x];
        __syncthreads();
    }
    const int v = sa.owned_begin + blockIdx.x * 64 + threadIdx.x;
    if (v >= sa.owned_end)
        return;
    if (sa.status[v] != 0)
        return;
    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;
    ma.exp_table = ACC ? exp_tab : nullptr;
    VoxelState<P> st;
    Moments<P> mo;
    sp_load<P>(sa, v, st, mo);
    if (FAST)
    {
        if (!sp_complete_theta<P, NEEDF, SpWhite<P> >(sa, v, st, mo))
            return;
    }
    double kk, trSA;
    if (Model::host_evaluated) // the residual about the centre the moments belong to ...
    {
        ma.lin_T = ka.cfg.n_times;
        ma.lin = sa.lin_cur + (size_t)v * ka.cfg.n_times * (P + 1);
    }
    residual_and_trace<Model, P>(ka, ma, v, st, mo, kk, trSA);
    update_noise<P>(ka, st, kk, trSA);
    if (Model::host_evaluated) // ... and the re-centre about the means of this iteration's first sweep
        ma.lin = sa.lin_next + (size_t)v * ka.cfg.n_times * (P + 1);
    int status = FVB_OK;
    if (!sa.locked_linear) // inference_vb.cc:695-696
    {
        status = recentre<Model, P, ACC>(ka, ma, v, st.m, mo, ACC || sp_precise(sa));
        kk = mo.s; // the centre is the mean now: k = y - g
        trSA = trace_SA<P>(st, mo);
    }
    if (status == FVB_OK && NEEDF)
    {
        // only the last of the reference's four F evaluations per iteration is observable
        // (resultFs is overwritten each time); it uses the prior term of the LAST voxel of the
        // first sweep (the reference reuses one local variable, inference_vb.cc:612,689,702)
        st.covValid = true;
        st.precValid = false;
        double F;
        bool finite = true;
        if (!calc_free_energy<P>(ka, st, kk, trSA, *sa.fprior_last, F, finite))
            status = FVB_BAD_RESULT;
        else if (!finite)
            status = FVB_BAD_FREE_ENERGY;
        else if (ka.out.free_energy)
            ka.out.free_energy[v] = F;
    }
    if (status != FVB_OK)
        sa.status[v] = status;
    sp_store_noise<P>(sa, v, st, mo);
}

// ---- result image (inference_vb.cc:757-762) ----------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void vb_spatial_pack_kernel(const SpatialArgs sa)
{
    typedef SpLayout<P> L;
    const KernelArgs &ka = sa.ka;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= ka.cfg.n_voxels)
        return;
    const size_t V = (size_t)ka.cfg.n_voxels;
    constexpr int n = P + 1;
    constexpr int nCov = n * (n + 1) / 2;
    const double *p = sa.state + v;
    double *dst = ka.out.mvn + v;
#pragma unroll
    for (int i = 0; i < L::PT; i++)
        dst[(size_t)i * V] = p[(size_t)(L::SIG + i) * V];
#pragma unroll
    for (int j = 0; j < P; j++)
        dst[(size_t)tri(P, j) * V] = 0.0;
    const double b = p[(size_t)L::B * V], c = p[(size_t)L::C * V];
    dst[(size_t)tri(P, P) * V] = b * b * c;
#pragma unroll
    for (int i = 0; i < P; i++)
        dst[(size_t)(nCov + i) * V] = p[(size_t)(L::M + i) * V];
    dst[(size_t)(nCov + P) * V] = b * c;
    dst[(size_t)(nCov + n) * V] = 1.0;
    if (ka.out.status)
        ka.out.status[v] = sa.status[v];
    if (ka.out.iterations)
        ka.out.iterations[v] = sa.it;
    if (ka.out.free_energy && !ka.cfg.need_f)
        ka.out.free_energy[v] = 1234.5678;
}

#endif // __HIPCC__

// Kernel table for one (model, P)
typedef void (*SpatialKernelFn)(const SpatialArgs);
typedef void (*SpatialThetaFn)(const SpatialArgs *, int, int, int);
typedef void (*SpatialPrepFn)(const SpatialArgs *, int, uint32_t);
typedef void (*SpatialSweepFn)(const SpatialArgs);
struct SpatialKernels
{
    SpatialKernelFn setup, ak_partial, ak_reduce, ak_final;
    SpatialThetaFn theta;
    SpatialKernelFn noise, pack;
    int state_rows;
    const char *name;
    // the split first sweep
    SpatialPrepFn prep;
    SpatialKernelFn noise_fast;
    SpatialSweepFn slab_sweep[3]; // the ordered part, built for 1, 2 and up to P swept parameters (what a lane keeps in registers grows with it)
    int lds_classes;              // 1: setup / noise / noise_fast keep cfg.phi_index in LDS (n_times bytes of dynamic LDS)
    SpatialKernelFn noise_acc, noise_fast_acc; // the second sweep of the iterations that end in a pointwise re-centre, or NULL
};
SpatialKernels get_spatial_kernels_poly(int P, bool need_f);
SpatialKernels get_spatial_kernels_linear(int P, bool need_f);
SpatialKernels get_spatial_kernels_exp(int P, bool need_f);
SpatialKernels get_spatial_kernels_more(int model, int P, bool need_f); // the larger parameter counts of the three above
SpatialKernels get_spatial_kernels_host(int P, bool need_f); // models evaluated on the host (HostLinModel)

#if defined(__HIPCC__)
template <class Model, int P, bool FAST>
SpatialKernelFn spatial_noise_acc(bool need_f)
{
    if constexpr (Model::model_id == FVB_MODEL_EXP)
        return need_f ? (SpatialKernelFn)vb_spatial_noise_kernel<Model, P, true, FAST, true>
                      : (SpatialKernelFn)vb_spatial_noise_kernel<Model, P, false, FAST, true>;
    else
        return nullptr;
}
#endif
#define FVB_SPATIAL_CASE(MODEL, TAG, PP)                                                                     \
    case PP:                                                                                                 \
        return SpatialKernels{ vb_spatial_setup_kernel<MODEL<PP>, PP>, vb_spatial_ak_partial_kernel<PP>,     \
            vb_spatial_ak_reduce_kernel<PP>, vb_spatial_ak_final_kernel<PP>,                                 \
            need_f ? (SpatialThetaFn)vb_spatial_theta_kernel<PP, true>                                       \
                   : (SpatialThetaFn)vb_spatial_theta_kernel<PP, false>,                                     \
            need_f ? (SpatialKernelFn)vb_spatial_noise_kernel<MODEL<PP>, PP, true>                           \
                   : (SpatialKernelFn)vb_spatial_noise_kernel<MODEL<PP>, PP, false>,                         \
            vb_spatial_pack_kernel<PP>, SpLayout<PP>::ROWS, "spatial<" TAG "," #PP ">",                      \
            need_f ? (SpatialPrepFn)vb_spatial_prep_kernel<PP, true> : (SpatialPrepFn)vb_spatial_prep_kernel<PP, false>, \
            need_f ? (SpatialKernelFn)vb_spatial_noise_kernel<MODEL<PP>, PP, true, true>                     \
                   : (SpatialKernelFn)vb_spatial_noise_kernel<MODEL<PP>, PP, false, true>,                   \
            { vb_spatial_slab_sweep_kernel<1>, vb_spatial_slab_sweep_kernel<2>,                               \
                vb_spatial_slab_sweep_kernel<(PP <= 4 ? 4 : 8)> }, 0,                                        \
            spatial_noise_acc<MODEL<PP>, PP, false>(need_f), spatial_noise_acc<MODEL<PP>, PP, true>(need_f) };

} // namespace fvb
