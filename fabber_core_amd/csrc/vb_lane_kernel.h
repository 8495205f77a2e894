/*
 * vb_lane_kernel.h - voxelwise VB, one LANE per voxel, everything in registers.
 *
 * Mapping (see DESIGN.md "Kernel mapping"): a wavefront owns 64 consecutive voxels of the
 * masked-voxel list. The time series image is [t][voxel], so for a fixed t the 64 lanes read 64
 * consecutive values: every global access of the kernel is a fully coalesced 256 B (float) /
 * 512 B (double) line. There is no cross-lane traffic at all: the J'J / J'r / r'r accumulations
 * that a wave-per-voxel mapping would do with shuffles are plain register FMAs here, the P x P
 * factorisation keeps all 64 lanes busy, and LDS is not needed because the Jacobian is never
 * stored: each re-linearisation streams once over t and keeps only the moments
 *        A = J' Q J,   u = J' Q (y - g),   s = (y - g)' Q (y - g)
 * of the linearised model about the current centre. Every quantity the update equations and
 * the free energy need is a function of (A, u, s):
 *        J' X J            = phi A                              (noisemodel_white.cc:303-305)
 *        J' X (y - g + J ml) = phi (u + A ml)                   (:321)
 *        k' Q k, k = y - g + J (ml - m)  = s - 2 d'u + d'A d,  d = m - ml   (:235,:252,:416)
 *        tr(Sigma J' Q J)  = tr(Sigma A)                        (:252,:417)
 *
 * Reference path reproduced per voxel: Vb::SetupPerVoxelDists (inference_vb.cc:207-247) and the
 * body of Vb::DoCalculationsVoxelwise (inference_vb.cc:423-571).
 */
#pragma once

#include "vb_math.h"
#include "vb_models.h"

// Waves per SIMD the register allocator must leave room for (2nd argument of __launch_bounds__ =
// waves per SIMD/EU on gfx950) and depth of the data pipeline, per parameter count. Measured on
// MI355X: with P <= 2 the streaming loop fits 170 VGPRs, so three waves
// and a short pipeline win (C2: 1.42 ms vs 1.58 ms); from P = 3 the loop alone needs ~130 VGPRs,
// three waves spill ~20 GB per launch and two waves with a deeper pipeline are both faster and
// quieter on HBM (C3: 21.0 ms / 6 GB of spills vs 22.2 ms / 20 GB; C4 model 17.3 vs 18.6 ms).
// From P = 7 one wave per SIMD with the whole 512-register file: two waves spill 450 - 2100 registers each,
// one spills 10 - 140 and is 1.2 - 1.7 x faster (linear P = 7 / 8: 2.05 -> 1.76 / 4.22 -> 2.47 ms per 262144
// voxels, four exponentials 6.0 -> 3.6 ms; for P <= 4 a single wave is 8 - 14 % slower).
// FVB_LANE_WAVES_PER_SIMD / FVB_PREFETCH_DEPTH override both for experiments.
namespace fvb
{
template <int P>
constexpr int lane_waves()
{
#ifdef FVB_LANE_WAVES_PER_SIMD
    return FVB_LANE_WAVES_PER_SIMD;
#else
    return P <= 2 ? 3 : (P >= 7 ? 1 : 2);
#endif
}
template <int P>
constexpr int lane_prefetch_depth()
{
#ifdef FVB_PREFETCH_DEPTH
    return FVB_PREFETCH_DEPTH;
#else
    return P <= 2 ? 2 : 4;
#endif
}
} // namespace fvb

namespace fvb
{
struct KernelArgs
{
    fvb_config cfg; // pointer members are device pointers
    fvb_outputs out;
    const void *data;
    const void *tiles; // the series re-laid per wavefront (retile_series below), or NULL: FEED_STRIDED kernels
    double *save;      // [lane_save_rows(P)][V] scratch for save/revert, or NULL
    int32_t n_unmasked; // T - #masked timepoints
    int32_t residual_mode; // k'Qk: 0 = moments with exact fallback, 1 = always exact, 2 = moments only
    int32_t precise_passes; // the first so many linearisations of a run evaluate the model pointwise (FVB_PRECISE_PASSES)
    double residual_tol;   // mode 0: fall back when k'Qk < residual_tol * (s + 2|d'u| + |d'Ad|)
};

template <int P>
constexpr int lane_save_rows()
{
    return 3 * P + 2 * (P * (P + 1) / 2) + 3;
}

#if defined(__HIPCC__)

template <int P>
struct VoxelState
{
    static constexpr int PT = P * (P + 1) / 2;
    double m[P];
    double Lam[PT];
    double Sig[PT];
    double logdetLam; // log|det Lam| from the factorisation that produced Sig (or -log|det Sig|)
    bool precValid, covValid;
    double pm[P], pprec[P]; // theta prior: means and (diagonal) precisions
    double b, c;            // noise posterior Gamma(scale b, shape c)
};

template <int P>
struct Moments
{
    static constexpr int PT = P * (P + 1) / 2;
    double A[PT];
    double u[P];
    double s;
    double ml[P];
    bool precise; // how the sweep that produced these moments evaluated the model
};

__device__ __forceinline__ double load_data(const KernelArgs &ka, size_t idx)
{
    return ka.cfg.data_f64 ? ((const double *)ka.data)[idx] : (double)((const float *)ka.data)[idx];
}

// Software pipeline over a voxel's time series. At a million voxels the image (400 MB) does not
// stay in L2 between iterations, so every pass re-reads it from the Infinity Cache / HBM while one
// timepoint is only ~0.2 us of arithmetic: the sample for t + DEPTH is requested while t is
// processed. The loop body is instantiated DEPTH times so that slot j of the pipe is a fixed
// register in copy j (a register that a load is still writing cannot be moved), and the loop is
// instantiated per element type with the loads of its main part unconditional, so that the
// compiler's wait before using the oldest sample leaves the younger loads in flight.
template <typename RAW, int DEPTH>
struct DataPipe
{
    static constexpr int D = DEPTH;
    RAW q[D]; // as loaded (the float -> double conversion would have to wait for the load)
    const RAW *p;
    __device__ __forceinline__ void start(const KernelArgs &ka, int v, size_t V, int T)
    {
        p = (const RAW *)ka.data + v;
#pragma unroll
        for (int j = 0; j < D; j++)
            q[j] = (j < T) ? p[(size_t)j * V] : RAW(0);
    }
    // sample t (which sits in slot j = t % D); the slot is refilled with sample t + D at once
    __device__ __forceinline__ double take_and_refill(size_t V, int t, int j)
    {
        const RAW raw = q[j];
        q[j] = p[(size_t)(t + D) * V];
        return (double)raw;
    }
    __device__ __forceinline__ double take(size_t V, int T, int t, int j)
    {
        const RAW raw = q[j];
        if (t + D < T)
            q[j] = p[(size_t)(t + D) * V];
        return (double)raw;
    }
};

// for (t = 0; t < T; t++) BODY(t, y_t)
#define FVB_STREAM_TIMEPOINTS_AS(RAW, DEPTH, KA, VOX, NV, NT, BODY)                                                 \
    {                                                                                                        \
        DataPipe<RAW, DEPTH> pipe_;                                                                               \
        pipe_.start((KA), (VOX), (NV), (NT));                                                                \
        constexpr int D_ = DEPTH;                                                                           \
        const int n_main_ = ((NT) > D_) ? (((NT)-D_) / D_) * D_ : 0; /* t + D < T for every t below */       \
        for (int t0_ = 0; t0_ < n_main_; t0_ += D_)                                                          \
        {                                                                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < D_; j_++)                                                \
                BODY(t0_ + j_, pipe_.take_and_refill((NV), t0_ + j_, j_));                                   \
        }                                                                                                    \
        for (int t0_ = n_main_; t0_ < (NT); t0_ += D_)                                                       \
        {                                                                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < D_; j_++)                                                \
            {                                                                                                \
                const int t_ = t0_ + j_;                                                                     \
                if (t_ < (NT))                                                                               \
                    BODY(t_, pipe_.take((NV), (NT), t_, j_));                                                \
            }                                                                                                \
        }                                                                                                    \
    }
#define FVB_FOR_EACH_TIMEPOINT(DEPTH, KA, VOX, NV, NT, BODY)                                                 \
    if ((KA).cfg.data_f64)                                                                                   \
        FVB_STREAM_TIMEPOINTS_AS(double, DEPTH, KA, VOX, NV, NT, BODY)                                       \
    else                                                                                                     \
        FVB_STREAM_TIMEPOINTS_AS(float, DEPTH, KA, VOX, NV, NT, BODY)

// MVNDist::GetCovariance (dist_mvn.cc:232-265)
template <int P>
__device__ __forceinline__ bool ensure_cov(VoxelState<P> &st)
{
    if (st.covValid)
        return true;
    int sign;
    double logabs;
    bool ok = mvn_invert<P>(st.Lam, st.Sig, logabs, sign);
    st.logdetLam = logabs;
    st.covValid = true;
    return ok;
}
// MVNDist::GetPrecisions (dist_mvn.cc:197-230)
template <int P>
__device__ __forceinline__ bool ensure_prec(VoxelState<P> &st)
{
    if (st.precValid)
        return true;
    int sign;
    double logabs;
    bool ok = mvn_invert<P>(st.Sig, st.Lam, logabs, sign);
    st.logdetLam = -logabs;
    st.precValid = true;
    return ok;
}

// LinearizedFwdModel::ReCentre (fwdmodel_linear.cc:126-182) fused with the J'J / J'r / r'r
// accumulations of UpdateTheta / UpdateNoise / CalcFreeEnergy. J(t,i) = (f(c + d e_i)(t) -
// f(c - d e_i)(t)) / (c2_i - c3_i) exactly as the reference, except that the division is a
// multiplication by the once-computed reciprocal (<= 1 ulp per Jacobian entry).
// precise: ask the model's sweep for its most precise evaluation (vb_models.h). The kernels do
// so for the first FVB_PRECISE_PASSES = 2 linearisations of a run (measured against the binary128
// ground truth, profiles/r2_c3_truth_*.json: with only the first one pointwise the error of the
// bi-exponential fit after 2-5 iterations is 1.5 x that of a CPU build, with two it is the CPU's): parameters that start at a Fabber-space mean of
// exactly 0 (log of a rate of 1) get the reference's minimum step of 1e-10 there, f2 - f3 is then
// ~1e-10 of f, and every rounding in f shows up a million-fold in J.
// ACC: the pointwise pass uses the half-ulp exp of vb_math.h (exp_acc) for the exponential transform and inside the
// exponential model (Sweep::step_acc; exponential model only). Only kernels that exist for the pointwise passes alone are built with it -
// the set-up kernel of a spatial run and the second-sweep kernel of its first iteration (vb_spatial.h) - because
// inlined next to a streaming loop exp_acc costs that loop its registers (see exp_acc).
template <class Model, int P, bool ACC = false>
__device__ __forceinline__ int recentre(const KernelArgs &ka, const ModelArgs &ma, int v, const double (&centre)[P],
    Moments<P> &mo, bool precise = false)
{
    constexpr int PT = P * (P + 1) / 2;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double tp[P], tp2[P], tp3[P], rden[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = centre[i] * 1e-5; // fwdmodel_linear.cc:157-161
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = centre[i] + delta;
        const double c3 = centre[i] - delta;
        tp[i] = (ACC && precise) ? to_model_acc(tr, centre[i], ma.exp_table) : to_model(tr, centre[i]); // fwdmodel.cc:375-379
        tp2[i] = (ACC && precise) ? to_model_acc(tr, c2, ma.exp_table) : to_model(tr, c2);
        tp3[i] = (ACC && precise) ? to_model_acc(tr, c3, ma.exp_table) : to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        mo.ml[i] = centre[i];
    }
#pragma unroll
    for (int i = 0; i < PT; i++)
        mo.A[i] = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        mo.u[i] = 0;
    mo.s = 0;
    bool bad_offset = false, bad_jac = false;
    // (one phi here, so the index only marks masked timepoints: not read at all when there are none)
    const uint8_t *phi_index = (ka.n_unmasked == T) ? nullptr : ka.cfg.phi_index;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(precise);
    mo.precise = precise;
    // ReCentre's non-finite tests (fwdmodel_linear.cc:134-140,174-181). With every timepoint in
    // the sums they are read off the sums after the pass - g through its running total, J(.,i)
    // through A_ii = sum_t J_ti^2 (a non-finite term makes the sum non-finite; a finite J whose
    // square overflows, |J| > 1e154, would be reported too) - instead of 3 instructions per value
    // and timepoint. With masked timepoints, whose J does not enter A, they stay per timepoint.
    double g_total = 0;
    auto step = [&](int t, double y_cur) {
        double g, J[P];
        if constexpr (ACC)
        {
            if (precise) // wave-uniform
                sweep.step_acc(ma, t, tp, tp2, tp3, rden, g, J);
            else
                sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
        }
        else
            sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
        if (phi_index)
        {
#pragma unroll
            for (int i = 0; i < P; i++)
                bad_jac |= !is_finite(J[i]);
        }
        if (phi_index)
            bad_offset |= !is_finite(g);
        else
            g_total += g;
        const bool unmasked = phi_index ? (phi_index[t] != 255) : true; // wave-uniform
        if (unmasked)
        {
            const double r = y_cur - g;
#pragma unroll
            for (int i = 0; i < P; i++)
            {
#pragma unroll
                for (int j = 0; j <= i; j++)
                    mo.A[tri(i, j)] += J[i] * J[j];
                mo.u[i] += J[i] * r;
            }
            mo.s += r * r;
        }
    };
    FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, step)
    if (!phi_index)
    {
        bad_offset = !is_finite(g_total);
#pragma unroll
        for (int i = 0; i < P; i++)
            bad_jac |= !is_finite(mo.A[tri(i, i)]);
    }
    return bad_offset ? FVB_BAD_OFFSET : (bad_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}

// ---------------------------------------------------------------------------------------------------
// Tiled series. The caller's image is [t][voxel]: consecutive timepoints of a voxel lie V elements
// apart (4 MB at a million voxels - a new page and a new DRAM row for every sample). The throughput
// kernels read each series ~51 times, so it is re-laid ONCE per run (retile_series, one read and one
// write of the image: ~0.2 ms per GB) as
//        tiles [wavefront = v / 64][group = t / G][lane = v % 64][G],   G = 16 / sizeof(element)
// i.e. a wavefront's whole series is ONE contiguous block of 64 * T elements, a lane fetches G
// consecutive timepoints with one 16-byte load and a wavefront instruction moves 1 KB.
// ---------------------------------------------------------------------------------------------------
enum
{
    FEED_TILES_F32 = 0, // float series, no masked timepoints: the ABI route and the fast one
    FEED_TILES_F64 = 1, // double series (the in-memory NEWMAT::Matrix route), no masked timepoints
    FEED_STRIDED = 2    // general: either element type read in place, masked timepoints, residual mode 1
};

template <int FEED>
struct FeedTraits
{
    typedef float raw;
};
template <>
struct FeedTraits<FEED_TILES_F64>
{
    typedef double raw;
};

template <typename RAW>
struct Tile
{
    static constexpr int G = 16 / (int)sizeof(RAW); // timepoints per 16-byte group
    typedef RAW vec __attribute__((ext_vector_type(16 / sizeof(RAW))));
    static FVB_HD size_t groups(int T)
    {
        return (size_t)((T + G - 1) / G);
    }
    // elements of one wavefront's block / of the whole workspace
    static FVB_HD size_t block_elems(int T)
    {
        return groups(T) * 64 * G;
    }
    static FVB_HD size_t bytes(int V, int T)
    {
        return (size_t)((V + 63) / 64) * block_elems(T) * sizeof(RAW);
    }
};

template <typename RAW>
__global__ __launch_bounds__(256) void retile_series(const RAW *__restrict__ data, RAW *__restrict__ tiles, int V, int T)
{
    typedef Tile<RAW> TL;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V)
        return;
    const int n_groups = (int)TL::groups(T);
    typename TL::vec *dst = (typename TL::vec *)(tiles + (size_t)(v >> 6) * TL::block_elems(T)) + (v & 63);
    const RAW *src = data + v;
    for (int g = 0; g < n_groups; g++)
    {
        typename TL::vec q;
#pragma unroll
        for (int j = 0; j < TL::G; j++)
        {
            const int t = g * TL::G + j;
            q[j] = (t < T) ? src[(size_t)t * V] : RAW(0);
        }
        dst[(size_t)g * 64] = q;
    }
}

// for (t = 0; t < T; t++) body(t, y_t) over a tiled series: trips of 8 timepoints, two register sets used
// alternately, every load unconditional (the last one re-reads the final trip), the last T mod 8 one by one.
template <typename RAW, class Body>
__device__ __forceinline__ void for_each_timepoint_tiled(const RAW *lane_tile, int T, Body body)
{
    typedef Tile<RAW> TL;
    constexpr int TRIP = 8;
    constexpr int NL = TRIP / TL::G;
    const int n_trips = T / TRIP;
    if (n_trips > 0)
    {
        const typename TL::vec *p = (const typename TL::vec *)lane_tile;
        typename TL::vec qa[NL], qb[NL];
        auto fetch = [&](typename TL::vec(&q)[NL], int trip) {
#pragma unroll
            for (int l = 0; l < NL; l++)
                q[l] = p[(size_t)(trip * NL + l) * 64];
        };
        auto process = [&](const typename TL::vec(&q)[NL], int t0) {
#pragma unroll
            for (int j = 0; j < TRIP; j++)
                body(t0 + j, (double)q[j / TL::G][j % TL::G]);
        };
        fetch(qa, 0);
        int k = 0;
        for (; k + 2 <= n_trips; k += 2)
        {
            fetch(qb, k + 1);
            process(qa, k * TRIP);
            fetch(qa, (k + 2 < n_trips) ? k + 2 : n_trips - 1);
            process(qb, (k + 1) * TRIP);
        }
        if (k < n_trips)
            process(qa, k * TRIP);
    }
#pragma nounroll
    for (int t = n_trips * TRIP; t < T; t++)
        body(t, (double)lane_tile[(size_t)(t / TL::G) * 64 * TL::G + (t % TL::G)]);
}

// recentre() for a tiled series without masked timepoints: the streaming pass of the throughput kernels.
// Same arithmetic in the same order as recentre() above (the moments of voxel v are bit-identical); what
// differs is how the samples arrive and that the code exists once:
//  * the main loop handles FVB_TRIP = 8 timepoints per trip with the position inside the trip known at
//    compile time, so "exact exponentials at t = 0 mod FVB_EXP_RESYNC, one multiplication otherwise"
//    is straight-line code; the trip's samples were requested one trip earlier (two register sets, used
//    alternately; the loads are unconditional - the last one re-reads the final trip - so that the
//    compiler's wait before a set's first use leaves the other set's loads in flight);
//  * the first linearisation of a run (precise, see recentre) and the last T mod 8 timepoints go through
//    one compact loop that decides per timepoint.
template <class Model, int P, typename RAW>
__device__ __forceinline__ int recentre_tiles(const KernelArgs &ka, const ModelArgs &ma, const RAW *lane_tile,
    const double (&centre)[P], Moments<P> &mo, bool precise, double *sweep_park = nullptr)
{
    typedef Tile<RAW> TL;
    constexpr int PT = P * (P + 1) / 2;
    constexpr int TRIP = 8;
    constexpr int NL = TRIP / TL::G; // 16-byte loads per trip
    static_assert(FVB_EXP_RESYNC == 1 || FVB_EXP_RESYNC == 2 || FVB_EXP_RESYNC == 4 || FVB_EXP_RESYNC == 8,
        "the resync period must divide the trip length");
    const int T = ka.cfg.n_times;
    double tp[P], tp2[P], tp3[P], rden[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = centre[i] * 1e-5; // fwdmodel_linear.cc:157-161
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = centre[i] + delta;
        const double c3 = centre[i] - delta;
        tp[i] = to_model(tr, centre[i]); // fwdmodel.cc:375-379
        tp2[i] = to_model(tr, c2);
        tp3[i] = to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        mo.ml[i] = centre[i];
    }
    if (sweep_park) // rows [tp | tp2 | tp3 | rden] x P, entry of lane l at [row * 64 + l]: see rescue_tiles
    {
        double *q = sweep_park + (threadIdx.x & 63);
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            q[(0 * P + i) * 64] = tp[i];
            q[(1 * P + i) * 64] = tp2[i];
            q[(2 * P + i) * 64] = tp3[i];
            q[(3 * P + i) * 64] = rden[i];
        }
    }
#pragma unroll
    for (int i = 0; i < PT; i++)
        mo.A[i] = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        mo.u[i] = 0;
    mo.s = 0;
    mo.precise = precise;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(false);
    double g_total = 0; // the non-finite tests of g and J are read off the sums, as in recentre()
    auto accumulate = [&](double y_cur, double g, const double(&J)[P]) {
        g_total += g;
        const double r = y_cur - g;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
#pragma unroll
            for (int j = 0; j <= i; j++)
                mo.A[tri(i, j)] += J[i] * J[j];
            mo.u[i] += J[i] * r;
        }
        mo.s += r * r;
    };
    int t = 0;
    const int n_trips = precise ? 0 : T / TRIP;
    if (n_trips > 0)
    {
        const typename TL::vec *p = (const typename TL::vec *)lane_tile; // group g of this lane: p[g * 64]
        typename TL::vec qa[NL], qb[NL];
        auto fetch = [&](typename TL::vec(&q)[NL], int trip) {
#pragma unroll
            for (int l = 0; l < NL; l++)
                q[l] = p[(size_t)(trip * NL + l) * 64];
        };
        auto process = [&](const typename TL::vec(&q)[NL], int t0) {
#pragma unroll
            for (int j = 0; j < TRIP; j++)
            {
                double g, J[P];
                if (j % FVB_EXP_RESYNC == 0)
                    sweep.template step_fast<true>(ma, t0 + j, tp, tp2, tp3, rden, g, J);
                else
                    sweep.template step_fast<false>(ma, t0 + j, tp, tp2, tp3, rden, g, J);
                accumulate((double)q[j / TL::G][j % TL::G], g, J);
            }
        };
        fetch(qa, 0);
        int k = 0;
        for (; k + 2 <= n_trips; k += 2)
        {
            fetch(qb, k + 1);
            process(qa, k * TRIP);
            fetch(qa, (k + 2 < n_trips) ? k + 2 : n_trips - 1);
            process(qb, (k + 1) * TRIP);
        }
        if (k < n_trips)
            process(qa, k * TRIP);
        t = n_trips * TRIP;
    }
#pragma nounroll
    for (; t < T; t++)
    {
        const double y_cur = (double)lane_tile[(size_t)(t / TL::G) * 64 * TL::G + (t % TL::G)];
        double g, J[P];
        sweep.step_any(ma, t, tp, tp2, tp3, rden, g, J, precise);
        accumulate(y_cur, g, J);
    }
    bool bad_jac = false;
#pragma unroll
    for (int i = 0; i < P; i++)
        bad_jac |= !is_finite(mo.A[tri(i, i)]);
    return !is_finite(g_total) ? FVB_BAD_OFFSET : (bad_jac ? FVB_BAD_JACOBIAN : FVB_OK);
}

// Prior::ApplyToMVN for every parameter (inference_vb.cc:460-463; priors.cc:108-181). Returns
// the value of the LAST prior's free-energy term ('=' not '+=' in the reference).
template <int P, bool NEEDF>
__device__ __forceinline__ bool apply_priors(const KernelArgs &ka, int v, int it, VoxelState<P> &st, double &Fprior)
{
    bool ok = true;
    bool any_ard = false; // wave-uniform: the covariance is made valid once, not once per parameter
#pragma unroll
    for (int k = 0; k < P; k++)
        any_ard |= (ka.cfg.prior_type[k] == FVB_PRIOR_ARD);
    if (any_ard)
        ok = ensure_cov<P>(st);
#pragma unroll
    for (int k = 0; k < P; k++)
    {
        const int type = ka.cfg.prior_type[k];
        double fk = 0;
        // (the new prior goes through two locals and ONE pair of stores: stores in the branches are merged by the
        // compiler into a store through a pointer phi, which keeps st.pm / st.pprec in scratch memory)
        double pm_k, pprec_k;
        if (type == FVB_PRIOR_ARD) // priors.cc:150-181
        {
            const double post_mean = st.m[k];
            const double post_cov = st.Sig[tri(k, k)];
            const double new_cov = post_mean * post_mean + post_cov;
            pprec_k = 1.0 / ((it == 0) ? ka.cfg.prior_var[k] : new_cov);
            // priors.cc:166 sets the mean in iteration 0 and leaves it alone afterwards: it IS the configured mean in
            // every iteration (nothing else writes the prior's mean; written unconditionally because `it == 0 ? load :
            // old` becomes a load through a selected POINTER, which keeps the array in scratch memory as well)
            pm_k = ka.cfg.prior_mean[k];
            if (NEEDF)
            {
                const double bb = 2 / new_cov;
                fk = -1.5 * (log(bb) + digamma(0.5)) - 0.5 - gammaln(0.5) - 0.5 * log(bb);
            }
        }
        else if (type == FVB_PRIOR_IMAGE) // priors.cc:133-142
        {
            pm_k = ka.cfg.image_prior[k][v];
            pprec_k = ka.cfg.prior_prec[k];
        }
        else // priors.cc:108-117
        {
            pm_k = ka.cfg.prior_mean[k];
            pprec_k = ka.cfg.prior_prec[k];
        }
        st.pm[k] = pm_k;
        st.pprec[k] = pprec_k;
        Fprior = fk;
    }
    return ok;
}

// eq (20), m = Sigma (phi J'Q(y - g + J m_l) + Lambda0 mu0), with the order of its operations FIXED
// (explicit fused multiply-adds): the spatial sweep evaluates the one entry that depends on a voxel's
// neighbours in a kernel of its own (vb_spatial.h) and must land on the same bits as update_theta.
//   base_i = phi (u_i + sum_j A_ij ml_j)      rhs_i = fma(prec0_i, mu0_i, base_i)     m_i = sum_j Sigma_ij rhs_j
template <int P>
__device__ __forceinline__ double theta_rhs_base(double phibar, const Moments<P> &mo, int i)
{
    double aml = 0;
#pragma unroll
    for (int j = 0; j < P; j++)
        aml = __builtin_fma(mo.A[tri(i, j)], mo.ml[j], aml);
    return phibar * (mo.u[i] + aml);
}
__device__ __forceinline__ double theta_rhs(double base, double prior_prec, double prior_mean)
{
    return __builtin_fma(prior_prec, prior_mean, base);
}
template <int P>
__device__ __forceinline__ double theta_mean(const double *Sig, const double (&rhs)[P], int i)
{
    double s = 0;
#pragma unroll
    for (int j = 0; j < P; j++)
        s = __builtin_fma(Sig[tri(i, j)], rhs[j], s);
    return s;
}

// WhiteNoiseModel::UpdateTheta (noisemodel_white.cc:275-363), one phi
template <int P>
__device__ __forceinline__ bool update_theta(VoxelState<P> &st, const Moments<P> &mo, double alpha)
{
    constexpr int PT = P * (P + 1) / 2;
    const double phibar = st.b * st.c; // GammaDist::CalcMean, dist_gamma.cc:21-24
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++)
            st.Lam[tri(i, j)] = phibar * mo.A[tri(i, j)] + ((i == j) ? st.pprec[i] : 0.0); // eq (19)
    st.precValid = true;
    st.covValid = false;
    if (alpha <= 0.0)
    {
        double rhs[P];
#pragma unroll
        for (int i = 0; i < P; i++)
            rhs[i] = theta_rhs(theta_rhs_base<P>(phibar, mo, i), st.pprec[i], st.pm[i]);
        if (!ensure_cov<P>(st))
            return false;
#pragma unroll
        for (int i = 0; i < P; i++)
            st.m[i] = theta_mean<P>(st.Sig, rhs, i); // eq (20)
    }
    else
    {
        // Levenberg-Marquardt form, noisemodel_white.cc:330-350
        double Delta[P], Mx[PT], Mi[PT];
#pragma unroll
        for (int i = 0; i < P; i++)
            Delta[i] = phibar * mo.u[i] + st.pprec[i] * st.pm[i] - st.pprec[i] * mo.ml[i];
#pragma unroll
        for (int i = 0; i < PT; i++)
            Mx[i] = st.Lam[i];
#pragma unroll
        for (int i = 0; i < P; i++)
            Mx[tri(i, i)] += alpha * st.Lam[tri(i, i)];
        double la;
        int sg;
        if (ldl_inverse<P>(Mx, Mi, la, sg))
        {
#pragma unroll
            for (int i = 0; i < P; i++)
            {
                double s = 0;
#pragma unroll
                for (int j = 0; j < P; j++)
                    s += Mi[tri(i, j)] * Delta[j];
                st.m[i] = mo.ml[i] + s;
            }
        } // singular: warn and keep the means (:347-350)
    }
    return true;
}

// k'Qk and tr(Sigma J'QJ) from the moments
// tr(Sigma J'QJ) = tr(Sigma A)  (noisemodel_white.cc:252,417)
template <int P>
__device__ __forceinline__ double trace_SA(const VoxelState<P> &st, const Moments<P> &mo)
{
    double tr = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j < P; j++)
            tr += st.Sig[tri(i, j)] * mo.A[tri(i, j)];
    return tr;
}

// k'Qk and tr(Sigma J'QJ) from the moments. k'Qk = s - 2 d'u + d'A d is algebraically the
// reference's sum of squares, but it cancels when the linear step explains away a residual that
// is orders of magnitude larger (voxels passing through huge parameter values): `lost` is set
// when the result is below tol x the magnitude of its terms (or not positive), and the caller
// then replaces it by the directly summed value (exact_residual below).
template <int P>
__device__ __forceinline__ void residual_terms(
    const VoxelState<P> &st, const Moments<P> &mo, double tol, double &kk, double &trSA, bool &lost)
{
    double d[P];
#pragma unroll
    for (int i = 0; i < P; i++)
        d[i] = st.m[i] - mo.ml[i];
    double du = 0, dAd = 0;
    trSA = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        du += d[i] * mo.u[i];
#pragma unroll
        for (int j = 0; j < P; j++)
        {
            dAd += d[i] * mo.A[tri(i, j)] * d[j];
            trSA += st.Sig[tri(i, j)] * mo.A[tri(i, j)];
        }
    }
    kk = mo.s - 2 * du + dAd;
    const double scale = mo.s + 2 * fabs(du) + fabs(dAd);
    lost = !(kk > tol * scale) && (scale > 0); // also true for NaN / negative values
}

// The reference's k = y - g(ml) + J (ml - m) summed directly (noisemodel_white.cc:235,252):
// one more streaming pass that re-evaluates the model and its finite-difference Jacobian about
// the OLD centre ml. Only run for wavefronts in which some voxel's moment form lost precision.
template <class Model, int P>
__device__ __forceinline__ double exact_residual(
    const KernelArgs &ka, const ModelArgs &ma, int v, const Moments<P> &mo, const double (&m)[P])
{
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double tp[P], tp2[P], tp3[P], rden[P], nd[P];
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const int tr = ka.cfg.transform[i];
        double delta = mo.ml[i] * 1e-5;
        if (delta < 0)
            delta = -delta;
        if (delta < 1e-10)
            delta = 1e-10;
        const double c2 = mo.ml[i] + delta;
        const double c3 = mo.ml[i] - delta;
        tp[i] = to_model(tr, mo.ml[i]);
        tp2[i] = to_model(tr, c2);
        tp3[i] = to_model(tr, c3);
        rden[i] = 1.0 / (c2 - c3);
        nd[i] = mo.ml[i] - m[i];
    }
    double kk = 0;
    // (one phi here, so the index only marks masked timepoints: not read at all when there are none)
    const uint8_t *phi_index = (ka.n_unmasked == T) ? nullptr : ka.cfg.phi_index;
    typename Model::Sweep sweep;
    sweep.init(ma, tp, tp2, tp3);
    sweep.set_precise(mo.precise); // the Jacobian as the re-centre about ml computed it
    auto step = [&](int t, double y_cur) {
        double g, J[P];
        sweep.eval_jac(ma, t, tp, tp2, tp3, rden, g, J);
        double Jd = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            FVB_MODEL_FP
            Jd += J[i] * nd[i];
        }
        const bool unmasked = phi_index ? (phi_index[t] != 255) : true;
        if (unmasked)
        {
            const double k = y_cur - g + Jd;
            kk += k * k;
        }
    };
    FVB_FOR_EACH_TIMEPOINT(lane_prefetch_depth<P>(), ka, v, V, T, step)
    return kk;
}

// The same quantity for the FEW voxels of a wavefront that need it (about 1 % of the
// voxel-iterations of the bi-exponential fit - but two in five wavefronts contain one). Instead
// of every lane streaming over its own series, the whole wavefront works on one such voxel at a
// time: its state is broadcast, the lanes that are still in the loop share out 64 timepoints at
// a time (pointwise model evaluation), put k_t^2 into an LDS row, and the row is added up in t
// order. ~0.6 k instructions per rescued voxel against ~10 k for the streaming pass. What a voxel
// gets depends on nothing but that voxel: every k_t^2 is the same whichever lane computes it and
// the additions run over t in order, as in the streaming pass and in the reference.
template <class Model, int P>
__device__ __forceinline__ double rescue_residual(const KernelArgs &ka, const ModelArgs &ma, int v,
    const Moments<P> &mo, const double (&m)[P], bool want, double *row /* LDS, 64 doubles per wave */)
{
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const int lane = threadIdx.x & 63;
    const uint8_t *phi_index = (ka.n_unmasked == T) ? nullptr : ka.cfg.phi_index;
    const unsigned long long active = __ballot(1);
    const int n_active = __popcll(active);
    const int rank = __popcll(active & ((1ull << lane) - 1ull));
    double result = 0;
    unsigned long long todo = __ballot(want);
    while (todo) // wave-uniform
    {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int vs = __shfl(v, src);
        double tp[P], tp2[P], tp3[P], rden[P], nd[P];
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            const double ml = __shfl(mo.ml[i], src);
            const double mi = __shfl(m[i], src);
            const int tr = ka.cfg.transform[i];
            double delta = ml * 1e-5;
            if (delta < 0)
                delta = -delta;
            if (delta < 1e-10)
                delta = 1e-10;
            const double c2 = ml + delta;
            const double c3 = ml - delta;
            tp[i] = to_model(tr, ml);
            tp2[i] = to_model(tr, c2);
            tp3[i] = to_model(tr, c3);
            rden[i] = 1.0 / (c2 - c3);
            nd[i] = ml - mi;
        }
        PointwiseSweep<Model, P> sweep;
        double acc = 0;
        for (int t_base = 0; t_base < T; t_base += 64)
        {
            for (int slot = rank; slot < 64; slot += n_active)
            {
                const int t = t_base + slot;
                double kk_t = 0;
                const bool unmasked = (t < T) && (phi_index ? (phi_index[t] != 255) : true);
                if (unmasked)
                {
                    const double y = load_data(ka, (size_t)t * V + vs);
                    double g, f2[P], f3[P];
                    sweep.eval(ma, t, tp, tp2, tp3, g, f2, f3);
                    double Jd = 0;
#pragma unroll
                    for (int i = 0; i < P; i++)
                    {
                        FVB_MODEL_FP
                        Jd += ((f2[i] - f3[i]) * rden[i]) * nd[i];
                    }
                    const double k = y - g + Jd;
                    kk_t = k * k;
                }
                row[slot] = kk_t;
            }
            // one wavefront per workgroup: its LDS operations complete in order; the fences keep
            // the compiler from moving the reads above the writes (and the next writes above the reads)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int n = (T - t_base < 64) ? T - t_base : 64;
            for (int i = 0; i < n; i++)
                acc += row[i]; // masked timepoints hold 0: adding it changes nothing
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == src)
            result = acc;
    }
    return result;
}

// a value that is the same in every lane of the wavefront, moved to scalar registers
__device__ __forceinline__ double wave_uniform(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The streaming passes over t need ~130 VGPRs of their own (sweep state, perturbed parameter
// vectors, moments); the P x P posterior is not touched while they run. With three waves per
// SIMD (170 VGPRs each) the register allocator would push it to scratch - which is HBM traffic
// and a long-latency reload every iteration - so the kernel parks it in LDS itself: one wave per
// workgroup, row i of the parked state at park[i * 64 + lane] (bank-conflict free), 26 rows =
// 13 KB per wave = the LDS share of a wave at 12 waves per CU.
// What is parked, in this order while it fits in 26 rows: Sigma; Lambda where it is live across
// the pass (only the kernels that evaluate F keep it); the prior means; the prior precisions.
template <int P, bool NEEDF>
struct ParkPlan
{
    static constexpr int PT = P * (P + 1) / 2;
    // rows of 512 B that fit a wave's share of the CU's 160 KB of LDS at lane_waves<P>() waves per SIMD
    static constexpr int BUDGET = lane_waves<P>() >= 3 ? 26 : 38;
    // Beside the parked state the wave's share holds the rescue row and, if they fit, the 4 P rows of the
    // perturbed parameter vectors (SweepPark below); those come before Lambda, which only the kernels
    // with F carry across the pass and only for their save / revert copy.
    static constexpr int OTHER = 1 + ((PT + 2 * P + 1 + 4 * P <= BUDGET) ? 4 * P : 0);
    static constexpr int SIG = 0;
    static constexpr bool HAS_LAM = false; // (the voxel loop keeps no Lambda across its stages)
    static constexpr int LAM = PT;
    static constexpr int PM = PT + (HAS_LAM ? PT : 0);
    static constexpr bool HAS_PM = (PM + P + OTHER <= BUDGET);
    static constexpr int PPREC = PM + (HAS_PM ? P : 0);
    static constexpr bool HAS_PPREC = HAS_PM && (PPREC + P + OTHER <= BUDGET);
    static constexpr int ROWS = PPREC + (HAS_PPREC ? P : 0);
};

template <int P, bool NEEDF>
__device__ __forceinline__ void park_state(double *park, const VoxelState<P> &st)
{
    typedef ParkPlan<P, NEEDF> PL;
    if (!park)
        return;
#pragma unroll
    for (int i = 0; i < PL::PT; i++)
        park[(PL::SIG + i) * 64] = st.Sig[i];
    if (PL::HAS_LAM)
    {
#pragma unroll
        for (int i = 0; i < PL::PT; i++)
            park[(PL::LAM + i) * 64] = st.Lam[i];
    }
    if (PL::HAS_PM)
    {
#pragma unroll
        for (int i = 0; i < P; i++)
            park[(PL::PM + i) * 64] = st.pm[i];
    }
    if (PL::HAS_PPREC)
    {
#pragma unroll
        for (int i = 0; i < P; i++)
            park[(PL::PPREC + i) * 64] = st.pprec[i];
    }
}

template <int P, bool NEEDF>
__device__ __forceinline__ void unpark_state(const double *park, VoxelState<P> &st)
{
    typedef ParkPlan<P, NEEDF> PL;
    if (!park)
        return;
#pragma unroll
    for (int i = 0; i < PL::PT; i++)
        st.Sig[i] = park[(PL::SIG + i) * 64];
    if (PL::HAS_LAM)
    {
#pragma unroll
        for (int i = 0; i < PL::PT; i++)
            st.Lam[i] = park[(PL::LAM + i) * 64];
    }
    if (PL::HAS_PM)
    {
#pragma unroll
        for (int i = 0; i < P; i++)
            st.pm[i] = park[(PL::PM + i) * 64];
    }
    if (PL::HAS_PPREC)
    {
#pragma unroll
        for (int i = 0; i < P; i++)
            st.pprec[i] = park[(PL::PPREC + i) * 64];
    }
}

// k'Qk (moments, or exact where needed) and tr(Sigma J'QJ) for the current (m, Sigma, ml)
template <class Model, int P, bool NEEDF = true>
__device__ __forceinline__ void residual_and_trace(const KernelArgs &ka, const ModelArgs &ma, int v,
    VoxelState<P> &st, const Moments<P> &mo, double &kk, double &trSA, double *park = nullptr, double *row = nullptr)
{
    bool lost;
    residual_terms<P>(st, mo, ka.residual_tol, kk, trSA, lost);
    const int mode = ka.residual_mode; // 0 adaptive, 1 always exact, 2 never (moments only)
    const bool want = (mode == 1) || (mode == 0 && lost);
    if (mode == 0 && row)
    {
        if (__any(want)) // wave-uniform
        {
            park_state<P, NEEDF>(park, st);
            const double exact = rescue_residual<Model, P>(ka, ma, v, mo, st.m, want, row);
            unpark_state<P, NEEDF>(park, st);
            if (want)
                kk = exact;
        }
        return;
    }
    if (__any(want)) // wave-uniform: the pass is taken by the whole wavefront or not at all
    {
        park_state<P, NEEDF>(park, st);
        const double exact = exact_residual<Model, P>(ka, ma, v, mo, st.m);
        unpark_state<P, NEEDF>(park, st);
        if (want)
            kk = exact; // per-voxel decision: a voxel's result never depends on its wave-mates
    }
    else if (mode == 2)
    {
        kk = (kk > 0.0) ? kk : ((kk <= 0.0) ? 0.0 : kk); // keep b positive; NaN passes through
    }
}

// LDS rows for the perturbed parameter vectors of the current linearisation (tp, tp2, tp3 and the
// reciprocal steps): 4 P rows next to the parked state and the rescue row, if a wave's share allows.
template <int P, bool NEEDF>
struct SweepPark
{
    static constexpr int ROWS = 4 * P;
    static constexpr bool FITS = ParkPlan<P, NEEDF>::ROWS + 1 + ROWS <= ParkPlan<P, NEEDF>::BUDGET;
};

// rescue_residual for the tile-fed kernels, whose wavefronts are always complete (the kernel's lanes past
// the voxel list repeat the last voxel). For one voxel at a time - lane src - lane l evaluates the model
// pointwise at the timepoints l, l + 64, ... of THAT voxel's series (which sits in the wavefront's own tile),
// adds its k_t^2 and a butterfly over the 64 lanes adds the partial sums in a fixed order: what a voxel gets
// depends on nothing but that voxel. The model-space parameter vectors of the linearisation are read from
// the LDS rows recentre_tiles left (a broadcast read each) instead of being transformed again by every lane.
template <class Model, int P, typename RAW>
__device__ __forceinline__ double rescue_tiles(const KernelArgs &ka, const ModelArgs &ma, const RAW *wave_tile,
    const double *sweep_park, const Moments<P> &mo, const double (&m)[P], int v, bool want)
{
    constexpr int G = Tile<RAW>::G;
    const int T = ka.cfg.n_times;
    const int lane = threadIdx.x & 63;
    double result = 0;
    // (a lane past the end of the voxel list repeats the last voxel, whose series is in THAT voxel's slot of
    // the tile: it takes that lane's result below instead of asking for one of its own)
    unsigned long long todo = __ballot(want && (v & 63) == lane);
    while (todo) // wave-uniform
    {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        // (the rescued voxel's vectors are the same in every lane: told so, they live in scalar registers - or in the
        // lanes of a spill VGPR - instead of 40 VGPRs, which the kernels that also carry F do not have: there the
        // rescue spilled 22 of them to scratch memory and read them back per timepoint, 15 GB of traffic per launch)
        double tp[P], tp2[P], tp3[P], rden[P], nd[P];
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            tp[i] = wave_uniform(sweep_park[(0 * P + i) * 64 + src]);
            tp2[i] = wave_uniform(sweep_park[(1 * P + i) * 64 + src]);
            tp3[i] = wave_uniform(sweep_park[(2 * P + i) * 64 + src]);
            rden[i] = wave_uniform(sweep_park[(3 * P + i) * 64 + src]);
            nd[i] = wave_uniform(__shfl(mo.ml[i], src) - __shfl(m[i], src));
        }
        PointwiseSweep<Model, P> sweep;
        double part = 0;
        for (int t = lane; t < T; t += 64)
        {
            const double y = (double)wave_tile[(size_t)(t / G) * 64 * G + (size_t)src * G + (t % G)];
            double g, f2[P], f3[P];
            sweep.eval(ma, t, tp, tp2, tp3, g, f2, f3);
            double Jd = 0;
#pragma unroll
            for (int i = 0; i < P; i++)
            {
                FVB_MODEL_FP
                Jd += ((f2[i] - f3[i]) * rden[i]) * nd[i];
            }
            const double k = y - g + Jd;
            part += k * k;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            part += __shfl_xor(part, off);
        if (lane == src)
            result = part;
    }
    return __shfl(result, v & 63);
}

// The same for the tile-fed kernels: modes 0 (adaptive, the default) and 2 only - "always exact" is an
// experiment switch and runs on the FEED_STRIDED kernel.
template <class Model, int P, bool NEEDF, typename RAW, bool SWEEP_PARKED>
__device__ __forceinline__ void residual_and_trace_adaptive(const KernelArgs &ka, const ModelArgs &ma, int v,
    VoxelState<P> &st, const Moments<P> &mo, double &kk, double &trSA, double *park, double *row,
    const RAW *wave_tile, const double *sweep_park)
{
    bool lost;
    residual_terms<P>(st, mo, ka.residual_tol, kk, trSA, lost);
    if (ka.residual_mode == 2)
    {
        kk = (kk > 0.0) ? kk : ((kk <= 0.0) ? 0.0 : kk); // keep b positive; NaN passes through
        return;
    }
#ifndef FVB_NO_RESCUE // (experiment switch: resource usage of the kernel without the rescue path)
    if (__builtin_expect(__any(lost), 0)) // wave-uniform
    {
        park_state<P, NEEDF>(park, st);
        double exact;
        if (SWEEP_PARKED)
            exact = rescue_tiles<Model, P, RAW>(ka, ma, wave_tile, sweep_park, mo, st.m, v, lost);
        else
            exact = rescue_residual<Model, P>(ka, ma, v, mo, st.m, lost, row);
        unpark_state<P, NEEDF>(park, st);
        if (lost)
            kk = exact;
    }
#endif
}

// WhiteNoiseModel::UpdateNoise (noisemodel_white.cc:228-273), one phi
template <int P>
__device__ __forceinline__ void update_noise(const KernelArgs &ka, VoxelState<P> &st, double kk, double trSA)
{
    const double tmp = kk + trSA;
    st.b = 1 / (tmp * 0.5 + 1 / ka.cfg.noise_prior_b[0]);                   // eq (22)
    st.c = ((double)ka.n_unmasked - 1) * 0.5 + ka.cfg.noise_prior_c[0];     // eq (21)
    if (ka.cfg.locked_noise_stdev > 0)
        st.b = 1 / st.c / ka.cfg.locked_noise_stdev / ka.cfg.locked_noise_stdev;
}

// WhiteNoiseModel::CalcFreeEnergy (noisemodel_white.cc:365-454), one phi. Returns false if
// a NEWMAT-type failure happened; *finite is cleared if F is not finite.
template <int P>
__device__ __forceinline__ bool calc_free_energy(
    const KernelArgs &ka, VoxelState<P> &st, double kk, double trSA, double Fprior, double &F, bool &finite)
{
    bool ok = ensure_prec<P>(st);
    const double si = st.b, ci = st.c;
    const double siPrior = ka.cfg.noise_prior_b[0], ciPrior = ka.cfg.noise_prior_c[0];
    const double nq = (double)ka.n_unmasked;
    const double expectedLogThetaDist = 0.5 * st.logdetLam - 0.5 * P * (LOG_2PI + 1);
    const double dg = digamma(ci) + log(si);
    const double expectedLogPhiDist = -gammaln(ci) - ci * log(si) - ci + (ci - 1) * dg;
    double parts = dg * (nq * 0.5 + ciPrior - 1);                              // [0]
    parts += -gammaln(ciPrior) - ciPrior * log(siPrior) - si * ci / siPrior;  // [9]
    parts += -0.5 * si * ci * kk - 0.5 * trSA;                                // [2] (trace unscaled)
    double logdetPrior = 0, quad = 0, trSL0 = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        logdetPrior += log(fabs(st.pprec[i]));
        const double dm = st.m[i] - st.pm[i];
        quad += dm * st.pprec[i] * dm;
        trSL0 += st.Sig[tri(i, i)] * st.pprec[i];
    }
    parts += 0.5 * logdetPrior - 0.5 * nq * LOG_2PI - 0.5 * P * LOG_2PI; // [3]
    parts += -0.5 * quad;                                                  // [4]
    parts += -0.5 * trSL0;                                                 // [5]
    F = -expectedLogThetaDist - expectedLogPhiDist + parts;
    finite = is_finite(F);
    F += Fprior; // Vb::CalculateF, inference_vb.cc:310
    return ok;
}

// The voxel loop's copy for save / revert (inference_vb.cc:432-434,451-458,516-525): posterior means and
// covariance with log|det Lambda| (Lambda itself is a temporary of UpdateTheta here), prior, noise.
// Same row count as save_state (the rows of Lambda stay unused).
template <int P>
__device__ __forceinline__ void save_posterior(const KernelArgs &ka, int v, const VoxelState<P> &st, bool logdet_valid)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double *p = ka.save + v;
    // (the rows' addresses are formed here, where they are used: left to itself the compiler forms all of them
    // once before the voxel loop - 27 pointers in 54 registers for P = 4, spilled and fetched back one by one)
    asm volatile("" : "+v"(p));
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.m[i];
#pragma unroll
    for (int i = 0; i < PT; i++)
        p[(size_t)(r++) * V] = st.Sig[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pm[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pprec[i];
    p[(size_t)(r++) * V] = st.b;
    p[(size_t)(r++) * V] = st.c;
    p[(size_t)(r++) * V] = st.logdetLam;
    p[(size_t)(r++) * V] = logdet_valid ? 1.0 : 0.0;
}

template <int P>
__device__ __forceinline__ void restore_posterior(const KernelArgs &ka, int v, VoxelState<P> &st, bool &logdet_valid)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const double *p = ka.save + v;
    asm volatile("" : "+v"(p)); // (see save_posterior)
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        st.m[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < PT; i++)
        st.Sig[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pm[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pprec[i] = p[(size_t)(r++) * V];
    st.b = p[(size_t)(r++) * V];
    st.c = p[(size_t)(r++) * V];
    st.logdetLam = p[(size_t)(r++) * V];
    logdet_valid = p[(size_t)(r++) * V] != 0.0;
    st.covValid = true;
    st.precValid = false;
}

// The free energy in the voxel loop. Vb::DoCalculationsVoxelwise calls CalculateF four times per iteration
// ("before", "theta", "phi", "lin": inference_vb.cc:468,477,485,495), but only the "lin" value is ever READ - by the
// convergence detector, the history and the result (:496-500,552). The other three are observable in two ways only:
// a non-finite value throws and stops the voxel there (noisemodel_white.cc:445-451), and when a later step of the
// same iteration throws, the local F - the last value that was computed - is what the voxel reports (:552,567).
// So the kernel evaluates F IN FULL once per iteration, at "lin". At the other three sites it forms the part without
// a logarithm in it,
//     partial = -(1/2 log|Lambda| - P/2 (log 2 pi + 1)) - 1/2 (m - mu0)' Lambda0 (m - mu0) - 1/2 tr(Sigma Lambda0)
//               - 1/2 b c k'Qk - 1/2 tr(Sigma J'QJ) - (n + P)/2 log 2 pi,
// checks that it and the arguments of the remaining terms are finite (b, c > 0 and finite, prior precisions finite
// and non-zero: exactly when lgamma(c), digamma(c), log b, log|Lambda0| are), and keeps it as `pending`; a voxel that
// fails later in the iteration completes the pending site's F after the loop (free_energy_rest: the cold path).
// The terms are added in a different order than noisemodel_white.cc:365-454 does (a few ulp of the largest term).
template <int P>
struct FreeEnergyConsts
{
    // wave-uniform, held in scalar registers: the shape of the noise posterior after ANY noise update,
    // c = (n - 1)/2 + c0 (eq 21), with its lgamma and digamma (fp64 lgamma is seven divisions, digamma a recurrence:
    // ~700 instructions), the constant -lgamma(c0) - c0 log b0, and - without ARD priors, whose precisions follow
    // the posterior - the prior's log-determinant, which Prior::ApplyToMVN then sets to the same value every time
    double c_post, lgamma_post, digamma_post, prior_const, prior_logdet;
    bool prior_is_const;
    static __device__ __forceinline__ double uniform(double x)
    {
        return wave_uniform(x);
    }
    __device__ __forceinline__ void init(const KernelArgs &ka)
    {
        const double b0 = ka.cfg.noise_prior_b[0], c0 = ka.cfg.noise_prior_c[0];
        c_post = ((double)ka.n_unmasked - 1) * 0.5 + c0; // update_noise's expression
        lgamma_post = uniform(gammaln(c_post));
        digamma_post = uniform(digamma(c_post));
        prior_const = uniform(-gammaln(c0) - c0 * log(b0));
        prior_is_const = true;
        double mant = 1.0;
        int expo = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
        {
            prior_is_const = prior_is_const && (ka.cfg.prior_type[i] != FVB_PRIOR_ARD);
            int e;
            mant *= frexp(fabs(ka.cfg.prior_prec[i]), &e);
            expo += e;
        }
        prior_logdet = uniform(0.5 * (log(mant) + expo * 0.6931471805599453));
    }
};

// 1/2 log|Lambda0| for a diagonal prior: one logarithm for the product of the precisions (mantissas multiplied,
// exponents added - see ldl_inverse)
template <int P>
__device__ __forceinline__ double prior_log_determinant(const VoxelState<P> &st)
{
    double mant = 1.0;
    int expo = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        int e;
        mant *= frexp(fabs(st.pprec[i]), &e);
        expo += e;
    }
    return 0.5 * (log(mant) + expo * 0.6931471805599453);
}

// the part of F without a logarithm (see above); false: the posterior precision could not be formed
template <int P>
__device__ __forceinline__ bool free_energy_partial(const KernelArgs &ka, VoxelState<P> &st, double kk, double trSA, bool &logdet_valid,
    double &theta_terms, double &data_terms)
{
    // log|det Lambda| comes with the covariance from UpdateTheta's inversion; only a posterior that arrived
    // as a covariance (the initial one) has to be inverted for it, as MVNDist::GetPrecisions would
    bool ok = true;
    if (!logdet_valid)
    {
        st.precValid = false;
        ok = ensure_prec<P>(st);
        logdet_valid = true;
    }
    const double nq = (double)ka.n_unmasked;
    double quad = 0, trSL0 = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
    {
        const double dm = st.m[i] - st.pm[i];
        quad += dm * st.pprec[i] * dm;
        trSL0 += st.Sig[tri(i, i)] * st.pprec[i];
    }
    theta_terms = -(0.5 * st.logdetLam - 0.5 * P * (LOG_2PI + 1)) - 0.5 * quad - 0.5 * trSL0;
    data_terms = -0.5 * st.b * st.c * kk - 0.5 * trSA - 0.5 * nq * LOG_2PI - 0.5 * P * LOG_2PI; // (trace unscaled, :416-417)
    return ok;
}

// the terms of F that depend on the noise posterior (b, c) alone; lgamma_c / digamma_c = those of c
__device__ __forceinline__ double free_energy_noise_terms(const KernelArgs &ka, double b, double c, double lgamma_c, double digamma_c,
    double prior_const)
{
    const double nq = (double)ka.n_unmasked;
    const double log_b = log(b);
    const double dg = digamma_c + log_b;
    const double expectedLogPhiDist = -lgamma_c - c * log_b - c + (c - 1) * dg;
    return -expectedLogPhiDist + dg * (nq * 0.5 + ka.cfg.noise_prior_c[0] - 1) + prior_const - b * c / ka.cfg.noise_prior_b[0];
}

// F of a site in full: hot at "lin" (the noise shape is the uniform c_post after any noise update, the prior's
// log-determinant a constant without ARD: ONE logarithm), general otherwise (a posterior restored by a revert, the
// pending site of a voxel that failed)
template <int P>
__device__ __forceinline__ double free_energy_full(const KernelArgs &ka, const FreeEnergyConsts<P> &fk, const VoxelState<P> &st, double b, double c,
    bool priors_applied, double theta_terms, double data_terms, double Fprior, bool &finite)
{
    double lg, dg;
    if (c == fk.c_post)
    {
        lg = fk.lgamma_post;
        dg = fk.digamma_post;
    }
    else
    {
        lg = gammaln(c);
        dg = digamma(c);
    }
    const double noise_terms = free_energy_noise_terms(ka, b, c, lg, dg, fk.prior_const);
    const double prior_logdet = (fk.prior_is_const && priors_applied) ? fk.prior_logdet : prior_log_determinant<P>(st);
    double F = theta_terms + noise_terms + data_terms + prior_logdet;
    finite = is_finite(F);
    return F + Fprior; // Vb::CalculateF, inference_vb.cc:310
}

// a site whose F nobody reads: would the reference's CalculateF have thrown? (see above)
template <int P>
__device__ __forceinline__ bool free_energy_would_be_finite(const VoxelState<P> &st, double theta_terms, double data_terms)
{
    bool ok = is_finite(theta_terms + data_terms) && st.b > 0 && st.c > 0 && is_finite(st.b * st.c) && is_finite(st.c);
#pragma unroll
    for (int i = 0; i < P; i++)
        ok = ok && is_finite(st.pprec[i]) && st.pprec[i] != 0;
    return ok;
}

template <int P>
__device__ __forceinline__ void save_state(const KernelArgs &ka, int v, const VoxelState<P> &st)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    double *p = ka.save + v;
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.m[i];
#pragma unroll
    for (int i = 0; i < PT; i++)
        p[(size_t)(r++) * V] = st.Lam[i];
#pragma unroll
    for (int i = 0; i < PT; i++)
        p[(size_t)(r++) * V] = st.Sig[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pm[i];
#pragma unroll
    for (int i = 0; i < P; i++)
        p[(size_t)(r++) * V] = st.pprec[i];
    p[(size_t)(r++) * V] = st.b;
    p[(size_t)(r++) * V] = st.c;
    p[(size_t)(r++) * V] = (st.precValid ? 1.0 : 0.0) + (st.covValid ? 2.0 : 0.0) + 4.0 * 0;
    // logdetLam is recomputed on demand after a restore (validity flags decide)
}

template <int P>
__device__ __forceinline__ void restore_state(const KernelArgs &ka, int v, VoxelState<P> &st)
{
    constexpr int PT = P * (P + 1) / 2;
    const size_t V = (size_t)ka.cfg.n_voxels;
    const double *p = ka.save + v;
    int r = 0;
#pragma unroll
    for (int i = 0; i < P; i++)
        st.m[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < PT; i++)
        st.Lam[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < PT; i++)
        st.Sig[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pm[i] = p[(size_t)(r++) * V];
#pragma unroll
    for (int i = 0; i < P; i++)
        st.pprec[i] = p[(size_t)(r++) * V];
    st.b = p[(size_t)(r++) * V];
    st.c = p[(size_t)(r++) * V];
    const int flags = (int)p[(size_t)(r++) * V];
    // Whichever representation was valid is kept; the other is re-derived (and with it
    // log|det Lam|) the next time it is asked for, as MVNDist does.
    if (flags & 1)
    {
        st.precValid = true;
        st.covValid = false;
    }
    else
    {
        st.precValid = false;
        st.covValid = true;
    }
}

// The voxel loop. FEED says how the series reaches the streaming pass (see the enum above): the two
// tile feeds share one compact pass (recentre_tiles), FEED_STRIDED is the general one (recentre).
// The loop of Vb::DoCalculationsVoxelwise is rotated so that the re-linearisation - the only part that
// streams over the series, i.e. nearly all of the kernel's code and time - exists ONCE: the reference's
//     ReCentre; do { priors, F, theta, F, noise, F, ReCentre, F, ++it } while (!Test(F)); [revert: ReCentre, F]
// runs as
//     for (;;) { ReCentre; first time: skip | after a revert: F, done | else: F, ++it, Test(F) -> done or
//                revert; priors, F, theta, F, noise, F }
// which executes the same steps in the same order.
// WATCH: a convergence detector that watches F may run (F-change, F-reduction, trial mode, LM: convergence.cc:73-378),
// with the save / revert copy of the posterior they ask for (inference_vb.cc:432-434,451-458,506-525). Without it - the
// reference's default `maxits` detector, with or without F - the detector is a counter and none of that code, nor the
// registers it holds across the streaming pass, exists in the kernel.
template <class Model, int P, bool NEEDF, int FEED, bool WATCH = NEEDF>
__global__ __launch_bounds__(64, lane_waves<P>()) void vb_lane_kernel(const KernelArgs ka)
{
    constexpr int PT = P * (P + 1) / 2;
    typedef typename FeedTraits<FEED>::raw RAW;
    // The lanes past the end of the voxel list (last wavefront only) do not leave: they repeat the last
    // voxel - same loads, same arithmetic, same stores of the same values - so that every wavefront is
    // complete wherever its lanes work together (the rescue of k'k below).
    const int v_lane = blockIdx.x * 64 + threadIdx.x;
    const int v = (v_lane < ka.cfg.n_voxels) ? v_lane : ka.cfg.n_voxels - 1;
    const int T = ka.cfg.n_times;
    const size_t V = (size_t)ka.cfg.n_voxels;

    ModelArgs ma;
    ma.iopt0 = ka.cfg.model_iopt[0];
    ma.dopt0 = ka.cfg.model_dopt[0];
    ma.design = ka.cfg.design;
    // this wavefront's block of the tiled series, and this lane's slot in group 0 of it
    const RAW *wave_tile = (FEED == FEED_STRIDED) ? nullptr : (const RAW *)ka.tiles + (size_t)blockIdx.x * Tile<RAW>::block_elems(T);
    const RAW *lane_tile = (FEED == FEED_STRIDED) ? nullptr : wave_tile + (size_t)(v & 63) * Tile<RAW>::G;

    VoxelState<P> st;
    Moments<P> mo;
    int status = FVB_OK;
    __shared__ double park_lds[ParkPlan<P, NEEDF>::ROWS * 64];
    __shared__ double rescue_row[64];
    // the perturbed parameter vectors of the current linearisation, for the rescue (tile feeds)
    constexpr bool SWEEP_PARKED = (FEED != FEED_STRIDED) && SweepPark<P, NEEDF>::FITS;
    __shared__ double sweep_lds[SWEEP_PARKED ? SweepPark<P, NEEDF>::ROWS * 64 : 1];
    double *park = park_lds + threadIdx.x;
    double *sweep_park = SWEEP_PARKED ? sweep_lds : nullptr;

    // ---- Vb::SetupPerVoxelDists, per-voxel part (inference_vb.cc:207-247) ----
    if (ka.cfg.init_mvn)
    {
        // MVNDist::Load + GetSubmatrix + WhiteParams::InputFromMVN
        // (dist_mvn.cc:347-374,136-166; noisemodel_white.cc:70-79)
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
        st.b = nv / nm; // GammaDist::SetMeanVariance, dist_gamma.cc:29-33
        st.c = nm / st.b;
    }
    else
    {
        // FwdModel::GetInitialPosterior (fwdmodel.cc:284-313)
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
            double data_max;
            if (FEED == FEED_STRIDED)
            {
                data_max = load_data(ka, v);
                for (int t = 1; t < T; t++)
                {
                    const double y = load_data(ka, (size_t)t * V + v);
                    data_max = (y > data_max) ? y : data_max;
                }
            }
            else
            {
                constexpr int G = Tile<RAW>::G;
                data_max = (double)lane_tile[0];
#pragma nounroll
                for (int t = 1; t < T; t++)
                {
                    const double y = (double)lane_tile[(size_t)(t / G) * 64 * G + (t % G)];
                    data_max = (y > data_max) ? y : data_max;
                }
            }
            Model::init_posterior(ma, data_max, st.m);
        }
#pragma unroll
        for (int i = 0; i < P; i++) // FwdModel::ToFabber, fwdmodel.cc:315-324
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
    for (int i = 0; i < P; i++) // fwd_prior = MVNDist(P): zero mean, identity (inference_vb.cc:159)
    {
        st.pm[i] = 0;
        st.pprec[i] = 1;
    }

    double F = 1234.5678; // inference_vb.cc:438
    double Fprior = 0;
    int it = 0;
    int hist_len = 0;
    bool setup_failed = false;

    ConvState conv;
    conv_init(conv, ka.cfg.convergence, ka.cfg.max_iterations, ka.cfg.max_trials, ka.cfg.min_fchange);
    conv_reset(conv);
    // Only the detectors that watch F save and revert, and they need F: the kernels built
    // without it carry no save / revert code at all.
    const bool use_save = WATCH && (ka.save != nullptr);
    enum
    {
        FIRST,     // inference_vb.cc:235 and :443 re-centre about the same means: one pass gives both
        ITERATING, // :490
        REVERTED   // :521
    };
    int phase = FIRST;
    int n_lin = 0; // linearisations done so far
    bool logdet_valid = false; // st.logdetLam belongs to st.Sig (see free_energy_partial)

    // The iteration as four stages that each end with (at most) ONE evaluation of the free energy, so that
    // CalculateF (inference_vb.cc:302-318) - lgamma, digamma, logarithms, a few hundred instructions - is in
    // the kernel once instead of at its four call sites :468, :477, :485, :495 (+ :524 after a revert):
    //   LINEARISE  ReCentre about the means                F(k = y - g)      "lin"   [skipped the first time]
    //   PRIORS     save if asked, Prior::ApplyToMVN        F(k = y - g)      "before"
    //   THETA      UpdateTheta, k'Qk and tr(Sigma J'QJ)    F(k, trace)       "theta"
    //   NOISE      UpdateNoise                             F(k, trace)       "phi"
    // A failure ends the voxel's loop exactly where the reference's exception would (F keeps its value).
    enum
    {
        LINEARISE,
        PRIORS,
        THETA,
        NOISE
    };
    int stage = LINEARISE;
    double kk = 0, trSA = 0;    // of the THETA stage, used again by NOISE
    FreeEnergyConsts<P> fk;
    if (NEEDF)
        fk.init(ka);
    // the last site of this iteration whose F was checked but not formed (see FreeEnergyConsts): its logarithm-free
    // part, and - set where a voxel fails, so that nothing more is carried through the loop - the noise posterior
    // that site saw
    bool pending = false, priors_applied = false;
    double pending_part = 0, pending_b = 0, pending_c = 0;
#define FVB_VOXEL_FAILS(CODE, B, C)                                                                          \
    {                                                                                                        \
        status = (CODE);                                                                                     \
        pending_b = (B);                                                                                     \
        pending_c = (C);                                                                                     \
        break;                                                                                               \
    }
    for (;;)
    {
        bool want_f = NEEDF;
        double f_kk, f_tr;
        double b_before = 0, c_before = 0; // NOISE: the noise posterior the pending "theta" site saw
        if (stage == LINEARISE)
        {
            park_state<P, NEEDF>(park, st);
            if (FEED == FEED_STRIDED)
                status = recentre<Model, P>(ka, ma, v, st.m, mo, n_lin < ka.precise_passes);
            else
                status = recentre_tiles<Model, P, RAW>(ka, ma, lane_tile, st.m, mo, n_lin < ka.precise_passes, sweep_park);
            n_lin = (phase == REVERTED) ? n_lin : n_lin + 1;
            unpark_state<P, NEEDF>(park, st);
            if (status != FVB_OK)
            {
                setup_failed = (phase == FIRST);
                FVB_VOXEL_FAILS(status, st.b, st.c)
            }
            if (phase == FIRST)
            {
                if (use_save)
                    save_posterior<P>(ka, v, st, logdet_valid); // :432-434
                want_f = false;
            }
        }
        else if (stage == PRIORS)
        {
            if (use_save && conv_need_save(conv)) // :451-458
                save_posterior<P>(ka, v, st, logdet_valid);
            if (!apply_priors<P, NEEDF>(ka, v, it, st, Fprior))
                FVB_VOXEL_FAILS(FVB_BAD_RESULT, st.b, st.c)
            priors_applied = true;
        }
        else if (stage == THETA)
        {
            if (!update_theta<P>(st, mo, WATCH ? conv_lm_alpha(conv) : 0.0) || !ensure_cov<P>(st)) // :470
                FVB_VOXEL_FAILS(FVB_BAD_RESULT, st.b, st.c)
            st.precValid = false; // Lambda is not kept: the covariance and log|det Lambda| are
            logdet_valid = true;
            if (FEED == FEED_STRIDED)
                residual_and_trace<Model, P, NEEDF>(ka, ma, v, st, mo, kk, trSA, park, rescue_row);
            else
                residual_and_trace_adaptive<Model, P, NEEDF, RAW, SWEEP_PARKED>(ka, ma, v, st, mo, kk, trSA, park, rescue_row, wave_tile, sweep_park);
        }
        else
        {
            b_before = st.b;
            c_before = st.c;
            update_noise<P>(ka, st, kk, trSA); // :479
        }
        if (want_f)
        {
            if (stage == LINEARISE || stage == PRIORS) // the centre is the current mean, so k = y - g
            {
                if (!ensure_cov<P>(st))
                    FVB_VOXEL_FAILS(FVB_BAD_RESULT, st.b, st.c)
                f_kk = mo.s;
                f_tr = trace_SA<P>(st, mo);
            }
            else
            {
                f_kk = kk;
                f_tr = trSA;
            }
            double theta_terms, data_terms;
            if (!free_energy_partial<P>(ka, st, f_kk, f_tr, logdet_valid, theta_terms, data_terms))
                FVB_VOXEL_FAILS(FVB_BAD_RESULT, st.b, st.c)
            if (stage == LINEARISE) // "lin" / "revert": the F that is read
            {
                bool fin = true;
                const double Fn = free_energy_full<P>(ka, fk, st, st.b, st.c, priors_applied && phase != REVERTED, theta_terms, data_terms, Fprior, fin);
                if (!fin)
                    FVB_VOXEL_FAILS(FVB_BAD_FREE_ENERGY, st.b, st.c)
                F = Fn;
                pending = false;
            }
            else
            {
                if (!free_energy_would_be_finite<P>(st, theta_terms, data_terms))
                {
                    // (the pending site of a failing "phi" is "theta", before this stage's noise update)
                    const bool noise_stage = (stage == NOISE);
                    FVB_VOXEL_FAILS(FVB_BAD_FREE_ENERGY, noise_stage ? b_before : st.b, noise_stage ? c_before : st.c)
                }
                pending_part = theta_terms + data_terms;
                pending = true;
            }
        }
        if (stage == LINEARISE)
        {
            if (phase == REVERTED) // :516-525 done
                break;
            if (phase == ITERATING)
            {
                if (ka.out.f_history && hist_len < ka.cfg.f_history_rows) // :496-497
                    ka.out.f_history[(size_t)hist_len * V + v] = F;
                hist_len++;
                ++it;
                if (WATCH ? conv_test(conv, F) : conv_test_counting(conv)) // :500
                {
                    if (use_save && conv_need_save(conv)) // :506-513
                        save_posterior<P>(ka, v, st, logdet_valid);
                    if (use_save && conv_need_revert(conv)) // :516-525
                    {
                        restore_posterior<P>(ka, v, st, logdet_valid);
                        phase = REVERTED;
                        continue; // LINEARISE again, about the restored means
                    }
                    break;
                }
            }
            phase = ITERATING;
        }
        stage = (stage + 1) & 3;
    }
#undef FVB_VOXEL_FAILS

    if (NEEDF && pending && status != FVB_OK)
    {
        // the voxel stopped after a site whose F was only checked: that F is what it reports (inference_vb.cc:552,567)
        bool fin;
        F = free_energy_full<P>(ka, fk, st, pending_b, pending_c, true, pending_part, 0.0, Fprior, fin);
    }
    // ---- result MVN: MVNDist(fwd_post, noise.OutputAsMVN()) packed as MVNDist::Save does
    // (inference_vb.cc:549-550; dist_mvn.cc:57-100,410-429; noisemodel_white.cc:55-68) ----
    if (!ensure_cov<P>(st))
    {
        // GetCovariance() threw for every element: the concatenating constructor stores zeros
#pragma unroll
        for (int i = 0; i < PT; i++)
            st.Sig[i] = 0;
        if (status == FVB_OK)
            status = FVB_BAD_RESULT;
    }
    {
        double *dst = ka.out.mvn + v;
        constexpr int n = P + 1;
        constexpr int nCov = n * (n + 1) / 2;
#pragma unroll
        for (int i = 0; i < PT; i++)
            dst[(size_t)i * V] = st.Sig[i];
#pragma unroll
        for (int j = 0; j < P; j++)
            dst[(size_t)tri(P, j) * V] = 0.0;
        dst[(size_t)tri(P, P) * V] = st.b * st.b * st.c; // GammaDist::CalcVariance
#pragma unroll
        for (int i = 0; i < P; i++)
            dst[(size_t)(nCov + i) * V] = st.m[i];
        dst[(size_t)(nCov + P) * V] = st.b * st.c; // GammaDist::CalcMean
        dst[(size_t)(nCov + n) * V] = 1.0;
    }
    if (ka.out.f_history && hist_len < ka.cfg.f_history_rows) // :553-554
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
