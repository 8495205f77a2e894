/*
 * vb_spatial_api.hip - host driver of spatial VB: neighbour lists (Vb::CalcNeighbours,
 * inference_vb.cc:830-964), level ordering, the iteration loop of Vb::DoCalculationsSpatial
 * (inference_vb.cc:605-725) as a sequence of launches on one stream. See vb_spatial.h.
 */
#include "vb_spatial_noise.h"
#include "vb_host_copy.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <cmath>
#include <string>
#include <thread>
#include <vector>

using namespace fvb;

extern "C" const char *fabber_vb_last_error(void);
namespace fvb
{
int api_fail(int code, const std::string &msg); // vb_api.hip
int api_validate(const fvb_config *cfg, bool allow_spatial);
int api_residual_mode();
int api_precise_passes();
double api_residual_tol();
void api_keep_pool_memory();
hipError_t api_pool_alloc(void **p, size_t bytes, hipStream_t stream);
hipError_t api_take_side_stream(hipStream_t *out, int *device);
void api_return_side_stream(hipStream_t s, int device);
}

namespace
{
#define FVB_HIP_CHECK(expr)                                                                                  \
    do                                                                                                       \
    {                                                                                                        \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return api_fail(-100 - (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

int sign_of(int x)
{
    return (x > 0) - (x < 0);
}

// First-neighbour table in the reference's order (+x, -x, +y, -y, +z, -z, limited by
// spatial-dims), -1 where there is no neighbour. Returns "" or an error message.
// dirs (optional): [V] which of the reference's six offsets (+x -x +y -y +z -z = 0 .. 5, inference_vb.cc:863-869) each
// listed neighbour was found with, 3 bits per list slot (7 = none): the split sweep's records are laid out by
// direction, and a sum over the listed neighbours in list order is then a sum over the directions in order
std::string build_neighbours(const int32_t *coords, int V, int dims, std::vector<int32_t> &nn, std::vector<int32_t> *dirs = nullptr)
{
    nn.assign((size_t)V * 6, -1);
    if (dirs)
        dirs->assign((size_t)V, 0777777);
    if (V == 0)
        return "";
    const int32_t *X = coords, *Y = coords + V, *Z = coords + 2 * (size_t)V;
    for (int v = 0; v + 1 < V; v++) // CheckCoordMatrixCorrectlyOrdered, :769-793
        if (sign_of(X[v + 1] - X[v]) + 10 * sign_of(Y[v + 1] - Y[v]) + 100 * sign_of(Z[v + 1] - Z[v]) <= 0)
            return "Coordinate matrix must be in correct order to use adjacency-based priors.";
    int xsize = 0, ysize = 0;
    for (int v = 0; v < V; v++)
    {
        xsize = std::max(xsize, (int)X[v] + 1);
        ysize = std::max(ysize, (int)Y[v] + 1);
    }
    std::vector<long long> offsets(V);
    for (int v = 0; v < V; v++)
        offsets[v] = (long long)Z[v] * xsize * ysize + (long long)Y[v] * xsize + X[v];
    const long long delta[6] = { 1, -1, xsize, -xsize, (long long)xsize * ysize, -(long long)xsize * ysize };
    const int max_delta = dims * 2 - 1;
    // The reference finds "the voxel at offset pos + delta" by binary search in the (sorted)
    // offsets. For a mask that fills a fair share of its bounding box the same question is one
    // look-up in a dense offset -> voxel map; the search is kept for sparse / odd geometries.
    const long long span = offsets[V - 1] - offsets[0] + 1;
    std::vector<int32_t> dense;
    if (span > 0 && span <= std::max<long long>(64LL * V, 1 << 20))
    {
        dense.assign((size_t)span, -1);
        for (int v = 0; v < V; v++)
            dense[(size_t)(offsets[v] - offsets[0])] = v;
    }
    auto find = [&](long long target) -> int {
        if (!dense.empty())
        {
            const long long rel = target - offsets[0];
            return (rel < 0 || rel >= span) ? -1 : dense[(size_t)rel];
        }
        auto it = std::lower_bound(offsets.begin(), offsets.end(), target);
        return (it == offsets.end() || *it != target) ? -1 : (int)(it - offsets.begin());
    };
    bool non_negative = true;
    for (int v = 0; v < V && non_negative; v++)
        non_negative = X[v] >= 0 && Y[v] >= 0 && Z[v] >= 0;
    if (non_negative && !dense.empty())
    {
        // With non-negative co-ordinates pos % xsize == x and pos % (xsize ysize) == y xsize + x,
        // so the four wrap-around tests (:906-925) read "x is on the last/first column" and "y is
        // on the last/first row"; and every relation found this way is mutual by construction
        // (the voxel found at pos + delta finds this one at its pos - delta), which is what the
        // reference verifies at :958-962.
        const long long base = offsets[0];
        for (int v = 0; v < V; v++)
        {
            const bool ok[6] = { X[v] < xsize - 1, X[v] > 0, Y[v] < ysize - 1, Y[v] > 0, true, true };
            const long long rel0 = offsets[v] - base;
            int32_t *row = &nn[(size_t)v * 6];
            int slot = 0;
            for (int n = 0; n <= max_delta; n++)
            {
                const long long rel = rel0 + delta[n];
                if (!ok[n] || rel < 0 || rel >= span)
                    continue;
                const int32_t found = dense[(size_t)rel];
                if (found >= 0)
                {
                    if (dirs)
                        (*dirs)[(size_t)v] = ((*dirs)[(size_t)v] & ~(7 << (3 * slot))) | (n << (3 * slot));
                    row[slot++] = found;
                }
            }
        }
        return "";
    }
    for (int v = 0; v < V; v++)
    {
        const long long pos = offsets[v];
        for (int n = 0; n <= max_delta; n++)
        {
            const int found = find(pos + delta[n]);
            if (found < 0)
                continue;
            if (n < 4) // wrap-around test, :906-925
            {
                bool ignore = false;
                if (delta[n] > 0)
                {
                    const long long test = delta[n + 2];
                    if (test > 0)
                        ignore = (pos % test) >= test - delta[n];
                }
                else
                {
                    const long long test = -delta[n + 2];
                    if (test > 0)
                        ignore = (pos % test) < -delta[n];
                }
                if (ignore)
                    continue;
            }
            // keep the reference's list order: entries are appended, so compact to the front
            int32_t *row = &nn[(size_t)v * 6];
            int slot = 0;
            while (row[slot] >= 0)
                slot++;
            row[slot] = (int32_t)found;
            if (dirs)
                (*dirs)[(size_t)v] = ((*dirs)[(size_t)v] & ~(7 << (3 * slot))) | (n << (3 * slot));
        }
    }
    // every neighbour relation must be mutual (:958-962)
    for (int v = 0; v < V; v++)
        for (int a = 0; a < 6 && nn[(size_t)v * 6 + a] >= 0; a++)
        {
            const int u = nn[(size_t)v * 6 + a];
            int back = 0;
            for (int b = 0; b < 6; b++)
                back += (nn[(size_t)u * 6 + b] == v);
            if (back != 1)
                return "Each of this voxel's neighbours must have this voxel as a neighbour";
        }
    return "";
}

// Device memory from the device's stream-ordered pool. The pool keeps what a run gives back (release threshold
// set to "never" the first time), so a caller that runs volume after volume pays for its ~20 allocations once:
// 4 ms per run of the 128^3 configuration with hipMalloc / hipFree.
struct DevMem
{
    void *p = nullptr;
    hipStream_t stream = nullptr;
    bool plain = false; // hipMalloc'ed (fine-grained memory another device writes into), not from the pool
    bool fine = false;  // ... and really fine-grained (alloc_fine falls back to ordinary device memory)
    ~DevMem()
    {
        reset();
    }
    void reset()
    {
        if (p && plain)
            (void)hipFree(p);
        else if (p)
            (void)hipFreeAsync(p, stream);
        p = nullptr;
        plain = false;
        fine = false;
    }
    // memory that a kernel on ANOTHER device writes while a kernel on this one polls it (the inboxes of the slab sweep
    // across devices): fine-grained, i.e. not held in this device's L2 between the polls
    hipError_t alloc_fine(size_t bytes)
    {
        plain = true;
        hipError_t e = hipExtMallocWithFlags(&p, bytes ? bytes : 8, hipDeviceMallocFinegrained);
        fine = (e == hipSuccess);
        if (e != hipSuccess)
        {
            // ordinary device memory: good for slabs that share a device; across devices a remote store might stay
            // invisible to the polling device's L2, so the caller takes the level-chunk pipeline then (gran_fine)
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes ? bytes : 8);
        }
        return e;
    }
    hipError_t alloc(size_t bytes, hipStream_t s = nullptr)
    {
        stream = s;
        return fvb::api_pool_alloc(&p, bytes ? bytes : 8, s);
    }
};

// ---- the same table built on the device -----------------------------------------------------------
// For the usual geometry (non-negative co-ordinates, a mask that fills a fair share of its bounding
// box) the table is a handful of independent look-ups per voxel: 22 ms of single-threaded host
// time plus a 50 MB upload for 128^3 voxels, well under a millisecond as three kernels on the
// co-ordinates (24 MB upload). Anything else takes the host path above.
struct GeomScan
{
    int32_t xmax, ymax, cmin, bad_order;
    int32_t zmin, zmax, lmin, lmax; // z and x + y + z: what the slab numbering of the split sweep needs
};

__global__ __launch_bounds__(256) void geom_scan_kernel(const int32_t *coords, int V, GeomScan *out)
{
    const int32_t *X = coords, *Y = coords + V, *Z = coords + 2 * (size_t)V;
    int xmax = 0, ymax = 0, cmin = 0, bad = 0;
    int zmin = INT_MAX, zmax = INT_MIN, lmin = INT_MAX, lmax = INT_MIN;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x)
    {
        xmax = max(xmax, X[v]);
        ymax = max(ymax, Y[v]);
        cmin = min(cmin, min(X[v], min(Y[v], Z[v])));
        zmin = min(zmin, Z[v]);
        zmax = max(zmax, Z[v]);
        lmin = min(lmin, X[v] + Y[v] + Z[v]);
        lmax = max(lmax, X[v] + Y[v] + Z[v]);
        if (v + 1 < V) // CheckCoordMatrixCorrectlyOrdered, inference_vb.cc:769-793
        {
            const int dx = X[v + 1] - X[v], dy = Y[v + 1] - Y[v], dz = Z[v + 1] - Z[v];
            const int key = ((dx > 0) - (dx < 0)) + 10 * ((dy > 0) - (dy < 0)) + 100 * ((dz > 0) - (dz < 0));
            bad |= (key <= 0);
        }
    }
    atomicMax(&out->xmax, xmax);
    atomicMax(&out->ymax, ymax);
    atomicMin(&out->cmin, cmin);
    atomicMin(&out->zmin, zmin);
    atomicMax(&out->zmax, zmax);
    atomicMin(&out->lmin, lmin);
    atomicMax(&out->lmax, lmax);
    if (bad)
        atomicOr(&out->bad_order, 1);
}

// Slab-major numbering of the split sweep on the device (vb_spatial.h, "slab form"): key = (slab, level) of a voxel;
// a histogram, the prefix sums (on the host: a few ten thousand keys) and one more pass that hands out the positions
// of a key's run in arrival order - which voxel of a run gets which of its positions changes no result.
__device__ __forceinline__ int slab_key(const int32_t *coords, int V, int v, int zmin, int dz, int lmin, int nl)
{
    const int x = coords[v], y = coords[(size_t)V + v], z = coords[2 * (size_t)V + v];
    return ((z - zmin) / dz) * nl + (x + y + z - lmin);
}
__global__ __launch_bounds__(256) void slab_count_kernel(const int32_t *coords, int V, int zmin, int dz, int lmin, int nl, int32_t *count)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V)
        atomicAdd(count + slab_key(coords, V, v, zmin, dz, lmin, nl), 1);
}
__global__ __launch_bounds__(256) void slab_place_kernel(const int32_t *coords, int V, int zmin, int dz, int lmin, int nl, int32_t *next,
    int32_t *pos_of)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V)
        pos_of[v] = atomicAdd(next + slab_key(coords, V, v, zmin, dz, lmin, nl), 1);
}

__global__ __launch_bounds__(256) void geom_dense_kernel(const int32_t *coords, int V, int xsize, int ysize, long long base,
    int32_t *dense)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V)
        return;
    const long long off = (long long)coords[2 * (size_t)V + v] * xsize * ysize + (long long)coords[(size_t)V + v] * xsize + coords[v];
    dense[off - base] = v;
}

// Vb::CalcNeighbours (inference_vb.cc:830-964) for non-negative co-ordinates: the wrap-around tests
// (:906-925) read "x is on the last / first column", "y is on the last / first row"
__global__ __launch_bounds__(256) void geom_neighbours_kernel(const int32_t *coords, int V, int xsize, int ysize, long long base,
    long long span, int max_delta, const int32_t *dense, int32_t *nn, int32_t *dirs)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V)
        return;
    const int x = coords[v], y = coords[(size_t)V + v], z = coords[2 * (size_t)V + v];
    const long long rel0 = (long long)z * xsize * ysize + (long long)y * xsize + x - base;
    const long long delta[6] = { 1, -1, xsize, -xsize, (long long)xsize * ysize, -(long long)xsize * ysize };
    const bool ok[6] = { x < xsize - 1, x > 0, y < ysize - 1, y > 0, true, true };
    int32_t row[6] = { -1, -1, -1, -1, -1, -1 };
    int slot = 0;
    int32_t dir = 0777777; // (see build_neighbours)
#pragma unroll
    for (int n = 0; n < 6; n++)
    {
        const long long rel = rel0 + delta[n];
        if (n > max_delta || !ok[n] || rel < 0 || rel >= span)
            continue;
        const int32_t found = dense[rel];
        if (found >= 0)
        {
#pragma unroll
            for (int q = 0; q < 6; q++) // (compile-time indices: the row stays in registers)
                if (q == slot)
                    row[q] = found;
            dir = (dir & ~(7 << (3 * slot))) | (n << (3 * slot));
            slot++;
        }
    }
    dirs[v] = dir;
#pragma unroll
    for (int q = 0; q < 6; q++)
        nn[(size_t)v * 6 + q] = row[q];
}

// Returns 0 (d_nn filled), 1 (geometry not suited: use the host path) or a negative error code.
struct DenseMap
{
    DevMem map; // [span] voxel at box offset base + i, -1 = none
    long long base = 0, span = 0;
    int xsize = 0, ysize = 0;
};

int build_neighbours_device(const int32_t *h_coords, int V, int dims, int32_t *d_nn, int32_t *d_dirs, hipStream_t stream, std::string &err,
    DevMem *keep_coords = nullptr, GeomScan *scan_out = nullptr, DenseMap *keep_dense = nullptr)
{
#define FVB_GEOM_CHECK(expr)                                                                                 \
    do                                                                                                       \
    {                                                                                                        \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
        {                                                                                                    \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                                         \
            return -100 - (int)e_;                                                                           \
        }                                                                                                    \
    } while (0)
    DevMem d_coords, d_scan, d_dense;
    FVB_GEOM_CHECK(d_coords.alloc(sizeof(int32_t) * 3 * (size_t)V, stream));
    FVB_GEOM_CHECK(d_scan.alloc(sizeof(GeomScan), stream));
    FVB_GEOM_CHECK(hipMemcpyAsync(d_coords.p, h_coords, sizeof(int32_t) * 3 * (size_t)V, hipMemcpyHostToDevice, stream));
    GeomScan scan0 = { 0, 0, 0, 0, INT_MAX, INT_MIN, INT_MAX, INT_MIN };
    FVB_GEOM_CHECK(hipMemcpyAsync(d_scan.p, &scan0, sizeof(GeomScan), hipMemcpyHostToDevice, stream));
    FVB_GEOM_CHECK(hipStreamSynchronize(stream)); // (scan0 is a local)
    const unsigned blocks = (unsigned)std::min(1024, (V + 255) / 256);
    hipLaunchKernelGGL(geom_scan_kernel, dim3(blocks), dim3(256), 0, stream, (const int32_t *)d_coords.p, V, (GeomScan *)d_scan.p);
    GeomScan scan;
    FVB_GEOM_CHECK(hipMemcpyAsync(&scan, d_scan.p, sizeof(scan), hipMemcpyDeviceToHost, stream));
    FVB_GEOM_CHECK(hipStreamSynchronize(stream));
    if (scan.bad_order)
    {
        err = "Coordinate matrix must be in correct order to use adjacency-based priors.";
        return -41;
    }
    if (scan.cmin < 0)
        return 1;
    const int xsize = scan.xmax + 1, ysize = scan.ymax + 1;
    const int32_t *X = h_coords, *Y = h_coords + V, *Z = h_coords + 2 * (size_t)V;
    const long long first = (long long)Z[0] * xsize * ysize + (long long)Y[0] * xsize + X[0];
    const long long last = (long long)Z[V - 1] * xsize * ysize + (long long)Y[V - 1] * xsize + X[V - 1];
    const long long span = last - first + 1;
    if (span <= 0 || span > std::max<long long>(64LL * V, 1 << 20))
        return 1;
    FVB_GEOM_CHECK(d_dense.alloc(sizeof(int32_t) * (size_t)span, stream));
    FVB_GEOM_CHECK(hipMemsetAsync(d_dense.p, 0xff, sizeof(int32_t) * (size_t)span, stream)); // -1
    const unsigned grid = (unsigned)((V + 255) / 256);
    hipLaunchKernelGGL(geom_dense_kernel, dim3(grid), dim3(256), 0, stream, (const int32_t *)d_coords.p, V, xsize, ysize, first,
        (int32_t *)d_dense.p);
    hipLaunchKernelGGL(geom_neighbours_kernel, dim3(grid), dim3(256), 0, stream, (const int32_t *)d_coords.p, V, xsize, ysize,
        first, span, dims * 2 - 1, (const int32_t *)d_dense.p, d_nn, d_dirs);
    FVB_GEOM_CHECK(hipGetLastError());
    FVB_GEOM_CHECK(hipStreamSynchronize(stream)); // the temporaries are freed on return
#undef FVB_GEOM_CHECK
    if (scan_out)
        *scan_out = scan;
    if (keep_coords) // the caller goes on with the co-ordinates on the device (slab numbering)
    {
        std::swap(keep_coords->p, d_coords.p);
        std::swap(keep_coords->stream, d_coords.stream);
    }
    if (keep_dense) // ... and with the map from box offsets to voxels (the prep kernel's tiles)
    {
        std::swap(keep_dense->map.p, d_dense.p);
        std::swap(keep_dense->map.stream, d_dense.stream);
        keep_dense->base = first;
        keep_dense->span = span;
        keep_dense->xsize = xsize;
        keep_dense->ysize = ysize;
    }
    return 0;
}

} // namespace

// Which statistics a configuration's state image carries (vb_spatial_noise.h), or -1: no spatial kernels for it
static int spatial_noise_kind(const fvb_config *cfg)
{
    if (cfg->noise == FVB_NOISE_WHITE)
        return cfg->n_phis == 1 ? FVB_SPNZ_WHITE : (cfg->n_phis == 2 ? FVB_SPNZ_PATTERN2 : (cfg->n_phis <= 4 ? FVB_SPNZ_PATTERN4 : (cfg->n_phis <= 8 ? FVB_SPNZ_PATTERN8 : -1)));
    if (cfg->noise == FVB_NOISE_AR1 && cfg->n_phis == 1 && cfg->ar_cross_terms == 0)
        return FVB_SPNZ_AR1;
    if (cfg->noise == FVB_NOISE_AR1 && cfg->n_phis == 2 && cfg->ar_cross_terms >= 0 && cfg->ar_cross_terms <= 2)
        return FVB_SPNZ_ARN2 + cfg->ar_cross_terms;
    return -1;
}
static const char *const spatial_noise_refusal
    = "spatial VB runs white noise with up to 8 noise precisions and AR(1) noise with one or two echoes";
// the kernel table of a configuration (setup == NULL: none was built for this model / parameter count / noise model)
static SpatialKernels spatial_kernels_for(const fvb_config *cfg)
{
    const int kind = spatial_noise_kind(cfg), P = cfg->n_params;
    const bool need_f = cfg->need_f != 0;
    if (kind < 0)
        return SpatialKernels{};
    if (kind >= FVB_SPNZ_ARN2 && kind <= FVB_SPNZ_ARN4)
        return get_spatial_kernels_nz_arn(cfg->model, P, need_f, kind);
    if (kind == FVB_SPNZ_PATTERN8)
        return get_spatial_kernels_nz_pattern8(cfg->model, P, need_f);
    switch (cfg->model)
    {
    case FVB_MODEL_POLY:
        return kind == FVB_SPNZ_WHITE ? get_spatial_kernels_poly(P, need_f) : get_spatial_kernels_nz_poly(P, need_f, kind);
    case FVB_MODEL_LINEAR:
        return kind == FVB_SPNZ_WHITE ? get_spatial_kernels_linear(P, need_f) : get_spatial_kernels_nz_linear(P, need_f, kind);
    case FVB_MODEL_EXP:
        return kind == FVB_SPNZ_WHITE ? get_spatial_kernels_exp(P, need_f) : get_spatial_kernels_nz_exp(P, need_f, kind);
    case FVB_MODEL_HOSTJAC:
        return kind == FVB_SPNZ_WHITE ? get_spatial_kernels_host(P, need_f) : get_spatial_kernels_nz_host(P, need_f, kind);
    default:
        return SpatialKernels{};
    }
}
// entries of the noise block of the result MVN (WhiteParams / Ar1cParams::OutputAsMVN)
static int spatial_noise_outputs(const fvb_config *cfg)
{
    return cfg->noise == FVB_NOISE_AR1 ? 2 + cfg->ar_cross_terms + cfg->n_phis : cfg->n_phis;
}

// One spatial VB run on one device: geometry, work buffers and the per-iteration steps. A single
// process drives it from run_spatial() below; with several slabs the caller interleaves the steps
// with its collectives (all-reduce of the a_K sums, halo exchange of the boundary planes).
struct fvb_spatial_run
{
    fvb_config cfg;
    fvb_spatial sp;
    SpatialKernels k;
    SpatialArgs sa;
    hipStream_t stream = nullptr;
    int V = 0, P = 0, owned_begin = 0, owned_end = 0;
    bool has_spatial = false;
    size_t noise_lds = 0; // dynamic LDS of the set-up and second-sweep kernels (noise-pattern: the class of every timepoint)
    std::vector<int32_t> level_begin;
    std::vector<long long> level_value; // the level (weighted co-ordinate sum) of each entry of level_begin
    int level_w[3] = { 1, 1, 1 };
    DevMem d_state, d_nn, d_nn_dir, d_order, d_aK, d_partials, d_fprior, d_status, d_sa, d_sums, d_seg_start;
    int n_segments = 0;
    double t_geometry_ms = 0, t_neighbours_ms = 0;
    // the split first sweep (vb_spatial.h): whole-volume runs, or one of several slabs that sweep together
    bool allow_fast = false, fast = false;
    bool multi_fast = false; // one of several slabs on several devices that sweep together (fabber_vb_run_spatial_host_multi)
    int device_share = 1;    // how many such slabs run on THIS device at once (a device listed several times)
    bool gran_fine = false;  // multi_fast: the inboxes are fine-grained memory (another DEVICE may write them)
    DevMem d_up_pos;
    DenseMap dense; // (kept from the neighbour table's kernels)
    std::vector<int32_t> h_pos_of; // (multi_fast: the numbering, for the slab below to address this slab's inboxes)
    int fast_prep(int it);
    int fast_sweep();
    int fast_noise(int it);
    int link_up(fvb_spatial_run &upper, int global_first, int upper_global_first);
    std::vector<int32_t> level_begin_counts; // voxels per level
    DevMem d_pos_of, d_level_pos, d_level_count, d_sw_f64, d_sw_i32, d_sw_sync, d_sw_gran, d_slab_first;
    int max_runs_per_slab = 0;
    bool slab_form = false; // (= fast) the voxels are numbered slab-major for vb_spatial_slab_sweep_kernel
    int sweep_fast(int it);
    int fast_failed(bool &failed);
    // the second-sweep kernel of iteration `it`: the instance with the half-ulp exp where the iteration ends in one
    // of the run's pointwise linearisations (vb_spatial.h: sp_precise) and such an instance was built
    SpatialKernelFn second_sweep(bool fast_form, int it) const
    {
        const bool pointwise = it + 1 < sa.ka.precise_passes && !sa.locked_linear;
        SpatialKernelFn acc = fast_form ? k.noise_fast_acc : k.noise_acc;
        return (pointwise && acc) ? acc : (fast_form ? k.noise_fast : k.noise);
    }
    // host-evaluated models: the linearisations the set-up re-centre reads (see HostLin below)
    const double *lin_cur = nullptr, *lin_next = nullptr;
    hipStream_t setup_stream = nullptr;
    int setup_device = 0;
    hipEvent_t setup_done = nullptr;
    ~fvb_spatial_run()
    {
        if (setup_stream)
        {
            (void)hipStreamSynchronize(setup_stream); // (before the buffers its kernel writes are given back)
            fvb::api_return_side_stream(setup_stream, setup_device); // (kept for the next run on this device)
        }
        if (setup_done)
            (void)hipEventDestroy(setup_done);
    }

    int open(const fvb_config *cfg_, const fvb_spatial *sp_, const void *d_data, const fvb_outputs *d_out, hipStream_t stream_);
    int ak_sums(double *host_sums);
    int ak_segment_sums(double *host_partials);
    int set_ak_sums(const double *host_sums);
    int sweep(int it);
    int sweep_levels(int it, long long lo, long long hi);
    int sweep_noise(int it);
    int copy_means(int v_begin, int v_count, double *host_means, int32_t *host_status, bool to_device);
    int finish();
};

int fvb_spatial_run::open(const fvb_config *cfg_, const fvb_spatial *sp_, const void *d_data, const fvb_outputs *d_out,
    hipStream_t stream_)
{
    cfg = *cfg_;
    sp = *sp_;
    stream = stream_;
    V = cfg.n_voxels;
    P = cfg.n_params;
    owned_begin = 0;
    owned_end = V;
    if (sp.owned_end > sp.owned_begin)
    {
        owned_begin = sp.owned_begin;
        owned_end = sp.owned_end;
    }
    if (owned_begin < 0 || owned_end > V)
        return api_fail(-45, "owned voxel range outside the local voxel list");
    if (spatial_noise_kind(&cfg) < 0)
        return api_fail(-44, spatial_noise_refusal);
    k = spatial_kernels_for(&cfg);
    if (!k.setup)
        return api_fail(-40, "no spatial kernel instantiation for this model / parameter count / noise model");
    noise_lds = k.lds_classes ? (size_t)cfg.n_times : 0;

    // ---- Vb::SetupPerVoxelDists for every local voxel (ghosts included: their initial means are what the
    // neighbouring slab starts from too) needs the series and the options only: it runs on a stream of its own
    // while the host and this run's stream work out the geometry ----
    FVB_HIP_CHECK(d_state.alloc(sizeof(double) * (size_t)k.state_rows * V, stream));
    FVB_HIP_CHECK(d_status.alloc(sizeof(int32_t) * (size_t)V, stream));
    int n_unmasked = cfg.n_times;
    double nz_count[8] = { (double)cfg.n_times, 0, 0, 0, 0, 0, 0, 0 }; // timepoints per noise precision (trace of Q_k)
    if (cfg.phi_index) // (a device pointer here: read it back once)
    {
        std::vector<uint8_t> h(cfg.n_times);
        FVB_HIP_CHECK(hipMemcpyAsync(h.data(), cfg.phi_index, h.size(), hipMemcpyDeviceToHost, stream));
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
        n_unmasked = 0;
        nz_count[0] = 0;
        for (int t = 0; t < cfg.n_times; t++)
        {
            n_unmasked += (h[t] != 255);
            if (h[t] < 8)
                nz_count[h[t]] += 1;
        }
    }
    {
        SpatialArgs early;
        memset(&early, 0, sizeof(early));
        early.lin_cur = lin_cur;
        early.lin_next = lin_next;
        early.ka.cfg = cfg;
        early.ka.out = *d_out;
        early.ka.data = d_data;
        early.ka.n_unmasked = n_unmasked;
        early.ka.residual_mode = api_residual_mode();
        early.ka.residual_tol = api_residual_tol();
        early.ka.precise_passes = api_precise_passes();
        early.state = (double *)d_state.p;
        early.status = (int32_t *)d_status.p;
        early.owned_begin = owned_begin;
        early.owned_end = owned_end;
        for (int i = 0; i < 8; i++)
            early.nz_count[i] = nz_count[i];
        early.locked_centres = sp.locked_centres;
        early.locked_linear = sp.locked_centres != nullptr;
        FVB_HIP_CHECK(fvb::api_take_side_stream(&setup_stream, &setup_device));
        FVB_HIP_CHECK(hipEventCreateWithFlags(&setup_done, hipEventDisableTiming));
        // (the two buffers were allocated in `stream`'s order; the series is the caller's, complete in `stream`'s order too)
        FVB_HIP_CHECK(hipEventRecord(setup_done, stream));
        FVB_HIP_CHECK(hipStreamWaitEvent(setup_stream, setup_done, 0));
        hipLaunchKernelGGL(k.setup, dim3((unsigned)((V + 63) / 64)), dim3(64), noise_lds, setup_stream, early);
        FVB_HIP_CHECK(hipGetLastError());
        FVB_HIP_CHECK(hipEventRecord(setup_done, setup_stream));
    }

    // ---- geometry: neighbour table on the device where the geometry allows, else on the host ----
    const auto t_start = std::chrono::steady_clock::now();
    FVB_HIP_CHECK(d_nn.alloc(sizeof(int32_t) * (size_t)V * 6, stream));
    FVB_HIP_CHECK(d_nn_dir.alloc(sizeof(int32_t) * (size_t)std::max(V, 1), stream));
    std::string err;
    DevMem d_coords;
    GeomScan scan;
    const int on_device = (V > 0 && !getenv("FVB_SPATIAL_HOST_GEOMETRY"))
        ? build_neighbours_device(sp.coords, V, sp.spatial_dims, (int32_t *)d_nn.p, (int32_t *)d_nn_dir.p, stream, err, &d_coords, &scan, &dense) : 1;
    if (on_device < 0)
        return api_fail(on_device, err);
    if (on_device == 1)
    {
        std::vector<int32_t> nn, dirs;
        err = build_neighbours(sp.coords, V, sp.spatial_dims, nn, &dirs);
        if (!err.empty())
            return api_fail(-41, err);
        FVB_HIP_CHECK(hipMemcpyAsync(d_nn.p, nn.data(), sizeof(int32_t) * (size_t)V * 6, hipMemcpyHostToDevice, stream));
        FVB_HIP_CHECK(hipMemcpyAsync(d_nn_dir.p, dirs.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, stream));
        FVB_HIP_CHECK(hipStreamSynchronize(stream)); // nn is a local
    }
    t_neighbours_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    const int32_t *X = sp.coords, *Y = sp.coords + V, *Z = sp.coords + 2 * (size_t)V;
    // Level function a x + b y + c z: a stencil offset that leads to a smaller voxel index must
    // lower the level, one that leads to a larger index must raise it. First neighbours only: (1,1,1). Second
    // neighbours too (the per-level kernel sums the neighbours of neighbours for types P, p, e.g. (x+1, y-1) which
    // has a smaller index - the sum is multiplied by the 0 of priors.cc:455, but a NaN in it is not lost): b > a and
    // c > b, so (1,2,3). The split form treats types P, p as local (vb_spatial.h) and numbers with (1,1,1).
    bool second_neighbours = false, minus_zero = false;
    for (int kk = 0; kk < P; kk++)
    {
        const bool second = (cfg.prior_type[kk] == FVB_PRIOR_SPATIAL_P || cfg.prior_type[kk] == FVB_PRIOR_SPATIAL_p);
        second_neighbours |= second;
        has_spatial |= cfg.prior_type[kk] >= FVB_PRIOR_SPATIAL_M;
        // (prec0 mean0 = -0: the sign of the reference's 0 x sum would decide the sign of a zero prior mean)
        const double pm0 = cfg.prior_prec[kk] * cfg.prior_mean[kk];
        minus_zero |= second && pm0 == 0 && std::signbit(pm0);
    }
    long long cy = 1, cz = 1;
    const int n_owned = owned_end - owned_begin;
    auto level_of = [&](int i) -> long long {
        const int v = owned_begin + i;
        return (long long)X[v] + cy * Y[v] + cz * Z[v];
    };
    // a few host threads over contiguous index ranges (the passes are memory-bound scans of the
    // co-ordinates); per-thread histograms keep the counting sort stable
    int nt = (n_owned >= (1 << 18)) ? (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())) : 1;
    if (const char *forced = getenv("FVB_SPATIAL_HOST_THREADS")) // tests: threads on small volumes
        nt = std::max(1, std::min(64, atoi(forced)));
    nt = std::max(1, std::min(nt, std::max(n_owned, 1)));
    auto chunk = [&](int t) { return (int)((long long)n_owned * t / nt); };
    auto parallel = [&](const std::function<void(int)> &body) {
        if (nt == 1)
            return body(0);
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++)
            pool.emplace_back(body, t);
        body(0);
        for (auto &th : pool)
            th.join();
    };
    long long lmin = 0, lmax = 0;
    auto scan_levels = [&]() {
        std::vector<long long> tmin(nt, 0), tmax(nt, 0);
        parallel([&](int t) {
            long long lo = 0, hi = 0;
            for (int i = chunk(t); i < chunk(t + 1); i++)
            {
                const long long l = level_of(i);
                lo = (i == chunk(t) || l < lo) ? l : lo;
                hi = (i == chunk(t) || l > hi) ? l : hi;
            }
            tmin[t] = lo;
            tmax[t] = hi;
        });
        bool first = true;
        lmin = lmax = 0;
        for (int t = 0; t < nt; t++)
            if (chunk(t + 1) > chunk(t))
            {
                lmin = (first || tmin[t] < lmin) ? tmin[t] : lmin;
                lmax = (first || tmax[t] > lmax) ? tmax[t] : lmax;
                first = false;
            }
    };
    scan_levels();
    // The level order (voxel ids sorted by level) is what the per-level launches walk; the slab form of the split
    // sweep numbers the voxels itself (below) and does without it - 2.5 ms of a 128^3 run's set-up.
    std::vector<int32_t> order(1);
    auto build_level_order = [&]() {
        order.assign(std::max(n_owned, 1), 0);
        level_begin.clear();
        level_value.clear();
        level_w[0] = 1;
        level_w[1] = (int)cy;
        level_w[2] = (int)cz;
        if (lmax - lmin < (1LL << 22))
        {
            // counting sort (stable: voxels of a level stay in index order)
            const size_t nl = (size_t)(lmax - lmin + 1);
            std::vector<std::vector<int32_t> > count(nt, std::vector<int32_t>(nl, 0));
            parallel([&](int t) {
                int32_t *c = count[t].data();
                for (int i = chunk(t); i < chunk(t + 1); i++)
                    c[(size_t)(level_of(i) - lmin)]++;
            });
            int32_t running = 0;
            for (size_t l = 0; l < nl; l++)
            {
                const int32_t begin = running;
                for (int t = 0; t < nt; t++) // thread order = index order
                {
                    const int32_t n = count[t][l];
                    count[t][l] = running; // becomes this thread's first slot in level l
                    running += n;
                }
                if (running > begin)
                {
                    level_begin.push_back(begin);
                    level_value.push_back(lmin + (long long)l);
                }
            }
            level_begin.push_back(n_owned);
            parallel([&](int t) {
                int32_t *c = count[t].data();
                for (int i = chunk(t); i < chunk(t + 1); i++)
                    order[c[(size_t)(level_of(i) - lmin)]++] = owned_begin + i;
            });
        }
        else
        {
            std::vector<int32_t> idx(n_owned);
            for (int i = 0; i < n_owned; i++)
                idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return level_of(a) < level_of(b); });
            for (int i = 0; i < n_owned; i++)
            {
                if (i == 0 || level_of(idx[i]) != level_of(idx[i - 1]))
                {
                    level_begin.push_back(i);
                    level_value.push_back(level_of(idx[i]));
                }
                order[i] = owned_begin + idx[i];
            }
            level_begin.push_back(n_owned);
        }
    };
    // ---- numbering for the split first sweep: the parameters the ordered part updates are those of types M, m ----
    int n_spatial = 0, spatial_param[FVB_MAX_PARAMS] = { 0 };
    for (int kk = 0; kk < P; kk++)
        if (cfg.prior_type[kk] == FVB_PRIOR_SPATIAL_M || cfg.prior_type[kk] == FVB_PRIOR_SPATIAL_m)
            spatial_param[n_spatial++] = kk;
    const bool whole = owned_begin == 0 && owned_end == V;
    const bool eligible = allow_fast && has_spatial && !minus_zero && (whole || multi_fast) && n_owned > 0 && !getenv("FVB_SPATIAL_PER_LEVEL");
    std::vector<int32_t> pos_of, level_pos, level_count, slab_first;
    int n_pos = 0, sl_width = 64, sl_max_run = 0;
    slab_form = false;
    if (eligible && lmax - lmin < (1LL << 22))
    {
        // Slab-major numbering: a slab = dz z-planes, inside a slab the voxels level by level (index order in a
        // level). dz: as few planes as keep the slabs within the chip's workgroups (every slab is one resident
        // workgroup): one plane per slab up to 192 planes. Thicker slabs mean fewer hand-overs between workgroups
        // (2.4 us each, one after the other) but longer runs and fewer groups per workgroup to hide the records'
        // latency: measured at 128^3, 0.51 ms per sweep with dz = 1, 0.65 with 2, 0.94 with 3, 1.04 with 4.
        // (with the co-ordinates on the device - the usual case - the numbering is three small kernels there; the
        // host does it with its threads otherwise: 3 ms at 128^3 against 0.2)
        // (a slab with ghost planes is numbered on the host: the device kernels number every local voxel)
        const bool on_dev = d_coords.p != nullptr && !getenv("FVB_SPATIAL_HOST_NUMBERING") && whole;
        int zmin = Z[owned_begin], zmax = Z[owned_begin];
        if (on_dev)
        {
            zmin = scan.zmin;
            zmax = scan.zmax;
        }
        else
            for (int v = owned_begin; v < owned_end; v++)
            {
                zmin = std::min(zmin, (int)Z[v]);
                zmax = std::max(zmax, (int)Z[v]);
            }
        const long long nz = (long long)zmax - zmin + 1;
        // The sweep's workgroups (1024 lanes, up to 128 KB of LDS: one per compute unit) wait for the slab below.
        // Workgroups are dispatched in index order, so the one waited for is resident or finished; all the same the
        // count stays within what THIS device (a partition of the chip in CPX mode has 32 compute units, not 256)
        // holds at once, three quarters of it at most, shared between the runs that sweep on it together. A wait
        // that does not end gives up (slab_wait_inbox) and the run is repeated with the per-level launches.
        int cus = 256, dev_now = 0;
        if (hipGetDevice(&dev_now) == hipSuccess)
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_now);
        const long long slab_cap = std::max(1LL, std::min(192LL, (long long)cus * 3 / 4 / std::max(1, device_share)));
        long long dz = std::max(1LL, (nz + slab_cap - 1) / slab_cap);
        if (const char *forced = getenv("FVB_SPATIAL_SLAB_DZ"))
            dz = std::max((nz + slab_cap - 1) / slab_cap, (long long)std::max(1, atoi(forced)));
        const long long n_slabs = (nz + dz - 1) / dz;
        const size_t nl = (size_t)(lmax - lmin + 1);
        if (sp.spatial_dims <= 3 && n_slabs * (long long)nl <= (1LL << 21))
        {
            const size_t nk = (size_t)n_slabs * nl;
            auto key_of = [&](int i) -> size_t {
                return (size_t)((Z[owned_begin + i] - zmin) / dz) * nl + (size_t)(level_of(i) - lmin);
            };
            const int nth = on_dev ? 1 : nt;
            std::vector<std::vector<int32_t> > count(nth, std::vector<int32_t>(nk, 0));
            DevMem d_keys;
            const unsigned vgrid = (unsigned)((V + 255) / 256);
            if (on_dev)
            {
                FVB_HIP_CHECK(d_keys.alloc(sizeof(int32_t) * nk, stream));
                FVB_HIP_CHECK(hipMemsetAsync(d_keys.p, 0, sizeof(int32_t) * nk, stream));
                hipLaunchKernelGGL(slab_count_kernel, dim3(vgrid), dim3(256), 0, stream, (const int32_t *)d_coords.p, V, zmin, (int)dz,
                    (int)lmin, (int)nl, (int32_t *)d_keys.p);
                FVB_HIP_CHECK(hipMemcpyAsync(count[0].data(), d_keys.p, sizeof(int32_t) * nk, hipMemcpyDeviceToHost, stream));
                FVB_HIP_CHECK(hipStreamSynchronize(stream));
            }
            else
                parallel([&](int t) {
                    int32_t *c = count[t].data();
                    for (int i = chunk(t); i < chunk(t + 1); i++)
                        c[key_of(i)]++;
                });
            slab_first.assign((size_t)n_slabs + 1, 0);
            int32_t running = 0;
            for (size_t key = 0; key < nk; key++)
            {
                if (key % nl == 0)
                    slab_first[key / nl] = (int32_t)level_pos.size();
                const int32_t begin = running;
                for (int t = 0; t < nth; t++) // thread order = index order
                {
                    const int32_t n = count[t][key];
                    count[t][key] = running;
                    running += n;
                }
                if (running > begin)
                {
                    level_pos.push_back(begin);
                    level_count.push_back(running - begin);
                    sl_max_run = std::max(sl_max_run, (int)(running - begin));
                }
            }
            slab_first[(size_t)n_slabs] = (int32_t)level_pos.size();
            if (sl_max_run <= 8192 && n_slabs <= slab_cap)
            {
                if (on_dev)
                {
                    // count[0] holds every key's first position now: hand the positions out on the device
                    FVB_HIP_CHECK(d_pos_of.alloc(sizeof(int32_t) * (size_t)V, stream));
                    FVB_HIP_CHECK(hipMemcpyAsync(d_keys.p, count[0].data(), sizeof(int32_t) * nk, hipMemcpyHostToDevice, stream));
                    hipLaunchKernelGGL(slab_place_kernel, dim3(vgrid), dim3(256), 0, stream, (const int32_t *)d_coords.p, V, zmin, (int)dz,
                        (int)lmin, (int)nl, (int32_t *)d_keys.p, (int32_t *)d_pos_of.p);
                    FVB_HIP_CHECK(hipGetLastError());
                    FVB_HIP_CHECK(hipStreamSynchronize(stream)); // (count[0] and d_keys go out of scope)
                }
                else
                {
                    pos_of.assign((size_t)V, 0);
                    parallel([&](int t) {
                        int32_t *c = count[t].data();
                        for (int i = chunk(t); i < chunk(t + 1); i++)
                            pos_of[(size_t)owned_begin + i] = c[key_of(i)]++;
                    });
                }
                n_pos = (n_owned + 15) / 16 * 16;
                // lanes per run: the next power of two from 64 that holds the longest run, 1024 at most
                while (sl_width < sl_max_run && sl_width < 1024)
                    sl_width *= 2;
                if (const char *forced = getenv("FVB_SPATIAL_SLAB_WIDTH")) // tests: lanes that take several voxels of a run
                    sl_width = std::max(64, std::min(1024, atoi(forced) / 64 * 64));
                if (1024 % sl_width != 0) // (the 1024 lanes are whole groups)
                    sl_width = 64;
                slab_form = true;
                level_begin_counts = level_count;
            }
            else
            {
                level_pos.clear();
                level_count.clear();
                sl_max_run = 0;
            }
        }
    }
    if (!slab_form)
    {
        // the per-level launches: the exact form, with the second-neighbour levels where types P, p are about
        if (second_neighbours)
        {
            cy = 2;
            cz = 3;
            scan_levels();
        }
        build_level_order();
    }
    fast = slab_form;
    if (slab_form && !whole)
    {
        // ghosts have no position: what stands in sw_npos for them says where their mean comes from (vb_spatial.h)
        for (int v = 0; v < owned_begin; v++)
            pos_of[(size_t)v] = FVB_NP_BELOW;
        for (int v = owned_end; v < V; v++)
            pos_of[(size_t)v] = FVB_NP_ABOVE;
    }
    if (multi_fast)
        h_pos_of = pos_of;
    t_geometry_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();

    // ---- device memory ----
    // segments of the a_K sums: every z-plane of the owned voxels, cut every 4096 voxels from its first
    std::vector<int32_t> seg_start;
    for (int v = owned_begin; v < owned_end; v++)
        if (v == owned_begin || Z[v] != Z[v - 1] || v - seg_start.back() >= 4096)
            seg_start.push_back(v);
    seg_start.push_back(owned_end);
    const int n_blocks = (int)seg_start.size() - 1;
    n_segments = n_blocks;
    FVB_HIP_CHECK(d_order.alloc(sizeof(int32_t) * order.size(), stream));
    FVB_HIP_CHECK(d_aK.alloc(sizeof(double) * FVB_MAX_PARAMS, stream));
    FVB_HIP_CHECK(d_sums.alloc(sizeof(double) * FVB_MAX_PARAMS * 2, stream));
    FVB_HIP_CHECK(d_partials.alloc(sizeof(double) * (size_t)std::max(n_blocks, 1) * P * 2, stream));
    FVB_HIP_CHECK(d_seg_start.alloc(sizeof(int32_t) * seg_start.size(), stream));
    FVB_HIP_CHECK(hipMemcpyAsync(d_seg_start.p, seg_start.data(), sizeof(int32_t) * seg_start.size(), hipMemcpyHostToDevice, stream));
    FVB_HIP_CHECK(d_fprior.alloc(sizeof(double), stream));
    FVB_HIP_CHECK(hipMemcpyAsync(d_order.p, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice, stream));
    double aK0[FVB_MAX_PARAMS];
    for (int i = 0; i < FVB_MAX_PARAMS; i++)
        aK0[i] = 1e-8; // priors.cc:185
    FVB_HIP_CHECK(hipMemcpyAsync(d_aK.p, aK0, sizeof(aK0), hipMemcpyHostToDevice, stream));
    FVB_HIP_CHECK(hipMemsetAsync(d_fprior.p, 0, sizeof(double), stream));
    FVB_HIP_CHECK(hipMemsetAsync(d_sums.p, 0, sizeof(double) * FVB_MAX_PARAMS * 2, stream));
    FVB_HIP_CHECK(hipMemsetAsync(d_partials.p, 0, sizeof(double) * (size_t)std::max(n_blocks, 1) * P * 2, stream));

    memset(&sa, 0, sizeof(sa));
    sa.lin_cur = lin_cur;
    sa.lin_next = lin_next;
    sa.ka.cfg = cfg;
    sa.ka.out = *d_out;
    sa.ka.data = d_data;
    sa.ka.save = nullptr;
    sa.ka.residual_mode = api_residual_mode();
    sa.ka.residual_tol = api_residual_tol();
    sa.ka.precise_passes = api_precise_passes();
    sa.state = (double *)d_state.p;
    sa.nn = (const int32_t *)d_nn.p;
    sa.nn_dir = (const int32_t *)d_nn_dir.p;
    sa.order = (const int32_t *)d_order.p;
    sa.aK = (double *)d_aK.p;
    sa.ak_sums = (double *)d_sums.p;
    sa.partials = (double *)d_partials.p;
    sa.seg_start = (const int32_t *)d_seg_start.p;
    sa.fprior_last = (double *)d_fprior.p;
    sa.status = (int32_t *)d_status.p;
    sa.spatial_dims = sp.spatial_dims;
    sa.update_first_iter = sp.update_first_iter;
    sa.spatial_speed = sp.spatial_speed;
    sa.q1 = sp.q1;
    sa.q2 = sp.q2;
    sa.n_blocks = n_blocks;
    sa.owned_begin = owned_begin;
    sa.owned_end = owned_end;
    sa.n_voxels_global = sp.n_voxels_global > 0 ? sp.n_voxels_global : V;
    for (int i = 0; i < 8; i++)
        sa.nz_count[i] = nz_count[i];
    sa.locked_centres = sp.locked_centres;
    sa.locked_linear = sp.locked_centres != nullptr;
    if (fast)
    {
        const size_t NP = (size_t)n_pos, ns = (size_t)n_spatial;
        if (!d_pos_of.p)
        {
            FVB_HIP_CHECK(d_pos_of.alloc(sizeof(int32_t) * (size_t)V, stream));
            FVB_HIP_CHECK(hipMemcpyAsync(d_pos_of.p, pos_of.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, stream));
        }
        FVB_HIP_CHECK(d_level_pos.alloc(sizeof(int32_t) * level_pos.size(), stream));
        FVB_HIP_CHECK(d_level_count.alloc(sizeof(int32_t) * level_count.size(), stream));
        FVB_HIP_CHECK(hipMemcpyAsync(d_level_pos.p, level_pos.data(), sizeof(int32_t) * level_pos.size(), hipMemcpyHostToDevice, stream));
        FVB_HIP_CHECK(hipMemcpyAsync(d_level_count.p, level_count.data(), sizeof(int32_t) * level_count.size(), hipMemcpyHostToDevice, stream));
        // doubles: x, pm, pre, rhsk, pprec, q [ns][NP] each; sigk [ns][ns][NP]; nbr [ns][3][NP]
        const size_t n_f64 = std::max<size_t>(1, (6 * ns + ns * ns + 3 * ns) * NP);
        FVB_HIP_CHECK(d_sw_f64.alloc(sizeof(double) * n_f64, stream));
        const size_t n_i32 = 5 * NP;
        FVB_HIP_CHECK(d_sw_i32.alloc(sizeof(int32_t) * n_i32, stream)); // npos [4][NP], alive [NP]
        FVB_HIP_CHECK(d_sw_sync.alloc(64, stream));                      // flags
        const size_t gran_bytes = std::max<size_t>(16, sizeof(unsigned long long) * 2 * ns * NP);
        if (multi_fast)
        {
            FVB_HIP_CHECK(d_sw_gran.alloc_fine(gran_bytes));
            gran_fine = d_sw_gran.fine;
        }
        else
            FVB_HIP_CHECK(d_sw_gran.alloc(gran_bytes, stream));
        FVB_HIP_CHECK(hipMemsetAsync(d_sw_gran.p, 0, gran_bytes, stream));
        sa.sl_remote = multi_fast ? 1 : 0;
        sa.sw_gran = (unsigned long long *)d_sw_gran.p;
        sa.sw_serial = 0;
        FVB_HIP_CHECK(hipMemsetAsync(d_sw_i32.p, 0, sizeof(int32_t) * n_i32, stream));
        FVB_HIP_CHECK(hipMemsetAsync(d_sw_f64.p, 0, sizeof(double) * n_f64, stream));
        FVB_HIP_CHECK(hipMemsetAsync(d_sw_sync.p, 0, 64, stream));
        double *f = (double *)d_sw_f64.p;
        sa.sw_x = f;
        sa.sw_pm = f + ns * NP;
        sa.sw_pre = f + 2 * ns * NP;
        sa.sw_rhsk = f + 3 * ns * NP;
        sa.sw_pprec = f + 4 * ns * NP;
        sa.sw_q = f + 5 * ns * NP;
        sa.sw_sigk = f + 6 * ns * NP;
        sa.sw_nbr = f + (6 * ns + ns * ns) * NP;
        FVB_HIP_CHECK(d_slab_first.alloc(sizeof(int32_t) * slab_first.size(), stream));
        FVB_HIP_CHECK(hipMemcpyAsync(d_slab_first.p, slab_first.data(), sizeof(int32_t) * slab_first.size(), hipMemcpyHostToDevice, stream));
        sa.n_slabs = (int32_t)slab_first.size() - 1;
        sa.sl_first_run = (const int32_t *)d_slab_first.p;
        sa.sl_width = sl_width;
        sa.sl_max_run = sl_max_run;
        max_runs_per_slab = 0;
        for (size_t b = 0; b + 1 < slab_first.size(); b++)
            max_runs_per_slab = std::max(max_runs_per_slab, (int)(slab_first[b + 1] - slab_first[b]));
        sa.sw_npos = (int32_t *)d_sw_i32.p;
        sa.sw_alive = (int32_t *)d_sw_i32.p + 4 * NP;
        sa.sw_flags = (int32_t *)d_sw_sync.p + 4;
        sa.pos_of = (const int32_t *)d_pos_of.p;
        sa.n_pos = n_pos;
        sa.n_spatial = n_spatial;
        for (int i = 0; i < n_spatial; i++)
            sa.spatial_param[i] = spatial_param[i];
        sa.sw_level_pos = (const int32_t *)d_level_pos.p;
        sa.sw_level_count = (const int32_t *)d_level_count.p;
        sa.n_levels = (int32_t)level_pos.size();
        if (dense.map.p && !getenv("FVB_SPATIAL_PREP_LINEAR")) // the prep kernel's tiles (vb_spatial.h)
        {
            const long long z0 = Z[owned_begin], z1 = Z[owned_end - 1];
            const long long tnx = (dense.xsize + 7) / 8, tny = (dense.ysize + 7) / 8, tiles = (z1 - z0 + 1) * tnx * tny;
            // (a mask that fills little of its box would spend the kernel on empty tiles)
            if (tiles > 0 && tiles * 64 <= 4LL * n_owned + 4096 && tiles < (1LL << 30))
            {
                sa.dense = (const int32_t *)dense.map.p;
                sa.dense_base = dense.base;
                sa.dense_span = dense.span;
                sa.xsize = dense.xsize;
                sa.ysize = dense.ysize;
                sa.tile_nx = (int32_t)tnx;
                sa.tile_ny = (int32_t)tny;
                sa.tile_z0 = (int32_t)z0;
                sa.n_tiles = (int32_t)tiles;
            }
        }
    }
    sa.ka.n_unmasked = n_unmasked;
    // the argument block the per-level launches read (nothing in it changes per launch)
    FVB_HIP_CHECK(d_sa.alloc(sizeof(SpatialArgs), stream));
    FVB_HIP_CHECK(hipMemcpyAsync(d_sa.p, &sa, sizeof(SpatialArgs), hipMemcpyHostToDevice, stream));
    FVB_HIP_CHECK(hipStreamSynchronize(stream)); // `sa`, nn, order are pageable host memory

    // everything after this waits for the set-up kernel (started at the top)
    FVB_HIP_CHECK(hipStreamWaitEvent(stream, setup_done, 0));
    return 0;
}

int fvb_spatial_run::ak_sums(double *host_sums)
{
    hipLaunchKernelGGL(k.ak_partial, dim3(sa.n_blocks), dim3(256), 0, stream, sa);
    hipLaunchKernelGGL(k.ak_reduce, dim3(1), dim3(64), 0, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    if (host_sums)
    {
        FVB_HIP_CHECK(hipMemcpyAsync(host_sums, d_sums.p, sizeof(double) * 2 * P, hipMemcpyDeviceToHost, stream));
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
    }
    return 0;
}

// the segments' partial sums [n_segments][P][2] (for callers that add up several slabs' segments in order)
int fvb_spatial_run::ak_segment_sums(double *host_partials)
{
    if (sa.n_blocks > 0)
        hipLaunchKernelGGL(k.ak_partial, dim3(sa.n_blocks), dim3(256), 0, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    FVB_HIP_CHECK(hipMemcpyAsync(host_partials, d_partials.p, sizeof(double) * 2 * P * (size_t)sa.n_blocks, hipMemcpyDeviceToHost, stream));
    FVB_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}

int fvb_spatial_run::set_ak_sums(const double *host_sums)
{
    if (host_sums)
        FVB_HIP_CHECK(hipMemcpyAsync(d_sums.p, host_sums, sizeof(double) * 2 * P, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k.ak_final, dim3(1), dim3(64), 0, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    if (host_sums)
        FVB_HIP_CHECK(hipStreamSynchronize(stream)); // the caller's buffer is pageable
    return 0;
}

int fvb_spatial_run::sweep(int it)
{
    int rc = sweep_levels(it, LLONG_MIN, LLONG_MAX);
    return rc ? rc : sweep_noise(it);
}

// first sweep, the levels with lo <= value < hi (one launch per level)
int fvb_spatial_run::sweep_levels(int it, long long lo, long long hi)
{
    sa.it = it;
    const SpatialArgs *sap = (const SpatialArgs *)d_sa.p;
    for (size_t l = 0; l + 1 < level_begin.size(); l++)
    {
        if (level_value[l] < lo || level_value[l] >= hi)
            continue;
        const int begin = level_begin[l], count = level_begin[l + 1] - level_begin[l];
        hipLaunchKernelGGL(k.theta, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, stream, sap, begin, count, it);
    }
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}

int fvb_spatial_run::sweep_noise(int it)
{
    sa.it = it;
    const int n_owned = owned_end - owned_begin;
    if (n_owned > 0)
        hipLaunchKernelGGL(second_sweep(false, it), dim3((unsigned)((n_owned + 63) / 64)), dim3(64), noise_lds, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}

// One iteration's first and second sweep with the split first sweep (see vb_spatial.h)
// the three launches of an iteration with the slab form of the split sweep, one by one (a run of several slabs on several
// devices puts its exchanges between them; one device runs them back to back: sweep_fast)
int fvb_spatial_run::fast_prep(int it)
{
    sa.it = it;
    sa.sw_serial++; // this sweep's number
    const int n_owned = owned_end - owned_begin;
    // (a multiple of 8 workgroups: the kernel deals them out to the XCDs in contiguous eighths of the voxel list)
    const int waves = sa.n_tiles > 0 ? sa.n_tiles : (n_owned + 63) / 64;
    hipLaunchKernelGGL(k.prep, dim3((unsigned)((waves + 7) / 8 * 8)), dim3(64), 0, stream, (const SpatialArgs *)d_sa.p, it, sa.sw_serial);
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}
int fvb_spatial_run::fast_sweep()
{
    if (sa.n_spatial == 0) // (types P, p only: nothing waits for a neighbour)
        return 0;
    const int which = sa.n_spatial <= 1 ? 0 : (sa.n_spatial == 2 ? 1 : 2);
    // one workgroup of 1024 lanes per slab (within the device's compute units, see the numbering)
    if (sa.n_spatial > 8)
        return api_fail(-40, "more than eight parameters with a first-neighbour prior");
    const size_t lds = sizeof(double) * (8 + 2 * (size_t)sa.n_spatial * sa.sl_max_run) + sizeof(int32_t) * (2 * (size_t)max_runs_per_slab + 2);
    if (lds > 48 * 1024)
        FVB_HIP_CHECK(hipFuncSetAttribute((const void *)k.slab_sweep[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k.slab_sweep[which], dim3((unsigned)sa.n_slabs), dim3(1024), lds, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}
int fvb_spatial_run::fast_noise(int it)
{
    sa.it = it;
    const int n_owned = owned_end - owned_begin;
    hipLaunchKernelGGL(second_sweep(true, it), dim3((unsigned)((n_owned + 63) / 64)), dim3(64), noise_lds, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}

// This slab's top plane hands its new means to the bottom plane of `upper` (another device, or another stream of this
// one): where in upper's granule array the inbox of every top-plane voxel's z+1 neighbour is. global_first = the global
// index of this slab's local voxel 0 (upper_global_first: of upper's).
int fvb_spatial_run::link_up(fvb_spatial_run &upper, int global_first, int upper_global_first)
{
    std::vector<int32_t> nn((size_t)V * 6);
    FVB_HIP_CHECK(hipMemcpyAsync(nn.data(), d_nn.p, sizeof(int32_t) * nn.size(), hipMemcpyDeviceToHost, stream));
    FVB_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int32_t> up_pos((size_t)sa.n_pos, -1);
    for (int v = owned_begin; v < owned_end; v++)
        for (int a = 0; a < 6; a++)
        {
            const int u = nn[(size_t)v * 6 + a];
            if (u < owned_end)
                continue; // (none, or not a ghost above)
            const long long in_upper = (long long)global_first + u - upper_global_first;
            if (in_upper < upper.owned_begin || in_upper >= upper.owned_end)
                return api_fail(-49, "slab decomposition: a ghost voxel is not owned by the slab above");
            up_pos[(size_t)h_pos_of[(size_t)v]] = upper.h_pos_of[(size_t)in_upper];
        }
    FVB_HIP_CHECK(d_up_pos.alloc(sizeof(int32_t) * up_pos.size(), stream));
    FVB_HIP_CHECK(hipMemcpyAsync(d_up_pos.p, up_pos.data(), sizeof(int32_t) * up_pos.size(), hipMemcpyHostToDevice, stream));
    sa.sw_up_pos = (const int32_t *)d_up_pos.p;
    sa.sw_gran_up = upper.sa.sw_gran;
    sa.up_n_pos = upper.sa.n_pos;
    FVB_HIP_CHECK(hipMemcpyAsync(d_sa.p, &sa, sizeof(SpatialArgs), hipMemcpyHostToDevice, stream));
    FVB_HIP_CHECK(hipStreamSynchronize(stream)); // (up_pos is a local)
    return 0;
}

int fvb_spatial_run::sweep_fast(int it)
{
    int rc = fast_prep(it);
    if (rc == 0)
        rc = fast_sweep();
    return rc ? rc : fast_noise(it);
}

int fvb_spatial_run::fast_failed(bool &failed)
{
    failed = false;
    if (!fast)
        return 0;
    int32_t flag = 0;
    FVB_HIP_CHECK(hipMemcpyAsync(&flag, sa.sw_flags, sizeof(flag), hipMemcpyDeviceToHost, stream));
    FVB_HIP_CHECK(hipStreamSynchronize(stream));
    failed = flag != 0;
    return 0;
}

int fvb_spatial_run::copy_means(int v_begin, int v_count, double *host_means, int32_t *host_status, bool to_device)
{
    if (v_begin < 0 || v_count < 0 || v_begin + v_count > V)
        return api_fail(-46, "voxel range outside the local voxel list");
    if (v_count == 0)
        return 0;
    // rows 0..P-1 of the state image are the posterior means (SpLayout::M)
    double *dev = (double *)d_state.p + v_begin;
    const size_t width = sizeof(double) * (size_t)v_count;
    if (host_means)
    {
        if (to_device)
            FVB_HIP_CHECK(copy_rows(dev, sizeof(double) * (size_t)V, host_means, width, width, P, hipMemcpyDefault, stream));
        else
            FVB_HIP_CHECK(copy_rows(host_means, width, dev, sizeof(double) * (size_t)V, width, P, hipMemcpyDefault, stream));
    }
    if (host_status)
    {
        int32_t *ds = (int32_t *)d_status.p + v_begin;
        if (to_device)
            FVB_HIP_CHECK(hipMemcpyAsync(ds, host_status, sizeof(int32_t) * (size_t)v_count, hipMemcpyDefault, stream));
        else
            FVB_HIP_CHECK(hipMemcpyAsync(host_status, ds, sizeof(int32_t) * (size_t)v_count, hipMemcpyDefault, stream));
    }
    FVB_HIP_CHECK(hipStreamSynchronize(stream));
    return 0;
}

int fvb_spatial_run::finish()
{
    sa.it = cfg.max_iterations;
    hipLaunchKernelGGL(k.pack, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, stream, sa);
    FVB_HIP_CHECK(hipGetLastError());
    FVB_HIP_CHECK(hipStreamSynchronize(stream)); // the work buffers are freed with the object
    return 0;
}

namespace
{
// A model that is evaluated on the host (FVB_MODEL_HOSTJAC) under spatial VB: the two places of the loop that
// run the model - the set-up re-centre and the re-centre that ends every iteration's second sweep
// (inference_vb.cc:235,695) - become a call of the caller's linearisation callback for the voxels still
// taking part, about the means the first sweep left, and an upload of g and J; the kernels read them through
// HostLinModel. Two device buffers [V][T (P + 1)]: the second sweep needs the linearisation about the OLD
// centre for k = y - g + J (centre - mean) next to the new one.
struct HostLin
{
    fvb_linearise_fn linearise = nullptr;
    void *user = nullptr;
    const double *init_means = nullptr; // host, [V][P]: the centres of the set-up re-centre
    size_t V = 0, T = 0;
    int P = 0;
    DevMem buf[2];
    int cur = 0; // buf[cur] belongs to the moments in the state
    std::vector<double> lin, means_rows, means;
    std::vector<int32_t> status, ids;
    size_t stride() const
    {
        return T * (size_t)(P + 1);
    }
    int open(const fvb_config *cfg, hipStream_t stream)
    {
        V = (size_t)cfg->n_voxels;
        T = (size_t)cfg->n_times;
        P = cfg->n_params;
        const double bytes = (double)V * stride() * sizeof(double);
        if (bytes > 48e9)
            return api_fail(-55, "host-evaluated model under spatial VB: two linearisation buffers of " + std::to_string((long long)(bytes / 1e9))
                + " GB each do not fit the budget (48 GB each)");
        if (getenv("FVB_SPATIAL_VERBOSE") || bytes > 4e9)
            fprintf(stderr, "[fvb spatial] host-evaluated model: the linearisations of the whole volume are resident twice on the device "
                            "and once on the host, %.2f GB each\n", bytes / 1e9);
        for (int i = 0; i < 2; i++)
            FVB_HIP_CHECK(buf[i].alloc(sizeof(double) * V * stride(), stream));
        lin.resize(V * stride());
        return 0;
    }
    // g and J of the voxels with status 0 about means [V][P] (voxel-major) into buf[which]
    int relinearise(const double *centres, const int32_t *voxel_status, int which, hipStream_t stream)
    {
        ids.clear();
        for (size_t v = 0; v < V; v++)
            if (!voxel_status || voxel_status[v] == 0)
                ids.push_back((int32_t)v);
        if (ids.empty())
            return 0;
        const bool all = ids.size() == V;
        const double *active = centres;
        if (!all)
        {
            means.resize(ids.size() * (size_t)P);
            for (size_t a = 0; a < ids.size(); a++)
                for (int i = 0; i < P; i++)
                    means[a * P + i] = centres[(size_t)ids[a] * P + i];
            active = means.data();
        }
        const int cb = linearise(user, (int32_t)ids.size(), ids.data(), active, lin.data());
        if (cb != 0)
            return api_fail(-54, "the model's linearisation callback failed (code " + std::to_string(cb) + ")");
        if (!all) // spread the active voxels' blocks out to their own places (back to front: in place)
            for (size_t a = ids.size(); a-- > 0;)
                if ((size_t)ids[a] != a)
                    memmove(lin.data() + (size_t)ids[a] * stride(), lin.data() + a * stride(), sizeof(double) * stride());
        FVB_HIP_CHECK(hipMemcpyAsync(buf[which].p, lin.data(), sizeof(double) * V * stride(), hipMemcpyHostToDevice, stream));
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
        return 0;
    }
};

int run_spatial(const fvb_config *cfg, const fvb_spatial *sp, const void *d_data, const fvb_outputs *d_out,
    hipStream_t stream, void (*progress_cb)(int, int), bool allow_fast = true, HostLin *hl = nullptr)
{
    const bool timing = getenv("FVB_SPATIAL_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t_start = now();
    fvb_spatial_run run;
    run.allow_fast = allow_fast && !hl; // (a host model's means must be complete before the second sweep starts)
    int rc;
    if (hl)
    {
        if ((rc = hl->open(cfg, stream)) != 0 || (rc = hl->relinearise(hl->init_means, nullptr, 0, stream)) != 0)
            return rc;
        hl->cur = 0;
        run.lin_cur = run.lin_next = (const double *)hl->buf[0].p;
    }
    rc = run.open(cfg, sp, d_data, d_out, stream);
    if (rc)
        return rc;
    const auto t_open = now();
    for (int it = 0; it < cfg->max_iterations; it++)
    {
        if (progress_cb)
            progress_cb(it, cfg->max_iterations); // inference_vb.cc:610
        if (run.has_spatial && (it > 0 || sp->update_first_iter))
        {
            if ((rc = run.ak_sums(nullptr)) != 0 || (rc = run.set_ak_sums(nullptr)) != 0)
                return rc;
        }
        if (hl)
        {
            // first sweep; the model about the new means (host); second sweep
            if ((rc = run.sweep_levels(it, LLONG_MIN, LLONG_MAX)) != 0)
                return rc;
            const size_t V = hl->V;
            const int P = hl->P;
            hl->means_rows.resize(V * (size_t)P);
            hl->status.resize(V);
            if ((rc = run.copy_means(0, (int)V, hl->means_rows.data(), hl->status.data(), false)) != 0)
                return rc;
            std::vector<double> centres(V * (size_t)P); // [V][P]
            for (int i = 0; i < P; i++)
                for (size_t v = 0; v < V; v++)
                    centres[v * P + i] = hl->means_rows[(size_t)i * V + v];
            const int spare = 1 - hl->cur;
            if ((rc = hl->relinearise(centres.data(), hl->status.data(), spare, stream)) != 0)
                return rc;
            run.sa.lin_cur = (const double *)hl->buf[hl->cur].p;
            run.sa.lin_next = (const double *)hl->buf[spare].p;
            if ((rc = run.sweep_noise(it)) != 0)
                return rc;
            hl->cur = spare;
            continue;
        }
        if ((rc = (run.fast ? run.sweep_fast(it) : run.sweep(it))) != 0)
            return rc;
    }
    bool failed = false;
    if ((rc = run.fast_failed(failed)) != 0)
        return rc;
    if (failed)
    {
        // a voxel failed DURING a first sweep (or the sweep's barrier gave up): the split sweep does not
        // reproduce what that does to the voxels after it. Nothing has been written to the outputs that the
        // repeat does not overwrite: do the run again with the per-level launches.
        if (timing || getenv("FVB_SPATIAL_VERBOSE"))
            fprintf(stderr, "[fvb spatial] split first sweep abandoned, repeating the run with per-level launches\n");
        return run_spatial(cfg, sp, d_data, d_out, stream, nullptr, false);
    }
    const auto t_enq = now();
    if ((rc = run.finish()) != 0)
        return rc;
    if (timing)
        fprintf(stderr, "[fvb spatial] V=%d levels/runs=%zu: geometry %.1f ms (neighbours %.1f), alloc+upload+setup %.1f ms, enqueue %.1f ms, drain %.1f ms\n",
            run.V, run.slab_form ? run.level_begin_counts.size() : run.level_begin.size() - 1, run.t_geometry_ms, run.t_neighbours_ms, ms(t_start, t_open) - run.t_geometry_ms, ms(t_open, t_enq),
            ms(t_enq, now()));
    return 0;
}
} // namespace

extern "C" {

static int32_t run_spatial_checked(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out, void *stream,
    void (*progress_cb)(int, int), HostLin *hl);

int32_t fabber_vb_run_spatial_device(const fvb_config *cfg, const fvb_spatial *sp, const void *data,
    const fvb_outputs *out, void *stream, void (*progress_cb)(int, int))
{
    if (cfg && cfg->model == FVB_MODEL_HOSTJAC)
        return api_fail(-56, "a model evaluated on the host runs spatial VB through fabber_vb_run_spatial_hostmodel_host");
    return run_spatial_checked(cfg, sp, data, out, stream, progress_cb, nullptr);
}

static int32_t run_spatial_checked(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out, void *stream,
    void (*progress_cb)(int, int), HostLin *hl)
{
    int rc = api_validate(cfg, true);
    if (rc)
        return rc;
    if (!sp || !sp->coords)
        return api_fail(-42, "spatial description / coordinates missing");
    if (sp->spatial_dims < 0 || sp->spatial_dims > 3)
        return api_fail(-43, "spatial-dims must be 0, 1, 2 or 3");
    if (spatial_noise_kind(cfg) < 0)
        return api_fail(-44, spatial_noise_refusal);
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    if (cfg->n_voxels == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    return run_spatial(cfg, sp, data, out, (hipStream_t)stream, progress_cb, true, hl);
}

int32_t fabber_vb_spatial_open(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    void *stream, fvb_spatial_run **run)
{
    if (!run)
        return api_fail(-47, "run handle pointer is NULL");
    *run = nullptr;
    int rc = api_validate(cfg, true);
    if (rc)
        return rc;
    if (!sp || !sp->coords)
        return api_fail(-42, "spatial description / coordinates missing");
    if (sp->spatial_dims < 0 || sp->spatial_dims > 3)
        return api_fail(-43, "spatial-dims must be 0, 1, 2 or 3");
    if (spatial_noise_kind(cfg) < 0)
        return api_fail(-44, spatial_noise_refusal);
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    if (cfg->n_voxels == 0 || !data)
        return api_fail(-21, "no voxels / data is NULL");
    fvb_spatial_run *r = new fvb_spatial_run();
    rc = r->open(cfg, sp, data, out, (hipStream_t)stream);
    if (rc)
    {
        delete r;
        return rc;
    }
    *run = r;
    return 0;
}

int32_t fabber_vb_spatial_ak_sums(fvb_spatial_run *run, double *sums)
{
    return run ? run->ak_sums(sums) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_ak_segment_sums(fvb_spatial_run *run, double *partials, int32_t *n_segments)
{
    if (!run || !n_segments)
        return api_fail(-47, "run handle is NULL");
    *n_segments = run->n_segments;
    return partials ? run->ak_segment_sums(partials) : 0;
}

int32_t fabber_vb_spatial_set_ak_sums(fvb_spatial_run *run, const double *sums)
{
    return run ? run->set_ak_sums(sums) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_sweep(fvb_spatial_run *run, int32_t iteration)
{
    return run ? run->sweep(iteration) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_sweep_levels(fvb_spatial_run *run, int32_t iteration, int64_t level_lo, int64_t level_hi)
{
    return run ? run->sweep_levels(iteration, level_lo, level_hi) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_sweep_noise(fvb_spatial_run *run, int32_t iteration)
{
    return run ? run->sweep_noise(iteration) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_level_weights(fvb_spatial_run *run, int32_t weights[3])
{
    if (!run || !weights)
        return api_fail(-47, "run handle is NULL");
    for (int i = 0; i < 3; i++)
        weights[i] = run->level_w[i];
    return 0;
}

int32_t fabber_vb_spatial_fprior(fvb_spatial_run *run, double *value, int32_t set)
{
    if (!run || !value)
        return api_fail(-47, "run handle is NULL");
    hipError_t e = set ? hipMemcpyAsync(run->d_fprior.p, value, sizeof(double), hipMemcpyDefault, run->stream)
                       : hipMemcpyAsync(value, run->d_fprior.p, sizeof(double), hipMemcpyDefault, run->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(run->stream);
    return e == hipSuccess ? 0 : api_fail(-100 - (int)e, hipGetErrorString(e));
}

int32_t fabber_vb_spatial_copy_means(fvb_spatial_run *run, int32_t v_begin, int32_t v_count, double *means, int32_t *status,
    int32_t to_device)
{
    return run ? run->copy_means(v_begin, v_count, means, status, to_device != 0) : api_fail(-47, "run handle is NULL");
}

int32_t fabber_vb_spatial_close(fvb_spatial_run *run)
{
    if (!run)
        return api_fail(-47, "run handle is NULL");
    const int rc = run->finish();
    delete run;
    return rc;
}

static int32_t run_spatial_host_impl(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    int32_t device, void (*progress_cb)(int, int), HostLin *hl);

int32_t fabber_vb_run_spatial_host(const fvb_config *cfg, const fvb_spatial *sp, const void *data,
    const fvb_outputs *out, int32_t device, void (*progress_cb)(int, int))
{
    if (cfg && cfg->model == FVB_MODEL_HOSTJAC)
        return api_fail(-56, "a model evaluated on the host runs spatial VB through fabber_vb_run_spatial_hostmodel_host");
    return run_spatial_host_impl(cfg, sp, data, out, device, progress_cb, nullptr);
}

int32_t fabber_vb_run_spatial_hostmodel_host(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    int32_t device, fvb_linearise_fn linearise, void *user, void (*progress_cb)(int, int))
{
    if (!cfg || cfg->model != FVB_MODEL_HOSTJAC)
        return api_fail(-56, "fabber_vb_run_spatial_hostmodel_host is for cfg->model = FVB_MODEL_HOSTJAC");
    if (!linearise)
        return api_fail(-50, "linearisation callback is NULL");
    if (!cfg->init_mvn)
        return api_fail(-52, "host-evaluated models need the initial posterior as init_mvn (the model's InitVoxelPosterior runs on the host)");
    if (sp && sp->locked_centres)
        return api_fail(-57, "locked linearisation centres are not available for host-evaluated models under spatial VB");
    // the centres of the set-up re-centre: the means of the initial posterior, voxel-major
    const size_t V = (size_t)cfg->n_voxels;
    const int P = cfg->n_params, n = P + spatial_noise_outputs(cfg), nCov = n * (n + 1) / 2;
    std::vector<double> init_means(V * (size_t)P);
    for (size_t v = 0; v < V; v++)
        for (int i = 0; i < P; i++)
            init_means[v * P + i] = cfg->init_mvn[(size_t)(nCov + i) * V + v];
    HostLin hl;
    hl.linearise = linearise;
    hl.user = user;
    hl.init_means = init_means.data();
    return run_spatial_host_impl(cfg, sp, data, out, device, progress_cb, &hl);
}

static int32_t run_spatial_host_impl(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    int32_t device, void (*progress_cb)(int, int), HostLin *hl)
{
    int rc = api_validate(cfg, true);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return api_fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    FVB_HIP_CHECK(hipSetDevice(device));
    const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times;
    if (V == 0)
        return 0;
    // (before anything is uploaded: a caller that falls back to the host-evaluated route on -40 has not paid for the
    // series on the device twice)
    if (spatial_noise_kind(cfg) < 0)
        return api_fail(-44, spatial_noise_refusal);
    if (!spatial_kernels_for(cfg).setup)
        return api_fail(-40, "no spatial kernel instantiation for this model / parameter count / noise model");
    const int P = cfg->n_params;
    const int n = P + spatial_noise_outputs(cfg), rows = n * (n + 1) / 2 + n + 1;
    const size_t esz = cfg->data_f64 ? 8 : 4;
    fvb_config d = *cfg;
    DevMem b_data, b_design, b_phi, b_init, b_img[FVB_MAX_PARAMS], b_mvn, b_f, b_status, b_it;
    FVB_HIP_CHECK(b_data.alloc(T * V * esz));
    FVB_HIP_CHECK(hipMemcpy(b_data.p, data, T * V * esz, hipMemcpyHostToDevice));
    if (cfg->design)
    {
        FVB_HIP_CHECK(b_design.alloc(sizeof(double) * T * P));
        FVB_HIP_CHECK(hipMemcpy(b_design.p, cfg->design, sizeof(double) * T * P, hipMemcpyHostToDevice));
        d.design = (const double *)b_design.p;
    }
    if (cfg->phi_index)
    {
        FVB_HIP_CHECK(b_phi.alloc(T));
        FVB_HIP_CHECK(hipMemcpy(b_phi.p, cfg->phi_index, T, hipMemcpyHostToDevice));
        d.phi_index = (const uint8_t *)b_phi.p;
    }
    if (cfg->init_mvn)
    {
        FVB_HIP_CHECK(b_init.alloc(sizeof(double) * rows * V));
        FVB_HIP_CHECK(hipMemcpy(b_init.p, cfg->init_mvn, sizeof(double) * rows * V, hipMemcpyHostToDevice));
        d.init_mvn = (const double *)b_init.p;
    }
    for (int kk = 0; kk < P; kk++)
        if (cfg->image_prior[kk])
        {
            FVB_HIP_CHECK(b_img[kk].alloc(sizeof(double) * V));
            FVB_HIP_CHECK(hipMemcpy(b_img[kk].p, cfg->image_prior[kk], sizeof(double) * V, hipMemcpyHostToDevice));
            d.image_prior[kk] = (const double *)b_img[kk].p;
        }
    fvb_outputs dout;
    memset(&dout, 0, sizeof(dout));
    FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * V));
    dout.mvn = (double *)b_mvn.p;
    if (out->free_energy)
    {
        FVB_HIP_CHECK(b_f.alloc(sizeof(double) * V));
        FVB_HIP_CHECK(hipMemset(b_f.p, 0xff, sizeof(double) * V)); // NaN: a voxel that fails before any F is evaluated keeps it
        dout.free_energy = (double *)b_f.p;
    }
    if (out->status)
    {
        FVB_HIP_CHECK(b_status.alloc(sizeof(int32_t) * V));
        dout.status = (int32_t *)b_status.p;
    }
    if (out->iterations)
    {
        FVB_HIP_CHECK(b_it.alloc(sizeof(int32_t) * V));
        dout.iterations = (int32_t *)b_it.p;
    }
    fvb_spatial dsp = *sp;
    DevMem b_locked;
    if (sp->locked_centres)
    {
        FVB_HIP_CHECK(b_locked.alloc(sizeof(double) * P * V));
        FVB_HIP_CHECK(hipMemcpy(b_locked.p, sp->locked_centres, sizeof(double) * P * V, hipMemcpyHostToDevice));
        dsp.locked_centres = (const double *)b_locked.p;
    }
    rc = run_spatial_checked(&d, &dsp, b_data.p, &dout, nullptr, progress_cb, hl);
    if (rc)
        return rc;
    FVB_HIP_CHECK(hipDeviceSynchronize());
    FVB_HIP_CHECK(hipMemcpy(out->mvn, dout.mvn, sizeof(double) * rows * V, hipMemcpyDeviceToHost));
    if (dout.free_energy)
        FVB_HIP_CHECK(hipMemcpy(out->free_energy, dout.free_energy, sizeof(double) * V, hipMemcpyDeviceToHost));
    if (dout.status)
        FVB_HIP_CHECK(hipMemcpy(out->status, dout.status, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.iterations)
        FVB_HIP_CHECK(hipMemcpy(out->iterations, dout.iterations, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    return 0;
}

// ---- spatial VB of one volume on several devices, driven by this one process ---------------------------------
// The decomposition and the schedule of fabber_core_amd/spatial_mgpu.py (one process per GPU over
// torch.distributed) inside the engine: z-slabs with ghost planes, the first sweep as a pipeline over chunks
// of 16 global levels (slab r sweeps chunk c at tick c + r and hands its top planes to slab r + 1 after every
// tick), the a_K sums added over the segments of the voxel list in voxel order, the prior term of the last
// voxel from the last slab, the second sweep, the exchange of the boundary planes both ways. The result is the
// single-device run bit for bit (tests/test_spatial_mgpu.py). Planes travel device to device
// (hipMemcpyPeerAsync between staging buffers; a device listed twice is a copy on itself).
namespace
{
struct SlabRun
{
    int dev = 0, g0 = 0, b = 0, e = 0, g1 = 0; // local list = global voxels [g0, g1), owned [b, e)
    hipStream_t stream = nullptr;
    fvb_config d;
    fvb_spatial sp;
    fvb_outputs dout;
    std::vector<int32_t> coords;
    DevMem b_data, b_design, b_phi, b_init, b_img[FVB_MAX_PARAMS], b_mvn, b_f, b_status, b_it, stage_means, stage_status;
    fvb_spatial_run *run = nullptr;
    ~SlabRun()
    {
        (void)hipSetDevice(dev);
        delete run; // (its buffers go back to this device's pool in its stream's order)
        DevMem *mine[] = { &b_data, &b_design, &b_phi, &b_init, &b_mvn, &b_f, &b_status, &b_it, &stage_means, &stage_status };
        for (DevMem *m : mine) // ... and this slab's, before the stream they are ordered on goes
            m->reset();
        for (DevMem &m : b_img)
            m.reset();
        if (stream)
        {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
    }
};

// means and status of n voxels: slab `from`, local index v_from -> slab `to`, local index v_to
int slab_transfer(SlabRun &from, int v_from, SlabRun &to, int v_to, int n, int P)
{
    if (n <= 0)
        return 0;
    FVB_HIP_CHECK(hipSetDevice(from.dev));
    int rc = from.run->copy_means(v_from, n, (double *)from.stage_means.p, (int32_t *)from.stage_status.p, false);
    if (rc)
        return rc;
    // (in the receiving slab's stream: ordered before the copy into its state, which ends with a wait for that
    // stream - so the sender's staging buffer is free again on return; the sender's copy above has completed)
    FVB_HIP_CHECK(hipSetDevice(to.dev));
    FVB_HIP_CHECK(hipMemcpyPeerAsync(to.stage_means.p, to.dev, from.stage_means.p, from.dev, sizeof(double) * (size_t)P * n, to.stream));
    FVB_HIP_CHECK(hipMemcpyPeerAsync(to.stage_status.p, to.dev, from.stage_status.p, from.dev, sizeof(int32_t) * (size_t)n, to.stream));
    return to.run->copy_means(v_to, n, (double *)to.stage_means.p, (int32_t *)to.stage_status.p, true);
}
} // namespace

static thread_local int s_unlink_pair = -1;
void fabber_vb_test_unlink_slab_pair(int32_t pair)
{
    s_unlink_pair = pair;
}

// One volume on several devices, in three steps a caller can time apart: the slabs and their part of the problem on
// their devices (open), a complete run on the resident data - geometry, set-up, every iteration, result images packed
// on the devices - as often as asked (run), the owned voxels' results into the caller's images (results).
struct fvb_spatial_multi
{
    fvb_config cfg;
    fvb_spatial sp;
    std::vector<int32_t> coords; // (a copy: sp.coords points here)
    std::vector<int> devs;
    int V = 0, T = 0, P = 0, world = 0, halo = 1, rows = 0, max_halo = 1;
    bool second = false, has_spatial = false, peers = true, want_f = false;
    size_t esz = 4;
    std::vector<std::unique_ptr<SlabRun> > slabs;
    const char *route = "";
    double ms_open = 0, ms_setup = 0, ms_loop = 0;

    // body(r) for every slab, each on a host thread of its own; the first failure (code and message) is the caller's
    int for_each_slab(const std::function<int(int)> &body)
    {
        std::vector<int> rcs(slabs.size(), 0);
        std::vector<std::string> errs(slabs.size());
        auto work = [&](int r) {
            rcs[(size_t)r] = body(r);
            if (rcs[(size_t)r] != 0)
                errs[(size_t)r] = fabber_vb_last_error(); // (thread-local: carried to the caller's thread)
        };
        std::vector<std::thread> pool;
        const bool threads = !getenv("FVB_SPATIAL_MULTI_SERIAL");
        for (int r = 1; r < (int)slabs.size() && threads; r++)
            pool.emplace_back(work, r);
        work(0);
        for (int r = 1; r < (int)slabs.size() && !threads; r++)
            work(r);
        for (auto &th : pool)
            th.join();
        for (size_t r = 0; r < slabs.size(); r++)
            if (rcs[r] != 0)
                return api_fail(rcs[r], errs[r]);
        return 0;
    }
    int plan(const fvb_config *cfg_, const fvb_spatial *sp_, const int32_t *devices, int32_t n_devices);
    int upload(const void *data, const fvb_outputs *out);
    int execute(void (*progress_cb)(int, int), bool no_fast);
    int run(void (*progress_cb)(int, int));
    int download(const fvb_outputs *out);
    void release();
    ~fvb_spatial_multi()
    {
        release();
    }
};

void fvb_spatial_multi::release()
{
    // (giving a slab's memory back unmaps it: a thread per slab)
    std::vector<std::thread> pool;
    for (size_t r = 1; r < slabs.size(); r++)
        pool.emplace_back([this, r]() { slabs[r].reset(); });
    if (!slabs.empty())
        slabs[0].reset();
    for (auto &th : pool)
        th.join();
    slabs.clear();
}

// ---- the slabs: cuts on z-plane boundaries, balanced by voxel count; fewer slabs if the planes do not go round ----
int fvb_spatial_multi::plan(const fvb_config *cfg_, const fvb_spatial *sp_, const int32_t *devices, int32_t n_devices)
{
    cfg = *cfg_;
    sp = *sp_;
    V = cfg.n_voxels;
    T = cfg.n_times;
    P = cfg.n_params;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return api_fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    if (devices)
    {
        if (n_devices <= 0)
            return api_fail(-31, "empty device list");
        for (int i = 0; i < n_devices; i++)
        {
            if (devices[i] < 0 || devices[i] >= visible)
                return api_fail(-31, "device index " + std::to_string(devices[i]) + " out of range (" + std::to_string(visible) + " visible)");
            devs.push_back(devices[i]);
        }
    }
    else
        for (int i = 0; i < visible; i++)
            devs.push_back(i);
    coords.assign(sp_->coords, sp_->coords + 3 * (size_t)V);
    sp.coords = coords.data();
    const int32_t *Z = coords.data() + 2 * (size_t)V;
    for (int k = 0; k < P; k++)
    {
        second |= (cfg.prior_type[k] == FVB_PRIOR_SPATIAL_P || cfg.prior_type[k] == FVB_PRIOR_SPATIAL_p);
        has_spatial |= cfg.prior_type[k] >= FVB_PRIOR_SPATIAL_M;
    }
    // the slabs of a run that sweep together write into each other's memory: every pair of neighbours must be peers
    for (size_t r = 0; r + 1 < devs.size() && peers; r++)
        if (devs[r] != devs[r + 1])
        {
            int can = 0;
            peers = hipDeviceCanAccessPeer(&can, devs[r], devs[r + 1]) == hipSuccess && can != 0;
            (void)hipGetLastError();
        }
    // ghost planes: the split form reads first neighbours only (types P, p are local there, and their a_K sums are
    // over first neighbours, priors.cc:280-301); the level-chunk pipeline's per-level kernel - what a run falls back
    // to - sums second neighbours too, so a problem with such priors keeps two planes either way
    halo = second ? 2 : 1;
    std::vector<int> plane_start;
    for (int v = 0; v < V; v++)
    {
        if (v > 0 && Z[v] < Z[v - 1])
            return api_fail(-41, "Coordinate matrix must be in correct order to use adjacency-based priors.");
        if (v == 0 || Z[v] != Z[v - 1])
            plane_start.push_back(v);
    }
    // Every slab keeps at least `halo` planes (its neighbours' ghosts must not reach past it) and leaves as many for
    // each slab after it; within that the cut falls on the plane boundary nearest to an equal share of the voxels.
    // A decomposition that does not work out (a very unbalanced mask, planes missing from the z range) is tried
    // again with one slab fewer, down to the one-device run - never refused.
    const int n_planes = (int)plane_start.size();
    world = (int)std::min<size_t>(devs.size(), std::max<size_t>(1, plane_start.size() / (size_t)(2 * halo)));
    for (; world > 1; world--)
    {
        std::vector<int> cut(1, 0); // plane index at which slab r starts
        for (int r = 1; r < world; r++)
        {
            const int lo = cut.back() + halo, hi = n_planes - (world - r) * halo;
            const double want = (double)V * r / world;
            int best = lo;
            for (int p = lo; p <= hi; p++)
                if (std::fabs(plane_start[p] - want) < std::fabs(plane_start[best] - want))
                    best = p;
            cut.push_back(best);
        }
        std::vector<int> bounds;
        for (int c : cut)
            bounds.push_back(plane_start[c]);
        bounds.push_back(V);
        slabs.clear();
        bool fits = true;
        for (int r = 0; r < world && fits; r++)
        {
            std::unique_ptr<SlabRun> sl(new SlabRun);
            sl->dev = devs[r];
            sl->b = bounds[r];
            sl->e = bounds[r + 1];
            sl->g0 = sl->b;
            sl->g1 = sl->e;
            if (r > 0)
                sl->g0 = (int)(std::lower_bound(Z, Z + V, Z[sl->b] - halo) - Z);
            if (r < world - 1)
                sl->g1 = (int)(std::upper_bound(Z, Z + V, Z[sl->e - 1] + halo) - Z);
            fits = !((r > 0 && sl->g0 < bounds[r - 1]) || (r < world - 1 && sl->g1 > bounds[r + 2]));
            slabs.push_back(std::move(sl));
        }
        if (fits)
            break;
    }
    if (world <= 1)
    {
        world = 1;
        slabs.clear();
    }
    const int n = P + spatial_noise_outputs(&cfg);
    rows = n * (n + 1) / 2 + n + 1;
    esz = cfg.data_f64 ? 8 : 4;
    max_halo = 1;
    for (int r = 0; r < (int)slabs.size(); r++)
        max_halo = std::max(max_halo, std::max(slabs[r]->b - slabs[r]->g0, slabs[r]->g1 - slabs[r]->e));
    return 0;
}

// ---- per slab: its part of the problem on its device (a host thread per slab: the devices work side by side) ----
int fvb_spatial_multi::upload(const void *data, const fvb_outputs *out)
{
    want_f = out->free_energy != nullptr;
    const bool want_status = out->status != nullptr, want_it = out->iterations != nullptr;
    auto upload_slab = [&](int r) -> int {
        SlabRun &sl = *slabs[r];
        const size_t Vl = (size_t)(sl.g1 - sl.g0);
        FVB_HIP_CHECK(hipSetDevice(sl.dev));
        FVB_HIP_CHECK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        hipStream_t st = sl.stream;
        auto upload_rows = [&](void *dst, const void *src, size_t elem, size_t nrows) {
            return copy_rows(dst, Vl * elem, (const char *)src + (size_t)sl.g0 * elem, (size_t)V * elem, Vl * elem, nrows,
                hipMemcpyHostToDevice, st);
        };
        sl.d = cfg;
        sl.d.n_voxels = (int32_t)Vl;
        FVB_HIP_CHECK(sl.b_data.alloc((size_t)T * Vl * esz, st));
        FVB_HIP_CHECK(upload_rows(sl.b_data.p, data, esz, (size_t)T));
        if (cfg.design)
        {
            FVB_HIP_CHECK(sl.b_design.alloc(sizeof(double) * (size_t)T * P, st));
            FVB_HIP_CHECK(hipMemcpyAsync(sl.b_design.p, cfg.design, sizeof(double) * (size_t)T * P, hipMemcpyHostToDevice, st));
            sl.d.design = (const double *)sl.b_design.p;
        }
        if (cfg.phi_index)
        {
            FVB_HIP_CHECK(sl.b_phi.alloc((size_t)T, st));
            FVB_HIP_CHECK(hipMemcpyAsync(sl.b_phi.p, cfg.phi_index, (size_t)T, hipMemcpyHostToDevice, st));
            sl.d.phi_index = (const uint8_t *)sl.b_phi.p;
        }
        if (cfg.init_mvn)
        {
            FVB_HIP_CHECK(sl.b_init.alloc(sizeof(double) * rows * Vl, st));
            FVB_HIP_CHECK(upload_rows(sl.b_init.p, cfg.init_mvn, sizeof(double), (size_t)rows));
            sl.d.init_mvn = (const double *)sl.b_init.p;
        }
        for (int k = 0; k < P; k++)
            if (cfg.image_prior[k])
            {
                FVB_HIP_CHECK(sl.b_img[k].alloc(sizeof(double) * Vl, st));
                FVB_HIP_CHECK(upload_rows(sl.b_img[k].p, cfg.image_prior[k], sizeof(double), 1));
                sl.d.image_prior[k] = (const double *)sl.b_img[k].p;
            }
        memset(&sl.dout, 0, sizeof(sl.dout));
        FVB_HIP_CHECK(sl.b_mvn.alloc(sizeof(double) * rows * Vl, st));
        sl.dout.mvn = (double *)sl.b_mvn.p;
        if (want_f)
        {
            FVB_HIP_CHECK(sl.b_f.alloc(sizeof(double) * Vl, st));
            sl.dout.free_energy = (double *)sl.b_f.p;
        }
        if (want_status)
        {
            FVB_HIP_CHECK(sl.b_status.alloc(sizeof(int32_t) * Vl, st));
            sl.dout.status = (int32_t *)sl.b_status.p;
        }
        if (want_it)
        {
            FVB_HIP_CHECK(sl.b_it.alloc(sizeof(int32_t) * Vl, st));
            sl.dout.iterations = (int32_t *)sl.b_it.p;
        }
        FVB_HIP_CHECK(sl.stage_means.alloc(sizeof(double) * (size_t)P * max_halo, st));
        FVB_HIP_CHECK(sl.stage_status.alloc(sizeof(int32_t) * (size_t)max_halo, st));
        FVB_HIP_CHECK(hipStreamSynchronize(st)); // (the uploads read pageable host memory)
        sl.coords.resize(3 * Vl);
        for (int dim = 0; dim < 3; dim++)
            std::copy(coords.begin() + (size_t)dim * V + sl.g0, coords.begin() + (size_t)dim * V + sl.g1, sl.coords.begin() + (size_t)dim * Vl);
        sl.sp = sp;
        sl.sp.coords = sl.coords.data();
        sl.sp.owned_begin = sl.b - sl.g0;
        sl.sp.owned_end = sl.e - sl.g0;
        sl.sp.n_voxels_global = V;
        return 0;
    };
    return for_each_slab(upload_slab);
}

// ---- a complete run on the resident data: set-up, the iterations, the packed result images (on the devices) ----
int fvb_spatial_multi::run(void (*progress_cb)(int, int))
{
    int rc = execute(progress_cb, false);
    s_unlink_pair = -1; // (the test hook holds for one attempt)
    if (rc == 1) // the slabs could not sweep together (or gave that up): the level-chunk pipeline, the exact form
        rc = execute(progress_cb, true);
    return rc;
}

// Returns 1 where the run has to be repeated as the level-chunk pipeline.
int fvb_spatial_multi::execute(void (*progress_cb)(int, int), bool no_fast)
{
    int rc;
    const int32_t *X = coords.data(), *Y = X + V, *Z = X + 2 * (size_t)V;
    const bool timing = getenv("FVB_SPATIAL_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    // all slabs sweep together with the slab form of the split sweep (vb_spatial.h) unless that was tried and abandoned,
    // the devices cannot reach each other's memory, or FVB_SPATIAL_PER_LEVEL asks for the level-chunk pipeline
    const bool try_fast = has_spatial && peers && !no_fast && !getenv("FVB_SPATIAL_PER_LEVEL") && !getenv("FVB_SPATIAL_MULTI_PIPELINE");
    // ---- per slab: a run handle (neighbour table, numbering, set-up) ----
    auto open_slab = [&](int r) -> int {
        SlabRun &sl = *slabs[r];
        FVB_HIP_CHECK(hipSetDevice(sl.dev));
        delete sl.run;
        sl.run = nullptr;
        if (sl.dout.free_energy)
            FVB_HIP_CHECK(hipMemsetAsync(sl.b_f.p, 0xff, sizeof(double) * (size_t)(sl.g1 - sl.g0), sl.stream)); // NaN (see fabber_vb_run_spatial_host)
        sl.run = new fvb_spatial_run;
        sl.run->allow_fast = sl.run->multi_fast = try_fast;
        sl.run->device_share = (int)std::count(devs.begin(), devs.begin() + world, sl.dev);
        return sl.run->open(&sl.d, &sl.sp, sl.b_data.p, &sl.dout, sl.stream);
    };
    if ((rc = for_each_slab(open_slab)) != 0)
        return rc;
    ms_setup = since(t_begin);
    const auto t_loop = std::chrono::steady_clock::now();
    // ---- all slabs sweep together: every slab's top plane writes into the inboxes of the slab above ----
    bool all_fast = try_fast;
    for (int r = 0; r < world; r++)
    {
        all_fast = all_fast && slabs[r]->run->slab_form;
        // (an inbox another DEVICE writes into must be fine-grained memory, or its stores may stay invisible to the polls)
        if (r > 0 && slabs[r]->dev != slabs[r - 1]->dev)
            all_fast = all_fast && slabs[r]->run->gran_fine;
    }
    if (try_fast && !all_fast) // (a slab the slab form does not take: the level-chunk pipeline for the whole run)
        return 1;
    if (all_fast)
    {
        for (int r = 0; r + 1 < world; r++)
        {
            FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
            if (slabs[r]->dev != slabs[r + 1]->dev)
            {
                hipError_t e = hipDeviceEnablePeerAccess(slabs[r + 1]->dev, 0);
                (void)hipGetLastError();
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                {
                    // (hipDeviceCanAccessPeer said yes: all the same, the pipeline needs no peer mapping)
                    if (getenv("FVB_SPATIAL_VERBOSE"))
                        fprintf(stderr, "[fvb spatial] hipDeviceEnablePeerAccess(%d -> %d): %s - level-chunk pipeline\n", slabs[r]->dev, slabs[r + 1]->dev, hipGetErrorString(e));
                    return 1;
                }
            }
            if (r == s_unlink_pair) // (test hook: the slab above never hears from this one)
                continue;
            if ((rc = slabs[r]->run->link_up(*slabs[r + 1]->run, slabs[r]->g0, slabs[r + 1]->g0)) != 0)
                return rc;
        }
        std::vector<double> partials_f, sums_f((size_t)P * 2);
        for (int it = 0; it < cfg.max_iterations; it++)
        {
            if (progress_cb)
                progress_cb(it, cfg.max_iterations); // inference_vb.cc:610
            if (has_spatial && (it > 0 || sp.update_first_iter))
            {
                std::fill(sums_f.begin(), sums_f.end(), 0.0);
                for (int r = 0; r < world; r++)
                {
                    SlabRun &sl = *slabs[r];
                    FVB_HIP_CHECK(hipSetDevice(sl.dev));
                    partials_f.assign((size_t)std::max(sl.run->n_segments, 1) * P * 2, 0.0);
                    if ((rc = sl.run->ak_segment_sums(partials_f.data())) != 0)
                        return rc;
                    for (int seg = 0; seg < sl.run->n_segments; seg++)
                        for (int j = 0; j < 2 * P; j++)
                            sums_f[j] = sums_f[j] + partials_f[(size_t)seg * 2 * P + j];
                }
                for (int r = 0; r < world; r++)
                {
                    FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                    if ((rc = slabs[r]->run->set_ak_sums(sums_f.data())) != 0)
                        return rc;
                }
            }
            // records of every slab, then ALL slabs' ordered sweeps at once (slab r + 1's lowest plane waits, voxel by
            // voxel, for what slab r's highest plane puts into its inboxes), then the second sweep
            for (int r = 0; r < world; r++)
            {
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = slabs[r]->run->fast_prep(it)) != 0)
                    return rc;
            }
            for (int r = 0; r < world; r++)
            {
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = slabs[r]->run->fast_sweep()) != 0)
                    return rc;
            }
            if (cfg.need_f) // the F term of the priors of the LAST voxel of the sweep is the last slab's
            {
                double fp = 0;
                FVB_HIP_CHECK(hipSetDevice(slabs[world - 1]->dev));
                if ((rc = fabber_vb_spatial_fprior(slabs[world - 1]->run, &fp, 0)) != 0)
                    return rc;
                for (int r = 0; r + 1 < world; r++)
                {
                    FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                    if ((rc = fabber_vb_spatial_fprior(slabs[r]->run, &fp, 1)) != 0)
                        return rc;
                }
            }
            for (int r = 0; r < world; r++)
            {
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = slabs[r]->run->fast_noise(it)) != 0)
                    return rc;
            }
            if (has_spatial)
                for (int r = 0; r + 1 < world; r++) // boundary planes both ways (the ghosts' means for the next iteration)
                {
                    SlabRun &lo = *slabs[r], &hi = *slabs[r + 1];
                    const int up_from = std::max(lo.b, hi.g0);
                    if ((rc = slab_transfer(lo, up_from - lo.g0, hi, up_from - hi.g0, lo.e - up_from, P)) != 0)
                        return rc;
                    const int down_to = std::min(hi.e, lo.g1);
                    if ((rc = slab_transfer(hi, hi.b - hi.g0, lo, hi.b - lo.g0, down_to - hi.b, P)) != 0)
                        return rc;
                }
        }
        bool any_failed = false;
        for (int r = 0; r < world; r++)
        {
            bool failed = false;
            FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
            if ((rc = slabs[r]->run->fast_failed(failed)) != 0)
                return rc;
            any_failed |= failed;
        }
        if (any_failed)
        {
            // a voxel failed during a first sweep (or an inbox never arrived): as on one device, the run is repeated
            // with the launches the split sweep does not need - here the level-chunk pipeline
            if (getenv("FVB_SPATIAL_VERBOSE"))
                fprintf(stderr, "[fvb spatial] slab sweep across devices abandoned, repeating the run with the level-chunk pipeline\n");
            return 1;
        }
    }
    // ---- (otherwise) the global level range and the pipeline's ticks ----
    const long long w0 = slabs[0]->run->level_w[0], w1 = slabs[0]->run->level_w[1], w2 = slabs[0]->run->level_w[2];
    long long lmin = LLONG_MAX, lmax = LLONG_MIN;
    for (int v = 0; v < V; v++)
    {
        const long long l = w0 * X[v] + w1 * Y[v] + w2 * Z[v];
        lmin = std::min(lmin, l);
        lmax = std::max(lmax, l);
    }
    long long chunk_levels = 16;
    if (const char *forced = getenv("FVB_SPATIAL_CHUNK_LEVELS")) // tests: other cuts of the level range
        chunk_levels = std::max(1, atoi(forced));
    const long long nchunks = std::max(1LL, (lmax - lmin + chunk_levels) / chunk_levels);
    std::vector<double> partials, sums((size_t)P * 2);
    for (int it = 0; it < (all_fast ? 0 : cfg.max_iterations); it++)
    {
        if (progress_cb)
            progress_cb(it, cfg.max_iterations); // inference_vb.cc:610
        if (has_spatial && (it > 0 || sp.update_first_iter))
        {
            // a_K: every slab's segment sums, added in the order of the voxel list (vb_spatial_ak_reduce_kernel's)
            std::fill(sums.begin(), sums.end(), 0.0);
            for (int r = 0; r < world; r++)
            {
                SlabRun &sl = *slabs[r];
                FVB_HIP_CHECK(hipSetDevice(sl.dev));
                partials.assign((size_t)std::max(sl.run->n_segments, 1) * P * 2, 0.0);
                if ((rc = sl.run->ak_segment_sums(partials.data())) != 0)
                    return rc;
                for (int seg = 0; seg < sl.run->n_segments; seg++)
                    for (int j = 0; j < 2 * P; j++)
                        sums[j] = sums[j] + partials[(size_t)seg * 2 * P + j];
            }
            for (int r = 0; r < world; r++)
            {
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = slabs[r]->run->set_ak_sums(sums.data())) != 0)
                    return rc;
            }
        }
        for (long long tick = 0; tick < nchunks + world - 1; tick++)
        {
            for (int r = 0; r < world; r++)
            {
                const long long c = tick - r;
                if (c < 0 || c >= nchunks)
                    continue;
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = slabs[r]->run->sweep_levels(it, lmin + c * chunk_levels, lmin + (c + 1) * chunk_levels)) != 0)
                    return rc;
            }
            if (!has_spatial)
                continue;
            for (int r = 0; r + 1 < world; r++) // whoever swept hands its top planes up
            {
                const long long c = tick - r;
                if (c < 0 || c >= nchunks)
                    continue;
                SlabRun &lo = *slabs[r], &hi = *slabs[r + 1];
                const int from = std::max(lo.b, hi.g0);
                if ((rc = slab_transfer(lo, from - lo.g0, hi, from - hi.g0, lo.e - from, P)) != 0)
                    return rc;
            }
        }
        if (cfg.need_f) // the F term of the priors of the LAST voxel of the sweep is the last slab's (inference_vb.cc:612,689,702)
        {
            double fp = 0;
            FVB_HIP_CHECK(hipSetDevice(slabs[world - 1]->dev));
            if ((rc = fabber_vb_spatial_fprior(slabs[world - 1]->run, &fp, 0)) != 0)
                return rc;
            for (int r = 0; r + 1 < world; r++)
            {
                FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
                if ((rc = fabber_vb_spatial_fprior(slabs[r]->run, &fp, 1)) != 0)
                    return rc;
            }
        }
        for (int r = 0; r < world; r++)
        {
            FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
            if ((rc = slabs[r]->run->sweep_noise(it)) != 0)
                return rc;
        }
        if (has_spatial)
            for (int r = 0; r + 1 < world; r++) // boundary planes both ways
            {
                SlabRun &lo = *slabs[r], &hi = *slabs[r + 1];
                const int up_from = std::max(lo.b, hi.g0);
                if ((rc = slab_transfer(lo, up_from - lo.g0, hi, up_from - hi.g0, lo.e - up_from, P)) != 0)
                    return rc;
                const int down_to = std::min(hi.e, lo.g1);
                if ((rc = slab_transfer(hi, hi.b - hi.g0, lo, hi.b - lo.g0, down_to - hi.b, P)) != 0)
                    return rc;
            }
    }
    // ---- every slab packs its voxels' results (on its device) ----
    for (int r = 0; r < world; r++)
    {
        FVB_HIP_CHECK(hipSetDevice(slabs[r]->dev));
        if ((rc = slabs[r]->run->finish()) != 0) // (ends with a wait for the slab's stream)
            return rc;
    }
    ms_loop = since(t_loop);
    route = all_fast ? "all slabs sweep together" : "level-chunk pipeline";
    if (timing)
        fprintf(stderr, "[fvb spatial] %d slabs (%s): geometry + set-up %.1f ms, %d iterations + result images %.1f ms\n", world, route, ms_setup,
            cfg.max_iterations, ms_loop);
    return 0;
}

// ---- results: the owned voxels of every slab go to the caller's images ----
int fvb_spatial_multi::download(const fvb_outputs *out)
{
    for (int r = 0; r < world; r++)
    {
        SlabRun &sl = *slabs[r];
        FVB_HIP_CHECK(hipSetDevice(sl.dev));
        const size_t Vl = (size_t)(sl.g1 - sl.g0), own = (size_t)(sl.e - sl.b), skip = (size_t)(sl.b - sl.g0);
        auto download_rows = [&](void *dst, const void *src, size_t elem, size_t nrows) {
            return copy_rows((char *)dst + (size_t)sl.b * elem, (size_t)V * elem, (const char *)src + skip * elem, Vl * elem, own * elem, nrows,
                hipMemcpyDeviceToHost, sl.stream);
        };
        FVB_HIP_CHECK(download_rows(out->mvn, sl.dout.mvn, sizeof(double), (size_t)rows));
        if (sl.dout.free_energy && out->free_energy)
            FVB_HIP_CHECK(download_rows(out->free_energy, sl.dout.free_energy, sizeof(double), 1));
        if (sl.dout.status && out->status)
            FVB_HIP_CHECK(download_rows(out->status, sl.dout.status, sizeof(int32_t), 1));
        if (sl.dout.iterations && out->iterations)
            FVB_HIP_CHECK(download_rows(out->iterations, sl.dout.iterations, sizeof(int32_t), 1));
        FVB_HIP_CHECK(hipStreamSynchronize(sl.stream));
    }
    return 0;
}

static int32_t spatial_multi_validate(const fvb_config *cfg, const fvb_spatial *sp, const fvb_outputs *out)
{
    int rc = api_validate(cfg, true);
    if (rc)
        return rc;
    if (!sp || !sp->coords)
        return api_fail(-42, "spatial description / coordinates missing");
    if (sp->spatial_dims < 0 || sp->spatial_dims > 3)
        return api_fail(-43, "spatial-dims must be 0, 1, 2 or 3");
    if (spatial_noise_kind(cfg) < 0)
        return api_fail(-44, spatial_noise_refusal);
    if (cfg->model == FVB_MODEL_HOSTJAC)
        return api_fail(-56, "a model evaluated on the host runs spatial VB on one device (fabber_vb_run_spatial_hostmodel_host)");
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    return 0;
}

int32_t fabber_vb_spatial_multi_open(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *wanted,
    const int32_t *devices, int32_t n_devices, fvb_spatial_multi **handle)
{
    if (!handle)
        return api_fail(-47, "handle pointer is NULL");
    *handle = nullptr;
    int rc = spatial_multi_validate(cfg, sp, wanted);
    if (rc)
        return rc;
    if (cfg->n_voxels == 0 || !data)
        return api_fail(-21, "no voxels / data is NULL");
    if (sp->locked_centres)
        return api_fail(-57, "locked linearisation centres run on one device (fabber_vb_run_spatial_host)");
    std::unique_ptr<fvb_spatial_multi> m(new fvb_spatial_multi);
    if ((rc = m->plan(cfg, sp, devices, n_devices)) != 0)
        return rc;
    if (m->world <= 1)
        return api_fail(-58, "the volume has too few planes for two slabs: fabber_vb_run_spatial_host / _device");
    if ((rc = m->upload(data, wanted)) != 0)
        return rc;
    *handle = m.release();
    return 0;
}

int32_t fabber_vb_spatial_multi_run(fvb_spatial_multi *handle, void (*progress_cb)(int, int))
{
    return handle ? handle->run(progress_cb) : api_fail(-47, "handle is NULL");
}

int32_t fabber_vb_spatial_multi_results(fvb_spatial_multi *handle, const fvb_outputs *out)
{
    if (!handle || !out || !out->mvn)
        return api_fail(-47, "handle or outputs.mvn is NULL");
    return handle->download(out);
}

int32_t fabber_vb_spatial_multi_slabs(fvb_spatial_multi *handle, int32_t *n_slabs, char *route, int32_t route_len)
{
    if (!handle)
        return api_fail(-47, "handle is NULL");
    if (n_slabs)
        *n_slabs = handle->world;
    if (route && route_len > 0)
        snprintf(route, (size_t)route_len, "%s", handle->route);
    return 0;
}

int32_t fabber_vb_spatial_multi_close(fvb_spatial_multi *handle)
{
    delete handle;
    return 0;
}

int32_t fabber_vb_run_spatial_host_multi(const fvb_config *cfg, const fvb_spatial *sp, const void *data, const fvb_outputs *out,
    const int32_t *devices, int32_t n_devices, void (*progress_cb)(int, int))
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    int rc = spatial_multi_validate(cfg, sp, out);
    if (rc)
        return rc;
    if (cfg->n_voxels == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    std::unique_ptr<fvb_spatial_multi> m(new fvb_spatial_multi);
    if ((rc = m->plan(cfg, sp, devices, n_devices)) != 0)
        return rc;
    // (locked centres: rare, and nothing a second device would speed up; too few planes for two slabs: the one-device run)
    if (sp->locked_centres || m->world <= 1)
        return fabber_vb_run_spatial_host(cfg, sp, data, out, m->devs[0], progress_cb);
    if ((rc = m->upload(data, out)) != 0)
        return rc;
    const double ms_up = since(t0);
    if ((rc = m->run(progress_cb)) != 0)
        return rc;
    const auto t_out = std::chrono::steady_clock::now();
    if ((rc = m->download(out)) != 0)
        return rc;
    const double ms_out = since(t_out);
    const auto t_free = std::chrono::steady_clock::now();
    m.reset();
    if (getenv("FVB_SPATIAL_TIMING"))
        fprintf(stderr, "[fvb spatial] fabber_vb_run_spatial_host_multi: upload %.1f ms, results %.1f ms, giving the slabs' memory back %.1f ms, %.1f ms in all\n",
            ms_up, ms_out, since(t_free), since(t0));
    return 0;
}

} // extern "C"

// Host-only helper (no GPU needed): the first-neighbour table the spatial driver builds,
// [n_voxels][6], 0-based ids, -1 = none. For the unit tests of the neighbour construction.
extern "C" int32_t fabber_vb_neighbours(const int32_t *coords, int32_t n_voxels, int32_t spatial_dims, int32_t *nn_out)
{
    std::vector<int32_t> nn;
    std::string err = build_neighbours(coords, n_voxels, spatial_dims, nn);
    if (!err.empty())
        return fvb::api_fail(-41, err);
    std::copy(nn.begin(), nn.end(), nn_out);
    return 0;
}
