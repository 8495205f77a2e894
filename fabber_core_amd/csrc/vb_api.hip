/*
 * vb_api.hip - C ABI of the voxelwise VB engine (include/fabber_vb.h): validation, kernel
 * selection, launch, and the host-pointer convenience entry points.
 *
 * There is no CPU fallback: if no HIP device is present or a launch fails the call returns a
 * negative code and fabber_vb_last_error() says why.
 */
#include "vb_dispatch.h"
#include "vb_host_copy.h"
#include "vb_wave_kernel.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace fvb;

namespace
{
thread_local std::string g_last_error;
int g_variant = 0; // 0 auto, 1 lane, 2 wave
int g_residual_mode = 0; // 0 adaptive, 1 always exact, 2 moments only
// linearisations of a run that evaluate the model pointwise (vb_lane_kernel.h); the environment variable is an experiment switch
int g_precise_passes = getenv("FVB_PRECISE_PASSES") ? atoi(getenv("FVB_PRECISE_PASSES")) : -1; // -1: by model, see run_device_as
bool g_tiled = getenv("FVB_NO_TILES") == nullptr; // experiment switch: FVB_NO_TILES=1 keeps the in-place (strided) feed
double g_residual_tol = 1e-10; // moments value keeps >= 6 significant digits where it is used

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define FVB_HIP_CHECK(expr)                                                                                  \
    do                                                                                                       \
    {                                                                                                        \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(-100 - (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

int noise_outputs(const fvb_config *cfg)
{
    // AR(1): (alphas, phi means), noisemodel_ar.cc:287-300
    return cfg->noise == FVB_NOISE_WHITE ? cfg->n_phis : 2 + cfg->ar_cross_terms + cfg->n_phis;
}

int validate(const fvb_config *cfg, bool allow_spatial = false, bool allow_no_noise = false)
{
    if (!cfg)
        return fail(-1, "config is NULL");
    if (cfg->abi_version != FVB_ABI_VERSION)
        return fail(-2, "fvb_config.abi_version mismatch");
    if (cfg->n_voxels < 0 || cfg->n_times <= 0)
        return fail(-3, "bad n_voxels / n_times");
    if (cfg->n_params <= 0 || cfg->n_params > (cfg->params_ext ? FVB_MAX_PARAMS_EXT : FVB_MAX_PARAMS))
        return fail(-4, cfg->params_ext ? "n_params out of range" : "n_params out of range (more than FVB_MAX_PARAMS parameters: fvb_config.params_ext)");
    // (allow_no_noise: the post-processing of a result image, which such problems have like any other)
    if (cfg->params_ext && allow_spatial && !allow_no_noise)
        return fail(-4, "a parameter table (more than FVB_MAX_PARAMS parameters) runs voxelwise VB and method=nlls only");
    // (a result image without noise entries - method=nlls - can only be post-processed)
    if ((cfg->n_phis <= 0 && !(allow_no_noise && cfg->n_phis == 0)) || cfg->n_phis > FVB_MAX_PHIS)
        return fail(-5, "n_phis out of range");
    if (cfg->noise != FVB_NOISE_WHITE && cfg->noise != FVB_NOISE_AR1)
        return fail(-6, "noise model not supported by this build");
    if (cfg->noise == FVB_NOISE_AR1) // noisemodel_ar.cc:322-349
    {
        if (cfg->n_phis != 1 && cfg->n_phis != 2)
            return fail(-6, "AR(1) noise: num-echoes must be 1 or 2");
        if (cfg->ar_cross_terms < 0 || cfg->ar_cross_terms > 2)
            return fail(-6, "AR(1) noise: unknown ar1-cross-terms");
        if (cfg->n_phis == 1 && cfg->ar_cross_terms != 0)
            return fail(-6, "AR(1) noise: you must use ar1-cross-terms=none with num-echoes=1");
        if (cfg->n_times % cfg->n_phis != 0)
            return fail(-6, "AR(1) noise: the number of timepoints is not a multiple of num-echoes");
    }
    if (cfg->convergence < FVB_CONV_MAXITS || cfg->convergence > FVB_CONV_LM)
        return fail(-7, "unknown convergence detector");
    if (cfg->max_iterations <= 0)
        return fail(-8, "max_iterations must be positive");
    if (cfg->convergence != FVB_CONV_MAXITS && !cfg->need_f)
        return fail(-9, "convergence detector uses F but need_f is 0");
    if (cfg->model == FVB_MODEL_LINEAR && !cfg->design)
        return fail(-10, "linear model needs a design matrix");
    if (cfg->model == FVB_MODEL_EXP && (cfg->n_params != 2 * cfg->model_iopt[0]))
        return fail(-11, "exp model: n_params != 2 * num-exps");
    if (cfg->model == FVB_MODEL_POLY && (cfg->n_params != cfg->model_iopt[0] + 1))
        return fail(-12, "poly model: n_params != degree + 1");
    for (int k = 0; k < cfg->n_params; k++)
    {
        // (validate() sees the configuration as the caller built it: with device entry points the table itself is device
        // memory and is not looked into here)
        if (cfg->params_ext)
            break;
        if (cfg->prior_type[k] == FVB_PRIOR_IMAGE && !cfg->image_prior[k])
            return fail(-13, "image prior without an image");
        if (cfg->prior_type[k] > FVB_PRIOR_ARD && !allow_spatial)
            return fail(-14, "spatial priors are not handled by the voxelwise engine (use fabber_vb_run_spatial_*)");
        if (cfg->prior_type[k] < 0 || cfg->prior_type[k] > FVB_PRIOR_SPATIAL_p)
            return fail(-14, "unknown prior type");
    }
    return 0;
}

// Below this many voxels the lane kernel (V/64 wavefronts, each taking the full per-voxel latency)
// leaves most of the 1024 SIMDs empty and the wave-per-voxel kernel (V wavefronts) finishes
// first. Measured crossovers on MI355X (profiles/r1_lane_vs_wave.jsonl): ~6k voxels for the
// bi-exponential C3 fit (lane 2.2 ms flat, wave 0.4 us per voxel), ~2.5k for the single-exponential
// C2 fit (lane 0.12 ms flat, wave 0.04 us per voxel).
constexpr int WAVE_KERNEL_BELOW_VOXELS = 4096;

LaneKernelInfo select_lane(const fvb_config *cfg)
{
    if (g_variant == 2 || (cfg->n_phis != 1 && cfg->noise == FVB_NOISE_WHITE && cfg->n_phis > 4))
        return LaneKernelInfo{ nullptr, 0, nullptr };
    if (cfg->noise == FVB_NOISE_AR1 && cfg->n_phis == 2) // two echoes: the two-pass kernel of vb_lane_arn_kernel.h
    {
        if (cfg->n_times % 2 != 0 || cfg->n_times < 8 || (g_variant == 0 && cfg->n_voxels < WAVE_KERNEL_BELOW_VOXELS))
            return LaneKernelInfo{ nullptr, 0, nullptr };
        const int n_alphas = 2 + cfg->ar_cross_terms;
        const bool f = cfg->need_f != 0;
        switch (cfg->model)
        {
        case FVB_MODEL_POLY:
            return get_lane_arn_kernel_poly(cfg->n_params, n_alphas, f);
        case FVB_MODEL_LINEAR:
            return get_lane_arn_kernel_linear(cfg->n_params, n_alphas, f);
        case FVB_MODEL_EXP:
            return get_lane_arn_kernel_exp(cfg->n_params, n_alphas, f);
        default:
            return LaneKernelInfo{ nullptr, 0, nullptr };
        }
    }
    if (cfg->n_phis != 1 && cfg->noise != FVB_NOISE_WHITE)
        return LaneKernelInfo{ nullptr, 0, nullptr };
    // ... for models whose re-linearisation is long enough to be worth sharing out over 64 lanes: T (2P + 1) model
    // evaluations per pass, an exponential counted twice. Below ~400 of them (C1: a quadratic over 10 timepoints = 70)
    // the wave kernel's per-iteration synchronisation outweighs what it shares, whatever the voxel count: 512 voxels
    // of C1 take 0.058 ms on the lane kernel against 0.105 ms.
    const double evaluations = (double)cfg->n_times * (2 * cfg->n_params + 1) * (cfg->model == FVB_MODEL_EXP ? 2.0 : 1.0);
    if (g_variant == 0 && cfg->noise == FVB_NOISE_WHITE && cfg->n_voxels < WAVE_KERNEL_BELOW_VOXELS && evaluations >= 400
        && wave_layout(cfg->n_times, cfg->n_params, cfg->n_phis).bytes <= 160 * 1024)
        return LaneKernelInfo{ nullptr, 0, nullptr };
    const bool need_f = cfg->need_f != 0;
    if (cfg->noise == FVB_NOISE_WHITE && cfg->n_phis > 1) // noise-pattern: 2 .. 4 precisions
    {
        if (cfg->n_times > 32768) // (the kernel keeps the class of every timepoint in LDS, one byte each)
            return LaneKernelInfo{ nullptr, 0, nullptr };
        const bool two = cfg->n_phis == 2;
        switch (cfg->model)
        {
        case FVB_MODEL_POLY:
            return two ? get_lane_pattern_kernel_poly_2(cfg->n_params) : get_lane_pattern_kernel_poly_4(cfg->n_params);
        case FVB_MODEL_LINEAR:
            return two ? get_lane_pattern_kernel_linear_2(cfg->n_params) : get_lane_pattern_kernel_linear_4(cfg->n_params);
        case FVB_MODEL_EXP:
            return two ? get_lane_pattern_kernel_exp_2(cfg->n_params) : get_lane_pattern_kernel_exp_4(cfg->n_params);
        default:
            return LaneKernelInfo{ nullptr, 0, nullptr };
        }
    }
    if (cfg->noise == FVB_NOISE_AR1)
    {
        switch (cfg->model)
        {
        case FVB_MODEL_POLY:
            return get_lane_ar_kernel_poly(cfg->n_params, need_f);
        case FVB_MODEL_LINEAR:
            return get_lane_ar_kernel_linear(cfg->n_params, need_f);
        case FVB_MODEL_EXP:
            return get_lane_ar_kernel_exp(cfg->n_params, need_f);
        default:
            return LaneKernelInfo{ nullptr, 0, nullptr };
        }
    }
    switch (cfg->model)
    {
    case FVB_MODEL_POLY:
        return get_lane_kernel_poly(cfg->n_params, need_f);
    case FVB_MODEL_LINEAR:
        return get_lane_kernel_linear(cfg->n_params, need_f);
    case FVB_MODEL_EXP:
        return get_lane_kernel_exp(cfg->n_params, need_f);
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}

bool needs_save(const fvb_config *cfg)
{
    return cfg->convergence == FVB_CONV_FREDUCE || cfg->convergence == FVB_CONV_TRIALMODE
        || cfg->convergence == FVB_CONV_LM;
}

int count_unmasked(const fvb_config *cfg, const uint8_t *phi_index_host)
{
    if (!phi_index_host)
        return cfg->n_times;
    int n = 0;
    for (int t = 0; t < cfg->n_times; t++)
        n += (phi_index_host[t] != 255);
    return n;
}

// ---- post-processing kernel: InferenceTechnique::SaveResults / Vb::SaveResults ----------------
template <int MAXP>
__global__ __launch_bounds__(256) void vb_postproc_kernel(
    const fvb_config cfg, const void *data, const double *mvn, const fvb_postproc pp, const int n_noise)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= cfg.n_voxels)
        return;
    const size_t V = (size_t)cfg.n_voxels;
    const int P = cfg.n_params, n = P + n_noise, T = cfg.n_times;
    const int nCov = n * (n + 1) / 2;
    double means[MAXP];
    for (int p = 0; p < P; p++)
    {
        const double m = mvn[(size_t)(nCov + p) * V + v];
        const double var = mvn[(size_t)(p * (p + 1) / 2 + p) * V + v];
        const int tr = cfg.params_ext ? cfg.params_ext->transform[p] : cfg.transform[p];
        // FwdModel::ToModel, fwdmodel.cc:326-337
        const double mm = to_model(tr, m);
        const double mv = to_model_var(tr, var);
        const double sd = sqrt(mv);
        means[p] = mm; // model-space value, what EvaluateModel receives
        if (pp.mean)
            pp.mean[(size_t)p * V + v] = mm;
        if (pp.var)
            pp.var[(size_t)p * V + v] = mv;
        if (pp.std)
            pp.std[(size_t)p * V + v] = sd;
        if (pp.zstat)
            pp.zstat[(size_t)p * V + v] = mm / sd;
    }
    for (int i = 0; i < n_noise; i++) // inference_vb.cc:981-989
    {
        const int q = P + i;
        if (pp.noise_mean)
            pp.noise_mean[(size_t)i * V + v] = mvn[(size_t)(nCov + q) * V + v];
        if (pp.noise_std)
            pp.noise_std[(size_t)i * V + v] = sqrt(mvn[(size_t)(q * (q + 1) / 2 + q) * V + v]);
    }
    if (pp.modelfit || pp.residuals) // inference.cc:181-243
    {
        ModelArgs ma;
        ma.iopt0 = cfg.model_iopt[0];
        ma.dopt0 = cfg.model_dopt[0];
        ma.design = cfg.design;
        for (int t = 0; t < T; t++)
        {
            const double fit = eval_model_runtime(cfg.model, ma, P, t, means);
            if (pp.modelfit)
                pp.modelfit[(size_t)t * V + v] = fit;
            if (pp.residuals)
            {
                const size_t idx = (size_t)t * V + v;
                const double y = cfg.data_f64 ? ((const double *)data)[idx] : (double)((const float *)data)[idx];
                pp.residuals[idx] = y - fit;
            }
        }
    }
}

} // namespace
namespace fvb
{
void api_keep_pool_memory();
hipError_t api_pool_alloc(void **p, size_t bytes, hipStream_t stream);
hipError_t api_pool_free(void *p, hipStream_t stream);
}
namespace
{
// RAII device buffer used only by the *_host entry points: from the device's stream-ordered pool, which keeps the
// memory between calls (hipMalloc + hipFree of the series-sized buffers were ~20 of the 50 ms a call on 1e6 voxels
// of the bi-exponential configuration took through host pointers)
// Device memory of ONE block of the pipelined host entry point: plain hipMalloc'd buffers that stay with the call's cached
// streams (PipeStreams) and are handed out again - to the block that takes the slot three blocks later, and to the next
// call. No stream-ordered pool here: upload, fit and download streams and two host threads work on a call, and with ROCm
// 7.2's runtime buffers taken from ONE pool by several streams came out overlapping - wrong results from the pipelined
// call, right ones with plain hipMalloc in the same code (tools/measure/runtime_check.py, runtime_check_capi.py); ROCm 7.0's
// runtime did not show it. A slot's buffers are reused only after the block that had them has been SEEN to finish.
struct BlockSlot
{
    struct Buf
    {
        void *p;
        size_t cap;
        bool used;
    };
    std::vector<Buf> bufs;
    hipError_t take(void **out, size_t bytes)
    {
        int best = -1;
        for (size_t i = 0; i < bufs.size(); i++)
            if (!bufs[i].used && bufs[i].cap >= bytes && (best < 0 || bufs[i].cap < bufs[(size_t)best].cap))
                best = (int)i;
        if (best < 0)
        {
            const size_t cap = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
            void *p = nullptr;
            const hipError_t e = hipMalloc(&p, cap);
            if (e != hipSuccess)
                return e;
            bufs.push_back(Buf{ p, cap, false });
            best = (int)bufs.size() - 1;
        }
        bufs[(size_t)best].used = true;
        *out = bufs[(size_t)best].p;
        return hipSuccess;
    }
    void reset()
    {
        for (Buf &b : bufs)
            b.used = false;
    }
    void destroy()
    {
        for (Buf &b : bufs)
            (void)hipFree(b.p);
        bufs.clear();
    }
};

struct DevBuf
{
    void *p = nullptr;
    hipStream_t stream = nullptr;
    BlockSlot *slot = nullptr; // the memory is the slot's (nothing to free here)
    ~DevBuf()
    {
        if (p && !slot)
            (void)fvb::api_pool_free(p, stream);
    }
    hipError_t alloc(size_t bytes, hipStream_t s = nullptr)
    {
        stream = s;
        if (slot)
            return slot->take(&p, bytes ? bytes : 8);
        return fvb::api_pool_alloc(&p, bytes ? bytes : 8, s);
    }
};

// A host fvb_param_table (fvb_config.params_ext: more than FVB_MAX_PARAMS parameters) on the device: the seven arrays,
// the image priors of the voxels [v0, v0 + Vb) and the table itself
struct DeviceParamTable
{
    DevBuf block;
    std::vector<std::unique_ptr<DevBuf> > images;
    const fvb_param_table *device = nullptr;
    int upload(const fvb_config *cfg, size_t v0, size_t Vb, hipStream_t stream)
    {
        const fvb_param_table *h = cfg->params_ext;
        const size_t P = (size_t)cfg->n_params, V = (size_t)cfg->n_voxels;
        // [table][transform, prior_type: int32 P each][5 double arrays][image pointers]
        const size_t off_i = sizeof(fvb_param_table), off_d = off_i + 2 * P * sizeof(int32_t) + (2 * P * sizeof(int32_t)) % 8;
        const size_t off_p = off_d + 5 * P * sizeof(double), bytes = off_p + P * sizeof(double *);
        std::vector<char> host(bytes, 0);
        FVB_HIP_CHECK(block.alloc(bytes, stream));
        char *dev = (char *)block.p;
        fvb_param_table t;
        t.transform = (const int32_t *)(dev + off_i);
        t.prior_type = t.transform + P;
        t.prior_mean = (const double *)(dev + off_d);
        t.prior_var = t.prior_mean + P;
        t.prior_prec = t.prior_var + P;
        t.post_mean = t.prior_prec + P;
        t.post_var = t.post_mean + P;
        t.image_prior = (const double *const *)(dev + off_p);
        memcpy(host.data(), &t, sizeof(t));
        memcpy(host.data() + off_i, h->transform, P * sizeof(int32_t));
        memcpy(host.data() + off_i + P * sizeof(int32_t), h->prior_type, P * sizeof(int32_t));
        const double *src[5] = { h->prior_mean, h->prior_var, h->prior_prec, h->post_mean, h->post_var };
        for (int a = 0; a < 5; a++)
            memcpy(host.data() + off_d + (size_t)a * P * sizeof(double), src[a], P * sizeof(double));
        const double **img = (const double **)(host.data() + off_p);
        for (size_t k = 0; k < P; k++)
        {
            if (h->prior_type[k] < 0 || h->prior_type[k] > FVB_PRIOR_ARD)
                return fail(-14, "a parameter table takes prior types N, I and ARD");
            if (h->prior_type[k] == FVB_PRIOR_IMAGE && !(h->image_prior && h->image_prior[k]))
                return fail(-13, "image prior without an image");
            if (h->image_prior && h->image_prior[k])
            {
                images.emplace_back(new DevBuf);
                FVB_HIP_CHECK(images.back()->alloc(sizeof(double) * Vb, stream));
                FVB_HIP_CHECK(hipMemcpyAsync(images.back()->p, h->image_prior[k] + v0, sizeof(double) * Vb, hipMemcpyHostToDevice, stream));
                img[k] = (const double *)images.back()->p;
            }
        }
        (void)V;
        FVB_HIP_CHECK(hipMemcpyAsync(block.p, host.data(), bytes, hipMemcpyHostToDevice, stream));
        FVB_HIP_CHECK(hipStreamSynchronize(stream)); // (`host` is a local; the image priors are the caller's pageable memory)
        device = (const fvb_param_table *)block.p;
        return 0;
    }
};

} // namespace

// helpers shared with vb_spatial_api.hip
namespace fvb
{
int api_fail(int code, const std::string &msg)
{
    return fail(code, msg);
}
int api_validate(const fvb_config *cfg, bool allow_spatial)
{
    return validate(cfg, allow_spatial);
}
int api_variant()
{
    return g_variant;
}
int api_residual_mode()
{
    return g_residual_mode;
}
int api_precise_passes()
{
    return g_precise_passes >= 0 ? g_precise_passes : 2; // (see run_device_as)
}
double api_residual_tol()
{
    return g_residual_tol;
}
// Work buffers of the size of the series are allocated by every run; they come from a stream-ordered memory pool
// that keeps freed memory instead of returning it to the driver at the next synchronisation. The pool is the
// LIBRARY'S OWN (one per device, created on first use): the process-wide default pool of an embedding application
// keeps its settings. fabber_vb_release_cached_memory gives everything back, on every device that was used.
struct PoolTable
{
    std::mutex lock;
    std::vector<std::pair<int, hipMemPool_t> > pools; // (device, pool); pool == nullptr: the default pool is used
};
static PoolTable &pool_table()
{
    static PoolTable t;
    return t;
}
// the current device's pool (nullptr: creation is not supported here - plain hipMallocAsync from the default pool)
hipMemPool_t api_pool()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    PoolTable &t = pool_table();
    std::lock_guard<std::mutex> hold(t.lock);
    for (auto &e : t.pools)
        if (e.first == dev)
            return e.second;
    hipMemPool_t pool = nullptr;
    hipMemPoolProps props;
    memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess)
    {
        uint64_t never = UINT64_MAX;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &never);
    }
    else
    {
        (void)hipGetLastError();
        pool = nullptr;
    }
    t.pools.push_back(std::make_pair(dev, pool));
    return pool;
}
hipError_t api_pool_free(void *p, hipStream_t stream)
{
    return hipFreeAsync(p, stream);
}
hipError_t api_pool_alloc(void **p, size_t bytes, hipStream_t stream)
{
    hipMemPool_t pool = api_pool();
    const hipError_t e = pool ? hipMallocFromPoolAsync(p, bytes, pool, stream) : hipMallocAsync(p, bytes, stream);
    if (e != hipSuccess)
        return e;
    // The allocation is waited for: what follows is often a copy from pageable host memory, which the runtime carries out
    // outside the stream's order - and under ROCm 7.2's runtime such a copy into memory the pool had only just mapped
    // arrived incomplete (tools/measure/runtime_check_release.py: wrong results from the first run after the pools had
    // been given back). The callers allocate at the start of a run, when the stream has nothing pending.
    return hipStreamSynchronize(stream);
}
void api_keep_pool_memory()
{
    (void)api_pool();
}
// A non-blocking stream beside the caller's (the spatial run's set-up kernel runs on one while the geometry is worked out):
// kept between runs, because creating and destroying a stream costs 0.5 - 1.2 ms each (tools/measure/host_timeline.sh) -
// of a 20 ms spatial run. Per device; fabber_vb_release_cached_memory destroys the idle ones.
struct SideStreams
{
    std::mutex lock;
    std::vector<std::pair<int, hipStream_t> > idle;
};
static SideStreams &side_streams()
{
    static SideStreams s;
    return s;
}
hipError_t api_take_side_stream(hipStream_t *out, int *device)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess)
        return e;
    *device = dev;
    {
        SideStreams &c = side_streams();
        std::lock_guard<std::mutex> hold(c.lock);
        for (size_t i = 0; i < c.idle.size(); i++)
            if (c.idle[i].first == dev)
            {
                *out = c.idle[i].second;
                c.idle.erase(c.idle.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
// (the stream must be idle: the caller has synchronised it)
void api_return_side_stream(hipStream_t s, int dev)
{
    SideStreams &c = side_streams();
    std::lock_guard<std::mutex> hold(c.lock);
    c.idle.push_back(std::make_pair(dev, s));
}
void destroy_side_streams()
{
    SideStreams &c = side_streams();
    std::lock_guard<std::mutex> hold(c.lock);
    int now = 0;
    (void)hipGetDevice(&now);
    for (auto &e : c.idle)
        if (hipSetDevice(e.first) == hipSuccess)
            (void)hipStreamDestroy(e.second);
    c.idle.clear();
    (void)hipSetDevice(now);
}

// keep_bytes: what every pool may keep for the next run (0 = give everything back). A pool that holds more is DESTROYED
// (the next allocation makes a new one) rather than trimmed: under ROCm 7.2's runtime the first run after
// hipMemPoolTrimTo(pool, 0) came back with wrong results - every voxel of a one-stream, one-block call
// (tools/measure/runtime_check_release.py; ROCm 7.0's runtime did not show it) - and a destroyed pool has no such state.
void api_release_pools(uint64_t keep_bytes)
{
    PoolTable &t = pool_table();
    std::lock_guard<std::mutex> hold(t.lock);
    int before = 0;
    (void)hipGetDevice(&before);
    for (size_t i = 0; i < t.pools.size();)
    {
        auto &e = t.pools[i];
        if (hipSetDevice(e.first) != hipSuccess)
        {
            i++;
            continue;
        }
        hipMemPool_t pool = e.second;
        const bool own = pool != nullptr;
        if (!own && hipDeviceGetDefaultMemPool(&pool, e.first) != hipSuccess)
        {
            i++;
            continue;
        }
        // (nothing to give back: no need to wait for the device - fabber_destroy calls this when the last handle goes)
        uint64_t reserved = UINT64_MAX;
        if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &reserved) == hipSuccess && reserved <= keep_bytes)
        {
            i++;
            continue;
        }
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        if (own)
        {
            (void)hipMemPoolDestroy(pool);
            t.pools.erase(t.pools.begin() + (long)i); // (api_pool() creates the next one)
            continue;
        }
        (void)hipMemPoolTrimTo(pool, (size_t)keep_bytes); // (the device's default pool, where no pool of our own could be made)
        i++;
    }
    (void)hipSetDevice(before);
}
} // namespace fvb

__global__ __launch_bounds__(64) void device_math_kernel(int what, int n, const double *in, double *out)
{
    __shared__ double exp_tab[64];
    exp_tab[threadIdx.x] = FVB_EXP_TABLE[threadIdx.x]; // (as the kernels that use exp_acc do: vb_spatial.h)
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n)
        return;
    if (what == 0)
        out[i] = exp_acc(in[i], exp_tab);
    else if (what == 1)
        out[i] = exp(in[i]);
    else
    {
        double a[10], r[10], logabs;
        int sign;
        for (int k = 0; k < 10; k++)
            a[k] = in[(size_t)i * 10 + k];
        const bool ok = mvn_invert<4>(a, r, logabs, sign);
        for (int k = 0; k < 10; k++)
            out[(size_t)i * 11 + k] = ok ? r[k] : __builtin_nan("");
        out[(size_t)i * 11 + 10] = logabs;
    }
}

namespace
{
void destroy_idle_pipe_streams();
}

extern "C" {

int32_t fabber_vb_mvn_rows(int32_t n)
{
    return n * (n + 1) / 2 + n + 1;
}

int32_t fabber_vb_abi_version(void)
{
    return FVB_ABI_VERSION;
}

int32_t fabber_vb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *fabber_vb_last_error(void)
{
    return g_last_error.c_str();
}

void fabber_vb_release_cached_memory(void)
{
    destroy_idle_pipe_streams();
    fvb::destroy_side_streams();
    fvb::api_release_pools(0);
}

void fabber_vb_trim_cached_memory(uint64_t keep_bytes)
{
    fvb::api_release_pools(keep_bytes);
}

int32_t fabber_vb_pin_host_buffer(void *ptr, uint64_t bytes)
{
    if (!ptr || bytes == 0)
        return fail(-23, "fabber_vb_pin_host_buffer: NULL pointer or empty range");
    FVB_HIP_CHECK(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable));
    return 0;
}

int32_t fabber_vb_unpin_host_buffer(void *ptr)
{
    if (!ptr)
        return fail(-23, "fabber_vb_unpin_host_buffer: NULL pointer");
    FVB_HIP_CHECK(hipHostUnregister(ptr));
    return 0;
}

void fabber_vb_set_variant(int32_t variant)
{
    g_variant = variant;
}

void fabber_vb_set_residual_mode(int32_t mode)
{
    g_residual_mode = mode;
}

void fabber_vb_set_residual_tolerance(double tol)
{
    g_residual_tol = tol;
}

const char *fabber_vb_kernel_name(const fvb_config *cfg)
{
    if (validate(cfg) != 0)
        return "invalid";
    LaneKernelInfo k = select_lane(cfg);
    if ((k.fn || k.fn_tiles_f32) && g_variant != 2)
        return k.name;
    return "wave";
}

namespace
{
int run_device_as(const fvb_config *cfg, const void *data, const fvb_outputs *out, hipStream_t stream, int n_unmasked,
    int kernel_voxels, BlockSlot *slot = nullptr);
}

int32_t fabber_vb_run_device_ex(const fvb_config *cfg, const void *data, const fvb_outputs *out, void *stream_,
    int32_t n_unmasked)
{
    return run_device_as(cfg, data, out, (hipStream_t)stream_, n_unmasked, cfg ? cfg->n_voxels : 0);
}

namespace
{
// kernel_voxels: the voxel count the kernel choice (lane / wave) is made for - the whole problem's when
// cfg describes one block of it (fabber_vb_run_host_multi), so that every block runs the same code.
// slot: the kernels' work buffers (tiled series, saved state) come from the block's slot (pipelined host entry point)
// instead of the stream-ordered pool.
int run_device_as(const fvb_config *cfg, const void *data, const fvb_outputs *out, hipStream_t stream, int n_unmasked,
    int kernel_voxels, BlockSlot *slot)
{
    int rc = validate(cfg);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return fail(-20, "outputs.mvn is required");
    if (cfg->n_voxels == 0)
        return 0;
    if (!data)
        return fail(-21, "data is NULL");
    if (cfg->noise == FVB_NOISE_AR1 && n_unmasked != cfg->n_times)
        return fail(-15, "Masked time points are not supported for the AR noise model"); // noisemodel_ar.cc:351-355
    api_keep_pool_memory();
    KernelArgs ka;
    ka.cfg = *cfg;
    ka.out = *out;
    ka.data = data;
    ka.tiles = nullptr;
    ka.save = nullptr;
    ka.n_unmasked = n_unmasked;
    ka.residual_mode = g_residual_mode;
    ka.residual_tol = g_residual_tol;
    // The first two linearisations of a run evaluate the model pointwise (vb_lane_kernel.h, recentre). One would do
    // for a single exponential on the population (C2 against the oracle: median 1.5e-10 and 99th percentile 1.1e-8
    // either way, 8 % faster) but not on its worst voxel: 1.5e-6 against 1.05e-6, across the 1e-6 every strict
    // comparison holds without raising it (tools/measure/c2_precise_passes.py) - so two, for every model.
    ka.precise_passes = g_precise_passes >= 0 ? g_precise_passes : 2;

    fvb_config choice = *cfg;
    choice.n_voxels = kernel_voxels;
    LaneKernelInfo lk = select_lane(&choice);
    auto work_alloc = [&](void **p, size_t bytes) { return slot ? slot->take(p, bytes) : api_pool_alloc(p, bytes, stream); };
    if (lk.fn || lk.fn_tiles_f32)
    {
        if (needs_save(cfg))
            FVB_HIP_CHECK(work_alloc((void **)&ka.save, sizeof(double) * (size_t)lk.save_rows * cfg->n_voxels));
        const unsigned grid = (unsigned)((cfg->n_voxels + 63) / 64);
        LaneKernelFn fn = lk.fn;
        void *tiles = nullptr;
        // White noise, no masked timepoints: the series is re-laid per wavefront once (one read and one
        // write of the image) and every pass of the voxel loop streams that block (vb_lane_kernel.h).
        // (the AR(1) kernels exist for the tiled series only)
        if (lk.fn_tiles_f32 && (!lk.fn || (n_unmasked == cfg->n_times && g_residual_mode != 1 && g_tiled)))
        {
            const int V = cfg->n_voxels, T = cfg->n_times;
            const unsigned rgrid = (unsigned)((V + 255) / 256);
            if (cfg->data_f64)
            {
                FVB_HIP_CHECK(work_alloc(&tiles, Tile<double>::bytes(V, T)));
                hipLaunchKernelGGL(retile_series<double>, dim3(rgrid), dim3(256), 0, stream, (const double *)data, (double *)tiles, V, T);
                fn = (cfg->convergence == FVB_CONV_MAXITS && lk.fn_tiles_f64_counting) ? lk.fn_tiles_f64_counting : lk.fn_tiles_f64;
            }
            else
            {
                FVB_HIP_CHECK(work_alloc(&tiles, Tile<float>::bytes(V, T)));
                hipLaunchKernelGGL(retile_series<float>, dim3(rgrid), dim3(256), 0, stream, (const float *)data, (float *)tiles, V, T);
                fn = (cfg->convergence == FVB_CONV_MAXITS && lk.fn_tiles_f32_counting) ? lk.fn_tiles_f32_counting : lk.fn_tiles_f32;
            }
            FVB_HIP_CHECK(hipGetLastError());
            ka.tiles = tiles;
        }
        // (the several-precisions kernels keep the T class bytes of the noise pattern in dynamic LDS)
        const size_t lds = (cfg->noise == FVB_NOISE_WHITE && cfg->n_phis > 1) ? (size_t)((cfg->n_times + 15) & ~15) : 0;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, ka);
        FVB_HIP_CHECK(hipGetLastError());
        for (void *work : { tiles, (void *)ka.save })
            if (work && !slot)
                FVB_HIP_CHECK(api_pool_free(work, stream));
        return 0;
    }
    return launch_wave_kernel(ka, stream, g_last_error);
}
} // namespace

int32_t fabber_vb_run_device(const fvb_config *cfg, const void *data, const fvb_outputs *out, void *stream)
{
    // The number of unmasked timepoints is needed as a scalar by the kernels; phi_index is a
    // device pointer here, so read it back (n_times bytes, once per call).
    int n_unmasked = cfg ? cfg->n_times : 0;
    if (cfg && cfg->phi_index && cfg->n_times > 0)
    {
        std::vector<uint8_t> h(cfg->n_times);
        FVB_HIP_CHECK(hipMemcpyAsync(h.data(), cfg->phi_index, h.size(), hipMemcpyDeviceToHost, (hipStream_t)stream));
        FVB_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        n_unmasked = count_unmasked(cfg, h.data());
    }
    return fabber_vb_run_device_ex(cfg, data, out, stream, n_unmasked);
}

namespace
{
// Voxels [v0, v1) of a host-resident problem on one device: the block's columns of every [row][voxel]
// image go up and down as 2-D copies (row pitch = the caller's n_voxels), the kernels see a problem of
// v1 - v0 voxels. Runs on `stream`; returns after the block's results are in the caller's arrays.
// The three stages of a block: in (allocate, upload), fit (the kernels), out (download). They may run on three
// different streams - the pipelined host entry point below uploads block b + 1 and downloads block b - 1 while block
// b is being fitted - ordered by the events up_done and fit_done.
struct HostBlock
{
    const fvb_config *cfg = nullptr;
    const void *data = nullptr;
    const fvb_outputs *out = nullptr;
    int v0 = 0, v1 = 0, kernel_voxels = 0, rows = 0;
    fvb_config d;
    fvb_outputs dout;
    DevBuf b_data, b_design, b_phi, b_init, b_img[FVB_MAX_PARAMS], b_mvn, b_small, b_hist;
    // F (8 bytes), history length, status, iterations (4 each) of a voxel: one device buffer, [F][hlen][status][it]
    static constexpr size_t SMALL_BYTES_PER_VOXEL = 8 + 4 + 4 + 4;
    size_t small_bytes = 0;
    DeviceParamTable ptable;
    hipEvent_t up_done = nullptr, fit_done = nullptr;
    BlockSlot *slot = nullptr; // pipelined entry point: where every device buffer of the block comes from
    ~HostBlock()
    {
        if (slot)
            slot->reset(); // (the owner destroys a block only after its work has been seen to finish)
        if (up_done)
            (void)hipEventDestroy(up_done);
        if (fit_done)
            (void)hipEventDestroy(fit_done);
    }
    int stage_in(hipStream_t stream)
    {
        const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times, Vb = (size_t)(v1 - v0);
        const int P = cfg->n_params;
        rows = fabber_vb_mvn_rows(P + noise_outputs(cfg));
        const size_t esz = cfg->data_f64 ? 8 : 4;
        auto upload = [&](void *dst, const void *src, size_t elem, size_t nrows) {
            return copy_rows(dst, Vb * elem, (const char *)src + (size_t)v0 * elem, V * elem, Vb * elem, nrows, hipMemcpyHostToDevice,
                stream);
        };
        d = *cfg;
        d.n_voxels = (int32_t)Vb;
        if (slot)
        {
            for (DevBuf *b : { &b_data, &b_design, &b_phi, &b_init, &b_mvn, &b_small, &b_hist })
                b->slot = slot;
            for (DevBuf &b : b_img)
                b.slot = slot;
        }
        FVB_HIP_CHECK(b_data.alloc(T * Vb * esz, stream));
        FVB_HIP_CHECK(upload(b_data.p, data, esz, T));
        if (cfg->design)
        {
            FVB_HIP_CHECK(b_design.alloc(sizeof(double) * T * P, stream));
            FVB_HIP_CHECK(hipMemcpyAsync(b_design.p, cfg->design, sizeof(double) * T * P, hipMemcpyHostToDevice, stream));
            d.design = (const double *)b_design.p;
        }
        if (cfg->phi_index)
        {
            FVB_HIP_CHECK(b_phi.alloc(T, stream));
            FVB_HIP_CHECK(hipMemcpyAsync(b_phi.p, cfg->phi_index, T, hipMemcpyHostToDevice, stream));
            d.phi_index = (const uint8_t *)b_phi.p;
        }
        if (cfg->init_mvn)
        {
            FVB_HIP_CHECK(b_init.alloc(sizeof(double) * rows * Vb, stream));
            FVB_HIP_CHECK(upload(b_init.p, cfg->init_mvn, sizeof(double), rows));
            d.init_mvn = (const double *)b_init.p;
        }
        if (cfg->params_ext) // more than FVB_MAX_PARAMS parameters: the per-parameter entries as a table on the device
        {
            const int rc = ptable.upload(cfg, (size_t)v0, Vb, stream);
            if (rc)
                return rc;
            d.params_ext = ptable.device;
        }
        for (int k = 0; k < P && !cfg->params_ext; k++)
            if (cfg->image_prior[k])
            {
                FVB_HIP_CHECK(b_img[k].alloc(sizeof(double) * Vb, stream));
                FVB_HIP_CHECK(upload(b_img[k].p, cfg->image_prior[k], sizeof(double), 1));
                d.image_prior[k] = (const double *)b_img[k].p;
            }
        memset(&dout, 0, sizeof(dout));
        FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * Vb, stream));
        dout.mvn = (double *)b_mvn.p;
        {
            // (sections the caller does not want are left out; every section starts on a multiple of 8 bytes because F
            // comes first and Vb is even for every block but possibly the last, whose int sections are padded)
            const size_t ints = (Vb + 1) / 2 * 2 * sizeof(int32_t);
            small_bytes = (out->free_energy ? sizeof(double) * Vb : 0) + (out->f_history_len ? ints : 0) + (out->status ? ints : 0)
                + (out->iterations ? ints : 0);
            FVB_HIP_CHECK(b_small.alloc(small_bytes, stream));
            char *q = (char *)b_small.p;
            if (out->free_energy)
            {
                dout.free_energy = (double *)q;
                q += sizeof(double) * Vb;
            }
            if (out->f_history_len)
            {
                dout.f_history_len = (int32_t *)q;
                q += ints;
            }
            if (out->status)
            {
                dout.status = (int32_t *)q;
                q += ints;
            }
            if (out->iterations)
                dout.iterations = (int32_t *)q;
        }
        if (out->f_history && cfg->f_history_rows > 0)
        {
            FVB_HIP_CHECK(b_hist.alloc(sizeof(double) * cfg->f_history_rows * Vb, stream));
            FVB_HIP_CHECK(hipMemsetAsync(b_hist.p, 0xff, sizeof(double) * cfg->f_history_rows * Vb, stream)); // NaN fill
            dout.f_history = (double *)b_hist.p;
        }
        FVB_HIP_CHECK(hipEventCreateWithFlags(&up_done, hipEventDisableTiming));
        FVB_HIP_CHECK(hipEventRecord(up_done, stream));
        return 0;
    }
    int fit(hipStream_t stream, int n_unmasked)
    {
        FVB_HIP_CHECK(hipStreamWaitEvent(stream, up_done, 0));
        int rc = run_device_as(&d, b_data.p, &dout, stream, n_unmasked, kernel_voxels, slot);
        if (rc)
            return rc;
        FVB_HIP_CHECK(hipEventCreateWithFlags(&fit_done, hipEventDisableTiming));
        FVB_HIP_CHECK(hipEventRecord(fit_done, stream));
        return 0;
    }
    // returns after the block's results are in the caller's arrays. bounce: pinned host memory of at least small_bytes
    // (the small arrays come down in one copy and are handed out from there), or NULL: one copy per array
    int stage_out(hipStream_t stream, void *bounce = nullptr)
    {
        const size_t V = (size_t)cfg->n_voxels, Vb = (size_t)(v1 - v0);
        auto download = [&](void *dst, const void *src, size_t elem, size_t nrows) {
            return copy_rows((char *)dst + (size_t)v0 * elem, V * elem, src, Vb * elem, Vb * elem, nrows, hipMemcpyDeviceToHost, stream);
        };
        FVB_HIP_CHECK(hipStreamWaitEvent(stream, fit_done, 0));
        if (bounce && small_bytes)
            FVB_HIP_CHECK(hipMemcpyAsync(bounce, b_small.p, small_bytes, hipMemcpyDeviceToHost, stream)); // (ahead of the big one)
        FVB_HIP_CHECK(download(out->mvn, dout.mvn, sizeof(double), rows));
        if (dout.f_history)
            FVB_HIP_CHECK(download(out->f_history, dout.f_history, sizeof(double), cfg->f_history_rows));
        if (!bounce)
        {
            if (dout.free_energy)
                FVB_HIP_CHECK(download(out->free_energy, dout.free_energy, sizeof(double), 1));
            if (dout.f_history_len)
                FVB_HIP_CHECK(download(out->f_history_len, dout.f_history_len, sizeof(int32_t), 1));
            if (dout.status)
                FVB_HIP_CHECK(download(out->status, dout.status, sizeof(int32_t), 1));
            if (dout.iterations)
                FVB_HIP_CHECK(download(out->iterations, dout.iterations, sizeof(int32_t), 1));
        }
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
        if (bounce && small_bytes)
        {
            const char *base = (const char *)b_small.p;
            auto hand_out = [&](void *dst, const void *dev, size_t elem) {
                memcpy((char *)dst + (size_t)v0 * elem, (const char *)bounce + ((const char *)dev - base), Vb * elem);
            };
            if (dout.free_energy)
                hand_out(out->free_energy, dout.free_energy, sizeof(double));
            if (dout.f_history_len)
                hand_out(out->f_history_len, dout.f_history_len, sizeof(int32_t));
            if (dout.status)
                hand_out(out->status, dout.status, sizeof(int32_t));
            if (dout.iterations)
                hand_out(out->iterations, dout.iterations, sizeof(int32_t));
        }
        return 0;
    }
};

int run_host_block(const fvb_config *cfg, const void *data, const fvb_outputs *out, int device, int v0, int v1,
    hipStream_t stream, int kernel_voxels)
{
    FVB_HIP_CHECK(hipSetDevice(device));
    if (v1 <= v0)
        return 0;
    HostBlock b;
    b.cfg = cfg;
    b.data = data;
    b.out = out;
    b.v0 = v0;
    b.v1 = v1;
    b.kernel_voxels = kernel_voxels;
    int rc = b.stage_in(stream);
    if (rc == 0)
        rc = b.fit(stream, count_unmasked(cfg, cfg->phi_index));
    if (rc == 0)
        rc = b.stage_out(stream);
    if (rc)
        (void)hipStreamSynchronize(stream); // (before the block's buffers go back to the pool)
    return rc;
}

// The whole problem through host pointers on one device, as a PIPELINE over blocks of voxels: while block b is being
// fitted, block b + 1 goes up and block b - 1 comes down (three streams; the caller's buffers are pageable, so a copy
// holds the host thread that issued it: this thread uploads and launches, a second one downloads). What a voxel gets
// depends on nothing but that voxel, and every block runs the kernel the whole problem would, so the result is the
// one-block run's bit for bit (tests/test_multi_device.py). C3 (1e6 voxels, 400 MB up, 176 MB down): the sum of its
// parts was 7.5 + 15.2 + 3.3 ms = 28 - 29 ms with overheads; piped 25 - 27 ms. What keeps it from 1.9 + 15.2 + 0.8 ms
// (rocprofv3 --kernel-trace --memory-copy-trace of one call, profiles/r3_host_pipeline.md): the runtime takes a 2-D
// copy from pageable memory as gather -> DMA at 57 GB/s -> a blit KERNEL that has to find room between the resident
// wavefronts of the fit (1.7 - 2.6 ms per block, and the fit of a block 25 % longer while it runs). Tried and
// dropped: gathering the rows into pinned slots with 12 host threads and one DMA per slot - the host's own memcpy
// (~20 GB/s) is then on the critical path: 32 - 37 ms. A caller that keeps its buffers can take the copies off the
// host altogether: fabber_vb_pin_host_buffer (below) - the same calls then are asynchronous rectangle DMAs.
// The four streams of a pipelined call (and a pinned bounce buffer for the blocks' small result arrays) are kept between
// calls: creating and destroying them was 4 - 5 ms of a 23 ms call (hipStreamCreateWithFlags 0.45 - 0.6 ms each,
// hipStreamDestroy 0.5 - 1.2 ms each: rocprofv3 --hip-runtime-trace, tools/measure/host_timeline.sh). A call takes a
// set of its device from the cache or makes one; fabber_vb_release_cached_memory destroys the idle ones.
struct PipeStreams
{
    int device = -1;
    hipStream_t up = nullptr, fit[2] = { nullptr, nullptr }, down = nullptr;
    void *bounce = nullptr; // pinned host memory
    size_t bounce_bytes = 0;
    BlockSlot slots[3];     // device memory of the (at most three) blocks in flight
    int create(int dev)
    {
        device = dev;
        FVB_HIP_CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        FVB_HIP_CHECK(hipStreamCreateWithFlags(&fit[0], hipStreamNonBlocking));
        FVB_HIP_CHECK(hipStreamCreateWithFlags(&fit[1], hipStreamNonBlocking));
        FVB_HIP_CHECK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
        return 0;
    }
    int need_bounce(size_t bytes)
    {
        if (bytes <= bounce_bytes)
            return 0;
        if (bounce)
            (void)hipHostFree(bounce);
        bounce = nullptr;
        bounce_bytes = 0;
        FVB_HIP_CHECK(hipHostMalloc(&bounce, bytes, hipHostMallocDefault));
        bounce_bytes = bytes;
        return 0;
    }
    void destroy()
    {
        for (hipStream_t s : { up, fit[0], fit[1], down })
            if (s)
                (void)hipStreamDestroy(s);
        if (bounce)
            (void)hipHostFree(bounce);
        for (BlockSlot &sl : slots)
            sl.destroy();
        up = fit[0] = fit[1] = down = nullptr;
        bounce = nullptr;
        bounce_bytes = 0;
    }
};
struct PipeStreamCache
{
    std::mutex lock;
    std::vector<PipeStreams> idle;
};
static PipeStreamCache &pipe_stream_cache()
{
    static PipeStreamCache c;
    return c;
}
static int acquire_pipe_streams(int device, PipeStreams &ps)
{
    {
        PipeStreamCache &c = pipe_stream_cache();
        std::lock_guard<std::mutex> hold(c.lock);
        for (size_t i = 0; i < c.idle.size(); i++)
            if (c.idle[i].device == device)
            {
                ps = c.idle[i];
                c.idle.erase(c.idle.begin() + (long)i);
                return 0;
            }
    }
    const int rc = ps.create(device);
    if (rc)
        ps.destroy();
    return rc;
}
static void release_pipe_streams(const PipeStreams &ps)
{
    PipeStreamCache &c = pipe_stream_cache();
    std::lock_guard<std::mutex> hold(c.lock);
    c.idle.push_back(ps);
}
void destroy_idle_pipe_streams()
{
    PipeStreamCache &c = pipe_stream_cache();
    std::lock_guard<std::mutex> hold(c.lock);
    int now = 0;
    (void)hipGetDevice(&now);
    for (PipeStreams &ps : c.idle)
    {
        if (hipSetDevice(ps.device) == hipSuccess)
            ps.destroy();
    }
    c.idle.clear();
    (void)hipSetDevice(now);
}

int run_host_pipelined(const fvb_config *cfg, const void *data, const fvb_outputs *out, int device, int block_voxels)
{
    FVB_HIP_CHECK(hipSetDevice(device));
    const int V = cfg->n_voxels;
    // The blocks: the upload is about twice as fast as the fit (56 GB/s of series against ~75 voxels/us on C3), so after
    // the first block the fits are what the call waits for - the first block is as small as fills the chip once (its
    // upload is the pipeline's fill, nothing overlaps it), the last one too (its download is the drain), the ones in
    // between the nominal size: a block more than twice its predecessor is still going up when that one's fit ends
    // (measured on C3, tools/measure/host_pipeline.py: 131072 / 262144 / 262144 / 213568 / 131072 voxels 23.1 ms,
    // four equal blocks 23.9, 131072 / 368960 / 368896 / 131072 25.1). FVB_HOST_BLOCK_SCHEDULE=a,b,c,... (voxels per
    // block, the last one takes the rest) overrides it for experiments.
    std::vector<int> bounds(1, 0);
    if (const char *sched = getenv("FVB_HOST_BLOCK_SCHEDULE"))
    {
        for (const char *p = sched; *p && bounds.back() < V;)
        {
            const int n = std::max(64, atoi(p) / 64 * 64);
            bounds.push_back(std::min(V, bounds.back() + n));
            while (*p && *p != ',')
                p++;
            if (*p == ',')
                p++;
        }
        if (bounds.back() < V)
            bounds.push_back(V);
    }
    else
    {
        const int edge = std::max(64, block_voxels / 2 / 64 * 64);
        if (V >= 2 * edge + block_voxels)
        {
            const int middle = (V - edge) / 64 * 64 - edge; // (every block but the last is whole wavefronts)
            bounds.push_back(edge);
            for (int done = 0; done < middle; done += block_voxels)
                bounds.push_back(edge + std::min(middle, done + block_voxels));
            bounds.push_back(V);
        }
        else
            for (int v = block_voxels; bounds.back() < V; v += block_voxels)
                bounds.push_back(std::min(V, v));
    }
    const int n_blocks = (int)bounds.size() - 1;
    const int n_unmasked = count_unmasked(cfg, cfg->phi_index);
    // (two streams take the blocks' kernels in turn: the first wavefronts of block b + 1 move into the SIMDs the last
    // stragglers of block b have left, instead of every block paying for its own tail)
    PipeStreams ps;
    {
        const int rc_streams = acquire_pipe_streams(device, ps);
        if (rc_streams)
            return rc_streams;
    }
    hipStream_t s_up = ps.up, s_fit[2] = { ps.fit[0], ps.fit[1] }, s_down = ps.down;
    {
        // the small result arrays of a block (F, history length, status, iterations) come down as ONE copy into pinned
        // memory and are handed out from there: four pageable copies of ~1 MB cost the download thread 0.15 - 0.3 ms each
        int widest = 0;
        for (int b = 0; b < n_blocks; b++)
            widest = std::max(widest, bounds[(size_t)b + 1] - bounds[(size_t)b]);
        const int rc_bounce = ps.need_bounce((size_t)widest * HostBlock::SMALL_BYTES_PER_VOXEL + 64);
        if (rc_bounce)
        {
            ps.destroy();
            return rc_bounce;
        }
    }
    std::vector<std::unique_ptr<HostBlock> > blocks((size_t)n_blocks);
    // blocks whose results are down: their buffers go back to the pool on a thread of their own (hipFreeAsync of a
    // block's buffers took 0.4 - 1.0 ms of the download thread per block)
    std::vector<std::unique_ptr<HostBlock> > finished;
    bool no_more_finished = false;
    std::mutex lock;
    std::condition_variable cv;
    int launched = 0;       // blocks whose fit has been enqueued
    int released = 0;       // blocks whose buffers have gone back to the pool
    bool abandoned = false; // the uploading side failed: no more blocks will come
    int rc_down = 0;
    std::string err_down;
    std::thread downloader([&] {
        if (hipSetDevice(device) != hipSuccess)
        {
            std::unique_lock<std::mutex> hold(lock);
            rc_down = -32;
            err_down = "cannot select the device in the download thread";
            cv.notify_all();
            return;
        }
        for (int b = 0; b < n_blocks; b++)
        {
            {
                std::unique_lock<std::mutex> hold(lock);
                cv.wait(hold, [&] { return launched > b || abandoned; });
                if (launched <= b)
                    return;
            }
            const int rc = blocks[(size_t)b]->stage_out(s_down, ps.bounce);
            std::unique_lock<std::mutex> hold(lock);
            if (rc && rc_down == 0)
            {
                rc_down = rc;
                err_down = g_last_error; // thread-local: carry it to the caller's thread
            }
            finished.push_back(std::move(blocks[(size_t)b])); // (to the reaper: at most three blocks are resident)
            cv.notify_all();
        }
    });
    std::thread reaper([&] {
        (void)hipSetDevice(device);
        for (;;)
        {
            std::unique_ptr<HostBlock> blk;
            {
                std::unique_lock<std::mutex> hold(lock);
                cv.wait(hold, [&] { return !finished.empty() || no_more_finished; });
                if (finished.empty())
                    return;
                blk = std::move(finished.front());
                finished.erase(finished.begin());
            }
            blk.reset(); // its buffers go back to the pool
            std::unique_lock<std::mutex> hold(lock);
            released++;
            cv.notify_all();
        }
    });
    int rc = 0;
    for (int b = 0; b < n_blocks && rc == 0; b++)
    {
        {
            // at most three blocks hold device buffers at once (with copies that really are asynchronous - buffers the
            // caller pinned - this loop would otherwise stage the whole problem before the first download ends)
            std::unique_lock<std::mutex> hold(lock);
            cv.wait(hold, [&] { return b - released < 3 || rc_down != 0; });
        }
        std::unique_ptr<HostBlock> blk(new HostBlock);
        blk->cfg = cfg;
        blk->data = data;
        blk->out = out;
        blk->v0 = bounds[(size_t)b];
        blk->v1 = bounds[(size_t)b + 1];
        blk->kernel_voxels = V;
        blk->slot = &ps.slots[b % 3]; // (free: block b - 3 has been reaped, see the wait above)
        rc = blk->stage_in(s_up);
        if (rc == 0)
            rc = blk->fit(s_fit[b & 1], n_unmasked);
        std::unique_lock<std::mutex> hold(lock);
        if (rc == 0)
        {
            blocks[(size_t)b] = std::move(blk);
            launched = b + 1;
        }
        else
        {
            (void)hipDeviceSynchronize(); // (before the failed block's buffers go back to the pool)
            abandoned = true;
        }
        cv.notify_all();
    }
    downloader.join();
    (void)hipStreamSynchronize(s_up);
    (void)hipStreamSynchronize(s_fit[0]);
    (void)hipStreamSynchronize(s_fit[1]);
    (void)hipStreamSynchronize(s_down);
    {
        std::unique_lock<std::mutex> hold(lock);
        no_more_finished = true;
        cv.notify_all();
    }
    reaper.join();
    blocks.clear();
    release_pipe_streams(ps);
    if (rc)
        return rc;
    if (rc_down)
        return fail(rc_down, err_down);
    return 0;
}
} // namespace

int32_t fabber_vb_run_host(const fvb_config *cfg, const void *data, const fvb_outputs *out, int32_t device)
{
    int rc = validate(cfg);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return fail(-20, "outputs.mvn is required");
    if (fabber_vb_device_count() <= 0)
        return fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    // large problems on a throughput (lane) kernel: upload, fit and download overlap block by block. A block holds
    // as many voxels as fill the chip twice (256 CUs x 4 SIMDs x 2 waves x 64 lanes = 131072): smaller ones leave
    // SIMDs idle in every block's tail, larger ones lengthen the pipeline's fill and drain.
    int block = 262144;
    if (const char *e = getenv("FVB_HOST_BLOCK_VOXELS")) // (0 = one block, the round-2 behaviour)
        block = atoi(e) / 64 * 64;
    fvb_config choice = *cfg;
    if (block > 0 && cfg->n_voxels >= 2 * block && (select_lane(&choice).fn || select_lane(&choice).fn_tiles_f32))
        return run_host_pipelined(cfg, data, out, device, block);
    return run_host_block(cfg, data, out, device, 0, cfg->n_voxels, nullptr, cfg->n_voxels);
}

int32_t fabber_vb_run_host_multi(const fvb_config *cfg, const void *data, const fvb_outputs *out,
    const int32_t *devices, int32_t n_devices, fvb_summary *summary)
{
    int rc = validate(cfg);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return fail(-20, "outputs.mvn is required");
    const int visible = fabber_vb_device_count();
    if (visible <= 0)
        return fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    std::vector<int> devs;
    if (devices)
    {
        if (n_devices <= 0)
            return fail(-31, "empty device list");
        for (int i = 0; i < n_devices; i++)
        {
            if (devices[i] < 0 || devices[i] >= visible)
                return fail(-31, "device index " + std::to_string(devices[i]) + " out of range (" + std::to_string(visible) + " visible)");
            devs.push_back(devices[i]);
        }
    }
    else
        for (int i = 0; i < visible; i++)
            devs.push_back(i);
    const int n = (int)devs.size();
    const long long V = cfg->n_voxels;
    // contiguous blocks, cut on multiples of 64 voxels (a wavefront of the lane kernels) so that a
    // voxel shares its wavefront with the same neighbours as in a one-device run
    std::vector<int> cut(n + 1, 0);
    for (int i = 1; i < n; i++)
        cut[i] = (int)std::min<long long>(V, ((V * i / n + 63) / 64) * 64);
    cut[n] = (int)V;
    std::vector<int> rcs(n, 0);
    std::vector<std::string> errs(n);
    std::vector<std::thread> pool;
    // (a device listed several times - a rehearsal of the N-block path on one GPU - takes its blocks one after the other:
    // several streams on one device's stream-ordered pool are what ROCm 7.2's runtime does not keep apart)
    std::vector<std::mutex> one_at_a_time((size_t)visible);
    for (int i = 0; i < n; i++)
        pool.emplace_back([&, i] {
            std::lock_guard<std::mutex> turn(one_at_a_time[(size_t)devs[i]]);
            hipStream_t st = nullptr;
            if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            {
                rcs[i] = -32;
                errs[i] = "cannot create a stream on device " + std::to_string(devs[i]);
                return;
            }
            // every block runs the kernel the whole problem would (kernel choice depends on the voxel count)
            rcs[i] = run_host_block(cfg, data, out, devs[i], cut[i], cut[i + 1], st, cfg->n_voxels);
            if (rcs[i])
                errs[i] = g_last_error; // thread-local: carry it to the caller's thread
            (void)hipStreamDestroy(st);
        });
    for (auto &t : pool)
        t.join();
    for (int i = 0; i < n; i++)
        if (rcs[i])
            return fail(rcs[i], "block " + std::to_string(i) + " (device " + std::to_string(devs[i]) + "): " + errs[i]);
    if (summary)
    {
        summary->sum_free_energy = 0;
        summary->sum_iterations = 0;
        summary->bad_voxels = 0;
        for (long long v = 0; v < V; v++)
        {
            const bool ok = !out->status || (out->status[v] & 0xff) == 0;
            if (out->status && !ok)
                summary->bad_voxels++;
            if (out->free_energy && ok)
                summary->sum_free_energy += out->free_energy[v];
            if (out->iterations)
                summary->sum_iterations += out->iterations[v];
        }
    }
    return 0;
}

int32_t fabber_vb_postproc_device(const fvb_config *cfg, const void *data, const double *mvn, const fvb_postproc *pp,
    void *stream)
{
    int rc = validate(cfg, true, true); // the output images do not depend on the prior types
    if (rc)
        return rc;
    if (!mvn || !pp)
        return fail(-22, "mvn / postproc outputs are NULL");
    if ((pp->residuals) && !data)
        return fail(-21, "residuals need the data");
    if (cfg->n_voxels == 0)
        return 0;
    const unsigned grid = (unsigned)((cfg->n_voxels + 255) / 256);
    if (cfg->n_params > FVB_MAX_PARAMS)
        hipLaunchKernelGGL(vb_postproc_kernel<FVB_MAX_PARAMS_EXT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *cfg, data, mvn, *pp,
            noise_outputs(cfg));
    else
        hipLaunchKernelGGL(vb_postproc_kernel<FVB_MAX_PARAMS>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *cfg, data, mvn, *pp,
            noise_outputs(cfg));
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}

int32_t fabber_vb_postproc_host(const fvb_config *cfg, const void *data, const double *mvn, const fvb_postproc *pp,
    int32_t device)
{
    int rc = validate(cfg, true, true);
    if (rc)
        return rc;
    if (fabber_vb_device_count() <= 0)
        return fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    FVB_HIP_CHECK(hipSetDevice(device));
    const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times;
    if (V == 0)
        return 0;
    const int P = cfg->n_params, N = noise_outputs(cfg);
    const int rows = fabber_vb_mvn_rows(P + N);
    const size_t esz = cfg->data_f64 ? 8 : 4;
    fvb_config d = *cfg;
    DevBuf b_data, b_design, b_mvn;
    DeviceParamTable ptable;
    if (cfg->params_ext)
    {
        fvb_config no_images = *cfg; // (the images of image priors play no part in the result images)
        fvb_param_table t = *cfg->params_ext;
        t.image_prior = nullptr;
        std::vector<int32_t> types((size_t)P, FVB_PRIOR_NORMAL);
        t.prior_type = types.data();
        no_images.params_ext = &t;
        if ((rc = ptable.upload(&no_images, 0, V, nullptr)) != 0)
            return rc;
        d.params_ext = ptable.device;
    }
    if (data)
    {
        FVB_HIP_CHECK(b_data.alloc(T * V * esz));
        FVB_HIP_CHECK(hipMemcpy(b_data.p, data, T * V * esz, hipMemcpyHostToDevice));
    }
    if (cfg->design)
    {
        FVB_HIP_CHECK(b_design.alloc(sizeof(double) * T * P));
        FVB_HIP_CHECK(hipMemcpy(b_design.p, cfg->design, sizeof(double) * T * P, hipMemcpyHostToDevice));
        d.design = (const double *)b_design.p;
    }
    FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * V));
    {
        // Only the rows the kernel reads go up: the means (contiguous) and, where a variance is asked for, the diagonal of
        // the covariance - for C3's mean images 40 MB of the 168 MB image (3 ms of a 30 ms fabber_dorun at PCIe rate).
        const int n = P + N, nCov = n * (n + 1) / 2;
        const size_t row = sizeof(double) * V;
        FVB_HIP_CHECK(hipMemcpy((char *)b_mvn.p + (size_t)nCov * row, (const char *)mvn + (size_t)nCov * row, (size_t)n * row, hipMemcpyHostToDevice));
        if (pp->var || pp->std || pp->zstat || pp->noise_std)
            for (int q = 0; q < n; q++)
            {
                const size_t r = (size_t)(q * (q + 1) / 2 + q);
                FVB_HIP_CHECK(hipMemcpy((char *)b_mvn.p + r * row, (const char *)mvn + r * row, row, hipMemcpyHostToDevice));
            }
        else // (the kernel loads the variance beside the mean whatever is asked for: give it something defined)
            for (int q = 0; q < n; q++)
                FVB_HIP_CHECK(hipMemsetAsync((char *)b_mvn.p + (size_t)(q * (q + 1) / 2 + q) * row, 0, row, nullptr));
    }
    struct Item
    {
        double *const *host;
        double **dev;
        size_t rows;
    };
    fvb_postproc dpp;
    memset(&dpp, 0, sizeof(dpp));
    Item items[8] = { { &pp->mean, &dpp.mean, (size_t)P }, { &pp->var, &dpp.var, (size_t)P },
        { &pp->std, &dpp.std, (size_t)P }, { &pp->zstat, &dpp.zstat, (size_t)P }, { &pp->modelfit, &dpp.modelfit, T },
        { &pp->residuals, &dpp.residuals, T }, { &pp->noise_mean, &dpp.noise_mean, (size_t)N },
        { &pp->noise_std, &dpp.noise_std, (size_t)N } };
    DevBuf bufs[8];
    for (int i = 0; i < 8; i++)
        if (*items[i].host)
        {
            FVB_HIP_CHECK(bufs[i].alloc(sizeof(double) * items[i].rows * V));
            *items[i].dev = (double *)bufs[i].p;
        }
    rc = fabber_vb_postproc_device(&d, data ? b_data.p : nullptr, (const double *)b_mvn.p, &dpp, nullptr);
    if (rc)
        return rc;
    FVB_HIP_CHECK(hipDeviceSynchronize());
    for (int i = 0; i < 8; i++)
        if (*items[i].host)
            FVB_HIP_CHECK(hipMemcpy(*items[i].host, *items[i].dev, sizeof(double) * items[i].rows * V, hipMemcpyDeviceToHost));
    return 0;
}

// ---- host-compiled twins of device building blocks, for unit tests without a GPU --------------
int32_t fabber_vb_convergence_trace(int32_t conv, int32_t max_iterations, int32_t max_trials, double min_fchange,
    const double *F, int32_t nF, int32_t *done, int32_t *save, int32_t *revert, double *alpha, int32_t stop_at_done)
{
    ConvState c;
    conv_init(c, conv, max_iterations, max_trials, min_fchange);
    conv_reset(c);
    int n = 0;
    for (int i = 0; i < nF; i++)
    {
        bool d = conv_test(c, F[i]);
        done[i] = d;
        save[i] = conv_need_save(c);
        revert[i] = conv_need_revert(c);
        alpha[i] = conv_lm_alpha(c);
        n++;
        if (d && stop_at_done)
            break;
    }
    return n;
}

double fabber_vb_gammaln(double x)
{
    return gammaln(x);
}
double fabber_vb_digamma(double x)
{
    return digamma(x);
}
double fabber_vb_exp_acc(double x) // (the host twin of the kernels' half-ulp exp, for tests/test_math_host.py)
{
    return exp_acc(x);
}
double fabber_vb_transform(int32_t which, int32_t tr, double x)
{
    switch (which)
    {
    case 0:
        return to_model(tr, x);
    case 1:
        return to_fabber(tr, x);
    case 2:
        return to_model_var(tr, x);
    default:
        return to_fabber_var(tr, x);
    }
}

// The same building blocks evaluated ON THE DEVICE (tests/test_hip_parity.py: the kernels are compiled with
// contraction allowed, so the error-free transforms inside exp_acc and the frexp-product log-determinant are checked
// as the device compiles them, not only through their host twin). what: 0 exp_acc with its table in LDS as the
// kernels keep it, 1 the device library's exp, 2 mvn_invert<4> of the packed 4 x 4 matrix at in[10 i ..]: out[11 i ..]
// = the packed inverse and log|det|. Host pointers; n values / matrices.
int32_t fabber_vb_device_math(int32_t what, int32_t n, const double *in, double *out)
{
    if (fabber_vb_device_count() <= 0)
        return fail(-30, "no HIP device available");
    if (n <= 0 || !in || !out || what < 0 || what > 2)
        return fail(-60, "fabber_vb_device_math: bad arguments");
    const size_t n_in = (size_t)n * (what == 2 ? 10 : 1), n_out = (size_t)n * (what == 2 ? 11 : 1);
    double *d_in = nullptr, *d_out = nullptr;
    FVB_HIP_CHECK(hipMalloc(&d_in, sizeof(double) * n_in));
    FVB_HIP_CHECK(hipMalloc(&d_out, sizeof(double) * n_out));
    FVB_HIP_CHECK(hipMemcpy(d_in, in, sizeof(double) * n_in, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(device_math_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, what, n, (const double *)d_in, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpy(out, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return e == hipSuccess ? 0 : fail(-100 - (int)e, hipGetErrorString(e));
}

// Host twin of the in-register LDL^T inverse for P = 1..6 (unit test of vb_math.h)
int32_t fabber_vb_ldl_inverse(int32_t P, const double *packed, double *inv_packed, double *logabs, int32_t *sign)
{
    bool ok = false;
    double la = 0;
    int sg = 1;
#define FVB_LDL_CASE(PP)                                                                                     \
    case PP:                                                                                                 \
    {                                                                                                        \
        double a[PP * (PP + 1) / 2], r[PP * (PP + 1) / 2];                                                   \
        for (int i = 0; i < PP * (PP + 1) / 2; i++)                                                          \
            a[i] = packed[i];                                                                                \
        ok = mvn_invert<PP>(a, r, la, sg);                                                                   \
        for (int i = 0; i < PP * (PP + 1) / 2; i++)                                                          \
            inv_packed[i] = r[i];                                                                            \
        break;                                                                                               \
    }
    switch (P)
    {
        FVB_LDL_CASE(1)
        FVB_LDL_CASE(2)
        FVB_LDL_CASE(3)
        FVB_LDL_CASE(4)
        FVB_LDL_CASE(5)
        FVB_LDL_CASE(6)
    default:
        return -1;
    }
    if (logabs)
        *logabs = la;
    if (sign)
        *sign = sg;
    return ok ? 0 : 1;
}

} // extern "C"
