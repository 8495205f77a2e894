/*
 * vb_hostmodel_api.hip - driver of voxelwise VB with a host-evaluated forward model
 * (vb_hostmodel.h): alternates the caller's linearisation callback with one step launch until
 * every voxel is done.
 */
#include "vb_hostmodel_ar.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace fvb;

namespace fvb
{
int api_fail(int code, const std::string &msg); // vb_api.hip
int api_validate(const fvb_config *cfg, bool allow_spatial);
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

struct DevMem
{
    void *p = nullptr;
    ~DevMem()
    {
        if (p)
            (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes)
    {
        return hipMalloc(&p, bytes ? bytes : 8);
    }
};
} // namespace

extern "C" int32_t fabber_vb_run_hostmodel_host(const fvb_config *cfg, const void *data, const fvb_outputs *out, int32_t device,
    fvb_linearise_fn linearise, void *user)
{
    int rc = api_validate(cfg, false);
    if (rc)
        return rc;
    if (!linearise)
        return api_fail(-50, "linearisation callback is NULL");
    const bool ar = cfg->noise == FVB_NOISE_AR1;
    if (!cfg->init_mvn)
        return api_fail(-52, "host-evaluated models need the initial posterior as init_mvn (the model's InitVoxelPosterior runs on the host)");
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return api_fail(-30, "no HIP device available (the VB engine has no CPU fallback)");
    FVB_HIP_CHECK(hipSetDevice(device));
    const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times;
    if (V == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    const int P = cfg->n_params, N = cfg->n_phis;
    const int n = P + (ar ? 2 + cfg->ar_cross_terms + N : N), rows = n * (n + 1) / 2 + n + 1;
    const WaveLayout L = wave_layout((int)T, P, N, ar);
    if (L.bytes > 160 * 1024)
        return api_fail(-41, "host-model step kernel: " + std::to_string(L.bytes) + " bytes of LDS needed exceed the 160 KB of a gfx950 CU");
    const size_t esz = cfg->data_f64 ? 8 : 4;
    const size_t lin_stride = T * (size_t)(P + 1);

    fvb_config d = *cfg;
    DevMem b_data, b_phi, b_init, b_img[FVB_MAX_PARAMS], b_mvn, b_f, b_hist, b_hlen, b_status, b_it;
    DevMem b_persist, b_scalars, b_lin, b_slot, b_means, b_phase;
    FVB_HIP_CHECK(b_data.alloc(T * V * esz));
    FVB_HIP_CHECK(hipMemcpy(b_data.p, data, T * V * esz, hipMemcpyHostToDevice));
    d.design = nullptr;
    if (cfg->phi_index)
    {
        FVB_HIP_CHECK(b_phi.alloc(T));
        FVB_HIP_CHECK(hipMemcpy(b_phi.p, cfg->phi_index, T, hipMemcpyHostToDevice));
        d.phi_index = (const uint8_t *)b_phi.p;
    }
    FVB_HIP_CHECK(b_init.alloc(sizeof(double) * rows * V));
    FVB_HIP_CHECK(hipMemcpy(b_init.p, cfg->init_mvn, sizeof(double) * rows * V, hipMemcpyHostToDevice));
    d.init_mvn = (const double *)b_init.p;
    for (int k = 0; k < P; k++)
        if (cfg->image_prior[k])
        {
            FVB_HIP_CHECK(b_img[k].alloc(sizeof(double) * V));
            FVB_HIP_CHECK(hipMemcpy(b_img[k].p, cfg->image_prior[k], sizeof(double) * V, hipMemcpyHostToDevice));
            d.image_prior[k] = (const double *)b_img[k].p;
        }
    fvb_outputs dout;
    memset(&dout, 0, sizeof(dout));
    FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * V));
    dout.mvn = (double *)b_mvn.p;
    if (out->free_energy)
    {
        FVB_HIP_CHECK(b_f.alloc(sizeof(double) * V));
        dout.free_energy = (double *)b_f.p;
    }
    if (out->f_history && cfg->f_history_rows > 0)
    {
        FVB_HIP_CHECK(b_hist.alloc(sizeof(double) * cfg->f_history_rows * V));
        FVB_HIP_CHECK(hipMemset(b_hist.p, 0xff, sizeof(double) * cfg->f_history_rows * V));
        dout.f_history = (double *)b_hist.p;
    }
    if (out->f_history_len)
    {
        FVB_HIP_CHECK(b_hlen.alloc(sizeof(int32_t) * V));
        dout.f_history_len = (int32_t *)b_hlen.p;
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

    HmArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.ka.cfg = d;
    ha.ka.out = dout;
    ha.ka.data = b_data.p;
    ha.ka.save = nullptr;
    int n_unmasked = (int)T;
    if (cfg->phi_index)
    {
        n_unmasked = 0;
        for (size_t t = 0; t < T; t++)
            n_unmasked += (cfg->phi_index[t] != 255);
    }
    ha.ka.n_unmasked = n_unmasked;
    if (ar && n_unmasked != (int)T)
        return api_fail(-15, "Masked time points are not supported for the AR noise model"); // noisemodel_ar.cc:351-355
    ha.ka.residual_mode = 1;
    ha.ka.residual_tol = 0;
    ha.L = L;
    ha.persist_doubles = L.part - L.b;
    FVB_HIP_CHECK(b_persist.alloc(sizeof(double) * (size_t)ha.persist_doubles * V));
    // the step kernel: white noise, or AR(1) with the alpha posterior kept per (echoes, alphas)
    typedef void (*StepFn)(const HmArgs);
    StepFn fn = cfg->need_f ? vb_wave_step_kernel<true> : vb_wave_step_kernel<false>;
    size_t scalars_bytes = sizeof(HmScalars);
    if (ar)
    {
        switch (N * 10 + 2 + cfg->ar_cross_terms)
        {
#define FVB_AR_STEP(KEY, NPHI, NA)                                                                           \
    case KEY:                                                                                                \
        fn = cfg->need_f ? vb_wave_ar_step_kernel<NPHI, NA, true> : vb_wave_ar_step_kernel<NPHI, NA, false>;  \
        scalars_bytes = sizeof(HmArScalars<NPHI, NA>);                                                       \
        break;
            FVB_AR_STEP(12, 1, 2)
            FVB_AR_STEP(22, 2, 2)
            FVB_AR_STEP(23, 2, 3)
            FVB_AR_STEP(24, 2, 4)
#undef FVB_AR_STEP
        default:
            return api_fail(-40, "AR(1) noise: num-echoes must be 1 or 2, cross terms need two echoes");
        }
    }
    FVB_HIP_CHECK(b_scalars.alloc(scalars_bytes * V));
    FVB_HIP_CHECK(hipMemset(b_scalars.p, 0, scalars_bytes * V)); // phase = HM_NEW
    // The voxels still running are worked through in batches: the linearisations of a batch (g and J, T (P + 1)
    // doubles per voxel) are what the host and the device hold at a time, in two buffers each side, so that the
    // host evaluates the model for the next batch while the device steps the current one. (One buffer for the whole
    // volume - 4 GB per million voxels at T = 100, P = 4 - made real volumes fail at allocation.)
    size_t batch_voxels = std::max<size_t>(1, std::min<size_t>(V, std::max<size_t>(4096, (size_t)(256u << 20) / (sizeof(double) * lin_stride))));
    if (const char *forced = getenv("FVB_HOSTMODEL_BATCH")) // tests: several batches on small volumes
        batch_voxels = std::max<size_t>(1, std::min<size_t>(V, (size_t)atol(forced)));
    DevMem b_lin2, b_slot2;
    FVB_HIP_CHECK(b_lin.alloc(sizeof(double) * lin_stride * batch_voxels));
    FVB_HIP_CHECK(b_lin2.alloc(sizeof(double) * lin_stride * batch_voxels));
    FVB_HIP_CHECK(b_slot.alloc(sizeof(int32_t) * batch_voxels));
    FVB_HIP_CHECK(b_slot2.alloc(sizeof(int32_t) * batch_voxels));
    FVB_HIP_CHECK(b_means.alloc(sizeof(double) * (size_t)P * V));
    FVB_HIP_CHECK(b_phase.alloc(sizeof(int32_t) * V));
    ha.persist = (double *)b_persist.p;
    ha.scalars = (HmScalars *)b_scalars.p;
    ha.ar_scalars = b_scalars.p;
    ha.lin = (const double *)b_lin.p;
    ha.batch_ids = (const int32_t *)b_slot.p;
    ha.means_out = (double *)b_means.p;
    ha.phase_out = (int32_t *)b_phase.p;

    if (L.bytes > 64 * 1024)
        FVB_HIP_CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes));

    // host side of the ping-pong
    std::vector<double> means((size_t)P * V), lin[2], active_means;
    lin[0].resize(lin_stride * batch_voxels);
    lin[1].resize(lin_stride * batch_voxels);
    std::vector<int32_t> phase(V, HM_NEW), ids;
    {
        const int nCov = n * (n + 1) / 2;
        for (size_t v = 0; v < V; v++)
            for (int i = 0; i < P; i++)
                means[v * P + i] = cfg->init_mvn[(size_t)(nCov + i) * V + v];
    }
    hipStream_t stream;
    FVB_HIP_CHECK(hipStreamCreate(&stream));
    struct StreamGuard
    {
        hipStream_t s;
        ~StreamGuard()
        {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    } stream_guard = { stream };
    hipEvent_t used[2]; // buffer pair i is free again when the launch that read it is over
    FVB_HIP_CHECK(hipEventCreateWithFlags(&used[0], hipEventDisableTiming));
    FVB_HIP_CHECK(hipEventCreateWithFlags(&used[1], hipEventDisableTiming));
    struct EventGuard
    {
        hipEvent_t *e;
        ~EventGuard()
        {
            (void)hipEventDestroy(e[0]);
            (void)hipEventDestroy(e[1]);
        }
    } event_guard = { used };
    void *dev_lin[2] = { b_lin.p, b_lin2.p }, *dev_ids[2] = { b_slot.p, b_slot2.p };
    // every iteration needs one step, a revert one more, trial / LM modes extra iterations
    const long max_steps = ((long)cfg->max_iterations + 2) * 12 + 8;
    for (long step = 0;; step++)
    {
        ids.clear();
        for (size_t v = 0; v < V; v++)
            if (phase[v] != HM_DONE)
                ids.push_back((int32_t)v);
        if (ids.empty())
            break;
        if (step >= max_steps)
            return api_fail(-53, "host-model loop did not terminate");
        int which = 0;
        for (size_t b0 = 0; b0 < ids.size(); b0 += batch_voxels, which ^= 1)
        {
            const size_t nb = std::min(batch_voxels, ids.size() - b0);
            active_means.resize(nb * (size_t)P);
            for (size_t a = 0; a < nb; a++)
                for (int i = 0; i < P; i++)
                    active_means[a * P + i] = means[(size_t)ids[b0 + a] * P + i];
            // (the launch that read this host / device buffer pair two batches ago has to be over; the host
            // works on this batch's model evaluations while the device steps the previous batch)
            if (b0 >= 2 * batch_voxels)
                FVB_HIP_CHECK(hipEventSynchronize(used[which]));
            // g [T] then J [T][P] per voxel of the batch, about active_means[a][.] (Fabber space)
            const int cb = linearise(user, (int32_t)nb, ids.data() + b0, active_means.data(), lin[which].data());
            if (cb != 0)
                return api_fail(-54, "the model's linearisation callback failed (code " + std::to_string(cb) + ")");
            FVB_HIP_CHECK(hipMemcpyAsync(dev_lin[which], lin[which].data(), sizeof(double) * lin_stride * nb, hipMemcpyHostToDevice, stream));
            FVB_HIP_CHECK(hipMemcpyAsync(dev_ids[which], ids.data() + b0, sizeof(int32_t) * nb, hipMemcpyHostToDevice, stream));
            ha.lin = (const double *)dev_lin[which];
            ha.batch_ids = (const int32_t *)dev_ids[which];
            hipLaunchKernelGGL(fn, dim3((unsigned)nb), dim3(64), L.bytes, stream, ha);
            FVB_HIP_CHECK(hipGetLastError());
            FVB_HIP_CHECK(hipEventRecord(used[which], stream));
        }
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
        FVB_HIP_CHECK(hipMemcpy(phase.data(), b_phase.p, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
        FVB_HIP_CHECK(hipMemcpy(means.data(), b_means.p, sizeof(double) * (size_t)P * V, hipMemcpyDeviceToHost));
    }

    FVB_HIP_CHECK(hipMemcpy(out->mvn, dout.mvn, sizeof(double) * rows * V, hipMemcpyDeviceToHost));
    if (dout.free_energy)
        FVB_HIP_CHECK(hipMemcpy(out->free_energy, dout.free_energy, sizeof(double) * V, hipMemcpyDeviceToHost));
    if (dout.f_history)
        FVB_HIP_CHECK(hipMemcpy(out->f_history, dout.f_history, sizeof(double) * cfg->f_history_rows * V, hipMemcpyDeviceToHost));
    if (dout.f_history_len)
        FVB_HIP_CHECK(hipMemcpy(out->f_history_len, dout.f_history_len, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.status)
        FVB_HIP_CHECK(hipMemcpy(out->status, dout.status, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.iterations)
        FVB_HIP_CHECK(hipMemcpy(out->iterations, dout.iterations, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    return 0;
}
