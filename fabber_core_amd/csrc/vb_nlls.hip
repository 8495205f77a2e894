/*
 * vb_nlls.hip - instantiations and C ABI of the non-linear least squares kernel
 * (vb_nlls_kernel.h; method=nlls, inference_nlls.cc).
 */
#include "vb_nlls_kernel.h"

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
int api_variant();                               // fabber_vb_set_variant: 0 auto, 1 lane, 2 wave

#define FVB_NLLS_CASE(MODEL, TAG, PP)                                                                        \
    case PP:                                                                                                 \
        return NllsKernelInfo{ nlls_lane_kernel<MODEL<PP>, PP>, "nlls<" TAG "," #PP ">" };

NllsKernelInfo get_nlls_kernel(int model, int P)
{
    switch (model)
    {
    case FVB_MODEL_POLY:
        switch (P)
        {
            FVB_NLLS_CASE(PolyModel, "poly", 1)
            FVB_NLLS_CASE(PolyModel, "poly", 2)
            FVB_NLLS_CASE(PolyModel, "poly", 3)
            FVB_NLLS_CASE(PolyModel, "poly", 4)
            FVB_NLLS_CASE(PolyModel, "poly", 5)
            FVB_NLLS_CASE(PolyModel, "poly", 6)
        }
        break;
    case FVB_MODEL_LINEAR:
        switch (P)
        {
            FVB_NLLS_CASE(LinearModel, "linear", 1)
            FVB_NLLS_CASE(LinearModel, "linear", 2)
            FVB_NLLS_CASE(LinearModel, "linear", 3)
            FVB_NLLS_CASE(LinearModel, "linear", 4)
            FVB_NLLS_CASE(LinearModel, "linear", 5)
            FVB_NLLS_CASE(LinearModel, "linear", 6)
        }
        break;
    case FVB_MODEL_EXP:
        switch (P)
        {
            FVB_NLLS_CASE(ExpModel, "exp", 2)
            FVB_NLLS_CASE(ExpModel, "exp", 4)
            FVB_NLLS_CASE(ExpModel, "exp", 6)
        }
        break;
    }
    return NllsKernelInfo{ nullptr, nullptr };
}
} // namespace fvb

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

int validate_nlls(const fvb_config *cfg, const fvb_nlls *nl)
{
    if (!cfg || !nl)
        return api_fail(-1, "config is NULL");
    if (cfg->abi_version != FVB_ABI_VERSION)
        return api_fail(-2, "fvb_config.abi_version mismatch");
    if (cfg->n_voxels < 0 || cfg->n_times <= 0)
        return api_fail(-3, "bad n_voxels / n_times");
    if (cfg->n_params <= 0 || cfg->n_params > (cfg->params_ext ? FVB_MAX_PARAMS_EXT : FVB_MAX_PARAMS))
        return api_fail(-4, "n_params out of range (more than FVB_MAX_PARAMS parameters: fvb_config.params_ext)");
    if (cfg->model == FVB_MODEL_LINEAR && !cfg->design)
        return api_fail(-10, "linear model needs a design matrix");
    if (cfg->model == FVB_MODEL_EXP && (cfg->n_params != 2 * cfg->model_iopt[0]))
        return api_fail(-11, "exp model: n_params != 2 * num-exps");
    if (cfg->model == FVB_MODEL_POLY && (cfg->n_params != cfg->model_iopt[0] + 1))
        return api_fail(-12, "poly model: n_params != degree + 1");
    if (nl->max_iterations < 0 || !(nl->lambda0 > 0) || !(nl->lambda_max > 0))
        return api_fail(-60, "bad minimiser settings");
    if (cfg->model != FVB_MODEL_POLY && cfg->model != FVB_MODEL_LINEAR && cfg->model != FVB_MODEL_EXP)
        return api_fail(-61, "method=nlls needs a forward model with a device body");
    if (!get_nlls_kernel(cfg->model, cfg->n_params).fn && wave_layout(cfg->n_times, cfg->n_params, 1).bytes > 160 * 1024)
        return api_fail(-61, "no NLLS kernel for this problem: no lane instantiation for the model / parameter count and the "
                             "series does not fit the 160 KB of LDS the wave-per-voxel kernel needs");
    return 0;
}
} // namespace

extern "C" {

void fabber_nlls_defaults(fvb_nlls *nl)
{
    nl->lm = 0;
    nl->max_iterations = 200;
    nl->cf_tolerance = 1e-8;
    nl->lambda0 = 0.1;
    nl->lambda_max = 1e20;
}

int32_t fabber_nlls_run_device(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    void *stream, int32_t n_unmasked)
{
    int rc = validate_nlls(cfg, nl);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    if (cfg->n_voxels == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    NllsArgs na;
    memset(&na, 0, sizeof(na));
    na.ka.cfg = *cfg;
    na.ka.out = *out;
    na.ka.data = data;
    na.ka.n_unmasked = n_unmasked;
    na.nl = *nl;
    const NllsKernelInfo k = get_nlls_kernel(cfg->model, cfg->n_params);
    const WaveLayout L = wave_layout(cfg->n_times, cfg->n_params, 1);
    // lane per voxel where an instantiation exists and there are enough voxels to fill the chip
    // (as the VB kernels, vb_api.cc); wave per voxel otherwise
    const bool wave_fits = L.bytes <= 160 * 1024;
    const int variant = api_variant();
    if (k.fn && variant != 2 && (variant == 1 || cfg->n_voxels >= 4096 || !wave_fits))
    {
        const unsigned grid = (unsigned)((cfg->n_voxels + 63) / 64);
        hipLaunchKernelGGL(k.fn, dim3(grid), dim3(64), 0, (hipStream_t)stream, na);
    }
    else
    {
        if (L.bytes > 64 * 1024)
            FVB_HIP_CHECK(hipFuncSetAttribute((const void *)nlls_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes));
        hipLaunchKernelGGL(nlls_wave_kernel, dim3((unsigned)cfg->n_voxels), dim3(64), L.bytes, (hipStream_t)stream, na, L);
    }
    FVB_HIP_CHECK(hipGetLastError());
    return 0;
}

int32_t fabber_nlls_run_host(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t device)
{
    int rc = validate_nlls(cfg, nl);
    if (rc)
        return rc;
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return api_fail(-30, "no HIP device available (the engine has no CPU fallback)");
    FVB_HIP_CHECK(hipSetDevice(device));
    const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times;
    if (V == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    const int P = cfg->n_params;
    const size_t rows = (size_t)P * (P + 1) / 2 + P + 1;
    const size_t esz = cfg->data_f64 ? 8 : 4;
    fvb_config d = *cfg;
    DevMem b_data, b_design, b_phi, b_mvn, b_status, b_it, b_cf;
    FVB_HIP_CHECK(b_data.alloc(T * V * esz));
    FVB_HIP_CHECK(hipMemcpy(b_data.p, data, T * V * esz, hipMemcpyHostToDevice));
    if (cfg->design)
    {
        FVB_HIP_CHECK(b_design.alloc(sizeof(double) * T * P));
        FVB_HIP_CHECK(hipMemcpy(b_design.p, cfg->design, sizeof(double) * T * P, hipMemcpyHostToDevice));
        d.design = (const double *)b_design.p;
    }
    int n_unmasked = (int)T;
    if (cfg->phi_index)
    {
        n_unmasked = 0;
        for (size_t t = 0; t < T; t++)
            n_unmasked += cfg->phi_index[t] != 255;
        FVB_HIP_CHECK(b_phi.alloc(T));
        FVB_HIP_CHECK(hipMemcpy(b_phi.p, cfg->phi_index, T, hipMemcpyHostToDevice));
        d.phi_index = (const uint8_t *)b_phi.p;
    }
    DevMem b_table;
    if (cfg->params_ext) // more than FVB_MAX_PARAMS parameters: the table on the device (the minimiser reads the transforms
    {                    // and the starting estimate; the prior entries are carried for the post-processing kernel)
        const fvb_param_table *h = cfg->params_ext;
        const size_t off_i = sizeof(fvb_param_table), off_d = off_i + 2 * (size_t)P * sizeof(int32_t) + (2 * (size_t)P * sizeof(int32_t)) % 8;
        const size_t off_p = off_d + 5 * (size_t)P * sizeof(double), bytes = off_p + (size_t)P * sizeof(double *);
        std::vector<char> host(bytes, 0);
        FVB_HIP_CHECK(b_table.alloc(bytes));
        char *dev = (char *)b_table.p;
        fvb_param_table t;
        t.transform = (const int32_t *)(dev + off_i);
        t.prior_type = t.transform + P;
        t.prior_mean = (const double *)(dev + off_d);
        t.prior_var = t.prior_mean + P;
        t.prior_prec = t.prior_var + P;
        t.post_mean = t.prior_prec + P;
        t.post_var = t.post_mean + P;
        t.image_prior = (const double *const *)(dev + off_p); // (all NULL: the minimiser has no priors)
        memcpy(host.data(), &t, sizeof(t));
        memcpy(host.data() + off_i, h->transform, (size_t)P * sizeof(int32_t));
        if (h->prior_type)
            memcpy(host.data() + off_i + (size_t)P * sizeof(int32_t), h->prior_type, (size_t)P * sizeof(int32_t));
        const double *src[5] = { h->prior_mean, h->prior_var, h->prior_prec, h->post_mean, h->post_var };
        for (int a = 0; a < 5; a++)
            if (src[a])
                memcpy(host.data() + off_d + (size_t)a * P * sizeof(double), src[a], (size_t)P * sizeof(double));
        FVB_HIP_CHECK(hipMemcpy(b_table.p, host.data(), bytes, hipMemcpyHostToDevice));
        d.params_ext = (const fvb_param_table *)b_table.p;
    }
    fvb_outputs dout;
    memset(&dout, 0, sizeof(dout));
    FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * V));
    dout.mvn = (double *)b_mvn.p;
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
    if (out->free_energy)
    {
        FVB_HIP_CHECK(b_cf.alloc(sizeof(double) * V));
        dout.free_energy = (double *)b_cf.p;
    }
    rc = fabber_nlls_run_device(&d, nl, b_data.p, &dout, nullptr, n_unmasked);
    if (rc)
        return rc;
    FVB_HIP_CHECK(hipDeviceSynchronize());
    FVB_HIP_CHECK(hipMemcpy(out->mvn, dout.mvn, sizeof(double) * rows * V, hipMemcpyDeviceToHost));
    if (dout.status)
        FVB_HIP_CHECK(hipMemcpy(out->status, dout.status, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.iterations)
        FVB_HIP_CHECK(hipMemcpy(out->iterations, dout.iterations, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.free_energy)
        FVB_HIP_CHECK(hipMemcpy(out->free_energy, dout.free_energy, sizeof(double) * V, hipMemcpyDeviceToHost));
    return 0;
}

// method=nlls with a forward model that exists only as host code: the minimiser's iterations run on the device,
// one launch of nlls_wave_step_kernel per trial point; the caller's callback evaluates the model (prediction and
// Jacobian about the trial point, as LinearizedFwdModel::ReCentre - NLLSCF::cf / grad / hess of
// inference_nlls.cc:223-290 use nothing else) for the voxels still running, in batches of two buffers so that the
// host works on the next batch while the device steps the current one.
int32_t fabber_nlls_run_hostmodel_host(const fvb_config *cfg, const fvb_nlls *nl, const void *data, const fvb_outputs *out,
    int32_t device, fvb_linearise_fn linearise, void *user)
{
    if (!cfg || !nl)
        return api_fail(-1, "config is NULL");
    if (cfg->abi_version != FVB_ABI_VERSION)
        return api_fail(-2, "fvb_config.abi_version mismatch");
    if (cfg->n_voxels < 0 || cfg->n_times <= 0)
        return api_fail(-3, "bad n_voxels / n_times");
    if (cfg->n_params <= 0 || cfg->n_params > FVB_MAX_PARAMS)
        return api_fail(-4, "n_params out of range");
    if (cfg->model != FVB_MODEL_HOSTJAC)
        return api_fail(-61, "fabber_nlls_run_hostmodel_host is for models evaluated on the host (FVB_MODEL_HOSTJAC)");
    if (nl->max_iterations < 0 || !(nl->lambda0 > 0) || !(nl->lambda_max > 0))
        return api_fail(-60, "bad minimiser settings");
    if (!linearise)
        return api_fail(-50, "linearisation callback is NULL");
    if (!out || !out->mvn)
        return api_fail(-20, "outputs.mvn is required");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return api_fail(-30, "no HIP device available (the engine has no CPU fallback)");
    FVB_HIP_CHECK(hipSetDevice(device));
    const size_t V = (size_t)cfg->n_voxels, T = (size_t)cfg->n_times;
    if (V == 0)
        return 0;
    if (!data)
        return api_fail(-21, "data is NULL");
    const int P = cfg->n_params;
    const size_t rows = (size_t)P * (P + 1) / 2 + P + 1;
    const size_t esz = cfg->data_f64 ? 8 : 4;
    const WaveLayout L = wave_layout((int)T, P, 1);
    if (L.bytes > 160 * 1024)
        return api_fail(-41, "NLLS step kernel: " + std::to_string(L.bytes) + " bytes of LDS needed exceed the 160 KB of a gfx950 CU");
    const size_t lin_stride = T * (size_t)(P + 1);

    fvb_config d = *cfg;
    d.design = nullptr;
    DevMem b_data, b_phi, b_mvn, b_status, b_it, b_cf, b_persist, b_scalars, b_lin[2], b_ids[2], b_means, b_phase;
    FVB_HIP_CHECK(b_data.alloc(T * V * esz));
    FVB_HIP_CHECK(hipMemcpy(b_data.p, data, T * V * esz, hipMemcpyHostToDevice));
    int n_unmasked = (int)T;
    if (cfg->phi_index)
    {
        n_unmasked = 0;
        for (size_t t = 0; t < T; t++)
            n_unmasked += cfg->phi_index[t] != 255;
        FVB_HIP_CHECK(b_phi.alloc(T));
        FVB_HIP_CHECK(hipMemcpy(b_phi.p, cfg->phi_index, T, hipMemcpyHostToDevice));
        d.phi_index = (const uint8_t *)b_phi.p;
    }
    fvb_outputs dout;
    memset(&dout, 0, sizeof(dout));
    FVB_HIP_CHECK(b_mvn.alloc(sizeof(double) * rows * V));
    dout.mvn = (double *)b_mvn.p;
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
    if (out->free_energy)
    {
        FVB_HIP_CHECK(b_cf.alloc(sizeof(double) * V));
        dout.free_energy = (double *)b_cf.p;
    }
    NllsHmArgs ha;
    memset(&ha, 0, sizeof(ha));
    ha.na.ka.cfg = d;
    ha.na.ka.out = dout;
    ha.na.ka.data = b_data.p;
    ha.na.ka.n_unmasked = n_unmasked;
    ha.na.nl = *nl;
    ha.L = L;
    ha.persist_doubles = L.part - L.b;
    size_t batch_voxels = std::max<size_t>(1, std::min<size_t>(V, std::max<size_t>(4096, (size_t)(256u << 20) / (sizeof(double) * lin_stride))));
    if (const char *forced = getenv("FVB_HOSTMODEL_BATCH")) // tests: several batches on small volumes
        batch_voxels = std::max<size_t>(1, std::min<size_t>(V, (size_t)atol(forced)));
    FVB_HIP_CHECK(b_persist.alloc(sizeof(double) * (size_t)ha.persist_doubles * V));
    FVB_HIP_CHECK(b_scalars.alloc(sizeof(NllsHmScalars) * V));
    FVB_HIP_CHECK(hipMemset(b_scalars.p, 0, sizeof(NllsHmScalars) * V)); // phase 0 = new
    for (int i = 0; i < 2; i++)
    {
        FVB_HIP_CHECK(b_lin[i].alloc(sizeof(double) * lin_stride * batch_voxels));
        FVB_HIP_CHECK(b_ids[i].alloc(sizeof(int32_t) * batch_voxels));
    }
    FVB_HIP_CHECK(b_means.alloc(sizeof(double) * (size_t)P * V));
    FVB_HIP_CHECK(b_phase.alloc(sizeof(int32_t) * V));
    ha.persist = (double *)b_persist.p;
    ha.scalars = (NllsHmScalars *)b_scalars.p;
    ha.means_out = (double *)b_means.p;
    ha.phase_out = (int32_t *)b_phase.p;
    if (L.bytes > 64 * 1024)
        FVB_HIP_CHECK(hipFuncSetAttribute((const void *)nlls_wave_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes));

    std::vector<double> means((size_t)P * V), lin[2], active_means;
    lin[0].resize(lin_stride * batch_voxels);
    lin[1].resize(lin_stride * batch_voxels);
    std::vector<int32_t> phase(V, 0), ids;
    for (size_t v = 0; v < V; v++)
        for (int i = 0; i < P; i++)
            means[v * P + i] = cfg->post_mean[i]; // the starting estimate, Fabber space (inference_nlls.cc:131)
    hipStream_t stream;
    FVB_HIP_CHECK(hipStreamCreate(&stream));
    hipEvent_t used[2];
    FVB_HIP_CHECK(hipEventCreateWithFlags(&used[0], hipEventDisableTiming));
    FVB_HIP_CHECK(hipEventCreateWithFlags(&used[1], hipEventDisableTiming));
    struct Guard
    {
        hipStream_t s;
        hipEvent_t *e;
        ~Guard()
        {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
            (void)hipEventDestroy(e[0]);
            (void)hipEventDestroy(e[1]);
        }
    } guard = { stream, used };
    // one launch per trial point: the first linearisation, then at most max_iterations trials
    const long max_steps = (long)nl->max_iterations + 2;
    for (long step = 0;; step++)
    {
        ids.clear();
        for (size_t v = 0; v < V; v++)
            if (phase[v] != 3)
                ids.push_back((int32_t)v);
        if (ids.empty())
            break;
        if (step >= max_steps)
            return api_fail(-53, "host-model NLLS loop did not terminate");
        int which = 0;
        for (size_t b0 = 0; b0 < ids.size(); b0 += batch_voxels, which ^= 1)
        {
            const size_t nb = std::min(batch_voxels, ids.size() - b0);
            active_means.resize(nb * (size_t)P);
            for (size_t a = 0; a < nb; a++)
                for (int i = 0; i < P; i++)
                    active_means[a * P + i] = means[(size_t)ids[b0 + a] * P + i];
            if (b0 >= 2 * batch_voxels)
                FVB_HIP_CHECK(hipEventSynchronize(used[which]));
            const int cb = linearise(user, (int32_t)nb, ids.data() + b0, active_means.data(), lin[which].data());
            if (cb != 0)
                return api_fail(-54, "the model's linearisation callback failed (code " + std::to_string(cb) + ")");
            FVB_HIP_CHECK(hipMemcpyAsync(b_lin[which].p, lin[which].data(), sizeof(double) * lin_stride * nb, hipMemcpyHostToDevice, stream));
            FVB_HIP_CHECK(hipMemcpyAsync(b_ids[which].p, ids.data() + b0, sizeof(int32_t) * nb, hipMemcpyHostToDevice, stream));
            ha.lin = (const double *)b_lin[which].p;
            ha.batch_ids = (const int32_t *)b_ids[which].p;
            hipLaunchKernelGGL(nlls_wave_step_kernel, dim3((unsigned)nb), dim3(64), L.bytes, stream, ha);
            FVB_HIP_CHECK(hipGetLastError());
            FVB_HIP_CHECK(hipEventRecord(used[which], stream));
        }
        FVB_HIP_CHECK(hipStreamSynchronize(stream));
        FVB_HIP_CHECK(hipMemcpy(phase.data(), b_phase.p, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
        FVB_HIP_CHECK(hipMemcpy(means.data(), b_means.p, sizeof(double) * (size_t)P * V, hipMemcpyDeviceToHost));
    }
    FVB_HIP_CHECK(hipMemcpy(out->mvn, dout.mvn, sizeof(double) * rows * V, hipMemcpyDeviceToHost));
    if (dout.status)
        FVB_HIP_CHECK(hipMemcpy(out->status, dout.status, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.iterations)
        FVB_HIP_CHECK(hipMemcpy(out->iterations, dout.iterations, sizeof(int32_t) * V, hipMemcpyDeviceToHost));
    if (dout.free_energy)
        FVB_HIP_CHECK(hipMemcpy(out->free_energy, dout.free_energy, sizeof(double) * V, hipMemcpyDeviceToHost));
    return 0;
}


} // extern "C"
