/*
 * vb_nlls.hip - instantiations and C ABI of the non-linear least squares kernel
 * (vb_nlls_kernel.h; method=nlls, inference_nlls.cc).
 */
#include "vb_nlls_kernel.h"

#include <hip/hip_runtime.h>

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
    if (cfg->n_params <= 0 || cfg->n_params > FVB_MAX_PARAMS)
        return api_fail(-4, "n_params out of range");
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

} // extern "C"
