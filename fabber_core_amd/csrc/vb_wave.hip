// Wave-per-voxel kernel: launch (see vb_wave_kernel.h for the mapping).
#include "vb_wave_ar_kernel.h"

#include <hip/hip_runtime.h>

namespace fvb
{
namespace
{
constexpr size_t LDS_PER_WORKGROUP_MAX = 160 * 1024; // gfx950: 160 KB per CU, all of it addressable by one workgroup
constexpr size_t LDS_DEFAULT_LIMIT = 64 * 1024;      // above this the kernel attribute has to be raised
} // namespace

int launch_wave_kernel(const KernelArgs &ka, hipStream_t stream, std::string &err)
{
    const fvb_config &cfg = ka.cfg;
    if (cfg.model != FVB_MODEL_POLY && cfg.model != FVB_MODEL_LINEAR && cfg.model != FVB_MODEL_EXP)
    {
        err = "wave kernel: model has no device body";
        return -40;
    }
    const bool ar = cfg.noise == FVB_NOISE_AR1;
    const WaveLayout L = wave_layout(cfg.n_times, cfg.n_params, cfg.n_phis, ar);
    if (L.bytes > LDS_PER_WORKGROUP_MAX)
    {
        err = "wave kernel: " + std::to_string(L.bytes) + " bytes of LDS needed for T=" + std::to_string(cfg.n_times)
            + ", P=" + std::to_string(cfg.n_params) + " exceed the 160 KB of a gfx950 CU";
        return -41;
    }
    auto fn = cfg.need_f ? vb_wave_kernel<true> : vb_wave_kernel<false>;
    if (ar) // one kernel per (echoes, alphas): the alpha posterior lives in registers
    {
        const int key = cfg.n_phis * 10 + 2 + cfg.ar_cross_terms;
        switch (key)
        {
        case 12:
            fn = cfg.need_f ? vb_wave_ar_kernel<1, 2, true> : vb_wave_ar_kernel<1, 2, false>;
            break;
        case 22:
            fn = cfg.need_f ? vb_wave_ar_kernel<2, 2, true> : vb_wave_ar_kernel<2, 2, false>;
            break;
        case 23:
            fn = cfg.need_f ? vb_wave_ar_kernel<2, 3, true> : vb_wave_ar_kernel<2, 3, false>;
            break;
        case 24:
            fn = cfg.need_f ? vb_wave_ar_kernel<2, 4, true> : vb_wave_ar_kernel<2, 4, false>;
            break;
        default:
            err = "AR(1) noise: num-echoes must be 1 or 2, cross terms need two echoes";
            return -40;
        }
    }
    if (L.bytes > LDS_DEFAULT_LIMIT)
    {
        hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.bytes);
        if (e != hipSuccess)
        {
            err = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e);
            return -100 - (int)e;
        }
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)cfg.n_voxels), dim3(64), L.bytes, stream, ka, L);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        err = std::string("vb_wave_kernel launch: ") + hipGetErrorString(e);
        return -100 - (int)e;
    }
    return 0;
}
} // namespace fvb
