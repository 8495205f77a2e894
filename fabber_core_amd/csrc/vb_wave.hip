// Wave-per-voxel kernel: launch (see vb_wave_kernel.h for the mapping).
#include "vb_wave_kernel.h"

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
    if (cfg.noise != FVB_NOISE_WHITE)
    {
        err = "wave kernel: only the white noise model is built (AR(1) runs on the lane kernels)";
        return -40;
    }
    if (cfg.model != FVB_MODEL_POLY && cfg.model != FVB_MODEL_LINEAR && cfg.model != FVB_MODEL_EXP)
    {
        err = "wave kernel: model has no device body";
        return -40;
    }
    const WaveLayout L = wave_layout(cfg.n_times, cfg.n_params, cfg.n_phis);
    if (L.bytes > LDS_PER_WORKGROUP_MAX)
    {
        err = "wave kernel: " + std::to_string(L.bytes) + " bytes of LDS needed for T=" + std::to_string(cfg.n_times)
            + ", P=" + std::to_string(cfg.n_params) + " exceed the 160 KB of a gfx950 CU";
        return -41;
    }
    auto fn = cfg.need_f ? vb_wave_kernel<true> : vb_wave_kernel<false>;
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
