/* vb_wave_kernel.h - generic wave-per-voxel kernel (runtime P, any noise pattern). */
#pragma once

#include "vb_lane_kernel.h"

#include <string>

namespace fvb
{
#if defined(__HIPCC__)
int launch_wave_kernel(const KernelArgs &ka, hipStream_t stream, std::string &err);
#endif
} // namespace fvb
