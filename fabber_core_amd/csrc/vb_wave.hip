// Generic wave-per-voxel kernel: placeholder until the LDS-staged implementation lands.
#include "vb_wave_kernel.h"

namespace fvb
{
int launch_wave_kernel(const KernelArgs &, hipStream_t, std::string &err)
{
    err = "no kernel instantiation for this model / parameter count / noise pattern";
    return -40;
}
} // namespace fvb
