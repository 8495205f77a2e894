// Spatial VB under the general AR(1) noise model (two echoes; SpArN, vb_spatial_noise.h): exp model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_arn_exp4(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_ARN_CASE(ExpModel, "exp", 4)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
