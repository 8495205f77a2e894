// Spatial VB under the general AR(1) noise model (two echoes; SpArN, vb_spatial_noise.h): linear model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_arn_linear2(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_ARN_CASE(LinearModel, "linear", 2)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
