// Spatial VB with 5 - 8 noise precisions (SpPattern<P, 8>, vb_spatial_noise.h): exp model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_p8_exp(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_P8_CASE(ExpModel, "exp", 2)
        FVB_SPATIAL_P8_CASE(ExpModel, "exp", 4)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
