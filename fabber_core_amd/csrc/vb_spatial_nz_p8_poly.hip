// Spatial VB with 5 - 8 noise precisions (SpPattern<P, 8>, vb_spatial_noise.h): poly model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_p8_poly(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_P8_CASE(PolyModel, "poly", 1)
        FVB_SPATIAL_P8_CASE(PolyModel, "poly", 2)
        FVB_SPATIAL_P8_CASE(PolyModel, "poly", 3)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
