// Spatial-VB kernel instantiations for several noise precisions and AR(1) noise (vb_spatial_noise.h), poly model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_poly(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_NZ_CASE(PolyModel, "poly", 1)
        FVB_SPATIAL_NZ_CASE(PolyModel, "poly", 2)
        FVB_SPATIAL_NZ_CASE(PolyModel, "poly", 3)
        FVB_SPATIAL_NZ_CASE(PolyModel, "poly", 4)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
