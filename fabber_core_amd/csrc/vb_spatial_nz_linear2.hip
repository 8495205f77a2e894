// Spatial-VB kernel instantiations for several noise precisions and AR(1) noise (vb_spatial_noise.h), linear model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_linear2(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 5)
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 6)
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
