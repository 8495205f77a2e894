// Spatial-VB kernel instantiations for several noise precisions and AR(1) noise (vb_spatial_noise.h), linear model
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_linear2(int P, bool need_f, int kind); // vb_spatial_nz_linear2.hip
SpatialKernels get_spatial_kernels_nz_linear(int P, bool need_f, int kind)
{
    switch (P)
    {
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 1)
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 2)
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 3)
        FVB_SPATIAL_NZ_CASE(LinearModel, "linear", 4)
    default:
        return get_spatial_kernels_nz_linear2(P, need_f, kind);
    }
}
} // namespace fvb
