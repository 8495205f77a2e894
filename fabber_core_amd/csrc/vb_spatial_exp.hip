// Spatial-VB kernel instantiations, exp model
#include "vb_spatial.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_exp(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_CASE(ExpModel, "exp", 2)
        FVB_SPATIAL_CASE(ExpModel, "exp", 4)
    default:
        return SpatialKernels{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr };
    }
}
} // namespace fvb
