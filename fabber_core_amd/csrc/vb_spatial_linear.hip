// Spatial-VB kernel instantiations, linear model
#include "vb_spatial.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_linear(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_CASE(LinearModel, "linear", 1)
        FVB_SPATIAL_CASE(LinearModel, "linear", 2)
        FVB_SPATIAL_CASE(LinearModel, "linear", 3)
        FVB_SPATIAL_CASE(LinearModel, "linear", 4)
        FVB_SPATIAL_CASE(LinearModel, "linear", 5)
        FVB_SPATIAL_CASE(LinearModel, "linear", 6)
    default:
        return get_spatial_kernels_more(FVB_MODEL_LINEAR, P, need_f); // vb_spatial_more.hip
    }
}
} // namespace fvb
