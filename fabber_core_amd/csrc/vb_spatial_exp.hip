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
        return get_spatial_kernels_more(FVB_MODEL_EXP, P, need_f); // vb_spatial_more.hip
    }
}
} // namespace fvb
