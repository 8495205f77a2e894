// Spatial-VB kernel instantiations, poly model
#include "vb_spatial.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_poly(int P, bool need_f)
{
    switch (P)
    {
        FVB_SPATIAL_CASE(PolyModel, "poly", 1)
        FVB_SPATIAL_CASE(PolyModel, "poly", 2)
        FVB_SPATIAL_CASE(PolyModel, "poly", 3)
        FVB_SPATIAL_CASE(PolyModel, "poly", 4)
    default:
        return get_spatial_kernels_more(FVB_MODEL_POLY, P, need_f); // vb_spatial_more.hip
    }
}
} // namespace fvb
