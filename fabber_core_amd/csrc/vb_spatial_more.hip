// Spatial-VB kernel instantiations for the larger parameter counts of the built-in models: polynomials of degree
// 4 and 5, three exponentials, design matrices with 7 and 8 regressors (a translation unit of their own: they compile
// beside the others)
#include "vb_spatial.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_more(int model, int P, bool need_f)
{
    if (model == FVB_MODEL_POLY)
        switch (P)
        {
            FVB_SPATIAL_CASE(PolyModel, "poly", 5)
            FVB_SPATIAL_CASE(PolyModel, "poly", 6)
        default:
            break;
        }
    if (model == FVB_MODEL_EXP)
        switch (P)
        {
            FVB_SPATIAL_CASE(ExpModel, "exp", 6)
        default:
            break;
        }
    if (model == FVB_MODEL_LINEAR)
        switch (P)
        {
            FVB_SPATIAL_CASE(LinearModel, "linear", 7)
            FVB_SPATIAL_CASE(LinearModel, "linear", 8)
        default:
            break;
        }
    return SpatialKernels{};
}
} // namespace fvb
