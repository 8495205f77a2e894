// Instantiations of the several-precisions lane kernel (vb_lane_pattern_kernel.h) for the polynomial model (fwdmodel_poly.cc), 2 moment sets
#include "vb_dispatch.h"
#include "vb_lane_pattern_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_pattern_kernel_poly_2(int P)
{
    switch (P)
    {
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 1, 2)
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 2, 2)
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 3, 2)
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 4, 2)
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 5, 2)
        FVB_LANE_PATTERN_CASE(PolyModel, "poly", 6, 2)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
