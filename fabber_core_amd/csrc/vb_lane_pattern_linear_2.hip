// Instantiations of the several-precisions lane kernel (vb_lane_pattern_kernel.h) for the design-matrix model (fwdmodel_linear.cc), 2 moment sets
#include "vb_dispatch.h"
#include "vb_lane_pattern_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_pattern_kernel_linear_2(int P)
{
    switch (P)
    {
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 1, 2)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 2, 2)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 3, 2)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 4, 2)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 5, 2)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 6, 2)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
