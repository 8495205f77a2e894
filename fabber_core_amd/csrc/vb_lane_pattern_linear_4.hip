// Instantiations of the several-precisions lane kernel (vb_lane_pattern_kernel.h) for the design-matrix model (fwdmodel_linear.cc), 4 moment sets
#include "vb_dispatch.h"
#include "vb_lane_pattern_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_pattern_kernel_linear_4(int P)
{
    switch (P)
    {
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 1, 4)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 2, 4)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 3, 4)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 4, 4)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 5, 4)
        FVB_LANE_PATTERN_CASE(LinearModel, "linear", 6, 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
