// Instantiations of the several-precisions lane kernel (vb_lane_pattern_kernel.h) for the multi-exponential model, 4 moment sets
#include "vb_dispatch.h"
#include "vb_lane_pattern_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_pattern_kernel_exp_4(int P)
{
    switch (P)
    {
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 2, 4)
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 4, 4)
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 6, 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
