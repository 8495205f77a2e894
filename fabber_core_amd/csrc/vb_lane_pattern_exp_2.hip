// Instantiations of the several-precisions lane kernel (vb_lane_pattern_kernel.h) for the multi-exponential model, 2 moment sets
#include "vb_dispatch.h"
#include "vb_lane_pattern_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_pattern_kernel_exp_2(int P)
{
    switch (P)
    {
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 2, 2)
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 4, 2)
        FVB_LANE_PATTERN_CASE(ExpModel, "exp", 6, 2)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
