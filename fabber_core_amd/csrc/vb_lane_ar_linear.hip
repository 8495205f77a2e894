// AR(1)-noise instantiations of the lane-per-voxel kernel, linear model
#include "vb_dispatch.h"
#include "vb_lane_ar_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_ar_kernel_linear(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_AR_CASE(LinearModel, "linear", 1)
        FVB_LANE_AR_CASE(LinearModel, "linear", 2)
        FVB_LANE_AR_CASE(LinearModel, "linear", 3)
        FVB_LANE_AR_CASE(LinearModel, "linear", 4)
        FVB_LANE_AR_CASE(LinearModel, "linear", 5)
        FVB_LANE_AR_CASE(LinearModel, "linear", 6)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
