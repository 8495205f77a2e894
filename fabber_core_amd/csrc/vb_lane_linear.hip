// Instantiations of the lane-per-voxel kernel for the design-matrix model (fwdmodel_linear.cc)
#include "vb_dispatch.h"

namespace fvb
{
LaneKernelInfo get_lane_kernel_linear(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_CASE(LinearModel, "linear", 1)
        FVB_LANE_CASE(LinearModel, "linear", 2)
        FVB_LANE_CASE(LinearModel, "linear", 3)
        FVB_LANE_CASE(LinearModel, "linear", 4)
        FVB_LANE_CASE(LinearModel, "linear", 5)
        FVB_LANE_CASE(LinearModel, "linear", 6)
    default:
        return get_lane_kernel_wide(FVB_MODEL_LINEAR, P, need_f); // 7 and 8 parameters: vb_lane_wide.hip
    }
}
} // namespace fvb
