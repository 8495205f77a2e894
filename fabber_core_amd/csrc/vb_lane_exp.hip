// Instantiations of the lane-per-voxel kernel for the multi-exponential model
// (examples/fwdmodel_exp.cc)
#include "vb_dispatch.h"

namespace fvb
{
LaneKernelInfo get_lane_kernel_exp(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_CASE(ExpModel, "exp", 2)
        FVB_LANE_CASE(ExpModel, "exp", 4)
        FVB_LANE_CASE(ExpModel, "exp", 6)
    default:
        return get_lane_kernel_wide(FVB_MODEL_EXP, P, need_f); // 7 and 8 parameters: vb_lane_wide.hip
    }
}
} // namespace fvb
