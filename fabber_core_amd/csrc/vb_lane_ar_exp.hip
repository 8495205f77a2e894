// AR(1)-noise instantiations of the lane-per-voxel kernel, exp model
#include "vb_dispatch.h"
#include "vb_lane_ar_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_ar_kernel_exp(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_AR_CASE(ExpModel, "exp", 2)
        FVB_LANE_AR_CASE(ExpModel, "exp", 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
