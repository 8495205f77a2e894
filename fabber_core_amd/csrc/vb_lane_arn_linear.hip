// AR(1) noise with two echoes (ar1-cross-terms none / same / dual): instantiations of the lane-per-voxel kernel,
// linear model
#include "vb_dispatch.h"
#include "vb_lane_arn_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_arn_kernel_linear(int P, int n_alphas, bool need_f)
{
    switch (P)
    {
        FVB_LANE_ARN_CASE(LinearModel, "linear", 2)
        FVB_LANE_ARN_CASE(LinearModel, "linear", 3)
        FVB_LANE_ARN_CASE(LinearModel, "linear", 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
