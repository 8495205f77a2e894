// AR(1) noise with two echoes (ar1-cross-terms none / same / dual): instantiations of the lane-per-voxel kernel,
// polynomial and exponential models
#include "vb_dispatch.h"
#include "vb_lane_arn_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_arn_kernel_poly(int P, int n_alphas, bool need_f)
{
    switch (P)
    {
        FVB_LANE_ARN_CASE(PolyModel, "poly", 2)
        FVB_LANE_ARN_CASE(PolyModel, "poly", 3)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
LaneKernelInfo get_lane_arn_kernel_exp(int P, int n_alphas, bool need_f)
{
    switch (P)
    {
        FVB_LANE_ARN_CASE(ExpModel, "exp", 2)
        FVB_LANE_ARN_CASE(ExpModel, "exp", 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
