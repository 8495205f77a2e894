// AR(1)-noise instantiations of the lane-per-voxel kernel, poly model
#include "vb_dispatch.h"
#include "vb_lane_ar_kernel.h"

namespace fvb
{
LaneKernelInfo get_lane_ar_kernel_poly(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_AR_CASE(PolyModel, "poly", 1)
        FVB_LANE_AR_CASE(PolyModel, "poly", 2)
        FVB_LANE_AR_CASE(PolyModel, "poly", 3)
        FVB_LANE_AR_CASE(PolyModel, "poly", 4)
    default:
        return LaneKernelInfo{ nullptr, 0, nullptr };
    }
}
} // namespace fvb
