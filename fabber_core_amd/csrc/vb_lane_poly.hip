// Instantiations of the lane-per-voxel kernel for the polynomial model (fwdmodel_poly.cc)
#include "vb_dispatch.h"

namespace fvb
{
LaneKernelInfo get_lane_kernel_poly(int P, bool need_f)
{
    switch (P)
    {
        FVB_LANE_CASE(PolyModel, "poly", 1)
        FVB_LANE_CASE(PolyModel, "poly", 2)
        FVB_LANE_CASE(PolyModel, "poly", 3)
        FVB_LANE_CASE(PolyModel, "poly", 4)
        FVB_LANE_CASE(PolyModel, "poly", 5)
        FVB_LANE_CASE(PolyModel, "poly", 6)
    default:
        return get_lane_kernel_wide(FVB_MODEL_POLY, P, need_f); // 7 and 8 parameters: vb_lane_wide.hip
    }
}
} // namespace fvb
