// Instantiations of the lane-per-voxel kernel for 7 and 8 parameters (polynomial of degree 6 / 7, design
// matrices with 7 / 8 regressors, four exponentials). With 36 entries per packed matrix these kernels spill
// (the posterior alone is 80 doubles per lane) and run at a fraction of the small-P rate, but one lane per voxel
// still beats the wave-per-voxel kernel they used to fall back to by a wide margin at volume sizes
// (profiles/r2_lane_vs_wave_wide.jsonl).
#include "vb_dispatch.h"

namespace fvb
{
LaneKernelInfo get_lane_kernel_wide(int model, int P, bool need_f)
{
    if (model == FVB_MODEL_POLY)
        switch (P)
        {
            FVB_LANE_CASE(PolyModel, "poly", 7)
            FVB_LANE_CASE(PolyModel, "poly", 8)
        default:
            break;
        }
    if (model == FVB_MODEL_LINEAR)
        switch (P)
        {
            FVB_LANE_CASE(LinearModel, "linear", 7)
            FVB_LANE_CASE(LinearModel, "linear", 8)
        default:
            break;
        }
    if (model == FVB_MODEL_EXP)
        switch (P)
        {
            FVB_LANE_CASE(ExpModel, "exp", 8)
        default:
            break;
        }
    return LaneKernelInfo{ nullptr, 0, nullptr };
}
} // namespace fvb
