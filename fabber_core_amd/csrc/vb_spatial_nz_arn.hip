// Spatial VB under the general AR(1) noise model and with 5 - 8 noise precisions: which translation unit holds the kernels of a (model, P)
#include "vb_spatial_noise.h"

namespace fvb
{
SpatialKernels get_spatial_kernels_nz_arn_linear2(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_arn_linear3(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_arn_linear4(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_arn_poly(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_arn_exp2(int P, bool need_f, int kind);
SpatialKernels get_spatial_kernels_nz_arn_exp4(int P, bool need_f, int kind);

SpatialKernels get_spatial_kernels_nz_arn(int model, int P, bool need_f, int kind)
{
    switch (model)
    {
    case FVB_MODEL_LINEAR:
        return P == 2 ? get_spatial_kernels_nz_arn_linear2(P, need_f, kind)
                      : (P == 3 ? get_spatial_kernels_nz_arn_linear3(P, need_f, kind) : get_spatial_kernels_nz_arn_linear4(P, need_f, kind));
    case FVB_MODEL_POLY:
        return get_spatial_kernels_nz_arn_poly(P, need_f, kind);
    case FVB_MODEL_EXP:
        return P == 2 ? get_spatial_kernels_nz_arn_exp2(P, need_f, kind) : get_spatial_kernels_nz_arn_exp4(P, need_f, kind);
    default:
        return SpatialKernels{};
    }
}

SpatialKernels get_spatial_kernels_nz_p8_linear(int P, bool need_f);
SpatialKernels get_spatial_kernels_nz_p8_poly(int P, bool need_f);
SpatialKernels get_spatial_kernels_nz_p8_exp(int P, bool need_f);
SpatialKernels get_spatial_kernels_nz_pattern8(int model, int P, bool need_f)
{
    switch (model)
    {
    case FVB_MODEL_LINEAR:
        return get_spatial_kernels_nz_p8_linear(P, need_f);
    case FVB_MODEL_POLY:
        return get_spatial_kernels_nz_p8_poly(P, need_f);
    case FVB_MODEL_EXP:
        return get_spatial_kernels_nz_p8_exp(P, need_f);
    default:
        return SpatialKernels{};
    }
}
} // namespace fvb
