/* vb_dispatch.h - lookup of the templated lane kernels, one translation unit per model and noise
 * model so that the instantiations compile in parallel. */
#pragma once

#include "vb_lane_kernel.h"

namespace fvb
{
typedef void (*LaneKernelFn)(const KernelArgs);

struct LaneKernelInfo
{
    LaneKernelFn fn; // reads the caller's [t][voxel] image in place (FEED_STRIDED: any element type, masked timepoints);
                     // NULL for the AR(1) kernels, which only exist for the tiled series
    int save_rows;
    const char *name;
    // the same loop fed from the tiled series (vb_lane_kernel.h), float / double elements - what runs unless
    // timepoints are masked
    LaneKernelFn fn_tiles_f32, fn_tiles_f64;
    // the tile-fed kernels of a run whose detector only counts iterations (convergence = maxits: the reference's
    // default) - F evaluated, nothing watching it; NULL: take the ones above
    LaneKernelFn fn_tiles_f32_counting = nullptr, fn_tiles_f64_counting = nullptr;
};

// Each returns {NULL,0,NULL} when (P, need_f) has no instantiation; the caller then falls back to
// the generic wave-per-voxel kernel.
LaneKernelInfo get_lane_kernel_poly(int P, bool need_f);
LaneKernelInfo get_lane_kernel_linear(int P, bool need_f);
LaneKernelInfo get_lane_kernel_exp(int P, bool need_f);
LaneKernelInfo get_lane_kernel_wide(int model, int P, bool need_f); // P = 7, 8 of any of the three
// AR(1) noise (vb_lane_ar_kernel.h)
LaneKernelInfo get_lane_ar_kernel_poly(int P, bool need_f);
LaneKernelInfo get_lane_ar_kernel_linear(int P, bool need_f);
LaneKernelInfo get_lane_ar_kernel_exp(int P, bool need_f);
// AR(1) noise, two echoes, 2 / 3 / 4 AR coefficients (vb_lane_arn_kernel.h)
LaneKernelInfo get_lane_arn_kernel_poly(int P, int n_alphas, bool need_f);
LaneKernelInfo get_lane_arn_kernel_linear(int P, int n_alphas, bool need_f);
LaneKernelInfo get_lane_arn_kernel_exp(int P, int n_alphas, bool need_f);
// white noise with several precisions (vb_lane_pattern_kernel.h): N = 2 or 4 moment sets, n_phis <= N used
LaneKernelInfo get_lane_pattern_kernel_poly_2(int P);
LaneKernelInfo get_lane_pattern_kernel_poly_4(int P);
LaneKernelInfo get_lane_pattern_kernel_linear_2(int P);
LaneKernelInfo get_lane_pattern_kernel_linear_4(int P);
LaneKernelInfo get_lane_pattern_kernel_exp_2(int P);
LaneKernelInfo get_lane_pattern_kernel_exp_4(int P);
} // namespace fvb

#define FVB_LANE_PATTERN_CASE(MODEL, TAG, PP, NN)                                                            \
    case PP:                                                                                                 \
        return LaneKernelInfo{ vb_lane_pattern_kernel<MODEL<PP>, PP, NN>, lane_pattern_save_rows<PP, NN>(),  \
            "lane_phis<" TAG "," #PP "," #NN ">", nullptr, nullptr };

#define FVB_LANE_CASE(MODEL, TAG, PP)                                                                        \
    case PP:                                                                                                 \
        if (need_f)                                                                                          \
            return LaneKernelInfo{ vb_lane_kernel<MODEL<PP>, PP, true, FEED_STRIDED>, lane_save_rows<PP>(),  \
                "lane<" TAG "," #PP ",F>", vb_lane_kernel<MODEL<PP>, PP, true, FEED_TILES_F32>,              \
                vb_lane_kernel<MODEL<PP>, PP, true, FEED_TILES_F64>,                                         \
                vb_lane_kernel<MODEL<PP>, PP, true, FEED_TILES_F32, false>,                                  \
                vb_lane_kernel<MODEL<PP>, PP, true, FEED_TILES_F64, false> };                                \
        return LaneKernelInfo{ vb_lane_kernel<MODEL<PP>, PP, false, FEED_STRIDED>, lane_save_rows<PP>(),     \
            "lane<" TAG "," #PP ">", vb_lane_kernel<MODEL<PP>, PP, false, FEED_TILES_F32>,                   \
            vb_lane_kernel<MODEL<PP>, PP, false, FEED_TILES_F64> };

#define FVB_LANE_AR_CASE(MODEL, TAG, PP)                                                                     \
    case PP:                                                                                                 \
        if (need_f)                                                                                          \
            return LaneKernelInfo{ nullptr, lane_ar_save_rows<PP>(), "lane_ar1<" TAG "," #PP ",F>",          \
                vb_lane_ar_kernel<MODEL<PP>, PP, true, FEED_TILES_F32>,                                      \
                vb_lane_ar_kernel<MODEL<PP>, PP, true, FEED_TILES_F64> };                                    \
        return LaneKernelInfo{ nullptr, lane_ar_save_rows<PP>(), "lane_ar1<" TAG "," #PP ">",                \
            vb_lane_ar_kernel<MODEL<PP>, PP, false, FEED_TILES_F32>,                                         \
            vb_lane_ar_kernel<MODEL<PP>, PP, false, FEED_TILES_F64> };

#define FVB_LANE_ARN_ONE(MODEL, TAG, PP, NA, FLAG, SUFFIX)                                                   \
    return LaneKernelInfo{ nullptr, lane_arn_save_rows<PP, NA>(), "lane_ar2<" TAG "," #PP "," #NA SUFFIX ">",  \
        vb_lane_arn_kernel<MODEL<PP>, PP, NA, FLAG, FEED_TILES_F32>, vb_lane_arn_kernel<MODEL<PP>, PP, NA, FLAG, FEED_TILES_F64> };
#define FVB_LANE_ARN_CASE(MODEL, TAG, PP)                                                                    \
    case PP:                                                                                                 \
        if (n_alphas == 2)                                                                                   \
        {                                                                                                    \
            if (need_f)                                                                                      \
                FVB_LANE_ARN_ONE(MODEL, TAG, PP, 2, true, ",F")                                              \
            FVB_LANE_ARN_ONE(MODEL, TAG, PP, 2, false, "")                                                   \
        }                                                                                                    \
        if (n_alphas == 3)                                                                                   \
        {                                                                                                    \
            if (need_f)                                                                                      \
                FVB_LANE_ARN_ONE(MODEL, TAG, PP, 3, true, ",F")                                              \
            FVB_LANE_ARN_ONE(MODEL, TAG, PP, 3, false, "")                                                   \
        }                                                                                                    \
        if (n_alphas == 4)                                                                                   \
        {                                                                                                    \
            if (need_f)                                                                                      \
                FVB_LANE_ARN_ONE(MODEL, TAG, PP, 4, true, ",F")                                              \
            FVB_LANE_ARN_ONE(MODEL, TAG, PP, 4, false, "")                                                   \
        }                                                                                                    \
        return LaneKernelInfo{ nullptr, 0, nullptr };
