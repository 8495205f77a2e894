/*
 * vb_host_copy.h - rows of a [row][voxel] HOST image to or from a device buffer (the host entry points' uploads and downloads)
 */
#pragma once

#include <hip/hip_runtime.h>

namespace fvb
{
// `rows` rows of `width` bytes, `spitch` / `dpitch` bytes apart: one contiguous copy where the rows touch, else one 2-D copy
inline hipError_t copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipMemcpyKind kind,
    hipStream_t stream)
{
    if (dpitch == width && spitch == width)
        return hipMemcpyAsync(dst, src, width * rows, kind, stream);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, stream);
}
} // namespace fvb
