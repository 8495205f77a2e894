/*
 * vb_host_copy.h - rows of a [row][voxel] HOST image to or from a device buffer (the host entry points' uploads and downloads)
 */
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

namespace fvb
{
// Rows of a [row][voxel] host image to or from a block's device buffer: ONE 2-D copy. FVB_COPY_ROWS=1 sends them row by
// row as 1-D transfers instead where the host side is page-locked (an experiment switch from the hunt for the wrong
// results under ROCm 7.2's runtime, which turned out to come from the stream-ordered memory pool - BlockSlot in
// vb_api.hip - and not from the copies: 2-D copies of pageable and of registered memory are right there too).
inline bool is_locked_host_memory(const void *p)
{
    static const bool rows_apart = getenv("FVB_COPY_ROWS") != nullptr;
    if (!rows_apart)
        return false;
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) != hipSuccess)
    {
        (void)hipGetLastError(); // (an ordinary pointer: not an error)
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}
inline hipError_t copy_rows(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows, hipMemcpyKind kind,
    hipStream_t stream, bool row_by_row)
{
    if (!row_by_row || rows <= 1 || (dpitch == width && spitch == width))
        return (dpitch == width && spitch == width) ? hipMemcpyAsync(dst, src, width * rows, kind, stream)
                                                    : hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, stream);
    for (size_t r = 0; r < rows; r++)
    {
        const hipError_t e = hipMemcpyAsync((char *)dst + r * dpitch, (const char *)src + r * spitch, width, kind, stream);
        if (e != hipSuccess)
            return e;
    }
    return hipSuccess;
}

} // namespace fvb
