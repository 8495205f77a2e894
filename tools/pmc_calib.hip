// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths the VB kernels use
// (MI355X_MICROARCH.md, HBM section: widths other than 16 B/lane are uncalibrated). Each kernel
// moves a known number of bytes through a buffer much larger than the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// lane-per-column walk over a [rows][cols] image, as the lane kernel reads its [t][voxel] data
template <typename T>
__global__ __launch_bounds__(64) void read_rows(const T *src, size_t rows, size_t cols, double *sink)
{
    const size_t v = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (v >= cols)
        return;
    double acc = 0;
    for (size_t t = 0; t < rows; t++)
        acc += (double)src[t * cols + v];
    if (acc == 12345.678)
        sink[0] = acc;
}

template <typename T>
__global__ __launch_bounds__(64) void write_rows(T *dst, size_t rows, size_t cols)
{
    const size_t v = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (v >= cols)
        return;
    for (size_t t = 0; t < rows; t++)
        dst[t * cols + v] = (T)(t + v);
}

int main()
{
    const size_t cols = 1 << 20, rows = 512; // 2 GiB of float, 4 GiB of double
    void *buf;
    double *sink;
    CHECK(hipMalloc(&buf, rows * cols * sizeof(double)));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 0, rows * cols * sizeof(double)));
    const unsigned grid = (unsigned)(cols / 64);
    for (int rep = 0; rep < 2; rep++)
    {
        hipLaunchKernelGGL(read_rows<float>, dim3(grid), dim3(64), 0, 0, (const float *)buf, rows, cols, sink);
        hipLaunchKernelGGL(read_rows<double>, dim3(grid), dim3(64), 0, 0, (const double *)buf, rows, cols, sink);
        hipLaunchKernelGGL(write_rows<float>, dim3(grid), dim3(64), 0, 0, (float *)buf, rows, cols);
        hipLaunchKernelGGL(write_rows<double>, dim3(grid), dim3(64), 0, 0, (double *)buf, rows, cols);
    }
    CHECK(hipDeviceSynchronize());
    printf("expected bytes: read f32 %zu, read f64 %zu, write f32 %zu, write f64 %zu\n", rows * cols * 4, rows * cols * 8,
        rows * cols * 4, rows * cols * 8);
    return 0;
}
