// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access width the renderer's record columns
// use: 8 bytes per lane, 512 contiguous bytes per wave access.  Streams a buffer far larger than the
// 256 MiB Infinity Cache once through a read kernel and once through a write kernel, so the counters can
// be compared with a known byte count:
//     hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/bin/calib_fetch
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/bin/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_read8(const double* __restrict__ a, size_t n, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 12345.678) out[0] = acc;  // never true for the zero-filled input; keeps the loads alive
}

__global__ __launch_bounds__(256) void calib_write8(double* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = 1.0;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(double);
    double *a = nullptr, *out = nullptr;
    CHECK(hipMalloc((void**)&a, bytes));
    CHECK(hipMalloc((void**)&out, 256));
    CHECK(hipMemset(a, 0, bytes));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, n, out);
        hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
    }
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    std::printf("calib_read8 and calib_write8: %zu bytes per launch, 3 launches each\n", bytes);
    CHECK(hipFree(a));
    CHECK(hipFree(out));
    return 0;
}
