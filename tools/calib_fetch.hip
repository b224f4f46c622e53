// Calibration of rocprofv3 FETCH_SIZE on gfx950 for THIS project's access widths (the guide only calibrates
// 16 B/lane streams): (a) 8 B/lane coalesced stream, (b) random 128-byte row gathers by 16-lane groups (the SpMM's
// access), each over a buffer far larger than the 256 MiB Infinity Cache, with a known number of bytes touched.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void stream8(const double *__restrict__ a, double *__restrict__ out, size_t n)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void stream16(const double2 *__restrict__ a, double *__restrict__ out, size_t n)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ void gather128(const double *__restrict__ tab, size_t rows, double *__restrict__ out, size_t ngather)
{
    // thread group of 16 lanes reads one random 128-byte row per step
    size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16;
    int k = threadIdx.x & 15;
    size_t ngroups = (size_t)gridDim.x * blockDim.x / 16;
    double s = 0;
    for (size_t i = g; i < ngather; i += ngroups) {
        uint64_t h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        size_t r = h % rows;
        s += tab[r * 16 + k];
    }
    if (s == 12345.678) out[0] = s;
}
int main()
{
    size_t bytes = (size_t)8 << 30;   // 8 GiB
    double *a, *out;
    hipMalloc(&a, bytes); hipMalloc(&out, 64);
    hipMemset(a, 0, bytes);
    size_t n = bytes / 8;
    hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, a, out, n);
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const double2 *)a, out, n / 2);
    size_t rows = bytes / 128, ngather = (size_t)1 << 26;   // 64 Mi gathers of 128 B = 8 GiB touched
    hipLaunchKernelGGL(gather128, dim3(8192), dim3(256), 0, 0, a, rows, out, ngather);
    hipDeviceSynchronize();
    printf("known bytes: stream8 %zu  stream16 %zu  gather128 %zu\n", bytes, bytes, ngather * 128);
    return 0;
}
