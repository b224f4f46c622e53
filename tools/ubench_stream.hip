// How fast can ONE wave stream a private contiguous region (the sweep SpMV's access pattern), as a function of the loads it
// keeps in flight?  Each wave reads `rows` word-rows of 64 lanes x W bytes from its own region with D loads outstanding.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o build/ubench_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int D>
__global__ __launch_bounds__(1024) void k_stream(const T *__restrict__ data, int rows, size_t region_words, unsigned long long *out)
{
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const T *p = data + gw * region_words + lane;
    T buf[D];
    unsigned long long acc = 0;
#pragma unroll
    for (int u = 0; u < D; ++u) buf[u] = p[(size_t)u * 64];
    for (int r = 0; r < rows; r += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const T v = buf[u];
            int rn = r + D + u;
            rn = rn < rows ? rn : rows - 1;
            buf[u] = p[(size_t)rn * 64];
            acc += (unsigned long long)v.x;
        }
    }
    out[gw * 64 + lane] = acc;
}

template <typename T, int D>
static void run(const char *name, int wgs, int threads, int rows, void *data, unsigned long long *out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t region_words = (size_t)rows * 64;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_stream<T, D>), dim3(wgs), dim3(threads), 0, 0, (const T *)data, rows, region_words, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = (double)wgs * threads / 64, bytes = waves * rows * 64 * sizeof(T);
    printf("{\"test\": \"wave_stream\", \"load\": \"%s\", \"in_flight\": %d, \"waves_per_cu\": %.1f, \"KB_per_wave\": %.0f, \"us\": %.1f, \"GBps_per_wave\": %.2f, \"chip_TBps\": %.2f}\n",
           name, D, waves / 256, rows * 64.0 * sizeof(T) / 1024, best * 1e3, bytes / waves / (best * 1e6), bytes / (best * 1e9));
    fflush(stdout);
}

int main()
{
    void *data; unsigned long long *out;
    const size_t bytes = 1ull << 30;
    CK(hipMalloc(&data, bytes));
    CK(hipMemset(data, 1, bytes));
    CK(hipMalloc(&out, 256 * 16 * 64 * 8));
    // one wave per CU, 292 KB each (the heaviest wave of the MovieLens-shaped sweep), then 16 waves per CU, 64 KB each
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int threads = cfg == 0 ? 64 : 1024;
        const int kb = cfg == 0 ? 292 : 64;
        const int rows8 = kb * 1024 / 512, rows16 = kb * 1024 / 1024;
        run<uint2, 4>("8B", 256, threads, rows8, data, out);
        run<uint2, 8>("8B", 256, threads, rows8, data, out);
        run<uint2, 16>("8B", 256, threads, rows8, data, out);
        run<uint2, 24>("8B", 256, threads, rows8, data, out);
        run<uint2, 32>("8B", 256, threads, rows8, data, out);
        run<uint4, 4>("16B", 256, threads, rows16, data, out);
        run<uint4, 8>("16B", 256, threads, rows16, data, out);
        run<uint4, 16>("16B", 256, threads, rows16, data, out);
        run<uint4, 24>("16B", 256, threads, rows16, data, out);
    }
    return 0;
}
