// micro-benchmark: latency of a dependent v_add_f64 chain on gfx950 (bounds the exact seed-row chain)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double *out, const double *in, int n, long long *cyc)
{
    double a = in[threadIdx.x];
    double b = in[64 + threadIdx.x];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a += b;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *in, *out; long long *cyc;
    hipMalloc(&in, 1024 * 8); hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + i * 1e-9;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads : {64, 256, 1024}) {
        int n = 100000;
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, in, n, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %4d: %.2f memtime-ticks per dependent v_add_f64 (s_memtime runs at 100MHz? see ratio)\n", threads, (double)c / (16.0 * n));
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, in, n, cyc); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("   wall %.3f ms -> %.2f ns per add\n", ms, ms * 1e6 / (16.0 * n));
    }
    return 0;
}
