// Gather-rate microbenchmarks behind the K = 1 SpMV design: how many 8-byte gathers per second does an
// MI355X sustain (a) from global memory through L1/L2 (table sizes of the three configs, plain / nontemporal / sc1 loads),
// (b) from LDS (ds_read_b64 at random addresses), (c) in the shape of the blocked SpMV's inner loop (coalesced 16-bit
// local indices from global, gather from LDS, one dependent fp64 add per entry), and (d) how fast a workgroup refills a
// 128 KB LDS block from L2.   hipcc -O3 --offload-arch=gfx950 tools/ubench_gather.hip -o build/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// skew 0: uniform; 2 / 3: product of two / three uniforms (the synthetic generator's user / item popularity)
__device__ __forceinline__ uint32_t draw(uint64_t h, uint32_t size, int skew)
{
    uint64_t u = splitmix(h);
    if (skew >= 2) u = __umul64hi(u, splitmix(h ^ 0x1111));
    if (skew >= 3) u = __umul64hi(u, splitmix(h ^ 0x2222));
    return (uint32_t)__umul64hi(u, (uint64_t)size);
}
__global__ void k_make_idx(uint32_t *idx, int64_t count, uint32_t size, int skew, uint64_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) idx[i] = draw(seed + (uint64_t)i, size, skew);
}
__global__ void k_make_idx16(uint16_t *idx, int64_t count, uint32_t size, int skew, uint64_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) idx[i] = (uint16_t)draw(seed + (uint64_t)i, size, skew);
}
__global__ void k_fill(double *t, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = 1.0 + (double)(i & 1023) * 0x1p-20;
}

// (a) global gathers: idx laid out [step][thread] (coalesced), U gathers in flight per lane, dependent adds in order
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_gather_global(const double *__restrict__ tab, const uint32_t *__restrict__ idx,
                                                       int steps, double *__restrict__ out)
{
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int p = 0; p < steps; p += U) {
        uint32_t ix[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ix[u] = idx[(int64_t)(p + u) * nthr + tid];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) v[u] = tab[ix[u]];
            else if (MODE == 1) v[u] = __builtin_nontemporal_load(tab + ix[u]);
            else v[u] = __hip_atomic_load(tab + ix[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[tid] = acc;
}

// (b, c) LDS gathers: a 1024-thread workgroup per CU holds BLK doubles of the table in LDS; 16-bit local indices are
// streamed from global [step][thread]; one dependent add per entry
template <int BLK, int U>
__global__ __launch_bounds__(1024) void k_gather_lds(const double *__restrict__ tab, const uint16_t *__restrict__ idx,
                                                     int steps, double *__restrict__ out)
{
    extern __shared__ double zs[];
    for (int i = threadIdx.x; i < BLK; i += blockDim.x) zs[i] = tab[i];
    __syncthreads();
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int p = 0; p < steps; p += U) {
        uint16_t ix[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) ix[u] = idx[(int64_t)(p + u) * nthr + tid];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = zs[ix[u] & (BLK - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[tid] = acc;
}
// the same with the indices packed four to a 64-bit load (one 8-byte load per lane per four entries)
template <int BLK>
__global__ __launch_bounds__(1024) void k_gather_lds_packed(const double *__restrict__ tab, const uint2 *__restrict__ idx4,
                                                            int steps4, double *__restrict__ out)
{
    extern __shared__ double zs[];
    for (int i = threadIdx.x; i < BLK; i += blockDim.x) zs[i] = tab[i];
    __syncthreads();
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int p = 0; p < steps4; p += 2) {
        const uint2 a = idx4[(int64_t)p * nthr + tid];
        const uint2 b = idx4[(int64_t)(p + 1) * nthr + tid];
        const double v0 = zs[a.x & (BLK - 1)], v1 = zs[(a.x >> 16) & (BLK - 1)];
        const double v2 = zs[a.y & (BLK - 1)], v3 = zs[(a.y >> 16) & (BLK - 1)];
        const double v4 = zs[b.x & (BLK - 1)], v5 = zs[(b.x >> 16) & (BLK - 1)];
        const double v6 = zs[b.y & (BLK - 1)], v7 = zs[(b.y >> 16) & (BLK - 1)];
        acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
    }
    out[tid] = acc;
}

// (d) LDS block refill: every workgroup copies `blocks` consecutive BLK-double blocks of the table into LDS, one after
// the other, with a barrier per block (no compute): the price of a sweep step's refill alone
template <int BLK>
__global__ __launch_bounds__(1024) void k_refill(const double *__restrict__ tab, int blocks, double *__restrict__ out)
{
    extern __shared__ double zs[];
    typedef double v2d __attribute__((ext_vector_type(2)));
    double acc = 0.0;
    for (int b = 0; b < blocks; ++b) {
        const v2d *src = reinterpret_cast<const v2d *>(tab + (size_t)b * BLK);
        v2d *dstp = reinterpret_cast<v2d *>(zs);
        for (int i = threadIdx.x; i < BLK / 2; i += blockDim.x) dstp[i] = src[i];
        __syncthreads();
        acc += zs[(threadIdx.x * 7 + b) & (BLK - 1)];
        __syncthreads();
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int CUS = 256;
    double *out;
    CK(hipMalloc(&out, (size_t)CUS * 2048 * 8 * sizeof(double)));
    // ---------------- (a) global gathers
    {
        const int64_t nthr = (int64_t)CUS * 2048;   // every wave slot of the chip
        const int steps = 128;
        const int64_t cnt = nthr * steps;
        uint32_t *idx;
        CK(hipMalloc(&idx, cnt * sizeof(uint32_t)));
        const uint32_t sizes[] = {224000u, 600000u, 6000000u};   // doubles: C3 1.8 MB, C2 4.8 MB, C4 48 MB
        for (uint32_t size : sizes) {
            double *tab;
            CK(hipMalloc(&tab, (size_t)size * sizeof(double)));
            hipLaunchKernelGGL(k_fill, dim3((size + 255) / 256), dim3(256), 0, 0, tab, (int64_t)size);
            for (int skew : {0, 3}) {
                hipLaunchKernelGGL(k_make_idx, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, idx, cnt, size, skew, 12345ull);
                for (int mode = 0; mode < 3; ++mode) {
                    float best = 1e30f;
                    for (int rep = 0; rep < 4; ++rep) {
                        CK(hipEventRecord(e0, 0));
                        if (mode == 0) hipLaunchKernelGGL((k_gather_global<0, 8>), dim3((unsigned)(nthr / 256)), dim3(256), 0, 0, tab, idx, steps, out);
                        else if (mode == 1) hipLaunchKernelGGL((k_gather_global<1, 8>), dim3((unsigned)(nthr / 256)), dim3(256), 0, 0, tab, idx, steps, out);
                        else hipLaunchKernelGGL((k_gather_global<2, 8>), dim3((unsigned)(nthr / 256)), dim3(256), 0, 0, tab, idx, steps, out);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        const float ms = time_ms(e0, e1);
                        if (ms < best) best = ms;
                    }
                    printf("{\"test\": \"global_gather8\", \"table_doubles\": %u, \"skew\": %d, \"mode\": \"%s\", \"gathers\": %lld, \"ms\": %.4f, \"Ggather_s\": %.1f}\n",
                           size, skew, mode == 0 ? "plain" : mode == 1 ? "nt" : "sc1", (long long)cnt, best, cnt / (best * 1e6));
                    fflush(stdout);
                }
            }
            CK(hipFree(tab));
        }
        CK(hipFree(idx));
    }
    // ---------------- (b, c) LDS gathers
    {
        constexpr int BLK = 16384;
        const int64_t nthr = (int64_t)CUS * 1024;
        const int steps = 256;
        const int64_t cnt = nthr * steps;
        uint16_t *idx;
        CK(hipMalloc(&idx, cnt * sizeof(uint16_t)));
        double *tab;
        CK(hipMalloc(&tab, (size_t)BLK * sizeof(double)));
        hipLaunchKernelGGL(k_fill, dim3((BLK + 255) / 256), dim3(256), 0, 0, tab, (int64_t)BLK);
        CK(hipFuncSetAttribute((const void *)(k_gather_lds<BLK, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, BLK * 8));
        CK(hipFuncSetAttribute((const void *)(k_gather_lds<BLK, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, BLK * 8));
        CK(hipFuncSetAttribute((const void *)(k_gather_lds_packed<BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, BLK * 8));
        for (int skew : {0, 3}) {
            hipLaunchKernelGGL(k_make_idx16, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, idx, cnt, (uint32_t)BLK, skew, 777ull);
            for (int var = 0; var < 3; ++var) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (var == 0) hipLaunchKernelGGL((k_gather_lds<BLK, 8>), dim3(CUS), dim3(1024), BLK * 8, 0, tab, idx, steps, out);
                    else if (var == 1) hipLaunchKernelGGL((k_gather_lds<BLK, 16>), dim3(CUS), dim3(1024), BLK * 8, 0, tab, idx, steps, out);
                    else hipLaunchKernelGGL((k_gather_lds_packed<BLK>), dim3(CUS), dim3(1024), BLK * 8, 0, tab, (const uint2 *)idx, steps / 4, out);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    const float ms = time_ms(e0, e1);
                    if (ms < best) best = ms;
                }
                printf("{\"test\": \"lds_gather8\", \"block_doubles\": %d, \"skew\": %d, \"variant\": \"%s\", \"gathers\": %lld, \"ms\": %.4f, \"Ggather_s\": %.1f, \"index_GBps\": %.0f}\n",
                       BLK, skew, var == 0 ? "u16 x8" : var == 1 ? "u16 x16" : "packed 4 per 8 B", (long long)cnt, best, cnt / (best * 1e6), cnt * 2 / (best * 1e6));
                fflush(stdout);
            }
        }
        // ---------------- (d) refill
        {
            double *big;
            const int blocks = 14;
            CK(hipMalloc(&big, (size_t)BLK * blocks * sizeof(double)));
            hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)BLK * blocks + 255) / 256)), dim3(256), 0, 0, big, (int64_t)BLK * blocks);
            CK(hipFuncSetAttribute((const void *)(k_refill<BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, BLK * 8));
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_refill<BLK>), dim3(CUS), dim3(1024), BLK * 8, 0, big, blocks, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                const float ms = time_ms(e0, e1);
                if (ms < best) best = ms;
            }
            printf("{\"test\": \"lds_refill\", \"block_doubles\": %d, \"blocks\": %d, \"ms\": %.4f, \"us_per_block\": %.2f, \"chip_TBps\": %.2f}\n", BLK, blocks,
                   best, best * 1e3 / blocks, (double)CUS * blocks * BLK * 8 / (best * 1e9));
            CK(hipFree(big));
        }
        CK(hipFree(idx));
        CK(hipFree(tab));
    }
    return 0;
}
