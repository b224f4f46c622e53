// The whole LSD radix sort of a small array by ONE 1024-thread workgroup (sort.hip: k_sort_small; build.hip: the one-launch
// graph build of ego-network-sized graphs calls it four times from inside its own kernel).
#pragma once
#include "common.h"

namespace rwr {

constexpr int SMALL_SORT_RADIX = 256;
constexpr int SMALL_SORT_THREADS = 1024;
// (one workgroup needs ~1.5 us per tile of 1024 keys and pass, the general path three launches ~ 50 us per pass whatever the
//  size: break-even near 32 K keys)
constexpr size_t SMALL_SORT_MAX = 20480;

// LDS the sort needs (declared by the calling kernel so that several stages can share one workgroup)
struct SmallSortLds {
    uint32_t single;                 // this pass's digit is the same for every key: the pass is a plain copy
    uint32_t run[SMALL_SORT_RADIX];
    uint32_t wcnt[SMALL_SORT_THREADS / WAVE][SMALL_SORT_RADIX];
    uint32_t wsum[SMALL_SORT_RADIX / WAVE];
};

// Stable sort of (key, u32 payload) pairs on key bits [0, key_bits): per pass a digit histogram (LDS atomics), its exclusive
// scan, and the stable scatter tile by tile with wave-ballot ranking; workgroup barriers where the general path has three
// kernel launches per pass.  Every thread of the 1024-thread workgroup must call it.  The result lands in (ka, va) when the
// number of passes is even, in (kb, vb) when it is odd.
template <typename KeyT>
__device__ __forceinline__ void sort_small_body(SmallSortLds &L, KeyT *ka, KeyT *kb, uint32_t *va, uint32_t *vb, uint32_t m, int key_bits)
{
    constexpr int RADIX = SMALL_SORT_RADIX;
    constexpr int NW = SMALL_SORT_THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    KeyT *kin = ka, *kout = kb;
    uint32_t *vin = va, *vout = vb;
    for (int shift = 0; shift < key_bits; shift += 8) {
        if (tid < RADIX) L.run[tid] = 0;
        if (tid == 0) L.single = 0;
        for (int i = tid; i < NW * RADIX; i += SMALL_SORT_THREADS) (&L.wcnt[0][0])[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < m; i += SMALL_SORT_THREADS) atomicAdd(&L.run[(unsigned)(kin[i] >> shift) & (RADIX - 1)], 1u);
        __syncthreads();
        // exclusive scan of the 256 digit counts (waves 0-3)
        uint32_t v = 0, incl = 0;
        if (tid < RADIX) {
            v = L.run[tid];
            if (v == m) L.single = 1;          // (e.g. the sign / exponent byte of a score key: every key in one bin)
            incl = v;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, WAVE);
                if (lane >= off) incl += o;
            }
            if (lane == WAVE - 1) L.wsum[wv] = incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t pre = 0;
            for (int q = 0; q < wv; ++q) pre += L.wsum[q];
            L.run[tid] = pre + incl - v;
        }
        __syncthreads();
        if (L.single) {                        // a stable sort on a constant digit keeps the order: copy (the buffers still alternate)
            for (uint32_t i = tid; i < m; i += SMALL_SORT_THREADS) { kout[i] = kin[i]; vout[i] = vin[i]; }
            __syncthreads();
            { KeyT *t = kin; kin = kout; kout = t; }
            { uint32_t *t = vin; vin = vout; vout = t; }
            continue;
        }
        // stable scatter, tile by tile in index order
        for (uint32_t base = 0; base < m; base += SMALL_SORT_THREADS) {
            const uint32_t idx = base + tid;
            const bool valid = idx < m;
            const KeyT key = valid ? kin[idx] : (KeyT)0;
            const uint32_t val = valid ? vin[idx] : 0u;
            const unsigned digit = (unsigned)(key >> shift) & (RADIX - 1);
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (digit >> b) & 1u;
                const unsigned long long mb = __ballot(valid && bit);
                peers &= bit ? mb : ~mb;
            }
            const unsigned rank_in_wave = __popcll(peers & ((1ull << lane) - 1ull));
            if (valid && rank_in_wave == 0) L.wcnt[wv][digit] = __popcll(peers);
            __syncthreads();
            uint32_t pos = 0;
            if (valid) {
                pos = L.run[digit] + rank_in_wave;
                for (int q = 0; q < wv; ++q) pos += L.wcnt[q][digit];
            }
            __syncthreads();
            if (tid < RADIX) {
                uint32_t sacc = 0;
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    sacc += L.wcnt[q][tid];
                    L.wcnt[q][tid] = 0;
                }
                L.run[tid] += sacc;
            }
            if (valid) {
                kout[pos] = key;
                vout[pos] = val;
            }
            __syncthreads();
        }
        { KeyT *t = kin; kin = kout; kout = t; }
        { uint32_t *t = vin; vin = vout; vout = t; }
    }
}

}  // namespace rwr
