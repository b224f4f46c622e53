// The whole LSD radix sort of a small array by ONE 1024-thread workgroup (sort.hip: k_sort_small; build.hip: the one-launch
// graph build of ego-network-sized graphs calls it four times from inside its own kernel).
#pragma once
#include "common.h"

namespace rwr {

constexpr int SMALL_SORT_RADIX = 256;
constexpr int SMALL_SORT_THREADS = 1024;
// (one workgroup needs ~1.5 us per 1024 keys and pass, the general path three launches ~ 50 us per pass whatever the
//  size: break-even near 32 K keys)
constexpr size_t SMALL_SORT_MAX = 20480;

// LDS the sort needs (declared by the calling kernel so that several stages can share one workgroup)
struct SmallSortLds {
    uint32_t single;                 // this pass's digit is the same for every key: the pass is a plain copy
    uint32_t wcnt[SMALL_SORT_THREADS / WAVE][SMALL_SORT_RADIX];   // per (wave, digit): count, then the wave's next output place
    uint32_t wsum[SMALL_SORT_RADIX / WAVE];
};

// Stable sort of (key, u32 payload) pairs on key bits [0, key_bits).  Every wave owns ONE CONTIGUOUS RANGE of the array (a
// multiple of 64 keys), so that a pass needs workgroup barriers only around its three phases, none per tile: (1) each wave
// counts the digits of its range (LDS atomics on its own row of wcnt); (2) 256 threads turn the (wave, digit) counts into
// output places -- digit-major, wave-minor, the order a stable sort keeps; (3) each wave scatters its range 64 keys at a
// time, ranking equal digits inside the step with wave ballots and advancing its own row of places.  Every thread of the
// 1024-thread workgroup must call it.  The result lands in (ka, va) when the number of passes is even, in (kb, vb) when
// it is odd.
template <typename KeyT>
__device__ __forceinline__ void sort_small_body(SmallSortLds &L, KeyT *ka, KeyT *kb, uint32_t *va, uint32_t *vb, uint32_t m, int key_bits)
{
    constexpr int RADIX = SMALL_SORT_RADIX;
    constexpr int NW = SMALL_SORT_THREADS / WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const uint32_t per = (((m + NW - 1) / NW) + WAVE - 1) / WAVE * WAVE;    // keys per wave
    const uint32_t lo = min(m, (uint32_t)wv * per), hi = min(m, lo + per);
    KeyT *kin = ka, *kout = kb;
    uint32_t *vin = va, *vout = vb;
    for (int shift = 0; shift < key_bits; shift += 8) {
        if (tid == 0) L.single = 0;
        for (int i = tid; i < NW * RADIX; i += SMALL_SORT_THREADS) (&L.wcnt[0][0])[i] = 0;
        __syncthreads();
        for (uint32_t i = lo + lane; i < hi; i += WAVE) atomicAdd(&L.wcnt[wv][(unsigned)(kin[i] >> shift) & (RADIX - 1)], 1u);
        __syncthreads();
        // thread t < 256 owns digit t: its count over the waves, the exclusive scan over the digits (waves 0-3), then the
        // place of every wave's first key of that digit
        uint32_t v = 0, incl = 0;
        if (tid < RADIX) {
#pragma unroll
            for (int q = 0; q < NW; ++q) v += L.wcnt[q][tid];
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
            uint32_t at = incl - v;
            for (int q = 0; q < wv; ++q) at += L.wsum[q];
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const uint32_t c = L.wcnt[q][tid];
                L.wcnt[q][tid] = at;
                at += c;
            }
        }
        __syncthreads();
        if (L.single) {                        // a stable sort on a constant digit keeps the order: copy (the buffers still alternate)
            for (uint32_t i = tid; i < m; i += SMALL_SORT_THREADS) { kout[i] = kin[i]; vout[i] = vin[i]; }
        } else {
            // stable scatter: the wave walks its range in index order; no other wave touches its row of places
            for (uint32_t base = lo; base < hi; base += WAVE) {
                const uint32_t idx = base + lane;
                const bool valid = idx < hi;
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
                volatile uint32_t *places = &L.wcnt[wv][0];
                const uint32_t at = valid ? places[digit] : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();                       // (every lane has read its place before a leader moves it on)
                if (valid && rank_in_wave == 0) places[digit] = at + (uint32_t)__popcll(peers);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (valid) {
                    kout[at + rank_in_wave] = key;
                    vout[at + rank_in_wave] = val;
                }
            }
        }
        __syncthreads();
        { KeyT *t = kin; kin = kout; kout = t; }
        { uint32_t *t = vin; vin = vout; vout = t; }
    }
}

}  // namespace rwr
