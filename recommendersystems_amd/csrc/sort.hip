// Stable LSD radix sort (8-bit digits) for gfx950.  Used by the device-side graph build
// (transpose = stable sort of the link list by target, which keeps the reference's addend
// order (source asc, list position asc) -- Model.cs:78,85) and by the ranking stage
// (Recommender.cs:35-38: score desc, id desc).
//
// Three kernels per pass:  per-block digit histogram -> exclusive scan over (digit, block)
// -> stable scatter.  Stability inside a block comes from ranking equal digits with
// wave-wide ballots (64-lane match) and carrying per-digit running offsets across the
// block's 256-element sub-tiles in index order.  Everything is integer work: the output is
// fully deterministic.
#include <cstdlib>

#include "common.h"
#include "sort_small.h"

namespace rwr {

constexpr int SORT_BLOCK = 256;            // threads
constexpr int SORT_ITEMS = 16;             // elements per thread
constexpr int SORT_CHUNK = SORT_BLOCK * SORT_ITEMS;
constexpr int RADIX = 256;

size_t radix_sort_temp_bytes(size_t m, int nseg)
{
    size_t nblocks = (m + SORT_CHUNK - 1) / SORT_CHUNK;
    if (nblocks == 0) nblocks = 1;
    // hist[seg][RADIX][nblocks] + tot[seg][RADIX]
    return ((size_t)nseg * RADIX * nblocks + (size_t)nseg * RADIX) * sizeof(uint32_t);
}

template <typename KeyT>
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_hist(const KeyT *__restrict__ keys, size_t m,
                                                          int shift, uint32_t *__restrict__ hist,
                                                          unsigned nblocks)
{
    __shared__ uint32_t h[RADIX];
    const int seg = blockIdx.y;
    const unsigned blk = blockIdx.x;
    keys += (size_t)seg * m;
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blk * SORT_CHUNK;
#pragma unroll
    for (int t = 0; t < SORT_ITEMS; ++t) {
        size_t idx = base + (size_t)t * SORT_BLOCK + threadIdx.x;
        if (idx < m) atomicAdd(&h[(unsigned)(keys[idx] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * RADIX + threadIdx.x) * nblocks + blk] = h[threadIdx.x];
}

// exclusive scan along the block dimension of one (segment, digit) row, in place;
// the row total goes to tot[seg][digit]
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scan_rows(uint32_t *__restrict__ hist,
                                                               uint32_t *__restrict__ tot,
                                                               unsigned nblocks)
{
    __shared__ uint32_t wsum[SORT_BLOCK / WAVE];
    __shared__ uint32_t carry_s;
    const int seg = blockIdx.y, digit = blockIdx.x;
    uint32_t *row = hist + ((size_t)seg * RADIX + digit) * nblocks;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (unsigned base = 0; base < nblocks; base += SORT_BLOCK) {
        unsigned i = base + threadIdx.x;
        uint32_t v = (i < nblocks) ? row[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
        }
        if (lane == WAVE - 1) wsum[wv] = incl;
        __syncthreads();
        uint32_t pre = carry_s;
        for (int q = 0; q < wv; ++q) pre += wsum[q];
        if (i < nblocks) row[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == SORT_BLOCK - 1) carry_s = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[(size_t)seg * RADIX + digit] = carry_s;
}

template <typename KeyT>
__global__ __launch_bounds__(SORT_BLOCK) void k_sort_scatter(
    const KeyT *__restrict__ keys, const uint32_t *__restrict__ vals, KeyT *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, size_t m, int shift, const uint32_t *__restrict__ hist,
    const uint32_t *__restrict__ tot, unsigned nblocks)
{
    __shared__ uint32_t run[RADIX];                       // next free slot per digit
    __shared__ uint32_t wcnt[SORT_BLOCK / WAVE][RADIX];   // per-wave digit counts of a sub-tile
    __shared__ uint32_t dbase[RADIX];
    const int seg = blockIdx.y;
    const unsigned blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    keys += (size_t)seg * m;
    vals += (size_t)seg * m;
    keys_out += (size_t)seg * m;
    vals_out += (size_t)seg * m;

    // exclusive scan of the 256 digit totals of this segment (4 waves x 64)
    {
        uint32_t v = tot[(size_t)seg * RADIX + tid];
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
        }
        if (lane == WAVE - 1) wcnt[0][wv] = incl;
        __syncthreads();
        uint32_t pre = 0;
        for (int q = 0; q < wv; ++q) pre += wcnt[0][q];
        dbase[tid] = pre + incl - v;
        __syncthreads();
    }
    run[tid] = dbase[tid] + hist[((size_t)seg * RADIX + tid) * nblocks + blk];
#pragma unroll
    for (int q = 0; q < SORT_BLOCK / WAVE; ++q) wcnt[q][tid] = 0;
    __syncthreads();

    const size_t base = (size_t)blk * SORT_CHUNK;
    for (int t = 0; t < SORT_ITEMS; ++t) {
        size_t idx = base + (size_t)t * SORT_BLOCK + tid;
        if (base + (size_t)t * SORT_BLOCK >= m) break;    // block-uniform
        bool valid = idx < m;
        KeyT key = valid ? keys[idx] : (KeyT)0;
        uint32_t val = valid ? vals[idx] : 0u;
        unsigned digit = (unsigned)(key >> shift) & (RADIX - 1);
        // 64-lane match on the digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            bool bit = (digit >> b) & 1u;
            unsigned long long mb = __ballot(valid && bit);
            peers &= bit ? mb : ~mb;
        }
        unsigned rank_in_wave = __popcll(peers & ((1ull << lane) - 1ull));
        unsigned cnt = __popcll(peers);
        if (valid && rank_in_wave == 0) wcnt[wv][digit] = cnt;
        __syncthreads();
        uint32_t pos = 0;
        if (valid) {
            pos = run[digit] + rank_in_wave;
            for (int q = 0; q < wv; ++q) pos += wcnt[q][digit];
        }
        __syncthreads();
        {
            uint32_t s = 0;
#pragma unroll
            for (int q = 0; q < SORT_BLOCK / WAVE; ++q) {
                s += wcnt[q][tid];
                wcnt[q][tid] = 0;
            }
            run[tid] += s;
        }
        if (valid) {
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
    }
}

// Few elements (an ego network's node arrays, its link list): ALL passes of the sort in ONE launch of one 1024-thread
// workgroup (sort_small.h).  The graph build of such a graph is launch-bound: this takes ~30 launches out of it.
template <typename KeyT>
__global__ __launch_bounds__(SMALL_SORT_THREADS) void k_sort_small(KeyT *ka, KeyT *kb, uint32_t *va, uint32_t *vb, uint32_t m,
                                                                   int key_bits)
{
    __shared__ SmallSortLds L;
    sort_small_body<KeyT>(L, ka, kb, va, vb, m, key_bits);
}

template <typename KeyT>
int32_t radix_sort_pairs(KeyT *keys, KeyT *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t m,
                         int nseg, int key_bits, void *temp, hipStream_t stream, bool *in_alt)
{
    *in_alt = false;
    if (m == 0 || nseg == 0) return RWR_OK;
    if (key_bits < 1 || key_bits > (int)(8 * sizeof(KeyT))) {
        set_error("radix_sort_pairs: %d key bits do not fit a %d-bit key", key_bits, (int)(8 * sizeof(KeyT)));
        return RWR_E_INVALID;
    }
    static const int small_env = [] { const char *e = RWR_TUNE_ENV("RWR_SMALL_SORT"); return e ? atoi(e) : 1; }();
    static const size_t small_max = [] { const char *e = RWR_TUNE_ENV("RWR_SMALL_SORT_MAX"); return e ? (size_t)atol(e) : SMALL_SORT_MAX; }();
    if (small_env && nseg == 1 && m <= small_max) {
        hipLaunchKernelGGL(k_sort_small<KeyT>, dim3(1), dim3(SMALL_SORT_THREADS), 0, stream, keys, keys_alt, vals, vals_alt,
                           (uint32_t)m, key_bits);
        RWR_HIP(hipGetLastError());
        *in_alt = (((key_bits + 7) / 8) & 1) != 0;
        return RWR_OK;
    }
    if (m >= 0xFFFFFFFFull) {
        set_error("radix_sort_pairs: segment of %zu elements exceeds the 32-bit payload range", m);
        return RWR_E_UNSUPPORTED;
    }
    unsigned nblocks = cdiv(m, SORT_CHUNK);
    uint32_t *hist = (uint32_t *)temp;
    uint32_t *tot = hist + (size_t)nseg * RADIX * nblocks;
    KeyT *kin = keys, *kout = keys_alt;
    uint32_t *vin = vals, *vout = vals_alt;
    bool alt = false;
    for (int shift = 0; shift < key_bits; shift += 8) {
        dim3 grid(nblocks, nseg);
        hipLaunchKernelGGL(k_sort_hist<KeyT>, grid, dim3(SORT_BLOCK), 0, stream, kin, m, shift, hist, nblocks);
        hipLaunchKernelGGL(k_sort_scan_rows, dim3(RADIX, nseg), dim3(SORT_BLOCK), 0, stream, hist, tot, nblocks);
        hipLaunchKernelGGL(k_sort_scatter<KeyT>, grid, dim3(SORT_BLOCK), 0, stream, kin, vin, kout, vout, m,
                           shift, hist, tot, nblocks);
        KeyT *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
        alt = !alt;
    }
    RWR_HIP(hipGetLastError());
    *in_alt = alt;
    return RWR_OK;
}

template int32_t radix_sort_pairs<uint32_t>(uint32_t *, uint32_t *, uint32_t *, uint32_t *, size_t, int, int,
                                            void *, hipStream_t, bool *);
template int32_t radix_sort_pairs<uint64_t>(uint64_t *, uint64_t *, uint32_t *, uint32_t *, size_t, int, int,
                                            void *, hipStream_t, bool *);

}  // namespace rwr
