// Ranking stage of Recommender.Recommendation (Recommender.cs:27-38, 42-51):
// candidates = ITEM nodes not LIKEd by the seed, ordered by (score desc, id desc).
//
// The ITEM rows are kept pre-sorted by id descending (graph build), so a STABLE sort on
// the score alone yields the reference's total order (ids are unique among items:
// DataLoader.cs:51-58, so List.Sort's instability cannot show).  Scores are mapped to
// order-preserving 64-bit keys; non-candidates (excluded by k_exclude: score -1, or a
// padding lane) get the largest key and sort to the end.
#include "engine.h"

#include <cstring>

#include <cstdlib>

namespace rwr {

template <int G>
__global__ __launch_bounds__(256) void k_rank_keys(int32_t n_items, const int32_t *__restrict__ item_order,
                                                   const double *__restrict__ X,
                                                   const int32_t *__restrict__ seeds,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    // thread = (item slot q, seed k); consecutive threads walk k first: X reads coalesce
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t q = t / G;
    const int k = (int)(t % G);
    if (q >= n_items) return;
    const int32_t row = item_order[q];
    const double s = X[(size_t)row * G + k];
    const bool cand = (seeds[k] >= 0) && (s >= 0.0);
    keys[(size_t)k * n_items + q] = cand ? ~f64_orderable(s) : ~0ull;
    vals[(size_t)k * n_items + q] = (uint32_t)q;
}

// counts[k] = number of candidates = first position holding the sentinel key
__global__ void k_rank_count(int32_t n_items, int G, const uint64_t *__restrict__ keys, int32_t *__restrict__ counts)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G) return;
    const uint64_t *kk = keys + (size_t)k * n_items;
    int32_t lo = 0, hi = n_items;
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (kk[mid] == ~0ull) hi = mid;
        else lo = mid + 1;
    }
    counts[k] = lo;
}

__global__ __launch_bounds__(256) void k_rank_emit(int32_t n_items, int G, const int32_t *__restrict__ slot_k,
                                                   int32_t top_n, const uint32_t *__restrict__ vals,
                                                   const int32_t *__restrict__ cand_counts,
                                                   const int32_t *__restrict__ item_order,
                                                   const int64_t *__restrict__ node_id,
                                                   const double *__restrict__ X, int64_t *__restrict__ out_id,
                                                   double *__restrict__ out_score, int32_t *__restrict__ out_counts)
{
    const int k = blockIdx.y;
    const int32_t orow = slot_k[k];              // batch position of this slot's seed
    if (orow < 0) return;
    int32_t cnt = cand_counts[k];
    if (cnt > top_n) cnt = top_n;
    const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) out_counts[orow] = cnt;
    if (q >= cnt) return;
    const int32_t row = item_order[vals[(size_t)k * n_items + q]];
    out_id[(size_t)orow * top_n + q] = node_id[row];
    out_score[(size_t)orow * top_n + q] = X[(size_t)row * G + k];
}

// ---------------------------------------------------------------------------------------
// Batched top-k without sorting everything: MSD radix SELECT on the 128-bit key
// (score, id) -- 8-bit digits from the top, one histogram pass per level over the tile's
// candidates -- until the bin holding the k-th best entry has at most SEL_CAP members;
// then every entry at or above that bin is collected (at most k-1+SEL_CAP) and only those
// are sorted (bitonic, in LDS).  Integer counting only: the result is exactly the first
// top_n entries of the full (score desc, id desc) order of Recommender.cs:35-38.
// Typical depth is two levels (sign/exponent byte, then 4 exponent + 4 mantissa bits).
// ---------------------------------------------------------------------------------------
constexpr int SEL_MAX_K = 1024;
constexpr int SEL_CAP = 3072;
constexpr int SEL_SLOTS = 4096;      // >= SEL_MAX_K - 1 + SEL_CAP
constexpr int SEL_LEVELS = 16;
constexpr int SEL_ROWS_PER_BLOCK = 4096;

struct SelState {
    uint64_t ph, pl;     // prefix: nbits <= 64: ph holds the top nbits of the score key (right-aligned);
                         // nbits > 64: ph = full score key, pl = top (nbits-64) bits of the id key
    int32_t nbits;       // bits decided so far
    int32_t k_rem;       // entries still to take from inside the prefix bin
    int32_t total;       // candidates of the segment (set at level 0)
    int32_t done;
    int32_t cand_cnt;    // collect cursor
    int32_t pad;
};

__device__ static inline int sel_cmp(uint64_t hi, uint64_t lo, const SelState &st)
{
    if (st.nbits == 0) return 0;
    if (st.nbits <= 64) {
        const uint64_t a = hi >> (64 - st.nbits);
        return a > st.ph ? 1 : (a < st.ph ? -1 : 0);
    }
    if (hi != st.ph) return hi > st.ph ? 1 : -1;
    const uint64_t b = lo >> (128 - st.nbits);
    return b > st.pl ? 1 : (b < st.pl ? -1 : 0);
}

__global__ void k_sel_init(int nseg, int32_t top_n, const int32_t *__restrict__ seeds, SelState *__restrict__ st)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nseg) return;
    SelState s{};
    s.k_rem = top_n;
    s.done = seeds[q] < 0 ? 1 : 0;
    st[q] = s;
}

// one level's decision for one segment from its 256 digit counts h (cleared for the next level).  AGENT: the counts were
// added by other workgroups of the running kernel (read with agent-scope loads); otherwise h is this workgroup's own memory
template <bool AGENT>
__device__ __forceinline__ void sel_decide(SelState &s, uint32_t *h, int level)
{
    auto cnt = [&](int b) -> uint32_t { return AGENT ? __hip_atomic_load(h + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : h[b]; };
    if (s.done) return;
    if (level == 0) {
        int64_t tot = 0;
        for (int b = 0; b < 256; ++b) tot += cnt(b);
        s.total = (int32_t)tot;
        if (tot <= s.k_rem) {                  // fewer candidates than top_n: take them all
            s.k_rem = (int32_t)tot;
            s.done = 1;                        // nbits stays 0: every key matches the empty prefix
            for (int b = 0; b < 256; ++b) h[b] = 0;
            return;
        }
    }
    int64_t cum = 0;
    int sel = 0;
    uint32_t bin = 0;
    for (int b = 255; b >= 0; --b) {
        const uint32_t c = cnt(b);
        if (cum + c >= (int64_t)s.k_rem) { sel = b; bin = c; break; }
        cum += c;
    }
    for (int b = 0; b < 256; ++b) h[b] = 0;
    s.k_rem -= (int32_t)cum;                   // entries above the chosen bin are all taken
    if (level < 8) s.ph = (s.ph << 8) | (uint64_t)sel;
    else s.pl = (s.pl << 8) | (uint64_t)sel;
    s.nbits += 8;
    if (bin <= (uint32_t)SEL_CAP || level == SEL_LEVELS - 1) s.done = 1;
}
// the digit of level `level` of a candidate's key, or -1 when it is excluded or lies outside the segment's prefix
__device__ __forceinline__ int sel_digit(double sc, int32_t row, const int64_t *__restrict__ node_id, const SelState &my, int level)
{
    if (!(sc >= 0.0)) return -1;                                 // excluded (Recommender.cs:29)
    const uint64_t hi = f64_orderable(sc);
    uint64_t lo = 0;
    if (level >= 8) lo = i64_orderable(node_id[row]);
    if (sel_cmp(hi, lo, my) != 0) return -1;
    return (level < 8) ? (int)((hi >> (56 - 8 * level)) & 255u) : (int)((lo >> (56 - 8 * (level - 8))) & 255u);
}

template <int G>
__global__ __launch_bounds__(256) void k_sel_hist(int32_t n, int32_t n_items, const int32_t *__restrict__ item_rows,
                                                  const int64_t *__restrict__ node_id,
                                                  const double *__restrict__ X, SelState *__restrict__ st,
                                                  uint32_t *__restrict__ ghist, int level, unsigned int *__restrict__ ticket)
{
    // ticket != nullptr: the workgroup that finishes LAST takes the level's decision for every segment (no k_sel_decide launch)
    constexpr int RL = 256 / G;
    // rows padded to 257 words: at one digit per seed (the usual case at the top levels) the G seeds of a wave would
    // otherwise all hit LDS bank (digit % 32) -- a G-way conflict on every atomic
    __shared__ uint32_t h[G][257];
    __shared__ SelState sst[G];
    __shared__ int any_active;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, k = tid % G, rl = tid / G;
    if (tid == 0) any_active = 0;
    __syncthreads();
    if (tid < G) {
        sst[tid] = st[tile * G + tid];
        if (!sst[tid].done) any_active = 1;
    }
    __syncthreads();
    if (!any_active && !ticket) return;
    for (int b = rl; b < 256; b += RL) h[k][b] = 0;
    __syncthreads();
    const double *x = X + (size_t)tile * (size_t)n * G;
    const SelState my = sst[k];
    const int32_t q0 = blockIdx.x * SEL_ROWS_PER_BLOCK;
    const int32_t q1 = (q0 + SEL_ROWS_PER_BLOCK < n_items) ? q0 + SEL_ROWS_PER_BLOCK : n_items;
    if (any_active && !my.done) {
        // 4 rows per trip: four independent (row index -> score) load chains in flight per thread
        for (int32_t qb = q0 + rl; qb < q1; qb += 4 * RL) {
            int32_t row[4];
            double sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int32_t q = qb + u * RL;
                row[u] = item_rows[q < q1 ? q : q1 - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sv[u] = x[(size_t)row[u] * G + k];
            int dg[4];                                           // digit of each of the 4 elements, -1 = not counted
#pragma unroll
            for (int u = 0; u < 4; ++u) dg[u] = (qb + u * RL >= q1) ? -1 : sel_digit(sv[u], row[u], node_id, my, level);
            // equal digits among the 4 (the rule at the top levels, where one exponent byte covers everything) are
            // counted with ONE LDS atomic: LDS atomics, not the loads, bound this kernel
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dg[u] < 0) continue;
                unsigned cnt = 1;
#pragma unroll
                for (int v = u + 1; v < 4; ++v)
                    if (dg[v] == dg[u]) { ++cnt; dg[v] = -1; }
                atomicAdd(&h[k][dg[u]], cnt);
            }
        }
    }
    __syncthreads();
    for (int b = rl; b < 256; b += RL) {
        const uint32_t c = h[k][b];
        if (c) atomicAdd(&ghist[((size_t)tile * G + k) * 256 + b], c);
    }
    if (!ticket) return;
    // the last workgroup to arrive decides (its own adds and everyone else's are complete and visible: fence + ticket)
    __shared__ int last_s;
    __threadfence();
    __syncthreads();
    if (tid == 0) last_s = atomicAdd(&ticket[level], 1u) == gridDim.x * gridDim.y - 1u;
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    // (the 256 counts of a segment are fetched by the 256 threads at once -- one thread reading them one by one through
    //  agent-scope loads took 10-50 us -- and cleared for the next level; the decision itself runs on the LDS copy)
    const int nseg = (int)gridDim.y * G;
    uint32_t *hc = &h[0][0];
    for (int q = 0; q < nseg; ++q) {
        uint32_t *gh = ghist + (size_t)q * 256;
        hc[tid] = __hip_atomic_load(gh + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gh[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            SelState sq = st[q];
            if (!sq.done) {
                sel_decide<false>(sq, hc, level);
                st[q] = sq;
            }
        }
        __syncthreads();
    }
}

// Levels L0 .. 15 for ONE tile by one workgroup: by then the prefix holds L0 bytes of the key and the segment is almost
// always decided (the bin of the k-th entry has at most SEL_CAP members) -- the kernel then ends at once, in place of
// 2 * (16 - L0) launches that would find nothing to do; when a segment is NOT decided (a tie across thousands of scores)
// the workgroup walks the tile's candidates itself, level by level.
template <int G>
__global__ __launch_bounds__(256) void k_sel_tail(int32_t n, int32_t n_items, const int32_t *__restrict__ item_rows,
                                                  const int64_t *__restrict__ node_id, const double *__restrict__ X,
                                                  SelState *__restrict__ st, int level0)
{
    constexpr int RL = 256 / G;
    __shared__ uint32_t h[G][257];
    __shared__ SelState sst[G];
    __shared__ int any_active;
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, k = tid % G, rl = tid / G;
    const double *x = X + (size_t)tile * (size_t)n * G;
    if (tid < G) sst[tid] = st[tile * G + tid];
    for (int level = level0; level < SEL_LEVELS; ++level) {
        if (tid == 0) any_active = 0;
        __syncthreads();
        if (tid < G && !sst[tid].done) any_active = 1;
        for (int b = rl; b < 256; b += RL) h[k][b] = 0;
        __syncthreads();
        if (!any_active) break;                                  // (uniform)
        const SelState my = sst[k];
        if (!my.done)
            for (int32_t q = rl; q < n_items; q += RL) {
                const int32_t row = item_rows[q];
                const int dg = sel_digit(x[(size_t)row * G + k], row, node_id, my, level);
                if (dg >= 0) atomicAdd(&h[k][dg], 1u);
            }
        __syncthreads();
        if (tid < G) sel_decide<false>(sst[tid], &h[tid][0], level);
        __syncthreads();
    }
    if (tid < G) st[tile * G + tid] = sst[tid];
}

__global__ void k_sel_decide(int nseg, SelState *__restrict__ st, uint32_t *__restrict__ ghist, int level)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nseg) return;
    SelState s = st[q];
    if (s.done) return;
    sel_decide<false>(s, ghist + (size_t)q * 256, level);
    st[q] = s;
}

struct SelCand {
    uint64_t hi, lo;
};

template <int G>
__global__ __launch_bounds__(256) void k_sel_collect(int32_t n, int32_t n_items, const int32_t *__restrict__ item_rows,
                                                     const int64_t *__restrict__ node_id,
                                                     const double *__restrict__ X, SelState *__restrict__ st,
                                                     const int32_t *__restrict__ seeds, SelCand *__restrict__ cand)
{
    constexpr int RL = 256 / G;
    __shared__ SelState sst[G];
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, k = tid % G, rl = tid / G;
    if (tid < G) sst[tid] = st[tile * G + tid];
    __syncthreads();
    if (seeds[tile * G + k] < 0) return;
    const SelState my = sst[k];
    const double *x = X + (size_t)tile * (size_t)n * G;
    const int32_t q0 = blockIdx.x * SEL_ROWS_PER_BLOCK;
    const int32_t q1 = (q0 + SEL_ROWS_PER_BLOCK < n_items) ? q0 + SEL_ROWS_PER_BLOCK : n_items;
    for (int32_t qb = q0 + rl; qb < q1; qb += 4 * RL) {
        int32_t row[4];
        double sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t q = qb + u * RL;
            row[u] = item_rows[q < q1 ? q : q1 - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = x[(size_t)row[u] * G + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (qb + u * RL >= q1) continue;
            const double s = sv[u];
            if (!(s >= 0.0)) continue;
            const uint64_t hi = f64_orderable(s);
            if (my.nbits > 0 && my.nbits <= 64 && (hi >> (64 - my.nbits)) < my.ph) continue;   // cheap reject
            const uint64_t lo = i64_orderable(node_id[row[u]]);
            if (sel_cmp(hi, lo, my) < 0) continue;
            const int slot = atomicAdd(&st[tile * G + k].cand_cnt, 1);
            if (slot < SEL_SLOTS) cand[((size_t)tile * G + k) * SEL_SLOTS + slot] = SelCand{hi, lo};
        }
    }
}

// one block per segment: bitonic sort (descending) of the collected candidates, emit the first top_n
__global__ __launch_bounds__(256) void k_sel_sort_emit(const int32_t *__restrict__ slot_k, int32_t top_n,
                                                       const SelState *__restrict__ st,
                                                       const SelCand *__restrict__ cand, int64_t *__restrict__ out_id,
                                                       double *__restrict__ out_score, int32_t *__restrict__ out_counts)
{
    extern __shared__ SelCand sc[];
    const int seg = blockIdx.x;
    const int32_t orow = slot_k[seg];            // batch position of this slot's seed
    if (orow < 0) return;
    int cnt = st[seg].cand_cnt;
    if (cnt > SEL_SLOTS) cnt = SEL_SLOTS;
    int N2 = 1;
    while (N2 < cnt) N2 <<= 1;
    const SelCand *c = cand + (size_t)seg * SEL_SLOTS;
    for (int i = threadIdx.x; i < N2; i += blockDim.x) sc[i] = (i < cnt) ? c[i] : SelCand{0ull, 0ull};
    __syncthreads();
    for (int size = 2; size <= N2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < N2 / 2; t += blockDim.x) {
                const int lo_i = 2 * t - (t & (stride - 1));
                const int hi_i = lo_i + stride;
                const bool desc = (lo_i & size) == 0;        // first half of each bitonic pair: descending
                const SelCand a = sc[lo_i], b = sc[hi_i];
                const bool a_lt_b = (a.hi < b.hi) || (a.hi == b.hi && a.lo < b.lo);
                if (a_lt_b == desc) {
                    sc[lo_i] = b;
                    sc[hi_i] = a;
                }
            }
            __syncthreads();
        }
    }
    int take = cnt < top_n ? cnt : top_n;
    if (threadIdx.x == 0) out_counts[orow] = take;
    for (int i = threadIdx.x; i < take; i += blockDim.x) {
        const SelCand v = sc[i];
        out_id[(size_t)orow * top_n + i] = (int64_t)(v.lo ^ 0x8000000000000000ull);
        const uint64_t u = (v.hi & 0x8000000000000000ull) ? (v.hi ^ 0x8000000000000000ull) : ~v.hi;
        double s;
        __builtin_memcpy(&s, &u, 8);
        out_score[(size_t)orow * top_n + i] = s;
    }
}

// Few items (an ego network: n_items <= SEL_SLOTS): the whole ranked list of a seed is one workgroup's bitonic sort in
// LDS on the 128-bit key (score, id) -- one launch instead of the ~27 of the segmented radix sort.  One block per seed
// slot of the tile; candidates = ITEM rows whose score is not the exclusion marker (Recommender.cs:27-31).
template <int G>
__global__ __launch_bounds__(256) void k_rank_small(int32_t n_items, const int32_t *__restrict__ item_rows,
                                                    const int64_t *__restrict__ node_id, const double *__restrict__ X,
                                                    const int32_t *__restrict__ seeds, const int32_t *__restrict__ slot_k,
                                                    int32_t top_n, int64_t *__restrict__ out_id,
                                                    double *__restrict__ out_score, int32_t *__restrict__ out_counts)
{
    extern __shared__ SelCand sc[];
    __shared__ int cnt_s;
    const int k = blockIdx.x;
    const int32_t orow = slot_k[k];
    if (orow < 0 || seeds[k] < 0) return;
    int N2 = 1;
    while (N2 < n_items) N2 <<= 1;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < N2; i += blockDim.x) {
        SelCand c{0ull, 0ull};                                   // below every real key: non-candidates sink to the end
        if (i < n_items) {
            const int32_t row = item_rows[i];
            const double sv = X[(size_t)row * G + k];
            if (sv >= 0.0) { c.hi = f64_orderable(sv); c.lo = i64_orderable(node_id[row]); ++mine; }
        }
        sc[i] = c;
    }
    if (mine) atomicAdd(&cnt_s, mine);
    __syncthreads();
    for (int size = 2; size <= N2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < N2 / 2; t += blockDim.x) {
                const int lo_i = 2 * t - (t & (stride - 1));
                const int hi_i = lo_i + stride;
                const bool desc = (lo_i & size) == 0;
                const SelCand a = sc[lo_i], b = sc[hi_i];
                const bool a_lt_b = (a.hi < b.hi) || (a.hi == b.hi && a.lo < b.lo);
                if (a_lt_b == desc) {
                    sc[lo_i] = b;
                    sc[hi_i] = a;
                }
            }
            __syncthreads();
        }
    }
    const int cnt = cnt_s;
    const int take = cnt < top_n ? cnt : top_n;
    if (threadIdx.x == 0) out_counts[orow] = take;
    for (int i = threadIdx.x; i < take; i += blockDim.x) {
        const SelCand v = sc[i];
        out_id[(size_t)orow * top_n + i] = (int64_t)(v.lo ^ 0x8000000000000000ull);
        const uint64_t u = (v.hi & 0x8000000000000000ull) ? (v.hi ^ 0x8000000000000000ull) : ~v.hi;
        double sv;
        __builtin_memcpy(&sv, &u, 8);
        out_score[(size_t)orow * top_n + i] = sv;
    }
}

#define RWR_DISPATCH_G(G, CALL)                          \
    switch (G) {                                         \
        case 1: { constexpr int GG = 1; CALL; } break;   \
        case 2: { constexpr int GG = 2; CALL; } break;   \
        case 4: { constexpr int GG = 4; CALL; } break;   \
        case 8: { constexpr int GG = 8; CALL; } break;   \
        case 16: { constexpr int GG = 16; CALL; } break; \
        case 32: { constexpr int GG = 32; CALL; } break; \
        default: { constexpr int GG = 64; CALL; } break; \
    }

int32_t rank_tile(rwr_graph *g, int G, const int32_t *d_slot_k_tile, int32_t top_n, const double *X,
                  const int32_t *d_seeds_tile, hipStream_t s)
{
    const int32_t m = g->n_items;
    if (m == 0) return RWR_OK;
    if (m <= SEL_SLOTS) {
        int N2 = 1;
        while (N2 < m) N2 <<= 1;
        RWR_DISPATCH_G(G, {
            (void)hipFuncSetAttribute((const void *)k_rank_small<GG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      SEL_SLOTS * (int)sizeof(SelCand));
            hipLaunchKernelGGL(k_rank_small<GG>, dim3((unsigned)G), dim3(256), (size_t)N2 * sizeof(SelCand), s, m, g->item_rows.p,
                               g->node_id.p, X, d_seeds_tile, d_slot_k_tile, top_n, g->d_out_id.p, g->d_out_score.p, g->d_counts.p);
        });
        RWR_HIP(hipGetLastError());
        return RWR_OK;
    }
    const size_t tot = (size_t)G * m;
    RWR_TRY(g->keys.ensure(tot));
    RWR_TRY(g->keys_alt.ensure(tot));
    RWR_TRY(g->vals.ensure(tot));
    RWR_TRY(g->vals_alt.ensure(tot));
    RWR_TRY(g->sort_temp.ensure(radix_sort_temp_bytes((size_t)m, G) + (size_t)G * sizeof(int32_t)));
    RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_rank_keys<GG>, dim3(cdiv(tot, 256)), dim3(256), 0, s, m,
                                         g->item_order.p, X, d_seeds_tile, g->keys.p, g->vals.p));
    RWR_HIP(hipGetLastError());
    bool alt = false;
    RWR_TRY(radix_sort_pairs<uint64_t>(g->keys.p, g->keys_alt.p, g->vals.p, g->vals_alt.p, (size_t)m, G, 64,
                                       g->sort_temp.p, s, &alt));
    const uint64_t *ks = alt ? g->keys_alt.p : g->keys.p;
    const uint32_t *vs = alt ? g->vals_alt.p : g->vals.p;
    int32_t *cand = (int32_t *)(g->sort_temp.p + radix_sort_temp_bytes((size_t)m, G));
    hipLaunchKernelGGL(k_rank_count, dim3(1), dim3(64), 0, s, m, G, ks, cand);
    int32_t width = top_n < m ? top_n : m;
    hipLaunchKernelGGL(k_rank_emit, dim3(cdiv((size_t)width, 256), G), dim3(256), 0, s, m, G, d_slot_k_tile, top_n, vs,
                       cand, g->item_order.p, g->node_id.p, X, g->d_out_id.p, g->d_out_score.p, g->d_counts.p);
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

// top-k for a whole tile group in one go (select path; top_n <= SEL_MAX_K)
int32_t rank_group_select(rwr_graph *g, int G, int tg, const int32_t *d_slot_k, int32_t top_n, const double *X,
                          const int32_t *d_seeds, hipStream_t s)
{
    const int32_t m = g->n_items;
    if (m == 0) return RWR_OK;
    const int nseg = tg * G;
    const size_t st_bytes = (size_t)nseg * sizeof(SelState);
    const size_t hist_bytes = (size_t)nseg * 256 * sizeof(uint32_t);
    const size_t cand_bytes = (size_t)nseg * SEL_SLOTS * sizeof(SelCand);
    const size_t ticket_bytes = SEL_LEVELS * sizeof(unsigned int);
    RWR_TRY(g->sort_temp.ensure(st_bytes + hist_bytes + ticket_bytes + cand_bytes + 64));
    SelState *st = (SelState *)g->sort_temp.p;
    uint32_t *ghist = (uint32_t *)(g->sort_temp.p + st_bytes);
    unsigned int *ticket = (unsigned int *)(g->sort_temp.p + st_bytes + hist_bytes);
    SelCand *cand = (SelCand *)(g->sort_temp.p + st_bytes + hist_bytes + ticket_bytes);
    RWR_HIP(hipMemsetAsync(ghist, 0, hist_bytes + ticket_bytes, s));
    hipLaunchKernelGGL(k_sel_init, dim3(cdiv((size_t)nseg, 64)), dim3(64), 0, s, nseg, top_n, d_seeds, st);
    const unsigned nblk = cdiv((size_t)m, SEL_ROWS_PER_BLOCK);
    // A call of a few seeds is launch-bound: 16 levels x (histogram + decision) are 32 launches of which, typically, the
    // first two or three find anything to do -- 0.35 of the 1.6 ms of a single-seed call on the 0.6 M-node graph.  There the
    // decision is taken by the histogram kernel's last workgroup, and after SEL_FUSED_LEVELS levels ONE workgroup per tile
    // finishes whatever is left (k_sel_tail).  Large batches keep the plain form (their launches are noise, and an undecided
    // segment should not be walked by a single workgroup).
    static const int fused_env = [] { const char *e = RWR_TUNE_ENV("RWR_SEL_FUSED"); return e ? atoi(e) : 1; }();
    constexpr int SEL_FUSED_LEVELS = 3;
    if (fused_env && nseg <= 64) {
        for (int level = 0; level < SEL_FUSED_LEVELS; ++level)
            RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_sel_hist<GG>, dim3(nblk, tg), dim3(256), 0, s, g->n, m, g->item_rows.p,
                                                 g->node_id.p, X, st, ghist, level, ticket));
        RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_sel_tail<GG>, dim3(tg), dim3(256), 0, s, g->n, m, g->item_rows.p, g->node_id.p, X, st,
                                             SEL_FUSED_LEVELS));
    } else {
        for (int level = 0; level < SEL_LEVELS; ++level) {
            RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_sel_hist<GG>, dim3(nblk, tg), dim3(256), 0, s, g->n, m, g->item_rows.p,
                                                 g->node_id.p, X, st, ghist, level, (unsigned int *)nullptr));
            hipLaunchKernelGGL(k_sel_decide, dim3(cdiv((size_t)nseg, 64)), dim3(64), 0, s, nseg, st, ghist, level);
        }
    }
    RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_sel_collect<GG>, dim3(nblk, tg), dim3(256), 0, s, g->n, m, g->item_rows.p,
                                         g->node_id.p, X, st, d_seeds, cand));
    (void)hipFuncSetAttribute((const void *)k_sel_sort_emit, hipFuncAttributeMaxDynamicSharedMemorySize,
                              SEL_SLOTS * (int)sizeof(SelCand));   // per launch: the attribute is per device
    hipLaunchKernelGGL(k_sel_sort_emit, dim3(nseg), dim3(256), SEL_SLOTS * sizeof(SelCand), s, d_slot_k, top_n, st, cand,
                       g->d_out_id.p, g->d_out_score.p, g->d_counts.p);
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

// ---------------------------------------------------------------------------------------
// Evaluation of a ranked list against a test set -- TweetRecommender/Experiment.cs:121-128.
// One block walks the list in rank order; hits are numbered with an ordered block-wide prefix
// (wave ballots), each hit's addend (double)nHits / (i + 1) is stored at its hit number, and a
// single thread finally adds the addends in rank order (the reference's summation order).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void eval_ranked_body(const int64_t *__restrict__ ranked, int32_t cnt,
                                                 const int64_t *__restrict__ test_sorted, int32_t n_test,
                                                 double *__restrict__ terms, int64_t *__restrict__ out_hits,
                                                 double *__restrict__ out_sum)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int32_t c0 = 0; c0 < cnt; c0 += 1024) {
        const int32_t pos = c0 + tid;
        bool hit = false;
        if (pos < cnt) {
            const int64_t id = ranked[pos];
            int32_t lo = 0, hi = n_test;                    // testSet.Contains (Experiment.cs:124)
            while (lo < hi) {
                const int32_t mid = lo + (hi - lo) / 2;
                const int64_t v = test_sorted[mid];
                if (v < id) lo = mid + 1;
                else hi = mid;
            }
            hit = lo < n_test && test_sorted[lo] == id;
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < wv; ++q) off += wave_cnt[q];
        if (hit) {
            const int h = off + __popcll(m & ((1ull << lane) - 1ull)) + 1;    // nHits after this hit (:125)
            if (h <= n_test) terms[h - 1] = (double)h / (double)(pos + 1);    // (double)nHits / (i + 1)  (:126)
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < 16; ++q) tot += wave_cnt[q];
            base_s += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        int nh = base_s < n_test ? base_s : n_test;
        double sum = 0.0;
        for (int h = 0; h < nh; ++h) sum += terms[h];                          // sumPrecision += ... in rank order
        *out_hits = base_s;
        *out_sum = sum;
    }
}

__global__ __launch_bounds__(1024) void k_eval_ranked(const int64_t *__restrict__ ranked, int32_t cnt,
                                                      const int64_t *__restrict__ test_sorted, int32_t n_test,
                                                      double *__restrict__ terms, int64_t *__restrict__ out_hits,
                                                      double *__restrict__ out_sum)
{
    eval_ranked_body(ranked, cnt, test_sorted, n_test, terms, out_hits, out_sum);
}
// one block per seed of a batch: row k of the ranked-list table against test set k (CSR)
__global__ __launch_bounds__(1024) void k_eval_ranked_batch(const int64_t *__restrict__ ranked, int64_t row_stride,
                                                            const int32_t *__restrict__ counts,
                                                            const int64_t *__restrict__ test_sorted,
                                                            const int64_t *__restrict__ test_ptr, double *__restrict__ terms,
                                                            int64_t *__restrict__ out_hits, double *__restrict__ out_sum)
{
    const int k = blockIdx.x;
    const int64_t t0 = test_ptr[k];
    eval_ranked_body(ranked + (size_t)k * row_stride, counts[k], test_sorted + t0, (int32_t)(test_ptr[k + 1] - t0), terms + t0,
                     out_hits + k, out_sum + k);
}

int32_t eval_ranked(rwr_graph *g, int32_t cnt, const int64_t *test_sorted_host, int64_t n_test, int64_t *n_hits,
                    double *sum_precision)
{
    hipStream_t s = g->stream;
    DevBuf<int64_t> d_test, d_hits;
    DevBuf<double> d_terms, d_sum;
    RWR_TRY(d_test.alloc((size_t)n_test));
    RWR_TRY(d_terms.alloc((size_t)n_test));
    RWR_TRY(d_hits.alloc(1));
    RWR_TRY(d_sum.alloc(1));
    // The test set and the two results cross through the handle's pinned buffer when it exists and they fit (an ego network's
    // fold: a few dozen ids): a copy from / to pageable memory is staged by the runtime under a process-wide lock and waits for
    // the device -- with the harness's ten threads (Program.cs:11) those copies, not the kernels, set the rate (8 K graphs/s
    // whatever the thread count).  The buffer holds the ranked list of the call that has just ended (small.hip), which this
    // entry point does not return: its score half is free.
    int64_t *pin = (g->sm_pin && n_test + 4 <= small_pin_words()) ? reinterpret_cast<int64_t *>(small_pin_scratch(g)) : nullptr;
    if (pin) {
        if (n_test > 0) memcpy(pin, test_sorted_host, sizeof(int64_t) * (size_t)n_test);
        if (n_test > 0) RWR_HIP(hipMemcpyAsync(d_test.p, pin, sizeof(int64_t) * (size_t)n_test, hipMemcpyHostToDevice, s));
    } else if (n_test > 0) {
        RWR_HIP(hipMemcpyAsync(d_test.p, test_sorted_host, sizeof(int64_t) * (size_t)n_test, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_eval_ranked, dim3(1), dim3(1024), 0, s, g->d_out_id.p, cnt, d_test.p, (int32_t)n_test, d_terms.p,
                       d_hits.p, d_sum.p);
    RWR_HIP(hipGetLastError());
    if (pin) {
        int64_t *o_hits = pin + n_test;
        double *o_sum = reinterpret_cast<double *>(pin + n_test + 1);
        RWR_HIP(hipMemcpyAsync(o_hits, d_hits.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        RWR_HIP(hipMemcpyAsync(o_sum, d_sum.p, sizeof(double), hipMemcpyDeviceToHost, s));
        RWR_HIP(hipStreamSynchronize(s));
        *n_hits = *o_hits;
        *sum_precision = *o_sum;
        g->sm_pin_count = -1;                                  // (the list's score half is gone)
        return RWR_OK;
    }
    RWR_HIP(hipMemcpyAsync(n_hits, d_hits.p, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(sum_precision, d_sum.p, sizeof(double), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    return RWR_OK;
}

// K ranked lists (rows of d_out_id, `row_stride` apart, lengths in d_counts) against K test sets given in CSR form, each
// already sorted and de-duplicated by the caller
int32_t eval_ranked_batch(rwr_graph *g, int32_t K, int64_t row_stride, const int64_t *test_ptr_host,
                          const int64_t *test_sorted_host, int64_t *n_hits, double *sum_precision)
{
    hipStream_t s = g->stream;
    const int64_t total = test_ptr_host[K];
    DevBuf<int64_t> d_test, d_ptr, d_hits;
    DevBuf<double> d_terms, d_sum;
    RWR_TRY(d_test.alloc((size_t)total));
    RWR_TRY(d_terms.alloc((size_t)total));
    RWR_TRY(d_ptr.alloc((size_t)K + 1));
    RWR_TRY(d_hits.alloc((size_t)K));
    RWR_TRY(d_sum.alloc((size_t)K));
    if (total > 0) RWR_HIP(hipMemcpyAsync(d_test.p, test_sorted_host, sizeof(int64_t) * (size_t)total, hipMemcpyHostToDevice, s));
    RWR_HIP(hipMemcpyAsync(d_ptr.p, test_ptr_host, sizeof(int64_t) * ((size_t)K + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_eval_ranked_batch, dim3((unsigned)K), dim3(1024), 0, s, g->d_out_id.p, row_stride, g->d_counts.p, d_test.p,
                       d_ptr.p, d_terms.p, d_hits.p, d_sum.p);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipMemcpyAsync(n_hits, d_hits.p, sizeof(int64_t) * (size_t)K, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(sum_precision, d_sum.p, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    return RWR_OK;
}

// one block per GRAPH of a batch (rwr_eval_graphs): each graph's own ranked list against its test set
struct EvalMultiArgs {
    const int64_t *ranked;
    const int32_t *count;
};
__global__ __launch_bounds__(1024) void k_eval_ranked_multi(const EvalMultiArgs *__restrict__ args,
                                                            const int64_t *__restrict__ test_sorted,
                                                            const int64_t *__restrict__ test_ptr, double *__restrict__ terms,
                                                            int64_t *__restrict__ out_hits, double *__restrict__ out_sum,
                                                            int32_t *__restrict__ out_len)
{
    const int k = blockIdx.x;
    const EvalMultiArgs a = args[k];
    const int64_t t0 = test_ptr[k];
    const int32_t cnt = a.count[0];
    if (threadIdx.x == 0) out_len[k] = cnt;
    eval_ranked_body(a.ranked, cnt, test_sorted + t0, (int32_t)(test_ptr[k + 1] - t0), terms + t0, out_hits + k, out_sum + k);
}

// the lists recommend_small_multi left in the graphs' tables against `count` test sets in CSR form (sorted, de-duplicated)
int32_t eval_ranked_multi(rwr_graph **gs, int32_t count, const int64_t *test_ptr_host, const int64_t *test_sorted_host,
                          int64_t *n_hits, double *sum_precision, int64_t *list_len, hipStream_t s)
{
    if (count <= 0) return RWR_OK;
    const int64_t total = test_ptr_host[count];
    DevBuf<int64_t> d_test, d_ptr, d_hits;
    DevBuf<double> d_terms, d_sum;
    DevBuf<int32_t> d_len;
    DevBuf<EvalMultiArgs> d_args;
    RWR_TRY(d_test.alloc((size_t)total));
    RWR_TRY(d_terms.alloc((size_t)total));
    RWR_TRY(d_ptr.alloc((size_t)count + 1));
    RWR_TRY(d_hits.alloc((size_t)count));
    RWR_TRY(d_sum.alloc((size_t)count));
    RWR_TRY(d_len.alloc((size_t)count));
    RWR_TRY(d_args.alloc((size_t)count));
    // every table goes through the thread's pinned buffers (build.hip: multi_pinned): [args][ptr][test ids] in, [hits][sums][len] out
    const size_t b_args = sizeof(EvalMultiArgs) * (size_t)count, b_ptr = 8 * ((size_t)count + 1), b_test = 8 * (size_t)total;
    void *in_v = nullptr, *out_v = nullptr;
    RWR_TRY(multi_pinned(3, b_args + b_ptr + b_test + 64, &in_v));
    RWR_TRY(multi_pinned(4, (size_t)count * 24 + 64, &out_v));
    EvalMultiArgs *h = static_cast<EvalMultiArgs *>(in_v);
    int64_t *h_ptr = reinterpret_cast<int64_t *>(static_cast<uint8_t *>(in_v) + b_args);
    int64_t *h_test = h_ptr + count + 1;
    int64_t *o_hits = static_cast<int64_t *>(out_v);
    double *o_sum = reinterpret_cast<double *>(o_hits + count);
    int32_t *h_len = reinterpret_cast<int32_t *>(o_sum + count);
    for (int32_t i = 0; i < count; ++i) { h[i].ranked = gs[i]->d_out_id.p; h[i].count = gs[i]->d_counts.p; }
    memcpy(h_ptr, test_ptr_host, b_ptr);
    if (total > 0) memcpy(h_test, test_sorted_host, b_test);
    if (total > 0) RWR_HIP(hipMemcpyAsync(d_test.p, h_test, b_test, hipMemcpyHostToDevice, s));
    RWR_HIP(hipMemcpyAsync(d_ptr.p, h_ptr, b_ptr, hipMemcpyHostToDevice, s));
    RWR_HIP(hipMemcpyAsync(d_args.p, h, b_args, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_eval_ranked_multi, dim3((unsigned)count), dim3(1024), 0, s, d_args.p, d_test.p, d_ptr.p, d_terms.p, d_hits.p,
                       d_sum.p, d_len.p);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipMemcpyAsync(o_hits, d_hits.p, sizeof(int64_t) * (size_t)count, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(o_sum, d_sum.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(h_len, d_len.p, sizeof(int32_t) * (size_t)count, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    memcpy(n_hits, o_hits, sizeof(int64_t) * (size_t)count);
    memcpy(sum_precision, o_sum, sizeof(double) * (size_t)count);
    if (list_len) for (int32_t i = 0; i < count; ++i) list_len[i] = h_len[i];
    return RWR_OK;
}

// A seed without explicit out-links is dangling (Graph.cs:64,86): all of its rank mass returns to it every iteration
// (Model.cs:94-98), so after ANY number of iterations rank = n at the seed and exactly 0 elsewhere.  Its ranked list is
// therefore known without iterating: the seed itself first if it is an ITEM (score n; it has no LIKE link, so it is
// not excluded), then every other item with score +0.0 in id-descending order (Recommender.cs:35-38) = item_order.
// (Without this shortcut such seeds force the radix select through all 16 digit levels: a 500K-way tie at score 0.)
__global__ __launch_bounds__(256) void k_emit_dangling(int n_d, const int32_t *__restrict__ d_rows /* batch positions */,
                                                       const int32_t *__restrict__ d_seed, int32_t n, int32_t n_items,
                                                       int32_t top_n, const int32_t *__restrict__ item_order,
                                                       const int64_t *__restrict__ node_id,
                                                       const uint8_t *__restrict__ node_type, int64_t *__restrict__ out_id,
                                                       double *__restrict__ out_score, int32_t *__restrict__ out_counts)
{
    const int d = blockIdx.y;
    if (d >= n_d) return;
    const int32_t orow = d_rows[d], seed = d_seed[d];
    const bool seed_is_item = node_type[seed] == RWR_NODE_ITEM;
    const int32_t cnt = top_n < n_items ? top_n : n_items;
    const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) out_counts[orow] = cnt;
    if (q >= cnt) return;
    if (!seed_is_item) {
        out_id[(size_t)orow * top_n + q] = node_id[item_order[q]];
        out_score[(size_t)orow * top_n + q] = 0.0;
        return;
    }
    if (q == 0) {
        out_id[(size_t)orow * top_n] = node_id[seed];
        out_score[(size_t)orow * top_n] = (double)n;
        return;
    }
    // position of the seed inside item_order (ids are unique): entries before it keep their place, later ones shift by one
    // -- found by a short scan of the first q entries only
    int32_t src = q - 1;
    for (int32_t t = 0; t <= src; ++t)
        if (item_order[t] == seed) { src = q; break; }
    out_id[(size_t)orow * top_n + q] = node_id[item_order[src]];
    out_score[(size_t)orow * top_n + q] = 0.0;
}

int32_t emit_dangling(rwr_graph *g, const std::vector<int32_t> &rows, const std::vector<int32_t> &seeds, int32_t top_n,
                      hipStream_t s)
{
    const int n_d = (int)rows.size();
    if (n_d == 0 || g->n_items == 0) return RWR_OK;
    DevBuf<int32_t> d_rows, d_seed;
    RWR_TRY(d_rows.alloc(n_d));
    RWR_TRY(d_seed.alloc(n_d));
    RWR_HIP(hipMemcpyAsync(d_rows.p, rows.data(), n_d * sizeof(int32_t), hipMemcpyHostToDevice, s));
    RWR_HIP(hipMemcpyAsync(d_seed.p, seeds.data(), n_d * sizeof(int32_t), hipMemcpyHostToDevice, s));
    const int32_t cnt = top_n < g->n_items ? top_n : g->n_items;
    for (int d0 = 0; d0 < n_d; d0 += 65535) {
        const int nd = n_d - d0 < 65535 ? n_d - d0 : 65535;
        hipLaunchKernelGGL(k_emit_dangling, dim3(cdiv((size_t)cnt, 256), nd), dim3(256), 0, s, nd, d_rows.p + d0,
                           d_seed.p + d0, g->n, g->n_items, top_n, g->item_order.p, g->node_id.p, g->node_type.p,
                           g->d_out_id.p, g->d_out_score.p, g->d_counts.p);
    }
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(s));   // d_rows / d_seed are released on return
    return RWR_OK;
}

int rank_select_max_k() { return SEL_MAX_K; }

}  // namespace rwr
