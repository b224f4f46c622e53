// Ego-network-sized graphs: the WHOLE Recommendation call as ONE kernel launch.
//
// The reference's real workload is thousands of small graphs (10^3 - 10^5 nodes), one Recommendation(0, 0.15f, nIter) each
// (TweetRecommender/Experiment.cs:69-109, Program.cs:61-66).  Through the general path such a call is ~100 launches of
// ~5 us kernels plus a few blocking copies: launch-bound, ~1.1 ms whatever the graph holds.  For graphs of up to
// SM_MAX_N nodes and SEL_SLOTS items one 512-thread workgroup does it all -- Model ctor (Model.cs:33-50), nIter x
// (deliverRanks + updateRanks) (Model.cs:68-108), the LIKE exclusion and the (score desc, id desc) sort
// (Recommender.cs:20-38) -- with workgroup barriers where the general path has kernel boundaries:
//   * rank vectors live in global memory (a few KB: L1 / L2 resident), written and read by the one workgroup only;
//   * per step, every thread forms the restart addends of its nodes (Model.cs:91,97) and the addends of the links INTO
//     the seed (Model.cs:84,87) and drops them at their place of the seed row's addend sequence M (LDS) -- for node i
//     ascending: first i's links into the seed in list order, then i's restart addend (Model.cs:85-93,96-97); the places
//     depend on the graph and the seed only and are computed once;
//   * wave 0 then folds M strictly in order: the first 1024 addends with real fp64 adds, the rest as an exact parallel
//     reduction inside the current binade (pf.h; chain_scan.hip explains the arithmetic) with real adds wherever the
//     running sum leaves a binade -- bit for bit the sequential chain;
//   * waves 1-7 meanwhile sum every other row in list order (weighted form, Model.cs:84,87): rows of >= 128 in-links
//     (the ego itself, hub items) by a whole wave -- parallel loads and products, sequential adds --, the rest a lane each;
//   * ranking: bitonic sort of the <= 4096 candidates on the 128-bit key (score, id) in LDS, as k_rank_small.
// Results are bitwise those of the general EXACT path (tests/test_gpu_parity.py runs both and compares with the oracle).
#include "engine.h"
#include "pf.h"

#include <climits>
#include <mutex>
#include <cstdlib>
#include <cstring>

namespace rwr {

constexpr int SM_THREADS = 512;                  // 8 waves: up to 256 registers per lane (the per-thread place tables + the reduction)
constexpr int SM_WAVES = SM_THREADS / WAVE;
constexpr int SM_MAX_N = 6144;                 // nodes (M holds up to 2 n addends: 96 KiB)
constexpr int SM_MAX_ITEMS = 4096;             // candidates of the LDS sort
constexpr int SM_MCAP = 2 * SM_MAX_N;
constexpr int SM_R = 16;                       // addends per lane and pass of the exact reduction
constexpr int SM_PASS = WAVE * SM_R;           // 1024 addends per pass
constexpr int SM_SEQ = 256;                    // addends folded one by one before the passes start (the sum crosses a binade
                                               // every few addends while it is small; measured best, as in k_spmv_exact_hub)

struct SmCand {
    uint64_t hi, lo;
};

__device__ __forceinline__ int sm_pad(int q) { return q + (q >> 6); }

// s + v[0] + v[1] + ... + v[cnt-1], strictly in order, v >= 0: one wave; M is the padded LDS array, [base, base + cnt)
__device__ double sm_fold_seq(double s, const double *M, int base, int cnt, int lane)
{
    for (int i0 = 0; i0 < cnt; i0 += WAVE) {
        const int left = cnt - i0;
        const double mine = (lane < left) ? M[sm_pad(base + i0 + lane)] : 0.0;
        if (!__any(mine != 0.0)) continue;                      // +0.0 addends leave a non-negative sum as it is
        // (same address in every lane: an LDS broadcast; a full line is unrolled so that its 64 reads run ahead of the adds)
        if (left >= WAVE) {
#pragma unroll
            for (int t = 0; t < WAVE; ++t) s += M[sm_pad(base + i0 + t)];
        } else {
            for (int t = 0; t < left; ++t) s += M[sm_pad(base + i0 + t)];
        }
    }
    return s;
}

// the same sum for one pass of SM_PASS addends as an exact parallel reduction (pf.h: wave_fold_exact)
__device__ double sm_fold_scan(double s, const double *M, int base, int cnt, int lane)
{
    double v[SM_R];
#pragma unroll
    for (int u = 0; u < SM_R; ++u) {
        const int q = lane * SM_R + u;
        v[u] = q < cnt ? M[sm_pad(base + q)] : 0.0;
    }
    return wave_fold_exact<SM_R>(s, v, lane);
}

__device__ __forceinline__ void small_rwr_body(
    int32_t n, int32_t n_items, int32_t n_long, int32_t seed, double c1, int32_t n_iter, int32_t top_n, int scan_ok,
    const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src, const double *__restrict__ in_w,
    const uint8_t *__restrict__ dangling, const int32_t *__restrict__ row_order, const int64_t *__restrict__ rowptr,
    const int32_t *__restrict__ dst, const uint8_t *__restrict__ etype, const int32_t *__restrict__ item_rows,
    const int64_t *__restrict__ node_id, double *X, double *Y, int32_t *tab, int64_t *__restrict__ out_id,
    double *__restrict__ out_score, int32_t *__restrict__ out_count, int64_t *__restrict__ pin_id,
    double *__restrict__ pin_score, int32_t *__restrict__ pin_count)
{
    extern __shared__ double sm_lds[];
    double *M = sm_lds;                                            // [SM_MCAP + SM_MCAP / 64 + 1] padded addend sequence
    double *prod = sm_lds + (SM_MCAP + SM_MCAP / 64 + 64);         // [SM_WAVES][2][WAVE] staging lines of the long rows
    // per lane row (position in row_order): {list start, (row id << 17) | list length} -- read every step from LDS instead of
    // chasing row_order -> in_ptr through global memory (two dependent round trips per row and step)
    int2 *desc = reinterpret_cast<int2 *>(prod + (size_t)SM_WAVES * 2 * WAVE);   // [SM_MAX_N]
    __shared__ int cnt_s;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;

    // ---- Model ctor (Model.cs:42-49): rank[seed] = n, everything else 0
    for (int i = tid; i < n; i += SM_THREADS) { X[i] = (i == seed) ? (double)n : 0.0; }
    // ---- places in the seed row's addend sequence (static for the call)
    const int64_t sp0 = in_ptr[seed];
    const int32_t sdeg = (int32_t)(in_ptr[seed + 1] - sp0);
    const int32_t mlen = n + sdeg;
    // tab[i] = place of node i's restart addend: i + (number of links into the seed with source <= i);
    // tab[n + q] = place of link q: (its source) + q -- the restart addends of all earlier nodes and the links before it
    // (the seed's in-list sources are parked in LDS for the bisections: M is not in use yet)
    int32_t *ssrc = reinterpret_cast<int32_t *>(M);
#pragma unroll 1
    for (int32_t q = tid; q < sdeg; q += SM_THREADS) {
        const int32_t sq = in_src[sp0 + q];
        ssrc[q] = sq;
        tab[n + q] = sq + q;
    }
    __syncthreads();
#pragma unroll 1
    for (int32_t i = tid; i < n; i += SM_THREADS) {
        int32_t lo = 0, hi = sdeg;
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if (ssrc[mid] <= i) lo = mid + 1; else hi = mid;
        }
        tab[i] = i + lo;
    }
#pragma unroll 1
    for (int32_t r = n_long + tid; r < n; r += SM_THREADS) {
        const int32_t j = row_order[r];
        const int64_t p = in_ptr[j];
        desc[r] = make_int2((int)p, (int)(((uint32_t)j << 17) | (uint32_t)(in_ptr[j + 1] - p)));
    }
    __syncthreads();

    for (int it = 0; it < n_iter; ++it) {
        // ---- addends of the seed's own row
#pragma unroll 2
        for (int32_t i = tid; i < n; i += SM_THREADS) {
            const double xi = X[i];
            const double rw = c1 * xi;                                         // Model.cs:84
            M[sm_pad(tab[i])] = dangling[i] ? xi : (xi - rw);                  // Model.cs:97 / :91
        }
#pragma unroll 2
        for (int32_t q = tid; q < sdeg; q += SM_THREADS) {
            const double rw = c1 * X[in_src[sp0 + q]];                         // Model.cs:84
            M[sm_pad(tab[n + q])] = rw * in_w[sp0 + q];                        // Model.cs:87
        }
        __syncthreads();
        if (wave == 0) {
            // ---- the seed's row: M folded in order (Model.cs:85-93,96-97)
            double s = 0.0;
            if (scan_ok) {
                const int first = mlen < SM_SEQ ? mlen : SM_SEQ;
                s = sm_fold_seq(s, M, 0, first, lane);
                for (int base = SM_SEQ; base < mlen; base += SM_PASS)
                    s = sm_fold_scan(s, M, base, (mlen - base) < SM_PASS ? (mlen - base) : SM_PASS, lane);
            } else {
                s = sm_fold_seq(s, M, 0, mlen, lane);
            }
            if (lane == 0) Y[seed] = s;
        } else {
            // ---- every other row, in-degree descending (row_order): long rows a wave each, the rest a lane each
            double *pb = prod + (size_t)wave * 2 * WAVE;
            int buf = 0;
            for (int32_t r = wave - 1; r < n_long; r += SM_WAVES - 1) {
                const int32_t j = row_order[r];
                if (j == seed) continue;
                int64_t p = in_ptr[j];
                const int64_t e = in_ptr[j + 1];
                double acc = 0.0;
                double cur = 0.0;
                if (p + lane < e) { const double rw = c1 * X[in_src[p + lane]]; cur = rw * in_w[p + lane]; }
                while (p < e) {
                    const int64_t pn = p + WAVE;
                    double nxt = 0.0;
                    if (pn + lane < e) { const double rw = c1 * X[in_src[pn + lane]]; nxt = rw * in_w[pn + lane]; }
                    pb[buf * WAVE + lane] = cur;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (e - p >= WAVE) {
#pragma unroll
                        for (int t = 0; t < WAVE; ++t) acc += pb[buf * WAVE + t];      // list order (Model.cs:85-88)
                    } else {
                        const int c = (int)(e - p);
                        for (int t = 0; t < c; ++t) acc += pb[buf * WAVE + t];
                    }
                    buf ^= 1;
                    cur = nxt;
                    p = pn;
                }
                if (lane == 0) Y[j] = acc;
            }
            for (int32_t r = n_long + (tid - WAVE); r < n; r += SM_THREADS - WAVE) {
                const int2 dr = desc[r];
                const int32_t j = (int32_t)((uint32_t)dr.y >> 17);
                if (j == seed) continue;
                int32_t p = dr.x;
                const int32_t e = p + (int32_t)((uint32_t)dr.y & 0x1FFFFu);
                double acc = 0.0;
                for (; p + 4 <= e; p += 4) {
                    const int32_t i0 = in_src[p], i1 = in_src[p + 1], i2 = in_src[p + 2], i3 = in_src[p + 3];
                    const double w0 = in_w[p], w1 = in_w[p + 1], w2 = in_w[p + 2], w3 = in_w[p + 3];
                    const double x0 = X[i0], x1 = X[i1], x2 = X[i2], x3 = X[i3];
                    double rw;
                    rw = c1 * x0; acc += rw * w0;                                   // Model.cs:84,87 -- list order
                    rw = c1 * x1; acc += rw * w1;
                    rw = c1 * x2; acc += rw * w2;
                    rw = c1 * x3; acc += rw * w3;
                }
                for (; p < e; ++p) { const double rw = c1 * X[in_src[p]]; acc += rw * in_w[p]; }
                Y[j] = acc;
            }
        }
        __syncthreads();                                           // (workgroup scope: the one workgroup's stores are visible to it)
        { double *t = X; X = Y; Y = t; }                           // Model.updateRanks (Model.cs:103-108)
    }

    // ---- Recommender.cs:20-24,29: the seed's RAW out-links of type LIKE are not candidates
    for (int64_t p = rowptr[seed] + tid; p < rowptr[seed + 1]; p += SM_THREADS)
        if (etype[p] == RWR_EDGE_LIKE) X[dst[p]] = -1.0;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    // ---- Recommender.cs:27-38: candidates = ITEM nodes not excluded, sorted by (score desc, id desc)
    SmCand *sc = reinterpret_cast<SmCand *>(sm_lds);               // (M is no longer needed)
    int N2 = 1;
    while (N2 < n_items) N2 <<= 1;
    int mine = 0;
    for (int i = tid; i < N2; i += SM_THREADS) {
        SmCand c{0ull, 0ull};
        if (i < n_items) {
            const int32_t row = item_rows[i];
            const double sv = X[row];
            if (sv >= 0.0) { c.hi = f64_orderable(sv); c.lo = i64_orderable(node_id[row]); ++mine; }
        }
        sc[i] = c;
    }
    if (mine) atomicAdd(&cnt_s, mine);
    __syncthreads();
    for (int size = 2; size <= N2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < N2 / 2; t += SM_THREADS) {
                const int lo_i = 2 * t - (t & (stride - 1));
                const int hi_i = lo_i + stride;
                const bool desc = (lo_i & size) == 0;
                const SmCand a = sc[lo_i], b = sc[hi_i];
                const bool a_lt_b = (a.hi < b.hi) || (a.hi == b.hi && a.lo < b.lo);
                if (a_lt_b == desc) { sc[lo_i] = b; sc[hi_i] = a; }
            }
            __syncthreads();
        }
    }
    const int cnt = cnt_s;
    const int take = cnt < top_n ? cnt : top_n;
    // the list goes to the device tables (rwr_recommend_eval walks it there) AND straight into pinned host memory, so that
    // the call ends with one stream synchronisation instead of three copies
    if (tid == 0) { out_count[0] = take; if (pin_count) pin_count[0] = take; }
    for (int i = tid; i < take; i += SM_THREADS) {
        const SmCand v = sc[i];
        const int64_t id = (int64_t)(v.lo ^ 0x8000000000000000ull);
        const uint64_t u = (v.hi & 0x8000000000000000ull) ? (v.hi ^ 0x8000000000000000ull) : ~v.hi;
        double sv;
        __builtin_memcpy(&sv, &u, 8);
        out_id[i] = id;
        out_score[i] = sv;
        if (pin_id) {                                  // (a batch of graphs keeps its lists on the device: multi.hip)
            pin_id[i] = id;
            pin_score[i] = sv;
        }
    }
}

__global__ __launch_bounds__(SM_THREADS) void k_small_rwr(
    int32_t n, int32_t n_items, int32_t n_long, int32_t seed, double c1, int32_t n_iter, int32_t top_n, int scan_ok,
    const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src, const double *__restrict__ in_w,
    const uint8_t *__restrict__ dangling, const int32_t *__restrict__ row_order, const int64_t *__restrict__ rowptr,
    const int32_t *__restrict__ dst, const uint8_t *__restrict__ etype, const int32_t *__restrict__ item_rows,
    const int64_t *__restrict__ node_id, double *X, double *Y, int32_t *tab, int64_t *__restrict__ out_id,
    double *__restrict__ out_score, int32_t *__restrict__ out_count, int64_t *__restrict__ pin_id,
    double *__restrict__ pin_score, int32_t *__restrict__ pin_count)
{
    small_rwr_body(n, n_items, n_long, seed, c1, n_iter, top_n, scan_ok, in_ptr, in_src, in_w, dangling, row_order, rowptr, dst, etype,
                   item_rows, node_id, X, Y, tab, out_id, out_score, out_count, pin_id, pin_score, pin_count);
}

// one workgroup per graph of a batch (rwr_eval_graphs): the ranked lists stay in each graph's device tables
struct SmallRwrArgs {
    int32_t n, n_items, n_long, seed, top_n, pad0;
    const int64_t *in_ptr; const int32_t *in_src; const double *in_w; const uint8_t *dangling; const int32_t *row_order;
    const int64_t *rowptr; const int32_t *dst; const uint8_t *etype; const int32_t *item_rows; const int64_t *node_id;
    double *X, *Y; int32_t *tab; int64_t *out_id; double *out_score; int32_t *out_count;
};
__global__ __launch_bounds__(SM_THREADS) void k_small_rwr_multi(const SmallRwrArgs *__restrict__ args, double c1, int32_t n_iter,
                                                                int scan_ok)
{
    const SmallRwrArgs a = args[blockIdx.x];
    small_rwr_body(a.n, a.n_items, a.n_long, a.seed, c1, n_iter, a.top_n, scan_ok, a.in_ptr, a.in_src, a.in_w, a.dangling, a.row_order,
                   a.rowptr, a.dst, a.etype, a.item_rows, a.node_id, a.X, a.Y, a.tab, a.out_id, a.out_score, a.out_count, nullptr,
                   nullptr, nullptr);
}

// the seed row's addend sequence (one restart addend per node + one addend per link INTO the seed; multi-edges can make
// the latter exceed n) must fit the LDS array M
bool small_path_seed_ok(const rwr_graph *g, int32_t seed)
{
    if (seed < 0 || seed >= g->n) return false;
    const int64_t sdeg = g->h_in_ptr[(size_t)seed + 1] - g->h_in_ptr[(size_t)seed];
    return (int64_t)g->n + sdeg <= (int64_t)SM_MCAP;
}

// layout of the pinned result buffer recommend_small leaves behind (rwr_recommend copies the list out of it)
const int64_t *small_pin_ids(const rwr_graph *g) { return reinterpret_cast<const int64_t *>(g->sm_pin); }
const double *small_pin_scores(const rwr_graph *g) { return reinterpret_cast<const double *>(small_pin_ids(g) + SM_MAX_ITEMS); }
// makes sure the handle has its pinned result buffer (SM_MAX_ITEMS ids, SM_MAX_ITEMS scores, a count)
int32_t small_pin_ensure(rwr_graph *g)
{
    if (!g->sm_pin) RWR_HIP(hipHostMalloc(&g->sm_pin, SM_MAX_ITEMS * 16 + 64, hipHostMallocMapped | hipHostMallocPortable));
    return RWR_OK;
}
// the score half of that buffer as scratch for an entry point that does not hand the list out (rank.hip: eval_ranked)
void *small_pin_scratch(rwr_graph *g) { return reinterpret_cast<int64_t *>(g->sm_pin) + SM_MAX_ITEMS; }
int64_t small_pin_words() { return SM_MAX_ITEMS; }

bool small_path_ok(const rwr_graph *g)
{
    static const int env = [] { const char *e = getenv("RWR_SMALL"); return e ? atoi(e) : 1; }();
    // one workgroup walks every link once per step: beyond ~65 K links the general path's whole-chip kernels are ahead
    // (measured, round 3: 4 000 nodes / 52 K links 0.54 vs 0.61 ms, 4 700 nodes / 76 K links 0.80 vs 0.66 ms, 5 500 nodes / 92 K links
    //  0.95 vs 0.60 ms; round 2, before the general path's chain was rebuilt: 55 K links 0.77 vs 1.37 ms) -- which is also the
    // largest graph the one-launch build takes (build.hip: STAGE_MAX_M)
    return env && g->n <= SM_MAX_N && g->n_items <= SM_MAX_ITEMS && g->n_items > 0 && g->nonneg && g->nnz <= 65536;
}

// One single-seed Recommendation as one launch.  The ranked list is left in d_out_id / d_out_score (row 0) on the device
// and in the handle's pinned host buffer; ids / scores (may be null) receive it, *count its length.  The caller has
// validated seed, d and top_n.
int32_t recommend_small(rwr_graph *g, int32_t seed, double d, int32_t n_iter, int32_t top_n, int64_t *ids, double *scores,
                        int32_t *count)
{
    const int32_t n = g->n;
    hipStream_t s = g->stream;
    RWR_TRY(ensure_in_w(g));                                       // (weighted form: ego networks carry MENTION weights)
    RWR_TRY(g->X.ensure((size_t)n));
    RWR_TRY(g->Y.ensure((size_t)n));
    RWR_TRY(g->d_out_id.ensure((size_t)SM_MAX_ITEMS + 64));
    RWR_TRY(g->d_out_score.ensure((size_t)SM_MAX_ITEMS + 64));
    RWR_TRY(g->d_counts.ensure(64));
    RWR_TRY(g->sm_tab.ensure((size_t)SM_MCAP));
    constexpr size_t smem = ((size_t)SM_MCAP + SM_MCAP / 64 + 64 + (size_t)SM_WAVES * 2 * WAVE) * sizeof(double) +
                            (size_t)SM_MAX_N * sizeof(int2);
    static_assert(smem >= (size_t)SM_MAX_ITEMS * sizeof(SmCand), "the sort re-uses the addend buffer");
    RWR_TRY(small_pin_ensure(g));
    {   // the kernel's LDS attribute: once per device and process
        static std::mutex attr_mu1;
        static bool attr_done1[64];
        std::lock_guard<std::mutex> lk(attr_mu1);
        if (g->device >= 0 && g->device < 64 && !attr_done1[g->device]) {
            RWR_HIP(hipFuncSetAttribute((const void *)k_small_rwr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_done1[g->device] = true;
        }
    }
    int64_t *pin_id = reinterpret_cast<int64_t *>(g->sm_pin);
    double *pin_score = reinterpret_cast<double *>(pin_id + SM_MAX_ITEMS);
    int32_t *pin_count = reinterpret_cast<int32_t *>(pin_score + SM_MAX_ITEMS);
    const double c1 = 1 - d;                                       // Model.cs:84
    const int scan_ok = (c1 >= 0.0 && c1 <= 1.0) ? 1 : 0;   // ranks >= 0: the binade reduction's precondition
    hipLaunchKernelGGL(k_small_rwr, dim3(1), dim3(SM_THREADS), smem, s, n, g->n_items, g->bin_end[0], seed, c1, n_iter, top_n,
                       scan_ok, g->in_ptr.p, g->in_src.p, g->in_w.p, g->dangling.p, g->row_order.p, g->rowptr.p, g->dst.p,
                       g->etype.p, g->item_rows.p, g->node_id.p, g->X.p, g->Y.p, g->sm_tab.p, g->d_out_id.p, g->d_out_score.p,
                       g->d_counts.p, pin_id, pin_score, pin_count);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(s));
    const int32_t cnt = *pin_count;
    if (ids && scores) {
        memcpy(ids, pin_id, sizeof(int64_t) * (size_t)cnt);
        memcpy(scores, pin_score, sizeof(double) * (size_t)cnt);
        for (int32_t i = cnt; i < top_n; ++i) { ids[i] = 0; scores[i] = 0.0; }     // (rest of the row: id 0 / score 0)
    }
    g->sm_pin_count = cnt;
    *count = cnt;
    return RWR_OK;
}


// One single-seed Recommendation on each of `count` ego-network-sized graphs as ONE launch (every graph must pass
// small_path_ok / small_path_seed_ok).  Enqueued on `s`; the full ranked lists are left in each graph's d_out_id /
// d_out_score / d_counts.  args_keep receives the kernel's argument table.
int32_t recommend_small_multi(rwr_graph **gs, const int32_t *seeds, int32_t count, double d, int32_t n_iter, hipStream_t s,
                              DevBuf<uint8_t> &args_keep)
{
    if (count <= 0) return RWR_OK;
    constexpr size_t smem = ((size_t)SM_MCAP + SM_MCAP / 64 + 64 + (size_t)SM_WAVES * 2 * WAVE) * sizeof(double) +
                            (size_t)SM_MAX_N * sizeof(int2);
    static std::mutex attr_mu;
    static bool attr_done[64];                   // per device, once per process
    int dev = 0;
    RWR_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(attr_mu);
        if (dev >= 0 && dev < 64 && !attr_done[dev]) {
            RWR_HIP(hipFuncSetAttribute((const void *)k_small_rwr_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_done[dev] = true;
        }
    }
    void *h_v = nullptr;
    RWR_TRY(multi_pinned(2, sizeof(SmallRwrArgs) * (size_t)count, &h_v));      // (pinned: see build.hip, multi_pinned)
    SmallRwrArgs *h = static_cast<SmallRwrArgs *>(h_v);
    for (int32_t i = 0; i < count; ++i) {
        rwr_graph *g = gs[i];
        RWR_TRY(ensure_in_w(g));
        RWR_TRY(g->X.ensure((size_t)g->n));
        RWR_TRY(g->Y.ensure((size_t)g->n));
        RWR_TRY(g->d_out_id.ensure((size_t)SM_MAX_ITEMS + 64));
        RWR_TRY(g->d_out_score.ensure((size_t)SM_MAX_ITEMS + 64));
        RWR_TRY(g->d_counts.ensure(64));
        RWR_TRY(g->sm_tab.ensure((size_t)SM_MCAP));
        SmallRwrArgs &a = h[i];
        a.n = g->n; a.n_items = g->n_items; a.n_long = g->bin_end[0]; a.seed = seeds[i]; a.top_n = g->n_items; a.pad0 = 0;
        a.in_ptr = g->in_ptr.p; a.in_src = g->in_src.p; a.in_w = g->in_w.p; a.dangling = g->dangling.p; a.row_order = g->row_order.p;
        a.rowptr = g->rowptr.p; a.dst = g->dst.p; a.etype = g->etype.p; a.item_rows = g->item_rows.p; a.node_id = g->node_id.p;
        a.X = g->X.p; a.Y = g->Y.p; a.tab = g->sm_tab.p; a.out_id = g->d_out_id.p; a.out_score = g->d_out_score.p;
        a.out_count = g->d_counts.p;
        g->sm_pin_count = -1;
        g->stats.seeds_done += 1;
    }
    RWR_TRY(args_keep.alloc(sizeof(SmallRwrArgs) * (size_t)count));
    SmallRwrArgs *d_args_p = reinterpret_cast<SmallRwrArgs *>(args_keep.p);
    RWR_HIP(hipMemcpyAsync(d_args_p, h, sizeof(SmallRwrArgs) * (size_t)count, hipMemcpyHostToDevice, s));
    const double c1 = 1 - d;                                       // Model.cs:84
    const int scan_ok = (c1 >= 0.0 && c1 <= 1.0) ? 1 : 0;
    hipLaunchKernelGGL(k_small_rwr_multi, dim3((unsigned)count), dim3(SM_THREADS), smem, s, d_args_p, c1, n_iter, scan_ok);
    RWR_HIP(hipGetLastError());
    // (no synchronisation: the evaluation kernel follows on the same stream; the caller keeps args_keep until that stream has
    //  been synchronised -- the device-memory cache is shared by all threads)
    return RWR_OK;
}

}  // namespace rwr
