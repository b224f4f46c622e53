// K = 1 on DENSE graphs (the MovieLens-shaped configuration: hundreds of links per node, a rank vector of a few MB):
// the single-seed SpMV with the gathered vector staged through LDS.
//
// Why: the list-order kernels of spmv.hip gather z with one 8-byte global load per entry; each moves a 64-byte sector
// from the L2 through the L1, so they run at the L2's request rate (~150 G gathers/s measured), an order of magnitude
// under what the 4-byte index stream costs in HBM time.  Here the SOURCE range is cut into blocks of BK_B nodes whose z
// values (64 KiB) sit in LDS; a workgroup walks the blocks in order and, for every row it owns, adds the entries whose
// source lies in the current block.  In-lists are sorted by source (stable transpose, build.hip), so the entries of a
// (row, block) pair are one contiguous piece of the row's list and the row's running sum simply carries on from block to
// block: not a single addition changes order -- bit for bit the sums of Model.deliverRanks (Model.cs:85-88).
// What is left in global memory traffic is the index stream (4 bytes per entry, read once), the block-boundary table
// (4 bytes per row and block) and one z block per workgroup and block step, served by the L2.
//
// Needs the value-free form (engine.h: rwr_graph::vf -- z, not (x, weight), is what is gathered) and a graph dense
// enough that a (row, block) piece holds several entries on average; otherwise the launcher leaves the step to spmv.hip.
//
// Work layout (static, so that every running sum lives in a register):
//   hub rows   in-degree >= hub_t: ONE WAVE per row and block piece -- 64 entries at a time are fetched coalesced, their
//              z values read from LDS in parallel, and the adds run in list order off an LDS staging line (the shape of
//              spmv_exact_wave).  FAST mode sums per lane and folds the wave once at the end.
//   lane rows  everything else, in groups of 64 rows of similar degree: one LANE per row, each walking its own piece.
// Hub h belongs to workgroup h % NWG (the longest rows are spread over all workgroups), group q to wave q % nwaves.
#include "engine.h"
#include "pf.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace rwr {

constexpr int BK_B = 8192;           // source nodes per LDS block (64 KiB of z; two blocks resident: one in use, one arriving)
constexpr int BK_THREADS = 512;      // 8 waves, one workgroup per CU: up to 256 registers per lane for the index queues
constexpr int BK_WAVES = BK_THREADS / WAVE;
constexpr int BK_MAXH = 4;           // hub rows per wave
constexpr int BK_MAXG = 2;           // lane groups per wave
constexpr int BK_PRE = BK_B / BK_THREADS;   // doubles per thread and block fetch

typedef int v4i_bu __attribute__((ext_vector_type(4), aligned(4)));

// bp[b * n + pos] = number of entries of row border[pos] whose source is < b * BK_B   (b = 0 .. nblk)
__global__ __launch_bounds__(256) void k_bk_bounds(int32_t n, int nblk, const int32_t *__restrict__ border,
                                                   const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                   int32_t *__restrict__ bp)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)n * (nblk + 1)) return;
    const int b = (int)(q / n);
    const int32_t pos = (int32_t)(q % n);
    const int32_t row = border[pos];
    const int64_t p0 = in_ptr[row];
    const int32_t deg = (int32_t)(in_ptr[row + 1] - p0);
    const int64_t target = (int64_t)b * BK_B;
    int32_t lo = 0, hi = deg;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)in_src[p0 + mid] < target) lo = mid + 1; else hi = mid;
    }
    bp[q] = lo;
}

template <bool FAST>
__global__ __launch_bounds__(BK_THREADS) void k_spmv_blocked(
    int32_t n, int nblk, int32_t n_hub, int32_t n_groups, const int64_t *__restrict__ in_ptr,
    const int32_t *__restrict__ in_src, const int32_t *__restrict__ border, const int32_t *__restrict__ bp,
    const double *__restrict__ z, double *__restrict__ y, double *__restrict__ zout, const double *__restrict__ w_src,
    const int32_t *__restrict__ seeds, int skip_seed_row, double c1, int dbg)
{
    extern __shared__ double lds[];
    double *zb = lds;                                          // [2][BK_B]
    double *pbase = lds + 2 * BK_B;                            // [BK_WAVES][2][WAVE] staging lines of the hub adds
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int wg = blockIdx.x, nwg = gridDim.x;
    const int64_t nwaves = (int64_t)nwg * BK_WAVES;
    const int32_t my_seed = skip_seed_row ? seeds[0] : -1;
    double *pb = pbase + (size_t)wave * 2 * WAVE;

    // ---- this wave's rows
    int32_t hpos[BK_MAXH];
    int64_t hbase[BK_MAXH];
    double hacc[BK_MAXH];
#pragma unroll
    for (int i = 0; i < BK_MAXH; ++i) {
        const int64_t h = (int64_t)wg + (int64_t)nwg * (wave + BK_WAVES * i);
        hpos[i] = h < n_hub ? (int32_t)h : -1;
        hbase[i] = hpos[i] >= 0 ? in_ptr[border[hpos[i]]] : 0;
        hacc[i] = 0.0;
    }
    int32_t gpos[BK_MAXG];
    int64_t gbase[BK_MAXG];
    double gacc[BK_MAXG];
#pragma unroll
    for (int i = 0; i < BK_MAXG; ++i) {
        const int64_t q = (int64_t)wave * nwg + wg + nwaves * i;
        const int64_t pos = (int64_t)n_hub + q * WAVE + lane;
        gpos[i] = (q < n_groups && pos < n) ? (int32_t)pos : -1;
        gbase[i] = gpos[i] >= 0 ? in_ptr[border[gpos[i]]] : 0;
        gacc[i] = 0.0;
    }

    // ---- block 0 straight into LDS
#pragma unroll
    for (int u = 0; u < BK_PRE; ++u) {
        const int64_t i = (int64_t)u * BK_THREADS + tid;
        zb[u * BK_THREADS + tid] = i < n ? z[i] : 0.0;
    }
    __syncthreads();

    // lane rows: every lane streams ITS row's index list through a two-chunk register queue (16 indices per chunk, the next
    // chunk always in flight); a row's list is consumed front to back across the blocks, so the queue never restarts and
    // the fetches run a chunk ahead of the adds whatever the piece lengths are (in_src is padded: over-reads are unused)
    int32_t gp[BK_MAXG], gpe[BK_MAXG], gc[BK_MAXG];
    v4i_bu gq[BK_MAXG][4], gqn[BK_MAXG][4];
#pragma unroll
    for (int i = 0; i < BK_MAXG; ++i) {
        gp[i] = 0;
        gc[i] = 0;
        gpe[i] = gpos[i] >= 0 ? bp[(size_t)n + gpos[i]] : 0;
        const int32_t *src = in_src + gbase[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gq[i][j] = *reinterpret_cast<const v4i_bu *>(src + 4 * j);
            gqn[i][j] = *reinterpret_cast<const v4i_bu *>(src + 16 + 4 * j);
        }
    }
    int32_t hp[BK_MAXH], hpe[BK_MAXH];
#pragma unroll
    for (int i = 0; i < BK_MAXH; ++i) {
        hp[i] = 0;
        hpe[i] = hpos[i] >= 0 ? bp[(size_t)n + hpos[i]] : 0;
    }

    int buf = 0;
    constexpr int D = 8;                                       // hub rows: index fetches in flight per lane
    for (int b = 0; b < nblk; ++b) {
        const double *zc = zb + (size_t)(b & 1) * BK_B;
        const int32_t s0 = b * BK_B;
        const bool more = b + 1 < nblk;
        // the next block's z values start their way in now and are parked in LDS after this block's adds
        double pre[BK_PRE];
        if (more) {
#pragma unroll
            for (int u = 0; u < BK_PRE; ++u) {
                const int64_t i = (int64_t)(b + 1) * BK_B + (int64_t)u * BK_THREADS + tid;
                pre[u] = i < n ? z[i] : 0.0;
            }
        }
        // next step's piece ends
        int32_t gpe_n[BK_MAXG], hpe_n[BK_MAXH];
#pragma unroll
        for (int i = 0; i < BK_MAXG; ++i) gpe_n[i] = (more && gpos[i] >= 0) ? bp[(size_t)(b + 2) * n + gpos[i]] : gpe[i];
#pragma unroll
        for (int i = 0; i < BK_MAXH; ++i) hpe_n[i] = (more && hpos[i] >= 0) ? bp[(size_t)(b + 2) * n + hpos[i]] : hpe[i];

        // ---- hub rows: a wave per piece, D rounds of 64 indices in flight
#pragma unroll
        for (int i = 0; i < BK_MAXH; ++i) {
            const int32_t ps = hp[i], pe = hpe[i];
            if (ps >= pe || (dbg & 1)) continue;
            const int32_t *src = in_src + hbase[i];
            int32_t q[D];
#pragma unroll
            for (int d = 0; d < D; ++d) q[d] = (ps + d * WAVE + lane < pe) ? src[ps + d * WAVE + lane] : -1;
            double acc = hacc[i];
            int32_t p = ps;
            while (p < pe) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (p < pe) {                                   // (wave-uniform)
                        const int32_t idx = q[d];
                        const int32_t pf = p + D * WAVE + lane;
                        q[d] = pf < pe ? src[pf] : -1;              // refill the slot: the fetch for D rounds ahead
                        const double cur = idx >= 0 ? zc[idx - s0] : 0.0;
                        if (FAST) {
                            acc += cur;
                        } else {
                            pb[buf * WAVE + lane] = cur;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            if (pe - p >= WAVE) {
#pragma unroll
                                for (int t = 0; t < WAVE; ++t) acc += pb[buf * WAVE + t];   // list order (Model.cs:85-88)
                            } else {
                                const int cnt = pe - p;
                                for (int t = 0; t < cnt; ++t) acc += pb[buf * WAVE + t];
                            }
                            buf ^= 1;
                        }
                        p += WAVE;
                    }
                }
            }
            hacc[i] = acc;
        }
        // ---- lane rows: a lane per piece, chunk by chunk off the register queue
#pragma unroll
        for (int i = 0; i < BK_MAXG; ++i) {
            const int32_t ps = gp[i], pe = gpe[i];
            bool go = ps < pe && !(dbg & 2);
            if (!__any(go)) continue;
            const int32_t *src = in_src + gbase[i];
            double acc = gacc[i];
            while (__any(go)) {
                if (go) {
                    const int32_t c0 = gc[i] * 16;
                    if (c0 >= pe) {
                        go = false;
                    } else {
                        double xv[16];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const v4i_bu v = gq[i][j];
                            const int32_t pos = c0 + 4 * j;
                            xv[4 * j + 0] = zc[((pos + 0 >= ps && pos + 0 < pe) ? v.x : s0) - s0];
                            xv[4 * j + 1] = zc[((pos + 1 >= ps && pos + 1 < pe) ? v.y : s0) - s0];
                            xv[4 * j + 2] = zc[((pos + 2 >= ps && pos + 2 < pe) ? v.z : s0) - s0];
                            xv[4 * j + 3] = zc[((pos + 3 >= ps && pos + 3 < pe) ? v.w : s0) - s0];
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            if (c0 + j >= ps && c0 + j < pe) acc += xv[j];       // list order (Model.cs:85-88)
                        if (c0 + 16 <= pe) {            // chunk consumed: the queued one moves up, the one after it is fetched
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                gq[i][j] = gqn[i][j];
                                gqn[i][j] = *reinterpret_cast<const v4i_bu *>(src + c0 + 32 + 4 * j);
                            }
                            gc[i] += 1;
                        } else {
                            go = false;                 // the rest of this chunk belongs to the next block
                        }
                    }
                }
            }
            gacc[i] = acc;
        }
        // ---- rotate to the next block
#pragma unroll
        for (int i = 0; i < BK_MAXG; ++i) { gp[i] = gpe[i]; gpe[i] = gpe_n[i]; }
#pragma unroll
        for (int i = 0; i < BK_MAXH; ++i) { hp[i] = hpe[i]; hpe[i] = hpe_n[i]; }
        if (more) {
            double *zn = zb + (size_t)((b + 1) & 1) * BK_B;
#pragma unroll
            for (int u = 0; u < BK_PRE; ++u) zn[u * BK_THREADS + tid] = pre[u];
        }
        __syncthreads();
    }

    // ---- results (+ the rows' own z for the next step)
#pragma unroll
    for (int i = 0; i < BK_MAXH; ++i) {
        if (hpos[i] < 0) continue;
        double acc = hacc[i];
        if (FAST) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
        }
        const int32_t row = border[hpos[i]];
        if (lane == 0 && row != my_seed) {
            y[row] = acc;
            if (zout) { const double rw = c1 * acc; zout[row] = rw * w_src[row]; }
        }
    }
#pragma unroll
    for (int i = 0; i < BK_MAXG; ++i) {
        if (gpos[i] < 0) continue;
        const int32_t row = border[gpos[i]];
        if (row != my_seed) {
            y[row] = gacc[i];
            if (zout) { const double rw = c1 * gacc[i]; zout[row] = rw * w_src[row]; }
        }
    }
}

// Decides whether the blocked kernel serves this graph and builds its tables (row order: hub rows by in-degree
// descending, then the lane rows by in-degree descending; block boundaries of every row).  Cheap: one sort of n keys on
// the host (the graphs this path takes have at most a few hundred thousand nodes) and one binary-search kernel.
int32_t blocked_prepare(rwr_graph *g)
{
    if (g->bk_state != 0) return RWR_OK;
    g->bk_state = -1;                                            // not eligible unless everything below holds
    // OFF by default: measured on the MovieLens-shaped configuration (MI355X, DESIGN.md "K = 1 on the dense configuration")
    // the block sweep loses to the L2-gather kernels of spmv.hip -- 28 barrier-separated, latency-bound block steps cost
    // 76 us before a single entry is added (FAST: 341 vs 198 us per step, EXACT: 651 vs 381).  Kept as a tested experiment
    // (RWR_SPMV_BLOCKED=1; bitwise equal results).
    static const int env = [] { const char *e = getenv("RWR_SPMV_BLOCKED"); return e ? atoi(e) : 0; }();
    static const double min_piece = [] { const char *e = getenv("RWR_BLOCKED_MIN_PIECE"); return e ? atof(e) : 4.0; }();
    const int32_t n = g->n;
    if (!env || !g->vf || n <= 0 || g->nnz <= 0) return RWR_OK;
    const int nblk = (int)(((int64_t)n + BK_B - 1) / BK_B);
    // average entries of a (row, block) piece, counting for a row only the blocks that can hold its sources at all
    // (bipartite graphs: half of them); below ~4 the per-piece bookkeeping costs more than the LDS gathers save
    const double piece = (double)g->nnz / ((double)n * (double)nblk);
    if (env < 2 && (piece < min_piece || nblk > 512)) return RWR_OK;
    hipDeviceProp_t prop;
    RWR_HIP(hipGetDeviceProperties(&prop, g->device));
    const int nwg = prop.multiProcessorCount;                    // one 1024-thread workgroup per CU (144 KiB of LDS each)
    const int64_t nwaves = (int64_t)nwg * BK_WAVES;
    // hub threshold: at least 1024 in-links, and no more hub rows than the waves can hold
    std::vector<int32_t> order((size_t)n);
    for (int32_t i = 0; i < n; ++i) order[i] = i;
    auto deg = [&](int32_t i) { return g->h_in_ptr[i + 1] - g->h_in_ptr[i]; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return deg(a) > deg(b); });
    static const int hub_env = [] { const char *e = getenv("RWR_BLOCKED_HUB"); return e ? atoi(e) : 1024; }();
    int64_t n_hub = 0;
    while (n_hub < n && deg(order[n_hub]) >= hub_env && n_hub < nwaves * BK_MAXH) ++n_hub;
    const int64_t n_groups = ((int64_t)n - n_hub + WAVE - 1) / WAVE;
    if (n_groups > nwaves * BK_MAXG) return RWR_OK;
    RWR_TRY(g->bk_border.alloc((size_t)n));
    RWR_TRY(g->bk_bp.alloc((size_t)n * (size_t)(nblk + 1)));
    RWR_HIP(hipMemcpyAsync(g->bk_border.p, order.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, g->stream));
    const int64_t tot = (int64_t)n * (nblk + 1);
    hipLaunchKernelGGL(k_bk_bounds, dim3(cdiv((size_t)tot, 256)), dim3(256), 0, g->stream, n, nblk, g->bk_border.p, g->in_ptr.p,
                       g->in_src.p, g->bk_bp.p);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(g->stream));                    // `order` is released on return
    g->bk_nblk = nblk;
    g->bk_nhub = (int32_t)n_hub;
    g->bk_ngroups = (int32_t)n_groups;
    g->bk_nwg = nwg;
    g->bk_state = 1;
    return RWR_OK;
}

bool blocked_ready(const rwr_graph *g) { return g->bk_state == 1; }

void launch_spmv_blocked(rwr_graph *g, const double *zin, double *Y, double *zout, const int32_t *seeds, int skip, double c1,
                         bool fast, hipStream_t s)
{
    constexpr size_t smem = (2 * (size_t)BK_B + (size_t)BK_WAVES * 2 * WAVE) * sizeof(double);
    // timing diagnostics only (results are wrong with any bit set): 1 = skip the hub rows, 2 = skip the lane rows
    static const int dbg = [] { const char *e = getenv("RWR_BLOCKED_DBG"); return e ? atoi(e) : 0; }();
    if (fast) {
        (void)hipFuncSetAttribute((const void *)k_spmv_blocked<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k_spmv_blocked<true>, dim3(g->bk_nwg), dim3(BK_THREADS), smem, s, g->n, g->bk_nblk, g->bk_nhub,
                           g->bk_ngroups, g->in_ptr.p, g->in_src.p, g->bk_border.p, g->bk_bp.p, zin, Y, zout, g->w_src.p, seeds,
                           skip, c1, dbg);
    } else {
        (void)hipFuncSetAttribute((const void *)k_spmv_blocked<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k_spmv_blocked<false>, dim3(g->bk_nwg), dim3(BK_THREADS), smem, s, g->n, g->bk_nblk, g->bk_nhub,
                           g->bk_ngroups, g->in_ptr.p, g->in_src.p, g->bk_border.p, g->bk_bp.p, zin, Y, zout, g->w_src.p, seeds,
                           skip, c1, dbg);
    }
}

}  // namespace rwr
