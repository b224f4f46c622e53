// K = 1: the SpMV of a single seed -- y = (1-d) P^T x  (Model.deliverRanks, Model.cs:78-88, one rank vector).
// Every row is summed in the reference's addend order (bitwise); rows are binned by in-degree so that long rows
// are served cooperatively (parallel loads / gathers / products, sequential adds).
#include "engine.h"
#include "pf.h"

#include <cstdlib>

namespace rwr {

// K = 1, EXACT mode: one lane per destination row, the row's in-list walked sequentially in the reference's addend
// order (bitwise).  Rows come in in-degree order, so the lanes of a wave have lists of similar length.  Each lane
// streams ITS list with wide loads -- 8 indices as two 16-byte loads, 8 weights as four -- so a fetched 64-byte
// sector is consumed whole by the lane that fetched it (per-entry 4- and 8-byte loads would move one sector per
// entry through the L1), and keeps the 8 gathers of x in flight before the first dependent add.
typedef int v4i_u __attribute__((ext_vector_type(4), aligned(4)));
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
// (bid, nblk): this workgroup's index and the number of workgroups serving the bin -- the four bins share ONE launch
// (k_spmv_exact_binned below), so that their tails overlap instead of queueing behind each other
// VF (all bodies): value-free form (engine.h: rwr_graph::vf) -- x is then the z vector, z[i] = ((1-d) x[i]) * w_src[i],
// the product Model.cs:84,87 forms for every link of source i; no weights are read, the row's sum is the list-order sum
// of the gathered z, and the row's own z for the next step goes to zout (null on the last step).
#define RWR_SPMV_STORE(J, ACC)                                                          \
    {                                                                                   \
        if ((J) != my_seed) {                                                           \
            y[(J)] = (ACC);                                                             \
            if (VF && zout) { const double rw__ = c1 * (ACC); zout[(J)] = rw__ * w_src[(J)]; } \
        }                                                                               \
        if (nz_out && (ACC) != 0.0) atomicOr(&nz_out[(uint32_t)(J) >> 5], 1u << ((J) & 31)); \
    }
template <bool VF>
__device__ __forceinline__ void spmv_exact_lane(int bid, int nblk, int32_t r0, int32_t n, const int64_t *__restrict__ in_ptr,
                                                const int32_t *__restrict__ in_src, const double *__restrict__ in_w,
                                                const int32_t *__restrict__ row_order, const double *__restrict__ x,
                                                double *__restrict__ y, int32_t my_seed, double c1,
                                                const uint32_t *__restrict__ act, uint32_t *__restrict__ nz_out,
                                                const double *__restrict__ w_src, double *__restrict__ zout)
{
    const int64_t stride = (int64_t)nblk * blockDim.x;
    for (int64_t r = (int64_t)r0 + (int64_t)bid * blockDim.x + threadIdx.x; r < n; r += stride) {
        const int32_t j = row_order[r];
        int64_t p = in_ptr[j];
        int64_t e = in_ptr[j + 1];
        // first iterations: a row none of whose in-neighbours holds a non-zero (k_mark_active) stays exactly +0.0
        if (act && !((act[(uint32_t)j >> 5] >> (j & 31)) & 1u)) e = p;
        double acc = 0.0;
        for (; p + 8 <= e; p += 8) {
            const v4i_u i0 = *reinterpret_cast<const v4i_u *>(in_src + p);
            const v4i_u i1 = *reinterpret_cast<const v4i_u *>(in_src + p + 4);
            if (VF) {
                const double x0 = x[i0.x], x1 = x[i0.y], x2 = x[i0.z], x3 = x[i0.w];
                const double x4 = x[i1.x], x5 = x[i1.y], x6 = x[i1.z], x7 = x[i1.w];
                acc += x0; acc += x1; acc += x2; acc += x3;     // z of the sources, in list order
                acc += x4; acc += x5; acc += x6; acc += x7;
            } else {
                const v2d_u w0 = *reinterpret_cast<const v2d_u *>(in_w + p);
                const v2d_u w1 = *reinterpret_cast<const v2d_u *>(in_w + p + 2);
                const v2d_u w2 = *reinterpret_cast<const v2d_u *>(in_w + p + 4);
                const v2d_u w3 = *reinterpret_cast<const v2d_u *>(in_w + p + 6);
                const double x0 = x[i0.x], x1 = x[i0.y], x2 = x[i0.z], x3 = x[i0.w];
                const double x4 = x[i1.x], x5 = x[i1.y], x6 = x[i1.z], x7 = x[i1.w];
                double rw;
                rw = c1 * x0; acc += rw * w0.x;      // Model.cs:84,87 -- in list order
                rw = c1 * x1; acc += rw * w0.y;
                rw = c1 * x2; acc += rw * w1.x;
                rw = c1 * x3; acc += rw * w1.y;
                rw = c1 * x4; acc += rw * w2.x;
                rw = c1 * x5; acc += rw * w2.y;
                rw = c1 * x6; acc += rw * w3.x;
                rw = c1 * x7; acc += rw * w3.y;
            }
        }
        {   // the last (up to 7) entries: all loads issued before the first dependent add
            const int cnt = (int)(e - p);
            int32_t ti[7];
            double tw[7], tx[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                ti[u] = u < cnt ? in_src[p + u] : 0;
                tw[u] = (!VF && u < cnt) ? in_w[p + u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) tx[u] = u < cnt ? x[ti[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (u < cnt) {
                    if (VF) acc += tx[u];
                    else { const double rw = c1 * tx[u]; acc += rw * tw[u]; }
                }
        }
        RWR_SPMV_STORE(j, acc)
    }
}

// K = 1, EXACT mode, LONG rows (in-degree >= 128: hub items, the ego of an ego network): one lane walking such a list
// alone pays a memory round trip per few entries and the longest row becomes the kernel's run time.  Here a whole wave
// serves the row: 64 consecutive entries are loaded coalesced, their 64 gathers of x fly together and the products
// rw * weight (Model.cs:84,87) are formed in parallel -- only the ADDS stay sequential, in list order, each taking
// the next product from its lane (v_readlane) into the wave-uniform accumulator:  8 cycles per entry instead of a
// memory latency, bit for bit the same sum.  The next 64 entries are fetched while the adds of the current ones run.
__device__ __forceinline__ double readlane_f64(double v, int t)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), t);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), t);
    return __hiloint2double(hi, lo);
}
template <bool VF>
__device__ __forceinline__ void spmv_exact_wave(int bid, int nblk, int32_t r0, int32_t r1, const int64_t *__restrict__ in_ptr,
                                                const int32_t *__restrict__ in_src, const double *__restrict__ in_w,
                                                const int32_t *__restrict__ row_order, const double *__restrict__ x,
                                                double *__restrict__ y, int32_t my_seed, double c1,
                                                const uint32_t *__restrict__ act, uint32_t *__restrict__ nz_out,
                                                const double *__restrict__ w_src, double *__restrict__ zout)
{
    // the 64 products of a round are parked in LDS (double-buffered per wave); every lane then reads them back one by
    // one from the SAME address (a broadcast read, no bank conflict) -- one LDS read + one add per entry, and the reads
    // of a round are independent of its adds, so they run ahead of them
    __shared__ double prod_s[4][2][WAVE];
    const int lane = threadIdx.x & (WAVE - 1);
    double(*pb)[WAVE] = prod_s[threadIdx.x / WAVE];
    const int64_t nwaves = ((int64_t)nblk * blockDim.x) / WAVE;
    int buf = 0;
    int64_t r = (int64_t)r0 + ((int64_t)bid * blockDim.x + threadIdx.x) / WAVE;
    int32_t j = -1;
    int64_t p = 0, e = 0;
    if (r < r1) { j = row_order[r]; p = in_ptr[j]; e = in_ptr[j + 1]; }
    for (; r < r1; r += nwaves) {
        // the NEXT row's index and list bounds are fetched while this row is being summed
        const int64_t rn = r + nwaves;
        int32_t jn = -1;
        int64_t pn_ = 0, en_ = 0;
        if (rn < r1) { jn = row_order[rn]; pn_ = in_ptr[jn]; en_ = in_ptr[jn + 1]; }
        if (act && !((act[(uint32_t)j >> 5] >> (j & 31)) & 1u)) e = p;
        double acc = 0.0;
        double cur = 0.0;
        if (p + lane < e) {
            if (VF) cur = x[in_src[p + lane]];
            else { const double rw = c1 * x[in_src[p + lane]]; cur = rw * in_w[p + lane]; }
        }
        while (p < e) {
            const int64_t pn = p + WAVE;
            double nxt = 0.0;
            if (pn + lane < e) {                       // issued ahead of the dependent adds below
                if (VF) nxt = x[in_src[pn + lane]];
                else { const double rw = c1 * x[in_src[pn + lane]]; nxt = rw * in_w[pn + lane]; }
            }
            pb[buf][lane] = cur;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (e - p >= WAVE) {
#pragma unroll
                for (int t = 0; t < WAVE; ++t) acc += pb[buf][t];
            } else {
                const int cnt = (int)(e - p);
                for (int t = 0; t < cnt; ++t) acc += pb[buf][t];
            }
            buf ^= 1;                                  // (the other buffer: no wait for this round's reads before the next write)
            cur = nxt;
            p = pn;
        }
        if (lane == 0) RWR_SPMV_STORE(j, acc)
        j = jn; p = pn_; e = en_;
    }
}

// The same idea for shorter rows: W (16 or 4) lanes share a row, 64 / W rows per wave.  The W products of a round are
// formed in parallel; every lane of the group then adds them in list order (the group's lanes all carry the row's
// accumulator), taking product t from lane t of its group.
template <int W, bool VF>
__device__ __forceinline__ void spmv_exact_group(int bid, int nblk, int32_t r0, int32_t r1, const int64_t *__restrict__ in_ptr,
                                                 const int32_t *__restrict__ in_src, const double *__restrict__ in_w,
                                                 const int32_t *__restrict__ row_order, const double *__restrict__ x,
                                                 double *__restrict__ y, int32_t my_seed, double c1,
                                                 const uint32_t *__restrict__ act, uint32_t *__restrict__ nz_out,
                                                 const double *__restrict__ w_src, double *__restrict__ zout)
{
    constexpr int RPW = WAVE / W;
    const int lane = threadIdx.x & (WAVE - 1), gl = lane % W, grp = lane / W;
    const int64_t nwaves = ((int64_t)nblk * blockDim.x) / WAVE;
    for (int64_t rb = (int64_t)r0 + (((int64_t)bid * blockDim.x + threadIdx.x) / WAVE) * RPW; rb < r1; rb += nwaves * RPW) {
        const int64_t r = rb + grp;
        int32_t j = -1;
        int64_t p = 0, e = 0;
        if (r < r1) {
            j = row_order[r];
            p = in_ptr[j];
            e = in_ptr[j + 1];
            if (act && !((act[(uint32_t)j >> 5] >> (j & 31)) & 1u)) e = p;
        }
        double acc = 0.0;
        double cur = 0.0;
        if (p + gl < e) {
            if (VF) cur = x[in_src[p + gl]];
            else { const double rw = c1 * x[in_src[p + gl]]; cur = rw * in_w[p + gl]; }
        }
        while (__any(p < e)) {
            const int64_t pn = p + W;
            double nxt = 0.0;
            if (pn + gl < e) {
                if (VF) nxt = x[in_src[pn + gl]];
                else { const double rw = c1 * x[in_src[pn + gl]]; nxt = rw * in_w[pn + gl]; }
            }
            const int64_t left = e - p;
            const int cnt = left > W ? W : (left > 0 ? (int)left : 0);
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const double v = __shfl(cur, t, W);
                if (t < cnt) acc += v;
            }
            cur = nxt;
            if (p < e) p = pn;
        }
        if (gl == 0 && j >= 0) RWR_SPMV_STORE(j, acc)
    }
}

// rows [ra, b0): a wave per row; [b0, b1): 16 lanes per row; [b1, b2): 4 lanes per row; [b2, n): a lane per row.
// Workgroups [0, nb0) serve the first bin, the next nb1 the second, ...
template <bool VF>
__global__ __launch_bounds__(256) void k_spmv_exact_binned(int nb0, int nb1, int nb2, int nb3, int32_t ra, int32_t b0, int32_t b1,
                                                           int32_t b2, int32_t n, const int64_t *__restrict__ in_ptr,
                                                           const int32_t *__restrict__ in_src,
                                                           const double *__restrict__ in_w,
                                                           const int32_t *__restrict__ row_order,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int32_t *__restrict__ seeds, double c1, int skip_seed_row,
                                                           const uint32_t *__restrict__ act, uint32_t *__restrict__ nz_out,
                                                           const double *__restrict__ w_src, double *__restrict__ zout)
{
    const int32_t my_seed = skip_seed_row ? seeds[0] : -1;
    int b = blockIdx.x;
    if (b < nb0) { spmv_exact_wave<VF>(b, nb0, ra, b0, in_ptr, in_src, in_w, row_order, x, y, my_seed, c1, act, nz_out, w_src, zout); return; }
    b -= nb0;
    if (b < nb1) { spmv_exact_group<16, VF>(b, nb1, b0, b1, in_ptr, in_src, in_w, row_order, x, y, my_seed, c1, act, nz_out, w_src, zout); return; }
    b -= nb1;
    if (b < nb2) { spmv_exact_group<4, VF>(b, nb2, b1, b2, in_ptr, in_src, in_w, row_order, x, y, my_seed, c1, act, nz_out, w_src, zout); return; }
    b -= nb2;
    spmv_exact_lane<VF>(b, nb3, b2, n, in_ptr, in_src, in_w, row_order, x, y, my_seed, c1, act, nz_out, w_src, zout);
}

// K = 1, EXACT mode, HUB rows (in-degree >= hub_t, 2048 by default: the most liked items of a dense graph, the ego of an ego network).
// A strictly sequential fp64 sum of L addends takes L dependent adds -- 29 000 of them for the top item of the
// MovieLens-shaped configuration, ~200 us at the ~16 cycles per entry of the wave body above: the whole SpMV waited for
// that one row.  When every addend is known to be >= 0 (weights, ranks, 1-d: `hub_scan`), the chain is an INTEGER sum
// inside each binade of the running sum (pf.h; chain_scan.hip explains the arithmetic), so it reduces in parallel and is
// still bit for bit the sequential result: the first few hundred entries are added one by one as above (the sum crosses a binade
// every few entries while it is small), the rest in passes of 1024 -- lane l fetches ITS 16 consecutive entries (64
// contiguous bytes of indices), gathers their values, and wave_fold_exact composes the 64 runs.  One wave per row; the
// kernel runs beside k_spmv_exact_binned on a stream of its own.
constexpr int HS_R = 8;    // (8, not 16: at 16 the kernel needs 244 registers per lane and cannot share a SIMD with the sweep kernel's waves)
constexpr int HS_PASS = WAVE * HS_R;
template <bool VF>
__global__ __launch_bounds__(WAVE, 3) void k_spmv_exact_hub(int32_t r0, int32_t r1, const int64_t *__restrict__ in_ptr,
                                                         const int32_t *__restrict__ in_src,
                                                         const double *__restrict__ in_w,
                                                         const int32_t *__restrict__ row_order,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         const int32_t *__restrict__ seeds, double c1, int skip_seed_row,
                                                         const double *__restrict__ w_src, double *__restrict__ zout, int prefix,
                                                         const uint32_t *__restrict__ act, uint32_t *__restrict__ nz_out)
{
    __shared__ double pb[2][WAVE];
    const int lane = threadIdx.x;
    const int32_t my_seed = skip_seed_row ? seeds[0] : -1;
    // a hub row is one long dependent chain: beside the sweep kernel (sweep.hip), whose waves are many and mostly waiting at
    // barriers, this wave should win the SIMD's issue arbitration
    __builtin_amdgcn_s_setprio(3);
    for (int32_t r = r0 + blockIdx.x; r < r1; r += gridDim.x) {
        const int32_t j = row_order[r];
        if (j == my_seed) continue;
        int64_t p = in_ptr[j];
        int64_t e = in_ptr[j + 1];
        // first iterations: a row none of whose in-neighbours holds a non-zero (k_mark_active) stays exactly +0.0
        if (act && !((act[(uint32_t)j >> 5] >> (j & 31)) & 1u)) e = p;
        double acc = 0.0;
        {   // the first `prefix` entries one by one (the shape of spmv_exact_wave)
            const int64_t ea = (e - p) < prefix ? e : p + prefix;
            int buf = 0;
            double cur = 0.0;
            if (p + lane < ea) {
                if (VF) cur = x[in_src[p + lane]];
                else { const double rw = c1 * x[in_src[p + lane]]; cur = rw * in_w[p + lane]; }
            }
            while (p < ea) {
                const int64_t pn = p + WAVE;
                double nxt = 0.0;
                if (pn + lane < ea) {
                    if (VF) nxt = x[in_src[pn + lane]];
                    else { const double rw = c1 * x[in_src[pn + lane]]; nxt = rw * in_w[pn + lane]; }
                }
                pb[buf][lane] = cur;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (ea - p >= WAVE) {
#pragma unroll
                    for (int t = 0; t < WAVE; ++t) acc += pb[buf][t];
                } else {
                    const int cnt = (int)(ea - p);
                    for (int t = 0; t < cnt; ++t) acc += pb[buf][t];
                }
                buf ^= 1;
                cur = nxt;
                p = pn;
            }
            p = ea;
        }
        // the rest in passes of HS_PASS: every lane's 16 consecutive entries, then the exact parallel fold.  Three passes are
        // in flight: the indices of pass k+2 and the gathers of pass k+1 are requested before pass k is folded, so the fold
        // (the only sequential part: it needs the running sum) never waits for memory -- the longest row of the
        // MovieLens-shaped graph (29 000 in-links) bounds the whole step, and it was spending two memory round trips per pass.
        auto load_idx = [&](int64_t pp, int32_t (&idx)[HS_R]) {
            const int64_t left = e - pp;
            const int64_t q0 = pp + (int64_t)lane * HS_R;
            if (left >= HS_PASS) {
#pragma unroll
                for (int u = 0; u < HS_R; u += 4) {
                    const v4i_u t = *reinterpret_cast<const v4i_u *>(in_src + q0 + u);
                    idx[u] = t.x; idx[u + 1] = t.y; idx[u + 2] = t.z; idx[u + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int u = 0; u < HS_R; ++u) idx[u] = (lane * HS_R + u < (int)left) ? in_src[q0 + u] : -1;
            }
        };
        auto gather = [&](int64_t pp, const int32_t (&idx)[HS_R], double (&v)[HS_R]) {
            const int64_t q0 = pp + (int64_t)lane * HS_R;
#pragma unroll
            for (int u = 0; u < HS_R; ++u) {
                v[u] = 0.0;
                if (idx[u] >= 0) {
                    if (VF) v[u] = x[idx[u]];
                    else { const double rw = c1 * x[idx[u]]; v[u] = rw * in_w[q0 + u]; }      // Model.cs:84,87
                }
            }
        };
        int32_t ia[HS_R], ib[HS_R];
        double va[HS_R], vb[HS_R];
        if (p < e) {
            load_idx(p, ia);
            gather(p, ia, va);
            if (p + HS_PASS < e) load_idx(p + HS_PASS, ib);
        }
        while (p < e) {
            if (p + HS_PASS < e) gather(p + HS_PASS, ib, vb);
            if (p + 2 * HS_PASS < e) load_idx(p + 2 * HS_PASS, ia);
            acc = wave_fold_exact<HS_R>(acc, va, lane);                                       // list order, bit for bit
            p += HS_PASS;
            if (p >= e) break;
            if (p + HS_PASS < e) gather(p + HS_PASS, ia, va);
            if (p + 2 * HS_PASS < e) load_idx(p + 2 * HS_PASS, ib);
            acc = wave_fold_exact<HS_R>(acc, vb, lane);
            p += HS_PASS;
        }
        if (lane == 0) RWR_SPMV_STORE(j, acc)
    }
}

// EXACT single seed: two phases (ITEM rows, then the others -- engine.h: row_order_x), each binned by in-degree:
// >= 128 in-links a wave per row; >= 32: 16 lanes per row; >= 4: 4 lanes per row; below: a lane per row
// (RWR_GROUP_ROWS = 0 keeps everything under 128 on the lane-per-row form; RWR_ROW_ORDER != 0 or RWR_SPMV_PHASES = 0: one
// pass in row_order).  One launch per phase.
void launch_spmv_exact(rwr_graph *g, const double *X, double *Y, const int32_t *seeds, double c1, int skip,
                       const uint32_t *act, uint32_t *nz_out, hipStream_t s, const double *zin, double *zout, bool hub_scan)
{
    static const bool by_degree = [] { const char *e = RWR_TUNE_ENV("RWR_ROW_ORDER"); return !e || atoi(e) == 0; }();
    static const int group_rows = [] { const char *e = RWR_TUNE_ENV("RWR_GROUP_ROWS"); return e ? atoi(e) : 2; }();
    // two phases pay once the rank vector no longer fits the L2s (measured: -17 % SpMV time on the 6 M-node graph, nothing on
    // the 0.6 M-node one, +10 % on the dense 0.2 M-node one, where the second launch only adds a tail)
    static const int phases_env = [] { const char *e = getenv("RWR_SPMV_PHASES"); return e ? atoi(e) : -1; }();
    const int phases = phases_env >= 0 ? phases_env : (g->n >= spmv_big_n() ? 1 : 0);
    const bool vf = zin != nullptr;
    const double *gs = vf ? zin : X;       // gather source
    // dense step of a value-free graph whose tables are built: the source-block sweep (sweep.hip) for every row below
    // hub_t in-links, the hub rows beside it as below
    const bool sweep = vf && hub_scan && !act && !nz_out && sweep_ready(g) && g->stream3 && s != g->stream3;
    // hub rows (>= hub_t in-links, first in the in-degree order): exact parallel reduction, one wave per row, on a stream of
    // its own beside the binned kernel (dense steps only, and only when every addend is known to be >= 0)
    static const int hub_env = [] { const char *e = getenv("RWR_HUB_SCAN"); return e ? atoi(e) : 1; }();
    // (also in the frontier iterations: a hub row that IS active costs its whole list there too -- left to the binned kernel's
    //  wave-per-row walk, the one active 24 K-link row of the MovieLens-shaped graph made the sparse first step 324 us long)
    const bool hubs = sweep || (hub_scan && hub_env && by_degree && g->stream3 && s != g->stream3);
    bool forked = false;
    auto fork = [&]() {
        if (forked) return;
        (void)hipEventRecord(g->ev_h0, s);
        (void)hipStreamWaitEvent(g->stream3, g->ev_h0, 0);
        forked = true;
    };
    // (graphs of ego-network size: the hub kernel runs a few microseconds; a side stream would cost two cross-stream hand-overs
    //  -- ~35 us per step measured on an 8 K-node ego network -- to overlap it with a 6 us binned kernel: it stays in line)
    const bool hubs_in_line = !sweep && g->n < 100000;
    auto launch_hubs = [&](const int32_t *order, int32_t ra, int32_t nh) {
        if (nh <= 0) return;
        hipStream_t sh = hubs_in_line ? s : g->stream3;
        if (!hubs_in_line) fork();
        const unsigned grid = (unsigned)(nh < 4096 ? nh : 4096);
        static const int prefix = [] { const char *e = RWR_TUNE_ENV("RWR_HUB_PREFIX"); return e ? atoi(e) : 256; }();
        // beside the sweep kernel: ask for 36 KB of (unused) LDS per workgroup, more than a CU has left beside a sweep
        // workgroup (160 - 128 KB), so that the hub rows run on the CUs the sweep leaves free (sweep.hip: RWR_SWEEP_WGS)
        static const int hub_lds_env = [] { const char *e = RWR_TUNE_ENV("RWR_HUB_LDS"); return e ? atoi(e) : 36864; }();
        const size_t hub_lds = (sweep && !g->sw_partial) ? (size_t)hub_lds_env : 0;   // (partial: the sweep holds every CU)
        if (vf)
            hipLaunchKernelGGL(k_spmv_exact_hub<true>, dim3(grid), dim3(WAVE), hub_lds, sh, ra, ra + nh, g->in_ptr.p, g->in_src.p,
                               g->in_w.p, order, gs, Y, seeds, c1, skip, g->w_src.p, zout, prefix, act, nz_out);
        else
            hipLaunchKernelGGL(k_spmv_exact_hub<false>, dim3(grid), dim3(WAVE), hub_lds, sh, ra, ra + nh, g->in_ptr.p, g->in_src.p,
                               g->in_w.p, order, gs, Y, seeds, c1, skip, g->w_src.p, zout, prefix, act, nz_out);
    };
    auto blocks_for = [](int64_t rows, int W) { const int64_t b = (rows * W + 255) / 256; return (int)(b < 0 ? 0 : (b > 16384 ? 16384 : b)); };
    // (sb: the stream of the row-binned kernel -- the main stream, or stream3 when the sweep holds the main stream)
    auto launch = [&](const int32_t *order, int32_t ra0, int32_t rows, const int32_t bins[3], int32_t nh, hipStream_t sb) {
        if (rows <= 0) return;
        if (!hubs) nh = 0;
        launch_hubs(order, ra0, nh);
        const int32_t ra = ra0 + nh;                    // the binned kernel starts behind the hub rows
        rows -= nh;
        int32_t bins_adj[3] = {bins[0] - nh, bins[1] - nh, bins[2] - nh};
        bins = bins_adj;
        const int32_t b0 = ra + (by_degree ? bins[0] : 0);
        const int32_t b1 = (by_degree && group_rows >= 1) ? ra + bins[1] : b0;
        const int32_t b2 = (by_degree && group_rows >= 2) ? ra + bins[2] : b1;
        const int32_t rend = ra + rows;
        const int nb0 = blocks_for(b0 - ra, 64), nb1 = blocks_for(b1 - b0, 16), nb2 = blocks_for(b2 - b1, 4), nb3 = blocks_for(rend - b2, 1);
        if (nb0 + nb1 + nb2 + nb3 <= 0) return;
        if (vf)
            hipLaunchKernelGGL(k_spmv_exact_binned<true>, dim3((unsigned)(nb0 + nb1 + nb2 + nb3)), dim3(256), 0, sb, nb0, nb1, nb2, nb3, ra, b0,
                               b1, b2, rend, g->in_ptr.p, g->in_src.p, g->in_w.p, order, gs, Y, seeds, c1, skip, act, nz_out, g->w_src.p, zout);
        else
            hipLaunchKernelGGL(k_spmv_exact_binned<false>, dim3((unsigned)(nb0 + nb1 + nb2 + nb3)), dim3(256), 0, sb, nb0, nb1, nb2, nb3, ra, b0,
                               b1, b2, rend, g->in_ptr.p, g->in_src.p, g->in_w.p, order, gs, Y, seeds, c1, skip, act, nz_out, g->w_src.p, zout);
    };
    if (sweep) {
#ifdef RWR_EXPERIMENTS
        // timing probes of the experiments build only (results are WRONG with either set): one of the two kernels alone
        static const int x_only = [] { const char *e = getenv("RWR_X_SWEEP_ONLY"); return e ? atoi(e) : 0; }();
        if (x_only != 1 && (g->x_hub[0] + g->x_hub[1] > 0 || g->sw_partial)) fork();
        if (x_only != 2) launch_sweep(g, zin, Y, zout, seeds, skip, c1, s);
        if (x_only != 1) {
            launch_hubs(g->row_order_x.p, 0, g->x_hub[0]);
            if (g->sw_partial) launch(g->row_order_x.p, g->x_rows[0], g->x_rows[1], g->x_bins[1], g->x_hub[1], g->stream3);
            else launch_hubs(g->row_order_x.p, g->x_rows[0], g->x_hub[1]);
        }
#else
        // (the sweep first: its one workgroup per CU needs 128 KB of LDS and 16 wave slots at once, which a CU already full
        //  of hub-row waves could not offer until they retire; the hub kernel then fills what is left beside it)
        if (g->x_hub[0] + g->x_hub[1] > 0 || g->sw_partial) fork();   // (the fork point lies BEFORE the sweep in the main stream)
        launch_sweep(g, zin, Y, zout, seeds, skip, c1, s);
        launch_hubs(g->row_order_x.p, 0, g->x_hub[0]);
        // the sweep serves the ITEM rows only (sweep.hip: more rows than its waves have accumulators for): the other rows by the
        // row-binned kernel beside it
        if (g->sw_partial) launch(g->row_order_x.p, g->x_rows[0], g->x_rows[1], g->x_bins[1], g->x_hub[1], g->stream3);
        else launch_hubs(g->row_order_x.p, g->x_rows[0], g->x_hub[1]);
#endif
    } else if (by_degree && phases) {
        launch(g->row_order_x.p, 0, g->x_rows[0], g->x_bins[0], g->x_hub[0], s);
        launch(g->row_order_x.p, g->x_rows[0], g->x_rows[1], g->x_bins[1], g->x_hub[1], s);
    } else {
        launch(g->row_order.p, 0, g->n, g->bin_end, g->bin_hub, s);
    }
    if (forked) {                                       // the step is complete when both kernels are
        (void)hipEventRecord(g->ev_h1, g->stream3);
        (void)hipStreamWaitEvent(s, g->ev_h1, 0);
    }
}

}  // namespace rwr
