// K = 1, dense steps, value-free graphs: the source-block SWEEP form of the single-seed SpMV
//     y[j] = sum over the in-links of j, in list order, of z[source]          (Model.deliverRanks, Model.cs:78-88)
//
// Why: an 8-byte gather through L1/L2 costs a whole 128-byte line on the L2->L1 path -- 230-250 G gathers/s for a 1.8 MB
// vector, 180-220 G/s for 4.8 MB, 60-90 G/s for 48 MB (profiles/r03_a_ubench_gather_rates.jsonl), which is where the
// row-binned kernels of spmv.hip sit -- while ds_read_b64 at random addresses delivers 2.4 T gathers/s.  So z is staged
// through LDS one SOURCE BLOCK (BN nodes) at a time, and every destination row is summed block by block:
//
//   * a lane owns a destination row for the whole launch and keeps its running sum in a register; the in-list of a row is
//     sorted by source (build.hip: stable transpose), so the entries that fall into block b are a contiguous piece of the
//     list and sweeping the blocks in ascending order adds them in exactly the reference's order -- bit for bit;
//   * the matrix is re-laid out ONCE per graph into the order the sweep consumes it ("source-block-major"): for every
//     (wave, block) the pieces of the wave's 64 rows, padded to the longest of them, as 16-bit block-local indices, four
//     to a 64-bit word, lane-interleaved -- the wave streams its share of the matrix with coalesced 512-byte loads and
//     never chases a row pointer.  Padding entries point at an LDS slot that holds +0.0 (a sum of non-negative addends is
//     never -0.0, so adding +0.0 leaves it bitwise unchanged); rows come in in-degree order, so the 64 rows of a wave have
//     pieces of similar length;
//   * per block a workgroup refills its LDS copy of z (128 KB, 1.7 us measured) between two barriers.
// Rows of >= hub_t in-links stay with k_spmv_exact_hub (a wave per row, exact parallel reduction) on its own stream.
#include "engine.h"

#include <cstdlib>
#include <vector>

namespace rwr {

namespace {

constexpr int SW_GR = 4;          // entries per lane and 64-bit word
constexpr int SW_MAXK = 4;        // slots (64-row groups) per wave

// slot s (64 consecutive rows of the sweep order) -> workgroup s % nwg, wave (s / nwg) % wpg, slot-of-wave s / (nwg * wpg):
// every workgroup receives an even sample of the in-degree order
__host__ __device__ inline int64_t sw_slot_of(int wg, int wave, int i, int nwg, int wpg) { return ((int64_t)i * wpg + wave) * nwg + wg; }

// pass 1: per (slot, block) the longest piece among the slot's 64 rows, in words of SW_GR entries
__global__ __launch_bounds__(256) void k_sw_count(int32_t n_sw, int32_t ns, int B, int BN, const int32_t *__restrict__ order,
                                                  const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                  uint32_t *__restrict__ scnt)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (s >= ns) return;
    const int64_t pos = s * WAVE + lane;
    int64_t p = 0, e = 0;
    if (pos < n_sw) { const int32_t j = order[pos]; p = in_ptr[j]; e = in_ptr[j + 1]; }
    for (int b = 0; b < B; ++b) {
        const int64_t lim = (int64_t)(b + 1) * BN;
        int cnt = 0;
        while (p < e && in_src[p] < lim) { ++p; ++cnt; }
        int mx = cnt;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(mx, off, WAVE); mx = o > mx ? o : mx; }
        if (lane == 0) scnt[s * B + b] = (uint32_t)((mx + SW_GR - 1) / SW_GR);
    }
}

// pass 2: per wave, the running word-row offsets of its (block, slot) chunks and their total
__global__ __launch_bounds__(256) void k_sw_wave_prefix(int nw, int nwg, int wpg, int K, int32_t ns, int B, const uint32_t *__restrict__ scnt,
                                                        uint32_t *__restrict__ woff, uint32_t *__restrict__ tot)
{
    const int gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= nw) return;
    const int wg = gw / wpg, wave = gw % wpg;
    uint32_t run = 0;
    for (int b = 0; b < B; ++b) {
        woff[(size_t)gw * B + b] = run;
        for (int i = 0; i < K; ++i) {
            const int64_t s = sw_slot_of(wg, wave, i, nwg, wpg);
            if (s < ns) run += scnt[s * B + b];
        }
    }
    tot[gw] = run;
}
// exclusive scan of the per-wave totals (one workgroup; nw is a few thousand)
__global__ __launch_bounds__(1024) void k_sw_scan(int nw, const uint32_t *__restrict__ tot, uint32_t *__restrict__ base,
                                                  unsigned long long *__restrict__ total_out)
{
    __shared__ unsigned long long part[1024];
    const int per = (nw + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = (lo + per < nw) ? lo + per : nw;
    unsigned long long sum = 0;
    for (int q = lo; q < hi; ++q) sum += tot[q];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; }
        *total_out = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int q = lo; q < hi; ++q) { base[q] = (uint32_t)run; run += tot[q]; }
}
// pass 3: the per-(wave, block) record the sweep reads: {first word-row of the chunk, piece lengths of the K slots}
__global__ __launch_bounds__(256) void k_sw_meta(int nw, int nwg, int wpg, int K, int32_t ns, int B, const uint32_t *__restrict__ scnt,
                                                 const uint32_t *__restrict__ woff, const uint32_t *__restrict__ base,
                                                 const uint32_t *__restrict__ tot, uint4 *__restrict__ meta)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (int64_t)nw * B) return;
    const int gw = (int)(q / B), b = (int)(q % B);
    const int wg = gw / wpg, wave = gw % wpg;
    uint32_t c[SW_MAXK] = {0, 0, 0, 0};
    for (int i = 0; i < K; ++i) {
        const int64_t s = sw_slot_of(wg, wave, i, nwg, wpg);
        if (s < ns) c[i] = scnt[s * B + b];
    }
    meta[q] = make_uint4(base[gw] + woff[q], c[0] | (c[1] << 16), c[2] | (c[3] << 16), tot[gw]);
}
// per workgroup: the blocks (ascending) in which any of its rows holds an entry -- wgblk[wg * (B + 1)] = their number, the ids
// behind it.  A workgroup visits only those: the refill, both barriers and the block record of every other block are skipped
// (on a bipartite graph the ITEM rows read the users' blocks only: 7 of 37 on the 0.6 M-node graph)
__global__ __launch_bounds__(64) void k_sw_wglist(int nwg, int wpg, int K, int32_t ns, int B, const uint32_t *__restrict__ scnt,
                                                  uint32_t *__restrict__ wgblk)
{
    const int wg = blockIdx.x * blockDim.x + threadIdx.x;
    if (wg >= nwg) return;
    uint32_t *out = wgblk + (size_t)wg * (B + 1);
    uint32_t cnt = 0;
    for (int b = 0; b < B; ++b) {
        bool any = false;
        for (int wave = 0; wave < wpg && !any; ++wave)
            for (int i = 0; i < K; ++i) {
                const int64_t s = sw_slot_of(wg, wave, i, nwg, wpg);
                if (s < ns && scnt[s * B + b]) { any = true; break; }
            }
        if (any) out[1 + cnt++] = (uint32_t)b;
    }
    out[0] = cnt;
}
// pass 4: the entry stream.  One wave per slot; lane l writes the words of ITS row (8 bytes each, 512 contiguous bytes
// per wave and word-row).
__global__ __launch_bounds__(256) void k_sw_fill(int32_t n_sw, int32_t ns, int nwg, int wpg, int B, int BN, const int32_t *__restrict__ order,
                                                 const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                 const uint4 *__restrict__ meta, uint2 *__restrict__ ents)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (s >= ns) return;
    const int wg = (int)(s % nwg);
    const int64_t t = s / nwg;
    const int wave = (int)(t % wpg), i = (int)(t / wpg);
    const int gw = wg * wpg + wave;
    const int64_t pos = s * WAVE + lane;
    int64_t p = 0, e = 0;
    if (pos < n_sw) { const int32_t j = order[pos]; p = in_ptr[j]; e = in_ptr[j + 1]; }
    for (int b = 0; b < B; ++b) {
        const uint4 m = meta[(size_t)gw * B + b];
        const uint32_t cs[SW_MAXK] = {m.y & 0xffffu, m.y >> 16, m.z & 0xffffu, m.z >> 16};
        uint32_t row = m.x;
        for (int q = 0; q < i; ++q) row += cs[q];
        const uint32_t c = cs[i];
        const int64_t lim = (int64_t)(b + 1) * BN;
        const int32_t lo = b * BN;
        for (uint32_t r = 0; r < c; ++r) {
            uint32_t v[SW_GR];
#pragma unroll
            for (int u = 0; u < SW_GR; ++u) {
                v[u] = (uint32_t)BN;                                    // padding: the +0.0 slot
                if (p < e) { const int32_t sidx = in_src[p]; if (sidx < lim) { v[u] = (uint32_t)(sidx - lo); ++p; } }
            }
            ents[((size_t)(row + r)) * WAVE + lane] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
        }
    }
}

// ------------------------------------------------------------------------------------------------ the sweep
// One workgroup per CU for the whole launch.  LDS holds ONE block of z (zs[BN + 2], slot BN = +0.0).
//
// Block change: barrier (every wave is done with the old block) -- every wave copies its share of the new block by LDS-DMA
// (global_load_lds_dwordx4: 1 KB per instruction, no registers, so the copy's run-time loop issues back to back) -- wait --
// barrier.  ~2 us per 128 KB block, not overlapped with the sums: a double-buffered variant with dedicated loader waves was
// slower (its copy of block b + 1 can only start when block b - 1 has been released, so each 64 KB copy still cost a full
// round trip: 2.6 us per block at twice the number of blocks).
//
// A wave's share of the matrix is ONE sequential stream of T word-rows (512 bytes: 4 entries per lane) in the order it is
// consumed -- block 0 {slot 0, slot 1, ...}, block 1 {...}, ... -- read with a ROLLING prefetch: SW_D word-rows are always
// in flight, each register pair being reloaded with word-row t + SW_D as soon as word-row t has been consumed, by buffer
// loads (descriptor + scalar row offset + one lane-offset VGPR).  Three things the first versions taught:
//   * a single wave must keep ~8 KB outstanding to stream at the rate the heaviest slot needs (64 rows of ~2000 in-links):
//     with one word-row in flight that wave alone took 230 us on the MovieLens-shaped graph;
//   * every load must be UNCONDITIONAL and the queue length static -- vector-memory returns are counted in order, so behind
//     a branch the compiler has to drain the whole queue (s_waitcnt vmcnt(0)) wherever an older load is consumed; rows
//     behind the stream's end are simply requested too (the buffer's range check returns zeros) and never consumed;
//   * no per-load 64-bit address registers: the register allocator recycled the destinations of in-flight loads as address
//     temporaries and every loop iteration waited on that false dependency.
// Piece ends are wave-uniform scalars: a group of SW_G word-rows inside one piece takes a straight-line path (16 LDS
// gathers, then 16 dependent adds); a group that holds a piece end is walked word-row by word-row, switching accumulator
// at a slot end and refilling LDS at a block end.
constexpr int SW_D = 16;          // word-rows in flight per wave (8 KB)
constexpr int SW_STREAM_AUX = 0;  // cache policy of the stream's loads (nt = 2, "read once, do not displace z in the L2s", measured: no difference)
constexpr int SW_G = 4;           // word-rows per group

#ifdef RWR_SWEEP_STAMPS   // compile-time option of the experiments build: per-wave cycle counts (tools/sweep_stamps.py)
__device__ unsigned long long *sw_stamp_buf = nullptr;   // [wave][4]: cycles in block changes, cycles in all, T, blocks
#define SW_TIC() const unsigned long long tic__ = __builtin_amdgcn_s_memtime();
#define SW_TOC() t_enter += __builtin_amdgcn_s_memtime() - tic__;
#else
#define SW_TIC()
#define SW_TOC()
#endif

template <int K>
__global__ __launch_bounds__(512) void k_sweep_lds(int32_t n, int32_t n_sw, int B, int BN, const uint4 *__restrict__ meta,
                                                    const uint2 *__restrict__ ents, const double *__restrict__ z,
                                                    const int32_t *__restrict__ order, double *__restrict__ y,
                                                    const int32_t *__restrict__ seeds, int skip_seed_row, double c1,
                                                    const double *__restrict__ w_src, double *__restrict__ zout,
                                                    const uint32_t *__restrict__ wgblk)
{
    extern __shared__ double zs[];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int wpg = blockDim.x / WAVE, nwg = gridDim.x, wg = blockIdx.x;
#ifdef RWR_SWEEP_STAMPS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long t_enter = 0;
#endif
    if (tid == 0) zs[BN] = 0.0;
    // pieces of 1 KB (128 doubles, 16 bytes per lane); lanes behind the vector's end re-read its last pair (the LDS slots
    // they fill are never referenced).  (Bare s_barrier, not __syncthreads(): its fence would be the same vmcnt(0), but the
    // LDS-only fences keep the compiler from adding more.)
    const int64_t last_pair = n >= 2 ? (((int64_t)n - 2) & ~(int64_t)1) : 0;
#define SW_ENTER_BLOCK()                                                                                                  \
    {                                                                                                                     \
        SW_TIC()                                                                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");                                                   \
        __builtin_amdgcn_s_barrier();                                  /* every wave is done with the previous block */   \
        const int64_t lo__ = (int64_t)b * BN;                                                                             \
        const int64_t left__ = (int64_t)n - lo__;                                                                         \
        const int cnt__ = left__ < BN ? (int)left__ : BN;                                                                 \
        for (int p__ = wave; p__ * 128 < cnt__; p__ += wpg) {                                                             \
            int64_t at__ = lo__ + (int64_t)p__ * 128 + lane * 2;                                                          \
            at__ = at__ < last_pair ? at__ : last_pair;                                                                   \
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(z + at__),                  \
                                             (void __attribute__((address_space(3))) *)(zs + (size_t)p__ * 128), 16, 0, 0); \
        }                                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                  \
        __builtin_amdgcn_s_barrier();                                  /* the block has landed */                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");                                                   \
        SW_TOC()                                                                                                          \
    }

    // (wave-uniform on purpose: the block records are fetched by SCALAR loads, which are not counted with the vector queue)
    const int cw = wave, ncw = wpg;
    const int gw = __builtin_amdgcn_readfirstlane(wg * ncw + cw);
    const uint4 *mrow = meta + (size_t)gw * B;
    double acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = 0.0;
    // the blocks this workgroup visits (workgroup-uniform: every wave walks the same list and meets the others at the same
    // barriers); the records of all other blocks hold empty pieces for every wave of the workgroup
    const uint32_t *wl = wgblk + (size_t)wg * (B + 1);
    const int nb = (int)__builtin_amdgcn_readfirstlane(wl[0]);
    ++wl;
    // record of block 0: {first word-row of the wave's stream, ..., word-rows of the whole stream}; then the first visited block's
    const uint4 m0 = mrow[0];
    const uint32_t T = __builtin_amdgcn_readfirstlane(m0.w);
    int kb = 0;                                                        // position in the list of visited blocks
    int b = nb > 0 ? (int)__builtin_amdgcn_readfirstlane(wl[0]) : 0;
    const uint4 mc = mrow[b];
    uint32_t my = __builtin_amdgcn_readfirstlane(mc.y), mz = __builtin_amdgcn_readfirstlane(mc.z);
    int bn = nb > 1 ? (int)__builtin_amdgcn_readfirstlane(wl[1]) : b;  // the next visited block and its record, fetched one block ahead
    uint4 mn = mrow[bn];
    typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint2 *>(ents + (size_t)__builtin_amdgcn_readfirstlane(m0.x) * WAVE), (short)0, (int)(T * (WAVE * 8u)), 0x00020000);
    const int lane8 = lane * 8;
    v2u_t r[SW_D];
#pragma unroll
    for (int u = 0; u < SW_D; ++u) r[u] = __builtin_amdgcn_raw_buffer_load_b64(srs, lane8, (int)(u * (WAVE * 8u)), SW_STREAM_AUX);

    int i = 0;
    const double *zc = zs;
    uint32_t pend = my & 0xffffu;                                      // end (stream position) of the current piece
    double a = 0.0;
    if (nb > 0) SW_ENTER_BLOCK()

#define SW_PIECE_LEN(I) ((((I) < 2 ? my : mz) >> (((I) & 1) * 16)) & 0xffffu)
    // the current piece (block b, slot i) is exhausted: park its sum, move to the next piece
#define SW_ADVANCE()                                                                                          \
    {                                                                                                         \
        _Pragma("unroll") for (int q__ = 0; q__ < K; ++q__) if (i == q__) acc[q__] = a;                       \
        ++i;                                                                                                  \
        if (i == K) {                                                                                         \
            i = 0;                                                                                            \
            ++kb;                                                                                             \
            if (kb < nb) {                                                                                    \
                b = bn;                                                                                       \
                my = __builtin_amdgcn_readfirstlane(mn.y);                                                    \
                mz = __builtin_amdgcn_readfirstlane(mn.z);                                                    \
                bn = (int)__builtin_amdgcn_readfirstlane(wl[kb + 1 < nb ? kb + 1 : nb - 1]);                  \
                mn = mrow[bn];                                                                                \
                SW_ENTER_BLOCK()                                                                              \
            }                                                                                                 \
        }                                                                                                     \
        if (kb < nb) {                                                                                        \
            _Pragma("unroll") for (int q__ = 0; q__ < K; ++q__) if (i == q__) a = acc[q__];                   \
            pend += SW_PIECE_LEN(i);                                                                          \
        }                                                                                                     \
    }
    // Deferred adds: a lone wave is latency-bound -- the heaviest wave of a workgroup runs alone while the others wait at the
    // block's barrier -- so the 16 LDS gathers of a group are issued BEFORE the 16 dependent adds of the previous group,
    // which then run under the gathers' latency (group buffers va / vb alternate; `pending` says the previous group's values
    // still wait to be added).  The order of the adds is untouched.
    double va[SW_G * SW_GR], vb[SW_G * SW_GR];
    bool pending = false;                                              // (wave-uniform)
#define SW_FLUSH(V) { _Pragma("unroll") for (int u__ = 0; u__ < SW_G * SW_GR; ++u__) a += V[u__]; }
#define SW_GROUP(GQ, CUR, PREV)                                                                                           \
    {                                                                                                                     \
        const uint32_t t0 = t + (GQ) * SW_G;                                                                              \
        if (t0 + SW_G <= pend) {                                                                                          \
            /* the whole group lies inside the current piece (which also means: inside the stream) */                     \
            _Pragma("unroll") for (int u = 0; u < SW_G; ++u) {                                                            \
                const v2u_t w = r[(GQ) * SW_G + u];                                                                       \
                CUR[4 * u] = zc[w.x & 0xffffu]; CUR[4 * u + 1] = zc[w.x >> 16];                                           \
                CUR[4 * u + 2] = zc[w.y & 0xffffu]; CUR[4 * u + 3] = zc[w.y >> 16];                                       \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                         /* gathers first, then the previous group's adds */ \
            if (pending) SW_FLUSH(PREV)                                /* list order (Model.cs:85-88); padding adds +0.0 */ \
            pending = true;                                                                                               \
        } else {                                                                                                          \
            if (pending) { SW_FLUSH(PREV) pending = false; }                                                              \
            _Pragma("unroll") for (int u = 0; u < SW_G; ++u) {                                                            \
                const uint32_t tt = t0 + u;                                                                               \
                if (tt < T) {                                                                                             \
                    while (tt == pend && kb < nb) SW_ADVANCE()                                                            \
                    const v2u_t w = r[(GQ) * SW_G + u];                                                                   \
                    const double v0 = zc[w.x & 0xffffu], v1 = zc[w.x >> 16], v2 = zc[w.y & 0xffffu], v3 = zc[w.y >> 16];  \
                    a += v0; a += v1; a += v2; a += v3;                                                                   \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
        _Pragma("unroll") for (int u = 0; u < SW_G; ++u)                                                                  \
            r[(GQ) * SW_G + u] = __builtin_amdgcn_raw_buffer_load_b64(srs, lane8, (int)((t0 + SW_D + u) * (WAVE * 8u)), SW_STREAM_AUX); \
    }
    static_assert((SW_D / SW_G) % 2 == 0, "the group buffers alternate");
    for (uint32_t t = 0; t < T; t += SW_D) {
#pragma unroll
        for (int gq = 0; gq < SW_D / SW_G; gq += 2) {
            // (`pending` always refers to the buffer the PREVIOUS group wrote, which is this group's PREV)
            SW_GROUP(gq, va, vb)
            SW_GROUP(gq + 1, vb, va)
        }
    }
    if (pending) SW_FLUSH(vb)                                          // (the last group of an iteration writes vb)
#undef SW_GROUP
#undef SW_FLUSH
    while (kb < nb) SW_ADVANCE()                                       // the pieces behind the stream's end are all empty
#undef SW_PIECE_LEN
#undef SW_ADVANCE
#undef SW_ENTER_BLOCK

    const int32_t my_seed = skip_seed_row ? seeds[0] : -1;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const int64_t pos = sw_slot_of(wg, cw, q, nwg, ncw) * WAVE + lane;
        if (pos < n_sw) {
            const int32_t j = order[pos];
            if (j != my_seed) {
                y[j] = acc[q];
                if (zout) { const double rw = c1 * acc[q]; zout[j] = rw * w_src[j]; }   // Model.cs:84,87 for the next step
            }
        }
    }
#ifdef RWR_SWEEP_STAMPS
    if (sw_stamp_buf && lane == 0) {
        unsigned long long *o = sw_stamp_buf + (size_t)gw * 4;
        o[0] = t_enter; o[1] = __builtin_amdgcn_s_memtime() - t_start; o[2] = T; o[3] = (unsigned long long)B;
    }
#endif
}

}  // namespace

// Decides whether the graph takes the sweep and builds its tables (once per graph build; graph_derive resets sw_state).
int32_t sweep_prepare(rwr_graph *g)
{
    if (g->sw_state != 0) return RWR_OK;
    g->sw_state = -1;
    static const int enable = [] { const char *e = getenv("RWR_SWEEP"); return e ? atoi(e) : 1; }();
    static const int64_t min_n = [] { const char *e = getenv("RWR_SWEEP_MIN_N"); return e ? atol(e) : 50000l; }();
    static const int bn_env = [] { const char *e = getenv("RWR_SWEEP_BN"); return e ? atoi(e) : 16384; }();
    if (!enable || !g->vf || g->n < min_n || g->nnz <= 0) return RWR_OK;
    int BN = bn_env;                                       // nodes per block: a multiple of 128 (the loaders move 1 KB pieces)
    if (BN < 128) BN = 128;
    if (BN > 16384) BN = 16384;                            // (BN + 2 doubles of LDS)
    BN &= ~127;
    const int wpg = 8;                                     // waves per workgroup (512 threads: up to 256 registers per lane)
    hipDeviceProp_t prop;
    RWR_HIP(hipGetDeviceProperties(&prop, g->device));
    const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int32_t n_hub = g->x_hub[0] + g->x_hub[1];
    // sweep order: the two-phase in-degree order (engine.h: row_order_x -- ITEM rows, then the others, each by in-degree
    // descending) without its hub rows.  The 64 rows of a slot then gather from the same region of the rank vector (an
    // item's in-links come from users, a user's mostly from items): with the one-phase order a slot mixed both kinds, its
    // lanes were idle in half of the blocks and the stream held 2x padding.
    const int32_t a0 = g->x_rows[0] - g->x_hub[0], a1 = g->x_rows[1] - g->x_hub[1];
    const int B = (int)(((int64_t)g->n + BN - 1) / BN);
    if (B > 512 || g->hub_t > SW_GR * 65535) return RWR_OK;             // (piece lengths are kept in 16 bits)
    // workgroups of the sweep (one per CU, each holding its CU's LDS): a share of the CUs is left to the hub-row kernel, whose
    // workgroups ask for more LDS than a CU has left beside a sweep workgroup and therefore land on the other CUs -- the two
    // kernels then never share a SIMD (sharing one slowed both: 93 us beside 88 us standalone became 184 us together)
    static const int wgs_env = [] { const char *e = getenv("RWR_SWEEP_WGS"); return e ? atoi(e) : 0; }();   // (tests: few workgroups force the ITEM-rows-only form on small graphs)
    int nwg = wgs_env > 0 ? wgs_env : (n_hub > 0 ? (ncu * 5) / 8 : ncu);
    if (nwg > ncu) nwg = ncu;
    if (nwg < 1) nwg = 1;
    int32_t n_sw = a0 + a1;
    int partial = 0;
    auto slots_of = [](int64_t rows) { return (int32_t)((rows + WAVE - 1) / WAVE); };
    if ((slots_of(n_sw) + nwg * wpg - 1) / (nwg * wpg) > SW_MAXK) {
        // too many rows for SW_MAXK slots per wave.  The ITEM rows alone may still fit -- and on a bipartite like-graph they
        // are the rows whose sources (the users) span few blocks, so their share of the step costs a handful of refills; the
        // other rows (users gathering from the far larger item range) stay with the row-binned kernel, which runs BESIDE
        // the sweep (spmv.hip).  Measured on the 0.6 M-node graph: 127 us for the binned kernel alone.
        const int need = (slots_of(a0) + wpg * SW_MAXK - 1) / (wpg * SW_MAXK);   // workgroups for at most SW_MAXK slots per wave
        if (a0 <= 0 || a1 <= 0 || need > ncu) return RWR_OK;
        partial = 1;
        n_sw = a0;
        if (nwg < need) nwg = need;
    }
    if (n_sw <= 0) return RWR_OK;
    const int nw = nwg * wpg;
    RWR_TRY(g->sw_order.ensure((size_t)n_sw));
    if (a0 > 0) RWR_HIP(hipMemcpyAsync(g->sw_order.p, g->row_order_x.p + g->x_hub[0], (size_t)a0 * sizeof(int32_t), hipMemcpyDeviceToDevice, g->stream));
    if (a1 > 0 && !partial) RWR_HIP(hipMemcpyAsync(g->sw_order.p + a0, g->row_order_x.p + g->x_rows[0] + g->x_hub[1], (size_t)a1 * sizeof(int32_t), hipMemcpyDeviceToDevice, g->stream));
    const int32_t hub0 = n_hub;
    const int32_t ns = slots_of(n_sw);
    const int K = (ns + nw - 1) / nw;
    if (K > SW_MAXK) return RWR_OK;
    hipStream_t s = g->stream;
    DevBuf<uint32_t> scnt, woff, tot, base;
    DevBuf<unsigned long long> total;
    RWR_TRY(scnt.alloc((size_t)ns * B));
    RWR_TRY(woff.alloc((size_t)nw * B));
    RWR_TRY(tot.alloc(nw));
    RWR_TRY(base.alloc(nw));
    RWR_TRY(total.alloc(1));
    RWR_TRY(g->sw_meta.ensure((size_t)nw * B));
    RWR_TRY(g->sw_wgblk.ensure((size_t)nwg * (B + 1)));
    const int32_t *order = g->sw_order.p;
    hipLaunchKernelGGL(k_sw_count, dim3(cdiv((size_t)ns, 4)), dim3(256), 0, s, n_sw, ns, B, BN, order, g->in_ptr.p, g->in_src.p, scnt.p);
    hipLaunchKernelGGL(k_sw_wave_prefix, dim3(cdiv((size_t)nw, 256)), dim3(256), 0, s, nw, nwg, wpg, K, ns, B, scnt.p, woff.p, tot.p);
    hipLaunchKernelGGL(k_sw_scan, dim3(1), dim3(1024), 0, s, nw, tot.p, base.p, total.p);
    hipLaunchKernelGGL(k_sw_meta, dim3(cdiv((size_t)nw * B, 256)), dim3(256), 0, s, nw, nwg, wpg, K, ns, B, scnt.p, woff.p, base.p,
                       tot.p, g->sw_meta.p);
    hipLaunchKernelGGL(k_sw_wglist, dim3(cdiv((size_t)nwg, 64)), dim3(64), 0, s, nwg, wpg, K, ns, B, scnt.p, g->sw_wgblk.p);
    RWR_HIP(hipGetLastError());
    unsigned long long h_total = 0;
    std::vector<uint32_t> h_flags;
    RWR_HIP(hipMemcpyAsync(&h_total, total.p, sizeof(h_total), hipMemcpyDeviceToHost, s));
    if (partial) {
        h_flags.resize((size_t)nwg * (B + 1));
        RWR_HIP(hipMemcpyAsync(h_flags.data(), g->sw_wgblk.p, h_flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    RWR_HIP(hipStreamSynchronize(s));
    if (h_total >= 0xFFFFFFFFull) return RWR_OK;            // word-row offsets are 32-bit
    if (partial) {
        // worth it only while the ITEM rows read few blocks (each refill costs every workgroup ~2.5 us)
        size_t refills = 0;
        for (int w = 0; w < nwg; ++w) refills += h_flags[(size_t)w * (B + 1)];
        if (refills > (size_t)16 * nwg) return RWR_OK;
    }
    RWR_TRY(g->sw_ent.ensure((size_t)h_total * WAVE + WAVE));
    hipLaunchKernelGGL(k_sw_fill, dim3(cdiv((size_t)ns, 4)), dim3(256), 0, s, n_sw, ns, nwg, wpg, B, BN, order, g->in_ptr.p, g->in_src.p,
                       g->sw_meta.p, g->sw_ent.p);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(s));
    g->sw_B = B; g->sw_BN = BN; g->sw_K = K; g->sw_nwg = nwg; g->sw_wpg = wpg; g->sw_hub0 = hub0;
    g->sw_rows = n_sw; g->sw_partial = partial;
    g->sw_words = (int64_t)h_total * WAVE;
    g->sw_state = 1;
#ifdef RWR_EXPERIMENTS
    fprintf(stderr, "[sweep] n %d hubs %d rows %d%s slots %d K %d blocks %d (BN %d) compute waves %d x %d words %lld (%.1f MB) nnz %lld\n", g->n, hub0,
            n_sw, partial ? " (ITEM rows only)" : "", ns, K, B, BN, nwg, wpg, (long long)g->sw_words, g->sw_words * 8 / 1e6, (long long)g->nnz);
#endif
    return RWR_OK;
}

bool sweep_ready(const rwr_graph *g) { return g->sw_state == 1; }

// the rows of sw_order; the caller runs the hub rows of both phases of row_order_x (k_spmv_exact_hub) beside it
void launch_sweep(rwr_graph *g, const double *zin, double *Y, double *zout, const int32_t *seeds, int skip, double c1, hipStream_t s)
{
    const size_t smem = ((size_t)g->sw_BN + 2) * sizeof(double);
    const int32_t n_sw = g->sw_rows;
    const int32_t *order = g->sw_order.p;
    const uint4 *meta = g->sw_meta.p;
    const uint2 *ents = g->sw_ent.p;
#ifdef RWR_SWEEP_STAMPS
    static unsigned long long *stamp_dev = nullptr;
    static int stamp_launch = 0;
    const char *stamp_path = getenv("RWR_X_SWEEP_STAMPS");
    const size_t stamp_words = (size_t)g->sw_nwg * g->sw_wpg * 4;
    if (stamp_path && !stamp_dev) {
        (void)hipMalloc((void **)&stamp_dev, stamp_words * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sw_stamp_buf), &stamp_dev, sizeof(stamp_dev));
    }
#endif
#define RWR_SWEEP(KK)                                                                                                              \
    {                                                                                                                              \
        (void)hipFuncSetAttribute((const void *)k_sweep_lds<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);           \
        hipLaunchKernelGGL(k_sweep_lds<KK>, dim3(g->sw_nwg), dim3(g->sw_wpg * WAVE), smem, s, g->n, n_sw, g->sw_B,                \
                           g->sw_BN, meta, ents, zin, order, Y, seeds, skip, c1, g->w_src.p, zout, g->sw_wgblk.p);               \
    }
    switch (g->sw_K) {
        case 1: RWR_SWEEP(1) break;
        case 2: RWR_SWEEP(2) break;
        case 3: RWR_SWEEP(3) break;
        default: RWR_SWEEP(4) break;
    }
#undef RWR_SWEEP
#ifdef RWR_SWEEP_STAMPS
    if (stamp_path && ++stamp_launch == 30) {           // one launch in the middle of a run: dump the stamps
        std::vector<unsigned long long> h(stamp_words);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), stamp_dev, stamp_words * 8, hipMemcpyDeviceToHost);
        if (FILE *f = fopen(stamp_path, "w")) {
            for (size_t w = 0; w < stamp_words / 4; ++w) fprintf(f, "%llu %llu %llu %llu\n", h[w * 4], h[w * 4 + 1], h[w * 4 + 2], h[w * 4 + 3]);
            fclose(f);
        }
    }
#endif
}

}  // namespace rwr
