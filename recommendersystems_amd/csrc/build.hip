// Device-side Graph.buildGraph (Graph.cs:51-88) + stable transpose into the in-neighbour CSR.
//
// Reference semantics reproduced exactly:
//   * links with type == UNDEFINED are dropped from the walk          (Graph.cs:59,72)
//   * sumWeights accumulates the explicit weights LEFT TO RIGHT       (Graph.cs:70-76)
//   * each explicit weight is divided by that sum (IEEE fp64 divide)  (Graph.cs:80-81)
//   * a node with no explicit link is dangling (graph[i] == null)     (Graph.cs:53,64,86)
// The transpose is a STABLE sort of the raw link list by target: the raw list is in
// (source asc, list position asc) order, so every in-neighbour list comes out in exactly
// the order in which Model.deliverRanks adds into nextRank[target] (Model.cs:78,85-88).
#include "engine.h"
#include "sort_small.h"

#include <chrono>
#include <mutex>
#include <vector>
#include <cstdlib>
#include <cstring>

namespace rwr {

// Graph.buildGraph for 64 consecutive source rows per wave (their links are one contiguous range of the flat list).
// The reference sums a row's explicit weights LEFT TO RIGHT (Graph.cs:70-76), so the adds of a row stay sequential -- one
// lane per row -- but nothing else does: the wave streams the range through LDS in tiles with coalesced loads, every lane
// walks ITS row's part of the tile from LDS (a row longer than a tile simply carries its running sum into the next one),
// and the per-link outputs (normalised weight, source row, sort key) are written in a second, fully coalesced sweep in
// which a link finds its row by bisecting the wave's 65 list offsets.  (One thread per row walking global memory on its
// own -- the first version -- ran at a tenth of this: 31 of the 54 ms of the 100 M-like graph's build.)
constexpr int RP_CAP = 2048;   // links per tile and wave
constexpr int RP_WPB = 4;      // waves per workgroup
#ifdef RWR_EXPERIMENTS
__device__ unsigned long long rp_dbg[16];
#define RP_STAMP(I) { if (r0 == 0 && lane == 0) rp_dbg[(I)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define RP_STAMP(I)
#endif
// (the body serves 64 rows starting at r0 with the calling wave's LDS tiles; CAP = links per tile)
template <int CAP>
__device__ __forceinline__ void row_prepare_body(
    int64_t r0, int lane, double *sw, uint8_t *st, int64_t *srp, double *ssum,
    int32_t n, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ etype, const double *__restrict__ w, double *__restrict__ w_norm,
    int32_t *__restrict__ esrc, uint32_t *__restrict__ skey, uint32_t *__restrict__ sval,
    uint8_t *__restrict__ dangling, double *__restrict__ w_src, int *__restrict__ flags)
{
    constexpr int RP_CAP = CAP;
    if (r0 >= n) return;
    RP_STAMP(0)
    const int nrows = (n - r0) < WAVE ? (int)(n - r0) : WAVE;
    const int64_t b = rowptr[r0 + (lane < nrows ? lane : nrows)];
    const int64_t e = rowptr[r0 + (lane < nrows ? lane + 1 : nrows)];
    srp[lane] = b;
    if (lane == 0) srp[nrows] = rowptr[r0 + nrows];
    const int64_t L0 = rowptr[r0], L1 = rowptr[r0 + nrows];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    RP_STAMP(1)
    // ---- sweep 1: every row's explicit weights summed in list order (Graph.cs:75)
    double sum = 0.0, first = 0.0;
    int64_t n_explicit = 0;
    bool uni = true, neg = false;
    int64_t p = b;
    if (L1 - L0 > 8 * (int64_t)RP_CAP) {
        // 64 LONG rows (hub items of a dense graph): a tile would hold a piece of one row only and the lanes would take
        // turns; here every lane walks its own row in global memory, eight entries in flight, all 64 rows side by side
        for (; p < e; p += 8) {
            double wv[8];
            uint8_t tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t q = p + u < e ? p + u : e - 1;
                wv[u] = w[q];
                tv[u] = etype[q];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (p + u < e && tv[u] != RWR_EDGE_UNDEFINED) {
                    const double wp = wv[u];
                    if (n_explicit == 0) first = wp;
                    else if (wp != first) uni = false;
                    sum += wp;                           // Graph.cs:75, list order
                    ++n_explicit;
                    if (!(wp >= 0.0)) neg = true;
                }
            }
        }
    }
    for (int64_t T0 = L0; T0 < L1 && L1 - L0 <= 8 * (int64_t)RP_CAP; T0 += RP_CAP) {
        const int64_t T1 = (T0 + RP_CAP < L1) ? T0 + RP_CAP : L1;
        // (eight loads per lane in flight: as a plain copy loop every iteration waited for its own load, and in the
        //  one-launch build of an ego network -- one workgroup, nothing to hide latency behind -- that was 130 of 270 us)
        for (int64_t c0 = T0; c0 < T1; c0 += 8 * WAVE) {
            double tw[8];
            uint8_t te[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t q = c0 + (int64_t)u * WAVE + lane;
                const int64_t qq = q < T1 ? q : T1 - 1;
                tw[u] = w[qq];
                te[u] = etype[qq];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t q = c0 + (int64_t)u * WAVE + lane;
                if (q < T1) { sw[q - T0] = tw[u]; st[q - T0] = te[u]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t pe = e < T1 ? e : T1;
        // (four LDS pairs fetched ahead of the adds: one entry per iteration paid two dependent LDS round trips each, and
        //  the row's adds -- sequential by the reference's definition -- waited for them)
        for (; p + 4 <= pe; p += 4) {
            uint8_t ty[4];
            double wq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { ty[u] = st[p - T0 + u]; wq[u] = sw[p - T0 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ty[u] != RWR_EDGE_UNDEFINED) {
                    const double wp = wq[u];
                    if (n_explicit == 0) first = wp;
                    else if (wp != first) uni = false;
                    sum += wp;                           // Graph.cs:75, list order
                    ++n_explicit;
                    if (!(wp >= 0.0)) neg = true;        // negative or NaN raw weight
                }
            }
        }
        for (; p < pe; ++p) {
            if (st[p - T0] != RWR_EDGE_UNDEFINED) {
                const double wp = sw[p - T0];
                if (n_explicit == 0) first = wp;
                else if (wp != first) uni = false;
                sum += wp;                               // Graph.cs:75, list order
                ++n_explicit;
                if (!(wp >= 0.0)) neg = true;            // negative or NaN raw weight
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // (the tile is overwritten next)
    }
    RP_STAMP(2)
    ssum[lane] = sum;
    if (lane < nrows) {
        dangling[r0 + lane] = (n_explicit == 0) ? 1 : 0;             // Graph.cs:64,86
        w_src[r0 + lane] = (n_explicit > 0) ? first / sum : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- sweep 2: per-link outputs, coalesced; four links per lane in flight (in the one-launch build of an ego network a
    // single workgroup runs this, and one link per lane and iteration paid a memory round trip each: 170 of its 270 us)
    bool bad = false;
    constexpr int SW2_U = 4;
    for (int64_t q0 = L0 + lane; q0 < L1; q0 += (int64_t)WAVE * SW2_U) {
        uint8_t et[SW2_U];
        int32_t tt[SW2_U];
        double ww[SW2_U];
#pragma unroll
        for (int u = 0; u < SW2_U; ++u) {
            const int64_t q = q0 + (int64_t)u * WAVE;
            const int64_t qq = q < L1 ? q : L1 - 1;
            et[u] = etype[qq];
            tt[u] = dst[qq];
            ww[u] = w[qq];
        }
#pragma unroll
        for (int u = 0; u < SW2_U; ++u) {
            const int64_t q = q0 + (int64_t)u * WAVE;
            if (q < L1) {
                int lo = 0, hi = nrows - 1;                  // the row holding link q: largest i with srp[i] <= q
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (srp[mid] <= q) lo = mid; else hi = mid - 1;
                }
                const bool ex = et[u] != RWR_EDGE_UNDEFINED;
                const int32_t t = tt[u];
                if (t < 0 || t >= n) bad = true;
                w_norm[q] = ex ? ww[u] / ssum[lo] : 0.0;     // Graph.cs:81
                esrc[q] = (int32_t)(r0 + lo);
                skey[q] = (ex && t >= 0 && t < n) ? (uint32_t)t : (uint32_t)n;   // UNDEFINED -> sentinel row n
                sval[q] = (uint32_t)q;
            }
        }
    }
    RP_STAMP(3)
    if (!uni) atomicOr(&flags[0], 1);
    if (bad) atomicOr(&flags[1], 1);
    if (lane < nrows && (neg || (n_explicit > 0 && !(sum > 0.0 && sum < __longlong_as_double(0x7ff0000000000000ll))))) atomicOr(&flags[3], 1);
}

__global__ __launch_bounds__(RP_WPB * WAVE) void k_row_prepare(
    int32_t n, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ dst,
    const uint8_t *__restrict__ etype, const double *__restrict__ w, double *__restrict__ w_norm,
    int32_t *__restrict__ esrc, uint32_t *__restrict__ skey, uint32_t *__restrict__ sval,
    uint8_t *__restrict__ dangling, double *__restrict__ w_src, int *__restrict__ flags)
{
    __shared__ double sw_all[RP_WPB][RP_CAP];
    __shared__ uint8_t st_all[RP_WPB][RP_CAP];
    __shared__ int64_t srp_all[RP_WPB][WAVE + 1];
    __shared__ double ssum_all[RP_WPB][WAVE];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
    row_prepare_body<RP_CAP>(((int64_t)blockIdx.x * RP_WPB + wave) * WAVE, lane, sw_all[wave], st_all[wave], srp_all[wave], ssum_all[wave],
                             n, rowptr, dst, etype, w, w_norm, esrc, skey, sval, dangling, w_src, flags);
}

// in_ptr[j] = first sorted position whose key >= j   (keys sorted ascending, sentinel n last)
// (the element-wise bodies below are shared by the one-element-per-thread kernels of the general build and by the one-launch
//  build of ego-network-sized graphs, k_build_small)
__device__ __forceinline__ void in_ptr_body(int64_t p, const uint32_t *__restrict__ skey, int64_t m, int32_t n,
                                            int64_t *__restrict__ in_ptr)
{
    if (p > m) return;
    // boundary between position p-1 and p
    int64_t lo = (p == 0) ? -1 : (int64_t)skey[p - 1];
    int64_t hi = (p == m) ? (int64_t)n : (int64_t)skey[p];
    for (int64_t j = lo + 1; j <= hi; ++j) in_ptr[j] = p;
}
__global__ __launch_bounds__(256) void k_in_ptr(const uint32_t *__restrict__ skey, int64_t m, int32_t n,
                                                int64_t *__restrict__ in_ptr)
{
    in_ptr_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, skey, m, n, in_ptr);
}

// (nnz_dev != nullptr: the number of explicit links is read on the device -- in_ptr[n] -- and `nnz` is only the launch's upper
//  bound: ego-network-sized graphs are built without a host round trip in the middle)
__global__ __launch_bounds__(256) void k_gather_in(const uint32_t *__restrict__ sval, int64_t nnz,
                                                   const int32_t *__restrict__ esrc,
                                                   const double *__restrict__ w_norm,
                                                   int32_t *__restrict__ in_src, double *__restrict__ in_w,
                                                   const int64_t *__restrict__ nnz_dev = nullptr)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nnz_dev) nnz = *nnz_dev;
    if (p >= nnz) return;
    uint32_t e = sval[p];
    in_src[p] = esrc[e];
    if (in_w) in_w[p] = w_norm[e];     // (value-free graphs keep no per-entry weights)
}

// value-free graphs: the weight of entry p is the one weight of its source
__global__ __launch_bounds__(256) void k_fill_in_w(int64_t nnz, const int32_t *__restrict__ in_src,
                                                   const double *__restrict__ w_src, double *__restrict__ in_w)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) in_w[p] = w_src[in_src[p]];
}

__device__ __forceinline__ void order_keys_body(int32_t i, int32_t n, const int64_t *__restrict__ in_ptr, uint32_t *__restrict__ dkey,
                                                uint32_t *__restrict__ dval, int *__restrict__ maxdeg, int order_mode, uint32_t top)
{
    if (i >= n) return;
    uint32_t deg = (uint32_t)(in_ptr[i + 1] - in_ptr[i]);
    // ascending sort of (top - deg) == in-degree descending (top = the link count, an upper bound of every in-degree: the
    // keys then have only as many bits as the link count, and the radix sort as few passes); mode 2: by power-of-two
    // degree class only (stable, so rows keep their natural order inside a class); mode 1: natural order
    dkey[i] = (order_mode == 1) ? 0u : (order_mode == 2) ? (uint32_t)__clz((int)deg) : (top - deg);
    dval[i] = (uint32_t)i;
    atomicMax(maxdeg, (int)deg);
}
__global__ __launch_bounds__(256) void k_order_keys(int32_t n, const int64_t *__restrict__ in_ptr,
                                                    const uint8_t *__restrict__ node_type,
                                                    const int64_t *__restrict__ node_id,
                                                    uint32_t *__restrict__ dkey, uint32_t *__restrict__ dval,
                                                    int *__restrict__ maxdeg, int order_mode, uint32_t top)
{
    order_keys_body(blockIdx.x * blockDim.x + threadIdx.x, n, in_ptr, dkey, dval, maxdeg, order_mode, top);
}

// single-seed SpMV order: phase (ITEM rows first) then in-degree descending
__device__ __forceinline__ void order_keys_phase_body(int32_t i, int32_t n, const int64_t *__restrict__ in_ptr,
                                                      const uint8_t *__restrict__ node_type, uint32_t *__restrict__ dkey,
                                                      uint32_t *__restrict__ dval, uint32_t top, int top_bits)
{
    if (i >= n) return;
    const uint32_t deg = (uint32_t)(in_ptr[i + 1] - in_ptr[i]);
    const uint32_t phase = node_type[i] == RWR_NODE_ITEM ? 0u : 1u;
    dkey[i] = (phase << top_bits) | (deg > top ? 0u : top - deg);   // (top_bits <= 31: the phase bit stays inside the key)
    dval[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_order_keys_phase(int32_t n, const int64_t *__restrict__ in_ptr,
                                                          const uint8_t *__restrict__ node_type,
                                                          uint32_t *__restrict__ dkey, uint32_t *__restrict__ dval,
                                                          uint32_t top, int top_bits)
{
    order_keys_phase_body(blockIdx.x * blockDim.x + threadIdx.x, n, in_ptr, node_type, dkey, dval, top, top_bits);
}

// items by id descending: ascending sort of ~orderable(id) over the ITEM rows only (every 64-bit key value is a
// legitimate id -- INT64_MIN maps to ~0 -- so non-items cannot be parked behind a sentinel key)
// (keys are taken relative to the largest id, so that they have only as many bits as the id RANGE needs)
__global__ void k_item_id_keys(int32_t n_items, const int32_t *__restrict__ item_rows, const int64_t *__restrict__ node_id,
                               uint64_t *__restrict__ ikey, uint32_t *__restrict__ ival, uint64_t top)
{
    int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_items) return;
    const int32_t i = item_rows[q];
    ikey[q] = top - i64_orderable(node_id[i]);
    ival[q] = (uint32_t)i;
}

__global__ void k_item_flag_keys(int32_t n, const uint8_t *__restrict__ node_type, uint32_t *__restrict__ key,
                                 uint32_t *__restrict__ val)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (node_type[i] == RWR_NODE_ITEM) ? 0u : 1u;
    val[i] = (uint32_t)i;
}

__global__ void k_u32_to_i32(const uint32_t *__restrict__ a, int32_t *__restrict__ b, int64_t m)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < m) b[p] = (int32_t)a[p];
}

// incremental rebuild: overwrite type and/or weight of the listed raw links
__global__ __launch_bounds__(256) void k_patch_links(int64_t count, const int64_t *__restrict__ idx,
                                                     const uint8_t *__restrict__ new_type, const double *__restrict__ new_w,
                                                     uint8_t *__restrict__ etype, double *__restrict__ w)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    const int64_t p = idx[q];
    if (new_type) etype[p] = new_type[q];
    if (new_w) w[p] = new_w[q];
}

// Ego-network-sized graphs: the six raw arrays arrive as ONE copy out of a pinned staging buffer and are dealt to their
// device arrays by this kernel (six pageable copies cost ~100 us of a ~0.5 ms build); what the host needs back (link count,
// flags, in_ptr, dangling) is written by k_stage_out straight into pinned host memory.
constexpr int32_t STAGE_MAX_N = 8192;
constexpr int64_t STAGE_MAX_M = 65536;
constexpr size_t STAGE_OUT_OFF = 1u << 20;                                    // inputs below, outputs above
constexpr size_t STAGE_BYTES = STAGE_OUT_OFF + 8 * (size_t)(STAGE_MAX_N + 1) + STAGE_MAX_N + 256;
struct StageLayout { size_t id, rp, w, dst, nt, et, total; };
static StageLayout stage_layout(int32_t n, int64_t m)
{
    StageLayout L;
    L.id = 0;
    L.rp = L.id + 8 * (size_t)n;
    L.w = L.rp + 8 * ((size_t)n + 1);
    L.dst = L.w + 8 * (size_t)m;
    L.nt = L.dst + ((4 * (size_t)m + 7) & ~(size_t)7);
    L.et = L.nt + (((size_t)n + 7) & ~(size_t)7);
    L.total = L.et + (size_t)m;
    return L;
}
__global__ __launch_bounds__(256) void k_stage_in(int32_t n, int64_t m, const uint8_t *__restrict__ st, StageLayout L,
                                                  int64_t *__restrict__ node_id, uint8_t *__restrict__ node_type,
                                                  int64_t *__restrict__ rowptr, int32_t *__restrict__ dst,
                                                  uint8_t *__restrict__ etype, double *__restrict__ w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        node_id[i] = reinterpret_cast<const int64_t *>(st + L.id)[i];
        node_type[i] = st[L.nt + i];
    }
    if (i <= n) rowptr[i] = reinterpret_cast<const int64_t *>(st + L.rp)[i];
    if (i < m) {
        w[i] = reinterpret_cast<const double *>(st + L.w)[i];
        dst[i] = reinterpret_cast<const int32_t *>(st + L.dst)[i];
        etype[i] = st[L.et + i];
    }
}
// out: [0] link count (int64), [8..24) flags, [32 ...) in_ptr (n+1 int64) when `full`, then dangling (n bytes)
__global__ __launch_bounds__(256) void k_stage_out(int32_t n, const int64_t *__restrict__ in_ptr, const int *__restrict__ flags,
                                                   const uint8_t *__restrict__ dangling, uint8_t *__restrict__ out, int full)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        reinterpret_cast<int64_t *>(out)[0] = in_ptr[n];
        for (int q = 0; q < 4; ++q) reinterpret_cast<int *>(out + 8)[q] = flags[q];
    }
    if (full) {
        if (i <= n) reinterpret_cast<int64_t *>(out + 32)[i] = in_ptr[i];
        if (i < n) (out + 32 + 8 * ((size_t)n + 1))[i] = dangling[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ego-network-sized graphs (n <= STAGE_MAX_N, m <= STAGE_MAX_M): the WHOLE device-side build -- unpack the staging copy,
// Graph.buildGraph's row pass, the stable transpose, the processing orders, the read-back into pinned memory -- as ONE launch
// of one 1024-thread workgroup, with workgroup barriers where the general build has 18 kernel boundaries.  The reference
// rebuilds such a graph per fold and methodology (Experiment.cs:69-105); its build was launch-bound: 18 dependent launches
// of a few microseconds of work each cost ~270 us, the work itself ~60.  Same element-wise bodies, same sorts, same order of
// operations per element as the general path (k_row_prepare, k_in_ptr, k_gather_in, k_order_keys*, k_item_*): bit-identical
// arrays (tests/test_gpu_parity.py: test_build_graph_bitwise, the golden fixtures, the hypothesis graphs, incremental rebuilds).
struct BuildSmallArgs {
    int32_t n, n_items, first, do_stage_in;
    int32_t zero_flags, pad0;        // the kernel clears the flag words itself (batch builds: no memset per graph)
    int64_t m;
    const uint8_t *stage; StageLayout L;
    int64_t *node_id; uint8_t *node_type; int64_t *rowptr; int32_t *dst; uint8_t *etype; double *w_raw;
    double *w_norm; int32_t *esrc; uint32_t *skey, *skey2, *sval, *sval2; uint8_t *dangling; double *w_src; int *flags;
    int64_t *in_ptr; int32_t *in_src; double *in_w;
    int32_t *row_order, *row_order_x, *item_rows, *item_order;
    uint64_t *ikey, *ikey2; uint32_t *ival, *ival2;
    int order_mode; uint32_t top; int top_bits; uint32_t ptop; int ptop_bits; int n_bits; uint64_t id_key_top; int id_key_bits;
    uint8_t *pin_out;
    unsigned long long *stamps;      // experiments build: s_memrealtime (100 MHz) at the stage boundaries; nullptr otherwise
};
#ifdef RWR_EXPERIMENTS
#define BS_STAMP(I) { if (a.stamps && tid == 0) a.stamps[(I)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define BS_STAMP(I)
#endif
constexpr int BS_CAP = 1024;                     // links per LDS tile and wave of the row pass
constexpr int BS_ROW_WAVES = 8;                  // waves that take part in the row pass (their tiles share the LDS with the sort's tables)
__device__ __forceinline__ void build_small_body(const BuildSmallArgs &a)
{
    constexpr int NT = SMALL_SORT_THREADS;
    __shared__ SmallSortLds sortlds;
    __shared__ double sw_all[BS_ROW_WAVES][BS_CAP];
    __shared__ uint8_t st_all[BS_ROW_WAVES][BS_CAP];
    __shared__ int64_t srp_all[BS_ROW_WAVES][WAVE + 1];
    __shared__ double ssum_all[BS_ROW_WAVES][WAVE];
    const int tid = threadIdx.x, wave = tid / WAVE, lane = tid & (WAVE - 1);
    const int32_t n = a.n;
    const int64_t m = a.m;
    BS_STAMP(0)
    if (a.zero_flags) {
        if (tid < 4) a.flags[tid] = 0;
        __syncthreads();
    }
    // ---- the staging copy's arrays (first build only; an incremental rebuild patches the resident arrays)
    if (a.do_stage_in) {
        const int64_t work = (m > (int64_t)n + 1) ? m : (int64_t)n + 1;
        for (int64_t i = tid; i < work; i += NT) {
            if (i < n) {
                a.node_id[i] = reinterpret_cast<const int64_t *>(a.stage + a.L.id)[i];
                a.node_type[i] = a.stage[a.L.nt + i];
            }
            if (i <= n) a.rowptr[i] = reinterpret_cast<const int64_t *>(a.stage + a.L.rp)[i];
            if (i < m) {
                a.w_raw[i] = reinterpret_cast<const double *>(a.stage + a.L.w)[i];
                a.dst[i] = reinterpret_cast<const int32_t *>(a.stage + a.L.dst)[i];
                a.etype[i] = a.stage[a.L.et + i];
            }
        }
        __syncthreads();
    }
    BS_STAMP(1)
    // ---- Graph.buildGraph: filter, list-order row sums, divide (Graph.cs:51-88); 64 rows per wave
    if (wave < BS_ROW_WAVES)
        for (int64_t r0 = (int64_t)wave * WAVE; r0 < n; r0 += (int64_t)BS_ROW_WAVES * WAVE)
            row_prepare_body<BS_CAP>(r0, lane, sw_all[wave], st_all[wave], srp_all[wave], ssum_all[wave], n, a.rowptr, a.dst, a.etype,
                                     a.w_raw, a.w_norm, a.esrc, a.skey, a.sval, a.dangling, a.w_src, a.flags);
    __syncthreads();
    BS_STAMP(2)
    // ---- stable sort of the raw links by target (UNDEFINED links carry the sentinel key n and go last)
    const uint32_t *k_sorted = a.skey, *v_sorted = a.sval;
    if (m > 0) {
        sort_small_body<uint32_t>(sortlds, a.skey, a.skey2, a.sval, a.sval2, (uint32_t)m, a.n_bits);
        if (((a.n_bits + 7) / 8) & 1) { k_sorted = a.skey2; v_sorted = a.sval2; }
    }
    __syncthreads();
    BS_STAMP(3)
    for (int64_t p = tid; p <= m; p += NT) in_ptr_body(p, k_sorted, m, n, a.in_ptr);
    __syncthreads();
    const int64_t nnz = a.in_ptr[n];
    for (int64_t p = tid; p < nnz; p += NT) {
        const uint32_t e = v_sorted[p];
        a.in_src[p] = a.esrc[e];
        a.in_w[p] = a.w_norm[e];
    }
    __syncthreads();
    BS_STAMP(4)
    // ---- destination-row processing orders.  Same permutations as the general build's two sorts -- (top - deg) ascending,
    // stable, for any bound `top` of the in-degrees; (phase, top - deg) ascending, stable -- in fewer radix passes: the first
    // keyed on the graph's own largest in-degree (one 8-bit pass for most ego networks instead of two), the second as ONE
    // stable pass on the phase bit over the first one's result (LSD order: least significant key first).
    for (int32_t i = tid; i < n; i += NT) atomicMax(a.flags + 2, (int)(a.in_ptr[i + 1] - a.in_ptr[i]));
    __syncthreads();
    const uint32_t maxdeg = (uint32_t)__hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int32_t i = tid; i < n; i += NT) {
        const uint32_t deg = (uint32_t)(a.in_ptr[i + 1] - a.in_ptr[i]);
        a.skey[i] = (a.order_mode == 1) ? 0u : (a.order_mode == 2) ? (uint32_t)__clz((int)deg) : (maxdeg - deg);
        a.sval[i] = (uint32_t)i;
    }
    __syncthreads();
    {
        int bits = 1;
        while (bits < 32 && (maxdeg >> bits)) ++bits;
        if (a.order_mode != 0) bits = 8;
        sort_small_body<uint32_t>(sortlds, a.skey, a.skey2, a.sval, a.sval2, (uint32_t)n, bits);
        const uint32_t *res = (((bits + 7) / 8) & 1) ? a.sval2 : a.sval;
        for (int32_t i = tid; i < n; i += NT) a.row_order[i] = (int32_t)res[i];
    }
    __syncthreads();
    for (int32_t q = tid; q < n; q += NT) {
        const int32_t i = a.row_order[q];
        a.skey[q] = a.node_type[i] == RWR_NODE_ITEM ? 0u : 1u;
        a.sval[q] = (uint32_t)i;
    }
    __syncthreads();
    sort_small_body<uint32_t>(sortlds, a.skey, a.skey2, a.sval, a.sval2, (uint32_t)n, 1);
    for (int32_t i = tid; i < n; i += NT) a.row_order_x[i] = (int32_t)a.sval2[i];       // (one pass: the alternate buffer)
    __syncthreads();
    BS_STAMP(5)
    if (a.first) {   // (the node arrays never change: an incremental rebuild keeps both item orders)
        for (int32_t i = tid; i < n; i += NT) {
            a.skey[i] = (a.node_type[i] == RWR_NODE_ITEM) ? 0u : 1u;
            a.sval[i] = (uint32_t)i;
        }
        __syncthreads();
        sort_small_body<uint32_t>(sortlds, a.skey, a.skey2, a.sval, a.sval2, (uint32_t)n, 8);
        for (int32_t i = tid; i < a.n_items; i += NT) a.item_rows[i] = (int32_t)a.sval2[i];    // (one pass: the result is in the alternate buffer)
        __syncthreads();
        if (a.n_items > 0) {
            for (int32_t q = tid; q < a.n_items; q += NT) {
                const int32_t i = a.item_rows[q];
                a.ikey[q] = a.id_key_top - i64_orderable(a.node_id[i]);
                a.ival[q] = (uint32_t)i;
            }
            __syncthreads();
            sort_small_body<uint64_t>(sortlds, a.ikey, a.ikey2, a.ival, a.ival2, (uint32_t)a.n_items, a.id_key_bits);
            const uint32_t *res = (((a.id_key_bits + 7) / 8) & 1) ? a.ival2 : a.ival;
            for (int32_t q = tid; q < a.n_items; q += NT) a.item_order[q] = (int32_t)res[q];
        }
        __syncthreads();
    }
    BS_STAMP(6)
    // ---- what the host needs, straight into pinned memory: link count, flags, in_ptr, dangling
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        reinterpret_cast<int64_t *>(a.pin_out)[0] = a.in_ptr[n];
        for (int q = 0; q < 4; ++q) reinterpret_cast<int *>(a.pin_out + 8)[q] = a.flags[q];
    }
    for (int64_t i = tid; i <= n; i += NT) {
        reinterpret_cast<int64_t *>(a.pin_out + 32)[i] = a.in_ptr[i];
        if (i < n) (a.pin_out + 32 + 8 * ((size_t)n + 1))[i] = a.dangling[i];
    }
    BS_STAMP(7)
}
__global__ __launch_bounds__(SMALL_SORT_THREADS) void k_build_small(BuildSmallArgs a) { build_small_body(a); }
// a batch of ego-network-sized graphs (rwr_eval_graphs): one workgroup per graph, ONE launch for all of them
__global__ __launch_bounds__(SMALL_SORT_THREADS) void k_build_small_multi(const BuildSmallArgs *__restrict__ args)
{
    build_small_body(args[blockIdx.x]);
}

static int bit_length(uint64_t v)
{
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

static int32_t graph_derive(rwr_graph *g, bool first);

static double bt_now()
{
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}
static const bool bt_on = [] { const char *e = RWR_TUNE_ENV("RWR_BUILD_TIMING"); return e && atoi(e) != 0; }();
#define BT(label) do { if (bt_on) { const double t__ = bt_now(); fprintf(stderr, "[build] %-28s %8.1f us\n", label, t__ - bt_t); bt_t = t__; } } while (0)

// uploads the RAW lists (they stay resident: the exclusion list reads them, Recommender.cs:20-24, and an incremental
// rebuild re-derives everything else from them), then derives the walk's data
static int32_t graph_build_upload(rwr_graph *g, const int64_t *node_id, const uint8_t *node_type, const int64_t *rowptr,
                                  const int32_t *dst, const uint8_t *etype, const double *w)
{
    const int32_t n = g->n;
    const int64_t m = g->nnz_raw;
    hipStream_t s = g->stream;
    double bt_t = bt_now();
    if (m >= 0xFFFFFFFFll) {
        set_error("rwr_graph_create: %lld links exceed this build's per-device limit of 2^32-2", (long long)m);
        return RWR_E_UNSUPPORTED;
    }
    g->h_rowptr.assign(rowptr, rowptr + n + 1);
    int32_t n_items = 0;
    uint64_t id_lo = ~0ull, id_hi = 0ull;               // range of the ITEM ids in the order-preserving unsigned form
    for (int32_t i = 0; i < n; ++i)
        if (node_type[i] == RWR_NODE_ITEM) {
            ++n_items;
            const uint64_t k = i64_orderable(node_id[i]);
            id_lo = k < id_lo ? k : id_lo;
            id_hi = k > id_hi ? k : id_hi;
        }
    g->n_items = n_items;
    // items by id DESCENDING = ascending (id_hi - key): as many key bits as the id range needs (ids of a loader are
    // small consecutive numbers: two radix passes instead of eight)
    g->id_key_top = n_items > 0 ? id_hi : 0;
    g->id_key_bits = n_items > 0 ? (bit_length(id_hi - id_lo) > 0 ? bit_length(id_hi - id_lo) : 1) : 1;

    RWR_TRY(g->node_id.alloc(n));
    RWR_TRY(g->node_type.alloc(n));
    RWR_TRY(g->rowptr.alloc((size_t)n + 1));
    RWR_TRY(g->dst.alloc(m));
    RWR_TRY(g->etype.alloc(m));
    RWR_TRY(g->w_raw.alloc(m));
    RWR_TRY(g->w_norm_raw.alloc(m));
    RWR_TRY(g->dangling.alloc(n));
    RWR_TRY(g->w_src.alloc(n));
    RWR_TRY(g->in_ptr.alloc((size_t)n + 1));
    RWR_TRY(g->row_order.alloc(n));
    RWR_TRY(g->row_order_x.alloc(n));
    g->h_is_item.assign((size_t)n, 0);
    for (int32_t i = 0; i < n; ++i) g->h_is_item[i] = node_type[i] == RWR_NODE_ITEM;
    RWR_TRY(g->item_order.alloc(n_items));
    RWR_TRY(g->item_rows.alloc(n_items));

    BT("host prep + allocs");
    static const int stage_env = [] { const char *e = RWR_TUNE_ENV("RWR_STAGE"); return e ? atoi(e) : 1; }();
    g->staged = stage_env && n <= STAGE_MAX_N && m <= STAGE_MAX_M;
    if (g->staged) {
        if (!g->sm_stage) RWR_HIP(hipHostMalloc(&g->sm_stage, STAGE_BYTES, hipHostMallocMapped | hipHostMallocPortable));
        const StageLayout L = stage_layout(n, m);
        uint8_t *st = static_cast<uint8_t *>(g->sm_stage);
        memcpy(st + L.id, node_id, 8 * (size_t)n);
        memcpy(st + L.rp, rowptr, 8 * ((size_t)n + 1));
        memcpy(st + L.nt, node_type, (size_t)n);
        if (m > 0) {
            memcpy(st + L.w, w, 8 * (size_t)m);
            memcpy(st + L.dst, dst, 4 * (size_t)m);
            memcpy(st + L.et, etype, (size_t)m);
        }
        if (!g->stage_dev_ext) {        // (a batch copies all its graphs' slots with one transfer: graphs_build_multi)
            RWR_TRY(g->d_stage.ensure(L.total + 8));
            RWR_HIP(hipMemcpyAsync(g->d_stage.p, st, L.total, hipMemcpyHostToDevice, s));
        }
        g->stage_pending = 1;           // (k_build_small unpacks the copy: graph_derive)
    } else {
        RWR_HIP(hipMemcpyAsync(g->node_id.p, node_id, sizeof(int64_t) * n, hipMemcpyHostToDevice, s));
        RWR_HIP(hipMemcpyAsync(g->node_type.p, node_type, (size_t)n, hipMemcpyHostToDevice, s));
        RWR_HIP(hipMemcpyAsync(g->rowptr.p, rowptr, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, s));
        if (m > 0) {
            RWR_HIP(hipMemcpyAsync(g->dst.p, dst, sizeof(int32_t) * m, hipMemcpyHostToDevice, s));
            RWR_HIP(hipMemcpyAsync(g->etype.p, etype, (size_t)m, hipMemcpyHostToDevice, s));
            RWR_HIP(hipMemcpyAsync(g->w_raw.p, w, sizeof(double) * m, hipMemcpyHostToDevice, s));
        }
    }
    BT("H2D enqueue");
    return RWR_OK;
}

int32_t graph_build(rwr_graph *g, const int64_t *node_id, const uint8_t *node_type, const int64_t *rowptr,
                    const int32_t *dst, const uint8_t *etype, const double *w)
{
    RWR_TRY(graph_build_upload(g, node_id, node_type, rowptr, dst, etype, w));
    return graph_derive(g, true);
}

// Incremental rebuild: patch type / weight of some raw links in place and re-derive.  The result is bit for bit
// the state rwr_graph_create would build from the patched lists (same kernels, same order), without the host
// re-sending the unchanged arrays: what the harness's per-fold / per-methodology rebuild (Experiment.cs:69-105)
// reduces to when only link types (FRIENDSHIP -> UNDEFINED, Experiment.cs:84-101) or a few links change.
int32_t graph_update_links(rwr_graph *g, int64_t count, const int64_t *idx, const uint8_t *etype, const double *w)
{
    hipStream_t s = g->stream;
    if (count > 0) {
        for (int64_t q = 0; q < count; ++q)
            if (idx[q] < 0 || idx[q] >= g->nnz_raw) {
                set_error("rwr_graph_update_links: link index %lld (entry %lld) is outside [0, %lld)", (long long)idx[q],
                          (long long)q, (long long)g->nnz_raw);
                return RWR_E_RANGE;
            }
        DevBuf<int64_t> d_idx;
        DevBuf<uint8_t> d_t;
        DevBuf<double> d_w;
        RWR_TRY(d_idx.alloc((size_t)count));
        RWR_HIP(hipMemcpyAsync(d_idx.p, idx, sizeof(int64_t) * count, hipMemcpyHostToDevice, s));
        if (etype) {
            RWR_TRY(d_t.alloc((size_t)count));
            RWR_HIP(hipMemcpyAsync(d_t.p, etype, (size_t)count, hipMemcpyHostToDevice, s));
        }
        if (w) {
            RWR_TRY(d_w.alloc((size_t)count));
            RWR_HIP(hipMemcpyAsync(d_w.p, w, sizeof(double) * count, hipMemcpyHostToDevice, s));
        }
        hipLaunchKernelGGL(k_patch_links, dim3(cdiv((size_t)count, 256)), dim3(256), 0, s, count, d_idx.p,
                           etype ? d_t.p : (const uint8_t *)nullptr, w ? d_w.p : (const double *)nullptr, g->etype.p,
                           g->w_raw.p);
        RWR_HIP(hipGetLastError());
        RWR_HIP(hipStreamSynchronize(s));   // the staging buffers are released on return
    }
    return graph_derive(g, false);
}

// host side of a finished build: bins of the in-degree orders, statistics
static int32_t derive_finish(rwr_graph *g, const int *h_flags, hipEvent_t e0, hipEvent_t e1)
{
    const int32_t n = g->n;
    double bt_t = bt_now();
    g->max_in_deg = h_flags[2];
    g->bin_end[0] = g->bin_end[1] = g->bin_end[2] = g->bin_huge = g->bin_hub = 0;
    static const int hub_t_env = [] { const char *e = getenv("RWR_HUB_T"); return e ? atoi(e) : 2048; }();   // (measured best on the MovieLens-shaped graph: 2048 / prefix 256)
    g->hub_t = hub_t_env < 128 ? 128 : hub_t_env;   // (hub rows are a prefix of the wave-per-row bin: >= 128 in-links)
    for (int ph = 0; ph < 2; ++ph) g->x_rows[ph] = g->x_hub[ph] = g->x_bins[ph][0] = g->x_bins[ph][1] = g->x_bins[ph][2] = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t deg = g->h_in_ptr[i + 1] - g->h_in_ptr[i];
        g->bin_huge += deg >= 2048;
        g->bin_hub += deg >= g->hub_t;
        g->bin_end[0] += deg >= 128;
        g->bin_end[1] += deg >= 32;
        g->bin_end[2] += deg >= 4;
        const int ph = g->h_is_item[i] ? 0 : 1;
        g->x_rows[ph] += 1;
        g->x_hub[ph] += deg >= g->hub_t;
        g->x_bins[ph][0] += deg >= 128;
        g->x_bins[ph][1] += deg >= 32;
        g->x_bins[ph][2] += deg >= 4;
    }
    BT("D2H in_ptr/dangling + bins");
    float ms = 0.f;
    if (e0 && e1) RWR_HIP(hipEventElapsedTime(&ms, e0, e1));      // (a batch build is not timed per graph)
    g->stats.build_ms = ms;
    g->stats.nnz = g->nnz;
    g->stats.uniform = g->uniform;
    g->stats.uniform_path = g->vf;
    return RWR_OK;
}

// ---- the one-launch build of an ego-network-sized graph, in two halves around its launch: a single graph launches
// k_build_small in between (graph_derive), a batch of graphs one k_build_small_multi for all of them (graphs_build_multi)
struct SmallBuild {
    DevBuf<int32_t> esrc;
    DevBuf<uint32_t> skey, skey2, sval, sval2, ival, ival2;
    DevBuf<uint64_t> ikey, ikey2;
    DevBuf<int> flags;
    BuildSmallArgs a{};
};
static bool small_build_ok(const rwr_graph *g)
{
    static const bool by_degree = [] { const char *e = RWR_TUNE_ENV("RWR_ROW_ORDER"); return !e || atoi(e) == 0; }();
    return g->staged && g->sm_stage && g->n <= STAGE_MAX_N && g->nnz_raw <= STAGE_MAX_M && by_degree;
}
static int32_t small_build_begin(rwr_graph *g, bool first, SmallBuild &b)
{
    const int32_t n = g->n;
    const int64_t m = g->nnz_raw;
    // the (key, value) buffers serve the link sort (m entries) AND, afterwards, the row-order / item sorts (n entries)
    const size_t kv = (size_t)(m > (int64_t)n ? m : (int64_t)n);
    RWR_TRY(b.esrc.alloc(m));
    RWR_TRY(b.skey.alloc(kv));
    RWR_TRY(b.skey2.alloc(kv));
    RWR_TRY(b.sval.alloc(kv));
    RWR_TRY(b.sval2.alloc(kv));
    RWR_TRY(b.flags.alloc(4));
    RWR_TRY(b.ikey.alloc(n));
    RWR_TRY(b.ikey2.alloc(n));
    RWR_TRY(b.ival.alloc(n));
    RWR_TRY(b.ival2.alloc(n));
    RWR_TRY(g->in_src.ensure((size_t)m + 64));
    RWR_TRY(g->in_w.ensure((size_t)(m > 0 ? m : 1)));
    BuildSmallArgs &a = b.a;
    a = BuildSmallArgs{};
    a.n = n; a.n_items = g->n_items; a.first = first ? 1 : 0; a.do_stage_in = g->stage_pending ? 1 : 0; a.m = m;
    a.zero_flags = 1;
    a.stage = g->stage_dev_ext ? g->stage_dev_ext : g->d_stage.p; a.L = stage_layout(n, m);
    a.node_id = g->node_id.p; a.node_type = g->node_type.p; a.rowptr = g->rowptr.p; a.dst = g->dst.p; a.etype = g->etype.p;
    a.w_raw = g->w_raw.p; a.w_norm = g->w_norm_raw.p; a.esrc = b.esrc.p; a.skey = b.skey.p; a.skey2 = b.skey2.p; a.sval = b.sval.p;
    a.sval2 = b.sval2.p; a.dangling = g->dangling.p; a.w_src = g->w_src.p; a.flags = b.flags.p; a.in_ptr = g->in_ptr.p;
    a.in_src = g->in_src.p; a.in_w = g->in_w.p; a.row_order = g->row_order.p; a.row_order_x = g->row_order_x.p;
    a.item_rows = g->item_rows.p; a.item_order = g->item_order.p; a.ikey = b.ikey.p; a.ikey2 = b.ikey2.p; a.ival = b.ival.p;
    a.ival2 = b.ival2.p;
    a.order_mode = 0;
    a.top = (uint32_t)m;                                   // (any bound of the in-degrees gives the same order)
    a.top_bits = bit_length((uint64_t)m) > 0 ? bit_length((uint64_t)m) : 1;
    a.ptop_bits = a.top_bits > 31 ? 31 : a.top_bits;
    a.ptop = a.top_bits > 31 ? 0x7FFFFFFFu : a.top;
    a.n_bits = bit_length((uint64_t)n);
    a.id_key_top = g->id_key_top; a.id_key_bits = g->id_key_bits;
    a.pin_out = g->sm_out ? g->sm_out : static_cast<uint8_t *>(g->sm_stage) + STAGE_OUT_OFF;
    a.stamps = nullptr;
    return RWR_OK;
}
// after the stream the build ran on has been synchronised
static int32_t small_build_end(rwr_graph *g, SmallBuild &b, hipEvent_t e0, hipEvent_t e1)
{
    static const int vf_env_s = [] { const char *e = getenv("RWR_VALUE_FREE"); return e ? atoi(e) : 1; }();
    const int32_t n = g->n;
    const uint8_t *pin = b.a.pin_out;
    int h_flags[4];
    g->stage_pending = 0;
    g->h_in_ptr.resize((size_t)n + 1);
    g->h_dangling.resize((size_t)n);
    const int64_t nnz_s = reinterpret_cast<const int64_t *>(pin)[0];
    memcpy(h_flags, pin + 8, sizeof(h_flags));
    memcpy(g->h_in_ptr.data(), pin + 32, sizeof(int64_t) * ((size_t)n + 1));
    memcpy(g->h_dangling.data(), pin + 32 + 8 * ((size_t)n + 1), (size_t)n);
    if (h_flags[1]) {
        set_error("rwr_graph_create: a link targets a node outside [0, %d)", n);
        return RWR_E_RANGE;
    }
    g->nnz = nnz_s;
    g->uniform = h_flags[0] ? 0 : 1;
    g->nonneg = h_flags[3] ? 0 : 1;
    g->vf = (g->uniform && g->nonneg && vf_env_s) ? 1 : 0;   // (in_w stays: small.hip reads it)
    return derive_finish(g, h_flags, e0, e1);
}

// Graph.buildGraph + transpose + processing orders, from the device-resident raw lists
static int32_t graph_derive(rwr_graph *g, bool first)
{
    g->sw_state = 0;   // the sweep form of the single-seed SpMV re-decides (and rebuilds its tables) on first use
    g->sw_meta.release();
    g->sw_order.release();
    g->sw_ent.release();
    g->sw_wgblk.release();
    g->sw_partial = 0;
    g->part_G = 0;   // a row-partitioned run sized for the previous matrix is over: rwr_part_step asks for a new rwr_part_begin

    const int32_t n = g->n;
    const int64_t m = g->nnz_raw;
    const int32_t n_items = g->n_items;
    hipStream_t s = g->stream;
    double bt_t = bt_now();
    hipEvent_t e0 = g->ev_a, e1 = g->ev_b;
    if (small_build_ok(g)) {
        // ego-network-sized graph: the whole device-side build is ONE launch (k_build_small) and one synchronisation
        SmallBuild b;
        RWR_TRY(small_build_begin(g, first, b));
        BT("derive allocs");
#ifdef RWR_EXPERIMENTS
        DevBuf<unsigned long long> stamps_d;
        if (bt_on) { RWR_TRY(stamps_d.alloc(8)); b.a.stamps = stamps_d.p; }
#endif
        RWR_HIP(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_build_small, dim3(1), dim3(SMALL_SORT_THREADS), 0, s, b.a);
        RWR_HIP(hipGetLastError());
        RWR_HIP(hipEventRecord(e1, s));
        BT("enqueue one-launch build");
        RWR_HIP(hipStreamSynchronize(s));                      // the ONE synchronisation of an ego-network-sized build
        BT("sync (whole build)");
#ifdef RWR_EXPERIMENTS
        if (bt_on) {
            unsigned long long hs[8];
            RWR_HIP(hipMemcpy(hs, stamps_d.p, sizeof(hs), hipMemcpyDeviceToHost));
            unsigned long long rp[16];
            (void)hipMemcpyFromSymbol(rp, HIP_SYMBOL(rp_dbg), sizeof(rp));
            fprintf(stderr, "[build] row pass of group 0 (us): setup %.1f sweep1 %.1f sweep2 %.1f; started %.1f after stage 1\n", (rp[1] - rp[0]) / 100.0,
                    (rp[2] - rp[1]) / 100.0, (rp[3] - rp[2]) / 100.0, (rp[0] - hs[1]) / 100.0);
            fprintf(stderr, "[build] k_build_small stages (us): unpack %.1f rows %.1f linksort %.1f in_ptr+gather %.1f orders %.1f items %.1f out %.1f\n",
                    (hs[1] - hs[0]) / 100.0, (hs[2] - hs[1]) / 100.0, (hs[3] - hs[2]) / 100.0, (hs[4] - hs[3]) / 100.0, (hs[5] - hs[4]) / 100.0,
                    (hs[6] - hs[5]) / 100.0, (hs[7] - hs[6]) / 100.0);
        }
#endif
        return small_build_end(g, b, e0, e1);
    }
    DevBuf<int32_t> esrc;
    DevBuf<uint32_t> skey, skey2, sval, sval2;
    DevBuf<uint8_t> temp;
    DevBuf<int> flags;
    // the (key, value) buffers serve the link sort (m entries) AND, afterwards, the row-order / item sorts (n entries):
    // a graph may have fewer links than nodes
    const size_t kv = (size_t)(m > (int64_t)n ? m : (int64_t)n);
    RWR_TRY(esrc.alloc(m));
    RWR_TRY(skey.alloc(kv));
    RWR_TRY(skey2.alloc(kv));
    RWR_TRY(sval.alloc(kv));
    RWR_TRY(sval2.alloc(kv));
    RWR_TRY(flags.alloc(4));
    size_t tbytes = radix_sort_temp_bytes((size_t)m, 1);
    size_t tb2 = radix_sort_temp_bytes((size_t)n, 1);
    RWR_TRY(temp.alloc(tbytes > tb2 ? tbytes : tb2));
    int h_flags[4] = {0, 0, 0, 0};   // [0] some row non-uniform, [1] bad target, [2] max in-degree, [3] a weight or row sum not in (0, inf)
    RWR_HIP(hipMemsetAsync(flags.p, 0, sizeof(h_flags), s));

    BT("derive allocs");
    RWR_HIP(hipEventRecord(e0, s));

    if (g->stage_pending) {   // (staged upload, general build: unpack the copy by its own kernel)
        const StageLayout L = stage_layout(n, m);
        const int64_t work = (m > (int64_t)n + 1) ? m : (int64_t)n + 1;
        hipLaunchKernelGGL(k_stage_in, dim3(cdiv((size_t)work, 256)), dim3(256), 0, s, n, m, g->d_stage.p, L, g->node_id.p,
                           g->node_type.p, g->rowptr.p, g->dst.p, g->etype.p, g->w_raw.p);
        RWR_HIP(hipGetLastError());
        g->stage_pending = 0;
    }

    hipLaunchKernelGGL(k_row_prepare, dim3(cdiv(n, RP_WPB * WAVE)), dim3(RP_WPB * WAVE), 0, s, n, g->rowptr.p, g->dst.p, g->etype.p,
                       g->w_raw.p, g->w_norm_raw.p, esrc.p, skey.p, sval.p, g->dangling.p, g->w_src.p, flags.p);
    RWR_HIP(hipGetLastError());

    // stable sort of the raw links by target (UNDEFINED links carry the sentinel key n and go last)
    bool alt = false;
    RWR_TRY(radix_sort_pairs<uint32_t>(skey.p, skey2.p, sval.p, sval2.p, (size_t)m, 1, bit_length((uint64_t)n),
                                       temp.p, s, &alt));
    uint32_t *k_sorted = alt ? skey2.p : skey.p;
    uint32_t *v_sorted = alt ? sval2.p : sval.p;
    hipLaunchKernelGGL(k_in_ptr, dim3(cdiv((size_t)m + 1, 256)), dim3(256), 0, s, k_sorted, m, n, g->in_ptr.p);
    RWR_HIP(hipGetLastError());
    int64_t nnz = 0;
    const bool staged = g->staged && g->sm_stage;
    uint8_t *pin_out = staged ? static_cast<uint8_t *>(g->sm_stage) + STAGE_OUT_OFF : nullptr;
    static const int vf_env = [] { const char *e = getenv("RWR_VALUE_FREE"); return e ? atoi(e) : 1; }();
    int64_t nnz_bound = 0;          // what the launches below are sized for
    if (staged) {
        // ego-network-sized graph: NO host round trip here.  The number of explicit links (<= m) stays on the device, the
        // in-lists are sized for m, the per-entry weights are always built (such graphs carry MENTION weights, and the
        // one-launch kernel of small.hip reads them anyway), and the checks wait for the one synchronisation at the end:
        // a link with a bad target was keyed to the sentinel row and is harmless until then.  (The mid-way synchronisation
        // cost ~45 us of host round trip and left the second half of the pipeline un-enqueued meanwhile.)
        nnz_bound = m;
        RWR_TRY(g->in_src.ensure((size_t)m + 64));
        RWR_TRY(g->in_w.ensure((size_t)(m > 0 ? m : 1)));
        if (m > 0) {
            hipLaunchKernelGGL(k_gather_in, dim3(cdiv((size_t)m, 256)), dim3(256), 0, s, v_sorted, m, esrc.p, g->w_norm_raw.p,
                               g->in_src.p, g->in_w.p, g->in_ptr.p + n);
            RWR_HIP(hipGetLastError());
        }
        BT("enqueue row pass + link sort + gather");
    } else {
        RWR_HIP(hipMemcpyAsync(&nnz, g->in_ptr.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        RWR_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
        BT("enqueue row pass + link sort");
        RWR_HIP(hipStreamSynchronize(s));
        BT("sync 1 (H2D + row pass + sort)");
        if (h_flags[1]) {
            set_error("rwr_graph_create: a link targets a node outside [0, %d)", n);
            return RWR_E_RANGE;
        }
        g->nnz = nnz;
        g->uniform = h_flags[0] ? 0 : 1;
        g->nonneg = h_flags[3] ? 0 : 1;
        g->vf = (g->uniform && g->nonneg && vf_env) ? 1 : 0;
        nnz_bound = nnz;
        RWR_TRY(g->in_src.ensure((size_t)nnz + 64));   // (padded: wide index loads may run past a row's end)
        if (g->vf) g->in_w.release();
        else RWR_TRY(g->in_w.ensure((size_t)nnz));
        if (nnz > 0) {
            hipLaunchKernelGGL(k_gather_in, dim3(cdiv((size_t)nnz, 256)), dim3(256), 0, s, v_sorted, nnz, esrc.p,
                               g->w_norm_raw.p, g->in_src.p, g->vf ? (double *)nullptr : g->in_w.p);
            RWR_HIP(hipGetLastError());
        }
    }

    // destination-row processing order (in-degree descending) and ITEM rows by id descending
    const char *om = RWR_TUNE_ENV("RWR_ROW_ORDER");
    const int order_mode = om ? atoi(om) : 0;
    DevBuf<uint64_t> ikey, ikey2;
    DevBuf<uint32_t> ival, ival2;
    RWR_TRY(ikey.alloc(n));
    RWR_TRY(ikey2.alloc(n));
    RWR_TRY(ival.alloc(n));
    RWR_TRY(ival2.alloc(n));
    // (every in-degree is <= the number of explicit links: `top - deg` keys need bit_length(nnz) bits, not 32; any bound of
    //  the in-degrees gives the same order)
    const uint32_t top = (uint32_t)nnz_bound;
    const int top_bits = bit_length((uint64_t)nnz_bound) > 0 ? bit_length((uint64_t)nnz_bound) : 1;
    hipLaunchKernelGGL(k_order_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, n, g->in_ptr.p, g->node_type.p,
                       g->node_id.p, skey.p, sval.p, flags.p + 2, order_mode, top);
    RWR_HIP(hipGetLastError());
    RWR_TRY(radix_sort_pairs<uint32_t>(skey.p, skey2.p, sval.p, sval2.p, (size_t)n, 1, order_mode == 0 ? top_bits : 8, temp.p, s, &alt));
    hipLaunchKernelGGL(k_u32_to_i32, dim3(cdiv(n, 256)), dim3(256), 0, s, alt ? sval2.p : sval.p, g->row_order.p,
                       (int64_t)n);
    // (from 2^31 links on the in-degree key is capped at 31 bits -- longer rows all sort first -- so that the phase bit fits)
    const int ptop_bits = top_bits > 31 ? 31 : top_bits;
    const uint32_t ptop = top_bits > 31 ? 0x7FFFFFFFu : top;
    hipLaunchKernelGGL(k_order_keys_phase, dim3(cdiv(n, 256)), dim3(256), 0, s, n, g->in_ptr.p, g->node_type.p, skey.p, sval.p,
                       ptop, ptop_bits);
    RWR_TRY(radix_sort_pairs<uint32_t>(skey.p, skey2.p, sval.p, sval2.p, (size_t)n, 1, ptop_bits + 1, temp.p, s, &alt));
    hipLaunchKernelGGL(k_u32_to_i32, dim3(cdiv(n, 256)), dim3(256), 0, s, alt ? sval2.p : sval.p, g->row_order_x.p,
                       (int64_t)n);
    if (first) {   // (the node arrays never change: an incremental rebuild keeps both item orders)
        // ITEM rows in ascending row order: one stable pass on the flag (item ? 0 : 1)
        hipLaunchKernelGGL(k_item_flag_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, n, g->node_type.p, skey.p, sval.p);
        RWR_TRY(radix_sort_pairs<uint32_t>(skey.p, skey2.p, sval.p, sval2.p, (size_t)n, 1, 8, temp.p, s, &alt));
        if (n_items > 0)
            hipLaunchKernelGGL(k_u32_to_i32, dim3(cdiv(n_items, 256)), dim3(256), 0, s, alt ? sval2.p : sval.p,
                               g->item_rows.p, (int64_t)n_items);
        if (n_items > 0) {
            hipLaunchKernelGGL(k_item_id_keys, dim3(cdiv(n_items, 256)), dim3(256), 0, s, n_items, g->item_rows.p,
                               g->node_id.p, ikey.p, ival.p, g->id_key_top);
            RWR_TRY(radix_sort_pairs<uint64_t>(ikey.p, ikey2.p, ival.p, ival2.p, (size_t)n_items, 1, g->id_key_bits, temp.p, s, &alt));
            hipLaunchKernelGGL(k_u32_to_i32, dim3(cdiv(n_items, 256)), dim3(256), 0, s, alt ? ival2.p : ival.p,
                               g->item_order.p, (int64_t)n_items);
        }
    }
    RWR_HIP(hipGetLastError());
    g->h_in_ptr.resize((size_t)n + 1);
    g->h_dangling.resize((size_t)n);
    if (staged) {
        hipLaunchKernelGGL(k_stage_out, dim3(cdiv((size_t)n + 1, 256)), dim3(256), 0, s, n, g->in_ptr.p, flags.p, g->dangling.p,
                           pin_out, 1);
        RWR_HIP(hipEventRecord(e1, s));
        BT("enqueue gather + orders");
        RWR_HIP(hipStreamSynchronize(s));                   // the ONE synchronisation of an ego-network-sized build
        nnz = reinterpret_cast<const int64_t *>(pin_out)[0];
        memcpy(h_flags, pin_out + 8, sizeof(h_flags));
        memcpy(g->h_in_ptr.data(), pin_out + 32, sizeof(int64_t) * ((size_t)n + 1));
        memcpy(g->h_dangling.data(), pin_out + 32 + 8 * ((size_t)n + 1), (size_t)n);
        BT("sync (whole build)");
        if (h_flags[1]) {
            set_error("rwr_graph_create: a link targets a node outside [0, %d)", n);
            return RWR_E_RANGE;
        }
        g->nnz = nnz;
        g->uniform = h_flags[0] ? 0 : 1;
        g->nonneg = h_flags[3] ? 0 : 1;
        g->vf = (g->uniform && g->nonneg && vf_env) ? 1 : 0;   // (in_w stays: see above)
    } else {
        RWR_HIP(hipMemcpyAsync(h_flags, flags.p, sizeof(h_flags), hipMemcpyDeviceToHost, s));
        RWR_HIP(hipEventRecord(e1, s));
        BT("enqueue gather + orders");
        RWR_HIP(hipStreamSynchronize(s));
        BT("sync 2 (orders)");
        RWR_HIP(hipMemcpy(g->h_in_ptr.data(), g->in_ptr.p, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyDeviceToHost));
        RWR_HIP(hipMemcpy(g->h_dangling.data(), g->dangling.p, (size_t)n, hipMemcpyDeviceToHost));
    }
    return derive_finish(g, h_flags, e0, e1);
}

int32_t ensure_in_w(rwr_graph *g)
{
    if (g->in_w.p && g->in_w.count >= (size_t)(g->nnz > 0 ? g->nnz : 1)) return RWR_OK;
    RWR_TRY(g->in_w.alloc((size_t)g->nnz));
    if (g->nnz > 0) {
        // (uniform rows: w_src[i] = first / sum is bit for bit w[p] / sum of every explicit link p of i, Graph.cs:81)
        hipLaunchKernelGGL(k_fill_in_w, dim3(cdiv((size_t)g->nnz, 256)), dim3(256), 0, g->stream, g->nnz, g->in_src.p,
                           g->w_src.p, g->in_w.p);
        RWR_HIP(hipGetLastError());
    }
    return RWR_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// A batch of ego-network-sized graphs built with ONE launch (rwr_eval_graphs, multi.hip).  The reference builds one such
// graph per fold and methodology (Experiment.cs:69-105) and ten of its threads do so at once (Program.cs:11); handled one
// graph per call that is two synchronisations and a dozen runtime calls per graph, and the threads queue up behind the
// runtime rather than the GPU.  Here every graph of the batch gets a workgroup of k_build_small_multi: one H2D copy of all
// staging slots, one launch, one synchronisation.
//   caller_index[i]: the graph's number in the caller's batch (for error messages).
// Pinned host buffers of one rwr_eval_graphs call, grown on demand: slot 0 the graphs' staging slots and read-back areas,
// the others the small tables the batch kernels take (argument arrays, test sets, results).  Every host<->device copy of a
// batch goes through pinned memory: an "asynchronous" copy from pageable memory is staged by the runtime under a process-wide
// lock and waits for the device.  Allocating pinned memory costs milliseconds, so the sets are recycled through a process-wide
// pool (a call takes one, gives it back at its end); the calling thread reaches its set through a thread-local pointer.
constexpr int MULTI_SLOTS = 6;
struct MultiPins {
    void *pin[MULTI_SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[MULTI_SLOTS] = {0, 0, 0, 0, 0, 0};
};
static std::mutex g_pins_mutex;
static std::vector<MultiPins *> g_pins;          // parked sets (never freed: a handful of MB per concurrent caller)
static thread_local MultiPins *tl_pins = nullptr;
void multi_pins_acquire()
{
    if (tl_pins) return;
    {
        std::lock_guard<std::mutex> lk(g_pins_mutex);
        if (!g_pins.empty()) { tl_pins = g_pins.back(); g_pins.pop_back(); }
    }
    if (!tl_pins) tl_pins = new MultiPins();
}
void multi_pins_release()
{
    if (!tl_pins) return;
    std::lock_guard<std::mutex> lk(g_pins_mutex);
    g_pins.push_back(tl_pins);
    tl_pins = nullptr;
}
int32_t multi_pinned(int slot, size_t bytes, void **out)
{
    if (!tl_pins) { set_error("multi_pinned: no pinned set acquired"); return RWR_E_INVALID; }
    MultiPins &a = *tl_pins;
    if (a.cap[slot] < bytes) {
        if (a.pin[slot]) { (void)hipHostFree(a.pin[slot]); a.pin[slot] = nullptr; a.cap[slot] = 0; }
        const size_t cap = bytes + bytes / 2 + 4096;
        RWR_HIP(hipHostMalloc(&a.pin[slot], cap, hipHostMallocMapped | hipHostMallocPortable));
        a.cap[slot] = cap;
    }
    *out = a.pin[slot];
    return RWR_OK;
}

bool graph_fits_small_build(int32_t n, int64_t m)
{
    static const int stage_env = [] { const char *e = RWR_TUNE_ENV("RWR_STAGE"); return e ? atoi(e) : 1; }();
    static const bool by_degree = [] { const char *e = RWR_TUNE_ENV("RWR_ROW_ORDER"); return !e || atoi(e) == 0; }();
    return stage_env && by_degree && n > 0 && n <= STAGE_MAX_N && m >= 0 && m <= STAGE_MAX_M;
}

int32_t graphs_build_multi(rwr_graph **gs, int32_t count, const rwr_graph_desc *descs, const int32_t *caller_index, hipStream_t s)
{
    if (count <= 0) return RWR_OK;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // arena: [staging slots of all graphs, contiguous][read-back areas]
    std::vector<size_t> in_off((size_t)count + 1, 0), out_off((size_t)count + 1, 0);
    for (int32_t i = 0; i < count; ++i) {
        const int32_t n = descs[i].n_nodes;
        const int64_t m = descs[i].rowptr[n];
        in_off[(size_t)i + 1] = in_off[i] + up(stage_layout(n, m).total + 8);
        out_off[(size_t)i + 1] = out_off[i] + up(32 + 8 * ((size_t)n + 1) + (size_t)n);
    }
    const size_t in_bytes = in_off[count], need = in_bytes + out_off[count];
    void *pin_v = nullptr, *args_v = nullptr;
    RWR_TRY(multi_pinned(0, need, &pin_v));
    RWR_TRY(multi_pinned(1, sizeof(BuildSmallArgs) * (size_t)count, &args_v));
    uint8_t *pin = static_cast<uint8_t *>(pin_v);
    BuildSmallArgs *h_args = static_cast<BuildSmallArgs *>(args_v);
    DevBuf<uint8_t> d_in;
    DevBuf<BuildSmallArgs> d_args;
    RWR_TRY(d_in.alloc(in_bytes));
    RWR_TRY(d_args.alloc((size_t)count));
    std::vector<SmallBuild> builds((size_t)count);
#ifdef RWR_EXPERIMENTS
    static const bool mt_on = [] { const char *e = getenv("RWR_X_MULTI_TIMING"); return e && atoi(e) != 0; }();
    double mt0 = bt_now(), mt1 = 0, mt2 = 0, mt3 = 0, mt4 = 0, mt_up = 0, mt_begin = 0;
#endif
    for (int32_t i = 0; i < count; ++i) {
        rwr_graph *g = gs[i];
        g->sm_stage = pin + in_off[i];
        g->sm_out = pin + in_bytes + out_off[i];
        g->stage_dev_ext = d_in.p + in_off[i];
        const rwr_graph_desc &D = descs[i];
#ifdef RWR_EXPERIMENTS
        const double u0 = bt_now();
#endif
        RWR_TRY(graph_build_upload(g, D.node_id, D.node_type, D.rowptr, D.dst, D.etype, D.w));
        if (!small_build_ok(g)) { set_error("graphs_build_multi: graph %d does not qualify for the one-launch build", caller_index[i]); return RWR_E_INVALID; }
#ifdef RWR_EXPERIMENTS
        const double u1 = bt_now();
        mt_up += u1 - u0;
#endif
        RWR_TRY(small_build_begin(g, true, builds[i]));
#ifdef RWR_EXPERIMENTS
        mt_begin += bt_now() - u1;
#endif
        h_args[i] = builds[i].a;
    }
#ifdef RWR_EXPERIMENTS
    mt1 = bt_now();
#endif
    RWR_HIP(hipMemcpyAsync(d_in.p, pin, in_bytes, hipMemcpyHostToDevice, s));
    RWR_HIP(hipMemcpyAsync(d_args.p, h_args, sizeof(BuildSmallArgs) * (size_t)count, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_build_small_multi, dim3((unsigned)count), dim3(SMALL_SORT_THREADS), 0, s, d_args.p);
    RWR_HIP(hipGetLastError());
#ifdef RWR_EXPERIMENTS
    mt2 = bt_now();
#endif
    RWR_HIP(hipStreamSynchronize(s));
#ifdef RWR_EXPERIMENTS
    mt3 = bt_now();
#endif
    for (int32_t i = 0; i < count; ++i) {
        rwr_graph *g = gs[i];
        const int32_t rc = small_build_end(g, builds[i], nullptr, nullptr);
        g->sm_stage = nullptr;            // (the arena belongs to the thread, not to the handle)
        g->sm_out = nullptr;
        g->stage_dev_ext = nullptr;
        g->staged = 0;                    // an incremental rebuild of this handle, should one follow, takes the general build
        if (rc != RWR_OK) {
            char msg[512];
            snprintf(msg, sizeof(msg), "%s", rwr_last_error());
            set_error("graph %d of the batch: %s", caller_index[i], msg);
            return rc;
        }
    }
#ifdef RWR_EXPERIMENTS
    mt4 = bt_now();
    if (mt_on)
        fprintf(stderr, "[multi build] %d graphs: prep %.0f us (upload %.0f, begin %.0f), enqueue %.0f, wait %.0f, finish %.0f\n", count, mt1 - mt0,
                mt_up, mt_begin, mt2 - mt1, mt3 - mt2, mt4 - mt3);
#endif
    return RWR_OK;
}

}  // namespace rwr
