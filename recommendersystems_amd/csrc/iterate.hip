// RWR power iteration  x <- (1-c) P^T x + c e   (Model.deliverRanks / updateRanks,
// Model.cs:76-108) for a batch of seeds, on gfx950.
//
// Data layout: the rank matrix is kept seed-minor, tile by tile:  X[tile][node][G]
// (G = seeds per tile = lanes per row, a power of two <= 64).  One lane owns one
// (destination row, seed) pair and walks the row's in-neighbour list sequentially, so
//   * every in-neighbour access of a lane group is one contiguous G*8-byte row of X
//     (a full 128-byte line at G = 16): coalesced, no per-element gathers;
//   * the addends arrive in exactly the reference's order (source asc, list position
//     asc), so every row except the seed's own is BITWISE what the C# computes -- no
//     reductions, no atomics (SURVEY.md section 3.3 point 1);
//   * pull form writes every y[j] exactly once: updateRanks (Model.cs:103-108) is a
//     pointer swap.
// The seed's own row receives, besides its in-links, the restart mass of EVERY node
// interleaved in node order (Model.cs:91-93,96-97): an n-term sequential fp64 chain per
// seed, reproduced literally (k_seed_chain, one lane per seed, on a
// second stream beside the SpMM, or the parallel binade scan of chain_scan.hip).
//
// Arithmetic per edge is the reference's:  rw = (1-d)*x_i  (Model.cs:84), then
// nextRank += rw * weight (Model.cs:87) -- two roundings, never an FMA (the file is built
// with -ffp-contract=off).
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>

namespace rwr {

// VF = value-free form (engine.h: rwr_graph::vf): X is then the z matrix -- z[i] = ((1-d) x[i]) * w_src[i], the one
// product Model.cs:87 adds for EVERY link of source i -- no per-entry value is read, a row's sum is the plain list-order
// sum of the gathered z, and the epilogue forms the row's own z for the next step (Zout, may be null on the last one).
template <int G, bool VF>
__global__ __launch_bounds__(256) void k_spmm(int32_t n, const int64_t *__restrict__ in_ptr,
                                              const int32_t *__restrict__ in_src,
                                              const double *__restrict__ in_w,
                                              const int32_t *__restrict__ row_order,
                                              const double *__restrict__ X, double *__restrict__ Y,
                                              const int32_t *__restrict__ seeds, double c1, int skip_seed_row,
                                              const double *__restrict__ w_src, double *__restrict__ Zout)
{
    constexpr int RPW = WAVE / G;   // destination rows per wave
    constexpr int U = 4;            // gathers in flight per lane
    const int tile = blockIdx.y;
    const size_t toff = (size_t)tile * (size_t)n * G;
    X += toff;
    Y += toff;
    if (VF && Zout) Zout += toff;
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = lane / G, k = lane % G;
    const int32_t my_seed = skip_seed_row ? seeds[tile * G + k] : -1;
    const int wpb = blockDim.x / WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t rb = ((int64_t)blockIdx.x * wpb + threadIdx.x / WAVE) * RPW; rb < n; rb += nwaves * RPW) {
        const int64_t r = rb + sub;
        int32_t j = -1;
        int64_t p = 0, e = 0;
        if (r < n) {
            j = row_order[r];
            p = in_ptr[j];
            e = in_ptr[j + 1];
        }
        double acc = 0.0;
        for (; p + U <= e; p += U) {
            int32_t idx[U];
            double wv[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) idx[u] = in_src[p + u];
#pragma unroll
            for (int u = 0; u < U; ++u) wv[u] = VF ? 0.0 : in_w[p + u];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = X[(size_t)idx[u] * G + k];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (VF) {
                    acc += xv[u];                // z of the source: the product of Model.cs:84,87, formed once per node
                } else {
                    double rw = c1 * xv[u];      // Model.cs:84
                    acc += rw * wv[u];           // Model.cs:87
                }
            }
        }
        for (; p < e; ++p) {
            if (VF) {
                acc += X[(size_t)in_src[p] * G + k];
            } else {
                double rw = c1 * X[(size_t)in_src[p] * G + k];
                acc += rw * in_w[p];
            }
        }
        if (j >= 0 && j != my_seed) {
            Y[(size_t)j * G + k] = acc;
            if (VF && Zout) { const double rw = c1 * acc; Zout[(size_t)j * G + k] = rw * w_src[j]; }
        }
    }
}

// Chunked variant for G >= 8: a lane group fetches G consecutive in-neighbour indices (and
// weights) of its row with ONE coalesced load -- lane k takes entry p+k -- and hands entry t
// to the whole group through the LDS crossbar (ds_bpermute), instead of G lanes loading the
// same address per edge.  All G row gathers of a chunk are issued before the first add, so
// each lane keeps up to G 8-byte gathers in flight; the adds then run in list order.
//
// Frontier awareness for the first iterations (x starts as a single non-zero row per seed and stays
// sparse for two or three steps): CHECK consults a per-tile bitmap "row of X has a non-zero" before a
// row gather and skips the gather of all-zero rows -- their addends are (1-d)*0*w = +0.0, which leave
// the non-negative accumulator bitwise unchanged.  WRITE records the non-zero rows of Y for the next step.
// VF: the value-free form (see k_spmm): X is the z matrix, one index load + one bpermute + one gather + ONE add per
// entry, no weight stream, no multiplies; the epilogue writes the row's next z.
template <int G, int CH, bool CHECK, bool WRITE, bool VF>
__global__ __launch_bounds__(256) void k_spmm_chunked(int32_t n, const int64_t *__restrict__ in_ptr,
                                                      const int32_t *__restrict__ in_src,
                                                      const double *__restrict__ in_w,
                                                      const int32_t *__restrict__ row_order,
                                                      const double *__restrict__ X, double *__restrict__ Y,
                                                      const int32_t *__restrict__ seeds, double c1,
                                                      int skip_seed_row, const uint32_t *__restrict__ nz_in,
                                                      uint32_t *__restrict__ nz_out,
                                                      const uint32_t *__restrict__ act,
                                                      const double *__restrict__ w_src, double *__restrict__ Zout)
{
    static_assert(G >= 8 && G <= 64, "chunked SpMM needs 8 <= G <= 64");
    constexpr int RPW = WAVE / G;
    const int tile = blockIdx.y;
    const size_t toff = (size_t)tile * (size_t)n * G;
    X += toff;
    Y += toff;
    if (VF && Zout) Zout += toff;
    const size_t nzw = ((size_t)n + 31) / 32;
    if (CHECK) nz_in += (size_t)tile * nzw;
    if (WRITE) nz_out += (size_t)tile * nzw;
    if (CHECK && act) act += (size_t)tile * nzw;
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = lane / G, k = lane % G;
    const int gbase = (lane - k) << 2;      // byte address of the group's lane 0 for ds_bpermute
    const int32_t my_seed = skip_seed_row ? seeds[tile * G + k] : -1;
    const int wpb = blockDim.x / WAVE;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t rb = ((int64_t)blockIdx.x * wpb + threadIdx.x / WAVE) * RPW; rb < n; rb += nwaves * RPW) {
        const int64_t r = rb + sub;
        int32_t j = -1;
        int64_t p = 0, e = 0;
        if (r < n) {
            j = row_order[r];
            p = in_ptr[j];
            e = in_ptr[j + 1];
            // first iterations: a row none of whose in-neighbours is non-zero (k_mark_active) stays exactly 0
            if (CHECK && act && !((act[(uint32_t)j >> 5] >> (j & 31)) & 1u)) e = p;
        }
        double acc = 0.0;
        // the wave iterates while ANY group still has entries; finished groups idle (cnt = 0)
        while (__any(p < e)) {
            const int64_t left = e - p;
            const int cnt = left > CH ? CH : (left > 0 ? (int)left : 0);
            int32_t my_idx = 0;
            double my_w = 0.0;
            if (k < cnt) {
                my_idx = in_src[p + k];
                if (!VF) my_w = in_w[p + k];
                if (CHECK) {   // one bitmap probe per entry (by the lane that fetched it); dead rows get the sign bit
                    const uint32_t wd = nz_in[(uint32_t)my_idx >> 5];
                    if (!((wd >> (my_idx & 31)) & 1u)) my_idx |= (int32_t)0x80000000;
                }
            }
            const int wlo = __double2loint(my_w), whi = __double2hiint(my_w);
            if (CHECK) {
                // sparse frontier: entries whose source row is all-zero add +0.0 -- skip them wholesale.  `um` is
                // the union over the wave's row groups of the chunk positions that are live in SOME group.
                const unsigned long long lm = __ballot(k < cnt && my_idx >= 0);
                unsigned long long um = lm;
                if (G < 64) {
#pragma unroll
                    for (int sh = G; sh < 64; sh <<= 1) um |= um >> sh;
                    um &= (1ull << (G & 63)) - 1ull;
                }
                if (__popcll(um) <= 4) {
                    while (um) {                                  // few live entries: take them one by one, in order
                        const int t = __builtin_ctzll(um);
                        um &= um - 1;
                        const int idx = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), my_idx);
                        if (VF) {
                            if (t < cnt && idx >= 0) acc += X[(size_t)idx * G + k];
                        } else {
                            const int lo = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), wlo);
                            const int hi = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), whi);
                            if (t < cnt && idx >= 0) {
                                const double rw = c1 * X[(size_t)idx * G + k];   // Model.cs:84
                                acc += rw * __hiloint2double(hi, lo);             // Model.cs:87
                            }
                        }
                    }
                    p += cnt;
                    continue;
                }
            }
            double xv[CH], wv[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int idx = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), my_idx);
                if (!VF) {
                    const int lo = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), wlo);
                    const int hi = __builtin_amdgcn_ds_bpermute(gbase + (t << 2), whi);
                    wv[t] = __hiloint2double(hi, lo);
                }
                xv[t] = (t < cnt && idx >= 0) ? X[(size_t)idx * G + k] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                if (t < cnt) {
                    if (VF) {
                        acc += xv[t];                // the source's z: fl(fl((1-d) x) * w), Model.cs:84,87
                    } else {
                        double rw = c1 * xv[t];      // Model.cs:84
                        acc += rw * wv[t];           // Model.cs:87
                    }
                }
            }
            p += cnt;
        }
        if (j >= 0 && j != my_seed) {
            Y[(size_t)j * G + k] = acc;
            if (VF && Zout) { const double rw = c1 * acc; Zout[(size_t)j * G + k] = rw * w_src[j]; }
        }
        if (WRITE) {
            const unsigned long long nzb = __ballot(acc != 0.0);
            const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << (sub * G));
            if (k == 0 && j >= 0 && (nzb & gmask)) atomicOr(&nz_out[(uint32_t)j >> 5], 1u << (j & 31));
        }
    }
}

// First iterations: destination rows that can become non-zero = out-neighbours (explicit links) of the rows of X
// that hold a non-zero.  One thread per bitmap word of the tile; pushes over the RAW out-links.
__global__ __launch_bounds__(256) void k_mark_active(int32_t n, const uint32_t *__restrict__ nz, uint32_t *__restrict__ act,
                                                     const int64_t *__restrict__ rowptr, const int32_t *__restrict__ dst,
                                                     const uint8_t *__restrict__ etype)
{
    // one WAVE per bitmap word; the lanes stride over the out-links of each non-zero row of that word
    const size_t nzw = ((size_t)n + 31) / 32;
    const int tile = blockIdx.y;
    const int lane = threadIdx.x & (WAVE - 1);
    const size_t wi = (size_t)blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (wi >= nzw) return;
    uint32_t w = nz[(size_t)tile * nzw + wi];
    uint32_t *a = act + (size_t)tile * nzw;
    while (w) {
        const int b = __builtin_ctz(w);
        w &= w - 1;
        const int64_t i = (int64_t)wi * 32 + b;
        const int64_t p1 = rowptr[i + 1];
        for (int64_t p = rowptr[i] + lane; p < p1; p += WAVE)
            if (etype[p] != RWR_EDGE_UNDEFINED) {
                const int32_t t = dst[p];
                atomicOr(&a[(uint32_t)t >> 5], 1u << (t & 31));
            }
    }
}

// EXACT mode: the seed's own row.  Model.cs:78-99 for target == seed: for i ascending,
// first i's links into the seed (list order), then the restart addend
// rank[i] - rw (non-dangling, Model.cs:91-93) or rank[i] (dangling, Model.cs:96-97).
template <int G>
__global__ __launch_bounds__(64) void k_seed_chain(int32_t n, int ntiles, const int64_t *__restrict__ in_ptr,
                                                   const int32_t *__restrict__ in_src,
                                                   const double *__restrict__ in_w,
                                                   const uint8_t *__restrict__ dangling,
                                                   const double *__restrict__ X, double *__restrict__ Y,
                                                   const int32_t *__restrict__ seeds, double c1,
                                                   uint32_t *__restrict__ nz_out)
{
    constexpr int U = 8;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;   // (tile, k)
    if (q >= ntiles * G) return;
    const int tile = q / G, k = q % G;
    const int32_t s = seeds[q];
    if (s < 0) return;
    const double *x = X + (size_t)tile * (size_t)n * G + k;
    int64_t p = in_ptr[s];
    const int64_t e = in_ptr[s + 1];
    int32_t nxt = (p < e) ? in_src[p] : INT_MAX;
    double acc = 0.0;
    int32_t i = 0;
    for (; i + U <= n; i += U) {
        double xv[U];
        uint8_t dg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[(size_t)(i + u) * G];
#pragma unroll
        for (int u = 0; u < U; ++u) dg[u] = dangling[i + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double rw = c1 * xv[u];
            while (nxt == i + u) {
                acc += rw * in_w[p];
                ++p;
                nxt = (p < e) ? in_src[p] : INT_MAX;
            }
            acc += dg[u] ? xv[u] : (xv[u] - rw);
        }
    }
    for (; i < n; ++i) {
        double xi = x[(size_t)i * G];
        double rw = c1 * xi;
        while (nxt == i) {
            acc += rw * in_w[p];
            ++p;
            nxt = (p < e) ? in_src[p] : INT_MAX;
        }
        acc += dangling[i] ? xi : (xi - rw);
    }
    Y[(size_t)tile * (size_t)n * G + (size_t)s * G + k] = acc;
    if (nz_out && acc != 0.0)
        atomicOr(&nz_out[(size_t)tile * (((size_t)n + 31) / 32) + ((uint32_t)s >> 5)], 1u << (s & 31));
}

// Holds the main stream until `expected` chain workgroups have checked in (or ~1 ms has passed: the spin is
// bounded, so a chain kernel that cannot become fully resident only costs the overlap, never a hang).
// Without it the chain kernel, although launched first on a high-priority stream, is dispatched only after the
// SpMM's ~half-million workgroups have all been issued, i.e. it runs after the SpMM instead of beside it.
__global__ void k_gate(const unsigned int *__restrict__ gate, unsigned int expected)
{
    const long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000) break;  // 1 ms
        __builtin_amdgcn_s_sleep(32);
    }
}

// EXACT mode helper: the addends of the links INTO each seed, ((1-d) x_src) * w in list order
// (Model.cs:84,87 for target == seed), computed in parallel ahead of the sequential fold.
// (VF: X is the z matrix, whose entries ARE those addends)
template <int G, bool VF>
__global__ __launch_bounds__(256) void k_seed_terms(int32_t n, const int64_t *__restrict__ in_ptr,
                                                    const int32_t *__restrict__ in_src,
                                                    const double *__restrict__ in_w, const double *__restrict__ X,
                                                    const int32_t *__restrict__ seeds, double c1,
                                                    const int64_t *__restrict__ evoff, double *__restrict__ evterm)
{
    const int slot = blockIdx.y;                 // tile * G + k
    const int32_t s = seeds[slot];
    if (s < 0) return;
    const int tile = slot / G, k = slot % G;
    const double *x = X + (size_t)tile * (size_t)n * G + k;
    const int64_t p0 = in_ptr[s], deg = in_ptr[s + 1] - p0;
    double *out = evterm + evoff[slot];
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < deg; q += (int64_t)gridDim.x * blockDim.x) {
        if (VF) {
            out[q] = x[(size_t)in_src[p0 + q] * G];
        } else {
            const double rw = c1 * x[(size_t)in_src[p0 + q] * G];
            out[q] = rw * in_w[p0 + q];
        }
    }
}

// EXACT mode while X is still sparse (iterations 0 and 1): all-zero rows add +0.0 to the seed row's chain, so the
// chain only has to visit the non-zero rows of X -- in ascending node order, taking the links INTO the seed that
// come from such a row first (Model.cs:85-93).  One wave per tile walks the tile's non-zero-row bitmap; lane = seed.
template <int G>
__global__ __launch_bounds__(64) void k_seed_chain_sparse(int32_t n, const int64_t *__restrict__ in_ptr,
                                                          const int32_t *__restrict__ in_src,
                                                          const uint8_t *__restrict__ dangling,
                                                          const double *__restrict__ X, double *__restrict__ Y,
                                                          const int32_t *__restrict__ seeds, double c1,
                                                          const int64_t *__restrict__ evoff,
                                                          const double *__restrict__ evterm,
                                                          const uint32_t *__restrict__ nz_x, uint32_t *__restrict__ nz_out,
                                                          unsigned int *__restrict__ gate)
{
    if (threadIdx.x == 0 && gate) __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int tile = blockIdx.x, lane = threadIdx.x;
    const size_t nzw = ((size_t)n + 31) / 32;
    const uint32_t *bits = nz_x + (size_t)tile * nzw;
    const double *x = X + (size_t)tile * (size_t)n * G;
    const bool consumer = lane < G;
    const int k = lane % G;
    int32_t s = -1;
    int64_t p = 0, e = 0;
    const double *termp = evterm;
    if (consumer) {
        s = seeds[tile * G + k];
        if (s >= 0) {
            p = in_ptr[s];
            e = in_ptr[s + 1];
            termp = evterm + evoff[tile * G + k] - p;
        }
    }
    double acc = 0.0;
    for (size_t w0 = 0; w0 < nzw; w0 += WAVE) {
        const uint32_t word = (w0 + lane < nzw) ? bits[w0 + lane] : 0u;
        unsigned long long mask = __ballot(word != 0u);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            uint32_t wv = (uint32_t)__builtin_amdgcn_readlane((int)word, l);
            while (wv) {
                const int b = __builtin_ctz(wv);
                wv &= wv - 1;
                const int32_t i = (int32_t)((w0 + l) * 32 + b);
                if (consumer && s >= 0) {
                    while (p < e && in_src[p] < i) ++p;                 // links from all-zero rows: addend +0.0
                    while (p < e && in_src[p] == i) { acc += termp[p]; ++p; }   // links i -> seed first (Model.cs:85-88)
                    const double xi = x[(size_t)i * G + k];
                    const double rw = c1 * xi;
                    acc += dangling[i] ? xi : (xi - rw);                // then the restart addend (Model.cs:91-93,96-97)
                }
            }
        }
    }
    if (consumer && s >= 0) {
        Y[(size_t)tile * (size_t)n * G + (size_t)s * G + k] = acc;
        if (nz_out && acc != 0.0)
            atomicOr(&nz_out[(size_t)tile * nzw + ((uint32_t)s >> 5)], 1u << (s & 31));
    }
}

// EXACT mode, role-specialised form (the default): wave 0 only FOLDS, waves 1-6 only STAGE.
//   stagers: stream the tile's rank matrix in 48 KiB chunks, two chunks ahead of the fold (2 x 16 coalesced loads
//            in flight per lane), turn each value into its restart addend
//                rr_i = dangling_i ? x_i : x_i - (1-d)*x_i          (Model.cs:91,97)
//            and write it to LDS (double-buffered);
//   folder : adds the staged addends strictly in node order, 16 at a time from registers.  The addends of the
//            links INTO a seed (computed by k_seed_terms) are prefetched per lane two links ahead into registers;
//            a 32-row block that holds such a link is walked row by row so that the link's addend lands before
//            the row's restart addend (Model.cs:85-93).  The folder issues no other global loads, so the n-term
//            chain runs near the dependent v_add_f64 rate (~7.5 cycles per row).
constexpr int CH3_NSW = 6;                      // stager waves
constexpr int CH3_NST = CH3_NSW * WAVE;         // stager threads
constexpr int CH3_LPT = 16;                     // loads per stager thread and chunk
constexpr int CH3_CE = CH3_NST * CH3_LPT;       // 6144 doubles per chunk (48 KiB)
template <int G>
__global__ __launch_bounds__(WAVE + CH3_NST) void k_seed_chain_roles(
    int32_t n, const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
    const uint8_t *__restrict__ dangling, const double *__restrict__ X, double *__restrict__ Y,
    const int32_t *__restrict__ seeds, double c1, const int64_t *__restrict__ evoff,
    const double *__restrict__ evterm, uint32_t *__restrict__ nz_out, unsigned int *__restrict__ gate)
{
    // check in: the main stream holds the SpMM back (k_gate) until the chain workgroups own their CUs
    if (threadIdx.x == 0 && gate) __hip_atomic_fetch_add(gate, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int CR = CH3_CE / G;                         // rows per chunk (a multiple of 32)
    static_assert(CR % 32 == 0, "chunk rows must be a multiple of 32");
    extern __shared__ double rr[];                         // [2][CH3_CE]
    const int tile = blockIdx.x;
    const double *x = X + (size_t)tile * (size_t)n * G;
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)n * G;
    const int nchunks = (int)((total + CH3_CE - 1) / CH3_CE);

    if (tid >= WAVE) {
        // ------------------------------------------------------------------ stagers
        const int st = tid - WAVE;
        double rega[CH3_LPT], regb[CH3_LPT];
        uint8_t dga[CH3_LPT], dgb[CH3_LPT];
#define CH3_LOAD(R, D, C)                                                       \
    {                                                                           \
        const int64_t base__ = (int64_t)(C) * CH3_CE;                           \
        _Pragma("unroll") for (int u = 0; u < CH3_LPT; ++u) {                   \
            int64_t el = base__ + (int64_t)u * CH3_NST + st;                    \
            el = el < total ? el : total - 1;                                   \
            R[u] = x[el];                                                       \
            D[u] = dangling[el / G];                                            \
        }                                                                       \
    }
#define CH3_STAGE(R, D, C)                                                                              \
    {                                                                                                   \
        const int64_t base__ = (int64_t)(C) * CH3_CE;                                                   \
        double *buf__ = rr + (size_t)((C) & 1) * CH3_CE;                                                \
        _Pragma("unroll") for (int u = 0; u < CH3_LPT; ++u) {                                           \
            const int off = u * CH3_NST + st;                                                           \
            const double xv = (base__ + off < total) ? R[u] : 0.0; /* past the end: +0.0, no effect */  \
            const double rw = c1 * xv;                                                                  \
            buf__[off] = D[u] ? xv : (xv - rw);                                                         \
        }                                                                                               \
    }
        CH3_LOAD(rega, dga, 0)
        if (nchunks > 1) CH3_LOAD(regb, dgb, 1)
        for (int c = 0; c < nchunks; c += 2) {
            CH3_STAGE(rega, dga, c)
            if (c + 2 < nchunks) CH3_LOAD(rega, dga, c + 2)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            if (c + 1 < nchunks) {
                CH3_STAGE(regb, dgb, c + 1)
                if (c + 3 < nchunks) CH3_LOAD(regb, dgb, c + 3)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                __builtin_amdgcn_s_barrier();
            }
        }
#undef CH3_LOAD
#undef CH3_STAGE
        return;
    }

    // ---------------------------------------------------------------------- folder (wave 0)
    const bool consumer = tid < G;
    const int k = tid % G;
    int32_t s = -1;
    int64_t p = 0, e = 0;                                  // p = index of the link AFTER the pending one
    const int32_t *srcp = in_src;
    const double *termp = evterm;
    int32_t nxt = INT_MAX, raw_s = INT_MAX;                // current link's source row; pending link's (raw load)
    double tcur = 0.0, raw_t = 0.0;
    bool has2 = false;
    if (consumer) {
        s = seeds[tile * G + k];
        if (s >= 0) {
            p = in_ptr[s];
            e = in_ptr[s + 1];
            termp = evterm + evoff[tile * G + k] - p;      // termp[link index] = addend of that link
            if (p < e) { nxt = srcp[p]; tcur = termp[p]; ++p; }
            if (p < e) { raw_s = srcp[p]; raw_t = termp[p]; has2 = true; ++p; }
        }
    }
    double acc = 0.0;
    __builtin_amdgcn_s_setprio(3);
    for (int c = 0; c < nchunks; ++c) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const double *buf = rr + (size_t)(c & 1) * CH3_CE;
        const int64_t row0 = (int64_t)c * CR;
        // 32-row blocks, two register sets in flight: the 32 LDS reads of the NEXT block are issued before the 32
        // dependent adds of the current one, so the adds (7.5 cycles each, the chain's floor) never wait for LDS.
        // A block that holds a link into one of the tile's seeds takes the same adds from the same registers, but
        // before row u every lane whose pending link comes from that row takes it first (Model.cs:85-88).
#define CH3_READ(V, R0) _Pragma("unroll") for (int u = 0; u < 32; ++u) V[u] = buf[((R0) + u) * G + k];
#define CH3_FOLD(V, R0)                                                                                     \
    {                                                                                                       \
        const bool evt = __any(consumer && nxt < row0 + (R0) + 32);                                         \
        if (!evt) {                                                                                         \
            _Pragma("unroll") for (int u = 0; u < 32; ++u) acc += V[u];                                     \
        } else {                                                                                            \
            _Pragma("unroll") for (int u = 0; u < 32; ++u) {                                                \
                const int32_t i = (int32_t)(row0 + (R0)) + u;                                               \
                if (__any(consumer && nxt == i)) {                                                          \
                    while (consumer && nxt == i) {                                                          \
                        acc += tcur;                                                                        \
                        nxt = has2 ? raw_s : INT_MAX; /* promote the pending link (loaded at the last take) */ \
                        tcur = raw_t;                                                                       \
                        has2 = p < e; /* and fetch the one after it; untouched until then */                \
                        const int64_t pc = has2 ? p : e - 1;                                                \
                        raw_s = srcp[pc];                                                                   \
                        raw_t = termp[pc];                                                                  \
                        p += has2 ? 1 : 0;                                                                  \
                    }                                                                                       \
                }                                                                                           \
                acc += V[u]; /* then the restart addend (Model.cs:91-93,96-97) */                           \
            }                                                                                               \
        }                                                                                                   \
    }
        double ra[32], rb[32];
        CH3_READ(ra, 0)
#pragma unroll 1
        for (int r0 = 0; r0 + 64 <= CR; r0 += 64) {
            CH3_READ(rb, r0 + 32)
            CH3_FOLD(ra, r0)
            if (r0 + 64 < CR) CH3_READ(ra, r0 + 64)
            CH3_FOLD(rb, r0 + 32)
        }
        if (CR % 64 != 0) CH3_FOLD(ra, CR - 32)        // odd number of 32-row blocks (G = 64): ra already holds the last one
#undef CH3_READ
#undef CH3_FOLD
    }
    __builtin_amdgcn_s_setprio(0);
    if (consumer && s >= 0) {
        Y[(size_t)tile * (size_t)n * G + (size_t)s * G + k] = acc;
        if (nz_out && acc != 0.0)
            atomicOr(&nz_out[(size_t)tile * (((size_t)n + 31) / 32) + ((uint32_t)s >> 5)], 1u << (s & 31));
    }
}

// Model ctor, Model.cs:42-49: rank[seed] = nNodes, everything else 0
// (value-free path: Z receives the seed's z, ((1-d) n) * w_src[seed])
__global__ void k_init_seeds(int32_t n, int ntiles, int G, double *__restrict__ X,
                             const int32_t *__restrict__ seeds, uint32_t *__restrict__ nz,
                             double *__restrict__ Z = nullptr, const double *__restrict__ w_src = nullptr, double c1 = 0.0)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ntiles * G) return;
    const int32_t s = seeds[q];
    if (s < 0) return;
    X[(size_t)(q / G) * (size_t)n * G + (size_t)s * G + (q % G)] = (double)n;
    if (Z) { const double rw = c1 * (double)n; Z[(size_t)(q / G) * (size_t)n * G + (size_t)s * G + (q % G)] = rw * w_src[s]; }
    if (nz) atomicOr(&nz[(size_t)(q / G) * (((size_t)n + 31) / 32) + ((uint32_t)s >> 5)], 1u << (s & 31));
}

// Recommender.cs:20-24,29: the seed's RAW out-links of type LIKE are not candidates.
// Marked by overwriting their (final) score with -1 -- valid scores are >= 0.
__global__ __launch_bounds__(64) void k_exclude(int32_t n, int ntiles, int G, const int64_t *__restrict__ rowptr,
                                                const int32_t *__restrict__ dst, const uint8_t *__restrict__ etype,
                                                double *__restrict__ X, const int32_t *__restrict__ seeds)
{
    // one wave per seed slot, the lanes stride over the seed's raw out-links
    const int q = blockIdx.x;
    if (q >= ntiles * G) return;
    const int32_t s = seeds[q];
    if (s < 0) return;
    double *x = X + (size_t)(q / G) * (size_t)n * G + (q % G);
    const int64_t p1 = rowptr[s + 1];
    for (int64_t p = rowptr[s] + threadIdx.x; p < p1; p += WAVE)
        if (etype[p] == RWR_EDGE_LIKE) x[(size_t)dst[p] * G] = -1.0;
}

// value-free path: the z of the seeds' own rows, once the seed-row kernel (chain / scan / restart reduction) has
// left their final rank in Y:  z = ((1-d) * y) * w_src   (the product of Model.cs:84,87 for the next step)
__global__ void k_seed_z(int32_t n, int ntiles, int G, const double *__restrict__ Y, double *__restrict__ Z,
                         const int32_t *__restrict__ seeds, const double *__restrict__ w_src, double c1)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ntiles * G) return;
    const int32_t s = seeds[q];
    if (s < 0) return;
    const size_t at = (size_t)(q / G) * (size_t)n * G + (size_t)s * G + (q % G);
    const double rw = c1 * Y[at];
    Z[at] = rw * w_src[s];
}
// value-free path, rank vector supplied by the caller (Model.deliverRanks on its own): z of every row
__global__ __launch_bounds__(256) void k_make_z(int64_t elems, int G, const double *__restrict__ X, double *__restrict__ Z,
                                                const double *__restrict__ w_src, double c1)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= elems) return;
    const double rw = c1 * X[q];
    Z[q] = rw * w_src[q / G];
}

// row-partitioned mode, first steps: z of the slab's rows AND the bitmap of the rows that hold a non-zero (global row numbers).
// A thread per row, counted from the 64-aligned row at or below the slab's first: a wave then covers exactly two words of the
// bitmap and writes them whole (no atomics; the words of rows outside the slab stay as the memset left them).
__global__ __launch_bounds__(256) void k_make_z_nz(int32_t lo, int32_t hi, int G, const double *__restrict__ X, double *__restrict__ Zs,
                                                   const double *__restrict__ w_src, double c1, uint32_t *__restrict__ nz)
{
    const int64_t j = ((int64_t)lo & ~(int64_t)63) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // global row
    bool any = false;
    if (j >= lo && j < hi) {
        const double w = w_src[j];
        const size_t at = (size_t)(j - lo) * G;
        for (int k = 0; k < G; ++k) {
            const double xv = X[(size_t)j * G + k];
            const double rw = c1 * xv;
            Zs[at + k] = rw * w;
            any = any || xv != 0.0;
        }
    }
    const unsigned long long m = __ballot(any);
    const int lane = threadIdx.x & (WAVE - 1);
    if (lane == 0 && (uint32_t)m) nz[(uint64_t)j >> 5] = (uint32_t)m;
    if (lane == 32 && (uint32_t)(m >> 32)) nz[(uint64_t)j >> 5] = (uint32_t)(m >> 32);
}

// ------------------------------------------------------------------------------ host side

template <int G>
static void launch_spmm(rwr_graph *g, int tg, const double *X, double *Y, const int32_t *seeds, double c1,
                        int skip, const uint32_t *nz_in, uint32_t *nz_out, hipStream_t s,
                        const uint32_t *act = nullptr, const double *Zin = nullptr, double *Zout = nullptr,
                        bool hub_scan = false)
{
    // Zin != nullptr: value-free form -- the kernels gather Zin (z of the current ranks) instead of X and read no weights
    const bool vf = Zin != nullptr;
    const double *GS = vf ? Zin : X;   // gather source
    constexpr int RPW = WAVE / G;
    unsigned want = cdiv((size_t)g->n, (size_t)RPW * 4);
    unsigned gx = want < 8192u ? want : 8192u;
    static const int variant = [] { const char *e = getenv("RWR_SPMM"); return e ? atoi(e) : 1; }();
    if constexpr (G == 1) {
        if (tg == 1 && variant != 0) {
            launch_spmv_exact(g, X, Y, seeds, c1, skip, act, nz_out, s, Zin, Zout, hub_scan);
            return;
        }
    }
    if constexpr (G >= 8) {
        if (variant != 0) {
#define RWR_SPMM_LAUNCH3(CH, CHK, WR, VFF)                                                                         \
    hipLaunchKernelGGL((k_spmm_chunked<G, CH, CHK, WR, VFF>), dim3(gx, tg), dim3(256), 0, s, g->n, g->in_ptr.p,    \
                       g->in_src.p, g->in_w.p, g->row_order.p, GS, Y, seeds, c1, skip, nz_in, nz_out, act,         \
                       g->w_src.p, Zout)
#define RWR_SPMM_LAUNCH2(CH, CHK, WR)                    \
    {                                                    \
        if (vf) RWR_SPMM_LAUNCH3(CH, CHK, WR, true);     \
        else RWR_SPMM_LAUNCH3(CH, CHK, WR, false);       \
    }
#define RWR_SPMM_LAUNCH(CH)                                  \
    {                                                        \
        if (nz_in && nz_out) RWR_SPMM_LAUNCH2(CH, true, true)    \
        else if (nz_in) RWR_SPMM_LAUNCH2(CH, true, false)        \
        else RWR_SPMM_LAUNCH2(CH, false, false)                  \
    }
            // entries per chunk = row gathers in flight per lane (variant 2: 8, variant 3: 4)
            if (variant == 2) RWR_SPMM_LAUNCH(8)
            else if (variant == 3) RWR_SPMM_LAUNCH(4)
            else RWR_SPMM_LAUNCH((G > 16 ? 16 : G))
#undef RWR_SPMM_LAUNCH3
#undef RWR_SPMM_LAUNCH2
#undef RWR_SPMM_LAUNCH
            return;
        }
    }
    if (vf)
        hipLaunchKernelGGL((k_spmm<G, true>), dim3(gx, tg), dim3(256), 0, s, g->n, g->in_ptr.p, g->in_src.p, g->in_w.p,
                           g->row_order.p, GS, Y, seeds, c1, skip, g->w_src.p, Zout);
    else
        hipLaunchKernelGGL((k_spmm<G, false>), dim3(gx, tg), dim3(256), 0, s, g->n, g->in_ptr.p, g->in_src.p, g->in_w.p,
                           g->row_order.p, GS, Y, seeds, c1, skip, g->w_src.p, Zout);
}
template <int G>
// the addends of the links into the seeds; tiny, runs on the MAIN stream ahead of the fork so that the chain kernel is
// the first thing its stream has to dispatch once the fork event fires (it must get its CUs before the SpMM's
// half-million workgroups flood the dispatcher, or it only starts when the SpMM drains)
static void launch_seed_terms(rwr_graph *g, int tg, const double *X, const int32_t *seeds, double c1,
                              const int64_t *evoff, hipStream_t s, const double *Zin)
{
    const unsigned term_blocks = g->max_in_deg > 256 * 8 ? 8u : cdiv((size_t)(g->max_in_deg > 0 ? g->max_in_deg : 1), 256);
    if (Zin)
        hipLaunchKernelGGL((k_seed_terms<G, true>), dim3(term_blocks, tg * G), dim3(256), 0, s, g->n, g->in_ptr.p, g->in_src.p,
                           g->in_w.p, Zin, seeds, c1, evoff, g->d_evterm.p);
    else
        hipLaunchKernelGGL((k_seed_terms<G, false>), dim3(term_blocks, tg * G), dim3(256), 0, s, g->n, g->in_ptr.p, g->in_src.p,
                           g->in_w.p, X, seeds, c1, evoff, g->d_evterm.p);
}
template <int G>
static void launch_chain(rwr_graph *g, int tg, const double *X, double *Y, const int32_t *seeds, double c1,
                         const int64_t *evoff, uint32_t *nz_out, unsigned int *gate, hipStream_t s,
                         const uint32_t *nz_sparse, int variant)
{
    if (nz_sparse) {   // X still sparse: visit only its non-zero rows
        hipLaunchKernelGGL(k_seed_chain_sparse<G>, dim3(tg), dim3(64), 0, s, g->n, g->in_ptr.p, g->in_src.p,
                           g->dangling.p, X, Y, seeds, c1, evoff, g->d_evterm.p, nz_sparse, nz_out, gate);
        return;
    }
    if (variant != 0) {   // 0 = simple reference kernel
        constexpr size_t smem = 2 * CH3_CE * sizeof(double);
        // (per launch, not once per process: the attribute belongs to the current device, and one process may hold
        //  graphs on several devices)
        (void)hipFuncSetAttribute((const void *)k_seed_chain_roles<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(k_seed_chain_roles<G>, dim3(tg), dim3(WAVE + CH3_NST), smem, s, g->n, g->in_ptr.p,
                           g->in_src.p, g->dangling.p, X, Y, seeds, c1, evoff, g->d_evterm.p, nz_out, gate);
        return;
    }
    hipLaunchKernelGGL(k_seed_chain<G>, dim3(cdiv((size_t)tg * G, 64)), dim3(64), 0, s, g->n, tg, g->in_ptr.p,
                       g->in_src.p, g->in_w.p, g->dangling.p, X, Y, seeds, c1, nz_out);
}
constexpr int RP_GRID = 512;
constexpr int RP_BLOCK = 256;

#define RWR_DISPATCH_G(G, CALL)                 \
    switch (G) {                                \
        case 1: { constexpr int GG = 1; CALL; } break;   \
        case 2: { constexpr int GG = 2; CALL; } break;   \
        case 4: { constexpr int GG = 4; CALL; } break;   \
        case 8: { constexpr int GG = 8; CALL; } break;   \
        case 16: { constexpr int GG = 16; CALL; } break; \
        case 32: { constexpr int GG = 32; CALL; } break; \
        default: { constexpr int GG = 64; CALL; } break; \
    }

struct EvPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t get()
    {
        if (used == ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[used++];
    }
    ~EvPool()
    {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
};

// declared in rank.hip
int32_t rank_group_select(rwr_graph *g, int G, int tg, const int32_t *d_slot_k, int32_t top_n, const double *X,
                          const int32_t *d_seeds, hipStream_t s);
int rank_select_max_k();
int32_t emit_dangling(rwr_graph *g, const std::vector<int32_t> &rows, const std::vector<int32_t> &seeds, int32_t top_n,
                      hipStream_t s);
int32_t rank_tile(rwr_graph *g, int G, const int32_t *d_slot_k_tile, int32_t top_n, const double *X,
                  const int32_t *d_seeds_tile, hipStream_t s);

static int resolve_G(const rwr_graph *g, int32_t K)
{
    int G = g->opts.tile_seeds;
    if (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64) return G;
    // 32 seeds per tile (256-byte rows) measured best on the 100M-link graph: half the matrix re-streaming and
    // half the per-entry instruction work of 16, while 64 gains nothing more and lengthens the seed-row chain
    // (on the 20M-link graph 16 is a little faster: the ranking stage scales with the tile width)
    // ... and 32 again wins on dense graphs (hundreds of links per node: the MovieLens-shaped config, +10 %)
    const int cap = (g->n >= spmv_big_n() || g->nnz / (g->n > 0 ? g->n : 1) >= 64) ? 32 : 16;
    int want = 1;
    while (want < K && want < cap) want <<= 1;
    return want;
}

// One tile group's power iteration: init() = Model ctor (Model.cs:33-50), step() = deliverRanks + updateRanks
// (Model.cs:76-108).  After step() `X` holds the new ranks and `Y` still holds the previous ones.
struct GroupIter {
    rwr_graph *g;
    int G, tg;
    const int32_t *d_seeds;
    const int64_t *d_evoff;
    double c1;
    double *X, *Y;
    double *Zc = nullptr, *Zn = nullptr;   // value-free path: z of the current ranks / of the ranks being produced
    uint32_t *nz_cur = nullptr, *nz_oth = nullptr;
    int nz_iters = 0;
    int64_t it = 0;
    bool scan = false;   // exact mode: seed-row chain by the parallel binade scan (chain_scan.hip)
    int chain_kind = 1;  // 0 simple one-lane loop, 1 auto, 2 scan, 3 role-specialised fold
    int act_iters = 0;   // iterations whose SpMM only visits the out-neighbours of non-zero rows
    int64_t dense_steps = 0;   // steps whose SpMM walked every row (no frontier bitmap)
    bool addends_nonneg = false;   // weights, ranks and 1-d all >= 0 and finite: exact parallel reductions are allowed

    GroupIter(rwr_graph *g_, int G_, int tg_, const int32_t *seeds, const int64_t *evoff, double d)
        : g(g_), G(G_), tg(tg_), d_seeds(seeds), d_evoff(evoff), c1(1 - d) /* Model.cs:84: (1 - dampingFactor) */,
          X(g_->X.p), Y(g_->Y.p), Zc(g_->vf ? g_->Z0.p : nullptr), Zn(g_->vf ? g_->Z1.p : nullptr) {}

    // fresh = Model ctor (rank = n at the seed, 0 elsewhere);  !fresh = X already holds a caller-supplied rank vector
    // (Model.deliverRanks called on its own): no frontier knowledge, and the binade scan only if those ranks are >= 0
    int32_t init(bool fresh = true, bool ranks_nonneg = true)
    {
        const int32_t n = g->n;
        hipStream_t s = g->stream;
        const size_t elems = (size_t)tg * (size_t)n * G;
        if (fresh) RWR_HIP(hipMemsetAsync(X, 0, elems * sizeof(double), s));
        if (fresh && Zc) RWR_HIP(hipMemsetAsync(Zc, 0, elems * sizeof(double), s));
        if (!fresh && Zc) hipLaunchKernelGGL(k_make_z, dim3(cdiv(elems, 256)), dim3(256), 0, s, (int64_t)elems, G, X, Zc, g->w_src.p, c1);
        // frontier bitmaps for the first iterations (chunked SpMM only)
        static const int nz_iters_env = [] { const char *e = RWR_TUNE_ENV("RWR_NZ_ITERS"); return e ? atoi(e) : 4; }();
        static const int spmm_variant = [] { const char *e = getenv("RWR_SPMM"); return e ? atoi(e) : 1; }();
        // (skipping +0.0 addends is only a bitwise no-op while every accumulator is >= +0.0: weights must be >= 0)
        nz_iters = (G >= 8 && spmm_variant != 0 && g->nonneg) ? nz_iters_env : 0;
        // iterations 0 and 1: the non-zero rows are few enough to mark their out-neighbours; every other row is 0
        // (iteration 1 only on sparse graphs: on dense ones -- hundreds of links per node -- marking the 2-hop
        //  neighbourhood costs more atomics than the skipped rows save)
        static const int act_env = [] { const char *e = getenv("RWR_ACT_ITERS"); return e ? atoi(e) : -1; }();
        act_iters = act_env >= 0 ? act_env : ((g->nnz / (g->n > 0 ? g->n : 1)) <= 64 ? 2 : 1);
        // single exact seed (lane-per-row SpMV): row-level skipping only, for exactly those iterations
        // (on multi-million-node sparse graphs a single seed's 3-hop frontier is still worth marking: measured -7 % per call
        //  on the 6 M-node graph, +10 % on the 0.6 M-node one)
        if (G == 1 && tg == 1 && act_env < 0 && act_iters == 2 && g->n >= spmv_big_n()) act_iters = 3;
        if (G == 1 && tg == 1 && spmm_variant != 0 && g->nonneg) nz_iters = act_iters;
        // (a single seed on a graph of ego-network size: marking the frontier takes longer than the dense step it replaces --
        //  measured 38 us against 8 at 12 K nodes, 25 against 28 at 120 K)
        static const int64_t act_min_n = [] { const char *e = RWR_TUNE_ENV("RWR_ACT_MIN_N"); return e ? atol(e) : 200000l; }();
        if (G == 1 && tg == 1 && act_env < 0 && n < act_min_n) nz_iters = act_iters = 0;
        if (!fresh) nz_iters = act_iters = 0;
        const size_t nzw = ((size_t)n + 31) / 32;
        nz_cur = nz_iters > 0 ? g->d_nz.p : nullptr;
        nz_oth = nz_iters > 0 ? g->d_nz.p + (size_t)tg * nzw : nullptr;
        if (nz_cur) RWR_HIP(hipMemsetAsync(nz_cur, 0, (size_t)tg * nzw * sizeof(uint32_t), s));
        if (fresh) hipLaunchKernelGGL(k_init_seeds, dim3(cdiv((size_t)tg * G, 64)), dim3(64), 0, s, n, tg, G, X, d_seeds, nz_cur, Zc, g->w_src.p, c1);
        RWR_HIP(hipGetLastError());
        it = 0;
        // Seed-row chain of the dense iterations.  The role-specialised fold (k_seed_chain_roles) takes ~20 cycles per
        // row whatever the batch, which a large batch hides behind its SpMM; the binade scan reads X twice but is
        // parallel, so it wins whenever the SpMM of the group is shorter than the fold: small tile groups, and above
        // all the single-seed call of the unmodified harness.  RWR_CHAIN: 0 simple kernel, 1 auto, 2 scan, 3 roles.
        //            opts.seed_row_kernel (1 fold, 2 scan, 3 simple) takes precedence over the environment.
        static const int chain_env = [] { const char *e = getenv("RWR_CHAIN"); return e ? atoi(e) : 1; }();
        // auto: per step the fold costs ~10 ns per node whatever the batch (hidden if the SpMM is longer); the scan
        // costs ~6.7 ps per (node, seed) on top of an SpMM of ~0.89 ps per (link, seed)  (measured, MI355X, 20 M- and
        // 200 M-link graphs) => scan while  seeds * (0.89 * links/node + 6.7) < 10000  (about 280 seeds there)
        static const double scan_work = [] { const char *e = RWR_TUNE_ENV("RWR_SCAN_WORK"); return e ? atof(e) : 10000.0; }();
        const int sel = g->opts.seed_row_kernel;
        chain_kind = sel == 1 ? 3 : sel == 2 ? 2 : sel == 3 ? 0 : chain_env;
        const double per_seed = 0.89 * (double)g->nnz / (double)(g->n > 0 ? g->n : 1) + 6.7;
        addends_nonneg = c1 >= 0.0 && c1 <= 1.0 && g->nonneg && ranks_nonneg;
        scan = c1 >= 0.0 && c1 <= 1.0 && g->nonneg && ranks_nonneg &&
               (chain_kind == 2 || (chain_kind == 1 && (double)tg * G * per_seed < scan_work));
        if (scan) RWR_TRY(chain_scan_prepare(g, G, tg, d_seeds, s));
        if (Zc && G == 1 && tg == 1 && addends_nonneg) RWR_TRY(sweep_prepare(g));   // single seed: the source-block sweep (sweep.hip)
        // (the simple one-lane reference kernel of the seed row walks the weighted in-lists itself)
        if (Zc && chain_kind == 0 && !scan) RWR_TRY(ensure_in_w(g));
        return RWR_OK;
    }

    // last = no further step follows: the value-free path need not form the next z
    int32_t step(EvPool &pool, std::vector<hipEvent_t> &spmm_ev, std::vector<hipEvent_t> &chain_ev, bool last = false)
    {
        const int32_t n = g->n;
        hipStream_t s = g->stream, s2 = g->stream2;
        const bool prof = g->opts.profile != 0;
        const size_t nzw = ((size_t)n + 31) / 32;
        constexpr int GATE_SLOTS = 64;
        static const int use_gate = [] { const char *e = RWR_TUNE_ENV("RWR_GATE"); return e ? atoi(e) : 1; }();
        static const int serial = [] { const char *e = RWR_TUNE_ENV("RWR_CHAIN_SERIAL"); return e ? atoi(e) : 0; }();
        unsigned int *gate_it = nullptr;
        const uint32_t *nz_in = (it < nz_iters) ? nz_cur : nullptr;
        uint32_t *nz_out = (it + 1 < nz_iters) ? nz_oth : nullptr;
        if (nz_out) RWR_HIP(hipMemsetAsync(nz_out, 0, (size_t)tg * nzw * sizeof(uint32_t), s));
        uint32_t *act = (nz_in && it < act_iters) ? g->d_nz.p + 2 * (size_t)tg * nzw : nullptr;
        if (act) {
            RWR_HIP(hipMemsetAsync(act, 0, (size_t)tg * nzw * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_mark_active, dim3(cdiv(nzw, 4), tg), dim3(256), 0, s, n, nz_in, act, g->rowptr.p,
                               g->dst.p, g->etype.p);
        }
        if (serial) s2 = s;
        // (while X is sparse the bitmap-walking chain serves a whole tile at once; for a single seed the scan is cheaper)
        const bool scan_now = scan && (!act || G == 1);
        bool scan_side = false;
        bool seed_z_done = false;
        if (scan_now) {
            // the parallel chain.  Batches: on the main stream, ahead of the SpMM (which skips the seed rows).  A single seed:
            // on the second stream BESIDE the SpMV -- the two read the same vectors and write disjoint rows, and for one seed
            // the chain's five small kernels take as long as the SpMV itself (C2: ~100 us each), so the step costs their
            // maximum instead of their sum.
            static const int side_env = [] { const char *e = RWR_TUNE_ENV("RWR_SCAN_SIDE"); return e ? atoi(e) : 1; }();
            // (only on graphs large enough for the kernels to outlast the fork / join: measured -29 % per call at 224 K nodes,
            //  neutral at 120 K, +19 % at 12 K)
            scan_side = side_env && G == 1 && tg == 1 && s2 != s && g->n >= 100000;
            hipStream_t sc = scan_side ? s2 : s;
            if (scan_side) {
                RWR_HIP(hipEventRecord(g->ev_fork, s));
                RWR_HIP(hipStreamWaitEvent(s2, g->ev_fork, 0));
            }
            // (a single seed per tile: the chain kernels gather the link terms from z themselves and leave the seed row's next z)
            const bool self = chain_scan_self_contained(G);
            seed_z_done = self;
            if (!(self && Zc)) RWR_DISPATCH_G(G, launch_seed_terms<GG>(g, tg, X, d_seeds, c1, d_evoff, sc, Zc));
            hipEvent_t c0 = nullptr, c1e = nullptr;
            if (prof) { c0 = pool.get(); c1e = pool.get(); RWR_HIP(hipEventRecord(c0, sc)); }
            RWR_TRY(chain_scan_step(g, G, tg, X, Y, d_seeds, d_evoff, c1, nz_out, sc, self ? Zc : nullptr,
                                    (self && Zc && !last) ? Zn : nullptr));
            if (prof) { RWR_HIP(hipEventRecord(c1e, sc)); chain_ev.push_back(c0); chain_ev.push_back(c1e); }
            if (scan_side) RWR_HIP(hipEventRecord(g->ev_join, s2));
            s2 = s;
        } else {
            // fork: the seed-row chain runs beside the SpMM on the second stream
            RWR_DISPATCH_G(G, launch_seed_terms<GG>(g, tg, X, d_seeds, c1, d_evoff, s, Zc));
            gate_it = (use_gate && s2 != s) ? g->d_gate.p + (it % GATE_SLOTS) : nullptr;
            if (gate_it) RWR_HIP(hipMemsetAsync(gate_it, 0, sizeof(unsigned int), s));
            RWR_HIP(hipEventRecord(g->ev_fork, s));
            if (s2 != s) RWR_HIP(hipStreamWaitEvent(s2, g->ev_fork, 0));
            hipEvent_t c0 = nullptr, c1e = nullptr;
            if (prof) { c0 = pool.get(); c1e = pool.get(); RWR_HIP(hipEventRecord(c0, s2)); }
            RWR_DISPATCH_G(G, launch_chain<GG>(g, tg, X, Y, d_seeds, c1, d_evoff, nz_out, gate_it, s2, act ? nz_in : nullptr, chain_kind));
            if (prof) { RWR_HIP(hipEventRecord(c1e, s2)); chain_ev.push_back(c0); chain_ev.push_back(c1e); }
            RWR_HIP(hipEventRecord(g->ev_join, s2));
        }
        if (gate_it && !scan_now) {
            const unsigned expected = (unsigned)(tg < 192 ? tg : 192);
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, s, gate_it, expected);
        }
        hipEvent_t a = nullptr, b = nullptr;
        if (prof) { a = pool.get(); b = pool.get(); RWR_HIP(hipEventRecord(a, s)); }
        double *zout = (Zc && !last) ? Zn : nullptr;
        RWR_DISPATCH_G(G, launch_spmm<GG>(g, tg, X, Y, d_seeds, c1, 1, nz_in, nz_out, s, act, Zc, zout, addends_nonneg));
        if (prof) { RWR_HIP(hipEventRecord(b, s)); spmm_ev.push_back(a); spmm_ev.push_back(b); g->spmm_ev_dense.push_back(nz_in ? 0 : 1); }
        if (!nz_in) { g->stats.spmm_dense_launches += 1; ++dense_steps; }
        if ((s2 != s && !scan_now) || scan_side) RWR_HIP(hipStreamWaitEvent(s, g->ev_join, 0));
        // value-free path: the seed rows' own z, now that the seed-row kernel has left their rank in Y
        if (zout && !seed_z_done) hipLaunchKernelGGL(k_seed_z, dim3(cdiv((size_t)tg * G, 64)), dim3(64), 0, s, n, tg, G, Y, zout, d_seeds, g->w_src.p, c1);
        RWR_HIP(hipGetLastError());
        { double *t = X; X = Y; Y = t; }   // Model.updateRanks (Model.cs:103-108)
        { double *t = Zc; Zc = Zn; Zn = t; }
        { uint32_t *tz = nz_cur; nz_cur = nz_oth; nz_oth = tz; }
        g->stats.spmm_launches += 1;
        g->stats.chain_launches += 1;
        ++it;
        return RWR_OK;
    }
};

int32_t iterate_group(rwr_graph *g, int G, int tg, const int32_t *d_seeds, const int64_t *d_evoff, double d,
                      int64_t n_iter, double **final_X, EvPool &pool, std::vector<hipEvent_t> &spmm_ev,
                      std::vector<hipEvent_t> &chain_ev, int64_t *dense_steps)
{
    GroupIter gi(g, G, tg, d_seeds, d_evoff, d);
    RWR_TRY(gi.init());
    for (int64_t it = 0; it < n_iter; ++it) RWR_TRY(gi.step(pool, spmm_ev, chain_ev, it + 1 == n_iter));
    *final_X = gi.X;
    *dense_steps = gi.dense_steps;
    return RWR_OK;
}

static int32_t ensure_workspace(rwr_graph *g, int G, int32_t K, int *TG_out)
{
    const size_t n = (size_t)g->n;
    const int ntiles = (int)cdiv((size_t)K, (size_t)G);
    size_t cap = (size_t)g->opts.workspace_bytes;
    if (cap == 0) {
        size_t fr = 0, tot = 0;
        RWR_HIP(hipMemGetInfo(&fr, &tot));
        // what is already held by the rank matrices counts as available
        fr += (g->X.count + g->Y.count + g->Z0.count + g->Z1.count) * sizeof(double);
        // three quarters of what is free go to the rank matrices; the rest stays for the buffers sized after them (frontier
        // bitmaps and seed slots below -- inside the retry loop --, chain-scan cells, ranking keys) and for other handles
        cap = fr / 2 + fr / 4;
    }
    const size_t mats = g->vf ? 4 : 2;   // X, Y (+ the value-free path's z of the current and of the next ranks)
    const size_t per_tile = mats * n * (size_t)G * sizeof(double);
    int TG = g->opts.tile_group > 0 ? g->opts.tile_group : (int)(cap / (per_tile ? per_tile : 1));
    if (TG < 1) TG = 1;
    if (TG > ntiles) TG = ntiles;
    if (TG > 65535 / G) TG = 65535 / G;   // grid.y of the per-slot kernels is TG * G
    // exact mode: every tile's chain workgroup must be resident beside the SpMM (one per CU, see k_gate)
    if (g->opts.tile_group <= 0 && TG > 192) TG = 192;
    // the rank matrices: if the device cannot give what the sizing above asked for (other handles of the process -- the
    // reference runs up to ten host threads, Program.cs:11 -- may have taken their share since hipMemGetInfo was read),
    // halve the tile group and try again instead of failing the call
    for (;;) {
        int32_t rc = g->X.ensure((size_t)TG * n * G);
        if (rc == RWR_OK) rc = g->Y.ensure((size_t)TG * n * G);
        if (rc == RWR_OK && g->vf) {
            rc = g->Z0.ensure((size_t)TG * n * G);
            if (rc == RWR_OK) rc = g->Z1.ensure((size_t)TG * n * G);
        }
        if (rc == RWR_OK) rc = g->d_seeds.ensure((size_t)ntiles * G);
        if (rc == RWR_OK) rc = g->d_nz.ensure(3 * (size_t)TG * ((n + 31) / 32));   // X, Y non-zero rows + active destination rows
        if (rc == RWR_OK) rc = g->d_gate.ensure(64);
        if (rc == RWR_OK) break;
        if (rc != RWR_E_NOMEM || TG <= 1 || g->opts.tile_group > 0) return rc;
        (void)hipGetLastError();
        g->X.release(); g->Y.release(); g->Z0.release(); g->Z1.release(); g->d_nz.release();
        TG = (TG + 1) / 2;
    }
    *TG_out = TG;
    return RWR_OK;
}

// Seeds are dealt to tile slots by in-degree rank, round-robin over the tiles, so that the links INTO the
// seeds (the only non-streaming work of the exact seed-row kernel) spread evenly over the tiles instead of
// piling up in the tile that would hold the batch's hottest seeds.  slot_k maps a slot back to the
// caller's batch position; padding slots hold seed -1.  Also: offsets of every slot's in-link term list.
static int32_t upload_seed_slots(rwr_graph *g, const int32_t *seeds, int32_t K, int G, std::vector<int32_t> *slot_k_out)
{
    const int ntiles = (int)cdiv((size_t)K, (size_t)G);
    const size_t slots = (size_t)ntiles * G;
    std::vector<int32_t> hs(slots, -1), sk(slots, -1);
    std::vector<int64_t> off(slots + 1, 0);
    std::vector<int32_t> order(K);
    for (int32_t k = 0; k < K; ++k) order[k] = k;
    auto indeg = [&](int32_t k) { return g->h_in_ptr[seeds[k] + 1] - g->h_in_ptr[seeds[k]]; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return indeg(a) > indeg(b); });
    for (int32_t r = 0; r < K; ++r) {
        const size_t slot = (size_t)(r % ntiles) * G + (size_t)(r / ntiles);
        hs[slot] = seeds[order[r]];
        sk[slot] = order[r];
    }
    for (size_t q = 0; q < slots; ++q) {
        int64_t deg = hs[q] >= 0 ? g->h_in_ptr[hs[q] + 1] - g->h_in_ptr[hs[q]] : 0;
        off[q + 1] = off[q] + deg;
    }
    RWR_TRY(g->d_seeds.ensure(slots));
    RWR_TRY(g->d_slot_k.ensure(slots));
    RWR_TRY(g->d_evoff.ensure(slots + 1));
    RWR_TRY(g->d_evterm.ensure((size_t)off[slots] + 1));
    RWR_HIP(hipMemcpy(g->d_seeds.p, hs.data(), slots * sizeof(int32_t), hipMemcpyHostToDevice));
    RWR_HIP(hipMemcpy(g->d_slot_k.p, sk.data(), slots * sizeof(int32_t), hipMemcpyHostToDevice));
    RWR_HIP(hipMemcpy(g->d_evoff.p, off.data(), (slots + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (slot_k_out) *slot_k_out = sk;
    return RWR_OK;
}

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static int32_t drain_events(std::vector<hipEvent_t> &v, double *acc)
{
    for (size_t i = 0; i + 1 < v.size(); i += 2) {
        float ms = 0.f;
        RWR_HIP(hipEventElapsedTime(&ms, v[i], v[i + 1]));
        *acc += ms;
    }
    v.clear();
    return RWR_OK;
}

int32_t recommend_batch(rwr_graph *g, const int32_t *seeds, int32_t K, double d, int32_t n_iter, int32_t top_n,
                        int64_t *ids, double *scores, int32_t *counts, int64_t row_stride)
{
    const double t_begin = now_ms();
    const int32_t n = g->n;
    if (!g->nonneg) {
        // the exclusion marker (-1) and the ranking keys assume scores >= 0, i.e. weights >= 0 and finite row sums -- what
        // the reference's loader produces (DataLoader.cs:293-294,431-432).  Model.run still works on such a graph.
        set_error("Recommendation needs non-negative finite link weights and positive row sums (a raw weight is negative or "
                  "NaN, or the explicit weights of a node sum to 0 or overflow); rwr_model_run accepts such graphs");
        return RWR_E_UNSUPPORTED;
    }
    if (!(d >= 0.0 && d <= 1.0)) {
        // outside [0, 1] ranks go negative (or NaN): the exclusion marker and the ranking keys assume scores >= 0
        set_error("Recommendation needs a damping factor in [0, 1] (got %g); rwr_model_run accepts any value", d);
        return RWR_E_UNSUPPORTED;
    }
    for (int32_t k = 0; k < K; ++k)
        if (seeds[k] < 0 || seeds[k] >= n) {
            set_error("seed %d (batch position %d) is outside [0, %d)", seeds[k], k, n);
            return RWR_E_RANGE;
        }
    if (K == 1 && small_path_ok(g) && small_path_seed_ok(g, seeds[0])) {
        // ego-network-sized graph, one seed (the unmodified harness's call, Experiment.cs:109): the whole call is one launch
        RWR_TRY(recommend_small(g, seeds[0], d, n_iter, top_n, ids, scores, counts));
        g->stats.seeds_done += 1;
        g->stats.total_wall_ms += now_ms() - t_begin;
        return RWR_OK;
    }
    // dangling seeds (no explicit out-link) are answered directly (see k_emit_dangling); the rest is iterated
    const int32_t K_all = K;
    std::vector<int32_t> live_seeds, live_rows, dang_seeds, dang_rows;
    const bool shortcut = top_n <= rank_select_max_k() && K_all > 1;
    for (int32_t k = 0; k < K_all; ++k) {
        if (shortcut && g->h_dangling[seeds[k]]) { dang_seeds.push_back(seeds[k]); dang_rows.push_back(k); }
        else { live_seeds.push_back(seeds[k]); live_rows.push_back(k); }
    }
    const bool any_dangling = !dang_seeds.empty();
    if (any_dangling) { seeds = live_seeds.data(); K = (int32_t)live_seeds.size(); }
    hipStream_t s = g->stream;
    const size_t out_all = (size_t)K_all * (size_t)top_n;
    // (every emitter indexes these tables by the caller's batch position < K_all: no padding rows are ever written)
    RWR_TRY(g->d_out_id.ensure(out_all + 64));
    RWR_TRY(g->d_out_score.ensure(out_all + 64));
    RWR_TRY(g->d_counts.ensure((size_t)K_all + 64));
    // output tables are indexed by the caller's batch position (K_all rows)
    RWR_HIP(hipMemsetAsync(g->d_out_id.p, 0, out_all * sizeof(int64_t), s));
    RWR_HIP(hipMemsetAsync(g->d_out_score.p, 0, out_all * sizeof(double), s));
    RWR_HIP(hipMemsetAsync(g->d_counts.p, 0, (size_t)K_all * sizeof(int32_t), s));
    if (K == 0) {   // every seed of the batch is dangling
        RWR_TRY(emit_dangling(g, dang_rows, dang_seeds, top_n, s));
        std::vector<int32_t> hc0((size_t)K_all);
        RWR_HIP(hipMemcpyAsync(hc0.data(), g->d_counts.p, hc0.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (ids && scores) {
            RWR_HIP(hipMemcpy2DAsync(ids, (size_t)row_stride * sizeof(int64_t), g->d_out_id.p, (size_t)top_n * sizeof(int64_t),
                                     (size_t)top_n * sizeof(int64_t), (size_t)K_all, hipMemcpyDeviceToHost, s));
            RWR_HIP(hipMemcpy2DAsync(scores, (size_t)row_stride * sizeof(double), g->d_out_score.p,
                                     (size_t)top_n * sizeof(double), (size_t)top_n * sizeof(double), (size_t)K_all,
                                     hipMemcpyDeviceToHost, s));
        }
        RWR_HIP(hipStreamSynchronize(s));
        for (int32_t k = 0; k < K_all; ++k) counts[k] = hc0[k];
        g->stats.seeds_done += K_all;
        g->stats.total_wall_ms += now_ms() - t_begin;
        return RWR_OK;
    }
    const int G = resolve_G(g, K);
    int TG = 1;
    RWR_TRY(ensure_workspace(g, G, K, &TG));
    const int ntiles = (int)cdiv((size_t)K, (size_t)G);
    std::vector<int32_t> slot_k;
    RWR_TRY(upload_seed_slots(g, seeds, K, G, &slot_k));
    if (any_dangling) {   // slots map to positions in the live list: translate to the caller's batch positions
        for (auto &v : slot_k) if (v >= 0) v = live_rows[v];
        RWR_HIP(hipMemcpy(g->d_slot_k.p, slot_k.data(), slot_k.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }

    EvPool pool;
    std::vector<hipEvent_t> spmm_ev, chain_ev, rank_ev, iter_ev;
    const bool prof = g->opts.profile != 0;
    g->spmm_ev_dense.clear();
    for (int t0 = 0; t0 < ntiles; t0 += TG) {
        const int tg = (ntiles - t0 < TG) ? (ntiles - t0) : TG;
        const int32_t *dseeds = g->d_seeds.p + (size_t)t0 * G;
        double *Xf = nullptr;
        hipEvent_t i0 = nullptr, i1 = nullptr;
        if (prof) { i0 = pool.get(); i1 = pool.get(); RWR_HIP(hipEventRecord(i0, s)); }
        int64_t dense_steps = 0;
        RWR_TRY(iterate_group(g, G, tg, dseeds, g->d_evoff.p + (size_t)t0 * G, d, n_iter, &Xf, pool, spmm_ev, chain_ev, &dense_steps));
        if (prof) { RWR_HIP(hipEventRecord(i1, s)); iter_ev.push_back(i0); iter_ev.push_back(i1); }
        int32_t real = 0;
        for (size_t q = (size_t)t0 * G; q < (size_t)(t0 + tg) * G; ++q) real += slot_k[q] >= 0;
        g->stats.spmm_seed_steps += (int64_t)real * n_iter;
        g->stats.spmm_dense_seed_steps += (int64_t)real * dense_steps;
        hipEvent_t a = nullptr, b = nullptr;
        if (prof) { a = pool.get(); b = pool.get(); RWR_HIP(hipEventRecord(a, s)); }
        hipLaunchKernelGGL(k_exclude, dim3((unsigned)(tg * G)), dim3(64), 0, s, n, tg, G, g->rowptr.p,
                           g->dst.p, g->etype.p, Xf, dseeds);
        RWR_HIP(hipGetLastError());
        static const int force_sort = [] { const char *e = RWR_TUNE_ENV("RWR_RANK_SORT"); return e ? atoi(e) : 0; }();
        if (top_n <= rank_select_max_k() && !force_sort) {
            RWR_TRY(rank_group_select(g, G, tg, g->d_slot_k.p + (size_t)t0 * G, top_n, Xf, dseeds, s));
        } else {
            for (int t = 0; t < tg; ++t) {
                RWR_TRY(rank_tile(g, G, g->d_slot_k.p + (size_t)(t0 + t) * G, top_n, Xf + (size_t)t * (size_t)n * G,
                                  dseeds + (size_t)t * G, s));
            }
        }
        if (prof) { RWR_HIP(hipEventRecord(b, s)); rank_ev.push_back(a); rank_ev.push_back(b); }
    }
    if (any_dangling) RWR_TRY(emit_dangling(g, dang_rows, dang_seeds, top_n, s));
    // results: K_all x top_n (device rows are top_n wide; host rows are row_stride wide)
    std::vector<int32_t> hc((size_t)K_all);
    RWR_HIP(hipMemcpyAsync(hc.data(), g->d_counts.p, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (ids && scores) {   // (NULL: the caller consumes the lists on the device, e.g. rwr_recommend_eval)
        RWR_HIP(hipMemcpy2DAsync(ids, (size_t)row_stride * sizeof(int64_t), g->d_out_id.p, (size_t)top_n * sizeof(int64_t),
                                 (size_t)top_n * sizeof(int64_t), (size_t)K_all, hipMemcpyDeviceToHost, s));
        RWR_HIP(hipMemcpy2DAsync(scores, (size_t)row_stride * sizeof(double), g->d_out_score.p,
                                 (size_t)top_n * sizeof(double), (size_t)top_n * sizeof(double), (size_t)K_all,
                                 hipMemcpyDeviceToHost, s));
    }
    RWR_HIP(hipStreamSynchronize(s));
    RWR_HIP(hipStreamSynchronize(g->stream2));
    for (int32_t k = 0; k < K_all; ++k) counts[k] = hc[k];
    if (prof) RWR_TRY(chain_scan_collect(g, s));
    if (prof) {
        for (size_t i = 0; i + 1 < spmm_ev.size(); i += 2)
            if (i / 2 < g->spmm_ev_dense.size() && g->spmm_ev_dense[i / 2]) {
                float ms = 0.f;
                RWR_HIP(hipEventElapsedTime(&ms, spmm_ev[i], spmm_ev[i + 1]));
                g->stats.spmm_dense_ms += ms;
            }
        RWR_TRY(drain_events(spmm_ev, &g->stats.spmm_ms));
        RWR_TRY(drain_events(chain_ev, &g->stats.chain_ms));
        RWR_TRY(drain_events(rank_ev, &g->stats.rank_ms));
        RWR_TRY(drain_events(iter_ev, &g->stats.iterate_wall_ms));
    }
    g->stats.tile_seeds = G;
    g->stats.tile_group = TG;
    g->stats.seeds_done += K_all;
    g->stats.total_wall_ms += now_ms() - t_begin;
    return RWR_OK;
}

// ---- Model.run() / run(double) / global model (Model.cs:14-31, 52-66, 110-115) -------------------------------

// sum over i of |a_i - b_i|  (checkConvergence, Model.cs:110-115) or of the restart addends of the global model;
// deterministic two-level tree (the reference sums sequentially: tolerance-level difference)
constexpr int RED_GRID = 256;
__global__ __launch_bounds__(256) void k_l1_partial(const double *__restrict__ a, const double *__restrict__ b, int32_t n,
                                                    double *__restrict__ part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = a[i], y = b[i];
        acc += (x > y) ? (x - y) : (y - x);                                  // Model.cs:113
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_absdiff(const double *__restrict__ a, const double *__restrict__ b, int32_t n,
                                                 double *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double x = a[i], y = b[i]; d[i] = (x > y) ? (x - y) : (y - x); }   // Math.Abs(rank - nextRank), Model.cs:113
}
__global__ __launch_bounds__(256) void k_rr_partial(const double *__restrict__ x, const uint8_t *__restrict__ dangling,
                                                    int32_t n, double c1, double *__restrict__ part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double xi = x[i], rw = c1 * xi;
        acc += dangling[i] ? xi : (xi - rw);                                 // Model.cs:91 / :97
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void k_sum_parts(const double *__restrict__ part, int nparts, double *__restrict__ out)
{
    double s = 0.0;
    for (int b = 0; b < nparts; ++b) s += part[b];
    *out = s;
}
__global__ void k_fill(double *__restrict__ x, int32_t n, double v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
// global model: every node receives restart mass / n  (restart[r] = 1/n, Model.cs:29,92-93,96-97)
__global__ void k_add_restart_share(double *__restrict__ y, int32_t n, const double *__restrict__ total, double inv_n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += *total * inv_n;
}

int32_t model_run(rwr_graph *g, int32_t seed, double d, int32_t run_mode, double value, double *rank_out,
                  int64_t *iters_out)
{
    const int32_t n = g->n;
    if (seed < -1 || seed >= n) {
        set_error("seed %d is outside [0, %d) (and is not -1 = global model)", seed, n);
        return RWR_E_RANGE;
    }
    // one model, one lane per row: the rank vector is contiguous
    const int G = 1;
    int TG = 1;
    RWR_TRY(ensure_workspace(g, G, 1, &TG));
    hipStream_t s = g->stream;
    EvPool pool;
    std::vector<hipEvent_t> a, b;
    static const int64_t max_iters = [] { const char *e = getenv("RWR_MAX_ITERS"); return e ? atoll(e) : (int64_t)1000000; }();
    const bool by_count = run_mode == RWR_RUN_ITERATIONS;
    // Model.cs:53: threshold = (1 / double.MaxValue) * n   (a subnormal-scale number: "until nothing changes")
    const double threshold = run_mode == RWR_RUN_DEFAULT_THRESHOLD ? (1 / 1.7976931348623157e308) * n : value;
    int64_t T = by_count ? (int64_t)value : max_iters;
    if (T < 0) T = 0;
    RWR_TRY(g->d_part.ensure(RED_GRID + 8));
    double *part = g->d_part.p, *scalar = g->d_part.p + RED_GRID;
    int64_t done = 0;
    bool converged = false;
    double *Xf = nullptr;

    if (seed >= 0) {
        RWR_TRY(upload_seed_slots(g, &seed, 1, 1, nullptr));
        GroupIter gi(g, G, 1, g->d_seeds.p, g->d_evoff.p, d);
        RWR_TRY(gi.init());
        while (done < T) {
            RWR_TRY(gi.step(pool, a, b));                                   // deliverRanks + updateRanks
            ++done;
            if (!by_count) {                                                // checkConvergence (Model.cs:58-65)
                // the reference's sequential sum of |rank[i] - nextRank[i]|, reproduced bit for bit by the binade scan
                // (d_evterm must exist for the scan's pointer arithmetic even though no link term is read)
                RWR_TRY(g->cs_diff.ensure((size_t)n));
                RWR_TRY(g->d_evterm.ensure(1));
                hipLaunchKernelGGL(k_absdiff, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, gi.Y, gi.X, n, g->cs_diff.p);
                RWR_TRY(chain_scan_sum(g, g->cs_diff.p, scalar, s));
                double diff = 0;
                RWR_HIP(hipMemcpyAsync(&diff, scalar, sizeof(double), hipMemcpyDeviceToHost, s));
                RWR_HIP(hipStreamSynchronize(s));
                if (diff < threshold) { converged = true; break; }
                a.clear(); b.clear(); pool.used = 0;                        // (synchronised above: safe to recycle)
            }
        }
        Xf = gi.X;
    } else {
        // global model (Model.cs:14-31): rank = 1, restart = 1/n.  Every row receives the restart mass of every node,
        // interleaved in node order in the reference; here: edge part in reference order + (tree-summed mass)/n.
        // Tolerance parity only.
        RWR_TRY(ensure_in_w(g));   // (the global model runs the weighted kernels)
        double *X = g->X.p, *Y = g->Y.p;
        const double c1 = 1 - d, inv_n = 1.0 / n;
        hipLaunchKernelGGL(k_fill, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, X, n, 1.0);
        int32_t no_seed = -1;
        RWR_HIP(hipMemcpyAsync(g->d_seeds.p, &no_seed, sizeof(int32_t), hipMemcpyHostToDevice, s));
        while (done < T) {
            hipLaunchKernelGGL(k_rr_partial, dim3(RED_GRID), dim3(256), 0, s, X, g->dangling.p, n, c1, part);
            hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1), 0, s, part, RED_GRID, scalar);
            launch_spmm<1>(g, 1, X, Y, g->d_seeds.p, c1, 0, nullptr, nullptr, s);
            hipLaunchKernelGGL(k_add_restart_share, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, Y, n, scalar, inv_n);
            RWR_HIP(hipGetLastError());
            { double *t = X; X = Y; Y = t; }
            ++done;
            if (!by_count) {
                hipLaunchKernelGGL(k_l1_partial, dim3(RED_GRID), dim3(256), 0, s, Y, X, n, part);
                hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1), 0, s, part, RED_GRID, scalar);
                double diff = 0;
                RWR_HIP(hipMemcpyAsync(&diff, scalar, sizeof(double), hipMemcpyDeviceToHost, s));
                RWR_HIP(hipStreamSynchronize(s));
                if (diff < threshold) { converged = true; break; }
            }
        }
        Xf = X;
    }
    if (!by_count && !converged) {
        set_error("rwr_model_run: no convergence within %lld iterations (RWR_MAX_ITERS)", (long long)max_iters);
        return RWR_E_UNSUPPORTED;
    }
    RWR_HIP(hipMemcpyAsync(rank_out, Xf, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    RWR_HIP(hipStreamSynchronize(g->stream2));
    if (iters_out) *iters_out = done;
    return RWR_OK;
}

// One Model.deliverRanks (Model.cs:76-100) on a rank vector supplied by the caller: backs the public step-by-step API
// (deliverRanks / updateRanks / checkConvergence, Model.cs:76,103,110) for hosts that drive the loop themselves.
int32_t model_deliver(rwr_graph *g, int32_t seed, double d, const double *rank_in, double *next_out)
{
    const int32_t n = g->n;
    if (seed < -1 || seed >= n) {
        set_error("seed %d is outside [0, %d) (and is not -1 = global model)", seed, n);
        return RWR_E_RANGE;
    }
    int TG = 1;
    RWR_TRY(ensure_workspace(g, 1, 1, &TG));
    hipStream_t s = g->stream;
    bool nonneg = true;
    for (int32_t i = 0; i < n; ++i)
        if (!(rank_in[i] >= 0.0)) { nonneg = false; break; }
    RWR_HIP(hipMemcpyAsync(g->X.p, rank_in, sizeof(double) * n, hipMemcpyHostToDevice, s));
    double *out = nullptr;
    if (seed >= 0) {
        RWR_TRY(upload_seed_slots(g, &seed, 1, 1, nullptr));
        EvPool pool;
        std::vector<hipEvent_t> a, b;
        GroupIter gi(g, 1, 1, g->d_seeds.p, g->d_evoff.p, d);
        RWR_TRY(gi.init(false, nonneg));
        RWR_TRY(gi.step(pool, a, b));
        RWR_HIP(hipStreamSynchronize(s));
        RWR_HIP(hipStreamSynchronize(g->stream2));
        out = gi.X;                                   // (step() swapped: X holds nextRank)
    } else {
        RWR_TRY(ensure_in_w(g));
        RWR_TRY(g->d_part.ensure(RED_GRID + 8));
        double *part = g->d_part.p, *scalar = g->d_part.p + RED_GRID;
        const double c1 = 1 - d, inv_n = 1.0 / n;
        int32_t no_seed = -1;
        RWR_TRY(g->d_seeds.ensure(1));
        RWR_HIP(hipMemcpyAsync(g->d_seeds.p, &no_seed, sizeof(int32_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_rr_partial, dim3(RED_GRID), dim3(256), 0, s, g->X.p, g->dangling.p, n, c1, part);
        hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1), 0, s, part, RED_GRID, scalar);
        launch_spmm<1>(g, 1, g->X.p, g->Y.p, g->d_seeds.p, c1, 0, nullptr, nullptr, s);
        hipLaunchKernelGGL(k_add_restart_share, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, g->Y.p, n, scalar, inv_n);
        RWR_HIP(hipGetLastError());
        out = g->Y.p;
    }
    RWR_HIP(hipMemcpyAsync(next_out, out, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    return RWR_OK;
}

// ---- row-partitioned mode (include/rwr.h: rwr_part_*) -----------------------------------------------------------

// restart mass of the slab's rows only: r[k] = sum over i in [lo, hi) of (dangling_i ? x_i : x_i - (1-d) x_i)
template <int G>
__global__ __launch_bounds__(RP_BLOCK) void k_slab_restart_partial(int32_t lo, int32_t hi,
                                                                   const uint8_t *__restrict__ dangling,
                                                                   const double *__restrict__ x,
                                                                   double *__restrict__ part, double c1)
{
    constexpr int RL = RP_BLOCK / G;
    __shared__ double sh[RP_BLOCK];
    const int k = threadIdx.x % G, rl = threadIdx.x / G;
    double acc = 0.0;
    for (int64_t i = (int64_t)lo + (int64_t)blockIdx.x * RL + rl; i < hi; i += (int64_t)gridDim.x * RL) {
        const double xi = x[(size_t)i * G + k];
        const double rw = c1 * xi;
        acc += dangling[i] ? xi : (xi - rw);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int half = RL / 2; half >= 1; half >>= 1) {
        if (rl < half) sh[threadIdx.x] += sh[threadIdx.x + half * G];
        __syncthreads();
    }
    if (rl == 0) part[(size_t)blockIdx.x * G + k] = sh[k];
}
__global__ void k_slab_restart_final(int G, int nblk, const double *__restrict__ part, double *__restrict__ r)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G) return;
    double R = 0.0;
    for (int b = 0; b < nblk; ++b) R += part[(size_t)b * G + k];
    r[k] = R;
}
__global__ void k_part_add_restart(int G, double *__restrict__ y, const double *__restrict__ r,
                                   const int32_t *__restrict__ seeds)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G) return;
    const int32_t s = seeds[k];
    if (s >= 0) y[(size_t)s * G + k] += r[k];
}

int32_t part_begin(rwr_graph *g, int32_t lo, int32_t hi, const int32_t *seeds, int32_t K, double d, double *x,
                   int32_t *G_out)
{
    const int32_t n = g->n;
    if (lo < 0 || hi > n || lo > hi) { set_error("rwr_part_begin: bad slab [%d, %d)", lo, hi); return RWR_E_RANGE; }
    if (K < 1 || K > 64) { set_error("rwr_part_begin: K must be 1..64 in row-partitioned mode"); return RWR_E_UNSUPPORTED; }
    for (int32_t k = 0; k < K; ++k)
        if (seeds[k] < 0 || seeds[k] >= n) { set_error("seed %d is outside [0, %d)", seeds[k], n); return RWR_E_RANGE; }
    int G = 1;
    while (G < K) G <<= 1;
    // value-free graphs (every BASELINE configuration): the slab step forms z = ((1-d) x) * w_src for the slab's rows and
    // the kernels gather z -- 4 instead of 12 matrix bytes per entry, and no in_w (2 GB per rank at 1/8 of the 1 G-like
    // graph) is ever materialised; other graphs run the weighted kernels on the caller's rank matrix
    if (g->vf) RWR_TRY(g->Z0.ensure((size_t)(hi - lo > 0 ? hi - lo : 1) * (size_t)G));
    else RWR_TRY(ensure_in_w(g));
    g->part_lo = lo; g->part_hi = hi; g->part_G = G; g->part_K = K; g->part_c1 = 1 - d;
    g->part_steps = 0;
    if (g->vf && g->nonneg && G >= 8) RWR_TRY(g->d_nz.ensure(2 * (((size_t)n + 31) / 32)));   // frontier of the first steps
    g->part_seeds.assign((size_t)G, -1);
    for (int32_t k = 0; k < K; ++k) g->part_seeds[k] = seeds[k];
    RWR_TRY(g->d_seeds.ensure(G));
    RWR_TRY(g->d_part.ensure((size_t)RP_GRID * G + G));
    RWR_HIP(hipMemcpy(g->d_seeds.p, g->part_seeds.data(), G * sizeof(int32_t), hipMemcpyHostToDevice));
    hipStream_t s = g->stream;
    RWR_HIP(hipMemsetAsync(x, 0, (size_t)n * G * sizeof(double), s));
    hipLaunchKernelGGL(k_init_seeds, dim3(1), dim3(64), 0, s, n, 1, G, x, g->d_seeds.p, (uint32_t *)nullptr);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(s));
    if (G_out) *G_out = G;
    return RWR_OK;
}

// y = (1-d) P_slab^T x over all n rows.  Value-free graphs: z of the slab's rows first (the in-lists of the slab graph
// hold in-slab sources only, so the kernels' gathers z[source * G + k] never leave [lo, hi): the buffer holds just the
// slab, addressed through a base pointer shifted by lo rows)
static int32_t part_spmm(rwr_graph *g, const double *x, double *y, hipStream_t s)
{
    const int G = g->part_G;
    const double c1 = g->part_c1;
    if (g->vf) {
        const int64_t rows = (int64_t)g->part_hi - g->part_lo;
        if (rows > 0 && (!g->Z0.p || g->Z0.count < (size_t)rows * G)) { set_error("rwr_part_step: call rwr_part_begin"); return RWR_E_INVALID; }
        const int64_t elems = rows * G;
        const double *zin = g->Z0.p - (size_t)g->part_lo * G;
        // the first steps after rwr_part_begin (the ranks are still concentrated around the seeds): mark the slab's non-zero rows
        // and -- the first two steps -- the destination rows their out-links reach, and let the SpMM skip every other row and
        // every gather of an all-zero source row: what the seed path does in its first iterations (GroupIter::init); exact
        // for ANY rank matrix (a skipped addend is +0.0), the step counter only decides whether the marking is worth its cost
        static const int part_act_env = [] { const char *e = RWR_TUNE_ENV("RWR_PART_ACT_STEPS"); return e ? atoi(e) : 2; }();
        static const int part_nz_env = [] { const char *e = RWR_TUNE_ENV("RWR_PART_NZ_STEPS"); return e ? atoi(e) : 4; }();
        const size_t nzw = ((size_t)g->n + 31) / 32;
        const bool frontier = g->part_steps < part_nz_env && g->nonneg && G >= 8 && rows > 0 && g->d_nz.p && g->d_nz.count >= 2 * nzw;
        const bool mark = frontier && g->part_steps < part_act_env;      // (later steps: only the per-entry probe of the sources)
        ++g->part_steps;
        if (frontier) {
            uint32_t *nz = g->d_nz.p, *act = g->d_nz.p + nzw;
            RWR_HIP(hipMemsetAsync(nz, 0, (mark ? 2 : 1) * nzw * sizeof(uint32_t), s));
            const int64_t first = (int64_t)g->part_lo & ~(int64_t)63;
            hipLaunchKernelGGL(k_make_z_nz, dim3(cdiv((size_t)((int64_t)g->part_hi - first), 256)), dim3(256), 0, s, g->part_lo, g->part_hi, G, x,
                               g->Z0.p, g->w_src.p, c1, nz);
            if (mark) hipLaunchKernelGGL(k_mark_active, dim3(cdiv(nzw, 4), 1), dim3(256), 0, s, g->n, nz, act, g->rowptr.p, g->dst.p, g->etype.p);
            RWR_DISPATCH_G(G, launch_spmm<GG>(g, 1, x, y, g->d_seeds.p, c1, 0, nz, nullptr, s, mark ? act : nullptr, zin, nullptr, false));
            return RWR_OK;
        }
        if (elems > 0)
            hipLaunchKernelGGL(k_make_z, dim3(cdiv((size_t)elems, 256)), dim3(256), 0, s, elems, G, x + (size_t)g->part_lo * G, g->Z0.p,
                               g->w_src.p + g->part_lo, c1);
        RWR_DISPATCH_G(G, launch_spmm<GG>(g, 1, x, y, g->d_seeds.p, c1, 0, nullptr, nullptr, s, nullptr, zin, nullptr, false));
    } else {
        if (!g->in_w.p) { set_error("rwr_part_step: call rwr_part_begin"); return RWR_E_INVALID; }
        RWR_DISPATCH_G(G, launch_spmm<GG>(g, 1, x, y, g->d_seeds.p, c1, 0, nullptr, nullptr, s));
    }
    return RWR_OK;
}

int32_t part_local_step(rwr_graph *g, const double *x, double *y, double *r)
{
    const int G = g->part_G;
    if (G == 0) { set_error("rwr_part_local_step: rwr_part_begin has not been called"); return RWR_E_INVALID; }
    hipStream_t s = g->stream;
    const double c1 = g->part_c1;
    RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_slab_restart_partial<GG>, dim3(RP_GRID), dim3(RP_BLOCK), 0, s, g->part_lo,
                                         g->part_hi, g->dangling.p, x, g->d_part.p, c1));
    hipLaunchKernelGGL(k_slab_restart_final, dim3(1), dim3(64), 0, s, G, RP_GRID, g->d_part.p, r);
    // the graph holds only this slab's out-links, so the in-lists contain only in-slab sources
    RWR_TRY(part_spmm(g, x, y, s));
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(s));
    return RWR_OK;
}

// One whole local step on the caller's stream, no host synchronisation: partial y over all rows, plus this slab's restart
// mass at the seeds' rows -- the sum over the ranks of y is then the next rank matrix (include/rwr.h: rwr_part_step).
int32_t part_step(rwr_graph *g, const double *x, double *y, hipStream_t stream)
{
    const int G = g->part_G;
    if (G == 0) { set_error("rwr_part_step: rwr_part_begin has not been called"); return RWR_E_INVALID; }
    hipStream_t s = stream;                              // exactly the caller's stream (NULL = the device's default stream)
    const double c1 = g->part_c1;
    double *r = g->d_part.p + (size_t)RP_GRID * G;       // (behind the per-block partials)
    RWR_DISPATCH_G(G, hipLaunchKernelGGL(k_slab_restart_partial<GG>, dim3(RP_GRID), dim3(RP_BLOCK), 0, s, g->part_lo,
                                         g->part_hi, g->dangling.p, x, g->d_part.p, c1));
    hipLaunchKernelGGL(k_slab_restart_final, dim3(1), dim3(64), 0, s, G, RP_GRID, g->d_part.p, r);
    RWR_TRY(part_spmm(g, x, y, s));
    hipLaunchKernelGGL(k_part_add_restart, dim3(1), dim3(64), 0, s, G, y, r, g->d_seeds.p);
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

int32_t part_finish_step(rwr_graph *g, double *y, const double *r)
{
    const int G = g->part_G;
    if (G == 0) { set_error("rwr_part_finish_step: rwr_part_begin has not been called"); return RWR_E_INVALID; }
    hipLaunchKernelGGL(k_part_add_restart, dim3(1), dim3(64), 0, g->stream, G, y, r, g->d_seeds.p);
    RWR_HIP(hipGetLastError());
    RWR_HIP(hipStreamSynchronize(g->stream));
    return RWR_OK;
}

int32_t part_rank(rwr_graph *g, double *x, int32_t top_n, int64_t *ids, double *scores, int32_t *counts)
{
    const int G = g->part_G, K = g->part_K;
    if (G == 0) { set_error("rwr_part_rank: rwr_part_begin has not been called"); return RWR_E_INVALID; }
    if (top_n < 1 || top_n > rank_select_max_k()) {
        set_error("rwr_part_rank: top_n must be 1..%d", rank_select_max_k());
        return RWR_E_UNSUPPORTED;
    }
    hipStream_t s = g->stream;
    // only the owner of a seed's row has its raw LIKE links (exclusion list): rank those, report -1 for the rest
    std::vector<int32_t> own((size_t)G, -1), slot_k((size_t)G, -1);
    for (int k = 0; k < K; ++k)
        if (g->part_seeds[k] >= g->part_lo && g->part_seeds[k] < g->part_hi) { own[k] = g->part_seeds[k]; slot_k[k] = k; }
    DevBuf<int32_t> d_own;
    RWR_TRY(d_own.alloc(G));
    RWR_TRY(g->d_slot_k.ensure(G));
    RWR_HIP(hipMemcpy(d_own.p, own.data(), G * sizeof(int32_t), hipMemcpyHostToDevice));
    RWR_HIP(hipMemcpy(g->d_slot_k.p, slot_k.data(), G * sizeof(int32_t), hipMemcpyHostToDevice));
    const size_t out_elems = (size_t)G * top_n;
    RWR_TRY(g->d_out_id.ensure(out_elems));
    RWR_TRY(g->d_out_score.ensure(out_elems));
    RWR_TRY(g->d_counts.ensure(G));
    RWR_HIP(hipMemsetAsync(g->d_out_id.p, 0, out_elems * sizeof(int64_t), s));
    RWR_HIP(hipMemsetAsync(g->d_out_score.p, 0, out_elems * sizeof(double), s));
    RWR_HIP(hipMemsetAsync(g->d_counts.p, 0, G * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_exclude, dim3((unsigned)G), dim3(64), 0, s, g->n, 1, G, g->rowptr.p, g->dst.p, g->etype.p, x, d_own.p);
    RWR_TRY(rank_group_select(g, G, 1, g->d_slot_k.p, top_n, x, d_own.p, s));
    std::vector<int32_t> hc((size_t)G);
    RWR_HIP(hipMemcpyAsync(hc.data(), g->d_counts.p, G * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(ids, g->d_out_id.p, (size_t)K * top_n * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemcpyAsync(scores, g->d_out_score.p, (size_t)K * top_n * sizeof(double), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < K; ++k) counts[k] = own[k] >= 0 ? hc[k] : -1;
    return RWR_OK;
}

}  // namespace rwr
