// EXACT mode, the seed's own row, as a PARALLEL reduction that is still bit-identical to the reference's
// strictly sequential chain.
//
// Model.deliverRanks (Model.cs:76-100) adds into nextRank[seed], for i ascending: i's links into the seed
// (Model.cs:85-88), then the restart addend  rank[i] - rw  (Model.cs:91-93) or  rank[i]  (dangling,
// Model.cs:96-97).  That is an n-term fp64 chain  s <- fl(s + a)  per seed; folded literally it runs at the
// dependent-add latency (k_seed_chain_roles in iterate.hip: ~20 cycles per row, 50 ms per step on a 6 M-node
// graph), which is hidden behind the SpMM of a large batch but IS the run time of a single-seed call.
//
// Binade scan.  All addends are >= 0 (0 <= d <= 1, ranks >= 0), so s only grows.  While s stays inside one
// binade [2^e, 2^(e+1)), ulp u = 2^(e-52) is constant and s = m*u with an integer m in [2^52, 2^53); then
//     fl(s + a) = (m + r(a/u)) * u,
// r = round to nearest integer, an exact half going to the side that makes m + r even.  Inside a binade the
// chain is therefore an INTEGER sum, except that a half-way addend's increment depends on the parity of the
// running m: every addend is a function  parity -> increment,  kept as the pair (d0, d1).  These functions
// compose associatively (PF / pf_compose below), so a block of rows reduces in parallel, in any tree shape, to
// one (d0, d1); the carry then needs O(1) per block:  m' = m + d[m & 1],  valid iff m' < 2^53 (monotone, so
// every prefix stayed in the binade too).  A block whose end leaves the binade -- or whose binade was
// mispredicted -- is redone by one wave with a lane-per-row scan and real fp64 adds at the crossing rows.
// tests/binade_scan_model.py is an executable model of exactly this arithmetic (checked against sequential
// sums in tests/test_binade_scan.py); the GPU parity tests compare the kernel with the oracle bit for bit.
//
// Pipeline per power-iteration step (all on the main stream, no spinning, no atomics besides the bitmap OR):
//   k_cs_block<G,false>  plain fp64 sum of each block's addends (any order)            -> approx
//   k_cs_plan            running approx prefix -> predicted biased exponent per block    -> e_pred
//   k_cs_block<G,true>   (d0, d1) of each block under its predicted binade             -> d0, d1
//   k_cs_carry<G>        one wave per seed, lane = block: the block functions of a 64-block window are themselves
//                        composed by a wave scan; the window is accepted up to the first block that leaves the
//                        binade (or whose binade was mispredicted), that block is redone row by row by the same
//                        wave, and the carry goes on behind it; writes Y[seed]
// once per batch: k_cs_links (where each block's share of the seed's in-link list starts).
// Per-(seed, block) cells are laid out [slot][block] so that a window is one coalesced load.
#include "engine.h"
#include "pf.h"

#include <cstdlib>

namespace rwr {

// rank-matrix elements per block (rows x G).  A single seed gets 1024-row blocks: a block that crosses a binade is
// redone row by row by ONE wave, and that serial redo, not the parallel passes, is what a single-seed step waits for.
// Wide tiles (G >= 16) get 16384-element blocks, walked by k_cs_block as four 4096-element pieces: the per-seed carry is a
// serial walk over the blocks, and at 32 seeds per row a 4096-element block is only 128 rows.
template <int G> struct CsGeom {
    static constexpr int EP = (G == 1) ? 1024 : 4096;   // elements per LDS piece of k_cs_block
    static constexpr int P = (G >= 16) ? 4 : 1;         // pieces per block
    static constexpr int E = EP * P;                    // elements per block = per (d0, d1) cell
    static constexpr int R = EP / 256;                  // rows per thread run inside a piece
    static constexpr int CH = E / G;                    // rows per block
};
__device__ __forceinline__ int cs_pad(int q) { return q + (q >> 4); }
// The addend of link l of the seed's in-list (Model.cs:84,87 for target == seed): precomputed by k_seed_terms (termp), or --
// value-free graphs, evoff == nullptr -- gathered here from the z matrix, whose entries ARE those addends (zt = the seed's
// column of its tile).
__device__ __forceinline__ double cs_term(const double *termp, const double *zt, const int32_t *srcp, int32_t l, int G)
{
    return zt ? zt[(size_t)srcp[l] * G] : termp[l];
}

// lnk[slot][c] = number of the seed's in-links whose source row is < c * CH  (c = 0..nchunks)
__global__ __launch_bounds__(256) void k_cs_links(int nchunks, int CH, const int64_t *__restrict__ in_ptr,
                                                  const int32_t *__restrict__ in_src,
                                                  const int32_t *__restrict__ seeds, int32_t *__restrict__ lnk)
{
    const int slot = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nchunks) return;
    const int32_t s = seeds[slot];
    int32_t out = 0;
    if (s >= 0) {
        const int64_t p0 = in_ptr[s];
        const int32_t deg = (int32_t)(in_ptr[s + 1] - p0);
        const int64_t target = (int64_t)c * CH;
        int32_t lo = 0, hi = deg;
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)in_src[p0 + mid] < target) lo = mid + 1; else hi = mid;
        }
        out = lo;
    }
    lnk[(size_t)slot * (size_t)(nchunks + 1) + c] = out;
}

// One workgroup per (block of E elements, tile), walked as P pieces of EP elements.  A piece's restart addends are
// staged in LDS (coalesced loads); thread (rl, k) then walks its run of R consecutive rows of seed k in row order, taking
// a row's links into the seed before the row's restart addend; the runs are combined in row order by an LDS tree and the
// pieces are composed in order.
template <int G, bool FUNCS>
__global__ __launch_bounds__(256) void k_cs_block(int32_t n, int nchunks, const uint8_t *__restrict__ dangling,
                                                  const double *__restrict__ X, const int32_t *__restrict__ seeds,
                                                  double c1, const int64_t *__restrict__ in_ptr,
                                                  const int32_t *__restrict__ in_src,
                                                  const int64_t *__restrict__ evoff, const double *__restrict__ evterm,
                                                  const int32_t *__restrict__ lnk, const int32_t *__restrict__ e_pred,
                                                  double *__restrict__ approx, long long *__restrict__ od0,
                                                  long long *__restrict__ od1)
{
    constexpr int EP = CsGeom<G>::EP, NP = CsGeom<G>::P, CS_R = CsGeom<G>::R;
    constexpr int CHP = EP / G, CH = CsGeom<G>::CH, RL = 256 / G;     // rows per piece / per block; row lanes
    static_assert(CHP == RL * CS_R, "run length");
    __shared__ double a_s[EP + EP / 16];
    __shared__ long long r0[256], r1[256];
    const int c = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
    const double *x = X + (size_t)tile * (size_t)n * G;
    const int64_t total = (int64_t)n * G;
    const int k = tid % G, rl = tid / G;
    const int slot = tile * G + k;
    const int32_t s = seeds[slot];
    // this block's share of the seed's in-link list (sorted by source row)
    int32_t l1 = 0, a0 = 0;
    const int32_t *srcp = in_src;
    const double *termp = evterm, *zt = nullptr;
    if (s >= 0) {
        const int32_t *lk = lnk + (size_t)slot * (size_t)(nchunks + 1) + c;
        a0 = lk[0];
        l1 = lk[1];
        if (l1 > a0) {
            srcp = in_src + in_ptr[s];
            if (evoff) termp = evterm + evoff[slot]; else zt = evterm + (size_t)tile * (size_t)n * G + k;
        }
    }
    const size_t oidx = (size_t)slot * nchunks + c;
    const int eb = FUNCS ? e_pred[oidx] : 0;
    PF ftot{0, 0};
    double atot = 0.0;
    for (int piece = 0; piece < NP; ++piece) {
        const int64_t row0 = (int64_t)c * CH + (int64_t)piece * CHP, el0 = row0 * G;
        if (piece) __syncthreads();                          // the previous piece's LDS reads are done
#pragma unroll
        for (int j = 0; j < EP / 256; ++j) {
            const int q = tid + 256 * j;
            const int64_t el = el0 + q;
            double a = 0.0;
            if (el < total) {
                const double xv = x[el];
                const double rw = c1 * xv;                       // Model.cs:84
                a = dangling[el / G] ? xv : (xv - rw);           // Model.cs:97 / :91
            }
            a_s[cs_pad(q)] = a;
        }
        __syncthreads();
        const int64_t ra = row0 + (int64_t)rl * CS_R;
        int32_t l = l1;
        if (l1 > a0) {                                           // first link whose source row is >= this run's first row
            int32_t lo = a0, hi = l1;
            while (lo < hi) {
                const int32_t mid = lo + ((hi - lo) >> 1);
                if ((int64_t)srcp[mid] < ra) lo = mid + 1; else hi = mid;
            }
            l = lo;
        }
        if constexpr (FUNCS) {
            PF f{0, 0};
#pragma unroll
            for (int u = 0; u < CS_R; ++u) {
                const int64_t row = ra + u;
                while (l < l1 && (int64_t)srcp[l] == row) { f = pf_compose(f, pf_of(cs_term(termp, zt, srcp, l, G), eb)); ++l; }   // Model.cs:85-88
                f = pf_compose(f, pf_of(a_s[cs_pad((rl * CS_R + u) * G + k)], eb));                        // Model.cs:91-93,96-97
            }
            r0[tid] = f.d0;
            r1[tid] = f.d1;
            __syncthreads();
            for (int st = 1; st < RL; st <<= 1) {
                if ((rl & (2 * st - 1)) == 0 && rl + st < RL) {
                    PF a{r0[tid], r1[tid]}, b{r0[tid + st * G], r1[tid + st * G]};
                    const PF r = pf_compose(a, b);
                    r0[tid] = r.d0;
                    r1[tid] = r.d1;
                }
                __syncthreads();
            }
            if (rl == 0) ftot = pf_compose(ftot, PF{r0[tid], r1[tid]});   // pieces in row order
        } else {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < CS_R; ++u) {
                const int64_t row = ra + u;
                while (l < l1 && (int64_t)srcp[l] == row) { acc += cs_term(termp, zt, srcp, l, G); ++l; }
                acc += a_s[cs_pad((rl * CS_R + u) * G + k)];
            }
            double *racc = reinterpret_cast<double *>(r0);
            racc[tid] = acc;
            __syncthreads();
            for (int st = 1; st < RL; st <<= 1) {
                if ((rl & (2 * st - 1)) == 0 && rl + st < RL) racc[tid] += racc[tid + st * G];
                __syncthreads();
            }
            if (rl == 0) atot += racc[tid];
        }
    }
    if (rl == 0) {
        if constexpr (FUNCS) { od0[oidx] = ftot.d0; od1[oidx] = ftot.d1; }
        else approx[oidx] = atot;
    }
}

// predicted biased exponent of the running sum at the start of every block (from the approximate block sums);
// one wave per seed, 64 blocks per step: shuffle scan of the block sums (any association will do for a prediction)
__global__ __launch_bounds__(64) void k_cs_plan(int nchunks, const int32_t *__restrict__ seeds,
                                                const double *__restrict__ approx, int32_t *__restrict__ e_pred)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (seeds[slot] < 0) return;
    const size_t base = (size_t)slot * nchunks;
    double carry = 0.0;
    for (int c0 = 0; c0 < nchunks; c0 += WAVE) {
        const int c = c0 + lane;
        const double ap = c < nchunks ? approx[base + c] : 0.0;
        double incl = ap;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const double o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
        }
        double excl = __shfl_up(incl, 1, WAVE);
        if (lane == 0) excl = 0.0;
        const double pre = carry + excl;
        if (c < nchunks) e_pred[base + c] = (int32_t)(((unsigned long long)__double_as_longlong(pre) >> 52) & 0x7ff);
        carry += __shfl(incl, WAVE - 1, WAVE);
    }
}

__device__ __forceinline__ double cs_from_m(int eb, long long M)
{
    return __longlong_as_double((long long)(((unsigned long long)eb << 52) | ((unsigned long long)M - CS_HID)));
}

// One wave redoes block c of seed kk exactly.  The block's CH rows are dealt to the lanes in runs of R = CH / 64
// consecutive rows.  Under the CURRENT binade every lane composes its run's parity functions (a row's links into the
// seed first, then its restart addend), one wave scan composes the runs, and the block is accepted up to the first
// run that leaves the binade; that run's adds are then done in real fp64, row by row, and the lanes behind it
// start over under the new binade.  While s is zero or subnormal the first run holding a non-zero addend is added
// in real fp64 the same way.  Returns the new s (wave-uniform).
template <int G>
__device__ double cs_redo_block(double s, int32_t n, int nchunks, int c, int tile, int kk,
                                const uint8_t *__restrict__ dangling, const double *__restrict__ X,
                                const int32_t *__restrict__ seeds, double c1, const int64_t *__restrict__ in_ptr,
                                const int32_t *__restrict__ in_src, const int64_t *__restrict__ evoff,
                                const double *__restrict__ evterm, const int32_t *__restrict__ lnk)
{
    constexpr int CH = CsGeom<G>::CH, R = CH / WAVE;
    static_assert(R >= 1 && R * WAVE == CH, "rows per lane");
    const int lane = threadIdx.x;
    const int slot = tile * G + kk;
    const int32_t sd = seeds[slot];
    const int32_t *lk = lnk + (size_t)slot * (size_t)(nchunks + 1) + c;
    const int32_t a0 = lk[0], a1 = lk[1];
    const int32_t *srcp = in_src + in_ptr[sd];
    const double *termp = evoff ? evterm + evoff[slot] : nullptr;
    const double *zt = evoff ? nullptr : evterm + (size_t)tile * (size_t)n * G + kk;
    const double *xk = X + (size_t)tile * (size_t)n * G + kk;
    // the block's restart addends, fetched in one burst (16 loads in flight per lane) and parked in LDS (run-major reads
    // are conflict-free with one pad slot per 64)
    __shared__ double red_a[CH + CH / 64 + 1];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool nzl = false;
#pragma unroll 16
    for (int i = lane; i < CH; i += WAVE) {
        const int64_t row = (int64_t)c * CH + i;
        double a = 0.0;
        if (row < n) {
            const double xv = xk[(size_t)row * G];
            const double rw = c1 * xv;                         // Model.cs:84
            a = dangling[row] ? xv : (xv - rw);                // Model.cs:97 / :91
        }
        red_a[i + (i >> 6)] = a;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t row0 = (int64_t)c * CH + (int64_t)lane * R;   // first row of this lane's run
    int32_t lb = a1;                                            // first link whose source row is >= row0
    if (a1 > a0) {
        int32_t lo = a0, hi = a1;
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)srcp[mid] < row0) lo = mid + 1; else hi = mid;
        }
        lb = lo;
    }
    const bool haslink = lb < a1 && (int64_t)srcp[lb] < row0 + R;
    // the run's links into the seed, fetched ONCE (row within the run, addend): every pass below walks the run again, and
    // chasing srcp[q] / the term through global memory row by row cost a lane with a link ~1.5 us per row and pass
    constexpr int LK = 4;
    int lu[LK] = {-1, -1, -1, -1};
    double lt[LK] = {0.0, 0.0, 0.0, 0.0};
    bool many = false;                                          // more than LK links in this run: the loops below go to memory
    if (haslink) {
        int32_t q = lb;
#pragma unroll
        for (int k = 0; k < LK; ++k)
            if (q < a1 && (int64_t)srcp[q] < row0 + R) { lu[k] = (int)((int64_t)srcp[q] - row0); lt[k] = cs_term(termp, zt, srcp, q, G); ++q; }
        many = q < a1 && (int64_t)srcp[q] < row0 + R;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) nzl = nzl || red_a[lane * R + u + ((lane * R + u) >> 6)] != 0.0;
    nzl = nzl || haslink;
    int start = 0;
    while (start < WAVE) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(s);
        const int eb = (int)((b >> 52) & 0x7ff);
        int L;
        if (eb == 0 || eb == 0x7ff) {
            // zero / subnormal (or non-finite) running sum: real adds; runs whose addends are all +0.0 are no-ops
            const unsigned long long nzm = __ballot(lane >= start && nzl);
            if (!nzm) break;
            L = __builtin_ctzll(nzm);
        } else {
            PF f{0, 0};
            if (lane >= start) {
                int32_t q = lb;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    if (many) {
                        while (q < a1 && (int64_t)srcp[q] == row0 + u) { f = pf_compose(f, pf_of(cs_term(termp, zt, srcp, q, G), eb)); ++q; }
                    } else if (haslink) {
#pragma unroll
                        for (int k = 0; k < LK; ++k) if (lu[k] == u) f = pf_compose(f, pf_of(lt[k], eb));   // list order (Model.cs:85-88)
                    }
                    f = pf_compose(f, pf_of(red_a[lane * R + u + ((lane * R + u) >> 6)], eb));
                }
            }
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                PF o;
                o.d0 = __shfl_up(f.d0, off, WAVE);
                o.d1 = __shfl_up(f.d1, off, WAVE);
                if (lane >= off) f = pf_compose(o, f);
            }
            const long long m = (long long)((b & CS_FRAC) | CS_HID);
            const long long M = m + ((m & 1) ? f.d1 : f.d0);
            const unsigned long long cross = __ballot(M >= CS_BIG);
            if (!cross) {
                s = cs_from_m(eb, __shfl(M, WAVE - 1, WAVE));
                break;
            }
            L = __builtin_ctzll(cross);
            if (L > 0) s = cs_from_m(eb, __shfl(M, L - 1, WAVE));   // the runs before the crossing (lanes < start hold m itself)
        }
        double t = s;
        if (lane == L) {
            int32_t q = lb;
#pragma unroll
            for (int u = 0; u < R; ++u) {
                if (many) {
                    while (q < a1 && (int64_t)srcp[q] == row0 + u) { t += cs_term(termp, zt, srcp, q, G); ++q; }   // Model.cs:85-88
                } else if (haslink) {
#pragma unroll
                    for (int k = 0; k < LK; ++k) if (lu[k] == u) t += lt[k];
                }
                t += red_a[lane * R + u + ((lane * R + u) >> 6)];                              // Model.cs:91-93,96-97
            }
        }
        s = __shfl(t, L, WAVE);
        start = L + 1;
    }
    return s;
}

template <int G>
__global__ __launch_bounds__(64) void k_cs_carry(int32_t n, int nchunks, const uint8_t *__restrict__ dangling,
                                                 const double *__restrict__ X, double *__restrict__ Y,
                                                 const int32_t *__restrict__ seeds, double c1,
                                                 const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                 const int64_t *__restrict__ evoff, const double *__restrict__ evterm,
                                                 const int32_t *__restrict__ lnk, const double *__restrict__ approx,
                                                 const int32_t *__restrict__ e_pred, const long long *__restrict__ d0,
                                                 const long long *__restrict__ d1, uint32_t *__restrict__ nz_out,
                                                 unsigned long long *__restrict__ redo_count, int direct,
                                                 double *__restrict__ zout = nullptr, const double *__restrict__ w_src = nullptr)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int tile = slot / G, k = slot % G;
    const int32_t sd = seeds[slot];
    if (sd < 0) return;                                   // padding slot (whole wave)
    const size_t base = (size_t)slot * nchunks;
    double s = 0.0;
    unsigned redo = 0;
    if (direct) {
        // a handful of blocks (ego-network sizes): nearly every block crosses a binade, so the parallel passes would be
        // redone anyway -- the wave walks the blocks itself and the step saves three launches
        for (int c = 0; c < nchunks; ++c)
            s = cs_redo_block<G>(s, n, nchunks, c, tile, k, dangling, X, seeds, c1, in_ptr, in_src, evoff, evterm, lnk);
        if (lane == 0) {
            Y[(size_t)tile * (size_t)n * G + (size_t)sd * G + k] = s;
            if (zout) { const double rw = c1 * s; zout[(size_t)tile * (size_t)n * G + (size_t)sd * G + k] = rw * w_src[sd]; }   // Model.cs:84,87
            if (nz_out && s != 0.0)
                atomicOr(&nz_out[(size_t)tile * (((size_t)n + 31) / 32) + ((uint32_t)sd >> 5)], 1u << (sd & 31));
        }
        return;
    }
    // window registers: (block sum, predicted exponent, d0, d1) of block c + lane
    double ap, apn = 0.0;
    int32_t ep, epn = 0;
    long long f0, f1, f0n = 0, f1n = 0;
#define CS_LOAD(C, AP, EP, F0, F1)                                   \
    {                                                                \
        const int cc__ = (C) + lane;                                 \
        const bool v__ = cc__ < nchunks;                             \
        const size_t i__ = base + (v__ ? cc__ : nchunks - 1);        \
        AP = v__ ? approx[i__] : 0.0;                                \
        EP = e_pred[i__];                                            \
        F0 = d0[i__];                                                \
        F1 = d1[i__];                                                \
    }
    int c = 0;
    int done = 0;                                         // blocks [c, c + done) of the window are already carried
    CS_LOAD(0, ap, ep, f0, f1)
    while (c < nchunks) {
        // the next window is requested once, when this one is entered (a redone block does NOT reload the window: the blocks
        // behind it are composed again from the registers under the new sum -- a reload cost a memory round trip per redo)
        if (done == 0 && c + WAVE < nchunks) CS_LOAD(c + WAVE, apn, epn, f0n, f1n)
        const unsigned long long b = (unsigned long long)__double_as_longlong(s);
        const int eb = (int)((b >> 52) & 0x7ff);
        const bool normal = eb != 0 && eb != 0x7ff;
        const bool nz = ap != 0.0 && lane >= done;        // (block sum 0 <=> every addend +0.0: the block is a no-op)
        PF f{0, 0};
        if (nz) { f.d0 = f0; f.d1 = f1; }
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            PF o;
            o.d0 = __shfl_up(f.d0, off, WAVE);
            o.d1 = __shfl_up(f.d1, off, WAVE);
            if (lane >= off) f = pf_compose(o, f);
        }
        const long long m = (long long)((b & CS_FRAC) | CS_HID);
        const long long M = m + ((m & 1) ? f.d1 : f.d0);
        const unsigned long long bad = __ballot(nz && (!normal || ep != eb || M >= CS_BIG));
        if (!bad) {
            if (normal) s = cs_from_m(eb, __shfl(M, WAVE - 1, WAVE));
            c += WAVE;
            done = 0;
            ap = apn; ep = epn; f0 = f0n; f1 = f1n;
            continue;
        }
        const int L = __builtin_ctzll(bad);
        if (L > 0 && normal) s = cs_from_m(eb, __shfl(M, L - 1, WAVE));   // the blocks before it (carried ones are identities)
        s = cs_redo_block<G>(s, n, nchunks, c + L, tile, k, dangling, X, seeds, c1, in_ptr, in_src, evoff, evterm, lnk);
        ++redo;
        done = L + 1;
        if (done >= WAVE || c + done >= nchunks) {        // the window is exhausted
            c += WAVE;
            done = 0;
            ap = apn; ep = epn; f0 = f0n; f1 = f1n;
        }
    }
#undef CS_LOAD
    if (lane == 0) {
        Y[(size_t)tile * (size_t)n * G + (size_t)sd * G + k] = s;
        if (nz_out && s != 0.0)
            atomicOr(&nz_out[(size_t)tile * (((size_t)n + 31) / 32) + ((uint32_t)sd >> 5)], 1u << (sd & 31));
        if (redo_count && redo) atomicAdd(redo_count, (unsigned long long)redo);
    }
}

// ---------------------------------------------------------------------------------------------- single seed (G = 1)
// A single-seed step WAITS for the chain (the SpMV beside it takes 110-180 us on the 0.2-0.6 M-node graphs), and the chain
// waited for the carry's redone blocks: the sum doubles ~20 times on its way from the first rank to d, ~12 of those
// crossings lie in block 0 and the others in one block each, and one wave redoing a 1024-row block costs 8-9 us (three or
// four dependent memory round trips by a lone wave, then two passes of 16 pf_of per lane) -- 100 of the carry's 128 us.
// The parallel pass can do nearly all of it ahead of time, because WHERE the sum leaves its binade is predictable:
//   * block 0 (more generally the first block holding a non-zero addend) starts from exactly +0.0, so its workgroup adds it
//     up sequentially -- literally the reference's loop -- and hands the carry the exact sum behind it         (CS_EXACT);
//   * a block whose approximate incoming and outgoing sums lie in adjacent binades e, e + 1 is SPLIT: the approximate
//     incoming mantissa locates the 4-row run in which the sum crosses; the block hands over the composed function of
//     the runs before it (under e), the run's four addends, and the composed function of the runs behind it (under
//     e + 1).  The carry applies function, four real adds, function -- and CHECKS each part (sum still inside e before
//     the run, inside e + 1 after the four adds and at the block's end).  A row's addend is ~10^10 ulps of the sum while
//     the approximate prefix is good to ~10^3, so the run is the right one except with probability ~10^-7; then, as for
//     any failed check, the block is redone exactly by the carry wave (cs_redo_block)                           (CS_SPLIT);
//   * anything else -- two crossings in a block behind the first, a link into the seed inside the split run, a subnormal
//     incoming sum -- is left to the carry's redo                                                               (CS_REDO).
// The prediction (k_cs_plan) is folded into the block kernel: every workgroup sums the approximate block sums in front of it.
constexpr int CS_PLAIN = 0, CS_SPLIT = 1, CS_EXACT = 2, CS_REDO = 3, CS_SPLIT2 = 4;
#ifdef RWR_EXPERIMENTS
__device__ int cs_carry_dbg = 0;     // RWR_X_CARRY_DBG=1: the carry prints what it did with its blocks
#endif
constexpr int CS_KIND_SHIFT = 12;                       // cs_e cell: predicted exponent | kind << 12
constexpr int CS_SIDE_WORDS = 32;                       // [0..7] first run's addends, [8,9] after (SPLIT2: the middle), [10] exact, [12..19] second run's addends, [20,21] after2
constexpr int CS_MX_LINKS = 8192;                       // links into the seed per 1024-row block that the exact-start block lays out in its scratch row
constexpr int CS_RUN_ADDENDS = 8;                       // a crossing ROW hands over up to 8 addends: its links into the seed (list order) and its restart addend
typedef double v2d_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 6) void k_cs_block1(int32_t n, int nchunks, const uint8_t *__restrict__ dangling,
                                                   const double *__restrict__ X, const int32_t *__restrict__ seeds, double c1,
                                                   const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                   const int64_t *__restrict__ evoff, const double *__restrict__ evterm,
                                                   const int32_t *__restrict__ lnk, double *__restrict__ approx,
                                                   int32_t *__restrict__ ek, long long *__restrict__ od0,
                                                   long long *__restrict__ od1, double *__restrict__ side, int own_sums,
                                                   double *__restrict__ mx)
{
    // own_sums: graphs of a few dozen blocks (ego-network sizes) skip the separate pass of approximate block sums
    // (k_cs_block<1,false>): every workgroup adds up the addends in front of its block itself -- at most a few tens of
    // thousands of rows out of the L2 -- and writes its own block's sum for the carry: one launch less per step
    constexpr int CH = CsGeom<1>::CH, CS_R = CsGeom<1>::R;      // 1024 rows per block, 4 per thread
    __shared__ double a_s[CH + CH / 16];
    __shared__ long long r0[256], r1[256];
    __shared__ long long w0[4], w1[4];
    __shared__ int rstar;
    const int c = blockIdx.x, slot = blockIdx.y, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const double *x = X + (size_t)slot * (size_t)n;
    const int32_t s = seeds[slot];
    const size_t base = (size_t)slot * nchunks, oidx = base + c;
    // this block's share of the seed's in-link list (sorted by source row)
    int32_t l1 = 0, a0 = 0;
    const int32_t *srcp = in_src;
    const double *termp = evterm, *zt = nullptr;
    if (s >= 0) {
        const int32_t *lk = lnk + (size_t)slot * (size_t)(nchunks + 1) + c;
        a0 = lk[0];
        l1 = lk[1];
        if (l1 > 0) {
            srcp = in_src + in_ptr[s];
            if (evoff) termp = evterm + evoff[slot]; else zt = evterm + (size_t)slot * (size_t)n;
        }
    }
    // the approximate sum in front of this block (any association will do for a prediction)
    const int64_t row0 = (int64_t)c * CH;
    double pa = 0.0, apl = 0.0;
    if (own_sums) {
        for (int64_t i = tid; i < row0; i += 256) {
            const double xv = x[i];
            const double rw = c1 * xv;
            pa += dangling[i] ? xv : (xv - rw);
        }
        for (int32_t l = tid; l < a0; l += 256) pa += cs_term(termp, zt, srcp, l, 1);
        for (int32_t l = a0 + tid; l < l1; l += 256) apl += cs_term(termp, zt, srcp, l, 1);
    } else {
        for (int i = tid; i < c; i += 256) pa += approx[base + i];
    }
    // this block's addends, staged in LDS
#pragma unroll
    for (int j = 0; j < CH / 256; ++j) {
        const int q = tid + 256 * j;
        const int64_t row = row0 + q;
        double a = 0.0;
        if (row < n) {
            const double xv = x[row];
            const double rw = c1 * xv;                       // Model.cs:84
            a = dangling[row] ? xv : (xv - rw);              // Model.cs:97 / :91
        }
        a_s[cs_pad(q)] = a;
        apl += own_sums ? a : 0.0;
    }
    double *racc = reinterpret_cast<double *>(r0), *racc2 = reinterpret_cast<double *>(r1);
    racc[tid] = pa;
    racc2[tid] = apl;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { racc[tid] += racc[tid + st]; racc2[tid] += racc2[tid + st]; }
        __syncthreads();
    }
#ifdef RWR_EXPERIMENTS
    const unsigned long long tq0 = __builtin_amdgcn_s_memrealtime();
#endif
    const double pre = racc[0];
    const double ap = own_sums ? racc2[0] : approx[oidx];
    __syncthreads();                                         // (r0 / r1 are reused below)
    if (own_sums && tid == 0) approx[oidx] = ap;             // (the carry skips blocks whose sum is +0.0)
    double *sd = side + oidx * CS_SIDE_WORDS;
    const int epre = (int)(((unsigned long long)__double_as_longlong(pre) >> 52) & 0x7ff);
    if (ap == 0.0) {                                         // every addend +0.0: a no-op for the carry
        if (tid == 0) { ek[oidx] = epre | (CS_PLAIN << CS_KIND_SHIFT); od0[oidx] = 0; od1[oidx] = 0; }
        return;
    }
    if (pre == 0.0) {
        // nothing non-zero in front: the sum enters as exactly +0.0 and one thread adds the block up in the reference's order.
        // The block's links into the seed -- an ego has in-links from most of its network: up to one per row -- are fetched by
        // all threads first (row within the block, addend) and parked in LDS: chased through global memory by the one adding
        // thread, each link cost two dependent round trips (~3 us: 60 us per step for 20 links, milliseconds for an ego).
        // The block's addend sequence M -- per row its links into the seed (list order), then its restart addend -- is laid out in
        // parallel in a global scratch row (link q of source row r_q: behind the r_q restart addends of the rows in front of
        // its row and the q links in front of it; row r's restart addend: behind r rows and the links of the rows up to and
        // including r), then ONE WAVE folds it: coalesced loads a line ahead, the adds strictly in order through an LDS line.
        __shared__ double pbx[2][WAVE];
        const int32_t nl = l1 - a0;
        // (a block with a handful of links -- any seed that is not an ego -- is added up by one thread straight from LDS: no
        //  global round trips at all, 8 us; the links wait in a small LDS table)
        constexpr int FEW = 64;
        __shared__ double few_term[FEW];
        __shared__ unsigned short few_row[FEW];
        if (nl <= FEW) {
            for (int32_t q = tid; q < nl; q += 256) {
                few_row[q] = (unsigned short)((int64_t)srcp[a0 + q] - row0);
                few_term[q] = cs_term(termp, zt, srcp, a0 + q, 1);
            }
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                int32_t l = 0;
                for (int u0 = 0; u0 < CH; u0 += 16) {          // 16 rows per batch: one LDS round trip, then the dependent adds
                    double v[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) v[k] = a_s[cs_pad(u0 + k)];
                    if (l < nl && (int)few_row[l] < u0 + 16) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            while (l < nl && (int)few_row[l] == u0 + k) { t += few_term[l]; ++l; }     // Model.cs:85-88
                            t += v[k];                                                                 // Model.cs:91-93,96-97
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) t += v[k];
                    }
                }
                sd[10] = t;
                ek[oidx] = CS_EXACT << CS_KIND_SHIFT;
                od0[oidx] = 0;
                od1[oidx] = 0;
            }
            return;
        }
        const bool staged = nl <= CS_MX_LINKS;
        double *Mg = mx + (size_t)slot * (size_t)(CH + CS_MX_LINKS);
        if (staged) {
            for (int32_t q = tid; q < nl; q += 256) {
                const int r = (int)((int64_t)srcp[a0 + q] - row0);
                Mg[r + q] = cs_term(termp, zt, srcp, a0 + q, 1);
            }
#pragma unroll
            for (int j = 0; j < CH / 256; ++j) {
                const int r = tid + 256 * j;
                int32_t lo = a0, hi = l1;
                while (lo < hi) {
                    const int32_t mid = lo + ((hi - lo) >> 1);
                    if ((int64_t)srcp[mid] <= row0 + r) lo = mid + 1; else hi = mid;
                }
                Mg[r + (lo - a0)] = a_s[cs_pad(r)];
            }
            __threadfence_block();
        }
        __syncthreads();
        if (wave == 0) {
            double t = 0.0;
            if (staged) {
                const int mlen = CH + nl;
                int buf = 0;
                // (four lines of 64 addends in flight: a line's 64 dependent adds take ~0.4 us, a global round trip more)
                double pre4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) pre4[d] = d * WAVE + lane < mlen ? Mg[d * WAVE + lane] : 0.0;
                for (int i0 = 0; i0 < mlen; i0 += 4 * WAVE) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int at = i0 + d * WAVE;
                        if (at >= mlen) break;                                   // (wave-uniform)
                        const double cur = pre4[d];
                        const int nx = at + 4 * WAVE + lane;
                        pre4[d] = nx < mlen ? Mg[nx] : 0.0;
                        pbx[buf][lane] = cur;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (mlen - at >= WAVE) {
#pragma unroll 16                                                                  // (16 LDS reads in flight: more would cost the whole kernel registers)
                            for (int q = 0; q < WAVE; ++q) t += pbx[buf][q];     // Model.cs:85-93,96-97, in the reference's order
                        } else {
                            const int left = mlen - at;
                            for (int q = 0; q < left; ++q) t += pbx[buf][q];
                        }
                        buf ^= 1;
                    }
                }
            } else if (lane == 0) {
                // (more than CS_MX_LINKS links into the seed from one block of 1024 rows: walked through global memory)
                int32_t l = a0;
                for (int u = 0; u < CH; ++u) {
                    while (l < l1 && (int64_t)srcp[l] == row0 + u) { t += cs_term(termp, zt, srcp, l, 1); ++l; }
                    t += a_s[cs_pad(u)];
                }
            }
          if (lane == 0) {
#ifdef RWR_EXPERIMENTS
            if (cs_carry_dbg == 3) printf("  block1 c=%d exact-start: %d links staged=%d, sequential part %.1f us\n", c, nl, (int)staged, (double)(__builtin_amdgcn_s_memrealtime() - tq0) * 0.01);
#endif
            sd[10] = t;
            ek[oidx] = CS_EXACT << CS_KIND_SHIFT;
            od0[oidx] = 0;
            od1[oidx] = 0;
          }
        }
        return;
    }
    const double post = pre + ap;
    const int epost = (int)(((unsigned long long)__double_as_longlong(post) >> 52) & 0x7ff);
    if (epre == 0 || epost >= 0x7ff || epost > epre + 2) {
#ifdef RWR_EXPERIMENTS
        if (cs_carry_dbg && tid == 0) printf("  block %d: redo kind, pre %.17g (e %d) ap %.17g post e %d\n", c, pre, epre, ap, epost);
#endif
        if (tid == 0) { ek[oidx] = epre | (CS_REDO << CS_KIND_SHIFT); od0[oidx] = 0; od1[oidx] = 0; }
        return;
    }
    // first link whose source row is >= this thread's run
    const int64_t ra = row0 + (int64_t)tid * CS_R;
    int32_t lb = l1;
    if (l1 > a0) {
        int32_t lo = a0, hi = l1;
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)srcp[mid] < ra) lo = mid + 1; else hi = mid;
        }
        lb = lo;
    }
    const bool haslink = lb < l1 && (int64_t)srcp[lb] < ra + CS_R;
    // the rows u >= ufrom of this thread's run as one function under binade eb (ufrom > 0: the rows behind a crossing row)
    auto run_pf_from = [&](int eb, int ufrom) {
        PF f{0, 0};
        int32_t l = lb;
#pragma unroll
        for (int u = 0; u < CS_R; ++u) {
            if (haslink)
                while (l < l1 && (int64_t)srcp[l] == ra + u) {                                                  // Model.cs:85-88
                    if (u >= ufrom) f = pf_compose(f, pf_of(cs_term(termp, zt, srcp, l, 1), eb));
                    ++l;
                }
            if (u >= ufrom) f = pf_compose(f, pf_of(a_s[cs_pad(tid * CS_R + u)], eb));                         // Model.cs:91-93,96-97
        }
        return f;
    };
    auto run_pf = [&](int eb) {
        PF f{0, 0};
        int32_t l = lb;
#pragma unroll
        for (int u = 0; u < CS_R; ++u) {
            if (haslink)
                while (l < l1 && (int64_t)srcp[l] == ra + u) { f = pf_compose(f, pf_of(cs_term(termp, zt, srcp, l, 1), eb)); ++l; }   // Model.cs:85-88
            f = pf_compose(f, pf_of(a_s[cs_pad(tid * CS_R + u)], eb));                                         // Model.cs:91-93,96-97
        }
        return f;
    };
    // ordered tree over the 256 runs; `keep` masks a run to the identity
    auto reduce = [&](PF f) {
        r0[tid] = f.d0;
        r1[tid] = f.d1;
        __syncthreads();
        for (int st = 1; st < 256; st <<= 1) {
            if ((tid & (2 * st - 1)) == 0) {
                const PF r = pf_compose(PF{r0[tid], r1[tid]}, PF{r0[tid + st], r1[tid + st]});
                r0[tid] = r.d0;
                r1[tid] = r.d1;
            }
            __syncthreads();
        }
        const PF r{r0[0], r1[0]};
        __syncthreads();
        return r;
    };
    const PF f = run_pf(epre);
#ifdef RWR_EXPERIMENTS
    if (cs_carry_dbg == 3 && tid == 0) printf("  block1 c=%d: links %d, lower bound + run function of thread 0: %.1f us\n", c, l1 - a0, (double)(__builtin_amdgcn_s_memrealtime() - tq0) * 0.01);
#endif
    if (epost == epre) {
        const PF tot = reduce(f);
        if (tid == 0) { ek[oidx] = epre | (CS_PLAIN << CS_KIND_SHIFT); od0[oidx] = tot.d0; od1[oidx] = tot.d1; }
        return;
    }
    // split: exclusive / inclusive prefix of the runs (wave scan + the four wave totals), located against the approximate
    // incoming mantissa.  scan() must be called by every thread.
    auto scan = [&](PF fr, PF &inc, PF &exc) {
        inc = fr;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            PF o;
            o.d0 = __shfl_up(inc.d0, off, WAVE);
            o.d1 = __shfl_up(inc.d1, off, WAVE);
            if (lane >= off) inc = pf_compose(o, inc);
        }
        __syncthreads();                                     // (the totals of an earlier scan have been read)
        if (lane == WAVE - 1) { w0[wave] = inc.d0; w1[wave] = inc.d1; }
        __syncthreads();
        PF prew{0, 0};
        for (int w = 0; w < wave; ++w) prew = pf_compose(prew, PF{w0[w], w1[w]});
        exc.d0 = __shfl_up(inc.d0, 1, WAVE);
        exc.d1 = __shfl_up(inc.d1, 1, WAVE);
        if (lane == 0) exc = PF{0, 0};
        exc = pf_compose(prew, exc);
        inc = pf_compose(prew, inc);
    };
    // the ROW in which mantissa mt (approximate) leaves binade eb: first the run (the scan's inclusive / exclusive functions),
    // then, by the thread that owns it, the row inside the run.  Through LDS come back: the function of everything in front of
    // that row, the row's own addend sequence -- its links into the seed in list order, then its restart addend
    // (Model.cs:85-93,96-97; the seed's in-links come from the nodes that hold most of the rank, so the crossing row is more
    // often than not one with such a link, and an ego has several links from one node), padded with +0.0 -- and the row's
    // place in the run.  ufrom: rows of this thread's run that are already behind an earlier crossing.  -1: no such run;
    // over: the row has more than CS_RUN_ADDENDS addends.
    __shared__ long long xb0, xb1;
    __shared__ double xa[CS_RUN_ADDENDS];
    __shared__ int xover, xcrow;
    auto locate = [&](long long mt, const PF &inc, const PF &exc, int eb, int ufrom, PF &front, bool &over, int &crow) -> int {
        if (tid == 0) rstar = -1;
        __syncthreads();
        if (mt + exc.d0 < CS_BIG && mt + inc.d0 >= CS_BIG) {       // (monotone: at most one run)
            rstar = tid;
            // pass 1: which row (functions only); pass 2: that row's addends straight into LDS (no private arrays: a dynamically
            // indexed one would put the whole kernel on scratch memory)
            PF running = exc;
            int32_t l = lb, lrow = lb;
            int found = -1;
            for (int u = 0; u < CS_R && found < 0; ++u) {
                const int32_t lu = l;
                PF rowpf{0, 0};
                if (haslink)
                    while (l < l1 && (int64_t)srcp[l] == ra + u) {
                        if (u >= ufrom) rowpf = pf_compose(rowpf, pf_of(cs_term(termp, zt, srcp, l, 1), eb));
                        ++l;
                    }
                if (u < ufrom) continue;
                rowpf = pf_compose(rowpf, pf_of(a_s[cs_pad(tid * CS_R + u)], eb));
                const PF incu = pf_compose(running, rowpf);
                if (mt + running.d0 < CS_BIG && mt + incu.d0 >= CS_BIG) {
                    found = u;
                    lrow = lu;
                    xb0 = running.d0;
                    xb1 = running.d1;
                }
                running = incu;
            }
            if (found >= 0) {
                int cnt = 0;
                for (int32_t q = lrow; haslink && q < l1 && (int64_t)srcp[q] == ra + found; ++q) {
                    if (cnt < CS_RUN_ADDENDS) xa[cnt] = cs_term(termp, zt, srcp, q, 1);
                    ++cnt;
                }
                if (cnt < CS_RUN_ADDENDS) xa[cnt] = a_s[cs_pad(tid * CS_R + found)];
                ++cnt;
                for (int q = cnt; q < CS_RUN_ADDENDS; ++q) xa[q] = 0.0;
                xover = cnt > CS_RUN_ADDENDS ? 1 : 0;
            }
            if (found < 0) { found = 0; xover = 1; }               // (cannot happen: the run-level test said the crossing is here)
            xcrow = found;
        }
        __syncthreads();
        const int r = rstar;
        front = PF{xb0, xb1};
        over = r >= 0 && xover != 0;
        crow = xcrow;
        return r;
    };
    auto redo_kind = [&](int why) {
#ifdef RWR_EXPERIMENTS
        if (cs_carry_dbg && tid == 0) printf("  block %d: redo kind, reason %d (1 first crossing row has too many addends, 2 sum behind it not in e+1, 3 second crossing row has too many); pre %.6g ap %.6g\n", c, why, pre, ap);
#endif
        (void)why;
        if (tid == 0) { ek[oidx] = epre | (CS_REDO << CS_KIND_SHIFT); od0[oidx] = 0; od1[oidx] = 0; }
    };
    PF inc, exc;
    scan(f, inc, exc);
    const long long mt = (long long)(((unsigned long long)__double_as_longlong(pre) & CS_FRAC) | CS_HID);
    PF before;
    bool over1 = false;
    int crow1 = 0;
    const int rs = locate(mt, inc, exc, epre, 0, before, over1, crow1);
    if (rs < 0) {
        // the approximate sums say the binade is left, the scan does not: hand over the whole block under e; the carry checks
        const PF tot = reduce(f);
        if (tid == 0) { ek[oidx] = epre | (CS_PLAIN << CS_KIND_SHIFT); od0[oidx] = tot.d0; od1[oidx] = tot.d1; }
        return;
    }
    if (over1) { redo_kind(1); return; }
    double run1[CS_RUN_ADDENDS];
#pragma unroll
    for (int q = 0; q < CS_RUN_ADDENDS; ++q) run1[q] = xa[q];
    PF g{0, 0};
    if (tid > rs) g = run_pf(epre + 1);
    else if (tid == rs) g = run_pf_from(epre + 1, crow1 + 1);    // (the rows of the crossing run behind the crossing row)
    auto emit_split = [&](const PF &after) {
        if (tid == 0) {
            ek[oidx] = epre | (CS_SPLIT << CS_KIND_SHIFT);
            od0[oidx] = before.d0;
            od1[oidx] = before.d1;
#pragma unroll
            for (int q = 0; q < CS_RUN_ADDENDS; ++q) sd[q] = run1[q];
            sd[8] = __longlong_as_double(after.d0);
            sd[9] = __longlong_as_double(after.d1);
        }
    };
    if (epost == epre + 1) {
        const PF after = reduce(g);                          // (its barriers also order the reads of xa above before any rewrite)
        emit_split(after);
        return;
    }
    // two crossings (the sum more than doubles inside the block: the ranks are still concentrated): the second run is located
    // the same way, under e + 1, from the approximate sum behind the first run
    double s1 = cs_from_m(epre, mt + before.d0);
#pragma unroll
    for (int q = 0; q < CS_RUN_ADDENDS; ++q) s1 += run1[q];
    const unsigned long long s1b = (unsigned long long)__double_as_longlong(s1);
    if ((int)((s1b >> 52) & 0x7ff) != epre + 1 || epre + 2 >= 0x7ff) { redo_kind(2); return; }   // (uniform: s1 is the same in every thread)
    PF inc2, exc2;
    scan(g, inc2, exc2);
    const long long mt1 = (long long)((s1b & CS_FRAC) | CS_HID);
    PF middle;
    bool over2 = false;
    int crow2 = 0;
    const int rs2 = locate(mt1, inc2, exc2, epre + 1, tid == rs ? crow1 + 1 : 0, middle, over2, crow2);
    if (rs2 < 0) {
        // no second crossing in sight after all: a single split, checked by the carry
        const PF after = reduce(g);
        emit_split(after);
        return;
    }
    if (over2) { redo_kind(3); return; }
    double run2[CS_RUN_ADDENDS];
#pragma unroll
    for (int q = 0; q < CS_RUN_ADDENDS; ++q) run2[q] = xa[q];
    PF g2{0, 0};
    if (tid > rs2) g2 = run_pf(epre + 2);
    else if (tid == rs2) g2 = run_pf_from(epre + 2, crow2 + 1);
    const PF after2 = reduce(g2);
    if (tid == 0) {
        ek[oidx] = epre | (CS_SPLIT2 << CS_KIND_SHIFT);
        od0[oidx] = before.d0;
        od1[oidx] = before.d1;
#pragma unroll
        for (int q = 0; q < CS_RUN_ADDENDS; ++q) { sd[q] = run1[q]; sd[12 + q] = run2[q]; }
        sd[8] = __longlong_as_double(middle.d0);
        sd[9] = __longlong_as_double(middle.d1);
        sd[20] = __longlong_as_double(after2.d0);
        sd[21] = __longlong_as_double(after2.d1);
    }
}

// The single-seed carry: k_cs_carry<1>'s window walk, with the split and exact blocks applied from the window's registers;
// only a failed check (or a CS_REDO block) costs a redone block.
__global__ __launch_bounds__(64) void k_cs_carry1(int32_t n, int nchunks, const uint8_t *__restrict__ dangling,
                                                  const double *__restrict__ X, double *__restrict__ Y,
                                                  const int32_t *__restrict__ seeds, double c1,
                                                  const int64_t *__restrict__ in_ptr, const int32_t *__restrict__ in_src,
                                                  const int64_t *__restrict__ evoff, const double *__restrict__ evterm,
                                                  const int32_t *__restrict__ lnk, const double *__restrict__ approx,
                                                  const int32_t *__restrict__ ek, const long long *__restrict__ d0,
                                                  const long long *__restrict__ d1, const double *__restrict__ side,
                                                  uint32_t *__restrict__ nz_out, unsigned long long *__restrict__ redo_count,
                                                  double *__restrict__ zout, const double *__restrict__ w_src)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int32_t sd = seeds[slot];
    if (sd < 0) return;                                   // padding slot (whole wave)
    const size_t base = (size_t)slot * nchunks;
    double s = 0.0;
    unsigned redo = 0;
#ifdef RWR_EXPERIMENTS
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_redo = 0;
    unsigned n_split = 0, n_exact = 0, redo_kind[4] = {0, 0, 0, 0};
#endif
    struct Win {
        double ap;
        int32_t ek;
        long long f0, f1;
        v2d_t a01, a23, a45, a67, af;                     // side words 0..9: first run's addends, the function behind it
        double ex;                                        // word 10: exact sum (CS_EXACT)
        v2d_t b01, b23, b45, b67, bf;                     // words 12..21: second run's addends, the function behind it (CS_SPLIT2)
    };
    auto load = [&](int C, Win &w) {
        const int cc = C + lane;
        const bool v = cc < nchunks;
        const size_t i = base + (v ? cc : nchunks - 1);
        w.ap = v ? approx[i] : 0.0;
        w.ek = ek[i];
        w.f0 = d0[i];
        w.f1 = d1[i];
        const v2d_t *sp = reinterpret_cast<const v2d_t *>(side + i * CS_SIDE_WORDS);
        w.a01 = sp[0]; w.a23 = sp[1]; w.a45 = sp[2]; w.a67 = sp[3]; w.af = sp[4];
        w.ex = side[i * CS_SIDE_WORDS + 10];
        w.b01 = sp[6]; w.b23 = sp[7]; w.b45 = sp[8]; w.b67 = sp[9]; w.bf = sp[10];
    };
    Win cur, nxt;
    int c = 0;
    int done = 0;                                         // blocks [c, c + done) of the window are already carried
    load(0, cur);
    nxt = cur;
    while (c < nchunks) {
        if (done == 0 && c + WAVE < nchunks) load(c + WAVE, nxt);
        unsigned long long b = (unsigned long long)__double_as_longlong(s);
        int eb = (int)((b >> 52) & 0x7ff);
        bool normal = eb != 0 && eb != 0x7ff;
        const int kind = cur.ek >> CS_KIND_SHIFT, ep = cur.ek & ((1 << CS_KIND_SHIFT) - 1);
        const bool nz = cur.ap != 0.0 && lane >= done;    // (block sum 0 <=> every addend +0.0: the block is a no-op)
        PF f{0, 0};
        if (nz && kind == CS_PLAIN) { f.d0 = cur.f0; f.d1 = cur.f1; }
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            PF o;
            o.d0 = __shfl_up(f.d0, off, WAVE);
            o.d1 = __shfl_up(f.d1, off, WAVE);
            if (lane >= off) f = pf_compose(o, f);
        }
        long long m = (long long)((b & CS_FRAC) | CS_HID);
        const long long M = m + ((m & 1) ? f.d1 : f.d0);
        const unsigned long long bad = __ballot(nz && (kind != CS_PLAIN || !normal || ep != eb || M >= CS_BIG));
        if (!bad) {
            if (normal) s = cs_from_m(eb, __shfl(M, WAVE - 1, WAVE));
            c += WAVE;
            done = 0;
            cur = nxt;
            continue;
        }
        const int L = __builtin_ctzll(bad);
        if (L > 0 && normal) s = cs_from_m(eb, __shfl(M, L - 1, WAVE));   // the blocks before it (carried ones are identities)
        // block c + L: its cells, wave-uniform
        const int kL = __shfl(kind, L, WAVE), eL = __shfl(ep, L, WAVE);
        b = (unsigned long long)__double_as_longlong(s);
        eb = (int)((b >> 52) & 0x7ff);
        normal = eb != 0 && eb != 0x7ff;
        bool handled = false;
        // the run's addends (uniform copies of lane L's registers), added in order with real fp64 adds; padding is +0.0
        auto add_run = [&](double t, const v2d_t &p01, const v2d_t &p23, const v2d_t &p45, const v2d_t &p67) {
            t += __shfl(p01.x, L, WAVE); t += __shfl(p01.y, L, WAVE);
            t += __shfl(p23.x, L, WAVE); t += __shfl(p23.y, L, WAVE);
            t += __shfl(p45.x, L, WAVE); t += __shfl(p45.y, L, WAVE);
            t += __shfl(p67.x, L, WAVE); t += __shfl(p67.y, L, WAVE);
            return t;
        };
        auto pf_at = [&](const v2d_t &p) { return PF{__double_as_longlong(__shfl(p.x, L, WAVE)), __double_as_longlong(__shfl(p.y, L, WAVE))}; };
        if (kL == CS_EXACT && b == 0ull) {
            s = __shfl(cur.ex, L, WAVE);
            handled = true;
        } else if ((kL == CS_SPLIT || kL == CS_SPLIT2) && normal && eL == eb && eb + 2 < 0x7ff) {
            // function, the run's real adds, function [, the second run's real adds, function] -- every part checked
            const PF before{__shfl(cur.f0, L, WAVE), __shfl(cur.f1, L, WAVE)};
            m = (long long)((b & CS_FRAC) | CS_HID);
            const long long M1 = m + ((m & 1) ? before.d1 : before.d0);
            if (M1 < CS_BIG) {                            // still inside the binade in front of the run
                double t = add_run(cs_from_m(eb, M1), cur.a01, cur.a23, cur.a45, cur.a67);
                unsigned long long tb = (unsigned long long)__double_as_longlong(t);
                if ((int)((tb >> 52) & 0x7ff) == eb + 1) {
                    const PF mid = pf_at(cur.af);
                    const long long m2 = (long long)((tb & CS_FRAC) | CS_HID);
                    const long long M2 = m2 + ((m2 & 1) ? mid.d1 : mid.d0);
                    if (M2 < CS_BIG) {
                        if (kL == CS_SPLIT) {
                            s = cs_from_m(eb + 1, M2);
                            handled = true;
                        } else {
                            t = add_run(cs_from_m(eb + 1, M2), cur.b01, cur.b23, cur.b45, cur.b67);
                            tb = (unsigned long long)__double_as_longlong(t);
                            if ((int)((tb >> 52) & 0x7ff) == eb + 2) {
                                const PF aft = pf_at(cur.bf);
                                const long long m3 = (long long)((tb & CS_FRAC) | CS_HID);
                                const long long M3 = m3 + ((m3 & 1) ? aft.d1 : aft.d0);
                                if (M3 < CS_BIG) {
                                    s = cs_from_m(eb + 2, M3);
                                    handled = true;
                                }
                            }
                        }
                    }
                }
            }
        }
#ifdef RWR_EXPERIMENTS
        if (handled) { if (kL == CS_SPLIT) ++n_split; else ++n_exact; }
        const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (!handled) {
            s = cs_redo_block<1>(s, n, nchunks, c + L, slot, 0, dangling, X, seeds, c1, in_ptr, in_src, evoff, evterm, lnk);
            ++redo;
#ifdef RWR_EXPERIMENTS
            t_redo += __builtin_amdgcn_s_memrealtime() - tr0;
            ++redo_kind[kL & 3];
#endif
        }
        done = L + 1;
        if (done >= WAVE || c + done >= nchunks) {        // the window is exhausted
            c += WAVE;
            done = 0;
            cur = nxt;
        }
    }
#ifdef RWR_EXPERIMENTS
    if (cs_carry_dbg && lane == 0)
        printf("carry1: %d blocks, applied %u split %u exact, redone %u (plain %u split %u exact %u redo %u), total %.2f us, in redo %.2f us\n",
               nchunks, n_split, n_exact, redo, redo_kind[0], redo_kind[1], redo_kind[2], redo_kind[3],
               (double)(__builtin_amdgcn_s_memrealtime() - t_begin) * 0.01, (double)t_redo * 0.01);
#endif
    if (lane == 0) {
        Y[(size_t)slot * (size_t)n + (size_t)sd] = s;
        // value-free path: the seed row's own z for the next step (the SpMV leaves the seed's row alone)
        if (zout) { const double rw = c1 * s; zout[(size_t)slot * (size_t)n + (size_t)sd] = rw * w_src[sd]; }   // Model.cs:84,87
        if (nz_out && s != 0.0) atomicOr(&nz_out[(size_t)slot * (((size_t)n + 31) / 32) + ((uint32_t)sd >> 5)], 1u << (sd & 31));
        if (redo_count && redo) atomicAdd(redo_count, (unsigned long long)redo);
    }
}

// ---------------------------------------------------------------------------------------------- host side

static inline int cs_block_elems(int G) { return G == 1 ? CsGeom<1>::E : (G >= 16 ? CsGeom<16>::E : CsGeom<2>::E); }
static inline int cs_nchunks(int32_t n, int G) { return (int)(((int64_t)n * G + cs_block_elems(G) - 1) / cs_block_elems(G)); }

#define CS_DISPATCH_G(G, CALL)                            \
    switch (G) {                                          \
        case 1: { constexpr int GG = 1; CALL; } break;    \
        case 2: { constexpr int GG = 2; CALL; } break;    \
        case 4: { constexpr int GG = 4; CALL; } break;    \
        case 8: { constexpr int GG = 8; CALL; } break;    \
        case 16: { constexpr int GG = 16; CALL; } break;  \
        case 32: { constexpr int GG = 32; CALL; } break;  \
        default: { constexpr int GG = 64; CALL; } break;  \
    }

// once per tile group: buffers + the per-block offsets into each seed's in-link list
int32_t chain_scan_prepare(rwr_graph *g, int G, int tg, const int32_t *d_seeds, hipStream_t s)
{
    const int nchunks = cs_nchunks(g->n, G);
    const size_t cells = (size_t)tg * nchunks * G;
    RWR_TRY(g->cs_approx.ensure(cells));
    RWR_TRY(g->cs_e.ensure(cells));
    RWR_TRY(g->cs_d0.ensure(cells));
    RWR_TRY(g->cs_d1.ensure(cells));
    if (G == 1) RWR_TRY(g->cs_side.ensure(cells * CS_SIDE_WORDS));
    if (G == 1) RWR_TRY(g->cs_mx.ensure((size_t)tg * (size_t)(CsGeom<1>::CH + CS_MX_LINKS)));
    RWR_TRY(g->cs_lnk.ensure((size_t)tg * G * (size_t)(nchunks + 1)));
    if (!g->cs_redo.p) {
        RWR_TRY(g->cs_redo.alloc(1));
        RWR_HIP(hipMemsetAsync(g->cs_redo.p, 0, sizeof(unsigned long long), s));
    }
    hipLaunchKernelGGL(k_cs_links, dim3(cdiv((size_t)nchunks + 1, 256), (unsigned)(tg * G)), dim3(256), 0, s, nchunks,
                       cs_block_elems(G) / G, g->in_ptr.p, g->in_src.p, d_seeds, g->cs_lnk.p);
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

// one step's seed-row chain for a tile group (the link terms d_evterm must already be on the stream)
static int32_t chain_scan_launch(rwr_graph *g, int G, int tg, const double *X, double *Y, const int32_t *d_seeds,
                                 const int64_t *d_evoff, double c1, uint32_t *nz_out, const int32_t *lnk, hipStream_t s,
                                 const double *zterms = nullptr, double *zout = nullptr);

static bool cs_split_enabled()
{
    static const int split_env = [] { const char *e = RWR_TUNE_ENV("RWR_SCAN_SPLIT"); return e ? atoi(e) : 1; }();
    return split_env != 0;
}
// single seed per tile: the chain gathers the link terms itself on value-free graphs (no k_seed_terms launch) and writes the
// seed row's z for the next step (no k_seed_z launch)
bool chain_scan_self_contained(int G) { return G == 1 && cs_split_enabled(); }

int32_t chain_scan_step(rwr_graph *g, int G, int tg, const double *X, double *Y, const int32_t *d_seeds,
                        const int64_t *d_evoff, double c1, uint32_t *nz_out, hipStream_t s, const double *zterms, double *zout)
{
    return chain_scan_launch(g, G, tg, X, Y, d_seeds, d_evoff, c1, nz_out, g->cs_lnk.p, s, zterms, zout);
}

// The reference's checkConvergence (Model.cs:110-115) is the same kind of chain: diff += |rank[i] - nextRank[i]| over
// all nodes in order, non-negative addends.  The scan machinery reproduces that sequential fp64 sum bit for bit:
// D holds the addends; with (1-d) := 0 a row's addend is D[i] - 0*D[i] = D[i], the link table is all zero (no in-link
// terms) and the "seed" slot 0 only names where the sum is written (*out).
int32_t chain_scan_sum(rwr_graph *g, const double *D, double *out, hipStream_t s)
{
    const int nchunks = cs_nchunks(g->n, 1);
    const size_t cells = (size_t)nchunks;
    RWR_TRY(g->cs_approx.ensure(cells));
    RWR_TRY(g->cs_e.ensure(cells));
    RWR_TRY(g->cs_d0.ensure(cells));
    RWR_TRY(g->cs_d1.ensure(cells));
    RWR_TRY(g->cs_side.ensure(cells * CS_SIDE_WORDS));
    RWR_TRY(g->cs_mx.ensure((size_t)(CsGeom<1>::CH + CS_MX_LINKS)));
    RWR_TRY(g->cs_lnk0.ensure((size_t)nchunks + 2));
    if (!g->cs_redo.p) {
        RWR_TRY(g->cs_redo.alloc(1));
        RWR_HIP(hipMemsetAsync(g->cs_redo.p, 0, sizeof(unsigned long long), s));
    }
    // words [0, nchunks]: the all-zero link table; word nchunks + 1: the slot's "seed" (node 0)
    RWR_HIP(hipMemsetAsync(g->cs_lnk0.p, 0, ((size_t)nchunks + 2) * sizeof(int32_t), s));
    RWR_TRY(g->d_evoff.ensure(2));
    return chain_scan_launch(g, 1, 1, D, out, g->cs_lnk0.p + nchunks + 1, g->d_evoff.p, 0.0, nullptr, g->cs_lnk0.p, s);
}

static int32_t chain_scan_launch(rwr_graph *g, int G, int tg, const double *X, double *Y, const int32_t *d_seeds,
                                 const int64_t *d_evoff, double c1, uint32_t *nz_out, const int32_t *lnk, hipStream_t s,
                                 const double *zterms, double *zout)
{
    const int nchunks = cs_nchunks(g->n, G);
    const dim3 grid((unsigned)nchunks, (unsigned)tg);
    static const int direct_max = [] { const char *e = RWR_TUNE_ENV("RWR_SCAN_DIRECT_BLOCKS"); return e ? atoi(e) : 8; }();
    // single seed per tile: link terms gathered from the z matrix when the caller hands it over (evoff = nullptr tells the
    // kernels), the seed row's next z written by the carry
    const bool self = chain_scan_self_contained(G);
#ifdef RWR_EXPERIMENTS
    static const int dbg_once = [] {
        const char *e = getenv("RWR_X_CARRY_DBG");
        const int v = e ? atoi(e) : 0;
        if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(cs_carry_dbg), &v, sizeof(v));
        return v;
    }();
    (void)dbg_once;
#endif
    const int64_t *evo = (self && zterms) ? nullptr : d_evoff;
    const double *evt = (self && zterms) ? zterms : g->d_evterm.p;
    // (a single seed takes the direct walk for one block only: with the crossings predicted the two-launch form costs ~20 us per
    //  step whatever the blocks hold, the direct walk 8-9 us per redone block -- 7 500 nodes: 1.83 -> 0.58 ms per call)
    if (nchunks <= (self ? 1 : direct_max)) {
        CS_DISPATCH_G(G, hipLaunchKernelGGL(k_cs_carry<GG>, dim3((unsigned)(tg * G)), dim3(64), 0, s, g->n, nchunks, g->dangling.p, X, Y,
                                            d_seeds, c1, g->in_ptr.p, g->in_src.p, evo, evt, lnk,
                                            g->cs_approx.p, g->cs_e.p, g->cs_d0.p, g->cs_d1.p, nz_out, g->cs_redo.p, 1,
                                            self ? zout : (double *)nullptr, g->w_src.p));
        RWR_HIP(hipGetLastError());
        return RWR_OK;
    }
    if (self) {
        // prediction, split and exact-start blocks in one pass; the carry applies them from registers
        static const int own_max = [] { const char *e = RWR_TUNE_ENV("RWR_SCAN_OWN_SUMS_BLOCKS"); return e ? atoi(e) : 48; }();
        const int own_sums = nchunks <= own_max ? 1 : 0;
        if (!own_sums)
            hipLaunchKernelGGL((k_cs_block<1, false>), grid, dim3(256), 0, s, g->n, nchunks, g->dangling.p, X, d_seeds, c1, g->in_ptr.p,
                               g->in_src.p, evo, evt, lnk, (const int32_t *)nullptr, g->cs_approx.p, (long long *)nullptr,
                               (long long *)nullptr);
        hipLaunchKernelGGL(k_cs_block1, grid, dim3(256), 0, s, g->n, nchunks, g->dangling.p, X, d_seeds, c1, g->in_ptr.p, g->in_src.p,
                           evo, evt, lnk, g->cs_approx.p, g->cs_e.p, g->cs_d0.p, g->cs_d1.p, g->cs_side.p, own_sums, g->cs_mx.p);
        hipLaunchKernelGGL(k_cs_carry1, dim3((unsigned)tg), dim3(64), 0, s, g->n, nchunks, g->dangling.p, X, Y, d_seeds, c1, g->in_ptr.p,
                           g->in_src.p, evo, evt, lnk, g->cs_approx.p, g->cs_e.p, g->cs_d0.p, g->cs_d1.p, g->cs_side.p, nz_out,
                           g->cs_redo.p, zout, g->w_src.p);
        RWR_HIP(hipGetLastError());
        return RWR_OK;
    }
    CS_DISPATCH_G(G, hipLaunchKernelGGL((k_cs_block<GG, false>), grid, dim3(256), 0, s, g->n, nchunks, g->dangling.p, X,
                                        d_seeds, c1, g->in_ptr.p, g->in_src.p, d_evoff, g->d_evterm.p, lnk,
                                        (const int32_t *)nullptr, g->cs_approx.p, (long long *)nullptr, (long long *)nullptr));
    hipLaunchKernelGGL(k_cs_plan, dim3((unsigned)(tg * G)), dim3(64), 0, s, nchunks, d_seeds, g->cs_approx.p, g->cs_e.p);
    CS_DISPATCH_G(G, hipLaunchKernelGGL((k_cs_block<GG, true>), grid, dim3(256), 0, s, g->n, nchunks, g->dangling.p, X,
                                        d_seeds, c1, g->in_ptr.p, g->in_src.p, d_evoff, g->d_evterm.p, lnk,
                                        g->cs_e.p, (double *)nullptr, g->cs_d0.p, g->cs_d1.p));
    CS_DISPATCH_G(G, hipLaunchKernelGGL(k_cs_carry<GG>, dim3((unsigned)(tg * G)), dim3(64), 0, s, g->n, nchunks, g->dangling.p, X, Y,
                                        d_seeds, c1, g->in_ptr.p, g->in_src.p, d_evoff, g->d_evterm.p, lnk,
                                        g->cs_approx.p, g->cs_e.p, g->cs_d0.p, g->cs_d1.p, nz_out, g->cs_redo.p, 0));
    RWR_HIP(hipGetLastError());
    return RWR_OK;
}

// after the caller has synchronised the stream: fold the redo counter into the statistics
int32_t chain_scan_collect(rwr_graph *g, hipStream_t s)
{
    if (!g->cs_redo.p) return RWR_OK;
    unsigned long long v = 0;
    RWR_HIP(hipMemcpyAsync(&v, g->cs_redo.p, sizeof(v), hipMemcpyDeviceToHost, s));
    RWR_HIP(hipMemsetAsync(g->cs_redo.p, 0, sizeof(v), s));
    RWR_HIP(hipStreamSynchronize(s));
    g->stats.chain_redo_blocks += (int64_t)v;
    return RWR_OK;
}

}  // namespace rwr
