// Parity functions of the binade scan (chain_scan.hip explains the arithmetic; tests/binade_scan_model.py is its executable
// model): inside one binade [2^e, 2^(e+1)) a running fp64 sum s = m * ulp of non-negative addends is an INTEGER sum, except
// that a half-way addend's increment depends on the parity of the running m -- so every addend is a function
// parity -> increment, (d0, d1), and such functions compose associatively.
#pragma once
#include "common.h"

namespace rwr {

constexpr unsigned long long CS_HID = 1ull << 52;
constexpr unsigned long long CS_FRAC = CS_HID - 1ull;
constexpr long long CS_BIG = 1ll << 53;

struct PF {
    long long d0, d1;   // increment of m for incoming parity 0 / 1
};

__device__ __forceinline__ PF pf_of(double a, int eb)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(a);
    int ea = (int)((b >> 52) & 0x7ff);
    unsigned long long mant = b & CS_FRAC;
    if (ea == 0) ea = 1; else mant |= CS_HID;
    const int sh = eb - ea;
    PF r;
    if (mant == 0ull || sh >= 64) {
        r.d0 = r.d1 = 0;
    } else if (sh <= 0) {
        r.d0 = r.d1 = CS_BIG;                       // a >= 2^e: leaves the binade
    } else {
        const unsigned long long k = mant >> sh, rem = mant & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
        if (rem > half) { r.d0 = r.d1 = (long long)(k + 1ull); }
        else if (rem == half) { r.d0 = (long long)(k + (k & 1ull)); r.d1 = (long long)(k + ((k + 1ull) & 1ull)); }
        else { r.d0 = r.d1 = (long long)k; }
    }
    return r;
}
// first f, then g
__device__ __forceinline__ PF pf_compose(PF f, PF g)
{
    PF r;
    r.d0 = f.d0 + ((f.d0 & 1) ? g.d1 : g.d0);
    r.d1 = f.d1 + (((f.d1 + 1) & 1) ? g.d1 : g.d0);
    r.d0 = r.d0 > CS_BIG ? CS_BIG : r.d0;
    r.d1 = r.d1 > CS_BIG ? CS_BIG : r.d1;
    return r;
}
__device__ __forceinline__ double pf_from_m(int eb, long long M)
{
    return __longlong_as_double((long long)(((unsigned long long)eb << 52) | ((unsigned long long)M - CS_HID)));
}

// s + v[0] + v[1] + ... strictly in order for ONE WAVE holding 64 * R addends >= 0 in registers -- lane l owns the run
// [l * R, (l + 1) * R) of the sequence -- as an exact parallel reduction: inside the current binade every run is a
// parity function, one wave scan composes the 64 runs; the run in which the sum leaves the binade is added with real fp64
// adds and the lanes behind it start over under the new binade (the shape of cs_redo_block, chain_scan.hip).  While s is
// zero or subnormal the first run holding a non-zero addend is added with real adds the same way.  Returns the new sum
// (wave-uniform), bit for bit the sequential one.
template <int R>
__device__ __forceinline__ double wave_fold_exact(double s, const double (&v)[R], int lane)
{
    bool nzl = false;
#pragma unroll
    for (int u = 0; u < R; ++u) nzl = nzl || v[u] != 0.0;
    int start = 0;
    while (start < WAVE) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(s);
        const int eb = (int)((b >> 52) & 0x7ff);
        int L;
        if (eb == 0 || eb == 0x7ff) {
            const unsigned long long nzm = __ballot(lane >= start && nzl);
            if (!nzm) break;
            L = __builtin_ctzll(nzm);
        } else {
            PF f{0, 0};
            if (lane >= start && nzl) {
#pragma unroll
                for (int u = 0; u < R; ++u) f = pf_compose(f, pf_of(v[u], eb));
            }
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                PF o;
                o.d0 = __shfl_up(f.d0, off, WAVE);
                o.d1 = __shfl_up(f.d1, off, WAVE);
                if (lane >= off) f = pf_compose(o, f);
            }
            const long long m = (long long)((b & CS_FRAC) | CS_HID);
            const long long Mv = m + ((m & 1) ? f.d1 : f.d0);
            const unsigned long long cross = __ballot(Mv >= CS_BIG);
            if (!cross) {
                s = pf_from_m(eb, __shfl(Mv, WAVE - 1, WAVE));
                break;
            }
            L = __builtin_ctzll(cross);
            if (L > 0) s = pf_from_m(eb, __shfl(Mv, L - 1, WAVE));
        }
        double t = s;
        if (lane == L) {
#pragma unroll
            for (int u = 0; u < R; ++u) t += v[u];
        }
        s = __shfl(t, L, WAVE);
        start = L + 1;
    }
    return s;
}

}  // namespace rwr
