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

}  // namespace rwr
