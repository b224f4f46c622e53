// librwr internal definitions (gfx950 only; built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/rwr.h"

namespace rwr {

void set_error(const char *fmt, ...);

#define RWR_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            rwr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                           __LINE__);                                                         \
            return (e__ == hipErrorOutOfMemory) ? RWR_E_NOMEM : RWR_E_HIP;                    \
        }                                                                                     \
    } while (0)

#define RWR_TRY(call)                  \
    do {                               \
        int32_t s__ = (call);          \
        if (s__ != RWR_OK) return s__; \
    } while (0)

constexpr int WAVE = 64;

// Tuning knobs (thresholds and A/B switches used while measuring) are read from the environment only by the experiments
// build (make exp, -DRWR_EXPERIMENTS: tools/ only, never shipped); the shipped library runs their defaults.  What it does
// read -- RWR_DEVICE, RWR_MODE and the selectors between equivalent correct code paths that the parity tests drive
//  -- goes through plain getenv.
#ifdef RWR_EXPERIMENTS
#define RWR_TUNE_ENV(NAME) getenv(NAME)
#else
#define RWR_TUNE_ENV(NAME) ((const char *)nullptr)
#endif

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// order-preserving map of a non-NaN double to uint64 (ascending); -0.0 is first
// canonicalised to +0.0 so that double.CompareTo's -0.0 == +0.0 holds for keys
__host__ __device__ static inline uint64_t f64_orderable(double x)
{
    x = x + 0.0;
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ static inline uint64_t i64_orderable(int64_t x)
{
    return (uint64_t)x ^ 0x8000000000000000ull;
}

// ------------------------------------------------------------------ device buffers
// Device memory of up to 32 MB comes from a per-device cache of power-of-two blocks (api.hip): the reference's workload is
// thousands of ego-network graphs, one Graph per fold from up to ten host threads (Program.cs:11, Experiment.cs:69-105),
// and every hipMalloc / hipFree of the ~40 buffers of a handle is a device-wide synchronisation point that serialises
// those threads.  A block is handed back only by code that has synchronised the stream(s) it was used on (handle
// destruction; the end of a call for its scratch buffers), so a recycled block is idle.  *got = bytes actually reserved.
void *pool_alloc(size_t bytes, size_t *got);
void pool_free(void *p, size_t got);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    size_t reserved = 0;      // bytes behind p (pool block size)
    int32_t alloc(size_t n_elems)
    {
        release();
        if (n_elems == 0) n_elems = 1;
        p = (T *)pool_alloc(n_elems * sizeof(T), &reserved);
        if (!p) {
            const hipError_t e = hipGetLastError();
            rwr::set_error("device allocation of %zu bytes failed: %s", n_elems * sizeof(T), hipGetErrorString(e));
            return (e == hipErrorOutOfMemory) ? RWR_E_NOMEM : RWR_E_HIP;
        }
        count = n_elems;
        return RWR_OK;
    }
    int32_t ensure(size_t n_elems)
    {
        if (p && count >= n_elems) return RWR_OK;
        return alloc(n_elems);
    }
    void release()
    {
        if (p) pool_free(p, reserved);
        p = nullptr;
        count = 0;
        reserved = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// ------------------------------------------------------------------ radix sort (sort.hip)
// Stable LSD radix sort of (key, u32 payload) pairs, nseg equal-length segments laid out
// [seg][m].  Sorts key bits [0, key_bits).  Result lands in (keys, vals) or (keys_alt,
// vals_alt); *in_alt tells which.  temp must hold radix_sort_temp_bytes(m, nseg).
size_t radix_sort_temp_bytes(size_t m, int nseg);
template <typename KeyT>
int32_t radix_sort_pairs(KeyT *keys, KeyT *keys_alt, uint32_t *vals, uint32_t *vals_alt, size_t m,
                         int nseg, int key_bits, void *temp, hipStream_t stream, bool *in_alt);

}  // namespace rwr
