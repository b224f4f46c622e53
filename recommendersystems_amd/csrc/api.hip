// C-ABI entry points of librwr (include/rwr.h).  No CPU fallback anywhere: without a
// usable gfx950 device every compute entry fails with RWR_E_NO_DEVICE.
#include <stdarg.h>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>

#include "engine.h"

namespace rwr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static int usable_devices()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// "is device d a gfx950" -- asked once per device and process: hipGetDeviceProperties goes to the driver under a runtime-wide
// lock, and the reference's harness creates a graph per fold and methodology from ten threads (Program.cs:11)
#ifdef RWR_EXPERIMENTS
static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
#endif
static int32_t check_gfx950(int device)
{
    static std::mutex mu;
    static int state[64];            // 0 unknown, 1 gfx950, 2 something else
    static char names[64][64];
    if (device < 0 || device >= 64) { set_error("device %d out of range", device); return RWR_E_NO_DEVICE; }
    std::lock_guard<std::mutex> lk(mu);
    if (state[device] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_error("hipGetDeviceProperties failed"); return RWR_E_HIP; }
        snprintf(names[device], sizeof(names[device]), "%s", prop.gcnArchName);
        state[device] = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 2;
    }
    if (state[device] != 1) {
        set_error("device %d is %s; librwr is built for gfx950 only", device, names[device]);
        return RWR_E_NO_DEVICE;
    }
    return RWR_OK;
}

// Makes the graph's device current for the duration of one entry point and gives the caller's device back on every
// return path: a host that shares the thread with PyTorch / RCCL keeps its own current device.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int32_t rc = RWR_OK;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != device) {
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) {
                set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
                rc = RWR_E_HIP;
                return;
            }
            switched = true;
        }
    }
    ~DeviceGuard()
    {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define RWR_BIND(g)                                                                                              \
    if ((g)->poisoned) {                                                                                         \
        rwr::set_error("this graph handle was invalidated by a failed rwr_graph_update_links; destroy it");       \
        return RWR_E_INVALID;                                                                                    \
    }                                                                                                            \
    rwr::DeviceGuard dev_guard__((g)->device);                                                                   \
    if (dev_guard__.rc != RWR_OK) return dev_guard__.rc

// Streams, events and the pinned result buffer of a handle are expensive to create and destroy (several hundred
// microseconds together) -- more than everything else the library does for an ego-network-sized graph, and the harness
// creates and drops one Graph per fold and methodology (Experiment.cs:69-105).  They are therefore recycled: a destroyed
// handle parks its set in a small per-process pool (per device), the next created handle on that device takes it.
struct HandleKit {
    int device = -1;
    hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_h0 = nullptr, ev_h1 = nullptr;
    void *pin = nullptr, *stage = nullptr;
};
static std::mutex g_kit_mutex;
static std::vector<HandleKit> g_kits;
constexpr size_t KIT_POOL_MAX = 32;

static bool kit_take(int device, HandleKit *out)
{
    std::lock_guard<std::mutex> lk(g_kit_mutex);
    for (size_t i = 0; i < g_kits.size(); ++i)
        if (g_kits[i].device == device) {
            *out = g_kits[i];
            g_kits.erase(g_kits.begin() + (long)i);
            return true;
        }
    return false;
}
static void kit_destroy(HandleKit &k)
{
    if (k.ev_fork) (void)hipEventDestroy(k.ev_fork);
    if (k.ev_join) (void)hipEventDestroy(k.ev_join);
    if (k.ev_a) (void)hipEventDestroy(k.ev_a);
    if (k.ev_b) (void)hipEventDestroy(k.ev_b);
    if (k.ev_h0) (void)hipEventDestroy(k.ev_h0);
    if (k.ev_h1) (void)hipEventDestroy(k.ev_h1);
    if (k.stream3) (void)hipStreamDestroy(k.stream3);
    if (k.stream) (void)hipStreamDestroy(k.stream);
    if (k.stream2) (void)hipStreamDestroy(k.stream2);
    if (k.pin) (void)hipHostFree(k.pin);
    if (k.stage) (void)hipHostFree(k.stage);
    k = HandleKit{};
}
static void kit_give(HandleKit &k)
{
    {
        std::lock_guard<std::mutex> lk(g_kit_mutex);
        if (k.stream && k.stream2 && k.stream3 && g_kits.size() < KIT_POOL_MAX) {
            g_kits.push_back(k);
            k = HandleKit{};
            return;
        }
    }
    kit_destroy(k);
}

// ---------------------------------------------------------------------------------------------- device-memory cache
// (common.h: DevBuf).  Blocks of 512 B ... 32 MB in power-of-two sizes, kept per device when handed back, up to
// POOL_CAP bytes per device; larger requests go straight to hipMalloc / hipFree.
namespace {
constexpr size_t POOL_MIN = 512, POOL_MAX = 32ull << 20, POOL_CAP = 2ull << 30;
constexpr int POOL_DEVICES = 16, POOL_CLASSES = 17;                 // 512 B << 16 = 32 MB
struct PoolDev { std::vector<void *> free_blocks[POOL_CLASSES]; size_t cached = 0; };
std::mutex g_pool_mutex;
PoolDev g_pool[POOL_DEVICES];
inline int pool_class(size_t bytes, size_t *size)
{
    size_t sz = POOL_MIN;
    int c = 0;
    while (sz < bytes) { sz <<= 1; ++c; }
    *size = sz;
    return c;
}
// In front of the shared cache, a cache per host thread that needs no lock: the harness's ten threads (Program.cs:11) each
// build and drop graphs of ~30 device arrays, and ten threads taking one mutex a thousand times per call spent more time
// in the mutex's slow path than in the rest of the library (measured: 25 ns per operation alone, 3.5 us with ten threads).
// A thread keeps what it frees, up to TL_CAP bytes / TL_PER_CLASS blocks per size; the rest, and everything it holds when it
// ends, goes to the shared cache.
constexpr size_t TL_CAP = 256ull << 20;
constexpr size_t TL_PER_CLASS = 512;
struct ThreadCache {
    int dev = -1;
    std::vector<void *> blocks[POOL_CLASSES];
    size_t bytes = 0;
    void flush()
    {
        if (dev < 0 || dev >= POOL_DEVICES) return;
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        size_t size = POOL_MIN;
        for (int c = 0; c < POOL_CLASSES; ++c, size <<= 1) {
            for (void *p : blocks[c]) {
                g_pool[dev].free_blocks[c].push_back(p);          // (may exceed POOL_CAP: nothing is freed from a destructor)
                g_pool[dev].cached += size;
            }
            blocks[c].clear();
        }
        bytes = 0;
    }
    ~ThreadCache() { flush(); }
};
thread_local ThreadCache tl_cache;
}  // namespace

void *pool_alloc(size_t bytes, size_t *got)
{
    void *p = nullptr;
    if (bytes > POOL_MAX) {
        *got = bytes;
        return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
    }
    size_t size = 0;
    const int c = pool_class(bytes, &size);
    *got = size;
    int dev = 0;
    const bool dev_ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < POOL_DEVICES;
    if (dev_ok && tl_cache.dev == dev && !tl_cache.blocks[c].empty()) {
        p = tl_cache.blocks[c].back();
        tl_cache.blocks[c].pop_back();
        tl_cache.bytes -= size;
        return p;
    }
    if (dev_ok) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        auto &fl = g_pool[dev].free_blocks[c];
        if (!fl.empty()) {
            p = fl.back();
            fl.pop_back();
            g_pool[dev].cached -= size;
            return p;
        }
    }
    if (hipMalloc(&p, size) == hipSuccess) return p;
    // out of memory with blocks parked in the cache: give them back to the driver and try once more
    (void)hipGetLastError();
    std::vector<void *> drop;
    if (dev >= 0 && dev < POOL_DEVICES) {
        if (tl_cache.dev == dev) tl_cache.flush();             // (this thread's own cache too; other threads' stay with them)
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        for (auto &fl : g_pool[dev].free_blocks) { drop.insert(drop.end(), fl.begin(), fl.end()); fl.clear(); }
        g_pool[dev].cached = 0;
    }
    for (void *q : drop) (void)hipFree(q);
    return hipMalloc(&p, size) == hipSuccess ? p : nullptr;
}

void pool_free(void *p, size_t got)
{
    if (!p) return;
    int dev = -1;
    if (got <= POOL_MAX && got >= POOL_MIN && (got & (got - 1)) == 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 &&
        dev < POOL_DEVICES) {
        // (the block belongs to the current device: every entry point binds the handle's device before it touches memory)
        size_t size = 0;
        const int c = pool_class(got, &size);
        if (tl_cache.dev != dev && tl_cache.bytes == 0) tl_cache.dev = dev;
        if (tl_cache.dev == dev && tl_cache.bytes + size <= TL_CAP && tl_cache.blocks[c].size() < TL_PER_CLASS) {
            tl_cache.blocks[c].push_back(p);
            tl_cache.bytes += size;
            return;
        }
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        if (g_pool[dev].cached + size <= POOL_CAP) {
            g_pool[dev].free_blocks[c].push_back(p);
            g_pool[dev].cached += size;
            return;
        }
    }
    (void)hipFree(p);
}

}  // namespace rwr

using namespace rwr;

extern "C" {

const char *rwr_version(void) { return RWR_VERSION_STRING " (gfx950, hip)"; }

int32_t rwr_device_count(void) { return usable_devices(); }

const char *rwr_last_error(void) { return g_err; }

int32_t rwr_graph_create(int32_t n, const int64_t *node_id, const uint8_t *node_type, const int64_t *rowptr,
                         const int32_t *dst, const uint8_t *etype, const double *w, const rwr_opts *opts,
                         rwr_graph **out)
{
    g_err[0] = 0;
    if (!out) { set_error("rwr_graph_create: out is NULL"); return RWR_E_INVALID; }
    *out = nullptr;
    if (n <= 0 || !node_id || !node_type || !rowptr) {
        set_error("rwr_graph_create: n must be > 0 and node_id/node_type/rowptr non-NULL");
        return RWR_E_INVALID;
    }
    if (rowptr[0] != 0) { set_error("rwr_graph_create: rowptr[0] must be 0"); return RWR_E_INVALID; }
    for (int32_t i = 0; i < n; ++i)
        if (rowptr[i + 1] < rowptr[i]) {
            set_error("rwr_graph_create: rowptr decreases at node %d", i);
            return RWR_E_INVALID;
        }
    const int64_t m = rowptr[n];
    if (m > 0 && (!dst || !etype || !w)) {
        set_error("rwr_graph_create: dst/etype/w must be non-NULL when there are links");
        return RWR_E_INVALID;
    }
    const int ndev = usable_devices();
    if (ndev <= 0) { set_error("no usable HIP device (librwr has no CPU fallback)"); return RWR_E_NO_DEVICE; }

    rwr_graph *g = new (std::nothrow) rwr_graph();
    if (!g) { set_error("out of host memory"); return RWR_E_NOMEM; }
    rwr_opts o{};
    o.struct_size = sizeof(rwr_opts);
    o.device = -1;
    o.mode = -1;
    if (opts) {
        size_t sz = opts->struct_size > 0 ? (size_t)opts->struct_size : sizeof(rwr_opts);
        if (sz > sizeof(rwr_opts)) sz = sizeof(rwr_opts);
        memcpy(&o, opts, sz);
    }
    if (o.device < 0) {
        const char *e = getenv("RWR_DEVICE");
        if (e && *e) o.device = atoi(e);
        else {
            int cur = 0;
            if (hipGetDevice(&cur) != hipSuccess) cur = 0;
            o.device = cur;
        }
    }
    if (o.mode < 0) {
        const char *e = getenv("RWR_MODE");
        o.mode = (e && strcmp(e, "fast") == 0) ? 1 : RWR_MODE_EXACT;
    }
    if (o.device >= ndev) {
        set_error("rwr_graph_create: device %d requested but only %d visible", o.device, ndev);
        delete g;
        return RWR_E_NO_DEVICE;
    }
    if (o.mode != RWR_MODE_EXACT) {
        // (mode 1 was FAST until ABI 3: re-associated sums with no ranking guarantee and no speed advantage -- removed)
        set_error("rwr_graph_create: unknown mode %d (the only arithmetic mode is RWR_MODE_EXACT = 0; FAST was removed)", o.mode);
        delete g;
        return RWR_E_INVALID;
    }
    g->opts = o;
    g->device = o.device;
    g->n = n;
    g->nnz_raw = m;
    int32_t rc = RWR_OK;
    auto fail = [&](int32_t code) {
        rwr_graph_destroy(g);
        return code;
    };
    rwr::DeviceGuard dev_guard(g->device);
    if (dev_guard.rc != RWR_OK) return fail(dev_guard.rc);
    if ((rc = check_gfx950(g->device)) != RWR_OK) return fail(rc);
    // the seed-row chain (stream2) is latency-bound and must not queue behind the SpMM's half-million
    // workgroups: give its stream the highest dispatch priority
    HandleKit kit;
    if (kit_take(g->device, &kit)) {   // recycled from a destroyed handle (idle: its streams were synchronised)
        g->stream = kit.stream; g->stream2 = kit.stream2; g->stream3 = kit.stream3;
        g->ev_fork = kit.ev_fork; g->ev_join = kit.ev_join; g->ev_a = kit.ev_a; g->ev_b = kit.ev_b;
        g->ev_h0 = kit.ev_h0; g->ev_h1 = kit.ev_h1;
        g->sm_pin = kit.pin;
        g->sm_stage = kit.stage;
    } else {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithPriority(&g->stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
            hipStreamCreateWithPriority(&g->stream3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_h0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_h1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&g->ev_a) != hipSuccess || hipEventCreate(&g->ev_b) != hipSuccess) {
            set_error("stream/event creation failed: %s", hipGetErrorString(hipGetLastError()));
            return fail(RWR_E_HIP);
        }
    }
    rc = graph_build(g, node_id, node_type, rowptr, dst, etype, w);
    if (rc != RWR_OK) return fail(rc);
    g->stats.struct_size = sizeof(rwr_stats);
    g->stats.n = n;
    g->stats.nnz_raw = m;
    g->stats.nnz = g->nnz;
    g->stats.uniform = g->uniform;
    g->stats.mode = g->opts.mode;
    *out = g;
    return RWR_OK;
}

int32_t rwr_graph_update_links(rwr_graph *g, int64_t count, const int64_t *link_index, const uint8_t *etype,
                               const double *w)
{
    g_err[0] = 0;
    if (!g) { set_error("rwr_graph_update_links: graph is NULL"); return RWR_E_INVALID; }
    if (count < 0 || (count > 0 && !link_index)) {
        set_error("rwr_graph_update_links: count must be >= 0 and link_index non-NULL when count > 0");
        return RWR_E_INVALID;
    }
    RWR_BIND(g);
    const int32_t rc = graph_update_links(g, count, link_index, etype, w);
    // a failure after the raw lists were patched leaves derived arrays that no longer match them
    if (rc != RWR_OK && rc != RWR_E_RANGE && rc != RWR_E_INVALID) g->poisoned = 1;
    return rc;
}

int32_t rwr_graph_destroy(rwr_graph *g)
{
    if (!g) return RWR_OK;
    rwr::DeviceGuard dev_guard(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->stream2) (void)hipStreamSynchronize(g->stream2);
    if (g->stream3) (void)hipStreamSynchronize(g->stream3);
    HandleKit kit;
    kit.device = g->device;
    kit.stream = g->stream; kit.stream2 = g->stream2; kit.stream3 = g->stream3;
    kit.ev_h0 = g->ev_h0; kit.ev_h1 = g->ev_h1;
    kit.ev_fork = g->ev_fork; kit.ev_join = g->ev_join; kit.ev_a = g->ev_a; kit.ev_b = g->ev_b;
    kit.pin = g->sm_pin;
    kit.stage = g->sm_stage;
    kit_give(kit);            // (parked for the next handle on this device, or destroyed when the pool is full)
    delete g;
    return RWR_OK;
}

int32_t rwr_graph_size(const rwr_graph *g, int32_t *n, int64_t *nnz_raw, int64_t *nnz_explicit)
{
    if (!g) { set_error("rwr_graph_size: graph is NULL"); return RWR_E_INVALID; }
    if (n) *n = g->n;
    if (nnz_raw) *nnz_raw = g->nnz_raw;
    if (nnz_explicit) *nnz_explicit = g->nnz;
    return RWR_OK;
}

int32_t rwr_graph_get_normalized(rwr_graph *g, double *w_out, uint8_t *dangling_out)
{
    g_err[0] = 0;
    if (!g) { set_error("rwr_graph_get_normalized: graph is NULL"); return RWR_E_INVALID; }
    RWR_BIND(g);
    if (w_out && g->nnz_raw > 0)
        RWR_HIP(hipMemcpy(w_out, g->w_norm_raw.p, sizeof(double) * (size_t)g->nnz_raw, hipMemcpyDeviceToHost));
    if (dangling_out) RWR_HIP(hipMemcpy(dangling_out, g->dangling.p, (size_t)g->n, hipMemcpyDeviceToHost));
    return RWR_OK;
}

int32_t rwr_eval_graphs(int32_t count, const rwr_graph_desc *graphs, const int32_t *seeds, float d, int32_t n_iter,
                        const int64_t *test_ptr, const int64_t *test_ids, const rwr_opts *opts, int64_t *n_hits,
                        double *sum_precision, int64_t *list_len)
{
    g_err[0] = 0;
#ifdef RWR_EXPERIMENTS
    const double t_enter = now_ms();
    double t_kit = 0, t_loop = 0;
#endif
    if (count < 0 || (count > 0 && (!graphs || !seeds || !test_ptr || !n_hits || !sum_precision))) {
        set_error("rwr_eval_graphs: bad argument");
        return RWR_E_INVALID;
    }
    if (count == 0) return RWR_OK;
    if (n_iter < 0) n_iter = 0;
    for (int32_t i = 0; i < count; ++i) {
        const rwr_graph_desc &D = graphs[i];
        if (D.n_nodes <= 0 || !D.node_id || !D.node_type || !D.rowptr) {
            set_error("rwr_eval_graphs: graph %d: n_nodes must be > 0 and node_id/node_type/rowptr non-NULL", i);
            return RWR_E_INVALID;
        }
        if (D.rowptr[0] != 0) { set_error("rwr_eval_graphs: graph %d: rowptr[0] must be 0", i); return RWR_E_INVALID; }
        for (int32_t q = 0; q < D.n_nodes; ++q)
            if (D.rowptr[q + 1] < D.rowptr[q]) { set_error("rwr_eval_graphs: graph %d: rowptr decreases at node %d", i, q); return RWR_E_INVALID; }
        if (D.rowptr[D.n_nodes] > 0 && (!D.dst || !D.etype || !D.w)) {
            set_error("rwr_eval_graphs: graph %d: dst/etype/w must be non-NULL when there are links", i);
            return RWR_E_INVALID;
        }
        if (seeds[i] < 0 || seeds[i] >= D.n_nodes) { set_error("graph %d: seed %d is outside [0, %d)", i, seeds[i], D.n_nodes); return RWR_E_RANGE; }
        if (test_ptr[i + 1] < test_ptr[i] || test_ptr[i + 1] - test_ptr[i] > 0x7FFFFFFF) {
            set_error("rwr_eval_graphs: test_ptr must be non-decreasing (graph %d)", i);
            return RWR_E_INVALID;
        }
    }
    if (test_ptr[count] > test_ptr[0] && !test_ids) { set_error("rwr_eval_graphs: test_ids is NULL"); return RWR_E_INVALID; }
    for (int32_t i = 0; i < count; ++i) { n_hits[i] = 0; sum_precision[i] = 0.0; if (list_len) list_len[i] = 0; }

    // a graph that the one-launch paths cannot take goes through the public single-graph calls
    auto one_by_one = [&](int32_t i) -> int32_t {
        const rwr_graph_desc &D = graphs[i];
        rwr_graph *g1 = nullptr;
        int32_t rc = rwr_graph_create(D.n_nodes, D.node_id, D.node_type, D.rowptr, D.dst, D.etype, D.w, opts, &g1);
        if (rc == RWR_OK)
            rc = rwr_recommend_eval(g1, seeds[i], d, n_iter, test_ids ? test_ids + test_ptr[i] : nullptr, test_ptr[i + 1] - test_ptr[i],
                                    n_hits + i, sum_precision + i, list_len ? list_len + i : nullptr);
        if (rc != RWR_OK) {
            char msg[600];
            snprintf(msg, sizeof(msg), "%s", g_err);
            set_error("graph %d of the batch: %s", i, msg);
        }
        if (g1) (void)rwr_graph_destroy(g1);
        return rc;
    };
    std::vector<int32_t> fast, slow;
    const bool d_ok = (double)d >= 0.0 && (double)d <= 1.0;
    for (int32_t i = 0; i < count; ++i) {
        const rwr_graph_desc &D = graphs[i];
        (d_ok && rwr::graph_fits_small_build(D.n_nodes, D.rowptr[D.n_nodes]) ? fast : slow).push_back(i);
    }
    if (!fast.empty()) {
        // the batch's stream comes from the handle pool (rwr_graph_create's); every graph of the batch works on it
        const int ndev = usable_devices();
        if (ndev <= 0) { set_error("no usable HIP device (librwr has no CPU fallback)"); return RWR_E_NO_DEVICE; }
        rwr_opts o{};
        o.struct_size = sizeof(rwr_opts);
        o.device = -1;
        o.mode = -1;
        if (opts) {
            size_t sz = opts->struct_size > 0 ? (size_t)opts->struct_size : sizeof(rwr_opts);
            if (sz > sizeof(rwr_opts)) sz = sizeof(rwr_opts);
            memcpy(&o, opts, sz);
        }
        if (o.device < 0) {
            const char *e = getenv("RWR_DEVICE");
            if (e && *e) o.device = atoi(e);
            else { int cur = 0; if (hipGetDevice(&cur) != hipSuccess) cur = 0; o.device = cur; }
        }
        if (o.mode < 0) o.mode = RWR_MODE_EXACT;
        if (o.device >= ndev) { set_error("rwr_eval_graphs: device %d requested but only %d visible", o.device, ndev); return RWR_E_NO_DEVICE; }
        if (o.mode != RWR_MODE_EXACT) { set_error("rwr_eval_graphs: unknown mode %d (the only arithmetic mode is RWR_MODE_EXACT = 0)", o.mode); return RWR_E_INVALID; }
        rwr::DeviceGuard dev_guard(o.device);
        if (dev_guard.rc != RWR_OK) return dev_guard.rc;
        RWR_TRY(check_gfx950(o.device));
        HandleKit kit;
        if (!kit_take(o.device, &kit)) {     // (a full set, as rwr_graph_create makes one, so that the pool keeps it afterwards)
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (hipStreamCreateWithFlags(&kit.stream, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithPriority(&kit.stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                hipStreamCreateWithPriority(&kit.stream3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                hipEventCreateWithFlags(&kit.ev_h0, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&kit.ev_h1, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&kit.ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&kit.ev_join, hipEventDisableTiming) != hipSuccess ||
                hipEventCreate(&kit.ev_a) != hipSuccess || hipEventCreate(&kit.ev_b) != hipSuccess) {
                set_error("stream/event creation failed: %s", hipGetErrorString(hipGetLastError()));
                kit_destroy(kit);
                return RWR_E_HIP;
            }
        }
        kit.device = o.device;
        rwr::multi_pins_acquire();
        // Every pinned set starts at sizes the usual batch (tens of ego networks) never outgrows: growing a buffer is a
        // hipHostFree + hipHostMalloc -- milliseconds each, the first a device synchronisation --, and which set a call
        // gets changes from call to call, so sets sized by their first batch would keep growing for many calls when ten
        // threads with batches of different sizes share them.
        {
            static const size_t floor_bytes[5] = {(size_t)16 << 20, (size_t)256 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)64 << 10};
            int32_t prc = RWR_OK;
            void *unused = nullptr;
            for (int sl = 0; sl < 5 && prc == RWR_OK; ++sl) prc = rwr::multi_pinned(sl, floor_bytes[sl], &unused);
            if (prc != RWR_OK) {
                rwr::multi_pins_release();
                kit_give(kit);
                return prc;
            }
        }
#ifdef RWR_EXPERIMENTS
        t_kit = now_ms();
#endif
        constexpr int32_t CHUNK = 256;                     // graphs per launch (a workgroup each)
        int32_t rc = RWR_OK;
        std::vector<rwr_graph *> gs;
        auto drop = [&]() {
            for (rwr_graph *g : gs) delete g;              // (device buffers go back to the pool; nothing else is owned)
            gs.clear();
        };
        for (size_t c0 = 0; c0 < fast.size() && rc == RWR_OK; c0 += CHUNK) {
            const int32_t nc = (int32_t)std::min<size_t>(CHUNK, fast.size() - c0);
            std::vector<rwr_graph_desc> descs((size_t)nc);
            for (int32_t q = 0; q < nc; ++q) {
                const int32_t i = fast[c0 + q];
                descs[q] = graphs[i];
                rwr_graph *g = new (std::nothrow) rwr_graph();
                if (!g) { set_error("out of host memory"); rc = RWR_E_NOMEM; break; }
                g->opts = o;
                g->device = o.device;
                g->n = graphs[i].n_nodes;
                g->nnz_raw = graphs[i].rowptr[graphs[i].n_nodes];
                g->stream = kit.stream;
                g->borrowed = 1;
                g->stats.struct_size = sizeof(rwr_stats);
                gs.push_back(g);
            }
            if (rc != RWR_OK) break;
#ifdef RWR_EXPERIMENTS
            static const bool mt_on = [] { const char *e = getenv("RWR_X_MULTI_TIMING"); return e && atoi(e) != 0; }();
            const double q0 = now_ms();
            double q1 = 0, q2 = 0, q3 = 0;
#endif
            rc = rwr::graphs_build_multi(gs.data(), nc, descs.data(), fast.data() + c0, kit.stream);
#ifdef RWR_EXPERIMENTS
            q1 = now_ms();
#endif
            if (rc != RWR_OK) break;
            // the one-launch call takes the graphs that pass its own limits (items, links, the seed's in-list)
            std::vector<rwr_graph *> run;
            std::vector<int32_t> run_seed, run_at;
            for (int32_t q = 0; q < nc; ++q) {
                const int32_t i = fast[c0 + q];
                rwr_graph *g = gs[q];
                if (g->n_items == 0) continue;             // (no candidates: hits 0, empty list)
                if (g->nonneg && rwr::small_path_ok(g) && rwr::small_path_seed_ok(g, seeds[i])) {
                    run.push_back(g);
                    run_seed.push_back(seeds[i]);
                    run_at.push_back(i);
                } else {
                    slow.push_back(i);
                }
            }
            if (!run.empty()) {
                const int32_t nr = (int32_t)run.size();
                rwr::DevBuf<uint8_t> args_keep;            // (lives until eval_ranked_multi has synchronised the stream)
                rc = rwr::recommend_small_multi(run.data(), run_seed.data(), nr, (double)d, n_iter, kit.stream, args_keep);
#ifdef RWR_EXPERIMENTS
                q2 = now_ms();
#endif
                if (rc == RWR_OK) {
                    // HashSet<long> semantics per test set (Experiment.cs:124): sorted, duplicates dropped
                    std::vector<int64_t> ts, tp((size_t)nr + 1, 0), hits((size_t)nr), lens((size_t)nr);
                    std::vector<double> sums((size_t)nr);
                    for (int32_t q = 0; q < nr; ++q) {
                        const int32_t i = run_at[q];
                        const size_t at = ts.size();
                        if (test_ptr[i + 1] > test_ptr[i]) ts.insert(ts.end(), test_ids + test_ptr[i], test_ids + test_ptr[i + 1]);
                        std::sort(ts.begin() + (long)at, ts.end());
                        ts.erase(std::unique(ts.begin() + (long)at, ts.end()), ts.end());
                        tp[(size_t)q + 1] = (int64_t)ts.size();
                    }
                    rc = rwr::eval_ranked_multi(run.data(), nr, tp.data(), ts.data(), hits.data(), sums.data(), lens.data(), kit.stream);
                    if (rc == RWR_OK)
                        for (int32_t q = 0; q < nr; ++q) {
                            const int32_t i = run_at[q];
                            n_hits[i] = hits[q];
                            sum_precision[i] = sums[q];
                            if (list_len) list_len[i] = lens[q];
                        }
                }
                if (rc != RWR_OK) (void)hipStreamSynchronize(kit.stream);   // (nothing may be in flight when the tables are released)
            }
#ifdef RWR_EXPERIMENTS
            q3 = now_ms();
#endif
            drop();
#ifdef RWR_EXPERIMENTS
            if (mt_on) fprintf(stderr, "[multi] %d graphs: build %.0f us, call enqueue %.0f, evaluation (incl. wait) %.0f, release %.0f\n", nc,
                               1e3 * (q1 - q0), 1e3 * (q2 - q1), 1e3 * (q3 - q2), 1e3 * (now_ms() - q3));
#endif
        }
        drop();
#ifdef RWR_EXPERIMENTS
        t_loop = now_ms();
#endif
        (void)hipStreamSynchronize(kit.stream);
        rwr::multi_pins_release();
        kit_give(kit);
#ifdef RWR_EXPERIMENTS
        {
            static const bool mt_on2 = [] { const char *e = getenv("RWR_X_MULTI_TIMING"); return e && atoi(e) != 0; }();
            if (mt_on2) fprintf(stderr, "[multi call] entry->kit %.0f us, chunks %.0f, tail %.0f\n", 1e3 * (t_kit - t_enter), 1e3 * (t_loop - t_kit),
                                1e3 * (now_ms() - t_loop));
        }
#endif
        if (rc != RWR_OK) return rc;
    }
    std::sort(slow.begin(), slow.end());
    for (int32_t i : slow) RWR_TRY(one_by_one(i));
    return RWR_OK;
}

int32_t rwr_recommend_batch(rwr_graph *g, const int32_t *seeds, int32_t K, float d, int32_t n_iter, int32_t top_n,
                            int64_t *ids, double *scores, int32_t *counts)
{
    g_err[0] = 0;
    if (!g || !seeds || !ids || !scores || !counts) { set_error("rwr_recommend_batch: NULL argument"); return RWR_E_INVALID; }
    if (K <= 0) { set_error("rwr_recommend_batch: K must be >= 1"); return RWR_E_INVALID; }
    if (top_n < 1) { set_error("rwr_recommend_batch: top_n must be >= 1"); return RWR_E_INVALID; }
    if (n_iter < 0) n_iter = 0;   // Model.run(int): a non-positive count runs no iteration (Model.cs:69)
    RWR_BIND(g);
    // Recommender.cs:14,16 -> Model.cs:33: the float is widened to double
    return recommend_batch(g, seeds, K, (double)d, n_iter, top_n, ids, scores, counts, top_n);
}

int32_t rwr_recommend(rwr_graph *g, int32_t seed, float d, int32_t n_iter, int32_t top_n, int64_t *out_id,
                      double *out_score, int64_t *inout_count)
{
    g_err[0] = 0;
    if (!g || !inout_count) { set_error("rwr_recommend: NULL argument"); return RWR_E_INVALID; }
    if (seed < 0 || seed >= g->n) { set_error("seed %d is outside [0, %d)", seed, g->n); return RWR_E_RANGE; }
    if (n_iter < 0) n_iter = 0;
    RWR_BIND(g);
    // Recommender.cs:42-51: topN <= 0 never truncates
    int64_t width = (top_n > 0 && top_n < g->n_items) ? top_n : g->n_items;
    if (width == 0) { *inout_count = 0; return RWR_OK; }
    if (width > 0x7FFFFFFF) { set_error("rwr_recommend: list too long"); return RWR_E_UNSUPPORTED; }
    int32_t cnt = 0;
    if (width <= rwr::small_pin_words() && out_id && out_score && *inout_count >= width &&
        !(rwr::small_path_ok(g) && rwr::small_path_seed_ok(g, seed))) {
        // a short list (top-N of a large graph): count, ids and scores come back into the handle's pinned buffer behind ONE
        // synchronisation (waiting for the count first and copying then cost a second host round trip: ~90 us of a 1.6 ms call)
        RWR_TRY(rwr::small_pin_ensure(g));
        int64_t *pid = const_cast<int64_t *>(rwr::small_pin_ids(g));
        double *psc = const_cast<double *>(rwr::small_pin_scores(g));
        g->sm_pin_count = -1;
        RWR_TRY(recommend_batch(g, &seed, 1, (double)d, n_iter, (int32_t)width, pid, psc, &cnt, width));
        memcpy(out_id, pid, sizeof(int64_t) * (size_t)cnt);
        memcpy(out_score, psc, sizeof(double) * (size_t)cnt);
        *inout_count = cnt;
        return RWR_OK;
    }
    // the ranked list stays on the device until its length is known, then goes straight into the caller's arrays
    // (no host staging copy: the full list of a 5 M-item graph is 80 MB)
    g->sm_pin_count = -1;
    RWR_TRY(recommend_batch(g, &seed, 1, (double)d, n_iter, (int32_t)width, nullptr, nullptr, &cnt, width));
    if (*inout_count < cnt || ((!out_id || !out_score) && cnt > 0)) {
        set_error("rwr_recommend: output holds %lld entries, %d needed", (long long)*inout_count, cnt);
        *inout_count = cnt;
        return RWR_E_CAPACITY;
    }
    if (cnt > 0 && g->sm_pin_count == cnt) {
        // ego-network-sized graph (small.hip): the kernel wrote the list into pinned host memory
        memcpy(out_id, rwr::small_pin_ids(g), sizeof(int64_t) * (size_t)cnt);
        memcpy(out_score, rwr::small_pin_scores(g), sizeof(double) * (size_t)cnt);
    } else if (cnt > 0) {
        RWR_HIP(hipMemcpyAsync(out_id, g->d_out_id.p, sizeof(int64_t) * (size_t)cnt, hipMemcpyDeviceToHost, g->stream));
        RWR_HIP(hipMemcpyAsync(out_score, g->d_out_score.p, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost, g->stream));
        RWR_HIP(hipStreamSynchronize(g->stream));
    }
    *inout_count = cnt;
    return RWR_OK;
}

int32_t rwr_recommend_eval(rwr_graph *g, int32_t seed, float d, int32_t n_iter, const int64_t *test_ids,
                           int64_t n_test, int64_t *n_hits, double *sum_precision, int64_t *list_len)
{
    g_err[0] = 0;
    if (!g || !n_hits || !sum_precision || (n_test > 0 && !test_ids) || n_test < 0) {
        set_error("rwr_recommend_eval: bad argument");
        return RWR_E_INVALID;
    }
    if (seed < 0 || seed >= g->n) { set_error("seed %d is outside [0, %d)", seed, g->n); return RWR_E_RANGE; }
    if (n_test > 0x7FFFFFFF) { set_error("rwr_recommend_eval: test set too large"); return RWR_E_UNSUPPORTED; }
    if (n_iter < 0) n_iter = 0;
    RWR_BIND(g);
    *n_hits = 0;
    *sum_precision = 0.0;
    if (list_len) *list_len = 0;
    const int32_t width = g->n_items;
    if (width == 0) return RWR_OK;
    int32_t cnt = 0;
    RWR_TRY(recommend_batch(g, &seed, 1, (double)d, n_iter, width, nullptr, nullptr, &cnt, width));
    std::vector<int64_t> ts(test_ids, test_ids + n_test);
    std::sort(ts.begin(), ts.end());
    ts.erase(std::unique(ts.begin(), ts.end()), ts.end());     // HashSet<long>
    RWR_TRY(eval_ranked(g, cnt, ts.data(), (int64_t)ts.size(), n_hits, sum_precision));
    if (list_len) *list_len = cnt;
    return RWR_OK;
}

int32_t rwr_recommend_eval_batch(rwr_graph *g, const int32_t *seeds, int32_t K, float d, int32_t n_iter,
                                 const int64_t *test_ptr, const int64_t *test_ids, int64_t *n_hits, double *sum_precision,
                                 int64_t *list_len)
{
    g_err[0] = 0;
    if (!g || !seeds || K < 0 || !test_ptr || !n_hits || !sum_precision) {
        set_error("rwr_recommend_eval_batch: bad argument");
        return RWR_E_INVALID;
    }
    for (int32_t k = 0; k < K; ++k) {
        if (seeds[k] < 0 || seeds[k] >= g->n) { set_error("seed %d (batch position %d) is outside [0, %d)", seeds[k], k, g->n); return RWR_E_RANGE; }
        if (test_ptr[k + 1] < test_ptr[k] || test_ptr[k + 1] - test_ptr[k] > 0x7FFFFFFF) {
            set_error("rwr_recommend_eval_batch: test_ptr must be non-decreasing (position %d)", k);
            return RWR_E_INVALID;
        }
    }
    if (K > 0 && test_ptr[K] > test_ptr[0] && !test_ids) { set_error("rwr_recommend_eval_batch: test_ids is NULL"); return RWR_E_INVALID; }
    if (n_iter < 0) n_iter = 0;
    RWR_BIND(g);
    for (int32_t k = 0; k < K; ++k) { n_hits[k] = 0; sum_precision[k] = 0.0; if (list_len) list_len[k] = 0; }
    const int32_t width = g->n_items;
    if (width == 0 || K == 0) return RWR_OK;
    // HashSet<long> semantics per test set (Experiment.cs:124): sorted, duplicates dropped; offsets rebuilt
    std::vector<int64_t> ts, tp((size_t)K + 1, 0);
    ts.reserve((size_t)(test_ptr[K] - test_ptr[0]));
    for (int32_t k = 0; k < K; ++k) {
        const size_t at = ts.size();
        ts.insert(ts.end(), test_ids + test_ptr[k], test_ids + test_ptr[k + 1]);
        std::sort(ts.begin() + at, ts.end());
        ts.erase(std::unique(ts.begin() + at, ts.end()), ts.end());
        tp[(size_t)k + 1] = (int64_t)ts.size();
    }
    // the full ranked lists stay on the device, `chunk` seeds at a time (16 bytes per candidate and seed)
    const int64_t per_seed = (int64_t)width * 16;
    int32_t chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(K, (1ll << 30) / std::max<int64_t>(per_seed, 1)));
    std::vector<int32_t> cnt((size_t)K);
    for (int32_t k0 = 0; k0 < K; k0 += chunk) {
        const int32_t kc = std::min(chunk, K - k0);
        RWR_TRY(recommend_batch(g, seeds + k0, kc, (double)d, n_iter, width, nullptr, nullptr, cnt.data() + k0, width));
        std::vector<int64_t> tpc((size_t)kc + 1);
        for (int32_t q = 0; q <= kc; ++q) tpc[q] = tp[(size_t)k0 + q] - tp[k0];
        RWR_TRY(eval_ranked_batch(g, kc, width, tpc.data(), ts.data() + tp[k0], n_hits + k0, sum_precision + k0));
    }
    if (list_len) for (int32_t k = 0; k < K; ++k) list_len[k] = cnt[k];
    return RWR_OK;
}

int32_t rwr_model_deliver(rwr_graph *g, int32_t seed, double d, const double *rank, double *next_rank)
{
    g_err[0] = 0;
    if (!g || !rank || !next_rank || rank == next_rank) { set_error("rwr_model_deliver: NULL or aliasing argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return model_deliver(g, seed, d, rank, next_rank);
}

int32_t rwr_model_run(rwr_graph *g, int32_t seed, double d, int32_t run_mode, double value, double *rank_out,
                      int64_t *iters_out)
{
    g_err[0] = 0;
    if (!g || !rank_out) { set_error("rwr_model_run: NULL argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    if (run_mode != RWR_RUN_ITERATIONS && run_mode != RWR_RUN_THRESHOLD && run_mode != RWR_RUN_DEFAULT_THRESHOLD) {
        set_error("rwr_model_run: unknown run_mode %d", run_mode);
        return RWR_E_INVALID;
    }
    RWR_TRY(model_run(g, seed, d, run_mode, value, rank_out, iters_out));
    return RWR_OK;
}

int32_t rwr_part_begin(rwr_graph *g, int32_t slab_lo, int32_t slab_hi, const int32_t *seeds, int32_t K, double d,
                       void *dev_x, int32_t *tile_seeds_out)
{
    g_err[0] = 0;
    if (!g || !seeds || !dev_x) { set_error("rwr_part_begin: NULL argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return part_begin(g, slab_lo, slab_hi, seeds, K, d, (double *)dev_x, tile_seeds_out);
}

int32_t rwr_part_local_step(rwr_graph *g, const void *dev_x, void *dev_y, void *dev_r)
{
    g_err[0] = 0;
    if (!g || !dev_x || !dev_y || !dev_r) { set_error("rwr_part_local_step: NULL argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return part_local_step(g, (const double *)dev_x, (double *)dev_y, (double *)dev_r);
}

int32_t rwr_part_step(rwr_graph *g, const void *dev_x, void *dev_y, void *stream)
{
    g_err[0] = 0;
    if (!g || !dev_x || !dev_y || dev_x == dev_y) { set_error("rwr_part_step: NULL or aliasing argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return part_step(g, (const double *)dev_x, (double *)dev_y, (hipStream_t)stream);
}

int32_t rwr_part_finish_step(rwr_graph *g, void *dev_y, const void *dev_r)
{
    g_err[0] = 0;
    if (!g || !dev_y || !dev_r) { set_error("rwr_part_finish_step: NULL argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return part_finish_step(g, (double *)dev_y, (const double *)dev_r);
}

int32_t rwr_part_rank(rwr_graph *g, void *dev_x, int32_t top_n, int64_t *ids, double *scores, int32_t *counts)
{
    g_err[0] = 0;
    if (!g || !dev_x || !ids || !scores || !counts) { set_error("rwr_part_rank: NULL argument"); return RWR_E_INVALID; }
    RWR_BIND(g);
    return part_rank(g, (double *)dev_x, top_n, ids, scores, counts);
}

int32_t rwr_get_stats(rwr_graph *g, rwr_stats *out)
{
    if (!g || !out) { set_error("rwr_get_stats: NULL argument"); return RWR_E_INVALID; }
    size_t sz = out->struct_size > 0 ? (size_t)out->struct_size : sizeof(rwr_stats);
    if (sz > sizeof(rwr_stats)) sz = sizeof(rwr_stats);
    rwr_stats s = g->stats;
    s.struct_size = (int32_t)sz;
    memcpy(out, &s, sz);
    return RWR_OK;
}

int32_t rwr_reset_stats(rwr_graph *g)
{
    if (!g) { set_error("rwr_reset_stats: NULL argument"); return RWR_E_INVALID; }
    rwr_stats &s = g->stats;
    s.spmm_ms = s.chain_ms = s.rank_ms = s.iterate_wall_ms = s.total_wall_ms = 0;
    s.spmm_launches = s.spmm_seed_steps = s.chain_launches = s.seeds_done = s.chain_redo_blocks = 0;
    s.spmm_dense_ms = 0;
    s.spmm_dense_launches = s.spmm_dense_seed_steps = 0;
    return RWR_OK;
}

}  // extern "C"
