"""Where a rwr_eval_graphs call spends its time: Python marshalling vs the C call, by batch size (synthetic ego-network-sized
graphs of tests/graphgen.py; `python tools/eval_graphs_profile.py`)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C                                                        # noqa: E402
from recommendersystems_amd import _lib, rwr_based as rb                  # noqa: E402
if os.environ.get("RWR_TOOLS_EXP_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librwr_exp.so")
from tests import graphgen as gg                                          # noqa: E402

rng = np.random.default_rng(5)
graphs = []
for k in range(192):
    g = gg.random_graph(500 + k, n_users=int(rng.integers(100, 300)), n_items=int(rng.integers(600, 2100)),
                        n_likes=int(rng.integers(1500, 6000)), n_friend=200, n_mention=100, n_author=50)
    graphs.append(rb.Graph.from_flat(**g))
tests = [[int(x) for x in rng.integers(1000, 9000, 30)] for _ in graphs]
lib = _lib.load()
real = lib.rwr_eval_graphs
t_c = [0.0]


class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter()
        r = real(*a)
        t_c[0] += time.perf_counter() - t0
        return r


lib.rwr_eval_graphs = Timed()
for nb in (1, 8, 48, 192):
    gs, ts = graphs[:nb], tests[:nb]
    rb.EvaluateGraphs(gs, [0] * nb, 0.15, 10, ts)
    t_c[0] = 0.0
    t = time.perf_counter()
    reps = max(1, 192 // nb)
    for _ in range(reps):
        rb.EvaluateGraphs(gs, [0] * nb, 0.15, 10, ts)
    dt = (time.perf_counter() - t) / reps
    print(f"batch {nb:4d}: {dt * 1e3:8.3f} ms per call = {dt / nb * 1e6:7.1f} us per graph = {nb / dt:9.0f} graphs/s; inside the C call {t_c[0] / reps * 1e3:8.3f} ms", flush=True)

# ---- scaling over host threads with the marshalling done beforehand (the C call releases the GIL)
import threading                                                          # noqa: E402

lib.rwr_eval_graphs = real


def marshal(gs, ts):
    K = len(gs)
    descs = (_lib.rwr_graph_desc * K)()
    keep = []
    for k, g in enumerate(gs):
        node_id, node_type, rowptr, dst, etype, w = g._flat
        descs[k] = _lib.rwr_graph_desc(int(node_id.shape[0]), 0, rb._p(node_id, C.c_int64), rb._p(node_type, C.c_uint8),
                                       rb._p(rowptr, C.c_int64), rb._p(dst, C.c_int32), rb._p(etype, C.c_uint8), rb._p(w, C.c_double))
    seeds = np.zeros(K, dtype=np.int32)
    ptr = np.zeros(K + 1, dtype=np.int64)
    for k, t in enumerate(ts):
        ptr[k + 1] = ptr[k] + len(t)
    ids = np.ascontiguousarray([x for t in ts for x in t], dtype=np.int64)
    hits = np.zeros(K, dtype=np.int64); sp = np.zeros(K, dtype=np.float64); ln = np.zeros(K, dtype=np.int64)
    opts = _lib.rwr_opts(C.sizeof(_lib.rwr_opts), -1, -1, 0, 0, 0, 0, 0, 0)
    keep = (descs, seeds, ptr, ids, hits, sp, ln, opts)
    args = (K, descs, rb._p(seeds, C.c_int32), C.c_float(0.15), 10, rb._p(ptr, C.c_int64), rb._p(ids, C.c_int64), C.byref(opts),
            rb._p(hits, C.c_int64), rb._p(sp, C.c_double), rb._p(ln, C.c_int64))
    return args, keep


for nt in (1, 2, 4, 10):
    per = 19
    packs = [marshal(graphs[t * per:(t + 1) * per], tests[t * per:(t + 1) * per]) for t in range(nt)]
    reps = 20

    def work(t):
        for _ in range(reps):
            rc = real(*packs[t][0])
            assert rc == 0

    for t in range(nt):
        real(*packs[t][0])
    th = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print(f"threads {nt:2d} x {per} graphs per call: {nt * per * reps / dt:9.0f} graphs/s ({dt / reps * 1e3:.2f} ms per round of calls)", flush=True)
