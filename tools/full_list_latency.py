"""C-ABI-level latency of rwr_recommend with the FULL ranked list (top_n <= 0), i.e. the unmodified harness call."""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd import synth, _lib
from recommendersystems_amd.rwr_based import Graph
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
no, U, I, E, K = synth.CONFIGS[cfg]
g = synth.bipartite(no, U, I, E)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
G = Graph.from_flat(**flat, profile=True); G.buildGraph()
lib = _lib.load()
n = U + I
ids = np.zeros(n, dtype=np.int64); sc = np.zeros(n, dtype=np.float64)
for rep in range(3):
    cnt = C.c_int64(n)
    G.reset_stats()
    t = time.perf_counter()
    _lib.check(lib.rwr_recommend(G._handle(), 0, C.c_float(0.15), 10, 0, ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                 sc.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt)))
    dt = time.perf_counter() - t
    st = G.stats()
print(f"{cfg}: rwr_recommend full list ({cnt.value} items) {dt*1e3:.1f} ms: spmm {st['spmm_ms']:.1f} chain {st['chain_ms']:.1f} rank {st['rank_ms']:.1f} ms")
