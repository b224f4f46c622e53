"""Harness-shaped load for profiling: build an ego-network-sized graph and run one Recommendation, many times
(rocprofv3 --hip-trace --kernel-trace --stats shows where the per-graph millisecond goes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystems_amd import synth, _lib                            # noqa: E402
if os.environ.get("RWR_TOOLS_EXP_LIB"):        # the experiments build (make -C recommendersystems_amd/csrc exp): RWR_BUILD_TIMING
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librwr_exp.so")
from recommendersystems_amd.rwr_based import Graph, Recommender           # noqa: E402

g = synth.bipartite(9, 60, 2000, 4000)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tb = tr = 0.0
for rep in range(N + 5):
    t0 = time.perf_counter()
    G = Graph.from_flat(**flat)
    G.buildGraph()
    t1 = time.perf_counter()
    Recommender(G).RecommendationArrays(0, 0.15, 10)
    t2 = time.perf_counter()
    G.close()
    if rep >= 5:
        tb += t1 - t0
        tr += t2 - t1
print(f"n={len(g['node_id'])} nnz={len(g['dst'])}: buildGraph {tb / N * 1e6:.0f} us, Recommendation {tr / N * 1e6:.0f} us per graph", flush=True)
