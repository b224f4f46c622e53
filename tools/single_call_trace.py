"""One configuration, the single-seed top-100 call repeated (for rocprofv3 --kernel-trace --hip-trace --stats)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystems_amd import synth                                  # noqa: E402
from recommendersystems_amd.rwr_based import Graph, Recommender           # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
no, U, I, E, K = synth.CONFIGS[cfg]
g = synth.bipartite(no, U, I, E)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
G = Graph.from_flat(**flat)
G.buildGraph()
rec = Recommender(G)
rec.RecommendationArrays(U // 2, 0.15, 10, 100)
N = 20
t = time.perf_counter()
for _ in range(N):
    rec.RecommendationArrays(U // 2, 0.15, 10, 100)
print(f"{cfg}: call {(time.perf_counter() - t) / N * 1e3:.2f} ms", flush=True)
