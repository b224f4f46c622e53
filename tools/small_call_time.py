"""Wall time of the C-ABI single-seed call on an ego-network-sized graph, by iteration count (fixed cost vs per-iteration
cost of the one-launch kernel of small.hip; RWR_SMALL=0 gives the general path for comparison)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystems_amd import synth                                  # noqa: E402
from recommendersystems_amd.rwr_based import Graph, Recommender           # noqa: E402

for (U, I, E) in ((60, 2000, 4000), (500, 3500, 30000), (1500, 4000, 60000)):
    g = synth.bipartite(9, U, I, E)
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    G = Graph.from_flat(**flat)
    G.buildGraph()
    rec = Recommender(G)
    out = []
    for T in (0, 1, 10, 20):
        rec.RecommendationArrays(0, 0.15, T)
        t = time.perf_counter()
        for _ in range(20):
            rec.RecommendationArrays(0, 0.15, T)
        out.append(f"T={T}: {(time.perf_counter() - t) / 20 * 1e6:.0f} us")
    t = time.perf_counter()
    for _ in range(20):
        rec.RecommendationArrays(0, 0.15, 10, 100)
    out.append(f"T=10 top-100: {(time.perf_counter() - t) / 20 * 1e6:.0f} us")
    print(f"n={U + I} nnz={len(g['dst'])} RWR_SMALL={os.environ.get('RWR_SMALL', '1')}: " + ", ".join(out), flush=True)
    G.close()
