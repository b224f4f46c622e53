"""Latency of the harness-shaped use: build a small graph, one Recommendation (full list), repeatedly."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd import synth
from recommendersystems_amd.rwr_based import Graph, Recommender
for (U, I, E) in ((60, 2000, 4000), (2000, 10000, 100000), (20000, 100000, 1000000)):
    g = synth.bipartite(9, U, I, E)
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    tb, tr = [], []
    for rep in range(6):
        t = time.perf_counter(); G = Graph.from_flat(**flat); G.buildGraph(); tb.append(time.perf_counter() - t)
        rec = Recommender(G)
        t = time.perf_counter(); r = rec.Recommendation(0, 0.15, 10); tr.append(time.perf_counter() - t)
        t = time.perf_counter(); r = rec.Recommendation(0, 0.15, 10); tr.append(time.perf_counter() - t)
        G.close()
    print(f"n={U+I} nnz={len(g['dst'])}: buildGraph {min(tb)*1e3:.2f} ms (first {tb[0]*1e3:.1f}), Recommendation first-call {min(tr[0::2])*1e3:.2f} ms, warm {min(tr[1::2])*1e3:.2f} ms")
