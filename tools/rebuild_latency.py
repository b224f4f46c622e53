"""Cost of a full rwr_graph_create (host arrays -> PCIe -> device build) against rwr_graph_update_links (patch a few
links on the device + device build) -- SURVEY.md 8f-2: the harness rebuilds an almost identical graph per fold."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd import synth
from recommendersystems_amd.rwr_based import Graph, Recommender

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
no, U, I, E, _ = synth.CONFIGS[cfg]
g = synth.bipartite(no, U, I, E)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
m = int(g["rowptr"][-1])
t = time.perf_counter(); G = Graph.from_flat(**flat, profile=True); G.buildGraph(); t_create = time.perf_counter() - t
b_create = G.stats()["build_ms"]
rng = np.random.default_rng(1)
idx = np.unique(rng.integers(0, m, m // 100)).astype(np.int64)        # relabel ~1 % of the links
et = np.zeros(len(idx), dtype=np.uint8)
t = time.perf_counter(); G.updateLinks(idx, etype=et); t_upd = time.perf_counter() - t
b_upd = G.stats()["build_ms"]
t = time.perf_counter(); G.updateLinks(idx, etype=np.ones(len(idx), dtype=np.uint8)); t_back = time.perf_counter() - t
r = Recommender(G).RecommendationBatch(synth.seeds_for(U, 8, 0, 8), 0.15, 10, 10)
print(f"{cfg}: {m} raw links; create {t_create*1e3:.0f} ms (device build {b_create:.1f} ms); update of {len(idx)} links "
      f"{t_upd*1e3:.0f} ms (device build {b_upd:.1f} ms); revert {t_back*1e3:.0f} ms; bytes over PCIe {m*13+8*(U+I)*2:,} vs {len(idx)*9:,}")
