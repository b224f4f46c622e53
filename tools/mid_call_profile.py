"""One mid-size ego network (8k / 12k / 120k nodes: beyond the one-launch path of small.hip), the single-seed call repeated
(for rocprofv3 --kernel-trace --stats).   python tools/mid_call_profile.py 12k"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendersystems_amd import synth, _lib                            # noqa: E402
if os.environ.get("RWR_TOOLS_EXP_LIB"):        # the experiments build (make -C recommendersystems_amd/csrc exp)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librwr_exp.so")
from recommendersystems_amd.rwr_based import Graph, Recommender           # noqa: E402

SIZES = {"8k": (1500, 6000, 60000), "12k": (2000, 10000, 100000), "120k": (20000, 100000, 1000000)}
U, I, E = SIZES[sys.argv[1] if len(sys.argv) > 1 else "12k"]
g = synth.bipartite(9, U, I, E)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
G = Graph.from_flat(**flat)
G.buildGraph()
rec = Recommender(G)
N = int(os.environ.get("MID_CALLS", "50"))
rec.RecommendationArrays(0, 0.15, 10)
t = time.perf_counter()
for _ in range(N):
    rec.RecommendationArrays(0, 0.15, 10)
print(f"n={U + I}: call {(time.perf_counter() - t) / N * 1e6:.0f} us", flush=True)
