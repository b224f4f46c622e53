"""Blocks the seed-row carry redoes per single-seed call (stats.chain_redo_blocks); with the experiments build and
RWR_X_CARRY_DBG=1 the carry kernel prints where its time goes."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import recommendersystems_amd._lib as _lib
if os.environ.get("RWR_TOOLS_EXP_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "librwr_exp.so")
from recommendersystems_amd import synth
from recommendersystems_amd.rwr_based import Graph, Recommender
for cfg in sys.argv[1:] or ("C2", "C3"):
    no, U, I, E, K = synth.CONFIGS[cfg]
    g = synth.bipartite(no, U, I, E)
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    G = Graph.from_flat(**flat, profile=True); G.buildGraph()
    rec = Recommender(G)
    for seed in (0, U // 2):
        rec.Recommendation(seed, 0.15, 10, 100)
        G.reset_stats()
        rec.Recommendation(seed, 0.15, 10, 100)
        st = G.stats()
        print(cfg, seed, "redo blocks", st["chain_redo_blocks"], "chain ms", st["chain_ms"], "launches", st["chain_launches"], flush=True)
