"""Latency of the single-seed entry points (the path the unmodified C# harness uses): Recommendation(seed, d, T)
with the full ranked list, and Model.run(T)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd import synth
from recommendersystems_amd.rwr_based import Graph, Recommender, Model
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else "exact"
no, U, I, E, K = synth.CONFIGS[cfg]
g = synth.bipartite(no, U, I, E)
flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
G = Graph.from_flat(**flat, profile=True, mode=mode); G.buildGraph()
n = U + I; nnz = int(g["rowptr"][-1])
# value-free path (uniform row weights): no per-entry value, + the n*8-byte scale vector
uni = bool(G.stats()["uniform_path"])
alg = synth.algorithmic_bytes_per_step(n, nnz, 1, 0 if uni else 8) + (8 * n if uni else 0)
rec = Recommender(G)
for seed in (0, U // 2):
    rec.Recommendation(seed, 0.15, 10, 100)
    G.reset_stats()
    t = time.perf_counter(); r = rec.Recommendation(seed, 0.15, 10, 100); t1 = time.perf_counter() - t
    st = G.stats()
    t = time.perf_counter(); r2 = rec.Recommendation(seed, 0.15, 10); t2 = time.perf_counter() - t
    m = Model(G, float(np.float32(0.15)), seed)
    t = time.perf_counter(); m.run(10); t3 = time.perf_counter() - t
    print(f"{cfg} {mode} K=1 SpMV: {st['spmm_ms']/10:.3f} ms per step = {alg/ (st['spmm_ms']/10*1e-3)/1e9:.0f} GB/s algorithmic ({alg/1e6:.0f} MB/step) = {alg/(st['spmm_ms']/10*1e-3)/8e12*100:.1f} % of 8 TB/s")
    print(f"{cfg} seed {seed}: top-100 {t1*1e3:.1f} ms (spmm {st['spmm_ms']:.1f} chain {st['chain_ms']:.1f} rank {st['rank_ms']:.1f}), "
          f"full list ({len(r2)}) {t2*1e3:.1f} ms, Model.run(10) {t3*1e3:.1f} ms")
