"""Child process of tests/test_gpu_env_paths.py: librwr reads its experiment knobs from the environment ONCE per process, so
code paths selected by them (RWR_VALUE_FREE, RWR_BIG_N, RWR_SPMV_PHASES, RWR_ACT_ITERS, ...) are reached by starting a fresh
interpreter with the variables set.  Compares the HIP path with the C restatement of the reference on seeded graphs:
bitwise.  Prints ENV_CHILD_OK <cases> on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import recommendersystems_amd as amd                    # noqa: E402
from oracle.c_oracle import FlatGraph                   # noqa: E402
from tests import graphgen as gg                        # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def row_uniform(g, seed):
    """Every row's explicit weights equal, but different from row to row (Graph.cs:79-81 then gives every link of a row
    the same normalised weight): still the value-free path."""
    rng = np.random.default_rng(seed)
    g = dict(g)
    w = np.array(g["w"], dtype=np.float64)
    rp = g["rowptr"]
    per_row = rng.choice([0.25, 1.0, 3.0, 0.1, 7.5], size=len(rp) - 1)
    for i in range(len(rp) - 1):
        w[rp[i]:rp[i + 1]] = per_row[i]
    g["w"] = w
    return g


def graphs():
    yield "uniform-small", gg.random_graph(4, n_users=30, n_items=90, n_likes=400, uniform=True), True
    yield "uniform-medium", gg.random_graph(12, n_users=3000, n_items=1000, n_likes=60000, uniform=True), True
    # uniform rows with UNDEFINED links and dangling nodes in between, weights differing from row to row
    g = gg.random_graph(21, n_users=400, n_items=1500, n_likes=9000, n_etc=10, n_friend=300, n_author=100, p_undefined=0.2)
    yield "row-uniform-undefined", row_uniform(g, 5), True
    # dense uniform graphs (tens of links per node) with hub rows (>= 1024 in-links): several LDS blocks of the sweep (sweep.hip)
    from recommendersystems_amd import synth
    # ... and 50 user rows of several thousand in-links each: the hub kernel of the exact single-seed SpMV (spmv.hip:
    # k_spmv_exact_hub, rows of >= 2048 in-links summed by the exact parallel reduction of pf.h)
    for name, (no, U, I, E) in {"dense-2-blocks": (7, 3000, 9000, 400_000), "dense-3-blocks": (8, 4000, 16500, 600_000),
                                "hub-rows": (12, 50, 20000, 300_000)}.items():
        sg = synth.bipartite(no, U, I, E)
        yield name, {k: sg[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}, True
    # 40 user rows of several thousand links each: the build's long-row path (build.hip: 64 rows holding more than 8 tiles
    # of links) and the hub kernel in its WEIGHTED form, with weights that differ inside a row and UNDEFINED links in between
    sg = synth.bipartite(11, 40, 7000, 200_000)
    rng = np.random.default_rng(3)
    lg = {k: sg[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    lg["w"] = rng.choice([0.5, 1.0, 2.0, 3.25], size=len(lg["w"]))
    lg["etype"] = np.where(rng.random(len(lg["etype"])) < 0.1, 0, 1).astype(np.uint8)
    yield "long-rows-weighted", lg, False
    yield "mixed-weights", gg.random_graph(11, n_users=700, n_items=2500, n_likes=20000, n_etc=20, n_friend=800,
                                           n_mention=500, n_author=300), False


def main():
    d32 = float(np.float32(0.15))
    cases = 0
    for name, g, expect_uniform in graphs():
        F = FlatGraph(**g)
        n_users = int((g["node_type"] == 1).sum())
        for mode in ("exact",):
            G = amd.Graph.from_flat(**g, mode=mode)
            G.buildGraph()
            st = G.stats()
            assert st["uniform"] == (1 if expect_uniform else 0), (name, st)
            want_vf = expect_uniform and os.environ.get("RWR_VALUE_FREE", "1") != "0"
            assert st["uniform_path"] == (1 if want_vf else 0), (name, st)
            rec = amd.Recommender(G)

            def check(got_ids, got_sc, ref_ids, ref_sc, what):
                assert np.array_equal(np.asarray(got_ids), np.asarray(ref_ids)), (name, mode, what, "ids differ")
                assert (bits(got_sc) == bits(ref_sc)).all(), (name, mode, what, "scores not bitwise equal")

            big = len(g["node_id"]) > 10000            # (the larger graphs: fewer single-seed cases, same code paths)
            for seed in ((0, n_users - 1) if big else (0, 7, n_users - 1)):
                for T in ((0, 2, 10) if big else (0, 1, 2, 4, 10)):
                    got = rec.Recommendation(seed, 0.15, T)
                    ri, rs = F.recommend(seed, 0.15, T)
                    check([r[0] for r in got], [r[1] for r in got], ri, rs, f"single seed {seed} T {T}")
                    cases += 1
            for K in ((3, 130) if big else (3, 40, 130)):
                seeds = (np.arange(K, dtype=np.int64) * n_users // K).astype(np.int32)
                bi, bs, bc = rec.RecommendationBatch(seeds, 0.15, 10, 20)
                oi, os_, oc = F.recommend_batch(seeds, 0.15, 10, 20)
                assert (bc == oc).all()
                check(bi, bs, oi, os_, f"batch {K}")
                cases += 1
            for seed in (1, n_users // 2):
                m = amd.Model(G, d32, seed)
                m.run(6)
                r, _ = F.model_run(d32, seed, 0, 6)
                if mode == "exact":
                    assert (bits(m.rank) == bits(r)).all(), (name, "model run")
                else:
                    assert np.abs(m.rank - r).max() <= 1e-6
                # step by step through the public API, continuing from the advanced state (host-held rank vector)
                m.deliverRanks()
                m.updateRanks()
                r7, _ = F.model_run(d32, seed, 0, 7)
                if mode == "exact":
                    assert (bits(m.rank) == bits(r7)).all(), (name, "model stepwise")
                else:
                    assert np.abs(m.rank - r7).max() <= 1e-6
                cases += 1
            if mode == "exact":
                m = amd.Model(G, d32, 2)
                m.run(1e-3)
                r, it = F.model_run(d32, 2, 1, 1e-3)
                assert m.iterations == it and (bits(m.rank) == bits(r)).all(), (name, "threshold run")
                mg = amd.Model(G, d32)                     # global model: tolerance parity
                mg.run(4)
                rg, _ = F.model_run(d32, -1, 0, 4)
                assert np.abs(mg.rank - rg).max() <= 1e-12 * max(1.0, np.abs(rg).max()), (name, "global model")
                cases += 2
            G.close()
    print("ENV_CHILD_OK", cases)


if __name__ == "__main__":
    main()
