"""Config C5 of BASELINE.md section 3 -- 10 M users x 50 M items, 1 G likes (2 G links) -- at FULL size on one MI355X
(the graph fits one GPU's 288 GB; the 8-GPU row-partitioned run of it is the driver's to launch).

Checked here: the size-independent properties of Recommendation (Recommender.cs:14-51), the two bitwise-equal seed-row
kernels against each other, the row-partitioned code path (rwr_part_*, world = 1: one slab = the whole matrix, no
exchange partner) against the seed-sharded path within 1e-9, and ONE seed against the CPU restatement, bitwise
(about a minute of one host core: RWR_TEST_C5_ORACLE=0 skips just that part)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 8           # 8 seeds for the whole job
T = 10
TOP_N = 100


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def c5():
    import torch
    from recommendersystems_amd import synth
    g = synth.config("C5")
    torch.cuda.empty_cache()                    # the generator's device-side sort scratch
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    seeds = synth.seeds_for(g["users"], K, 0, K)
    return g, flat, seeds


def test_c5_full_size_on_one_gpu(c5):
    import recommendersystems_amd as amd
    from recommendersystems_amd.partitioned import PartitionedRecommender
    g, flat, seeds = c5
    U, I = g["users"], g["items"]
    n = U + I
    rp, dst = flat["rowptr"], flat["dst"]

    out = {}
    for kern in ("scan", "fold"):               # two implementations of the same seed-row arithmetic (iterate.hip / chain_scan.hip)
        G = amd.Graph.from_flat(**flat, seed_row_kernel=kern)
        G.buildGraph()
        assert G.stats()["uniform"] == 1 and G.stats()["nnz"] == int(rp[-1])
        rec = amd.Recommender(G)
        out[kern] = rec.RecommendationBatch(seeds, 0.15, T, TOP_N)
        if kern == "scan":
            ids, sc, cnt = out[kern]
            # determinism, order, candidates, exclusion (Recommender.cs:20-38); ids == node index in the synthetic graphs
            ids2, sc2, cnt2 = rec.RecommendationBatch(seeds, 0.15, T, TOP_N)
            assert (ids == ids2).all() and (bits(sc) == bits(sc2)).all() and (cnt == cnt2).all()
            assert (cnt == TOP_N).all()
            for k in range(K):
                key = list(zip((-sc[k]).tolist(), (-ids[k]).tolist()))
                assert key == sorted(key)
                assert (ids[k] >= U).all() and (ids[k] < n).all() and (sc[k] >= 0).all()
                liked = set(dst[rp[seeds[k]]:rp[seeds[k] + 1]].tolist())
                assert not (liked & set(ids[k].tolist()))
            # a batch row equals the single-seed call (the unmodified harness's call shape, Experiment.cs:109)
            one = rec.Recommendation(int(seeds[K // 2]), 0.15, T, TOP_N)
            assert [r[0] for r in one] == ids[K // 2].tolist() and (bits([r[1] for r in one]) == bits(sc[K // 2])).all()
            # rank mass n is conserved
            m = amd.Model(G, float(np.float32(0.15)), int(seeds[1]))
            m.run(T)
            assert abs(m.rank.sum() - n) < 1e-7 * n and (m.rank >= 0).all()
        G.close()
    assert (out["fold"][0] == out["scan"][0]).all() and (bits(out["fold"][1]) == bits(out["scan"][1])).all()
    assert (out["fold"][2] == out["scan"][2]).all()
    ids, sc, cnt = out["scan"]

    # the row-partitioned code path at full size (one slab): same lists, scores to tolerance (tree-summed restart mass)
    pr = PartitionedRecommender(flat, rank=0, world=1)
    pi, ps, pc = pr.RecommendationBatch(seeds, 0.15, T, TOP_N)
    pr.backend.graph.close()
    assert (pi == ids).all() and (pc == cnt).all()
    assert np.abs(ps - sc).max() <= 1e-9 * max(1.0, float(np.abs(sc).max()))

    if os.environ.get("RWR_TEST_C5_ORACLE", "1") != "0":
        from oracle.c_oracle import FlatGraph
        F = FlatGraph(**flat)
        k = K // 2
        oi, os_, oc = F.recommend_batch(seeds[k:k + 1], 0.15, T, TOP_N, n_threads=1)
        assert (oi[0] == ids[k]).all() and (bits(os_[0]) == bits(sc[k])).all() and oc[0] == cnt[k]
