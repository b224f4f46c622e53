"""Parity at BASELINE.json's full sizes through size-independent properties, on every single-GPU configuration of
BASELINE.md section 3 -- C2 (100 K users x 500 K items, 10 M likes), C3 (MovieLens-25M-shaped: 162 K x 62 K, 25 M likes)
and C4 (1 M x 5 M, 100 M likes: the graph the headline metric is quoted on) -- plus an oracle spot check, bitwise, on a
few seeds of each.  The oracle cannot sweep these sizes in seconds, the properties can.  (C5, the 1 G-like graph: see
tests/test_gpu_c5.py.)"""
import numpy as np
import pytest

from oracle.c_oracle import FlatGraph

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module", params=["C2", "C3", "C4"])
def big(request):
    import recommendersystems_amd as amd
    from recommendersystems_amd import synth
    g = synth.config(request.param)
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}
    G = amd.Graph.from_flat(**flat)
    G.buildGraph()
    assert G.stats()["uniform"] == 1            # unit raw weights (DataLoader.cs:293-294): the value-free path serves these
    yield amd, synth, g, flat, G
    G.close()                                   # the next configuration needs the HBM


def test_full_size_properties(big):
    amd, synth, g, flat, G = big
    U, I = g["users"], g["items"]
    n = U + I
    rec = amd.Recommender(G)
    K = 96                                                    # 6 tiles of 16 (or 3 of 32), heavy and light seeds
    seeds = synth.seeds_for(U, K, 0, K)
    ids, sc, cnt = rec.RecommendationBatch(seeds, 0.15, 10, 100)
    # (1) determinism: a second run is bitwise identical
    ids2, sc2, cnt2 = rec.RecommendationBatch(seeds, 0.15, 10, 100)
    assert (ids == ids2).all() and (bits(sc) == bits(sc2)).all() and (cnt == cnt2).all()
    assert (cnt == 100).all()
    rp, dst = flat["rowptr"], flat["dst"]
    for k in range(K):
        # (2) order: score descending, then id descending (Recommender.cs:35-38)
        key = list(zip((-sc[k]).tolist(), (-ids[k]).tolist()))
        assert key == sorted(key)
        # (3) candidates are ITEM nodes the seed has not LIKEd (Recommender.cs:20-31); ids == node index here
        assert (ids[k] >= U).all() and (ids[k] < n).all()
        liked = set(dst[rp[seeds[k]]:rp[seeds[k] + 1]].tolist())
        assert not (liked & set(ids[k].tolist()))
        assert (sc[k] >= 0).all()
    # (4) a batch row equals the single-seed call, and the top-N overload is a prefix of the full list
    for k in (0, 37, K - 1):
        single = rec.Recommendation(int(seeds[k]), 0.15, 10, 100)
        assert [r[0] for r in single] == ids[k].tolist() and (bits([r[1] for r in single]) == bits(sc[k])).all()
    full = rec.Recommendation(int(seeds[5]), 0.15, 10)
    assert len(full) == I - len(set(dst[rp[seeds[5]]:rp[seeds[5] + 1]].tolist()))
    assert [r[0] for r in full[:100]] == ids[5].tolist()
    fk = [(-r[1], -r[0]) for r in full]
    assert fk == sorted(fk)
    # (5) rank mass n is conserved and ranks are non-negative
    m = amd.Model(G, float(np.float32(0.15)), int(seeds[3]))
    m.run(10)
    assert abs(m.rank.sum() - n) < 1e-7 * n and (m.rank >= 0).all()
    # (6) oracle spot check, bitwise (EXACT mode is the default)
    F = FlatGraph(**flat)
    spot = seeds[[0, K // 2]]
    oi, os_, oc = F.recommend_batch(spot, 0.15, 10, 100, n_threads=2)
    assert (oi == ids[[0, K // 2]]).all() and (bits(os_) == bits(sc[[0, K // 2]])).all()
    r, _ = F.model_run(float(np.float32(0.15)), int(seeds[3]), 0, 10)
    assert (bits(r) == bits(m.rank)).all()


def test_removed_fast_mode_is_rejected(big):
    """Mode 1 (FAST until ABI 3: re-associated sums, no ranking guarantee) is gone: rwr_graph_create refuses it."""
    amd, synth, g, flat, G = big
    from recommendersystems_amd import _lib
    Gf = amd.Graph.from_flat(**flat)
    Gf._opts.mode = 1
    with pytest.raises(Exception):
        Gf.buildGraph()


def test_monotone_id_relabelling_keeps_scores(big):
    """Ids only break ties (Recommender.cs:36-37): an order-preserving relabelling of the ids leaves every score and
    every position unchanged."""
    amd, synth, g, flat, G = big
    flat2 = dict(flat)
    flat2["node_id"] = flat["node_id"] * 3 + 7
    G2 = amd.Graph.from_flat(**flat2)
    G2.buildGraph()
    seeds = synth.seeds_for(g["users"], 32, 0, 32)
    ids, sc, _ = amd.Recommender(G).RecommendationBatch(seeds, 0.15, 10, 50)
    ids2, sc2, _ = amd.Recommender(G2).RecommendationBatch(seeds, 0.15, 10, 50)
    assert (ids2 == ids * 3 + 7).all() and (bits(sc) == bits(sc2)).all()
    G2.close()


def test_seed_row_kernels_agree_at_full_size(big):
    """The sequential fold and the parallel binade scan of the seed's own row are two implementations of the same
    arithmetic: at full size their results must be identical bit for bit (and the scan must have carried most blocks
    without redoing them)."""
    amd, synth, g, flat, G = big
    seeds = synth.seeds_for(g["users"], 40, 0, 40)
    out = {}
    for kern in ("fold", "scan"):
        Gk = amd.Graph.from_flat(**flat, seed_row_kernel=kern, profile=True)
        Gk.buildGraph()
        out[kern] = amd.Recommender(Gk).RecommendationBatch(seeds, 0.15, 10, 100)
        if kern == "scan":
            n = g["users"] + g["items"]
            blocks = -(-n * Gk.stats()["tile_seeds"] // 4096) * len(seeds) * 10
            assert 0 < Gk.stats()["chain_redo_blocks"] < 0.1 * blocks
        Gk.close()
    assert (out["fold"][0] == out["scan"][0]).all() and (bits(out["fold"][1]) == bits(out["scan"][1])).all()
    assert (out["fold"][2] == out["scan"][2]).all()
