"""Property-based checks: random small graphs with every feature the reference's containers allow --
multi-edges, self-links, UNDEFINED links, zero/odd weights, arbitrary signed 64-bit ids, nodes of every type --
run through both oracle restatements (CPU) and through the HIP path (GPU, bitwise in EXACT mode)."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import rwr_oracle as po
from oracle.c_oracle import FlatGraph


@st.composite
def graphs(draw):
    n = draw(st.integers(1, 14))
    ids = draw(st.lists(st.integers(-2 ** 63, 2 ** 63 - 1), min_size=n, max_size=n, unique=True))
    types = draw(st.lists(st.integers(0, 3), min_size=n, max_size=n))
    rowptr = [0]
    dst, etype, w = [], [], []
    for _ in range(n):
        deg = draw(st.integers(0, 5))
        for _ in range(deg):
            dst.append(draw(st.integers(0, n - 1)))
            etype.append(draw(st.sampled_from([0, 1, 1, 2, 3, 4, 5])))
            w.append(draw(st.sampled_from([1.0, 1.0, 0.5, 3.0, 0.1, 2.718281828459045, 1e-3, 7.0])))
        rowptr.append(len(dst))
    seed = draw(st.integers(0, n - 1))
    T = draw(st.integers(0, 6))
    return dict(node_id=np.array(ids, dtype=np.int64), node_type=np.array(types, dtype=np.uint8),
                rowptr=np.array(rowptr, dtype=np.int64), dst=np.array(dst, dtype=np.int32),
                etype=np.array(etype, dtype=np.uint8), w=np.array(w, dtype=np.float64)), seed, T


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def literal(g, seed, T, dense=True):
    nodes, edges = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"], g["etype"], g["w"])
    G = po.Graph(nodes, edges)
    G.buildGraph()
    m = po.Model(G, po.widen_float(0.15), seed, dense_restart=dense)
    m.run(T)
    rec = po.Recommender(G, dense_restart=dense).Recommendation(seed, 0.15, T)
    return m.rank, rec


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(graphs())
def test_two_restatements_agree_bitwise(case):
    g, seed, T = case
    rank, rec = literal(g, seed, T, dense=True)
    rank_s, rec_s = literal(g, seed, T, dense=False)
    assert (bits(rank) == bits(rank_s)).all() and rec == rec_s
    F = FlatGraph(**g)
    r, _ = F.model_run(po.widen_float(0.15), seed, 0, T)
    ids, sc = F.recommend(seed, 0.15, T)
    assert (bits(r) == bits(rank)).all()
    assert ids.tolist() == [x[0] for x in rec] and (bits(sc) == bits([x[1] for x in rec])).all()
    keys = [(-x[1], -x[0]) for x in rec]
    assert keys == sorted(keys)                                            # (score desc, id desc)


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("RWR_HYP_EXAMPLES", "60")), deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(graphs())
def test_hip_path_matches_literal_oracle_bitwise(case):
    import recommendersystems_amd as amd
    g, seed, T = case
    rank, rec = literal(g, seed, T, dense=True)
    G = amd.Graph.from_flat(**g)
    G.buildGraph()
    try:
        m = amd.Model(G, po.widen_float(0.15), seed)
        m.run(T)
        assert (bits(m.rank) == bits(rank)).all()
        got = amd.Recommender(G).Recommendation(seed, 0.15, T)
        assert [x[0] for x in got] == [x[0] for x in rec]
        assert (bits([x[1] for x in got]) == bits([x[1] for x in rec])).all()
        n = len(g["node_id"])
        seeds = np.array([seed, (seed + 1) % n, 0], dtype=np.int32)
        ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, T, 5)
        assert cnt[0] == min(5, len(rec)) and ids[0, :cnt[0]].tolist() == [x[0] for x in rec[:5]]
        assert (bits(sc[0, :cnt[0]]) == bits([x[1] for x in rec[:5]])).all()
    finally:
        G.close()


@st.composite
def patched_graphs(draw):
    g, seed, T = draw(graphs())
    m = len(g["dst"])
    k = draw(st.integers(0, min(m, 6)))
    idx = draw(st.lists(st.integers(0, max(m - 1, 0)), min_size=k, max_size=k, unique=True)) if m else []
    new_t = [draw(st.sampled_from([0, 1, 2, 5])) for _ in idx]
    new_w = [draw(st.sampled_from([1.0, 0.25, 4.0, 1e-3])) for _ in idx]
    return g, seed, T, idx, new_t, new_w


@pytest.mark.gpu
@settings(max_examples=int(os.environ.get("RWR_HYP_EXAMPLES", "60")), deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(patched_graphs())
def test_hip_incremental_rebuild_matches_literal_oracle(case):
    """rwr_graph_update_links on random graphs: after patching a few links' types and weights the device state must be the
    literal restatement's on the patched containers (ranks bitwise, ranked list identical)."""
    import recommendersystems_amd as amd
    g, seed, T, idx, new_t, new_w = case
    G = amd.Graph.from_flat(**g)
    G.buildGraph()
    try:
        et = g["etype"].copy()
        w = g["w"].copy()
        for p, t, x in zip(idx, new_t, new_w):
            et[p] = t
            w[p] = x
        G.updateLinks(np.array(idx, dtype=np.int64), etype=np.array(new_t, dtype=np.uint8), w=np.array(new_w, dtype=np.float64))
        g2 = dict(g, etype=et, w=w)
        rank, rec = literal(g2, seed, T, dense=False)
        m = amd.Model(G, po.widen_float(0.15), seed)
        m.run(T)
        assert (bits(m.rank) == bits(rank)).all()
        got = amd.Recommender(G).Recommendation(seed, 0.15, T)
        assert [x[0] for x in got] == [x[0] for x in rec]
        assert (bits([x[1] for x in got]) == bits([x[1] for x in rec])).all()
    finally:
        G.close()
