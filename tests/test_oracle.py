"""Pins the oracle: hand-derived KATs, constant pins, and bitwise
agreement of the two independent restatements (literal Python vs flat C), including
dense-restart == sparse-restart.  The reference has no tests/fixtures of its own
(parity unpinned by the reference; see oracle/rwr_oracle.py header)."""
import struct

import numpy as np
import pytest

from oracle import rwr_oracle as po
from oracle.c_oracle import FlatGraph
from tests import graphgen as gg


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def py_graph(g):
    nodes, edges = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"], g["etype"], g["w"])
    G = po.Graph(nodes, edges)
    G.buildGraph()
    return G


def test_constant_pins():
    # SURVEY.md F5 / section 8c
    d = po.widen_float(0.15)
    assert struct.pack(">d", d).hex() == "3fc3333340000000"
    assert struct.pack(">d", 1 - d).hex() == "3feb333330000000"
    inv = 1 / 1.7976931348623157e308          # Model.cs:53: a subnormal (must not be flushed)
    assert 0 < inv < 2.2250738585072014e-308 and inv * 4 == 2.2250738585072014e-308


@pytest.mark.parametrize("dense", [True, False])
def test_kat1(dense):
    G = py_graph(gg.kat1())
    assert [[(l.targetNode, l.weight) for l in G.graph[i]] for i in range(4)] == \
        [[(1, 1.0)], [(0, 0.5), (2, 0.5)], [(1, 0.5), (3, 0.5)], [(2, 1.0)]]
    expect = {1: [2, 2, 0, 0], 2: [2.5, 1, 0.5, 0], 3: [2.25, 1.375, 0.25, 0.125]}
    for T, e in expect.items():
        m = po.Model(G, po.widen_float(0.5), 0, dense_restart=dense)
        m.run(T)
        assert m.rank == e
    rec = po.Recommender(G, dense_restart=dense).Recommendation(0, 0.5, 3)
    assert rec == [(13, 0.125)]          # item 1 (id 11) is LIKEd by the seed -> excluded


@pytest.mark.parametrize("dense", [True, False])
def test_kat2(dense):
    G = py_graph(gg.kat2())
    assert G.graph[1] is None
    for T, e in {1: [1, 1], 2: [1.5, 0.5], 3: [1.25, 0.75]}.items():
        m = po.Model(G, po.widen_float(0.5), 0, dense_restart=dense)
        m.run(T)
        assert m.rank == e
    assert po.Recommender(G).Recommendation(0, 0.5, 3) == []


def test_kats_c_oracle():
    f = FlatGraph(**gg.kat1())
    r, _ = f.model_run(0.5, 0, 0, 3)
    assert list(r) == [2.25, 1.375, 0.25, 0.125]
    ids, sc = f.recommend(0, 0.5, 3)
    assert list(ids) == [13] and list(sc) == [0.125]
    f2 = FlatGraph(**gg.kat2())
    r, _ = f2.model_run(0.5, 0, 0, 3)
    assert list(r) == [1.25, 0.75]
    ids, sc = f2.recommend(0, 0.5, 3)
    assert len(ids) == 0


CASES = [
    dict(seed=1, n_users=12, n_items=30, n_likes=80, n_etc=3, n_friend=10, n_mention=12, n_author=8),
    dict(seed=2, n_users=40, n_items=25, n_likes=300, n_etc=0, n_friend=30, n_mention=20, n_author=10),
    dict(seed=3, n_users=5, n_items=60, n_likes=40, n_etc=2, p_undefined=0.3, n_mention=6),
    dict(seed=4, n_users=30, n_items=90, n_likes=400, uniform=True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"g{c['seed']}")
def test_python_vs_c_bitwise(case):
    g = gg.random_graph(**case)
    G = py_graph(g)
    F = FlatGraph(**g)
    n = F.n
    # buildGraph: normalised weights and dangling flags
    for i in range(n):
        expl = [e for e in range(g["rowptr"][i], g["rowptr"][i + 1]) if g["etype"][e] != 0]
        if G.graph[i] is None:
            assert F.dangling[i] == 1 and not expl
        else:
            assert F.dangling[i] == 0
            assert [l.weight.hex() for l in G.graph[i]] == [float(F.w_norm[e]).hex() for e in expl]
    d = po.widen_float(0.15)
    for seed in (0, n // 3, case["n_users"] - 1):
        for T in (1, 2, 5, 10):
            m_dense = po.Model(G, d, seed, dense_restart=True); m_dense.run(T)
            m_sparse = po.Model(G, d, seed, dense_restart=False); m_sparse.run(T)
            r_c, _ = F.model_run(d, seed, 0, T)
            r_cd, _ = F.model_run(d, seed, 0, T, dense=True)
            assert (bits(m_dense.rank) == bits(m_sparse.rank)).all()
            assert (bits(m_dense.rank) == bits(r_c)).all()
            assert (bits(r_cd) == bits(r_c)).all()
            # mass n is conserved up to rounding
            assert abs(sum(m_dense.rank) - n) < 1e-9 * n
        rec = po.Recommender(G, dense_restart=False).Recommendation(seed, 0.15, 10)
        ids, sc = F.recommend(seed, 0.15, 10)
        assert [r[0] for r in rec] == list(ids)
        assert (bits([r[1] for r in rec]) == bits(sc)).all()
        top = po.Recommender(G, dense_restart=False).Recommendation(seed, 0.15, 10, 7)
        ids7, sc7 = F.recommend(seed, 0.15, 10, 7)
        assert [r[0] for r in top] == list(ids7) == list(ids[:7])
        assert len(po.Recommender(G, dense_restart=False).Recommendation(seed, 0.15, 10, 0)) == len(rec)  # topN<=0 -> all


@pytest.mark.parametrize("case", CASES[:2], ids=lambda c: f"g{c['seed']}")
def test_global_and_convergence(case):
    g = gg.random_graph(**case)
    G = py_graph(g)
    F = FlatGraph(**g)
    d = 0.15
    m = po.Model(G, d); m.run(4)
    r, _ = F.model_run(d, -1, 0, 4, dense=True)
    assert (bits(m.rank) == bits(r)).all()
    m2 = po.Model(G, d, 0, dense_restart=False); it = m2.run(1e-9)
    r2, it2 = F.model_run(d, 0, 1, 1e-9)
    assert it == it2 and (bits(m2.rank) == bits(r2)).all()


def test_batch_matches_single():
    g = gg.random_graph(**CASES[1])
    F = FlatGraph(**g)
    seeds = [0, 3, 7, 11]
    ids, sc, cnt = F.recommend_batch(seeds, 0.15, 6, 5, n_threads=2)
    for k, s in enumerate(seeds):
        i1, s1 = F.recommend(s, 0.15, 6, 5)
        assert cnt[k] == len(i1)
        assert list(ids[k, :cnt[k]]) == list(i1) and (bits(sc[k, :cnt[k]]) == bits(s1)).all()


def test_evaluate_literal_vs_c():
    """Experiment.cs:121-128 restated twice (hits, running-precision sum over the full ranked list)."""
    from oracle.c_oracle import evaluate as c_eval
    g = gg.random_graph(**CASES[1])
    G = py_graph(g)
    rec = po.Recommender(G, dense_restart=False).Recommendation(0, 0.15, 10)
    ids = [r[0] for r in rec]
    rng = np.random.default_rng(5)
    test = set(rng.choice(ids, 7, replace=False).tolist()) | {123456789}      # one id that is not in the list
    h, sp = po.evaluate(rec, test)
    hc, spc = c_eval(ids, sorted(test))
    assert h == hc == 7 and struct.pack(">d", sp) == struct.pack(">d", spc)
    # hand check: hits at 0-based ranks r1 < r2 < ... contribute 1/(r1+1) + 2/(r2+1) + ...
    pos = sorted(ids.index(t) for t in test if t in ids)
    acc = 0.0
    for k, p in enumerate(pos):
        acc += float(k + 1) / (p + 1)
    assert acc == sp
