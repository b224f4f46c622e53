"""Parity tests proper: the HIP path, called through the C-ABI (ctypes), against the oracle
on the same seeded inputs.  Every rank value and score must be bitwise the oracle's, every ranked
list identical (north_star asks for identical top-k lists and scores within 1e-6: bitwise implies both)."""
import ctypes as C

import numpy as np
import pytest

from oracle import rwr_oracle as po
from oracle.c_oracle import FlatGraph
from tests import graphgen as gg

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def amd():
    import recommendersystems_amd as amd
    from recommendersystems_amd import _lib
    assert _lib.load().rwr_device_count() >= 1, "no gfx950 device: the HIP path cannot run"
    return amd


def dev_graph(amd, g, **opts):
    G = amd.Graph.from_flat(**g, **opts)
    G.buildGraph()
    return G


SMALL = [
    dict(seed=1, n_users=12, n_items=30, n_likes=80, n_etc=3, n_friend=10, n_mention=12, n_author=8),
    dict(seed=2, n_users=40, n_items=25, n_likes=300, n_etc=0, n_friend=30, n_mention=20, n_author=10),
    dict(seed=3, n_users=5, n_items=60, n_likes=40, n_etc=2, p_undefined=0.3, n_mention=6),
    dict(seed=4, n_users=30, n_items=90, n_likes=400, uniform=True),
]
MEDIUM = [
    dict(seed=11, n_users=700, n_items=2500, n_likes=20000, n_etc=20, n_friend=800, n_mention=500, n_author=300),
    dict(seed=12, n_users=3000, n_items=1000, n_likes=60000, uniform=True),
]


def test_kats_via_mirror_classes(amd):
    """SURVEY.md section 8c KATs through the reference-shaped classes (dict containers)."""
    for kat, seed_rank, rec_expect in ((gg.kat1(), [2.25, 1.375, 0.25, 0.125], [(13, 0.125)]),
                                       (gg.kat2(), [1.25, 0.75], [])):
        nodes = {i: amd.Node(int(kat["node_id"][i]), amd.NodeType(int(kat["node_type"][i])))
                 for i in range(len(kat["node_id"]))}
        edges = {i: [amd.ForwardLink(int(kat["dst"][e]), amd.EdgeType.LIKE, 1.0)
                     for e in range(kat["rowptr"][i], kat["rowptr"][i + 1])] for i in range(len(nodes))}
        graph = amd.Graph(nodes, edges)
        graph.buildGraph()
        model = amd.Model(graph, 0.5, 0)
        model.run(3)
        assert model.rank.tolist() == seed_rank
        assert amd.Recommender(graph).Recommendation(0, 0.5, 3) == rec_expect


@pytest.mark.parametrize("case", SMALL + MEDIUM, ids=lambda c: f"g{c['seed']}")
def test_build_graph_bitwise(amd, case):
    g = gg.random_graph(**case)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    wn, dg = G.normalized()
    assert (bits(wn) == bits(F.w_norm)).all()
    assert (dg == F.dangling).all()
    st = G.stats()
    assert st["nnz"] == int((g["etype"] != 0).sum())
    assert st["uniform"] == (1 if case.get("uniform") else 0) or not case.get("uniform")


@pytest.mark.parametrize("case", SMALL, ids=lambda c: f"g{c['seed']}")
@pytest.mark.parametrize("tile_seeds", [0, 1, 4, 16, 64])
@pytest.mark.parametrize("seed_row_kernel", ["fold", "scan"])
def test_small_exact_bitwise_vs_literal_python(amd, case, tile_seeds, seed_row_kernel):
    g = gg.random_graph(**case)
    nodes, edges = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"], g["etype"], g["w"])
    PG = po.Graph(nodes, edges)
    PG.buildGraph()
    G = dev_graph(amd, g, tile_seeds=tile_seeds, seed_row_kernel=seed_row_kernel)
    rec = amd.Recommender(G)
    n = len(nodes)
    for seed in (0, n // 3, case["n_users"] - 1):
        for T in (0, 1, 2, 5, 10):
            m = po.Model(PG, po.widen_float(0.15), seed, dense_restart=True)
            m.run(T)
            dm = amd.Model(G, po.widen_float(0.15), seed)
            dm.run(T)
            assert (bits(dm.rank) == bits(m.rank)).all(), (seed, T)
        ref = po.Recommender(PG).Recommendation(seed, 0.15, 10)
        got = rec.Recommendation(seed, 0.15, 10)
        assert [r[0] for r in got] == [r[0] for r in ref]
        assert (bits([r[1] for r in got]) == bits([r[1] for r in ref])).all()
        assert rec.Recommendation(seed, 0.15, 10, 7) == got[:7]
        assert rec.Recommendation(seed, 0.15, 10, 0) == got          # topN <= 0 -> whole list
        assert rec.Recommendation(seed, 0.15, 10, 10 ** 6) == got
        ai, asc = rec.RecommendationArrays(seed, 0.15, 10)
        assert ai.tolist() == [r[0] for r in got] and asc.tolist() == [r[1] for r in got]


@pytest.mark.parametrize("case", MEDIUM, ids=lambda c: f"g{c['seed']}")
def test_medium_batch_vs_c_oracle(amd, case):
    g = gg.random_graph(**case)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G)
    K = 70
    seeds = np.linspace(0, case["n_users"] - 1, K).astype(np.int32)
    ids, sc, cnt = rec.RecommendationBatch(seeds, 0.15, 10, 25)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 10, 25)
    assert (cnt == oc).all()
    assert (ids == oi).all()                       # top-k id lists identical
    assert (bits(sc) == bits(os_)).all()           # (north_star asks for 1e-6; the engine is bitwise)
    # single-seed full list == batch prefix
    full = rec.Recommendation(int(seeds[5]), 0.15, 10)
    assert [r[0] for r in full[:cnt[5]]] == ids[5, :cnt[5]].tolist()
    fi, fs = F.recommend(int(seeds[5]), 0.15, 10)
    assert [r[0] for r in full] == fi.tolist()
    assert (bits([r[1] for r in full]) == bits(fs)).all()


def test_properties_exact(amd):
    g = gg.random_graph(21, n_users=400, n_items=1500, n_likes=9000, n_friend=300, n_mention=200)
    G = dev_graph(amd, g)
    n = len(g["node_id"])
    m = amd.Model(G, po.widen_float(0.15), 5)
    m.run(10)
    assert abs(m.rank.sum() - n) < 1e-9 * n                      # mass n conserved
    rec = amd.Recommender(G).Recommendation(5, 0.15, 10)
    liked = {int(g["dst"][e]) for e in range(g["rowptr"][5], g["rowptr"][6]) if g["etype"][e] == gg.EDGE_LIKE}
    liked_ids = {int(g["node_id"][j]) for j in liked}
    assert not (liked_ids & {r[0] for r in rec})                 # exclusion list never appears
    keys = [(-r[1], -r[0]) for r in rec]
    assert keys == sorted(keys)                                  # (score desc, id desc)
    n_items = int((g["node_type"] == gg.NODE_ITEM).sum())
    assert len(rec) == n_items - len({j for j in liked if g["node_type"][j] == gg.NODE_ITEM})


def test_error_behaviour(amd):
    from recommendersystems_amd import _lib
    g = gg.kat1()
    G = dev_graph(amd, g)
    with pytest.raises(amd.RwrError) as ei:
        amd.Recommender(G).Recommendation(99, 0.5, 3)
    assert ei.value.status == _lib.RWR_E_RANGE
    bad = dict(g)
    bad["dst"] = g["dst"].copy()
    bad["dst"][0] = 77
    with pytest.raises(amd.RwrError) as ei:
        dev_graph(amd, bad)
    assert ei.value.status == _lib.RWR_E_RANGE
    # KeyNotFoundException analogue on the dictionary form (Recommender.cs:21)
    nodes = {0: amd.Node(1, amd.NodeType.USER), 1: amd.Node(2, amd.NodeType.ITEM)}
    graph = amd.Graph(nodes, {1: []})
    graph.buildGraph()
    with pytest.raises(KeyError):
        amd.Recommender(graph).Recommendation(0, 0.5, 1)
    # capacity error reports the needed size
    lib = _lib.load()
    cnt = C.c_int64(0)
    st = lib.rwr_recommend(G._handle(), 0, C.c_float(0.5), 3, 0, None, None, C.byref(cnt))
    assert st == _lib.RWR_E_CAPACITY and cnt.value == 1


@pytest.mark.parametrize("T", [0, 1, 2, 3])
@pytest.mark.parametrize("top_n", [1, 50, 1024, 1025])
def test_topk_select_ties_and_zeros(amd, T, top_n):
    """Few iterations leave most items at score exactly 0: the k-th entry then sits inside a huge tie
    that only the id (descending) breaks (Recommender.cs:36-37) -- the radix
    select has to descend through all score digits into the id digits.  top_n = 1025 takes the
    full-sort path instead; both must equal the oracle."""
    g = gg.random_graph(31, n_users=300, n_items=4000, n_likes=5000, n_friend=100, n_mention=50)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    seeds = np.array([0, 1, 5, 17, 33, 100, 299], dtype=np.int32)       # K not a multiple of the tile width
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, T, top_n)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, T, top_n)
    assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all()


def test_topk_more_than_candidates(amd):
    g = gg.random_graph(32, n_users=50, n_items=40, n_likes=300)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    seeds = np.arange(0, 50, dtype=np.int32)
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, 10, 64)     # only <= 40 items exist
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 10, 64)
    assert (cnt == oc).all() and (cnt <= 40).all()
    assert (ids == oi).all() and (bits(sc) == bits(os_)).all()


def test_structural_ties_keep_id_order(amd):
    """Items with identical in-neighbour sets get bitwise-equal scores; their order is id descending."""
    n_users, n_items = 6, 40
    node_id = np.concatenate([np.arange(6), 1000 + np.random.default_rng(3).permutation(n_items)]).astype(np.int64)
    node_type = np.array([gg.NODE_USER] * n_users + [gg.NODE_ITEM] * n_items, dtype=np.uint8)
    lists = {i: [] for i in range(n_users + n_items)}
    for j in range(n_items):                     # every item liked by users 1 and 2 only -> all items symmetric
        for u in (1, 2):
            lists[u].append(n_users + j)
            lists[n_users + j].append(u)
    lists[0] = [n_users + 0]
    lists[n_users + 0].append(0)
    g = gg._from_lists(node_id, node_type, lists)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G).Recommendation(1, 0.15, 6)
    ids, sc = F.recommend(1, 0.15, 6)
    assert [r[0] for r in rec] == ids.tolist() and (bits([r[1] for r in rec]) == bits(sc)).all()
    bi, bs, bc = amd.Recommender(G).RecommendationBatch(np.array([1, 2, 0], dtype=np.int32), 0.15, 6, 10)
    oi, os_, oc = F.recommend_batch(np.array([1, 2, 0], dtype=np.int32), 0.15, 6, 10)
    assert (bi == oi).all() and (bits(bs) == bits(os_)).all() and (bc == oc).all()


@pytest.mark.parametrize("case", SMALL[:3], ids=lambda c: f"g{c['seed']}")
def test_model_threshold_and_default_run(amd, case):
    """Model.run(double) / Model.run() (Model.cs:52-66,110-115): ranks bitwise (the kernels are the exact ones),
    iteration count equal to the oracle's (the L1 distance is tree-summed on the GPU, sequential in the
    reference: only a distance within rounding of the threshold could differ)."""
    g = gg.random_graph(**case)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    d = po.widen_float(0.15)
    for seed in (0, case["n_users"] - 1):
        for thr in (1e-3, 1e-9):
            m = amd.Model(G, d, seed)
            m.run(thr)
            r, it = F.model_run(d, seed, 1, thr)
            assert m.iterations == it
            assert (bits(m.rank) == bits(r)).all()
    # run(): threshold (1/double.MaxValue)*n -- iterate until the ranks stop changing at all
    # (some graphs never reach an exact fixed point -- the ranks keep flipping in the last bit and the reference
    # itself would loop forever; only graphs on which the oracle converges are compared)
    r, it = F.model_run(d, 0, 2, 0.0, max_iter=20000)
    if it < 20000:
        m = amd.Model(G, d, 0)
        m.run()
        assert m.iterations == it and (bits(m.rank) == bits(r)).all()


@pytest.mark.parametrize("case", SMALL[:2] + MEDIUM[:1], ids=lambda c: f"g{c['seed']}")
def test_global_model_tolerance(amd, case):
    """Global (non-personalised) model, Model.cs:14-31: tolerance parity only -- the reference
    interleaves n restart addends into every row; the GPU adds the tree-summed mass / n once per row."""
    g = gg.random_graph(**case)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    n = len(g["node_id"])
    for T in (1, 4, 10):
        m = amd.Model(G, 0.15)
        m.run(T)
        r, _ = F.model_run(0.15, -1, 0, T, dense=(n < 500))
        assert np.abs(m.rank - r).max() <= 1e-12 * max(1.0, np.abs(r).max())
        assert abs(m.rank.sum() - n) < 1e-9 * n
    m = amd.Model(G, 0.15)
    m.run(1e-10)
    r, it = F.model_run(0.15, -1, 1, 1e-10)
    assert abs(m.iterations - it) <= 1 and np.abs(m.rank - r).max() <= 1e-9


@pytest.mark.parametrize("world", [1, 2, 3])
def test_row_partitioned_logical_ranks(amd, world):
    """BASELINE config 5 on ONE device: `world` logical ranks, each with the HIP slab backend (rwr_part_step),
    partials summed in rank order in place of the RCCL reduce-scatter (every rank gets ITS slab of the sum; the rest of
    its buffer is poisoned with NaN).  Tolerance parity (partial sums re-associate): identical top-k lists, scores within
    1e-9 of the oracle."""
    import torch
    from recommendersystems_amd import partitioned as pt
    g = gg.random_graph(51, n_users=500, n_items=1800, n_likes=15000, n_etc=10, n_friend=500, n_mention=300,
                        n_author=200)
    F = FlatGraph(**g)
    seeds = np.array([0, 7, 123, 250, 499], dtype=np.int32)
    bounds = pt.slab_bounds(g["rowptr"], world)
    bes = [pt.HipSlabBackend(pt.slab_graph(g, int(bounds[r]), int(bounds[r + 1])), int(bounds[r]), int(bounds[r + 1]))
           for r in range(world)]
    d = float(np.float32(0.15))
    state = [list(be.begin(seeds, d)[:2]) for be in bes]
    G = state[0][0].numel() // len(g["node_id"])
    T = 10
    for it in range(T):
        for be, (x, y) in zip(bes, state):
            be.step(x, y)                                        # rwr_part_step: asynchronous, on torch's current stream
        ysum = torch.stack([st[1] for st in state]).sum(0)       # what the collective delivers
        for r, st in enumerate(state):
            if it + 1 == T:
                st[1].copy_(ysum)                                # last step: all-reduce (ranking needs every row)
            else:                                                # reduce-scatter: a rank receives ITS slab only -- the rest
                st[1].fill_(float("nan"))                        # of the buffer is poisoned to prove it is never read
                lo, hi = int(bounds[r]) * G, int(bounds[r + 1]) * G
                st[1][lo:hi] = ysum[lo:hi]
            st[0], st[1] = st[1], st[0]
    state = [(x, y, None) for x, y in state]
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 10, 20)
    ids = np.zeros_like(oi); sc = np.zeros_like(os_); cnt = np.zeros_like(oc); owners = np.zeros(len(seeds), dtype=int)
    for be, (x, y, r) in zip(bes, state):
        i, s, c = be.rank(x, 20)
        own = c >= 0
        owners += own
        ids[own], sc[own], cnt[own] = i[own], s[own], c[own]
    assert (owners == 1).all()
    assert (cnt == oc).all() and (ids == oi).all()
    assert np.abs(sc - os_).max() <= 1e-9
    if world == 1:   # the driver class itself, world size 1 (no process group needed)
        pr = pt.PartitionedRecommender(g)
        i2, s2, c2 = pr.RecommendationBatch(seeds, 0.15, 10, 20)
        assert (i2 == oi).all() and (c2 == oc).all() and np.abs(s2 - os_).max() <= 1e-9
        many = (np.arange(150, dtype=np.int64) * 500 // 150).astype(np.int32)       # beyond one 64-seed tile
        i3, s3, c3 = pr.RecommendationBatch(many, 0.15, 10, 20)
        o3 = F.recommend_batch(many, 0.15, 10, 20)
        assert (i3 == o3[0]).all() and (c3 == o3[2]).all() and np.abs(s3 - o3[1]).max() <= 1e-9


def test_recommend_eval_bitwise(amd):
    """Section 8(f)-1: Hits / running precision of Experiment.cs:121-128 computed on the device."""
    from oracle.c_oracle import evaluate as c_eval
    g = gg.random_graph(61, n_users=400, n_items=3000, n_likes=12000, n_friend=200, n_mention=100)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rng = np.random.default_rng(9)
    item_ids = g["node_id"][g["node_type"] == gg.NODE_ITEM]
    for seed in (0, 57, 399):
        ids, _ = F.recommend(seed, 0.15, 8)
        test = set(rng.choice(item_ids, 40, replace=False).tolist()) | {-5}
        hits, sp, ln = amd.Recommender(G).RecommendationEval(seed, 0.15, 8, test)
        oh, osp = c_eval(ids, sorted(test))
        assert ln == len(ids) and hits == oh
        assert np.float64(sp).view(np.uint64) == np.float64(osp).view(np.uint64)
    assert amd.Recommender(G).RecommendationEval(0, 0.15, 3, set())[:2] == (0, 0.0)


def test_recommend_eval_batch_bitwise(amd):
    """rwr_recommend_eval_batch: K seeds with K test sets (CSR) -- entry k equals the single-seed call and the oracle's
    evaluation of the oracle's full list: empty sets, duplicates, ids that are no items, a dangling seed, K = 1."""
    from oracle.c_oracle import evaluate as c_eval
    g = gg.random_graph(62, n_users=300, n_items=2500, n_likes=9000, n_friend=150, n_mention=80)
    et = g["etype"].copy()
    et[g["rowptr"][7]:g["rowptr"][8]] = gg.EDGE_UNDEFINED          # user 7 becomes dangling
    g = dict(g, etype=et)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G)
    rng = np.random.default_rng(3)
    item_ids = g["node_id"][g["node_type"] == gg.NODE_ITEM]
    seeds = np.array([0, 7, 57, 299, 120, 0, 33], dtype=np.int32)
    tests = []
    for k in range(len(seeds)):
        t = rng.choice(item_ids, int(rng.integers(0, 60)), replace=False).tolist()
        if k == 2:
            t = t + t[:5] + [-9, 10 ** 15]                            # duplicates and ids that are no items
        if k == 4:
            t = []
        tests.append(t)
    hits, sp, ln = rec.RecommendationEvalBatch(seeds, 0.15, 8, tests)
    for k, sd in enumerate(seeds):
        ids, _ = F.recommend(int(sd), 0.15, 8)
        oh, osp = c_eval(ids, sorted(set(tests[k])))
        assert ln[k] == len(ids) and hits[k] == oh, k
        assert np.float64(sp[k]).view(np.uint64) == np.float64(osp).view(np.uint64), k
        one = rec.RecommendationEval(int(sd), 0.15, 8, set(tests[k]))
        assert one == (int(hits[k]), float(sp[k]), int(ln[k])), k
    h1, s1, l1 = rec.RecommendationEvalBatch(seeds[:1], 0.15, 8, tests[:1])
    assert (h1[0], s1[0], l1[0]) == (hits[0], sp[0], ln[0])
    h0, s0, l0 = rec.RecommendationEvalBatch(seeds[:0], 0.15, 8, [])
    assert len(h0) == 0
    with pytest.raises(Exception):
        rec.RecommendationEvalBatch(np.array([0, 99999], dtype=np.int32), 0.15, 8, [[], []])


def test_eval_graphs_batch_bitwise(amd):
    """rwr_eval_graphs: many graphs, one seed and one test set each -- entry k equals create + RecommendationEval + destroy
    on graph k and the oracle's evaluation of the oracle's list.  The batch mixes graphs of the one-launch paths (most),
    weighted and relabelled (UNDEFINED) links, a dangling seed, an empty test set, a graph without items, a graph beyond the
    one-launch call's item limit and one beyond the one-launch build's node limit (both take the single-graph path inside)."""
    from oracle.c_oracle import evaluate as c_eval
    rng = np.random.default_rng(11)
    graphs, seeds, tests = [], [], []
    for k in range(14):
        g = gg.random_graph(100 + k, n_users=int(rng.integers(20, 400)), n_items=int(rng.integers(30, 2500)),
                            n_likes=int(rng.integers(200, 9000)), n_friend=int(rng.integers(0, 300)),
                            n_mention=int(rng.integers(0, 200)), n_author=int(rng.integers(0, 100)),
                            p_undefined=0.15 if k % 3 == 0 else 0.0, uniform=(k % 4 == 1))
        graphs.append(g)
        seeds.append(int(rng.integers(0, 20)))
    # a dangling seed: every out-link of user 3 of graph 2 relabelled UNDEFINED
    et = graphs[2]["etype"].copy()
    et[graphs[2]["rowptr"][3]:graphs[2]["rowptr"][4]] = gg.EDGE_UNDEFINED
    graphs[2] = dict(graphs[2], etype=et)
    seeds[2] = 3
    # beyond the one-launch call (items > 4096) and beyond the one-launch build (nodes > 8192)
    graphs.append(gg.random_graph(300, n_users=200, n_items=5000, n_likes=12000))
    seeds.append(5)
    graphs.append(gg.random_graph(301, n_users=1500, n_items=9000, n_likes=30000))
    seeds.append(9)
    # no ITEM node at all
    gi = gg.random_graph(302, n_users=50, n_items=40, n_likes=300)
    gi = dict(gi, node_type=np.where(gi["node_type"] == gg.NODE_ITEM, gg.NODE_USER, gi["node_type"]).astype(np.uint8))
    graphs.append(gi)
    seeds.append(1)
    for k, g in enumerate(graphs):
        item_ids = g["node_id"][g["node_type"] == gg.NODE_ITEM]
        t = rng.choice(item_ids, min(len(item_ids), int(rng.integers(0, 50))), replace=False).tolist() if len(item_ids) else []
        if k == 1:
            t = []
        if k == 4:
            t = t + t[:3] + [-5, 10 ** 14]
        tests.append(t)
    Gs = [amd.Graph.from_flat(**{k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}) for g in graphs]
    hits, sp, ln = amd.EvaluateGraphs(Gs, seeds, 0.15, 7, tests)
    for k, g in enumerate(graphs):
        F = FlatGraph(**g)
        ids, _ = F.recommend(int(seeds[k]), 0.15, 7)
        oh, osp = c_eval(ids, sorted(set(tests[k])))
        assert ln[k] == len(ids) and hits[k] == oh, k
        assert np.float64(sp[k]).view(np.uint64) == np.float64(osp).view(np.uint64), k
        G1 = dev_graph(amd, g)
        one = amd.Recommender(G1).RecommendationEval(int(seeds[k]), 0.15, 7, set(tests[k]))
        assert one == (int(hits[k]), float(sp[k]), int(ln[k])), k
        G1.close()
    # twice the same batch (the thread's pinned arena and the handle pool are reused), and an empty one
    h2, s2, l2 = amd.EvaluateGraphs(Gs, seeds, 0.15, 7, tests)
    assert (h2 == hits).all() and (s2.view(np.uint64) == sp.view(np.uint64)).all() and (l2 == ln).all()
    h0, s0, l0 = amd.EvaluateGraphs([], [], 0.15, 7, [])
    assert len(h0) == 0
    # errors name the graph: a seed out of range, a link target out of range
    with pytest.raises(Exception, match="graph 1"):
        amd.EvaluateGraphs(Gs[:2], [0, 10 ** 6], 0.15, 7, tests[:2])
    bad = dict(graphs[0], dst=graphs[0]["dst"].copy())
    bad["dst"][0] = 10 ** 6
    Gb = amd.Graph.from_flat(**{k: bad[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")})
    with pytest.raises(Exception, match="graph 1"):
        amd.EvaluateGraphs([Gs[0], Gb], seeds[:2], 0.15, 7, tests[:2])
    # (the number is the caller's: graph 15 of this batch is beyond the one-launch build and is not among the graphs built
    #  together, so the broken graph is number 16 for the caller but the 16th -- position 15 -- of that group)
    with pytest.raises(Exception, match="graph 16"):
        amd.EvaluateGraphs(Gs[:16] + [Gb], seeds[:16] + [seeds[0]], 0.15, 7, tests[:16] + [tests[0]])
    # ... and the library is usable afterwards
    h3, _, _ = amd.EvaluateGraphs(Gs[:3], seeds[:3], 0.15, 7, tests[:3])
    assert (h3 == hits[:3]).all()


def test_eval_graphs_from_concurrent_host_threads(amd):
    """rwr_eval_graphs from several host threads at once (the harness's threads each own an ego network, Program.cs:11): every
    call's results equal those of the same batch evaluated alone.  The threads share the device-memory cache, the pool of
    pinned sets and the pool of streams."""
    import threading
    rng = np.random.default_rng(23)
    batches = []
    for t in range(5):
        graphs, seeds, tests = [], [], []
        for k in range(7 + t):
            g = gg.random_graph(700 + 20 * t + k, n_users=int(rng.integers(30, 300)), n_items=int(rng.integers(50, 1800)),
                                n_likes=int(rng.integers(300, 6000)), n_friend=int(rng.integers(0, 200)), n_mention=int(rng.integers(0, 100)))
            graphs.append(amd.Graph.from_flat(**g))
            seeds.append(int(rng.integers(0, 30)))
            ids = g["node_id"][g["node_type"] == gg.NODE_ITEM]
            tests.append(rng.choice(ids, min(len(ids), 25), replace=False).tolist())
        batches.append((graphs, seeds, tests))
    alone = [amd.EvaluateGraphs(gs, sd, 0.15, 9, ts) for gs, sd, ts in batches]
    errors, got = [], [None] * len(batches)

    def worker(t):
        try:
            gs, sd, ts = batches[t]
            for _ in range(6):
                got[t] = amd.EvaluateGraphs(gs, sd, 0.15, 9, ts)
        except Exception as e:                                   # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(batches))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for (h, sp, ln), (h2, sp2, ln2) in zip(alone, got):
        assert (h == h2).all() and (sp.view(np.uint64) == sp2.view(np.uint64)).all() and (ln == ln2).all()


@pytest.mark.parametrize("n_users,n_items", [(3000, 5200), (900, 9000)])
def test_ego_network_beyond_the_one_launch_call(amd, n_users, n_items):
    """An ego network too large for small.hip (more than 6144 nodes or 4096 items) through the general single-seed path: the
    ego (node 0, the seed) has SEVERAL in-links from most of its network (FRIENDSHIP + FOLLOW + weighted MENTION from the
    same node, as the loader builds them: DataLoader.cs:256-436), so every block of the seed-row chain is full of links into
    the seed -- the exact-start block, the crossing rows and the redone blocks all walk them.  Bitwise against the oracle,
    list and scores, over several iteration counts; only a handful of chain blocks may need the exact redo."""
    rng = np.random.default_rng(n_users)
    n = n_users + n_items
    lists = [[] for _ in range(n)]          # (target, type, weight)
    for u in range(1, n_users):
        if rng.random() < 0.85:
            lists[u].append((0, gg.EDGE_FRIENDSHIP, 1.0))
            lists[0].append((u, gg.EDGE_FRIENDSHIP, 1.0))
        if rng.random() < 0.7:
            lists[u].append((0, gg.EDGE_FOLLOW, 1.0))
        if rng.random() < 0.5:
            lists[u].append((0, gg.EDGE_MENTION, float(rng.integers(1, 9))))
        for v in rng.integers(1, n_users, 3):
            if int(v) != u:
                lists[u].append((int(v), gg.EDGE_FOLLOW, 1.0))
    for u in range(n_users):
        for it in rng.integers(0, n_items, 80 if u == 0 else int(rng.integers(2, 25))):
            lists[u].append((n_users + int(it), gg.EDGE_LIKE, 1.0))
            lists[n_users + int(it)].append((u, gg.EDGE_LIKE, 1.0))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    for i in range(n):
        rowptr[i + 1] = rowptr[i] + len(lists[i])
    flat = [x for ls in lists for x in ls]
    g = dict(node_id=rng.permutation(np.arange(10, 10 + 3 * n, 3, dtype=np.int64)),
             node_type=np.array([gg.NODE_USER] * n_users + [gg.NODE_ITEM] * n_items, dtype=np.uint8), rowptr=rowptr,
             dst=np.array([x[0] for x in flat], dtype=np.int32), etype=np.array([x[1] for x in flat], dtype=np.uint8),
             w=np.array([x[2] for x in flat], dtype=np.float64))
    F = FlatGraph(**g)
    G = dev_graph(amd, g, profile=True)
    rec = amd.Recommender(G)
    for T in (1, 2, 5, 10):
        got = rec.Recommendation(0, 0.15, T)
        ids, sc = F.recommend(0, 0.15, T)
        assert [r[0] for r in got] == ids.tolist(), T
        assert (bits([r[1] for r in got]) == bits(sc)).all(), T
    st = G.stats()
    assert st["chain_redo_blocks"] <= 12, st["chain_redo_blocks"]          # (18 steps, 8-10 blocks each)
    hits, sp, ln = rec.RecommendationEval(0, 0.15, 10, set(int(x) for x in ids[:30:3]))
    assert (hits, ln) == (10, len(ids))
    G.close()


def test_cpp_host_mirror_runs_the_kats():
    """include/recommenders/rwr_based.hpp + tests/cpp/experiment_like.cpp: the caller pattern of
    Experiment.cs:104-128 in C++ against librwr (built by __graft_entry__.build())."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "experiment_like")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "experiment_like: ok" in out.stdout


def test_concurrent_host_threads_independent_handles(amd):
    """The reference runs up to 10 worker threads, each with its own Graph/Recommender (Program.cs:11,61-66;
    Experiment.cs:71,104,108): handles must be usable concurrently from distinct host threads."""
    import threading
    cases = [dict(seed=80 + t, n_users=200 + 30 * t, n_items=900, n_likes=5000, n_friend=100, n_mention=60) for t in range(6)]
    graphs = [gg.random_graph(**c) for c in cases]
    expect = []
    for g in graphs:
        F = FlatGraph(**g)
        expect.append([F.recommend(s, 0.15, 10) for s in (0, 11, 57)])
    results = [None] * len(graphs)
    errors = []

    def worker(t):
        try:
            G = dev_graph(amd, graphs[t])
            rec = amd.Recommender(G)
            out = []
            for _ in range(3):                                   # repeat to overlap with the other threads
                out = [rec.Recommendation(s, 0.15, 10) for s in (0, 11, 57)]
            results[t] = out
            G.close()
        except Exception as e:                                   # pragma: no cover
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(graphs))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for t in range(len(graphs)):
        for got, (ids, sc) in zip(results[t], expect[t]):
            assert [r[0] for r in got] == ids.tolist()
            assert (bits([r[1] for r in got]) == bits(sc)).all()


def test_degenerate_graphs(amd):
    # a single user, no links, no items
    g = gg._from_lists(np.array([5], dtype=np.int64), np.array([gg.NODE_USER], dtype=np.uint8), {0: []})
    G = dev_graph(amd, g)
    assert amd.Recommender(G).Recommendation(0, 0.15, 4) == []
    m = amd.Model(G, 0.15, 0)
    m.run(3)
    assert m.rank.tolist() == [1.0]                              # dangling seed: all mass returns to it
    # items only reachable by nobody: every candidate has score 0, ordered by id descending
    node_id = np.array([1, 40, 20, 30], dtype=np.int64)
    node_type = np.array([gg.NODE_USER, gg.NODE_ITEM, gg.NODE_ITEM, gg.NODE_ITEM], dtype=np.uint8)
    g = gg._from_lists(node_id, node_type, {0: [], 1: [], 2: [], 3: []})
    G = dev_graph(amd, g)
    assert amd.Recommender(G).Recommendation(0, 0.15, 5) == [(40, 0.0), (30, 0.0), (20, 0.0)]
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(np.array([0, 0], dtype=np.int32), 0.15, 5, 2)
    assert ids.tolist() == [[40, 30], [40, 30]] and cnt.tolist() == [2, 2]
    # a batch whose seeds are all dangling on a graph WITHOUT items: empty lists, counts 0 (not stale memory)
    g = gg._from_lists(np.array([0], dtype=np.int64), np.array([gg.NODE_UNDEFINED], dtype=np.uint8), {0: []})
    G = dev_graph(amd, g)
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(np.array([0, 0, 0], dtype=np.int32), 0.15, 1, 5)
    assert cnt.tolist() == [0, 0, 0]
    # an ITEM whose id is INT64_MIN (its id-descending sort key is all ones -- no sentinel may share it), among non-items
    lo, hi = -2 ** 63, 2 ** 63 - 1
    node_id = np.array([0, 1, -1, hi, lo, 3], dtype=np.int64)
    node_type = np.array([gg.NODE_USER, gg.NODE_UNDEFINED, gg.NODE_ETC, gg.NODE_ITEM, gg.NODE_ITEM, gg.NODE_USER], dtype=np.uint8)
    g = gg._from_lists(node_id, node_type, {i: [] for i in range(6)})
    G = dev_graph(amd, g)
    assert amd.Recommender(G).Recommendation(0, 0.15, 3) == [(hi, 0.0), (lo, 0.0)]
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(np.array([0, 5, 0], dtype=np.int32), 0.15, 0, 5)
    assert cnt.tolist() == [2, 2, 2] and ids[:, :2].tolist() == [[hi, lo]] * 3
    F = FlatGraph(**g)
    oi, os_, oc = F.recommend_batch(np.array([0, 5, 0], dtype=np.int32), 0.15, 0, 5)
    assert (oi == ids).all() and (oc == cnt).all()


@pytest.mark.parametrize("tile_seeds", [1, 8, 32])
def test_seed_row_binade_scan_bitwise(amd, tile_seeds):
    """The parallel form of the seed-row chain (chain_scan.hip) on a graph large enough for many blocks per seed, with
    weights spanning 60 binades (half-way addends, tiny addends that vanish, addends larger than the running sum), hub
    seeds with thousands of in-links and dangling rows; bitwise against the C restatement, and against the sequential
    fold kernel; the carry must have redone some blocks (binade crossings) but not all of them."""
    rng = np.random.default_rng(99)
    U, I = 30000, 50000
    n = U + I
    lists = {i: [] for i in range(n)}
    wl = {i: [] for i in range(n)}
    def link(a, b, w):
        lists[a].append(b); wl[a].append(w)
    us = (rng.random(300000) ** 2 * U).astype(np.int64)
    vs = (rng.random(300000) ** 3 * I).astype(np.int64)
    seen = set()
    for u, v in zip(us.tolist(), vs.tolist()):
        if (u, v) in seen:
            continue
        seen.add((u, v))
        w = float(2.0 ** rng.integers(-30, 30)) if rng.random() < 0.5 else float(rng.random() * 2.0 ** rng.integers(-30, 30))
        link(u, U + v, w)
        link(U + v, u, 1.0)
    for u in range(0, 3000):                                     # hub seed 5: thousands of in-links
        link(U + (u * 7) % I, 5, 0.5)
    node_id = rng.permutation(n).astype(np.int64) * 3 - 1000
    node_type = np.array([gg.NODE_USER] * U + [gg.NODE_ITEM] * I, dtype=np.uint8)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    for i in range(n):
        rowptr[i + 1] = rowptr[i] + len(lists[i])
    dst = np.fromiter((t for i in range(n) for t in lists[i]), dtype=np.int32, count=int(rowptr[n]))
    w = np.fromiter((x for i in range(n) for x in wl[i]), dtype=np.float64, count=int(rowptr[n]))
    g = dict(node_id=node_id, node_type=node_type, rowptr=rowptr, dst=dst, etype=np.full(len(dst), gg.EDGE_LIKE, dtype=np.uint8), w=w)
    F = FlatGraph(**g)
    K = {1: 1, 8: 11, 32: 70}[tile_seeds]
    seeds = np.unique(np.concatenate([[5, 0, U - 1], rng.integers(0, U, K)]))[:max(K, 1)].astype(np.int32)
    if tile_seeds == 1:
        seeds = np.array([5], dtype=np.int32)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 7, 30)
    outs = {}
    for kern in ("scan", "fold"):
        G = dev_graph(amd, g, tile_seeds=tile_seeds, seed_row_kernel=kern, profile=True)
        ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, 7, 30)
        assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all(), kern
        for sd in seeds[:2]:
            m = amd.Model(G, po.widen_float(0.15), int(sd))
            m.run(7)
            r, _ = F.model_run(po.widen_float(0.15), int(sd), 0, 7)
            assert (bits(m.rank) == bits(r)).all(), (kern, int(sd))
        outs[kern] = G.stats()["chain_redo_blocks"]
        G.close()
    nblocks = -(-n * tile_seeds // 4096)
    assert outs["fold"] == 0
    # (a single seed's crossings are predicted in the parallel pass -- chain_scan.hip: k_cs_block1 -- and none may need a redo)
    assert (0 if tile_seeds == 1 else 1) <= outs["scan"] < 0.5 * nblocks * len(seeds) * 7, outs


def test_dangling_seeds_in_a_batch(amd):
    """Seeds without explicit out-links are answered without iterating (rank = n at the seed, 0 elsewhere, for any T):
    users with no links, users whose links are all UNDEFINED, and a dangling ITEM seed (which leads its own list)."""
    rng = np.random.default_rng(4)
    g = gg.random_graph(91, n_users=60, n_items=300, n_likes=900, n_friend=40)
    n = len(g["node_id"])
    # make users 50..59 dangling: relabel all their links UNDEFINED; add two isolated nodes (a user and an item)
    et = g["etype"].copy()
    for u in range(50, 60):
        et[g["rowptr"][u]:g["rowptr"][u + 1]] = gg.EDGE_UNDEFINED
    node_id = np.concatenate([g["node_id"], [10 ** 12, -77]]).astype(np.int64)
    node_type = np.concatenate([g["node_type"], [gg.NODE_USER, gg.NODE_ITEM]]).astype(np.uint8)
    rowptr = np.concatenate([g["rowptr"], [g["rowptr"][-1], g["rowptr"][-1]]]).astype(np.int64)
    g2 = dict(node_id=node_id, node_type=node_type, rowptr=rowptr, dst=g["dst"], etype=et, w=g["w"])
    F = FlatGraph(**g2)
    G = dev_graph(amd, g2)
    seeds = np.array([0, 55, 3, n, n + 1, 59, 7, 50], dtype=np.int32)        # mix of live and dangling seeds
    for T in (0, 4):
        for top_n in (1, 20, 400):
            ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, T, top_n)
            oi, os_, oc = F.recommend_batch(seeds, 0.15, T, top_n)
            assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all(), (T, top_n)
    only = np.array([55, n + 1, 59], dtype=np.int32)                           # a batch of dangling seeds only
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(only, 0.15, 6, 10)
    oi, os_, oc = F.recommend_batch(only, 0.15, 6, 10)
    assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all()


def test_hub_nodes_long_rows(amd):
    """One item liked by every user and one user who likes every item: in-lists of 20 000 / 3 000 entries (long
    sequential rows, the hub-row reduction of the K = 1 SpMV, a seed with thousands of in-links for the chain)."""
    U, I = 20000, 3000
    rng = np.random.default_rng(12)
    lists = {i: [] for i in range(U + I)}
    def like(u, v):
        lists[u].append(U + v); lists[U + v].append(u)
    for u in range(U):
        like(u, 0)                                   # item 0: liked by everyone
    for v in range(1, I):
        like(7, v)                                   # user 7: likes everything
    for _ in range(40000):
        u, v = int(rng.integers(0, U)), int(rng.integers(1, I))
        if U + v not in lists[u]:
            like(u, v)
    node_id = np.arange(U + I, dtype=np.int64) * 5 + 3
    node_type = np.array([gg.NODE_USER] * U + [gg.NODE_ITEM] * I, dtype=np.uint8)
    g = gg._from_lists(node_id, node_type, lists)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G)
    seeds = np.array([7, 0, 19999, 1234] + list(range(100, 128)), dtype=np.int32)
    ids, sc, cnt = rec.RecommendationBatch(seeds, 0.15, 10, 50)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 10, 50)
    assert (cnt == oc).all() and (ids == oi).all()
    assert (bits(sc) == bits(os_)).all()
    assert rec.Recommendation(7, 0.15, 10, 50) == []   # user 7 likes every item: nothing left to recommend
    single = rec.Recommendation(1234, 0.15, 10, 50)    # K = 1 path
    si, ss = F.recommend(1234, 0.15, 10, 50)
    assert [r[0] for r in single] == si.tolist() and len(si) == 50
    assert (bits([r[1] for r in single]) == bits(ss)).all()


def _same_results(amd, G, F, seeds, T=6, top_n=15):
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, T, top_n)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, T, top_n)
    assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all()
    wn, dg = G.normalized()
    assert (bits(wn) == bits(F.w_norm)).all() and (dg == F.dangling).all()
    full = amd.Recommender(G).Recommendation(int(seeds[0]), 0.15, T)
    fi, fs = F.recommend(int(seeds[0]), 0.15, T)
    assert [r[0] for r in full] == fi.tolist() and (bits([r[1] for r in full]) == bits(fs)).all()


def test_incremental_rebuild_matches_fresh_build(amd):
    """rwr_graph_update_links: relabel FRIENDSHIP -> UNDEFINED as Experiment.cs:84-101 does, change raw
    weights, make a node dangling, revert -- after every patch the device state must equal a graph built from scratch
    from the patched lists (checked against the oracle on those lists), without re-sending the unchanged arrays."""
    g = gg.random_graph(77, n_users=400, n_items=1200, n_likes=7000, n_etc=10, n_friend=900, n_mention=300, n_author=200)
    seeds = np.array([0, 3, 50, 399, 120], dtype=np.int32)
    G = dev_graph(amd, g, profile=True)
    _same_results(amd, G, FlatGraph(**g), seeds)
    handle = G._h.value
    # 1. the harness's relabel: every FRIENDSHIP link becomes UNDEFINED
    et = g["etype"].copy()
    fr = np.flatnonzero(et == gg.EDGE_FRIENDSHIP)
    assert len(fr) > 100
    et[fr] = gg.EDGE_UNDEFINED
    G.updateLinks(fr, etype=et[fr])
    g1 = dict(g, etype=et)
    _same_results(amd, G, FlatGraph(**g1), seeds)
    assert G.stats()["nnz"] == int((et != 0).sum()) and G._h.value == handle
    # 2. weights only (MENTION-like), types untouched
    w = g["w"].copy()
    mi = np.flatnonzero(et == gg.EDGE_MENTION)[::2]
    w[mi] = w[mi] * 3.0 + 0.125
    G.updateLinks(mi, w=w[mi])
    g2 = dict(g1, w=w)
    _same_results(amd, G, FlatGraph(**g2), seeds)
    # 3. types and weights together; node 3 loses every link (becomes dangling), a LIKE of seed 0 becomes PURCHASE
    et3 = et.copy(); w3 = w.copy()
    r3 = np.arange(g["rowptr"][3], g["rowptr"][4])
    et3[r3] = gg.EDGE_UNDEFINED
    l0 = np.flatnonzero(et3[g["rowptr"][0]:g["rowptr"][1]] == gg.EDGE_LIKE)[:1] + g["rowptr"][0]
    et3[l0] = gg.EDGE_PURCHASE; w3[l0] = 2.5
    idx = np.concatenate([r3, l0])
    G.updateLinks(idx, etype=et3[idx], w=w3[idx])
    g3 = dict(g2, etype=et3, w=w3)
    _same_results(amd, G, FlatGraph(**g3), seeds)
    # 4. back to the original lists: same results as the very first build
    diff = np.flatnonzero((et3 != g["etype"]) | (w3 != g["w"]))
    G.updateLinks(diff, etype=g["etype"][diff], w=g["w"][diff])
    _same_results(amd, G, FlatGraph(**g), seeds)
    # errors leave the graph usable
    with pytest.raises(amd.RwrError) as ei:
        G.updateLinks(np.array([len(et)], dtype=np.int64), etype=np.array([1], dtype=np.uint8))
    assert ei.value.status == 2                                   # RWR_E_RANGE
    G.updateLinks(np.zeros(0, dtype=np.int64))                     # count 0: plain rebuild
    _same_results(amd, G, FlatGraph(**g), seeds)
    G.close()


def test_rebuild_of_mutated_dictionaries_sends_only_the_diff(amd):
    """The reference-shaped classes: the host mutates ForwardLink.type in place (Experiment.cs:90-97) and calls
    buildGraph() again; the mirror sends only the changed links (same native handle) and the results equal the oracle's
    on the mutated containers."""
    g = gg.random_graph(31, n_users=60, n_items=200, n_likes=900, n_friend=150, n_mention=40)
    nodes = {i: amd.Node(int(g["node_id"][i]), amd.NodeType(int(g["node_type"][i]))) for i in range(len(g["node_id"]))}
    edges = {i: [amd.ForwardLink(int(g["dst"][e]), amd.EdgeType(int(g["etype"][e])), float(g["w"][e]))
                 for e in range(g["rowptr"][i], g["rowptr"][i + 1])] for i in range(len(nodes))}
    graph = amd.Graph(nodes, edges)
    graph.buildGraph()
    handle = graph._h.value
    before = amd.Recommender(graph).Recommendation(0, 0.15, 8)
    for i in edges:                                               # Experiment.cs:84-101
        for l in edges[i]:
            if l.type == amd.EdgeType.FRIENDSHIP:
                l.type = amd.EdgeType.UNDEFINED
    graph.buildGraph()
    assert graph._h.value == handle
    after = amd.Recommender(graph).Recommendation(0, 0.15, 8)
    pn, pe = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"],
                          np.where(g["etype"] == gg.EDGE_FRIENDSHIP, 0, g["etype"]).astype(np.uint8), g["w"])
    PG = po.Graph(pn, pe); PG.buildGraph()
    ref = po.Recommender(PG).Recommendation(0, 0.15, 8)
    assert [r[0] for r in after] == [r[0] for r in ref] and (bits([r[1] for r in after]) == bits([r[1] for r in ref])).all()
    assert after != before
    # a structural change (one more link) falls back to a full re-create
    edges[1].append(amd.ForwardLink(5, amd.EdgeType.FOLLOW, 1.0))
    graph.buildGraph()
    assert len(graph.graph[1]) == sum(1 for l in edges[1] if l.type != amd.EdgeType.UNDEFINED)


def test_negative_weights_take_the_general_path(amd):
    """Raw weights < 0 are outside the loader's domain (DataLoader.cs:293-294,431-432) but the containers allow them: ranks
    can then go negative, so the zero-skipping frontier kernels and the binade scan (both rely on ranks >= 0) step aside;
    Model.run stays bitwise equal to the literal restatement, Recommendation refuses loudly."""
    g = gg.random_graph(5, n_users=50, n_items=120, n_likes=700, n_friend=60, n_mention=50)
    w = g["w"].copy()
    rng = np.random.default_rng(3)
    pick = rng.choice(len(w), 40, replace=False)
    w[pick] = -0.25 * w[pick]                               # row sums stay non-zero (weights are >= 0.3 in magnitude mix)
    g2 = dict(g, w=w)
    nodes, edges = po.from_flat(g2["node_id"], g2["node_type"], g2["rowptr"], g2["dst"], g2["etype"], g2["w"])
    PG = po.Graph(nodes, edges)
    PG.buildGraph()
    for tile_seeds in (0, 16):
        G = dev_graph(amd, g2, tile_seeds=tile_seeds)
        for seed in (0, 7):
            m = po.Model(PG, po.widen_float(0.15), seed, dense_restart=True)
            m.run(6)
            dm = amd.Model(G, po.widen_float(0.15), seed)
            dm.run(6)
            assert (bits(dm.rank) == bits(m.rank)).all()
        with pytest.raises(amd.RwrError) as ei:
            amd.Recommender(G).Recommendation(0, 0.15, 5)
        assert ei.value.status == 7 and "non-negative" in str(ei.value)
        G.close()


@pytest.mark.parametrize("case", SMALL[:3] + MEDIUM[:1], ids=lambda c: f"g{c['seed']}")
def test_model_stepwise_public_api(amd, case):
    """Model.deliverRanks / updateRanks / checkConvergence as separate public steps (Model.cs:76,103,110), driven by the
    host exactly as the reference allows: every intermediate nextRank bitwise, checkConvergence identical; run() on an
    already advanced model CONTINUES from its rank (Model.cs:68-73); a rank vector edited by the host (halved, with a
    negative entry) still propagates bitwise (general kernels)."""
    g = gg.random_graph(**case)
    nodes, edges = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"], g["etype"], g["w"])
    PG = po.Graph(nodes, edges)
    PG.buildGraph()
    G = dev_graph(amd, g)
    d = po.widen_float(0.15)
    seed = case["n_users"] // 2
    ref, dev = po.Model(PG, d, seed, dense_restart=False), amd.Model(G, d, seed)
    for step in range(4):
        ref.deliverRanks(); dev.deliverRanks()
        assert (bits(dev.nextRank) == bits(ref.nextRank)).all(), step
        for thr in (1e-300, 0.5, 10.0, 1e9):
            assert dev.checkConvergence(thr) == ref.checkConvergence(thr)
        with pytest.raises(RuntimeError):
            dev.deliverRanks()                                   # nextRank not yet folded into rank
        ref.updateRanks(); dev.updateRanks()
        assert (bits(dev.rank) == bits(ref.rank)).all() and not dev.nextRank.any()
    # run(int) continues from the current state
    a, b = po.Model(PG, d, seed, dense_restart=False), amd.Model(G, d, seed)
    a.run(3); b.run(3)
    a.run(2); b.run(2)
    fresh = amd.Model(G, d, seed); fresh.run(5)
    assert (bits(b.rank) == bits(a.rank)).all() and (bits(b.rank) == bits(fresh.rank)).all()
    # threshold run on an advanced model: same number of further iterations, same ranks
    it_ref = a.run(1e-3); b.run(1e-3)
    assert b.iterations == it_ref and (bits(b.rank) == bits(a.rank)).all()
    # host-edited rank vector (public field): halved, one entry negative
    a, b = po.Model(PG, d, seed, dense_restart=False), amd.Model(G, d, seed)
    a.run(2); b.run(2)
    for m in (a, b):
        m.rank = [x * 0.5 for x in m.rank] if isinstance(m.rank, list) else m.rank * 0.5
        m.rank[1] = -0.75
    a.deliverRanks(); b.deliverRanks()
    assert (bits(b.nextRank) == bits(np.array(a.nextRank))).all()
    # global model: tolerance parity per step
    ga, gb = po.Model(PG, d), amd.Model(G, d)
    for _ in range(3):
        ga.deliverRanks(); gb.deliverRanks()
        assert np.allclose(gb.nextRank, np.array(ga.nextRank), rtol=1e-12, atol=1e-12)
        ga.updateRanks(); gb.updateRanks()
    G.close()


@pytest.mark.parametrize("d", [0.0, 1.0, 0.999, 1e-7])
def test_extreme_restart_probabilities(amd, d):
    """d = 0 (no restart: the chain only carries dangling mass), d = 1 (no walk), and values next to them: single seed
    (binade scan) and batch (both seed-row kernels) stay bitwise equal to the restatement."""
    g = gg.random_graph(8, n_users=300, n_items=900, n_likes=5000, n_etc=10, n_friend=200, n_mention=100)
    F = FlatGraph(**g)
    seeds = np.array([0, 5, 150, 299], dtype=np.int32)
    oi, os_, oc = F.recommend_batch(seeds, d, 7, 20)
    for kern in ("scan", "fold"):
        G = dev_graph(amd, g, seed_row_kernel=kern)
        ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, d, 7, 20)
        assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all(), kern
        m = amd.Model(G, po.widen_float(d), 5)
        m.run(7)
        r, _ = F.model_run(po.widen_float(d), 5, 0, 7)
        assert (bits(m.rank) == bits(r)).all(), kern
        G.close()


def test_fewer_links_than_nodes(amd):
    """Mostly isolated nodes (links << nodes): the build's sort buffers serve the link sort AND the node-order sorts, so
    they must be sized for the larger of the two (they once were sized by the links alone: rows went missing from the
    processing order).  Dangling seed through the single-seed path (no shortcut), live seeds, batch."""
    rng = np.random.default_rng(5)
    U, I = 6000, 9000
    lists = {i: [] for i in range(U + I)}
    for _ in range(700):
        u, v = int(rng.integers(0, U)), int(rng.integers(0, I))
        if U + v not in lists[u]:
            lists[u].append(U + v); lists[U + v].append(u)
    node_id = rng.permutation(U + I).astype(np.int64)
    node_type = np.array([gg.NODE_USER] * U + [gg.NODE_ITEM] * I, dtype=np.uint8)
    g = gg._from_lists(node_id, node_type, lists)
    assert len(g["dst"]) < U + I
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    live = [u for u in range(U) if lists[u]][:3]
    dead = [u for u in range(U) if not lists[u]][:2]
    for sd in live + dead:
        for T in (1, 2, 3, 6):
            m = amd.Model(G, po.widen_float(0.15), sd)
            m.run(T)
            r, _ = F.model_run(po.widen_float(0.15), sd, 0, T)
            assert (bits(m.rank) == bits(r)).all(), (sd, T)
        got = amd.Recommender(G).Recommendation(sd, 0.15, 5, 30)
        oi, os_ = F.recommend(sd, 0.15, 5, 30)
        assert [x[0] for x in got] == oi.tolist() and (bits([x[1] for x in got]) == bits(os_)).all()
    seeds = np.array(live + dead, dtype=np.int32)
    ids, sc, cnt = amd.Recommender(G).RecommendationBatch(seeds, 0.15, 6, 30)
    oi, os_, oc = F.recommend_batch(seeds, 0.15, 6, 30)
    assert (cnt == oc).all() and (ids == oi).all() and (bits(sc) == bits(os_)).all()


@pytest.mark.parametrize("uniform", [True, False], ids=["value-free", "weighted"])
def test_hub_row_exact_reduction_adversarial(amd, uniform):
    """A row of thousands of in-links is summed by the exact PARALLEL reduction of pf.h (spmv.hip: k_spmv_exact_hub) and
    must equal the reference's strictly sequential sum (Model.cs:85-88) bit for bit -- also when addends are exact halves
    of the running sum's ulp (ties to even, parity-dependent), when the sum crosses many binades, and across zeros and
    subnormals.  5 000 users each LIKE item 0 (one out-link each: weight 1); d = 0.5f, so the addend of user u is exactly
    rank[u] / 2, and the host puts adversarial values into the public rank vector before deliverRanks()."""
    U = 5000
    n = U + 2
    node_id = np.arange(100, 100 + n, dtype=np.int64)
    node_type = np.array([gg.NODE_USER] * U + [gg.NODE_ITEM] * 2, dtype=np.uint8)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    dst, w = [], []
    for u in range(U):
        dst.append(U)                       # u -> item 0
        w.append(1.0)
        if not uniform and u == 7:          # one row with two different weights: the graph takes the weighted kernels
            dst.append(U + 1)
            w.append(3.0)
        rowptr[u + 1] = len(dst)
    rowptr[U + 1:] = len(dst)
    g = dict(node_id=node_id, node_type=node_type, rowptr=rowptr, dst=np.array(dst, dtype=np.int32),
             etype=np.ones(len(dst), dtype=np.uint8), w=np.array(w, dtype=np.float64))
    G = dev_graph(amd, g)
    assert G.stats()["uniform_path"] == (1 if uniform else 0)
    F = FlatGraph(**g)
    rng = np.random.default_rng(2024)
    for trial in range(6):
        kind = trial % 3
        if kind == 0:      # a large head, then thousands of exact half-ulps and near-halves of it
            x = rng.choice([2.0 ** -51, 3 * 2.0 ** -52, 2.0 ** -52, 5 * 2.0 ** -53, 0.0], size=U)
            x[0] = 2.0
        elif kind == 1:    # binade crossings all the way: magnitudes spread over 70 binades
            x = np.ldexp(1.0 + rng.integers(0, 4, U) / 4.0, rng.integers(-60, 10, U).astype(np.int32))
        else:              # subnormal and tiny addends first, zeros in between, normal ones later
            x = np.concatenate([np.full(U // 2, 5e-324) * rng.integers(0, 9, U // 2), rng.random(U - U // 2) * 2.0 ** -1000])
            x[rng.integers(0, U, 50)] = 1.0
        m = amd.Model(G, 0.5, 3)
        m.rank = np.concatenate([x, [0.0, 0.0]])
        m.deliverRanks()
        # the reference's sum, one add at a time in source order (Python floats are binary64)
        wcol = g["w"][g["rowptr"][:U]] / np.array([g["w"][g["rowptr"][u]:g["rowptr"][u + 1]].sum() for u in range(U)])
        acc = 0.0
        for u in range(U):
            acc += (0.5 * float(x[u])) * float(wcol[u])
        assert bits([m.nextRank[U]])[0] == bits([acc])[0], (uniform, trial)
    G.close()


def test_small_path_falls_back_when_the_seed_row_is_too_long(amd):
    """The one-launch kernel of small.hip keeps the seed row's addend sequence (n restart addends + one addend per link
    INTO the seed) in LDS.  Multi-edges (the loader de-duplicates on (target, type), DataLoader.cs:64-70) can make a seed's
    in-list longer than n: such a call must take the general path and still equal the oracle bit for bit."""
    U, I = 2100, 4000                                     # (<= 6144 nodes, <= 4096 items: the graph qualifies)
    n = U + I
    node_id = np.arange(n, dtype=np.int64) * 5 + 3
    node_type = np.array([gg.NODE_USER] * U + [gg.NODE_ITEM] * I, dtype=np.uint8)
    lists = [[] for _ in range(n)]
    for v in range(I):                                   # every item links to user 0 twice (two types) and to one more user
        lists[U + v] += [(0, gg.EDGE_LIKE, 1.0), (0, gg.EDGE_AUTHORSHIP, 2.0), (1 + v % (U - 1), gg.EDGE_LIKE, 1.0)]
        lists[1 + v % (U - 1)].append((U + v, gg.EDGE_LIKE, 1.0))
        if v % 7 == 0:
            lists[0].append((U + v, gg.EDGE_LIKE, 1.0))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    dst, et, w = [], [], []
    for i in range(n):
        for (t, y, wt) in lists[i]:
            dst.append(t); et.append(y); w.append(wt)
        rowptr[i + 1] = len(dst)
    g = dict(node_id=node_id, node_type=node_type, rowptr=rowptr, dst=np.array(dst, dtype=np.int32),
             etype=np.array(et, dtype=np.uint8), w=np.array(w, dtype=np.float64))
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G)
    assert n + 2 * I > 2 * 6144                          # seed 0's sequence does not fit small.hip's LDS array
    for seed in (0, 1):                                  # seed 0: 8 000 in-links (general path); seed 1: a few (one launch)
        for T in (0, 3, 10):
            got = rec.Recommendation(seed, 0.15, T)
            ri, rs = F.recommend(seed, 0.15, T)
            assert [r[0] for r in got] == ri.tolist() and (bits([r[1] for r in got]) == bits(rs)).all(), (seed, T)
    G.close()


def test_error_paths_of_this_round(amd):
    """Damping factor outside [0, 1] (ranks would go negative: the exclusion marker and the ranking keys assume scores >= 0)
    is refused by the Recommendation entries with RWR_E_UNSUPPORTED while Model.run still follows the reference; the
    row-partitioned step refuses to run before rwr_part_begin; a refused call leaves the handle usable."""
    import ctypes as C
    from recommendersystems_amd import _lib
    g = gg.random_graph(5, n_users=50, n_items=200, n_likes=900, n_friend=40)
    F = FlatGraph(**g)
    G = dev_graph(amd, g)
    rec = amd.Recommender(G)
    for bad in (1.5, -0.25, float("nan")):
        with pytest.raises(amd.RwrError) as ei:
            rec.Recommendation(0, bad, 5)
        assert ei.value.status == _lib.RWR_E_UNSUPPORTED and "damping" in str(ei.value)
        with pytest.raises(amd.RwrError) as ei:
            rec.RecommendationBatch(np.array([0, 1], dtype=np.int32), bad, 5, 10)
        assert ei.value.status == _lib.RWR_E_UNSUPPORTED
    m = amd.Model(G, 1.5, 3)                              # d > 1: negative ranks, general kernels, still the reference's numbers
    m.run(4)
    r, _ = F.model_run(1.5, 3, 0, 4)
    assert (bits(m.rank) == bits(r)).all()
    lib = _lib.load()
    buf = (C.c_double * (len(g["node_id"]) * 2))()
    st = lib.rwr_part_step(G._handle(), C.cast(buf, C.c_void_p), C.cast(C.byref(buf, 8 * len(g["node_id"])), C.c_void_p), None)
    assert st == _lib.RWR_E_INVALID and b"rwr_part_begin" in lib.rwr_last_error()
    got = rec.Recommendation(0, 0.15, 5)                  # the handle is still good
    ri, rs = F.recommend(0, 0.15, 5)
    assert [x[0] for x in got] == ri.tolist() and (bits([x[1] for x in got]) == bits(rs)).all()
    G.close()
