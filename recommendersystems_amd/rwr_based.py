"""Host-side mirror of the reference's public surface ``namespace Recommenders.RWRBased``
(Recommenders/RWRBased/{Graph,Model,Recommender}.cs) over the librwr C-ABI.

Same names, argument meaning and error behaviour as the C# types, so the parity tests read
like the reference's caller (TweetRecommender/Experiment.cs:104-109).  The reference is
C#, which cannot be built in this image; the C# shim that a .NET host would use is under
csharp/ and P/Invokes exactly the functions used here.

All arithmetic happens in hand-written HIP kernels on the GPU; this module only flattens the
reference's containers into the SoA layout of include/rwr.h and wraps results.
"""
from __future__ import annotations

import ctypes as C
import enum
import itertools
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _lib


class NodeType(enum.IntEnum):
    """Recommender.cs:4."""
    UNDEFINED = 0
    USER = 1
    ITEM = 2
    ETC = 3


class EdgeType(enum.IntEnum):
    """Recommender.cs:5."""
    UNDEFINED = 0
    LIKE = 1
    FRIENDSHIP = 2
    FOLLOW = 3
    MENTION = 4
    AUTHORSHIP = 5
    PURCHASE = 6
    ETC = 7


class Node:
    """struct Node (Graph.cs:4-17)."""
    __slots__ = ("id", "type")

    def __init__(self, id: int, type: NodeType = NodeType.UNDEFINED):
        self.id = int(id)
        self.type = NodeType(type)


class ForwardLink:
    """struct ForwardLink (Graph.cs:19-35); fields are public and mutable, as the harness
    relies on (DataLoader.cs:66,424; Experiment.cs:90-97)."""
    __slots__ = ("targetNode", "type", "weight")

    def __init__(self, targetNode: int, type=EdgeType.UNDEFINED, weight: Optional[float] = None):
        # ForwardLink(int, double) and ForwardLink(int, EdgeType, double)
        if weight is None:
            type, weight = EdgeType.UNDEFINED, type
        self.targetNode = int(targetNode)
        self.type = EdgeType(type)
        self.weight = float(weight)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Graph:
    """class Graph (Graph.cs:37-94).  ``nodes``/``edges`` are the caller's dictionaries
    (kept by reference, Graph.cs:46-47); ``buildGraph()`` hands the RAW links to
    rwr_graph_create, which filters/normalises/transposes on the device."""

    incremental_rebuilds = 0     # buildGraph() calls (all instances) that went through rwr_graph_update_links

    def __init__(self, nodes: Dict[int, Node], edges: Dict[int, List[ForwardLink]], *, mode: Optional[str] = None,
                 device: int = -1, tile_seeds: int = 0, tile_group: int = 0, profile: bool = False,
                 workspace_bytes: int = 0, seed_row_kernel: Optional[str] = None):
        self.nodes = nodes
        self.edges = edges
        self._graph_cache = None
        self._h = C.c_void_p()
        self._opts = _lib.rwr_opts(C.sizeof(_lib.rwr_opts), device,
                                   {None: -1, "exact": _lib.RWR_MODE_EXACT}[mode],
                                   tile_seeds, tile_group, 1 if profile else 0, workspace_bytes,
                                   {None: 0, "auto": 0, "fold": 1, "scan": 2, "simple": 3}[seed_row_kernel], 0)
        self._flat = None
        self._desc = None       # (rwr_graph_desc bytes of _flat: _marshal_graphs)
        self._sent = None

    # -- flat constructors (the layout of include/rwr.h), used by the bench for big graphs
    @classmethod
    def from_flat(cls, node_id, node_type, rowptr, dst, etype, w, **opts) -> "Graph":
        g = cls(None, None, **opts)
        g._flat = (np.ascontiguousarray(node_id, dtype=np.int64), np.ascontiguousarray(node_type, dtype=np.uint8),
                   np.ascontiguousarray(rowptr, dtype=np.int64), np.ascontiguousarray(dst, dtype=np.int32),
                   np.ascontiguousarray(etype, dtype=np.uint8), np.ascontiguousarray(w, dtype=np.float64))
        return g

    def _flatten(self):
        n = len(self.nodes)
        node_id = np.empty(n, dtype=np.int64)
        node_type = np.empty(n, dtype=np.uint8)
        for i in range(n):
            nd = self.nodes[i]                 # keys must be 0..n-1 (Graph.cs:52,55)
            node_id[i] = nd.id
            node_type[i] = int(nd.type)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        for i in range(n):
            rowptr[i + 1] = rowptr[i] + (len(self.edges[i]) if i in self.edges else 0)
        m = int(rowptr[n])
        dst = np.empty(m, dtype=np.int32)
        etype = np.empty(m, dtype=np.uint8)
        w = np.empty(m, dtype=np.float64)
        e = 0
        for i in range(n):
            if i in self.edges:
                for l in self.edges[i]:
                    dst[e] = l.targetNode
                    etype[e] = int(l.type)
                    w[e] = l.weight
                    e += 1
        return node_id, node_type, rowptr, dst, etype, w

    def buildGraph(self) -> None:
        """Graph.buildGraph (Graph.cs:51-88) -> rwr_graph_create.

        Called again on the same object after the caller mutated its dictionaries (the harness relabels link types
        between runs, Experiment.cs:84-101): when nodes, list lengths and targets are unchanged, only the links whose
        type or weight differ are sent (rwr_graph_update_links); the device state is the same either way."""
        lib = _lib.load()
        flat = self._flat if self._flat is not None else self._flatten()
        node_id, node_type, rowptr, dst, etype, w = flat
        if self._h and self._flat is None and self._sent is not None:
            o_id, o_type, o_rowptr, o_dst, o_etype, o_w = self._sent
            if (o_rowptr.shape == rowptr.shape and (o_rowptr == rowptr).all() and (o_dst == dst).all()
                    and (o_id == node_id).all() and (o_type == node_type).all()):
                changed = np.flatnonzero((o_etype != etype) | (o_w.view(np.uint64) != w.view(np.uint64))).astype(np.int64)
                self.updateLinks(changed, etype[changed], w[changed])
                self._sent = flat
                Graph.incremental_rebuilds += 1
                return
        if self._h:
            lib.rwr_graph_destroy(self._h)
            self._h = C.c_void_p()
        self._n = int(node_id.shape[0])
        self._rowptr = rowptr
        _lib.check(lib.rwr_graph_create(self._n, _p(node_id, C.c_int64), _p(node_type, C.c_uint8),
                                        _p(rowptr, C.c_int64), _p(dst, C.c_int32), _p(etype, C.c_uint8),
                                        _p(w, C.c_double), C.byref(self._opts), C.byref(self._h)))
        self._sent = flat if self._flat is None else None     # (dictionary graphs: fresh arrays, kept for the next diff)
        self._graph_cache = None

    def updateLinks(self, link_index, etype=None, w=None) -> None:
        """rwr_graph_update_links: new type / raw weight for the raw links at the given flattened positions, then the
        device-side rebuild.  (Flat graphs: the caller's arrays are not touched; keep them in step yourself.)"""
        idx = np.ascontiguousarray(link_index, dtype=np.int64)
        et = None if etype is None else np.ascontiguousarray(etype, dtype=np.uint8)
        ww = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
        if (et is not None and et.shape != idx.shape) or (ww is not None and ww.shape != idx.shape):
            raise ValueError("etype / w must have one entry per link index")
        _lib.check(_lib.load().rwr_graph_update_links(
            self._handle(), int(idx.shape[0]), _p(idx, C.c_int64),
            None if et is None else _p(et, C.c_uint8), None if ww is None else _p(ww, C.c_double)))
        self._graph_cache = None

    def size(self) -> int:
        """Graph.size() (Graph.cs:91-93)."""
        return len(self.nodes) if self.nodes is not None else int(self._flat[0].shape[0])

    def _handle(self):
        if not self._h:
            raise RuntimeError("Graph.buildGraph() has not been called")
        return self._h

    def normalized(self) -> Tuple[np.ndarray, np.ndarray]:
        """(w_norm per raw link, dangling per node) -- rwr_graph_get_normalized."""
        lib = _lib.load()
        m = int(self._rowptr[-1])
        wn = np.zeros(max(m, 1), dtype=np.float64)
        dg = np.zeros(self._n, dtype=np.uint8)
        _lib.check(lib.rwr_graph_get_normalized(self._handle(), _p(wn, C.c_double), _p(dg, C.c_uint8)))
        return wn[:m], dg

    @property
    def graph(self) -> Dict[int, Optional[List[ForwardLink]]]:
        """The public field Graph.graph (Graph.cs:43): normalised explicit links per node,
        None for dangling nodes; materialised lazily from the device."""
        if self._graph_cache is None:
            wn, dg = self.normalized()
            out = {}
            flat = self._flat if self._flat is not None else self._flatten()
            _, _, rowptr, dst, etype, _ = flat
            for i in range(self._n):
                if dg[i]:
                    out[i] = None
                else:
                    out[i] = [ForwardLink(int(dst[e]), EdgeType(int(etype[e])), float(wn[e]))
                              for e in range(int(rowptr[i]), int(rowptr[i + 1])) if etype[e] != 0]
            self._graph_cache = out
        return self._graph_cache

    def stats(self) -> dict:
        st = _lib.rwr_stats()
        st.struct_size = C.sizeof(_lib.rwr_stats)
        _lib.check(_lib.load().rwr_get_stats(self._handle(), C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def reset_stats(self) -> None:
        _lib.check(_lib.load().rwr_reset_stats(self._handle()))

    def close(self) -> None:
        if self._h:
            _lib.load().rwr_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):   # the reference types have no Dispose: release on finalisation
        try:
            self.close()
        except Exception:
            pass


def _desc_raw(flat) -> bytes:
    node_id, node_type, rowptr, dst, etype, w = flat
    return bytes(_lib.rwr_graph_desc(int(node_id.shape[0]), 0, _p(node_id, C.c_int64), _p(node_type, C.c_uint8),
                                     _p(rowptr, C.c_int64), _p(dst, C.c_int32), _p(etype, C.c_uint8), _p(w, C.c_double)))


def _marshal_graphs(graphs, testSets):
    """The arguments of rwr_eval_graphs for a batch: the descriptor array, the test sets as (ptr, ids), and the arrays that
    must stay alive during the call.  A Graph made from flat arrays keeps its descriptor (its six pointers) with the arrays,
    so that a batch of such graphs costs the interpreter one bytes.join -- with ten host threads behind one interpreter lock
    (bench.py --config C1) the per-graph ctypes conversions were most of the call."""
    K = len(graphs)
    alive, raw = [], []
    for g in graphs:
        if g._flat is not None:
            if g._desc is None:
                g._desc = _desc_raw(g._flat)
            raw.append(g._desc)
        else:
            flat = g._flatten()                 # (dictionary graphs: fresh arrays every time, the caller may have changed the lists)
            alive.append(flat)
            raw.append(_desc_raw(flat))
    descs = (_lib.rwr_graph_desc * max(K, 1))()
    if K:
        C.memmove(descs, b"".join(raw), K * C.sizeof(_lib.rwr_graph_desc))
    ptr = np.zeros(K + 1, dtype=np.int64)
    if K:
        np.cumsum(np.fromiter(map(len, testSets), dtype=np.int64, count=K), out=ptr[1:])
    total = int(ptr[K])
    if total == 0:
        ids = np.zeros(1, dtype=np.int64)
    elif all(isinstance(t, np.ndarray) for t in testSets):
        ids = np.ascontiguousarray(np.concatenate(testSets), dtype=np.int64)
    else:
        ids = np.fromiter(itertools.chain.from_iterable(testSets), dtype=np.int64, count=total)
    return descs, ptr, ids, alive


def EvaluateGraphs(graphs, seeds, dampingFactor: float, nIteration: int, testSets, *, device: int = -1):
    """The loop body of the reference's harness (Experiment.cs:69-134) for MANY graphs at once (rwr_eval_graphs):
    graphs[k].buildGraph(); Recommender(graphs[k]).RecommendationEval(seeds[k], d, T, testSets[k]) for every k, with one
    build launch, one iteration launch and one evaluation launch for all ego-network-sized graphs of the batch.  The Graph
    objects are NOT built by this call (no device handle is left behind).  Returns (nHits[K], sumPrecision[K], listLen[K])."""
    K = len(graphs)
    if len(seeds) != K or len(testSets) != K:
        raise ValueError("one seed and one test set per graph")
    descs, ptr, ids, _alive = _marshal_graphs(graphs, testSets)
    seeds_a = np.ascontiguousarray(seeds, dtype=np.int32)
    hits = np.zeros(K, dtype=np.int64)
    sp = np.zeros(K, dtype=np.float64)
    ln = np.zeros(K, dtype=np.int64)
    opts = _lib.rwr_opts(C.sizeof(_lib.rwr_opts), device, -1, 0, 0, 0, 0, 0, 0)
    _lib.check(_lib.load().rwr_eval_graphs(K, descs, _p(seeds_a, C.c_int32), C.c_float(dampingFactor), int(nIteration),
                                           _p(ptr, C.c_int64), _p(ids, C.c_int64), C.byref(opts), _p(hits, C.c_int64),
                                           _p(sp, C.c_double), _p(ln, C.c_int64)))
    return hits, sp, ln


class Model:
    """class Model (Model.cs:5-116) backed by rwr_model_run."""

    def __init__(self, graph: Graph, dampingFactor: float, targetNode: Optional[int] = None):
        self.graph = graph
        self.nNodes = graph.size()
        self.dampingFactor = float(dampingFactor)
        self._seed = -1 if targetNode is None else int(targetNode)
        n = self.nNodes
        if targetNode is None:                                  # Model.cs:14-31
            self.rank = np.ones(n)
            self.restart = np.full(n, 1.0 / n)
        else:                                                   # Model.cs:33-50
            self.rank = np.zeros(n)
            self.restart = np.zeros(n)
            if 0 <= targetNode < n:
                self.rank[targetNode] = float(n)
                self.restart[targetNode] = 1.0
        self.nextRank = np.zeros(n)
        self.iterations = 0

    def _ctor_state(self) -> bool:
        """rank / nextRank / restart are still what the constructor left (then the whole run can stay on the device)."""
        n = self.nNodes
        if self.nextRank.shape != (n,) or self.rank.shape != (n,) or np.any(self.nextRank != 0):
            return False
        if self._seed < 0:
            return bool(np.all(self.rank == 1.0))
        if not (0 <= self._seed < n):
            return bool(np.all(self.rank == 0.0))
        r = self.rank
        return bool(r[self._seed] == float(n) and np.count_nonzero(r) == (1 if n else 0))

    def run(self, arg=None) -> None:
        """run(int) / run(double) / run()  (Model.cs:68-73, 57-66, 52-55).  From the constructor's state the whole loop
        runs on the device (rwr_model_run); on a model that has already been advanced -- the reference's run() continues
        from the current rank -- it is driven step by step through deliverRanks / updateRanks / checkConvergence."""
        lib = _lib.load()
        if isinstance(arg, (int, np.integer)) and not isinstance(arg, bool):
            mode, value = _lib.RWR_RUN_ITERATIONS, float(arg)
        elif arg is None:
            mode, value = _lib.RWR_RUN_DEFAULT_THRESHOLD, 0.0
        else:
            mode, value = _lib.RWR_RUN_THRESHOLD, float(arg)
        if not self._ctor_state():
            if mode == _lib.RWR_RUN_ITERATIONS:                  # Model.cs:68-73
                for _ in range(int(value)):
                    self.deliverRanks()
                    self.updateRanks()
                self.iterations = int(value)
                return
            threshold = (1.0 / 1.7976931348623157e308) * self.nNodes if mode == _lib.RWR_RUN_DEFAULT_THRESHOLD else value
            it = 0
            while True:                                          # Model.cs:57-66
                self.deliverRanks()
                it += 1
                done = self.checkConvergence(threshold)
                self.updateRanks()
                if done:
                    break
            self.iterations = it
            return
        out = np.zeros(self.nNodes, dtype=np.float64)
        it = C.c_int64(0)
        _lib.check(lib.rwr_model_run(self.graph._handle(), self._seed, self.dampingFactor, mode, value,
                                     _p(out, C.c_double), C.byref(it)))
        self.rank = out
        self.nextRank = np.zeros(self.nNodes)
        self.iterations = int(it.value)

    def deliverRanks(self) -> None:
        """Model.deliverRanks (Model.cs:76-100): nextRank <- one propagation of the current rank (rwr_model_deliver)."""
        n = self.nNodes
        if np.any(self.nextRank != 0):
            raise RuntimeError("deliverRanks() on a non-zero nextRank (the reference would add on top of it): call updateRanks() first")
        expect = np.full(n, 1.0 / n) if self._seed < 0 else np.zeros(n)
        if 0 <= self._seed < n:
            expect[self._seed] = 1.0
        if not np.array_equal(np.asarray(self.restart, dtype=np.float64), expect):
            raise RuntimeError("Model.restart was modified: only the constructors' restart vectors are supported")
        rank = np.ascontiguousarray(self.rank, dtype=np.float64)
        out = np.empty(n, dtype=np.float64)
        _lib.check(_lib.load().rwr_model_deliver(self.graph._handle(), self._seed, self.dampingFactor,
                                                 _p(rank, C.c_double), _p(out, C.c_double)))
        self.nextRank = out

    def updateRanks(self) -> None:
        """Model.updateRanks (Model.cs:103-108)."""
        self.rank = np.array(self.nextRank, dtype=np.float64)
        self.nextRank = np.zeros(self.nNodes)

    def checkConvergence(self, threshold: float) -> bool:
        """Model.checkConvergence (Model.cs:110-115): sequential sum of |rank - nextRank| < threshold."""
        if self.nNodes == 0:
            return 0.0 < threshold
        diff = np.cumsum(np.abs(np.asarray(self.rank, dtype=np.float64) - self.nextRank))   # cumsum adds left to right
        return bool(diff[-1] < threshold)


class Recommender:
    """class Recommender (Recommender.cs:7-52)."""

    def __init__(self, graph: Graph):
        self.graph = graph

    def Recommendation(self, idxTargetUser: int, dampingFactor: float, nIteration: int,
                       topN: Optional[int] = None) -> List[Tuple[int, float]]:
        """Recommendation(int, float, int[, int topN]) -> list of (item id, score), i.e. the
        reference's List<KeyValuePair<long,double>> (Recommender.cs:14-40, 42-51)."""
        g = self.graph
        if g.edges is not None and idxTargetUser not in g.edges:
            raise KeyError(idxTargetUser)        # graph.edges[idxTargetUser], Recommender.cs:21
        lib = _lib.load()
        cap = g.size() if topN is None or topN <= 0 else min(g.size(), int(topN))
        ids = np.empty(max(cap, 1), dtype=np.int64)
        sc = np.empty(max(cap, 1), dtype=np.float64)
        cnt = C.c_int64(cap)
        _lib.check(lib.rwr_recommend(g._handle(), int(idxTargetUser), C.c_float(dampingFactor), int(nIteration),
                                     0 if topN is None else int(topN), _p(ids, C.c_int64), _p(sc, C.c_double),
                                     C.byref(cnt)))
        c = int(cnt.value)
        return list(zip(ids[:c].tolist(), sc[:c].tolist()))

    def RecommendationArrays(self, idxTargetUser: int, dampingFactor: float, nIteration: int,
                             topN: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """The same ranked list as Recommendation(), as two arrays (ids, scores) -- for graphs whose full list has
        millions of entries, where building Python tuples costs far more than the GPU call."""
        g = self.graph
        if g.edges is not None and idxTargetUser not in g.edges:
            raise KeyError(idxTargetUser)
        cap = g.size() if topN is None or topN <= 0 else min(g.size(), int(topN))
        ids = np.empty(max(cap, 1), dtype=np.int64)
        sc = np.empty(max(cap, 1), dtype=np.float64)
        cnt = C.c_int64(cap)
        _lib.check(_lib.load().rwr_recommend(g._handle(), int(idxTargetUser), C.c_float(dampingFactor), int(nIteration),
                                             0 if topN is None else int(topN), _p(ids, C.c_int64), _p(sc, C.c_double),
                                             C.byref(cnt)))
        c = int(cnt.value)
        return ids[:c], sc[:c]

    def RecommendationEval(self, idxTargetUser: int, dampingFactor: float, nIteration: int, testSet):
        """Recommendation + the harness's evaluation of it (Experiment.cs:109,121-128) without bringing the list
        to the host: returns (nHits, sumPrecision, len(list)); MAP contribution = sumPrecision / nHits."""
        g = self.graph
        if g.edges is not None and idxTargetUser not in g.edges:
            raise KeyError(idxTargetUser)
        t = np.ascontiguousarray(sorted(testSet), dtype=np.int64)
        hits, sp, ln = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        _lib.check(_lib.load().rwr_recommend_eval(g._handle(), int(idxTargetUser), C.c_float(dampingFactor),
                                                  int(nIteration), _p(t, C.c_int64), len(t), C.byref(hits),
                                                  C.byref(sp), C.byref(ln)))
        return int(hits.value), float(sp.value), int(ln.value)

    def RecommendationEvalBatch(self, seeds, dampingFactor: float, nIteration: int, testSets):
        """RecommendationEval for K seeds of this graph, each with its own test set (rwr_recommend_eval_batch):
        returns (nHits[K], sumPrecision[K], listLen[K])."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        K = int(seeds.shape[0])
        if len(testSets) != K:
            raise ValueError("one test set per seed")
        ptr = np.zeros(K + 1, dtype=np.int64)
        for k, t in enumerate(testSets):
            ptr[k + 1] = ptr[k] + len(t)
        ids = np.ascontiguousarray([x for t in testSets for x in t], dtype=np.int64) if ptr[K] else np.zeros(1, dtype=np.int64)
        hits = np.zeros(K, dtype=np.int64)
        sp = np.zeros(K, dtype=np.float64)
        ln = np.zeros(K, dtype=np.int64)
        _lib.check(_lib.load().rwr_recommend_eval_batch(self.graph._handle(), _p(seeds, C.c_int32), K, C.c_float(dampingFactor),
                                                        int(nIteration), _p(ptr, C.c_int64), _p(ids, C.c_int64),
                                                        _p(hits, C.c_int64), _p(sp, C.c_double), _p(ln, C.c_int64)))
        return hits, sp, ln

    def RecommendationBatch(self, seeds, dampingFactor: float, nIteration: int, topN: int):
        """Batch entry (an addition, see include/rwr.h): (ids[K,topN], scores[K,topN], counts[K])."""
        lib = _lib.load()
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        K = int(seeds.shape[0])
        ids = np.zeros((K, topN), dtype=np.int64)
        sc = np.zeros((K, topN), dtype=np.float64)
        counts = np.zeros(K, dtype=np.int32)
        _lib.check(lib.rwr_recommend_batch(self.graph._handle(), _p(seeds, C.c_int32), K, C.c_float(dampingFactor),
                                           int(nIteration), int(topN), _p(ids, C.c_int64), _p(sc, C.c_double),
                                           _p(counts, C.c_int32)))
        return ids, sc, counts
