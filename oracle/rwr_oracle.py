"""CPU oracle (literal restatement) of the reference's RWR hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recommendersystems_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED BY THE REFERENCE: ChangUk/RecommenderSystems ships no tests,
fixtures or golden vectors and its C# cannot be
built here (no dotnet/mono/csc).  This restatement is pinned instead by
  * the two hand-derived known-answer tests of SURVEY.md section 8c (dyadic values,
    exact in binary64, independent of summation order), and
  * bitwise agreement with the independent flat-array C restatement
    (``oracle/rwr_oracle.c``) on randomised graphs.

It walks the reference's code statement by statement, on the same containers
(dict of nodes, dict of link lists), with plain Python floats (IEEE binary64,
no FMA: CPython never contracts a*b+c).  Citations are file:line relative to
/root/reference/.
"""
from __future__ import annotations

import struct
from functools import cmp_to_key

# Recommenders/RWRBased/Recommender.cs:4-5 -- member order gives the integer values.
NODE_UNDEFINED, NODE_USER, NODE_ITEM, NODE_ETC = range(4)
(EDGE_UNDEFINED, EDGE_LIKE, EDGE_FRIENDSHIP, EDGE_FOLLOW, EDGE_MENTION,
 EDGE_AUTHORSHIP, EDGE_PURCHASE, EDGE_ETC) = range(8)


def widen_float(d: float) -> float:
    """(double)(float)d -- Recommender.cs:14 takes ``float dampingFactor`` and
    Recommender.cs:16 -> Model.cs:33 widens it to double."""
    return struct.unpack("<f", struct.pack("<f", d))[0]


class Node:
    """Graph.cs:4-17."""
    __slots__ = ("id", "type")

    def __init__(self, id: int, type: int = NODE_UNDEFINED):
        self.id = id
        self.type = type


class ForwardLink:
    """Graph.cs:19-35 (a mutable struct: copies are value copies)."""
    __slots__ = ("targetNode", "type", "weight")

    def __init__(self, targetNode: int, type: int = EDGE_UNDEFINED, weight: float = 0.0):
        self.targetNode = targetNode
        self.type = type
        self.weight = weight

    def copy(self) -> "ForwardLink":
        return ForwardLink(self.targetNode, self.type, self.weight)


class Graph:
    """Graph.cs:37-94."""

    def __init__(self, nodes: dict, edges: dict):
        self.nodes = nodes                      # Graph.cs:46
        self.edges = edges                      # Graph.cs:47
        self.graph = {}                         # Graph.cs:48

    def buildGraph(self) -> None:
        """Graph.cs:51-88."""
        for i in range(len(self.nodes)):        # :52
            forwardLinks = None                 # :53
            if i in self.edges:                 # :55
                nExplicitLinks = 0              # :57-61
                for forwardLink in self.edges[i]:
                    if forwardLink.type != EDGE_UNDEFINED:
                        nExplicitLinks += 1
                if nExplicitLinks > 0:          # :64
                    forwardLinks = [None] * nExplicitLinks  # :66
                    idx = 0
                    sumWeights = 0.0            # :70
                    for link in self.edges[i]:  # :71-77
                        if link.type != EDGE_UNDEFINED:
                            forwardLinks[idx] = link.copy()   # struct copy
                            idx += 1
                            sumWeights += link.weight
                    for f in range(nExplicitLinks):           # :80-81
                        forwardLinks[f].weight /= sumWeights
            self.graph[i] = forwardLinks        # :86

    def size(self) -> int:
        return len(self.nodes)                  # Graph.cs:91-93


class Model:
    """Model.cs:5-116.  ``dense_restart=True`` runs the reference's O(n^2)
    restart loops literally; False skips the addends whose restart weight is
    zero, which is bit-identical (adds +0.0 to a non-negative accumulator,
    SURVEY.md F8) and is asserted equal in tests/test_oracle.py."""

    def __init__(self, graph: Graph, dampingFactor: float, targetNode: int | None = None,
                 dense_restart: bool = True):
        self.graph = graph
        self.nNodes = graph.size()
        self.dampingFactor = dampingFactor
        n = self.nNodes
        self.dense_restart = dense_restart
        if targetNode is None:
            # Model.cs:14-31
            self.rank = [1.0] * n
            self.nextRank = [0.0] * n
            self.restart = [(1.0 / n) for _ in range(n)]
        else:
            # Model.cs:33-50 (rank[seed] = nNodes, an int widened to double)
            self.rank = [float(n) if i == targetNode else 0.0 for i in range(n)]
            self.nextRank = [0.0] * n
            self.restart = [1.0 if i == targetNode else 0.0 for i in range(n)]
        self._restart_nz = [r for r in range(n) if self.restart[r] != 0.0]

    def run(self, arg=None) -> int:
        """Model.cs:52-73.  int -> fixed iterations; float -> threshold; None ->
        threshold (1/double.MaxValue)*n (Model.cs:53).  Returns #deliverRanks."""
        if isinstance(arg, int) and not isinstance(arg, bool):
            for _ in range(arg):                # :69-72
                self.deliverRanks()
                self.updateRanks()
            return arg
        if arg is None:
            dbl_max = 1.7976931348623157e308
            threshold = (1 / dbl_max) * self.graph.size()   # :53
        else:
            threshold = arg
        it = 0
        while True:                             # :58-65
            self.deliverRanks()
            it += 1
            if self.checkConvergence(threshold):
                self.updateRanks()
                return it
            self.updateRanks()

    def deliverRanks(self) -> None:
        """Model.cs:76-100."""
        forwardLinks = self.graph.graph
        rank, nextRank, restart = self.rank, self.nextRank, self.restart
        n = self.nNodes
        rs = range(n) if self.dense_restart else self._restart_nz
        for i in range(n):                      # :78
            links = forwardLinks[i]             # :79
            if links is not None and len(links) > 0:   # :80
                rank_randomWalk = (1 - self.dampingFactor) * rank[i]   # :84
                for link in links:              # :85-88
                    nextRank[link.targetNode] += rank_randomWalk * link.weight
                rank_restart = rank[i] - rank_randomWalk               # :91
                for r in rs:                    # :92-93
                    nextRank[r] += rank_restart * restart[r]
            else:
                for r in rs:                    # :96-97
                    nextRank[r] += rank[i] * restart[r]

    def updateRanks(self) -> None:
        """Model.cs:103-108."""
        for i in range(self.nNodes):
            self.rank[i] = self.nextRank[i]
            self.nextRank[i] = 0.0

    def checkConvergence(self, threshold: float) -> bool:
        """Model.cs:110-115."""
        diff = 0.0
        for i in range(self.nNodes):
            a, b = self.rank[i], self.nextRank[i]
            diff += (a - b) if a > b else (b - a)
        return diff < threshold


def _cmp(x, y) -> int:
    return (x > y) - (x < y)


class Recommender:
    """Recommender.cs:7-52."""

    def __init__(self, graph: Graph, dense_restart: bool = True):
        self.graph = graph
        self.dense_restart = dense_restart

    def Recommendation(self, idxTargetUser: int, dampingFactor: float, nIteration: int,
                       topN: int | None = None):
        if topN is not None:
            # Recommender.cs:42-51: the Count == topN test never fires for topN <= 0
            recommendation = self.Recommendation(idxTargetUser, dampingFactor, nIteration)
            top = []
            for kv in recommendation:
                top.append(kv)
                if len(top) == topN:
                    break
            return top
        graph = self.graph
        d = widen_float(dampingFactor)          # float parameter, :14 -> Model.cs:33
        model = Model(graph, d, idxTargetUser, dense_restart=self.dense_restart)   # :16
        model.run(int(nIteration))              # :17
        linksOfTargetUser = []                  # :20-24 (raw edges; KeyError == KeyNotFoundException)
        for link in graph.edges[idxTargetUser]:
            if link.type == EDGE_LIKE:
                linksOfTargetUser.append(link.targetNode)
        recommendation = []                     # :27-31
        for i in range(model.nNodes):
            if graph.nodes[i].type == NODE_ITEM and i not in linksOfTargetUser:
                recommendation.append((graph.nodes[i].id, model.rank[i]))

        def compare(one, another):              # :35-38 (double.CompareTo, long.CompareTo)
            result = _cmp(one[1], another[1]) * -1
            return result if result != 0 else _cmp(one[0], another[0]) * -1
        recommendation.sort(key=cmp_to_key(compare))
        return recommendation


def evaluate(recommendation, testSet):
    """TweetRecommender/Experiment.cs:121-128: hits and the running-precision sum over the FULL ranked list.
    Returns (nHits, sumPrecision); the harness reports nHits and sumPrecision / nHits (:131-138)."""
    nHits = 0
    sumPrecision = 0.0
    for i in range(len(recommendation)):                      # :123
        if recommendation[i][0] in testSet:                   # :124
            nHits += 1
            sumPrecision += float(nHits) / (i + 1)            # :126  (double)nHits / (i + 1)
    return nHits, sumPrecision


# ---------------------------------------------------------------------------
# helpers shared by tests: flat (CSR) <-> dictionary form, hex encoding
# ---------------------------------------------------------------------------

def from_flat(node_id, node_type, rowptr, dst, etype, w, has_key=None):
    """Build the reference's containers from the flat layout the C-ABI takes
    (include/rwr.h).  ``has_key[i] == False`` models ``!edges.ContainsKey(i)``;
    by default every node has a (possibly empty) list."""
    n = len(node_id)
    nodes = {i: Node(int(node_id[i]), int(node_type[i])) for i in range(n)}
    edges = {}
    for i in range(n):
        if has_key is not None and not has_key[i]:
            continue
        edges[i] = [ForwardLink(int(dst[e]), int(etype[e]), float(w[e]))
                    for e in range(int(rowptr[i]), int(rowptr[i + 1]))]
    return nodes, edges


def f64_hex(x: float) -> str:
    return struct.pack(">d", x).hex()


def hex_f64(h: str) -> float:
    return struct.unpack(">d", bytes.fromhex(h))[0]
