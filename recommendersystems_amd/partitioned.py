"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Two regimes:

* seed sharding (graph fits one GPU, BASELINE.json config 4): seeds are independent -- the reference creates a
  fresh Model per call (Recommender.cs:16) -- so every rank holds the whole graph and takes a contiguous block
  of the batch.  NO data-path collective; `gather_seed_shards` only collects the small result tables.

* row partition (graph beyond one GPU, config 5): the transition matrix is split by SOURCE rows into contiguous
  slabs balanced by link count.  A rank computes a partial next rank matrix over ALL rows from its slab of the
  current one (plus its slab's restart mass, added locally at the seeds' rows), and since it will only ever READ
  its own slab again, every power-iteration step ends with a REDUCE-SCATTER of the partial matrix by slabs -- half
  the bytes of an all-reduce and no second collective for the restart scalars; only the last step all-reduces,
  because ranking needs every row.  The step is enqueued asynchronously on the stream the collective runs on: no
  host synchronisation inside the loop.  This is the one real exchange step of the path.  An ADDITION to the
  reference (which has no distributed mode); parity with the single-GPU result is to tolerance, not bitwise
  (partial sums re-associate).

The compute backend is the HIP library (`HipSlabBackend`); tests inject a numpy stand-in to exercise this host
logic on CPU with gloo.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------- seed sharding

def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` items for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_seed_shards(ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, total: int, group=None):
    """Collects the per-rank (K_r x top_n) result tables of a seed-sharded batch into the full
    (total x top_n) tables on every rank.  Results only -- the power iteration itself never communicates."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ids, scores, counts
    rank = dist.get_rank(group)
    top_n = ids.shape[1]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    full_ids = torch.zeros((total, top_n), dtype=torch.int64, device=dev)
    full_sc = torch.zeros((total, top_n), dtype=torch.float64, device=dev)
    full_cnt = torch.zeros((total,), dtype=torch.int32, device=dev)
    lo, hi = shard_bounds(total, world, rank)
    assert hi - lo == ids.shape[0]
    full_ids[lo:hi] = torch.from_numpy(ids).to(dev)
    full_sc[lo:hi] = torch.from_numpy(scores).to(dev)
    full_cnt[lo:hi] = torch.from_numpy(counts).to(dev)
    # disjoint supports: a sum is an exact gather
    dist.all_reduce(full_ids, group=group)
    dist.all_reduce(full_sc, group=group)
    dist.all_reduce(full_cnt, group=group)
    return full_ids.cpu().numpy(), full_sc.cpu().numpy(), full_cnt.cpu().numpy()


# ------------------------------------------------------------------------------------------- row partition

def slab_bounds(rowptr: np.ndarray, world: int) -> np.ndarray:
    """world+1 node boundaries so that every slab holds about the same number of links (deterministic)."""
    n = len(rowptr) - 1
    total = int(rowptr[-1])
    targets = (np.arange(1, world, dtype=np.int64) * total) // world
    cuts = np.searchsorted(rowptr[1:], targets, side="left") + 1 if world > 1 else np.array([], dtype=np.int64)
    b = np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)
    return np.maximum.accumulate(b)


def slab_graph(flat: dict, lo: int, hi: int) -> dict:
    """The rank-local graph: FULL node arrays, out-links only for rows [lo, hi) (rowptr flat elsewhere)."""
    rp = np.asarray(flat["rowptr"], dtype=np.int64)
    e0, e1 = int(rp[lo]), int(rp[hi])
    rowptr = np.empty_like(rp)
    rowptr[:lo] = 0
    rowptr[lo:hi + 1] = rp[lo:hi + 1] - e0
    rowptr[hi + 1:] = e1 - e0
    return dict(node_id=flat["node_id"], node_type=flat["node_type"], rowptr=rowptr,
                dst=flat["dst"][e0:e1], etype=flat["etype"][e0:e1], w=flat["w"][e0:e1])


class HipSlabBackend:
    """rwr_part_* of include/rwr.h on this process's GPU; rank matrices are torch CUDA tensors so that
    torch.distributed (RCCL) can all-reduce them in place."""

    def __init__(self, local_flat: dict, lo: int, hi: int, device: int = -1, **opts):
        import torch
        from . import _lib
        from .rwr_based import Graph
        self.torch, self._lib = torch, _lib
        self.lo, self.hi = lo, hi
        self.n = int(len(local_flat["node_id"]))
        self.graph = Graph.from_flat(**local_flat, device=device, **opts)
        self.graph.buildGraph()
        self.dev = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)

    def begin(self, seeds: np.ndarray, d: float):
        lib, torch = self._lib.load(), self.torch
        K = int(len(seeds))
        G = 1
        while G < K:
            G <<= 1
        self.K, self.G = K, G
        x = torch.empty(self.n * G, dtype=torch.float64, device=self.dev)
        y = torch.empty(self.n * G, dtype=torch.float64, device=self.dev)
        r = torch.zeros(G, dtype=torch.float64, device=self.dev)
        s = np.ascontiguousarray(seeds, dtype=np.int32)
        g_out = C.c_int32(0)
        torch.cuda.synchronize()
        self._lib.check(lib.rwr_part_begin(self.graph._handle(), self.lo, self.hi, s.ctypes.data_as(C.POINTER(C.c_int32)),
                                           K, C.c_double(d), C.c_void_p(x.data_ptr()), C.byref(g_out)))
        assert g_out.value == G
        return x, y, r

    def step(self, x, y):
        """One whole local step (partial y over all rows + this slab's restart mass at the seeds' rows), enqueued on
        torch's current stream -- the one the collective that follows is ordered after; no host synchronisation."""
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        self._lib.check(self._lib.load().rwr_part_step(self.graph._handle(), C.c_void_p(x.data_ptr()),
                                                       C.c_void_p(y.data_ptr()), C.c_void_p(stream)))

    def local_step(self, x, y, r):
        self.torch.cuda.synchronize()
        self._lib.check(self._lib.load().rwr_part_local_step(self.graph._handle(), C.c_void_p(x.data_ptr()),
                                                             C.c_void_p(y.data_ptr()), C.c_void_p(r.data_ptr())))

    def finish_step(self, y, r):
        self.torch.cuda.synchronize()
        self._lib.check(self._lib.load().rwr_part_finish_step(self.graph._handle(), C.c_void_p(y.data_ptr()),
                                                              C.c_void_p(r.data_ptr())))

    def rank(self, x, top_n: int):
        self.torch.cuda.synchronize()
        ids = np.zeros((self.K, top_n), dtype=np.int64)
        sc = np.zeros((self.K, top_n), dtype=np.float64)
        cnt = np.zeros(self.K, dtype=np.int32)
        self._lib.check(self._lib.load().rwr_part_rank(
            self.graph._handle(), C.c_void_p(x.data_ptr()), top_n, ids.ctypes.data_as(C.POINTER(C.c_int64)),
            sc.ctypes.data_as(C.POINTER(C.c_double)), cnt.ctypes.data_as(C.POINTER(C.c_int32))))
        return ids, sc, cnt

    def to_exchange(self, t):      # tensors handed to torch.distributed
        return t

    def from_numpy(self, a):
        return self.torch.from_numpy(a).to(self.dev)


MAX_TILE = 64      # seeds per rank-matrix tile of the row-partitioned mode (rwr_part_begin)


def reduce_scatter_slabs(t, G: int, bounds, rank: int, group=None, recv=None):
    """Reduce-scatter(sum) of the partial rank matrix t[n * G] by the (uneven) node slabs: afterwards rank r holds the
    complete rows [bounds[r], bounds[r + 1]) of the sum in place; the rest of t is unspecified (never read again).

    ONE collective per call: an all-to-all with uneven splits -- every rank sends slab j of its partial matrix to rank j,
    i.e. (w - 1) / w of the matrix leaves each rank, on a full mesh one slab per peer link (the direct
    reduce-scatter) -- followed by a LOCAL sum of the w received pieces in rank order, which makes the result independent
    of the collective's internal schedule (deterministic from run to run).  `recv` (optional) is a reusable buffer of at
    least w * (own slab) * G elements; returns the buffer used."""
    import torch
    import torch.distributed as dist
    world = len(bounds) - 1
    in_splits = [(int(bounds[r + 1]) - int(bounds[r])) * G for r in range(world)]
    own = in_splits[rank]
    if recv is None or recv.numel() < world * own or recv.dtype != t.dtype or recv.device != t.device:
        recv = torch.empty(max(world * own, 1), dtype=t.dtype, device=t.device)
    out = recv[:world * own]
    dist.all_to_all_single(out, t, output_split_sizes=[own] * world, input_split_sizes=in_splits, group=group)
    if own > 0:
        pieces = out.view(world, own)
        dst = t[int(bounds[rank]) * G:int(bounds[rank + 1]) * G]
        dst.copy_(pieces[0])
        for r in range(1, world):                     # rank order: a fixed summation order
            dst.add_(pieces[r])
    return recv


class PartitionedRecommender:
    """Recommendation over a source-row-partitioned transition matrix (one slab per rank)."""

    def __init__(self, flat: dict, rank: int = 0, world: int = 1, group=None, backend_factory=None, **opts):
        self.rank, self.world, self.group = rank, world, group
        self.bounds = slab_bounds(np.asarray(flat["rowptr"], dtype=np.int64), world)
        self.lo, self.hi = int(self.bounds[rank]), int(self.bounds[rank + 1])
        local = slab_graph(flat, self.lo, self.hi)
        factory = backend_factory or HipSlabBackend
        self.backend = factory(local, self.lo, self.hi, **opts)
        self.exchanged_bytes = 0          # payload handed to collectives by this rank (measurement)
        self.collectives = 0              # data-path collectives issued by this rank (one per iteration)
        self._recv = None                 # receive buffer of the slab exchange (w pieces of this rank's slab)

    def _all_reduce(self, t):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, group=self.group)      # sum; RCCL over xGMI on GPUs
            self.collectives += 1

    def _reduce_scatter(self, t, G):
        if self.world > 1:
            self._recv = reduce_scatter_slabs(t, G, self.bounds, self.rank, self.group, self._recv)
            self.collectives += 1

    def RecommendationBatch(self, seeds, dampingFactor: float, nIteration: int, topN: int):
        """Same result (to tolerance) on every rank as Recommender.RecommendationBatch on one GPU.
        dampingFactor crosses as float and is widened (Recommender.cs:14,16 -> Model.cs:33).  Batches beyond one tile
        (64 seeds) are run tile after tile."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        if len(seeds) > MAX_TILE:
            parts = [self.RecommendationBatch(seeds[i:i + MAX_TILE], dampingFactor, nIteration, topN)
                     for i in range(0, len(seeds), MAX_TILE)]
            return tuple(np.concatenate([p[j] for p in parts]) for j in range(3))
        be = self.backend
        d = float(np.float32(dampingFactor))
        x, y, _ = be.begin(seeds, d)
        G = int(x.numel() // be.n)
        T = int(nIteration)
        for it in range(T):
            be.step(x, y)                               # y = (1-d) P_slab^T x over all rows (+ own restart mass), asynchronous
            ex = be.to_exchange(y)
            if it + 1 == T:
                self._all_reduce(ex)                    # last step: ranking needs every row on the seed's owner
            else:
                self._reduce_scatter(ex, G)             # a rank only reads ITS slab of x again: half an all-reduce
            self.exchanged_bytes += int(ex.numel()) * 8
            x, y = y, x                                 # Model.updateRanks
        ids, sc, cnt = be.rank(x, topN)                 # owner ranks only; -1 elsewhere
        if self.world > 1:
            import torch
            import torch.distributed as dist
            own = cnt >= 0
            t_ids = be.from_numpy(np.where(own[:, None], ids, 0))
            t_sc = be.from_numpy(np.where(own[:, None], sc, 0.0))
            t_cnt = be.from_numpy(np.where(own, cnt, 0).astype(np.int32))
            t_own = be.from_numpy(own.astype(np.int32))
            for t in (t_ids, t_sc, t_cnt, t_own):
                dist.all_reduce(t, group=self.group)    # every seed has exactly one owner: sum == gather
            assert bool((t_own == 1).all())
            ids, sc, cnt = t_ids.cpu().numpy(), t_sc.cpu().numpy(), t_cnt.cpu().numpy()
            del torch
        return ids, sc, cnt
