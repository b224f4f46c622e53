"""Two processes sharing the one GPU of the test box (gloo for the process group; RCCL needs one GPU per rank): the
real HIP backends under the multi-process host logic -- seed-sharded batch + result gather, and the row-partitioned
iteration with its per-step reduce-scatter (all-reduce on the last step)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import recommendersystems_amd as amd
        from recommendersystems_amd import partitioned as pt
        from oracle.c_oracle import FlatGraph
        from tests import graphgen as gg
        torch.cuda.set_device(0)
        g = gg.random_graph(77, n_users=600, n_items=2500, n_likes=20000, n_friend=400, n_mention=300, n_author=200)
        F = FlatGraph(**g)
        # (1) seed sharding: graph replicated, each rank computes its block on the GPU, results gathered
        all_seeds = np.arange(0, 600, 7, dtype=np.int32)
        lo, hi = pt.shard_bounds(len(all_seeds), world, rank)
        G = amd.Graph.from_flat(**g)
        G.buildGraph()
        li, ls, lc = amd.Recommender(G).RecommendationBatch(all_seeds[lo:hi], 0.15, 10, 15)
        fi, fs, fc = pt.gather_seed_shards(li, ls, lc, len(all_seeds))
        oi, os_, oc = F.recommend_batch(all_seeds, 0.15, 10, 15)
        ok1 = bool((fi == oi).all() and (fs.view(np.uint64) == os_.view(np.uint64)).all() and (fc == oc).all())
        G.close()
        # (2) row partition: HIP slab backend, all-reduce of the rank matrix through the process group each step
        seeds = np.array([0, 11, 300, 599], dtype=np.int32)

        pr = pt.PartitionedRecommender(g, rank=rank, world=world)

        def via_host(t):                             # gloo reduces host tensors; RCCL takes the CUDA tensor as is
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        pr._all_reduce = via_host

        def rs_via_host(t, G):                       # reduce-scatter by slabs; everything outside the own slab poisoned
            h = t.cpu()
            pt.reduce_scatter_slabs(h, G, pr.bounds, rank)
            keep = h[pr.lo * G:pr.hi * G].clone()
            h.fill_(float("nan"))
            h[pr.lo * G:pr.hi * G] = keep
            t.copy_(h)
        pr._reduce_scatter = rs_via_host
        ids, sc, cnt = pr.RecommendationBatch(seeds, 0.15, 10, 12)
        pi, ps, pc = F.recommend_batch(seeds, 0.15, 10, 12)
        ok2 = bool((ids == pi).all() and (cnt == pc).all() and np.abs(sc - ps).max() <= 1e-9)
        q.put((rank, ok1, ok2))
    finally:
        dist.destroy_process_group()


def test_two_processes_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for rank, ok1, ok2 in sorted(q.get(timeout=10) for _ in range(world)):
        assert ok1, (rank, "seed-sharded gather differs from the oracle")
        assert ok2, (rank, "row-partitioned result differs from the oracle")
