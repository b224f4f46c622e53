#!/usr/bin/env python3
"""RWR seeds/s + achieved HBM GB/s on the synthetic bipartite like-graphs of BASELINE.md.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One STEP = one pass of the hot path over one batch: Recommendation for `seeds_per_gpu`
seeds on every GPU (graph resident in HBM; T power iterations + exclusion + top-100
ranking + D2H of the K x 100 result).  Seeds are independent (a fresh Model per call,
Recommender.cs:16), so ranks shard the seed set with NO data-path collective
("scaling": "weak": seeds per GPU fixed).  value = all seeds of all ranks / max-over-ranks
wall time of the K steps.

The line also carries
  roofline      -- dominant kernel (the batched SpMM): ALGORITHMIC bytes (SURVEY.md 8d
                   formula) / its launch durations measured with HIP events on the library's
                   own stream inside the timed region;
  cpu_baseline  -- the oracle's C restatement ("port": the reference is C# and cannot be
                   built here) timed on this host's cores on a bounded seed sample, plus a
                   bitwise comparison of that sample with the GPU result.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
T_ITER = 10                   # the reference has no default (Program.cs:33); fixed and reported
TOP_N = 100
DAMPING = 0.15                # Experiment.cs:109


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4", help="tiny | C2 | C3 | C4 | C5 (BASELINE.md section 3)")
    ap.add_argument("--seeds-per-gpu", type=int, default=0)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"])
    ap.add_argument("--tile-seeds", type=int, default=0)
    ap.add_argument("--tile-group", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seeds", type=int, default=0, help="seeds in the CPU sample (0 = one per thread)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline")
    ap.add_argument("--partition", default="seeds", choices=["seeds", "rows"],
                    help="seeds: graph replicated, seeds sharded, no collective (config 4, the default);  rows: transition "
                         "matrix partitioned by source rows, all-reduce of the rank matrix per iteration (config 5)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from recommendersystems_amd import synth
    from recommendersystems_amd.rwr_based import Graph, Recommender

    t0 = time.time()
    no, U, I, E, K_cfg = synth.CONFIGS[args.config]
    g = synth.bipartite(no, U, I, E)
    K = args.seeds_per_gpu or K_cfg
    n = U + I
    nnz = int(g["rowptr"][-1])
    log(rank, f"config {args.config}: users {U} items {I} likes {g['likes']} (requested {E}) n {n} nnz {nnz} "
              f"generated in {time.time() - t0:.1f}s")
    flat = {k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")}

    if args.partition == "rows":
        return bench_row_partitioned(args, rank, local_rank, world, flat, U, I, g, n, nnz)

    t0 = time.time()
    G = Graph.from_flat(**flat, mode=args.mode, device=local_rank, tile_seeds=args.tile_seeds,
                        tile_group=args.tile_group, profile=True)
    G.buildGraph()
    rec = Recommender(G)
    t_create = time.time() - t0
    torch.cuda.empty_cache()          # the generator's sort scratch: give it back before the library sizes its workspace
    seeds = synth.seeds_for(U, K * world, rank * K, K)
    log(rank, f"graph resident on GPU in {t_create:.1f}s (device build {G.stats()['build_ms']:.0f} ms); "
              f"{K} seeds per GPU, T={T_ITER}, top_n={TOP_N}, mode={args.mode}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ids, sc, cnt = rec.RecommendationBatch(seeds, DAMPING, T_ITER, TOP_N)
    G.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, sc, cnt = rec.RecommendationBatch(seeds, DAMPING, T_ITER, TOP_N)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = G.stats()

    if rank == 0:
        seeds_total = K * world * args.steps
        value = seeds_total / elapsed
        # dominant kernel: batched SpMM.  One launch = one power-iteration step over the seeds of
        # its tile group; algorithmic bytes per SURVEY.md 8d (matrix once per launch, v = 8 bytes
        # of value per entry are stored and read, gathers counted as n*K*8, not nnz*K*8)
        launches = st["spmm_launches"]
        bytes_total = launches * (nnz * 12 + (n + 1) * 8 + n) + st["spmm_seed_steps"] * (16 * n + 12)
        spmm_s = st["spmm_ms"] / 1e3
        achieved = bytes_total / spmm_s / 1e9 if spmm_s > 0 else 0.0
        # HBM-side bytes per SpMM launch from the committed rocprofv3 PMC passes of this same command (the PMC passes
        # cannot run inside the timed run); null when no measurement exists for this configuration
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("mode") == args.mode and tj.get("tile_seeds") == st["tile_seeds"] and K == K_cfg:
                traffic = tj["bytes_per_launch"]
        out = {
            "metric": "RWR seeds/sec + achieved HBM GB/s on 100M-edge bipartite graph, 1/2/4/8 GPUs",
            "value": value, "unit": "seeds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: synthetic bipartite {U} users x {I} items, "
                                   f"{g['likes']} likes (nnz {nnz}), {K} seeds/GPU, T={T_ITER}, "
                                   f"d=0.15f, top_n={TOP_N}, mode={args.mode}",
                       "users": U, "items": I, "likes": int(g["likes"]), "nnz": nnz, "seeds_per_gpu": K,
                       "iterations": T_ITER, "top_n": TOP_N, "mode": args.mode,
                       "tile_seeds": st["tile_seeds"], "tile_group": st["tile_group"],
                       "parallelism": f"seed-sharded x{world} (graph replicated, no collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "k_spmm", "launches": launches,
                         "avg_launch_ms": st["spmm_ms"] / max(launches, 1),
                         "algorithmic_bytes_per_launch": bytes_total / max(launches, 1)},
            "phases_ms_per_step": {"spmm": st["spmm_ms"] / args.steps, "seed_row": st["chain_ms"] / args.steps,
                                   "iterate_span": st["iterate_wall_ms"] / args.steps,
                                   "rank": st["rank_ms"] / args.steps,
                                   "call_wall": st["total_wall_ms"] / args.steps},
            "graph_build_ms": st["build_ms"], "graph_create_s": t_create,
        }
        if world == 1:
            # the unmodified harness's call shape (Experiment.cs:109): ONE seed per call.  Outside the timed region;
            # reported beside the batch figure because it is what a drop-in user of the C# host sees per call.
            s0 = int(seeds[len(seeds) // 2])
            rec.Recommendation(s0, DAMPING, T_ITER, TOP_N)
            t1 = time.perf_counter()
            rec.Recommendation(s0, DAMPING, T_ITER, TOP_N)
            out["single_seed_call"] = {"ms": 1e3 * (time.perf_counter() - t1), "seed": s0, "top_n": TOP_N,
                                       "iterations": T_ITER, "mode": args.mode}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(flat, seeds, ids, sc, cnt, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_row_partitioned(args, rank, local_rank, world, flat, U, I, g, n, nnz):
    """Config 5 shape: every rank owns a slab of source rows; one step = one batch of K (<= 64) seeds for the WHOLE job;
    per iteration one all-reduce(sum) of the n x G rank matrix over RCCL.  "scaling": "strong" (the batch is fixed,
    the matrix is split).  Tolerance parity (partial sums re-associate)."""
    import torch
    import torch.distributed as dist
    from recommendersystems_amd import synth
    from recommendersystems_amd.partitioned import PartitionedRecommender
    K = args.seeds_per_gpu or 8
    t0 = time.time()
    pr = PartitionedRecommender(flat, rank=rank, world=world, device=local_rank)
    t_create = time.time() - t0
    seeds = synth.seeds_for(U, K, 0, K)
    for _ in range(args.warmup):
        ids, sc, cnt = pr.RecommendationBatch(seeds, DAMPING, T_ITER, TOP_N)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids, sc, cnt = pr.RecommendationBatch(seeds, DAMPING, T_ITER, TOP_N)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        G = 1
        while G < K:
            G <<= 1
        step_bytes = synth.algorithmic_bytes_per_step(n, nnz, K, 8)
        out = {"metric": "RWR seeds/sec + achieved HBM GB/s on 100M-edge bipartite graph, 1/2/4/8 GPUs",
               "value": K * args.steps / elapsed, "unit": "seeds/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{args.config} row-partitioned x{world}: {U} users x {I} items, {g['likes']} likes, "
                                      f"{K} seeds for the whole job, T={T_ITER}, top_n={TOP_N}",
                          "parallelism": f"source-row slabs x{world}, all-reduce of the {n}x{G} rank matrix per iteration",
                          "exchange_bytes_per_iteration": n * G * 8},
               "roofline": {"bound": "hbm", "achieved": T_ITER * step_bytes * args.steps / elapsed / 1e9,
                            "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                            "frac": T_ITER * step_bytes * args.steps / elapsed / 1e9 / (HBM_PEAK_GBPS * world),
                            "traffic": None, "kernel": "whole step (local SpMM + all-reduce + ranking), wall clock"},
               "graph_create_s": t_create}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(flat, seeds, ids, sc, cnt, args):
    """The reference's algorithm (push-style, one seed per thread, sequential per seed) as restated
    in oracle/rwr_oracle.c, timed on this host on a bounded sample of the same workload, and used
    to check the GPU result of that sample bitwise."""
    from oracle.c_oracle import FlatGraph, max_threads
    # the GPU box gives one GPU's job a share of 16 host cores (more threads only oversubscribe them)
    cores = min(max_threads(), len(os.sched_getaffinity(0)), args.cpu_threads)
    ks = args.cpu_seeds or 3 * cores          # ~10-30 s of CPU work on the 100 M-like graph
    ks = min(ks, len(seeds))
    F = FlatGraph(**flat)
    sample = np.ascontiguousarray(seeds[:ks])
    t0 = time.perf_counter()
    oi, os_, oc = F.recommend_batch(sample, DAMPING, T_ITER, TOP_N, n_threads=cores)
    dt = time.perf_counter() - t0
    same_ids = bool((oi == ids[:ks]).all() and (oc == cnt[:ks]).all())
    same_bits = bool((os_.view(np.uint64) == sc[:ks].view(np.uint64)).all())
    return {"value": ks / dt, "unit": "seeds/s", "cores": cores, "kind": "port",
            "sample": f"{ks} seeds of the same batch (one per thread, T={T_ITER}, top_n={TOP_N}) in {dt:.1f}s",
            "gpu_topk_ids_identical": same_ids, "gpu_scores_bitwise_equal": same_bits,
            "gpu_max_abs_score_diff": float(np.abs(os_ - sc[:ks]).max()) if ks else 0.0}


if __name__ == "__main__":
    main()
