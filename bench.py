#!/usr/bin/env python3
"""RWR seeds/s + achieved HBM GB/s on the synthetic bipartite like-graphs of BASELINE.md.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS
a rank; typed as plain `python bench.py --gpus N` it starts the N ranks itself (child processes through
torch.distributed.run, rendezvous on 127.0.0.1 -- this parent never touches the GPU) and relays rank 0's JSON line.

One STEP = one pass of the hot path over one batch: Recommendation for `seeds_per_gpu`
seeds on every GPU (graph resident in HBM; T power iterations + exclusion + top-100
ranking + D2H of the K x 100 result).  Seeds are independent (a fresh Model per call,
Recommender.cs:16), so ranks shard the seed set with NO data-path collective
("scaling": "weak": seeds per GPU fixed).  value = all seeds of all ranks / max-over-ranks
wall time of the K steps.

The line also carries
  roofline      -- dominant kernel (the batched SpMM), DENSE launches only (every row walked, every entry's source row
                   gathered; the first iterations of a run skip the rows that are still exactly zero and are NOT credited
                   with the full byte count): ALGORITHMIC bytes (SURVEY.md 8d formula) / their durations measured with HIP
                   events on the library's own stream inside the timed region; roofline.k1_spmv is the same for the
                   single-seed SpMV (the kernel BASELINE.json's 70 % target names), measured after the timed region;
  cpu_baseline  -- the oracle's C restatement ("port": the reference is C# and cannot be
                   built here) timed on this host's cores on a bounded seed sample, plus a
                   bitwise comparison of that sample with the GPU result.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
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


TRAFFIC_KERNEL_SOURCES = ("iterate.hip", "build.hip", "common.h", "pf.h")


def kernel_source_sha() -> str:
    """Fingerprint of the sources a traffic measurement belongs to: the file that holds the dominant kernel (k_spmm*,
    iterate.hip), the one that lays out what it reads (build.hip) and the device-side headers they include.  profiles/traffic_*.json carries the
    one it was taken at; the GPU box has no .git, so a commit id is not available at run time."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "recommendersystems_amd", "csrc")
    for f in TRAFFIC_KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` typed as is: start the N ranks (torch.distributed.run, one per GPU) as CHILD processes
    and relay their output; this parent has not initialised the GPU and never does."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C4", help="tiny | C1 | C2 | C3 | C4 | C5 (BASELINE.md section 3; C1 = the harness's own "
                                                   "workload: ego networks x 16 methodologies x folds from 10 host threads)")
    ap.add_argument("--ego-networks", type=int, default=4, help="C1: synthetic ego networks (each gives 16 x 3 graphs)")
    ap.add_argument("--host-threads", type=int, default=10, help="C1: concurrent host threads (Program.cs:11)")
    ap.add_argument("--seeds-per-gpu", type=int, default=0)
    ap.add_argument("--mode", default="exact", choices=["exact"])
    ap.add_argument("--tile-seeds", type=int, default=0)
    ap.add_argument("--tile-group", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seeds", type=int, default=0, help="seeds in the CPU sample (0 = one per thread)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only rehearsal of the multi-rank plumbing (gloo): launcher, rank/seed-shard logic, barrier and "
                         "max-over-ranks reduction, with NO graph and NO kernels; prints a line marked dry_run (never a result)")
    ap.add_argument("--partition", default="seeds", choices=["seeds", "rows"],
                    help="seeds: graph replicated, seeds sharded, no collective (config 4, the default);  rows: transition "
                         "matrix partitioned by source rows, all-reduce of the rank matrix per iteration (config 5)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.config == "C1":
        return bench_c1(args)

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
        uni = bool(st.get("uniform_path", 0))
        # Dominant kernel: the batched SpMM.  One launch = one power-iteration step over the seeds of its tile group;
        # algorithmic bytes per SURVEY.md 8d: matrix once per launch (4-byte index + v bytes of value per entry: v = 8 on the
        # general path, v = 0 on the value-free uniform-weight path, which adds the n*8-byte per-source scale vector), row
        # offsets, dangling flags, read X once + write Y once (gathers counted as n*K*8, never nnz*K*8).
        # Only the DENSE launches are priced: the frontier launches of the first iterations skip most rows.
        m_bytes = matrix_bytes(n, nnz, uni)
        d_launches, d_steps, d_ms = st["spmm_dense_launches"], st["spmm_dense_seed_steps"], st["spmm_dense_ms"]
        bytes_dense = d_launches * m_bytes + d_steps * (16 * n + 12)
        achieved = bytes_dense / (d_ms / 1e3) / 1e9 if d_ms > 0 else 0.0
        # HBM-side bytes per DENSE SpMM launch from the committed rocprofv3 PMC passes of this same command (counter passes
        # cannot run inside the timed run).  The file carries the fingerprint of the kernel sources it was measured at:
        # null when the sources have changed since, or when no measurement exists for this configuration.
        traffic, traffic_note = None, "no PMC measurement committed for this configuration"
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("csrc_sha") != kernel_source_sha():
                traffic_note = f"stale: measured at kernel sources {tj.get('csrc_sha')}, running {kernel_source_sha()}"
            elif tj.get("mode") == args.mode and tj.get("tile_seeds") == st["tile_seeds"] and K == K_cfg:
                traffic, traffic_note = tj["bytes_per_launch"], "from the committed PMC file, not measured in this run: " + tj.get("source", "")
            else:
                traffic_note = "measured for another mode / tile width / batch size"
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
                       "matrix_path": "value-free (uniform row weights)" if uni else "weighted",
                       "parallelism": f"seed-sharded x{world} (graph replicated, no collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "k_spmm (dense launches: every row walked, no frontier skipping)",
                         "launches": d_launches, "avg_launch_ms": d_ms / max(d_launches, 1),
                         "algorithmic_bytes_per_launch": bytes_dense / max(d_launches, 1),
                         "all_launches": {"launches": st["spmm_launches"], "avg_launch_ms": st["spmm_ms"] / max(st["spmm_launches"], 1),
                                          "note": "includes the frontier launches of the first iterations (rows still exactly "
                                                  "zero are skipped); not priced against the roofline"}},
            "phases_ms_per_step": {"spmm": st["spmm_ms"] / args.steps, "seed_row": st["chain_ms"] / args.steps,
                                   "iterate_span": st["iterate_wall_ms"] / args.steps,
                                   "rank": st["rank_ms"] / args.steps,
                                   "call_wall": st["total_wall_ms"] / args.steps},
            "graph_build_ms": st["build_ms"], "graph_create_s": t_create,
        }
        # the second roofline, beside the first and never instead of it: what the launch actually moves across the L2 <-> fabric
        # boundary (PMC, from the committed file -- see traffic_note) over its duration, against the rate the memory system
        # sustains for random row gathers (MI355X_MICROARCH.md: 7.4-7.9 TB/s for a table beyond the L2s).  frac ~ 1 with
        # traffic >> algorithmic bytes reads: the rate is spent, only bytes are left to save.
        if traffic is not None and d_launches > 0 and d_ms > 0:
            fab = traffic / (d_ms / d_launches / 1e3) / 1e9
            out["roofline"]["fabric"] = {"bound": "L2<->fabric random-row gathers", "achieved": fab, "ceiling": 7650.0,
                                         "ceiling_range": [7400.0, 7900.0], "unit": "GB/s", "frac": fab / 7650.0,
                                         "traffic_over_algorithmic": traffic / (bytes_dense / d_launches),
                                         "note": "traffic = PMC bytes per dense launch from the committed profile (fingerprint-guarded), "
                                                 "not measured in this run; duration = this run's HIP events"}
        single = None
        if world == 1:
            # the unmodified harness's call shape (Experiment.cs:109): ONE seed per call.  Outside the timed region;
            # reported beside the batch figure because it is what a drop-in user of the C# host sees per call -- and its
            # SpMV is the kernel BASELINE.json's 70 % target names (roofline.k1_spmv).
            s0 = int(seeds[len(seeds) // 2])
            rec.Recommendation(s0, DAMPING, T_ITER, TOP_N)
            G.reset_stats()
            reps = 5
            t1 = time.perf_counter()
            for _ in range(reps):
                single = rec.Recommendation(s0, DAMPING, T_ITER, TOP_N)
            t_call = (time.perf_counter() - t1) / reps
            s1 = G.stats()
            out["single_seed_call"] = {"ms": 1e3 * t_call, "seed": s0, "top_n": TOP_N, "iterations": T_ITER, "mode": args.mode}
            k1_launches, k1_ms = s1["spmm_dense_launches"], s1["spmm_dense_ms"]
            k1_bytes = matrix_bytes(n, nnz, uni) + 16 * n + 12
            k1_ach = k1_launches * k1_bytes / (k1_ms / 1e3) / 1e9 if k1_ms > 0 else 0.0
            out["roofline"]["k1_spmv"] = {
                "kernel": "k_spmv (single seed, dense steps)", "bound": "hbm", "achieved": k1_ach, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": k1_ach / HBM_PEAK_GBPS, "launches": k1_launches,
                "avg_launch_ms": k1_ms / max(k1_launches, 1), "algorithmic_bytes_per_launch": k1_bytes}
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(flat, seeds, ids, sc, cnt, args, single_seed=(s0, single))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_c1(args):
    """BASELINE.json configs[0] the way the reference's host runs it (Program.cs:11,61-66; Experiment.cs:46,69,104-128): for
    every ego network, 16 methodologies x 3 folds = 48 slightly different graphs, each built (Graph ctor + buildGraph),
    walked from seed 0 (Recommendation(0, 0.15f, T)) and evaluated (Hits / AP over the full list), from 10 concurrent host
    threads.  `value`: every thread hands its share of the graphs to one rwr_eval_graphs call; `one_call_per_graph`: the same
    threads with a handle per graph (create + recommend_eval + destroy).  The graphs are produced beforehand by the restated loader (tests/tweet_harness.py, the
    reference's DataLoader / SQLiteAdapter flow over synthetic <ego>.sqlite files) and flattened; the timed region is
    rwr_graph_create + rwr_recommend_eval + rwr_graph_destroy per graph through the C-ABI.  One STEP = one pass over all
    graphs.  cpu_baseline = the same loop over the C restatement of the reference (oracle/), same thread count; every
    (hits, sum of precisions) pair is compared bit for bit."""
    import tempfile
    import threading
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    from recommendersystems_amd.rwr_based import Graph, Recommender
    from tests import tweet_harness as th
    api = th.gpu_api()
    tmp = tempfile.mkdtemp(prefix="c1_")
    sizes = [(55, 120, 900), (80, 250, 2500), (110, 400, 5000), (60, 150, 1500), (130, 500, 8000), (70, 200, 2000)]
    graphs = []                     # (flat arrays, test set, ego, methodology, fold)
    t0 = time.time()
    for e in range(args.ego_networks):
        nf, nt, ntw = sizes[e % len(sizes)]
        ego = 1000 + 100000 * e
        path = os.path.join(tmp, f"{ego}.sqlite")
        th.make_ego_db(path, ego=ego, seed=7 + e, n_friends=nf, n_thirdparty=nt, n_tweets=ntw)
        for m in range(16):
            for fold in range(3):
                ld = th.DataLoader(api, path, ego, 3)
                if not ld.checkEgoNetworkValidation():
                    raise SystemExit("synthetic ego network failed the 50 likes / 50 friends validation")
                ld.graphConfiguration(m, fold)
                if m in th.RELABEL:                                  # Experiment.cs:84-101
                    for ls in ld.allLinks.values():
                        for l in ls:
                            if l.type == api.FRIENDSHIP:
                                l.type = api.UNDEFINED
                flat = Graph(ld.allNodes, ld.allLinks)._flatten()
                graphs.append((flat, np.ascontiguousarray(sorted(ld.testSet), dtype=np.int64), ego, m, fold))
    nodes = [int(f[0][0].shape[0]) for f in graphs]
    links = [int(f[0][3].shape[0]) for f in graphs]
    log(0, f"C1: {args.ego_networks} ego networks -> {len(graphs)} graphs ({min(nodes)}..{max(nodes)} nodes, "
           f"{min(links)}..{max(links)} links) loaded in {time.time() - t0:.1f}s")
    names = ("node_id", "node_type", "rowptr", "dst", "etype", "w")

    from concurrent.futures import ThreadPoolExecutor
    from recommendersystems_amd.rwr_based import EvaluateGraphs
    flat_graphs = [Graph.from_flat(**dict(zip(names, f[0]))) for f in graphs]     # (host arrays only: nothing is built here)
    pool = ThreadPoolExecutor(max_workers=args.host_threads)     # the harness's threads live for the whole run (Program.cs:11)

    def run_on_threads(worker):
        for f in [pool.submit(worker, t) for t in range(args.host_threads)]:
            f.result()

    def gpu_pass(results):
        # a handle per graph: create + recommend_eval + destroy, graphs dealt to the threads as they become free
        nxt = iter(range(len(graphs)))
        lock = threading.Lock()

        def worker(_t):
            while True:
                with lock:
                    i = next(nxt, None)
                if i is None:
                    return
                flat, test, _, _, _ = graphs[i]
                G = Graph.from_flat(**dict(zip(names, flat)))
                G.buildGraph()
                results[i] = Recommender(G).RecommendationEval(0, DAMPING, T_ITER, test)
                G.close()
        run_on_threads(worker)

    def gpu_pass_batched(results):
        # the same threads, each handing its share of the graphs to ONE rwr_eval_graphs call (a workgroup per graph)
        nt = args.host_threads
        per = (len(graphs) + nt - 1) // nt

        def worker(t):
            lo, hi = t * per, min(len(graphs), (t + 1) * per)
            if lo >= hi:
                return
            h, sp, ln = EvaluateGraphs(flat_graphs[lo:hi], [0] * (hi - lo), DAMPING, T_ITER, [graphs[i][1] for i in range(lo, hi)])
            for i in range(lo, hi):
                results[i] = (int(h[i - lo]), float(sp[i - lo]), int(ln[i - lo]))
        run_on_threads(worker)

    def timed(fn, results):
        for _ in range(args.warmup):
            fn(results)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn(results)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    res_single = [None] * len(graphs)
    elapsed_single = timed(gpu_pass, res_single)
    res = [None] * len(graphs)
    elapsed = timed(gpu_pass_batched, res)
    value = len(graphs) * args.steps / elapsed
    out = {"metric": "ego-network evaluations/s (Graph.buildGraph + Recommendation(0, 0.15f, T) + Hits/AP per graph)", "value": value,
           "unit": "graphs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"C1: {args.ego_networks} synthetic ego networks x 16 methodologies x 3 folds = {len(graphs)} graphs "
                                  f"({min(nodes)}..{max(nodes)} nodes, {min(links)}..{max(links)} links), seed 0, T={T_ITER}, d=0.15f, "
                                  f"full list evaluated, {args.host_threads} host threads (Program.cs:11)",
                      "graphs": len(graphs), "host_threads": args.host_threads, "iterations": T_ITER},
           "ms_per_graph": 1e3 * elapsed / (args.steps * len(graphs)),
           "entry_point": "rwr_eval_graphs: every host thread hands its share of the graphs to one call (one build launch, one "
                          "iteration launch, one evaluation launch per call, a workgroup per graph)",
           "one_call_per_graph": {"value": len(graphs) * args.steps / elapsed_single, "unit": "graphs/s",
                                  "entry_points": "rwr_graph_create + rwr_recommend_eval + rwr_graph_destroy per graph, same threads",
                                  "results_identical_to_batched": res_single == res},
           "roofline": None, "roofline_note": "latency-bound by design: every graph is one workgroup of the build kernel and one "
                                              "of the iteration kernel; see DESIGN.md 3.6"}
    # the same entry point from NATIVE host threads (the reference's host is compiled code: Program.cs:11): Python threads
    # serialise on the interpreter between calls, so the rate above understates what a C# / C++ host gets.  Built and run as a
    # child process (tools/eval_graphs_threads.cpp, its own synthetic graphs of the same size range); absent g++ = skipped.
    try:
        exe = os.path.join(tmp, "egt")
        pkg = os.path.join(ROOT, "recommendersystems_amd")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "eval_graphs_threads.cpp"), "-o", exe, os.path.join(pkg, "librwr.so"),
                               "-Wl,-rpath," + pkg], stderr=subprocess.DEVNULL)
        per = (len(graphs) + args.host_threads - 1) // args.host_threads
        txt = subprocess.run([exe, str(per), "30"], capture_output=True, text=True, timeout=120).stdout
        out["native_host_threads"] = {"note": f"tools/eval_graphs_threads.cpp: T std::threads x {per} graphs per rwr_eval_graphs call x 30 calls "
                                              "(600..2100-item synthetic graphs built in C++), graphs/s",
                                      "by_threads": [json.loads(l) for l in txt.splitlines() if l.startswith("{")]}
    except Exception as e:                                   # noqa: BLE001
        out["native_host_threads"] = {"skipped": repr(e)[:200]}
    if not args.no_cpu_baseline:
        from oracle.c_oracle import FlatGraph, evaluate
        cres = [None] * len(graphs)

        def cpu_pass():
            nxt = iter(range(len(graphs)))
            lock = threading.Lock()

            def worker(_t):
                while True:
                    with lock:
                        i = next(nxt, None)
                    if i is None:
                        return
                    flat, test, _, _, _ = graphs[i]
                    F = FlatGraph(*flat)
                    ids, _ = F.recommend(0, DAMPING, T_ITER)
                    cres[i] = evaluate(ids, test) + (len(ids),)
            run_on_threads(worker)
        t1 = time.perf_counter()
        cpu_pass()
        t_cpu = time.perf_counter() - t1
        same = all(int(a[0]) == int(b[0]) and np.float64(a[1]).view(np.uint64) == np.float64(b[1]).view(np.uint64)
                   and int(a[2]) == int(b[2]) for a, b in zip(res, cres))
        # result.dat lines (Experiment.cs:144-153): per (ego, methodology) hits and MAP over the folds, both sides
        def lines(rs):
            acc = {}
            for (flat, test, ego, m, fold), r in zip(graphs, rs):
                h, ap = acc.get((ego, m), (0, 0.0))
                acc[(ego, m)] = (h + int(r[0]), ap + (0.0 if int(r[0]) == 0 else float(r[1]) / int(r[0])))
            return [f"{ego}\t{m}\t3\t{T_ITER}\t{h}\t{repr(ap / 3)}" for (ego, m), (h, ap) in sorted(acc.items())]
        out["cpu_baseline"] = {"value": len(graphs) / t_cpu, "unit": "graphs/s", "cores": min(args.host_threads, os.cpu_count() or 1),
                               "kind": "port", "sample": f"one pass over the same {len(graphs)} graphs, {args.host_threads} host threads "
                                                          f"(build + Recommendation + evaluation in oracle/rwr_oracle.c) in {t_cpu:.1f}s",
                               "hits_and_precision_sums_bitwise_equal": bool(same),
                               "result_dat_lines_identical": lines(res) == lines(cres), "result_dat_lines": len(lines(res))}
    print(json.dumps(out), flush=True)


def matrix_bytes(n: int, nnz: int, uniform_path: bool) -> int:
    """Matrix-side algorithmic bytes of one SpMM / SpMV launch: 4-byte index + v-byte value per entry
    (v = 0 on the value-free path, which reads the n*8-byte per-source scale vector instead), row offsets, dangling flags."""
    return nnz * (4 + (0 if uniform_path else 8)) + (n + 1) * 8 + n + (8 * n if uniform_path else 0)


def dry_run(args, rank, world):
    """CPU rehearsal of the multi-rank plumbing over gloo: rank / seed-shard logic, barrier, max-over-ranks timing,
    rank 0's single JSON line.  No graph, no kernels, no result: the line is marked dry_run."""
    import torch
    import torch.distributed as dist
    from recommendersystems_amd import synth
    if world > 1:
        dist.init_process_group("gloo")
    no, U, I, E, K_cfg = synth.CONFIGS[args.config]
    K = args.seeds_per_gpu or K_cfg
    seeds = synth.seeds_for(U, K * world, rank * K, K)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0, float(seeds[0]), float(seeds[-1])], dtype=torch.float64)
    firsts = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    if world > 1:
        dist.all_gather(firsts, t)
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm[0])
    else:
        firsts, elapsed = [t], float(t[0])
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "seeds_per_gpu": K, "elapsed_max_s": elapsed,
                          "shards": [[int(f[1]), int(f[2])] for f in firsts]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_row_partitioned(args, rank, local_rank, world, flat, U, I, g, n, nnz):
    """Config 5 shape: every rank owns a slab of source rows; one step = one batch of K seeds for the WHOLE job; per
    iteration one reduce-scatter(sum) of the partial n x G rank matrix by slabs over RCCL (a rank only reads its own slab
    again; the last iteration all-reduces, ranking needs every row).  "scaling": "strong" (the batch is fixed, the matrix
    is split).  Tolerance parity (partial sums re-associate)."""
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
                          "parallelism": f"source-row slabs x{world}, reduce-scatter of the {n}x{G} partial rank matrix per "
                                         f"iteration (all-reduce on the last one)",
                          "exchange_payload_bytes_per_iteration": n * G * 8,
                          "exchange_note": "reduce-scatter: each rank sends (world-1)/world of the payload once; the "
                                           "round-1 all-reduce moved twice that, plus a second collective for the restart scalars"},
               "roofline": {"bound": "hbm", "achieved": T_ITER * step_bytes * args.steps / elapsed / 1e9,
                            "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                            "frac": T_ITER * step_bytes * args.steps / elapsed / 1e9 / (HBM_PEAK_GBPS * world),
                            "traffic": None, "kernel": "whole step (local SpMM + all-reduce + ranking), wall clock"},
               "graph_create_s": t_create}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(flat, seeds, ids, sc, cnt, args, single_seed=None):
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
    out = {"value": ks / dt, "unit": "seeds/s", "cores": cores, "kind": "port",
           "sample": f"{ks} seeds of the same batch (one per thread, T={T_ITER}, top_n={TOP_N}) in {dt:.1f}s",
           "gpu_topk_ids_identical": same_ids, "gpu_scores_bitwise_equal": same_bits,
           "gpu_max_abs_score_diff": float(np.abs(os_ - sc[:ks]).max()) if ks else 0.0}
    if single_seed is not None and single_seed[1] is not None:
        # the single-seed call (its own kernels: SpMV, binade scan) against the same restatement
        s0, got = single_seed
        ri, rs = F.recommend(s0, DAMPING, T_ITER, TOP_N)
        gi = np.array([r[0] for r in got], dtype=np.int64)
        gs = np.array([r[1] for r in got], dtype=np.float64)
        out["single_seed_ids_identical"] = bool(gi.shape == ri.shape and (gi == ri).all())
        out["single_seed_scores_bitwise_equal"] = bool(gs.shape == rs.shape and (gs.view(np.uint64) == rs.view(np.uint64)).all())
    out["faithful_dense_restart"] = faithful_cost(cores)
    return out


def faithful_cost(cores: int):
    """The reference's own loop structure is O(nnz + n^2) per iteration: for EVERY source node a dense `for r in 0..n`
    restart loop although restart[] is one-hot (Model.cs:92-93,96-97).  The port above skips the +0.0 addends (bitwise
    identical).  Timed here on a 10^4-node graph, one core, both forms, results compared bitwise: the factor the reference
    itself would pay (it grows with n)."""
    from oracle.c_oracle import FlatGraph
    from recommendersystems_amd import synth
    g = synth.bipartite(9, 2_000, 8_000, 60_000)
    F = FlatGraph(**{k: g[k] for k in ("node_id", "node_type", "rowptr", "dst", "etype", "w")})
    d = float(np.float32(DAMPING))
    t0 = time.perf_counter()
    r_sparse, _ = F.model_run(d, 0, 0, T_ITER)
    t_sparse = time.perf_counter() - t0
    t0 = time.perf_counter()
    r_dense, _ = F.model_run(d, 0, 0, T_ITER, dense=True)
    t_dense = time.perf_counter() - t0
    return {"graph": "synthetic 2000 users x 8000 items, 60000 likes (n = 10^4)", "iterations": T_ITER, "cores": 1,
            "sparse_restart_ms": 1e3 * t_sparse, "reference_loop_ms": 1e3 * t_dense, "factor": t_dense / max(t_sparse, 1e-9),
            "bitwise_equal": bool((r_sparse.view(np.uint64) == r_dense.view(np.uint64)).all())}


if __name__ == "__main__":
    main()
