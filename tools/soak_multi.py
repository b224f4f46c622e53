"""Randomised differential soak of rwr_eval_graphs (many graphs per call) against the C restatement: batches of 1..40 graphs
in the loader's shape around the limits of the one-launch paths (6144 nodes, 4096 items, 65536 links: some graphs of a batch
lie beyond them and take the single-graph calls inside), random seeds (dangling ones too), damping factors, iteration counts
and test sets.    python tools/soak_multi.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd.rwr_based import EvaluateGraphs, Graph
from oracle.c_oracle import FlatGraph, evaluate
from tests import graphgen as gg

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
base = int(os.environ.get("SOAK_SEED", "4242"))
first = int(os.environ.get("SOAK_FIRST", "0"))
runs = graphs_done = 0
t_last = time.time()


def b(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


while time.time() < t_end:
    case = first + runs
    rng = np.random.default_rng(base + case)
    nb = int(rng.choice([1, 2, 5, 12, 40]))
    d = float(rng.choice([0.15, 0.15, 0.5, 0.9]))
    T = int(rng.integers(0, 12))
    graphs, seeds, tests, params_all = [], [], [], []
    for k in range(nb):
        big = rng.random() < 0.1
        nu = int(rng.integers(2, 900 if not big else 2500))
        ni = int(rng.integers(1, 3000 if not big else 5500))
        params = dict(seed=int(rng.integers(0, 2 ** 31)), n_users=nu, n_items=ni, n_likes=int(rng.integers(0, 5 * (nu + ni))),
                      n_etc=int(rng.choice([0, 0, 7, 60])), p_undefined=float(rng.choice([0.0, 0.05, 0.4])),
                      n_friend=int(rng.integers(0, 2 * nu)), n_mention=int(rng.integers(0, nu)), n_author=int(rng.integers(0, nu)))
        if rng.random() < 0.3:
            params.update(uniform=True, n_mention=0)
        g = gg.random_graph(**params)
        graphs.append(g)
        params_all.append(params)
        seeds.append(int(rng.integers(0, nu)))
        n = len(g["node_id"])
        tests.append([int(x) for x in rng.choice(g["node_id"], size=min(n, int(rng.integers(0, 40))), replace=False)])
    Gs = [Graph.from_flat(**g) for g in graphs]
    hits, sp, ln = EvaluateGraphs(Gs, seeds, d, T, tests)
    for k, g in enumerate(graphs):
        F = FlatGraph(**g)
        fi, _ = F.recommend(seeds[k], d, T)
        eh, esp = evaluate(fi, sorted(set(tests[k])))
        if (int(hits[k]), int(ln[k])) != (eh, len(fi)) or b([sp[k]])[0] != b([esp])[0]:
            print("MISMATCH case", case, "graph", k, params_all[k], dict(seed=seeds[k], d=d, T=T), (int(hits[k]), float(sp[k]), int(ln[k])),
                  (eh, esp, len(fi)), flush=True)
            sys.exit(1)
    runs += 1
    graphs_done += nb
    if time.time() - t_last > 60:
        print(f"... {runs} batches / {graphs_done} graphs so far", flush=True)
        t_last = time.time()
print(f"soak ok: {runs} random batches, {graphs_done} graphs through rwr_eval_graphs, hits / sum of precisions / list length equal to the C "
      f"restatement's bit for bit")
