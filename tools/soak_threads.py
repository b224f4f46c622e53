"""Concurrency soak: several host threads, each with its own graphs (the reference's threading model: one worker per ego
network, at most 10 at a time -- Program.cs:11,61-66), hammer the C-ABI at the same time; every result is compared with
the C restatement computed by the same thread.    python tools/soak_threads.py [seconds] [threads]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from recommendersystems_amd.rwr_based import Graph, Model, Recommender
from oracle.c_oracle import FlatGraph
from tests import graphgen as gg

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t_end = time.time() + budget
errors, counts = [], [0] * nthreads


def b(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def worker(tid):
    rng = np.random.default_rng(9000 + tid)
    try:
        while time.time() < t_end and not errors:
            nu, ni = int(rng.integers(5, 600)), int(rng.integers(5, 2000))
            g = gg.random_graph(int(rng.integers(0, 2 ** 31)), n_users=nu, n_items=ni, n_likes=int(rng.integers(10, 4 * (nu + ni))),
                                n_friend=int(rng.integers(0, nu)), n_mention=int(rng.integers(0, nu // 2 + 1)))
            F = FlatGraph(**g)
            G = Graph.from_flat(**g, seed_row_kernel=str(rng.choice(["auto", "fold", "scan"])))
            G.buildGraph()
            rec = Recommender(G)
            for _ in range(3):
                K = int(rng.choice([1, 4, 40])); T = int(rng.integers(1, 11))
                seeds = rng.integers(0, nu, K).astype(np.int32)
                ids, sc, cnt = rec.RecommendationBatch(seeds, 0.15, T, 30)
                oi, os_, oc = F.recommend_batch(seeds, 0.15, T, 30, n_threads=1)
                if not ((cnt == oc).all() and (ids == oi).all() and (b(sc) == b(os_)).all()):
                    errors.append((tid, "batch")); return
                full = rec.Recommendation(int(seeds[0]), 0.15, T)
                fi, fs = F.recommend(int(seeds[0]), 0.15, T)
                if [x[0] for x in full] != fi.tolist() or not (b([x[1] for x in full]) == b(fs)).all():
                    errors.append((tid, "full")); return
                m = Model(G, float(np.float32(0.15)), int(seeds[0])); m.run(T)
                r, _ = F.model_run(float(np.float32(0.15)), int(seeds[0]), 0, T)
                if not (b(m.rank) == b(r)).all():
                    errors.append((tid, "model")); return
            G.close()
            counts[tid] += 1
    except Exception as e:       # noqa: BLE001
        errors.append((tid, repr(e)))


ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for t in ths:
    t.start()
for t in ths:
    t.join()
if errors:
    print("MISMATCH / ERROR", errors[:4])
    sys.exit(1)
print(f"soak ok: {nthreads} threads, {sum(counts)} graphs, every result bitwise equal to the C restatement")
