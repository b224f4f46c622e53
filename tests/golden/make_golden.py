#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the LITERAL Python restatement (oracle/rwr_oracle.py, dense
O(n^2) restart loop, i.e. the reference's code path statement by statement).

The reference ships no fixtures and its C# cannot run here, so these vectors pin
the oracle <-> HIP path, not the oracle <-> C# relation (that one rests on the hand-derived KATs and
on the two restatements agreeing).  Doubles are stored as big-endian hex so nothing is lost in JSON.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import rwr_oracle as po          # noqa: E402
from tests import graphgen as gg             # noqa: E402

CASES = {
    "kat1": (gg.kat1(), [0], 0.5),
    "kat2": (gg.kat2(), [0], 0.5),
    "mixed_small": (gg.random_graph(101, n_users=15, n_items=40, n_likes=120, n_etc=3, n_friend=12,
                                    n_mention=15, n_author=9), [0, 7, 14], 0.15),
    "undefined_heavy": (gg.random_graph(102, n_users=8, n_items=50, n_likes=70, n_etc=2, p_undefined=0.35,
                                        n_mention=8), [0, 3], 0.15),
    "uniform_likes": (gg.random_graph(103, n_users=25, n_items=70, n_likes=350, uniform=True), [0, 11, 24], 0.15),
}


def main():
    for name, (g, seeds, d) in CASES.items():
        nodes, edges = po.from_flat(g["node_id"], g["node_type"], g["rowptr"], g["dst"], g["etype"], g["w"])
        G = po.Graph(nodes, edges)
        G.buildGraph()
        out = {
            "name": name, "damping_float": d,
            "node_id": g["node_id"].tolist(), "node_type": g["node_type"].tolist(),
            "rowptr": g["rowptr"].tolist(), "dst": g["dst"].tolist(), "etype": g["etype"].tolist(),
            "w_hex": [po.f64_hex(float(x)) for x in g["w"]],
            "w_norm_hex": [], "dangling": [], "runs": [],
        }
        for i in range(len(nodes)):
            links = G.graph[i]
            out["dangling"].append(1 if links is None else 0)
            it = iter(links or [])
            for e in range(int(g["rowptr"][i]), int(g["rowptr"][i + 1])):
                out["w_norm_hex"].append(po.f64_hex(next(it).weight) if g["etype"][e] != 0 else po.f64_hex(0.0))
        for seed in seeds:
            for T in (1, 3, 10):
                m = po.Model(G, po.widen_float(d), seed, dense_restart=True)
                m.run(T)
                rec = po.Recommender(G).Recommendation(seed, d, T)
                out["runs"].append({"seed": seed, "iterations": T,
                                    "rank_hex": [po.f64_hex(x) for x in m.rank],
                                    "rec_ids": [r[0] for r in rec],
                                    "rec_scores_hex": [po.f64_hex(r[1]) for r in rec]})
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print(name, len(nodes), "nodes", len(out["dst"]), "links", len(out["runs"]), "runs")


if __name__ == "__main__":
    main()
