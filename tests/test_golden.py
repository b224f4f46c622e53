"""Committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py from the literal
restatement): the C oracle must reproduce them on CPU, the HIP path through the C-ABI on the GPU."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import rwr_oracle as po
from oracle.c_oracle import FlatGraph

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def load(path):
    d = json.load(open(path))
    g = dict(node_id=np.array(d["node_id"], dtype=np.int64), node_type=np.array(d["node_type"], dtype=np.uint8),
             rowptr=np.array(d["rowptr"], dtype=np.int64), dst=np.array(d["dst"], dtype=np.int32),
             etype=np.array(d["etype"], dtype=np.uint8),
             w=np.array([po.hex_f64(h) for h in d["w_hex"]], dtype=np.float64))
    return d, g


def hexes(a):
    return [po.f64_hex(float(x)) for x in a]


def test_fixtures_present():
    assert len(FILES) >= 5


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-5])
def test_c_oracle_reproduces_golden(path):
    d, g = load(path)
    F = FlatGraph(**g)
    assert hexes(F.w_norm) == d["w_norm_hex"] and F.dangling.tolist() == d["dangling"]
    for run in d["runs"]:
        r, _ = F.model_run(po.widen_float(d["damping_float"]), run["seed"], 0, run["iterations"])
        assert hexes(r) == run["rank_hex"]
        ids, sc = F.recommend(run["seed"], d["damping_float"], run["iterations"])
        assert ids.tolist() == run["rec_ids"] and hexes(sc) == run["rec_scores_hex"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-5])
def test_hip_reproduces_golden(path):
    import recommendersystems_amd as amd
    d, g = load(path)
    G = amd.Graph.from_flat(**g)
    G.buildGraph()
    wn, dg = G.normalized()
    assert hexes(wn) == d["w_norm_hex"] and dg.tolist() == d["dangling"]
    rec = amd.Recommender(G)
    for run in d["runs"]:
        m = amd.Model(G, po.widen_float(d["damping_float"]), run["seed"])
        m.run(run["iterations"])
        assert hexes(m.rank) == run["rank_hex"]
        got = rec.Recommendation(run["seed"], d["damping_float"], run["iterations"])
        assert [r[0] for r in got] == run["rec_ids"] and hexes([r[1] for r in got]) == run["rec_scores_hex"]
