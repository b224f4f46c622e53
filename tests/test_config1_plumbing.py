"""BASELINE.json configs[0] ("TweetRecommender RWR on an ego-network SQLite graph, 1 seed user"): the reference's host flow
replayed end to end (tests/tweet_harness.py) over the oracle's classes and over the GPU classes with the SAME caller code;
the result.dat lines and every ranked list must be identical."""
import os

import pytest

from tests import tweet_harness as th

METHODS = list(range(16))          # every Methodology of Experiment.cs:7-15 / DataLoader.cs:144-212


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("ego") / "1000.sqlite")
    th.make_ego_db(p)
    return p


def test_loader_builds_a_valid_ego_network(db):
    api = th.oracle_api()
    ld = th.DataLoader(api, db, 1000, 3)
    assert ld.checkEgoNetworkValidation() and ld.cntLikesOfEgoUser >= 50
    ld.graphConfiguration(th.ALL, 0)
    n = len(ld.allNodes)
    assert sorted(ld.allNodes) == list(range(n)) and ld.allNodes[0].id == 1000          # ego is node 0 (DataLoader.cs:258)
    types = {l.type for ls in ld.allLinks.values() for l in ls}
    assert {api.LIKE, api.FRIENDSHIP, api.FOLLOW, api.MENTION, api.AUTHORSHIP} <= types
    assert any(l.weight != 1.0 for ls in ld.allLinks.values() for l in ls if l.type == api.MENTION)
    assert len(ld.testSet) == ld.cntLikesOfEgoUser // 3
    lines, _ = th.run_k_fold(api, db, 1000, [th.BASELINE], 3, 5)
    assert len(lines) == 1 and lines[0].split("\t")[:4] == ["1000", "0", "3", "5"]


def test_all_sixteen_methodologies_on_the_oracle(db):
    """The 16 feature sets give 16 result lines; the three relabel cases (Experiment.cs:84-101) leave no FRIENDSHIP link in
    the walk, and building before or after the relabel is the same graph for the oracle's classes too."""
    api = th.oracle_api()
    a_lines, a_lists = th.run_k_fold(api, db, 1000, METHODS, 3, 10)
    b_lines, b_lists = th.run_k_fold(api, db, 1000, list(th.RELABEL), 3, 10, relabel_after_build=True)
    assert len(a_lines) == 16 and [l.split("\t")[1] for l in a_lines] == [str(m) for m in METHODS]
    assert b_lines == [a_lines[m] for m in th.RELABEL]
    assert len({l.split("\t", 2)[2] for l in a_lines}) > 4      # the feature sets do change the outcome


@pytest.mark.gpu
def test_result_dat_identical_on_gpu_and_oracle(db):
    import recommendersystems_amd as amd
    o_lines, o_lists = th.run_k_fold(th.oracle_api(), db, 1000, METHODS, 3, 10)
    ev = lambda rec, test, T: rec.RecommendationEval(0, 0.15, T, test)
    before = amd.Graph.incremental_rebuilds
    g_lines, g_lists = th.run_k_fold(th.gpu_api(), db, 1000, METHODS, 3, 10, evaluate=ev, relabel_after_build=True)
    # the three FRIENDSHIP -> UNDEFINED methodologies were rebuilt in place (rwr_graph_update_links), once per fold
    assert amd.Graph.incremental_rebuilds - before == 3 * len(th.RELABEL)
    assert g_lines == o_lines                                  # hits and MAP identical to the last bit
    for (m, f, a), (_, _, b) in zip(o_lists, g_lists):
        assert [x[0] for x in a] == [x[0] for x in b], (m, f)
        assert [x[1].hex() for x in a] == [float(x[1]).hex() for x in b], (m, f)


@pytest.mark.gpu
def test_result_dat_identical_through_the_batch_entry_point(db):
    """All 48 graphs of the ego network (16 methodologies x 3 folds) handed to ONE rwr_eval_graphs call: the same result.dat."""
    import recommendersystems_amd as amd
    o_lines, _ = th.run_k_fold(th.oracle_api(), db, 1000, METHODS, 3, 10)
    g_lines = th.run_k_fold_batched(th.gpu_api(), db, 1000, METHODS, 3, 10, amd.EvaluateGraphs)
    assert g_lines == o_lines
