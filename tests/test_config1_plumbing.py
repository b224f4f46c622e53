"""BASELINE.json configs[0] ("TweetRecommender RWR on an ego-network SQLite graph, 1 seed user"): the reference's host flow
replayed end to end (tests/tweet_harness.py) over the oracle's classes and over the GPU classes with the SAME caller code;
the result.dat lines and every ranked list must be identical."""
import os

import pytest

from tests import tweet_harness as th

METHODS = [th.BASELINE, th.INCL_FRIENDSHIP, th.INCL_MENTIONCOUNT, th.ALL, th.EXCL_FRIENDSHIP,
           th.INCL_FOLLOWSHIP_ON_THIRDPARTY_AND_AUTHORSHIP]


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


@pytest.mark.gpu
def test_result_dat_identical_on_gpu_and_oracle(db):
    o_lines, o_lists = th.run_k_fold(th.oracle_api(), db, 1000, METHODS, 3, 10)
    ev = lambda rec, test, T: rec.RecommendationEval(0, 0.15, T, test)
    g_lines, g_lists = th.run_k_fold(th.gpu_api(), db, 1000, METHODS, 3, 10, evaluate=ev)
    assert g_lines == o_lines                                  # hits and MAP identical to the last bit
    for (m, f, a), (_, _, b) in zip(o_lists, g_lists):
        assert [x[0] for x in a] == [x[0] for x in b], (m, f)
        assert [x[1].hex() for x in a] == [float(x[1]).hex() for x in b], (m, f)
