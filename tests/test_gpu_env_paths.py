"""Code paths that librwr selects from environment variables read once per process, each reached in a fresh child
process (tests/env_child.py) and compared with the C restatement of the reference: EXACT bitwise, FAST within 1e-6.

* value-free matrix path on / off (RWR_VALUE_FREE): uniform-row-weight graphs must give bit-identical results either way
  (Graph.cs:79-81, Model.cs:84-88: the product fl(fl((1-d) rank[i]) * weight) is the same double for every link of i);
* the "graph beyond the L2s" paths -- two-phase row order of the single-seed SpMV, three frontier iterations, FAST single
  seed on the list-order kernels, 32-seed tiles -- which the default threshold only selects from 2 M nodes on (the 100 M-like
  headline configuration): RWR_BIG_N lowers the threshold so that oracle-sized graphs run them;
* the general single-seed path on ego-network-sized graphs (RWR_SMALL=0), which by default take the one-launch kernel of
  small.hip -- both must equal the oracle bit for bit;
* the source-block sweep of the single-seed SpMV (sweep.hip), which by default serves graphs of >= 50 000 nodes only:
  RWR_SWEEP_MIN_N=0 sends the oracle-sized graphs through it, RWR_SWEEP_BN shrinks the LDS block so that they span many
  blocks, RWR_HUB_T lowers the hub-row threshold so that hub kernel and sweep share the rows of one step, RWR_SWEEP_WGS
  leaves the sweep so few workgroups that it can only take the ITEM rows (its form on graphs of more than ~0.5 M rows) and
  skips the blocks they do not read, with the other rows on the row-binned kernel beside it;
* the frontier iterations of a single seed (rows none of whose in-neighbours is non-zero are skipped), which by default
  only graphs of >= 200 000 nodes take: RWR_ACT_ITERS forces them, with RWR_HUB_T low enough that hub rows are among the
  skipped and the walked ones."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_child(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_child.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "ENV_CHILD_OK" in p.stdout


@pytest.mark.parametrize("env", [
    {"RWR_VALUE_FREE": "1"},
    {"RWR_VALUE_FREE": "0"},
    {"RWR_BIG_N": "100", "RWR_SPMV_PHASES": "1", "RWR_ACT_ITERS": "3"},
    {"RWR_BIG_N": "100", "RWR_VALUE_FREE": "0"},
    {"RWR_SWEEP_MIN_N": "0", "RWR_SMALL": "0"},
    {"RWR_SWEEP_MIN_N": "0", "RWR_SMALL": "0", "RWR_SWEEP_BN": "512"},
    {"RWR_SWEEP_MIN_N": "0", "RWR_SMALL": "0", "RWR_SWEEP_BN": "128", "RWR_HUB_T": "96"},
    {"RWR_SWEEP_MIN_N": "0", "RWR_SMALL": "0", "RWR_SWEEP_WGS": "4"},
    {"RWR_SWEEP_MIN_N": "0", "RWR_SMALL": "0", "RWR_SWEEP_WGS": "3", "RWR_SWEEP_BN": "512", "RWR_HUB_T": "96"},
    {"RWR_SWEEP": "0"},
    {"RWR_SMALL": "0", "RWR_ACT_ITERS": "2", "RWR_HUB_T": "96"},
    {"RWR_SMALL": "0", "RWR_ACT_ITERS": "2", "RWR_HUB_T": "96", "RWR_VALUE_FREE": "0"},
    {"RWR_HUB_SCAN": "0"},
    {"RWR_SMALL": "0"},
    {"RWR_SMALL": "0", "RWR_VALUE_FREE": "0"},
    {"RWR_SPMM": "0"},
    {"RWR_CHAIN": "0"},
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_env_selected_paths_match_the_oracle(env):
    run_child(env)
