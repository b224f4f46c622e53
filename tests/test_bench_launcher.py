"""`python bench.py --gpus N` typed as is must start its N ranks itself (the driver's scaling runs call it that way).
CPU rehearsal over gloo: the parent launches the ranks through torch.distributed.run on 127.0.0.1, the ranks shard the
seed set, meet at the barrier, reduce the max time, and rank 0 prints ONE JSON line.  No graph and no kernels run here
(--dry-run); the GPU path of the same launcher is exercised by tests/test_gpu_multiprocess.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    out = _run(["--gpus", "2", "--dry-run", "--config", "tiny", "--seeds-per-gpu", "8"])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["seeds_per_gpu"] == 8
    (a0, a1), (b0, b1) = out["shards"]
    assert a0 <= a1 < b0 <= b1                      # contiguous, disjoint seed blocks per rank
    assert out["elapsed_max_s"] >= 0.02             # max over ranks (rank 1 sleeps longer)


def test_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1", "--dry-run", "--config", "tiny"])
    assert out["n_gpus"] == 1 and len(out["shards"]) == 1


def test_under_torchrun_the_process_is_a_rank():
    """Launched by torch.distributed.run (WORLD_SIZE set) the script must NOT start a second launcher."""
    env = {k: v for k, v in os.environ.items()}
    import socket
    with socket.socket() as sk:                     # a free rendezvous port (a fixed one collides with parallel test runs)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--config", "tiny"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
