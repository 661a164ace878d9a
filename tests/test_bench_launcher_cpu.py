"""bench.py --gpus N starts its N ranks itself: dry run on CPU (gloo, world_size 2), both
when bench.py is the launcher and when torch.distributed.run is; a real run on a box without GPUs fails per rank
and the launcher says which ranks failed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_launcher_spawns_world2_gloo():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=240, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launched_by"] == "bench.py"
    assert out["max_dt"] == pytest.approx(0.002)  # MAX over ranks of (rank + 1) ms


def test_under_torchrun_it_is_a_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", BENCH, "--gpus", "2", "--dry-run"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launched_by"] == "torch.distributed.run"


def test_without_gpus_every_rank_fails_and_is_named():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=240, env=_env(), cwd=ROOT)
    assert p.returncode != 0
    assert "rank 0 exit" in p.stderr and "rank 1 exit" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_one_dead_rank_ends_the_launch_quickly_and_is_named():
    """rank 1 dies before the rendezvous; rank 0 would wait in it for the collective library's timeout (minutes): the
    launcher polls its children, terminates the survivors and exits non-zero, naming the rank"""
    import time
    env = _env()
    env["DDRL_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "rank 1 exit 3" in p.stderr and "terminated by the launcher" in p.stderr
    assert time.time() - t0 < 30          # (the interpreter start + torch import dominate; the launcher reacts within 50 ms)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
