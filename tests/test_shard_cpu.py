"""The N > 1 layout on CPU (gloo, world_size 2): agent blocks, seeds independent of the sharding,
and the CMA-ES fitness all-gather giving every rank the single-process result bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ddrl_amd.shard import agent_seeds, all_gather_fitness, block_range, block_sizes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blocks_partition():
    for n in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            rs = [block_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert max(block_sizes(n, world)) - min(block_sizes(n, world)) <= 1
    with pytest.raises(ValueError):
        block_range(8, 2, 2)


def test_seeds_do_not_depend_on_world():
    one = agent_seeds(1000, 37, 0, 1)
    for world in (2, 3, 8):
        assert np.array_equal(np.concatenate([agent_seeds(1000, 37, r, world) for r in range(world)]), one)


def test_all_gather_world1():
    f = np.arange(5.0)
    assert np.array_equal(all_gather_fitness(f, 5, 0, 1), f)
    with pytest.raises(ValueError):
        all_gather_fitness(f, 6, 0, 1)


@pytest.mark.parametrize("flavour", ["host", "dev"])
@pytest.mark.parametrize("lam", [8, 9])          # even and uneven blocks
def test_sharded_cmaes_gloo_world2(ora, tmp_path, lam, flavour):
    """flavour dev: ShardedCmaes.generation_dev -- tensors gathered in place, the gathered buffer's address into update_dev"""
    N, gens = 6, 5
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + lam + (20 if flavour == "dev" else 0)),
           os.path.join(ROOT, "tests", "shard_worker.py"), out, str(lam), str(N), str(gens), flavour]
    subprocess.run(cmd, check=True, env=env, timeout=240, cwd=ROOT)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(2)]
    # single-process reference run
    es = ora.Cmaes(0.1 * (np.arange(N) + 1), [0.5] * N, 3, lam)
    fits = []
    for _ in range(gens):
        X = es.sample()
        if flavour == "dev":
            import torch
            t = torch.from_numpy(X)
            f = (t * t).sum(1).numpy()
        else:
            f = (X * X).sum(1)
        fits.append(f.tolist())
        es.update(f)
    for r in res:
        assert r["world"] == 2
        assert r["fits"] == fits                                  # gathered fitness = full fitness, bit for bit
        assert r["sigma"] == es.get("sigma") and r["xmean"] == es.vec("xmean").tolist()
        assert r["index"] == es.index().tolist()
    assert res[0]["block"][1] == res[1]["block"][0] and res[1]["block"][1] == lam
    assert res[0]["seeds"] + res[1]["seeds"] == list(range(1000, 1037))
