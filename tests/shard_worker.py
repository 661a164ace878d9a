"""world_size-2 gloo worker for tests/test_shard_cpu.py (CPU only: the oracle stands in for the GPU
Cmaes object; what is under test is the sharding + all-gather logic of ddrl_amd/shard.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, lam, N, gens = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import oracle as ora
    from ddrl_amd.shard import ShardedCmaes, agent_seeds, block_range
    es = ora.Cmaes(0.1 * (np.arange(N) + 1), [0.5] * N, 3, lam)
    dev_path = len(sys.argv) > 5 and sys.argv[5] == "dev"
    if dev_path:
        # the device flavour of the loop (ShardedCmaes.generation_dev: torch tensors, all_gather_into_tensor on them, the
        # gathered buffer's ADDRESS handed to update_dev) with the oracle behind the three calls it makes of a Cmaes
        import ctypes
        import torch

        class DevFlavour:
            def __init__(self, es):
                self.es, self.X = es, None

            def sample_dev(self):
                self.X = torch.from_numpy(self.es.sample())

            def population_torch(self):
                return self.X

            def update_dev(self, ptr):
                fit = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(lam,)).copy()
                self.es.update(fit)
        sh = ShardedCmaes(DevFlavour(es), lam, rank, world, device="cpu")
    else:
        sh = ShardedCmaes(es, lam, rank, world, device="cpu")
    fits = []
    for _ in range(gens):
        if dev_path:
            fits.append(sh.generation_dev(lambda Xb: (Xb * Xb).sum(1)).tolist())
        else:
            fits.append(sh.generation(lambda x: float(np.sum(x * x))).tolist())
    res = dict(rank=rank, world=world, block=block_range(lam, rank, world), sigma=es.get("sigma"),
               xmean=es.vec("xmean").tolist(), index=es.index().tolist(), fits=fits,
               seeds=agent_seeds(1000, 37, rank, world).tolist())
    with open("%s.%d" % (out, rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
