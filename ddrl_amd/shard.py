"""Multi-GPU layout of the hot path: one process per GPU, agents / individuals sharded over ranks.

DDPG and NFAC agents are fully independent (own replay, nets, optimiser state): rank r owns a
contiguous block of agents and there is NO data-path collective.  CMA-ES has the one
real exchange of the path: every rank evaluates lambda / world individuals and the lambda fitness
values are all-gathered (RCCL on GPUs, gloo in the CPU tests) before every rank applies the identical
ranking + distribution update to its replicated (m, sigma, C, B, D, ps, pc).
"""
import numpy as np


def block_range(n_total, rank, world):
    """[lo, hi) of the contiguous block rank owns; blocks differ by at most one element."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside 0..%d" % (rank, world - 1))
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def block_sizes(n_total, world):
    return [block_range(n_total, r, world)[1] - block_range(n_total, r, world)[0] for r in range(world)]


def agent_seeds(base_seed, n_total, rank, world):
    """Per-agent seeds that do not depend on how the population is sharded."""
    lo, hi = block_range(n_total, rank, world)
    return (base_seed + np.arange(lo, hi)).astype(np.uint32)


def all_gather_fitness(local_fitness, n_total, rank, world, device=None):
    """All-gather of the per-individual fitness (fp64, as c-cmaes' rgFunVal): local block -> [n_total].
    torch.distributed must be initialised when world > 1 (backend nccl = RCCL on GPUs, gloo on CPU).
    Uneven blocks are padded to the largest block for the collective."""
    local = np.ascontiguousarray(local_fitness, np.float64)
    lo, hi = block_range(n_total, rank, world)
    if local.shape != (hi - lo,):
        raise ValueError("rank %d owns %d individuals, got %s" % (rank, hi - lo, local.shape))
    if world == 1:
        return local.copy()
    import torch
    import torch.distributed as dist
    sizes = block_sizes(n_total, world)
    m = max(sizes)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    buf = torch.zeros(m, dtype=torch.float64, device=dev)
    buf[: hi - lo] = torch.from_numpy(local).to(dev)
    out = torch.empty(world * m, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, m)
    return np.concatenate([out[r, : sizes[r]] for r in range(world)])


def all_gather_fitness_torch(local, n_total, rank, world):
    """The same collective on torch tensors that stay where they are: `local` (fp64, this rank's block, on the GPU under
    RCCL / on the CPU under gloo) -> tensor [n_total] on the same device, ready for ddrl_cmaes_update_dev.  One
    all_gather_into_tensor of world x max-block doubles (32 KiB at lambda = 4096: latency-bound over xGMI)."""
    import torch
    import torch.distributed as dist
    lo, hi = block_range(n_total, rank, world)
    if tuple(local.shape) != (hi - lo,) or local.dtype != torch.float64:
        raise ValueError("rank %d owns %d individuals (fp64), got %s %s" % (rank, hi - lo, tuple(local.shape), local.dtype))
    if world == 1:
        return local.contiguous()
    sizes = block_sizes(n_total, world)
    m = max(sizes)
    buf = torch.zeros(m, dtype=torch.float64, device=local.device)
    buf[: hi - lo] = local
    out = torch.empty(world * m, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    if min(sizes) == m:
        return out
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])


class ShardedCmaes:
    """CMA-ES generation loop over `world` ranks: every rank holds the same distribution (same seed ->
    same z stream -> same population), evaluates its block of individuals, and the update runs on the
    all-gathered fitness.  `es` is any object with sample() -> X [lambda x N] and update(fitness)
    (ddrl_amd.Cmaes on a GPU)."""

    def __init__(self, es, lam, rank, world, device=None):
        self.es, self.lam, self.rank, self.world, self.device = es, lam, rank, world, device
        self.lo, self.hi = block_range(lam, rank, world)

    def generation(self, fitness_fn):
        X = self.es.sample()
        local = np.asarray([fitness_fn(X[k]) for k in range(self.lo, self.hi)], np.float64)
        fit = all_gather_fitness(local, self.lam, self.rank, self.world, self.device)
        self.es.update(fit)
        return fit

    def generation_dev(self, fitness_fn_torch):
        """The same generation without leaving the device: the population is a torch view of the library's buffer,
        `fitness_fn_torch(X_block)` returns this rank's fitness as an fp64 tensor on that device, the all-gather runs
        on device buffers (RCCL), and the gathered tensor's address goes straight into ddrl_cmaes_update_dev.
        Every rank samples the whole population (2 lambda N^2 flops: 28 us at lambda = 4096, N = 631); what is
        sharded is the evaluation."""
        import torch
        self.es.sample_dev()
        X = self.es.population_torch()
        local = fitness_fn_torch(X[self.lo:self.hi]).to(torch.float64)
        fit = all_gather_fitness_torch(local, self.lam, self.rank, self.world)
        if fit.is_cuda:
            torch.cuda.current_stream().synchronize()       # the library works on its own stream
        self.es.update_dev(fit.data_ptr())
        return fit

