#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X.

metric: actor+critic minibatch updates/sec (node) for the 1024-agent HalfCheetah DDPG sweep.
One "step" = one population update = one pass of the fupd loop body of DeepQNAg::end_episode
(DeepQNAg.hpp:343-466) for EVERY agent on this GPU (device-side replay index draw + the fused
update kernel).  Workload (config 5): G agents per GPU, S=18, A=6, "mid" nets (q=100,100 / a=100,100),
B=64, DDPG + EWC (penalty live, Fisher accumulating, ewc_for_critic=1) + inverting gradients,
replay of 10 000 synthetic transitions per agent, per-agent hyper-parameter sweep.  Everything is
resident in HBM when the timed region starts.  Agents are independent: ranks share nothing
(weak scaling, no data-path collective); torch.distributed is used for the barrier and the
max-over-ranks only.

  python bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZES = {"small": ((20,), (50, 7)), "mid": ((100, 100), (100, 100)), "large": ((400, 300), (400, 300))}
HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md chip table (spec; 6.29e12 measured copy rate)
MFMA_F32_PEAK = 157.3e12   # FLOP/s dense fp32 matrix


def n_params(dims):
    return sum(dims[i + 1] * dims[i] + dims[i + 1] for i in range(len(dims) - 1))


def p_weights(dims):
    return sum(dims[i + 1] * dims[i] for i in range(len(dims) - 1))


def algorithmic_work(S, A, B, ha, hq, ewc):
    """SURVEY.md 8(d) / BASELINE.md section 4: bytes and flops of ONE agent-update."""
    da, dq = [S] + list(ha) + [A], [S + A] + list(hq) + [1]
    Na, Nc, Pa, Pc = n_params(da), n_params(dq), p_weights(da), p_weights(dq)
    # Adam r/w of w,m,v (24) + target r/w (8) + w re-read (4) = 36 B/param; EWC: +8 (F-hat, w*) +8 (Fisher EMA r/w)
    per_param = 36 + (16 if ewc else 0)
    byts = per_param * (Na + Nc) + 4 * B * (2 * S + A + 2)
    flops = 2 * B * (4 * Pa + 6 * Pc - 2 * (S + A) * hq[0] + A * hq[0] - S * ha[0])
    return byts, flops, Na, Nc


def synth_rows(agent, n, S, A):
    """BASELINE.md section 3 synthetic replay, generator seeded 1000 + agent_id."""
    rng = np.random.RandomState(1000 + agent)
    s = rng.uniform(-1, 1, (n, S)); a = rng.uniform(-1, 1, (n, A)); s2 = rng.uniform(-1, 1, (n, S))
    r = rng.normal(0, 1, (n, 1)); goal = (rng.uniform(0, 1, (n, 1)) < 0.01)
    return np.concatenate([s, a, s2, r, goal], axis=1).astype(np.float32)


def cpu_baseline(S, A, B, ha, hq, ewc, seconds, threads):
    """The fp64 CPU oracle (oracle/ddpg_ref.c: a restatement standing in for the unavailable Caffe
    CPU path, kind "port"), one independent agent per thread (ctypes releases the GIL), timed on
    this box's host cores for a bounded sample of the same workload."""
    from oracle import oracle as ora
    ora.lib()
    agents = []
    for t in range(threads):
        ag = ora.Ddpg(S, A, B, ha, hq, replay_cap=2000, ewc=100.0 if ewc else -1.0, ewc_decay=-1.0,
                      ewc_for_critic=1, seed=3000 + t)
        rng = np.random.RandomState(2000 + t)
        for n in ag.nets:
            pass
        wa = rng.normal(0, 0.01, ag.ann.np); wq = rng.normal(0, 0.01, ag.qnn.np)
        ag.ann.set_weights(wa); ag.ann_target.set_weights(wa); ag.qnn.set_weights(wq); ag.qnn_target.set_weights(wq)
        rows = synth_rows(t, 2000, S, A).astype(np.float64)
        for row in rows:
            ag.insert(row[:S], row[S:S + A], row[2 * S + A], row[S + A:2 * S + A], row[2 * S + A + 1] != 0)
        if ewc:
            ag.update(); ag.ann.update_best_param(1.0); ag.qnn.update_best_param(1.0); ag.ewc_setup()
        agents.append(ag)
    counts = [0] * threads
    stop = time.perf_counter() + seconds

    def work(i):
        while time.perf_counter() < stop:
            agents[i].update()
            counts[i] += 1

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    return sum(counts) / dt, sum(counts), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--agents", type=int, default=1024, help="agents per GPU")
    ap.add_argument("--size", default="mid", choices=sorted(SIZES))
    ap.add_argument("--no-ewc", action="store_true")
    ap.add_argument("--replay", type=int, default=10000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--updates-per-launch", type=int, default=8,
                    help="consecutive updates per agent fused into one launch (ddrl_pop_ddpg_update_n): the reference's "
                         "fupd loop, 1 + force_more_update = 8 in its ddpg.ini; bit-identical to separate launches")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    S, A, B = 18, 6, 64
    ha, hq = SIZES[args.size]
    ewc = not args.no_ewc
    G = args.agents

    # CPU baseline first: before anything initialises the GPU in this process
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        ups, n, dt = cpu_baseline(S, A, B, ha, hq, ewc, args.cpu_seconds, threads)
        cpu = {"value": ups, "unit": "updates/s", "cores": threads, "kind": "port",
               "sample": "%d updates of the same per-agent workload (%s nets, B=64, EWC %s) by the fp64 oracle, "
                         "one agent per thread, %.1f s" % (n, args.size, "on" if ewc else "off", dt)}

    import torch
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import ddrl_amd
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=B, replay_memory=args.replay,
                              hidden_unit_a=ha, hidden_unit_q=hq, inverting_grad=1,
                              ewc=100.0 if ewc else -1.0, ewc_decay=-1.0, ewc_fisher_beta=0.999,
                              ewc_for_critic=1, ewc_best_param_method=0, device=local_rank)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=2000 + rank)
    rng = np.random.RandomState(3000 + rank)
    pop.set_hyper("alpha_a", 10 ** rng.uniform(-5, -2, G)); pop.set_hyper("alpha_v", 10 ** rng.uniform(-5, -2, G))
    pop.set_hyper("tau_soft_update", 10 ** rng.uniform(-4, -1, G))
    pop.seed_replay((3000 + rank * G + np.arange(G)).astype(np.uint32))
    for g in range(G):
        pop.replay_load(g, synth_rows(rank * G + g, args.replay, S, A))
    upl = max(1, min(args.updates_per_launch, 64))
    if max(ha + hq) > 128:
        upl = 1                      # the layer-by-layer path launches per update anyway

    def run_updates(k):              # k updates of every agent ("steps"), upl per launch
        full, rem = divmod(k, upl)
        for _ in range(full):
            pop.update_n(upl) if upl > 1 else pop.update()
        if rem:
            pop.update_n(rem) if rem > 1 else pop.update()

    # bring EWC to its steady state: Fisher accumulated, snapshot taken, penalty live
    run_updates(upl)
    if ewc:
        pop.update_best_param(np.ones(G))
        pop.ewc_setup()
    run_updates(args.warmup)
    pop.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pop.sync()

    pop.set_profiling(True)
    pop.profile_read()
    barrier()
    t0 = time.perf_counter()
    run_updates(args.steps)
    pop.sync()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = pop.profile_read()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    byts, flops, Na, Nc = algorithmic_work(S, A, B, ha, hq, ewc)
    value = world * G * args.steps / dt
    kavg = kernel_ms * 1e-3 / max(launches, 1)          # per launch, as rocprofv3 reports it
    upl_eff = args.steps / max(launches, 1)             # updates per launch actually run
    achieved = byts * G * upl_eff / kavg
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tp):
        try:
            # measured HBM bytes per launch (PMC, profiles/README.md) for 1 and for 8 updates per launch; a launch with
            # another count (--steps not a multiple of 8 leaves a shorter last launch) is priced per update from the nearer one
            tj = json.load(open(tp))
            per_update = {n: tj.get("bytes_per_launch_%d_%s_upl%d" % (G, args.size, n)) for n in (1, 8)}
            per_update = {n: v / n for n, v in per_update.items() if v}
            if per_update:
                near = min(per_update, key=lambda n: abs(n - upl_eff))
                traffic = per_update[near] * upl_eff
        except Exception:
            traffic = None
    out = {
        "metric": "actor+critic minibatch updates/sec (node), 1024-agent HalfCheetah DDPG sweep",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 5: DDPG%s + inverting gradients, HalfCheetah dims S=18 A=6, %s nets a=%s q=%s, B=64, "
                               "%d agents per GPU, replay %d/agent, device mt19937 replay sampling; %d consecutive updates per launch "
                               "(the fupd loop, force_more_update=%d)"
                               % ("+EWC" if ewc else "", args.size, "x".join(map(str, ha)), "x".join(map(str, hq)), G, args.replay, upl, upl - 1),
                   "agents_per_gpu": G, "agents_total": world * G, "parallelism": "agents sharded, no collective"},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "kernel": "ddpg_update_kernel" if max(ha + hq) <= 128 else "wide_* layer kernels, optimiser inside the dW kernels (one update = 45 launches)",
                     "kernel_avg_ms": kavg * 1e3, "launches": launches,
                     "updates_per_launch": upl_eff, "algorithmic_bytes_per_launch": byts * G * upl_eff,
                     "algorithmic_bytes_per_agent_update": byts,
                     "mfma_f32_frac": flops * G * upl_eff / kavg / MFMA_F32_PEAK, "flops_per_agent_update": flops},
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
