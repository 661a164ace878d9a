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

N > 1: when the process was not started by torch.distributed.run (no RANK / WORLD_SIZE in the environment),
this process only LAUNCHES: it starts N fresh children, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set), before anything here has touched the GPU, relays rank 0's JSON line and
exits with the worst child status.  At N > 1 the line carries both sweeps: `value` is the weak-scaling one
(--agents per GPU, default 1024), `strong` the metric's own configuration (1024 agents in total = 1024 / N per GPU).
--dry-run: the launcher / rendezvous / clock reduction without GPU or library (gloo; tests/test_bench_launcher_cpu.py).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
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


def cpu_baseline(S, A, B, ha, hq, ewc, seconds, threads, replay):
    """The fp64 CPU oracle (oracle/ddpg_ref.c: a restatement standing in for the unavailable Caffe
    CPU path, kind "port"), one independent agent per thread (ctypes releases the GIL), timed on
    this box's host cores for a bounded sample of the same workload."""
    from oracle import oracle as ora
    ora.lib()
    agents = []
    for t in range(threads):
        ag = ora.Ddpg(S, A, B, ha, hq, replay_cap=replay, ewc=100.0 if ewc else -1.0, ewc_decay=-1.0,
                      ewc_for_critic=1, seed=3000 + t)
        rng = np.random.RandomState(2000 + t)
        wa = rng.normal(0, 0.01, ag.ann.np); wq = rng.normal(0, 0.01, ag.qnn.np)
        ag.ann.set_weights(wa); ag.ann_target.set_weights(wa); ag.qnn.set_weights(wq); ag.qnn_target.set_weights(wq)
        rows = synth_rows(t, replay, S, A).astype(np.float64)
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


PAIR_MODE = 0
RESIDENT = 0
LOOP = 1
W8 = 0
KERNEL_SOURCES = ("kernels_ddpg.hip", "kernels_ddpg_pair.hip", "ddpg_update_body.h", "ddpg_device.h", "net_device.h", "mlp_device.h", "pop_internal.h")


def kernel_hash():
    """identifies the update kernel's sources: ties a PMC traffic figure to the build it was measured on"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        p = os.path.join(ROOT, "ddrl_amd", "csrc", f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def canned_traffic(G, size, upl_eff):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic_latest.json; bench.py cannot collect
    PMC counters itself).  Only reported when that file was measured on the kernel sources of this tree."""
    tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(tp):
        return None, "not measured"
    try:
        tj = json.load(open(tp))
    except Exception:
        return None, "profiles/traffic_latest.json unreadable"
    if tj.get("kernel_hash") != kernel_hash():
        return None, "stale: profiles/traffic_latest.json was measured on other kernel sources (%s)" % tj.get("kernel_hash", "unhashed, round 1")
    per_update = {n: tj.get("bytes_per_launch_%d_%s_upl%d" % (G, size, n)) for n in (1, 8)}
    per_update = {n: v / n for n, v in per_update.items() if v}
    if not per_update:
        return None, "no PMC pass for this workload"
    near = min(per_update, key=lambda n: abs(n - upl_eff))
    return per_update[near] * upl_eff, "canned: %s" % tj.get("source", "profiles/traffic_latest.json")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run: this process never touches the GPU; it starts N
    fresh interpreters, one per GPU, and relays rank 0's line.  (The other way round -- re-executing a process
    that has initialised HIP -- is what must never happen.)  Every child is polled: when one exits non-zero the others
    (which would sit in the rendezvous or a barrier until the collective library's timeout) are terminated -- they are
    fresh children of this launcher, nothing is re-executed -- and the launcher exits non-zero naming the rank."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, DDRL_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)   # never blocks the polling below
    reader.start()
    first_bad = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            first_bad = bad
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    if first_bad:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        failed = set(r for r, _ in first_bad)
        sys.stderr.write("bench.py: ranks failed: %s; final status: %s\n" % (
            ", ".join("rank %d exit %d" % rc for rc in first_bad),
            ", ".join("rank %d exit %d%s" % (r, p.returncode, "" if r in failed or p.returncode == 0 else " (terminated by the launcher)")
                      for r, p in enumerate(procs))))
        sys.exit(1)
    reader.join(timeout=10)
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()


def build_population(G, rank, first_agent, S, A, B, ha, hq, ewc, replay, local_rank):
    import ddrl_amd
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=B, replay_memory=replay,
                              hidden_unit_a=ha, hidden_unit_q=hq, inverting_grad=1,
                              ewc=100.0 if ewc else -1.0, ewc_decay=-1.0, ewc_fisher_beta=0.999,
                              ewc_for_critic=1, ewc_best_param_method=0, device=local_rank)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=2000 + rank)
    pop.set_option("pair", PAIR_MODE)
    pop.set_option("resident", RESIDENT)
    pop.set_option("loop", LOOP)
    pop.set_option("w8", W8)
    # per-agent hyper-parameters and replay depend on the agent's id in the whole sweep, not on the sharding
    ids = first_agent + np.arange(G)
    hyp = np.array([np.random.RandomState(3000 + i).uniform(size=3) for i in ids]).reshape(G, 3)
    pop.set_hyper("alpha_a", 10 ** (-5 + 3 * hyp[:, 0])); pop.set_hyper("alpha_v", 10 ** (-5 + 3 * hyp[:, 1]))
    pop.set_hyper("tau_soft_update", 10 ** (-4 + 3 * hyp[:, 2]))
    pop.seed_replay((3000 + ids).astype(np.uint32))
    for g in range(G):
        pop.replay_load(g, synth_rows(int(ids[g]), replay, S, A))
    return pop


SETUP_UPDATES = 64


def timed_sweep(pop, G, steps, warmup, upl, ewc, barrier):
    """W untimed + K timed population updates; returns (seconds, kernel_ms, launches)"""
    def run_updates(k):              # k updates of every agent ("steps"), upl per launch
        full, rem = divmod(k, upl)
        for _ in range(full):
            pop.update_n(upl) if upl > 1 else pop.update()
        if rem:
            pop.update_n(rem) if rem > 1 else pop.update()

    # set-up, before the W warm-up steps the caller asked for: one launch accumulates a Fisher average, the EWC snapshot is taken
    # from it (host-side bookkeeping with its own synchronisations: the GPU idles through it), and SETUP_UPDATES updates then
    # run the state a sweep is in -- penalty live, Fisher average restarting -- long enough for the GPU to be back in its loaded
    # state when the warm-up starts.  With the snapshot taken right before the warm-up instead, a short run (K = 20) measured
    # 0.377 ms per update; with these 0.353 ms (same box; 0.350 ms at K = 200).  Reported as config.setup_updates.
    run_updates(upl)
    if ewc:
        pop.update_best_param(np.ones(G))
        pop.ewc_setup()
    run_updates(max(upl, SETUP_UPDATES // upl * upl))
    run_updates(warmup)
    pop.sync()
    pop.set_profiling(True)
    pop.profile_read()
    barrier(pop)
    t0 = time.perf_counter()
    run_updates(steps)
    pop.sync()
    barrier(pop)
    dt = time.perf_counter() - t0
    kernel_ms, launches = pop.profile_read()
    pop.set_profiling(False)
    return dt, kernel_ms, launches


NFAC = dict(S=45, A=17, hidden=(64, 64), T=500, episodes=4, fitted=10, critic_steps=10, actor_steps=10)


def nfac_rows(agent, T, S, A):
    rng = np.random.RandomState(5000 + agent)
    s = np.clip(np.cumsum(rng.normal(0, 0.1, (T + 1, S)), axis=0), -1, 1)
    pa = np.tanh(rng.normal(0, 0.7, (T, A))); a = np.clip(pa + rng.normal(0, 0.2, (T, A)), -1, 1)
    return np.concatenate([s[:-1], pa, a, s[1:], rng.normal(0, 1, (T, 1)), np.zeros((T, 1))], axis=1).astype(np.float32)


def nfac_cpu_baseline(seconds, threads):
    """oracle/nfac_ref.c (fp64 restatement, kind "port") on the same per-agent workload, one agent per thread"""
    from oracle import oracle as ora
    ora.lib()
    c = NFAC
    ends = [c["T"] * (k + 1) // c["episodes"] for k in range(c["episodes"])]
    agents, data = [], []
    for t in range(threads):
        ag = ora.Nfac(c["S"], c["A"], c["hidden"], c["hidden"], lambda_=0.9, gae=1, number_fitted_iteration=c["fitted"],
                      stoch_iter_actor=c["actor_steps"], stoch_iter_critic=c["critic_steps"], ewc=100.0, ewc_for_critic=1)
        rng = np.random.RandomState(2000 + t)
        ag.ann.set_weights(rng.normal(0, 0.1, ag.ann.np)); ag.vnn.set_weights(rng.normal(0, 0.1, ag.vnn.np))
        rows = nfac_rows(t, c["T"], c["S"], c["A"]).astype(np.float64)
        S, A = c["S"], c["A"]
        data.append((rows[:, :S], rows[:, S:S + A], rows[:, S + A:S + 2 * A], rows[:, S + 2 * A:2 * S + 2 * A], rows[:, 2 * S + 2 * A],
                     rows[:, 2 * S + 2 * A + 1].astype(np.int32)))
        agents.append(ag)
    counts = [0] * threads
    stop = time.perf_counter() + seconds

    def work(i):
        while time.perf_counter() < stop:
            agents[i].update(*data[i], ends)
            counts[i] += 1

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return sum(counts) / (time.perf_counter() - t0), sum(counts), time.perf_counter() - t0


def nfac_main(args, rank, local_rank, world):
    out = nfac_measure(args, rank, local_rank, world)
    if out is not None:
        print(json.dumps(out))


def nfac_measure(args, rank, local_rank, world, cpu_leg=True):
    """--workload nfac: SURVEY.md 8(d) config 3.  A step = one NFAC(lambda)-V update (OfflineCaclaAg::end_episode's learning
    branch, OfflineCaclaAg.hpp:423-594) of every agent on this GPU: 10 fitted iterations x 10 full-batch critic steps + 10
    actor steps on a 500-row trajectory, EWC live on both nets.  Agents are sharded over ranks, no collective."""
    c = NFAC
    G = args.agents if args.agents != 1024 else 256
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and cpu_leg:
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        ups, n, dt = nfac_cpu_baseline(args.cpu_seconds, threads)
        cpu = {"value": ups, "unit": "agent updates/s", "cores": threads, "kind": "port", "host_cpu_count": os.cpu_count(),
               "sample": "%d updates of the same per-agent workload by the fp64 oracle (oracle/nfac_ref.c), one agent per thread on %d threads, %.1f s"
                         % (n, threads, dt)}
    import torch
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import ddrl_amd
    cfg = ddrl_amd.NfacConfig(n_agents=G, nb_sensors=c["S"], nb_motors=c["A"], hidden_unit_a=c["hidden"], hidden_unit_v=c["hidden"],
                              max_trajectory=512, ewc=100.0, ewc_fisher_beta=0.999, ewc_for_critic=1, ewc_best_param_method=1,
                              number_fitted_iteration=c["fitted"], stoch_iter_actor=c["actor_steps"], stoch_iter_critic=c["critic_steps"],
                              device=local_rank)
    cfg.set("lambda", 0.9); cfg.set("gae", 1)
    pop = ddrl_amd.NfacPopulation(cfg, init_seed=2000 + rank)
    ends = [c["T"] * (k + 1) // c["episodes"] for k in range(c["episodes"])]

    def load():
        for g in range(G):
            pop.traj_load(g, nfac_rows(rank * G + g, c["T"], c["S"], c["A"]), ends)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pop.sync()

    load(); pop.update()
    pop.update_best_param(np.ones(G)); pop.ewc_setup()
    for _ in range(args.warmup):
        load(); pop.update()
    steps = args.steps
    dt = 0.0
    pop.set_profiling(True); pop.profile_read()
    for _ in range(steps):                 # the trajectories are consumed by an update: reloading them is outside the timed region
        load()
        barrier()
        t0 = time.perf_counter()
        pop.update()
        barrier()
        dt += time.perf_counter() - t0
    kernel_ms, launches = pop.profile_read()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.destroy_process_group()
    if rank != 0:
        return None
    Pw = c["S"] * 64 + 64 * 64 + 64                       # the critic's weights
    flops_launch = G * c["T"] * (4 * Pw + 2 * (64 * 64 + 64))   # forward + dW over all layers, dX below the first: 2 flop / MAC
    kavg = kernel_ms * 1e-3 / max(launches, 1)
    return ({
        "metric": "NFAC(lambda)-V agent updates/sec (config 3)", "value": world * G * steps / dt, "unit": "agent updates/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 3: NFAC(lambda)-V, S=45 A=17, 64x64 nets, %d agents per GPU, T=500 rows in 4 episodes, lambda 0.9 gae, "
                               "10 fitted iterations x 10 critic steps + 10 actor steps per update, EWC live on both nets" % G,
                   "agents_per_gpu": G, "agents_total": world * G, "parallelism": "agents sharded, no collective",
                   "adam_iterations_per_s": world * G * steps * (c["fitted"] * c["critic_steps"] + c["actor_steps"]) / dt},
        "roofline": {"bound": "mfma", "achieved": flops_launch / kavg / 1e12, "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s",
                     "frac": flops_launch / kavg / MFMA_F32_PEAK, "traffic": None,
                     "kernel": "nfac_grad_kernel<0, ...> (the critic's full-batch gradient: %d of the %d launches of an update)"
                               % (c["fitted"] * c["critic_steps"], 2 * c["fitted"] * c["critic_steps"] + 2 * c["actor_steps"] + 3 * c["fitted"] + 4),
                     "kernel_avg_ms": kavg * 1e3, "launches": launches, "algorithmic_flops_per_launch": flops_launch},
        "cpu_baseline": cpu})


def cmaes_measure(N, lam, dev, gens):
    """SURVEY.md 8(d) config 4 on one GPU (tools/bench_cmaes.py): generations / s of CMA-ES at dimension N, lambda individuals,
    synthetic quadratic fitness.  dev False = parity mode (c-cmaes' own random stream + the host Householder / QL
    eigendecomposition, bit-identical to the goldens), True = throughput mode (device RNG, device eigensolver, device fitness)"""
    import torch
    import ddrl_amd
    es = ddrl_amd.Cmaes(np.full(N, 0.1), np.full(N, 0.3), 3, lam)
    es.set_option("device_rng", int(dev)); es.set_option("device_eigen", int(dev))
    target = torch.linspace(-0.5, 0.5, N, device="cuda", dtype=torch.float64) if dev else np.linspace(-0.5, 0.5, N)
    ts = []
    for g in range(gens):
        t0 = time.perf_counter()
        if dev:
            es.sample_dev()
            f = ((es.population_torch().double() - target) ** 2).sum(1)
            torch.cuda.synchronize()
            es.update_dev(f.data_ptr())
        else:
            X = es.sample()
            es.update(((X - target) ** 2).sum(1))
        es.get("sigma")
        ts.append(time.perf_counter() - t0)
    es.close()
    per = float(np.median(ts))
    flops = 2.0 * lam * N * N + 2.0 * (lam // 2) * N * N
    return {"generations_per_s": 1.0 / per, "ms_per_generation": per * 1e3, "mfma_f32_frac_whole_generation": flops / per / MFMA_F32_PEAK}


def extra_measurements(args, S, A, B, ewc, upl, barrier, budget_s=75.0):
    """The secondary configurations on the driver's record (N = 1 only, after the headline sweep, outside its timed region):
    the metric's own per-GPU share of a node (128 agents), the small and large nets, NFAC config 3, CMA-ES config 4.  Each
    entry is its own short sweep of the same kind as the headline one; an entry that would overrun the budget is skipped and
    says so."""
    t_start = time.perf_counter()
    out = {}

    def left():
        return budget_s - (time.perf_counter() - t_start)

    def ddpg_entry(name, G, size, steps, warmup, replay):
        if left() < 8:
            out[name] = {"skipped": "time budget"}
            return
        ha, hq = SIZES[size]
        u = upl if max(ha + hq) <= 128 else 1
        pop = build_population(G, 0, 0, S, A, B, ha, hq, ewc, replay, 0)
        dt, kms, launches = timed_sweep(pop, G, steps, warmup, u, ewc, barrier)
        del pop
        byts, flops, _, _ = algorithmic_work(S, A, B, ha, hq, ewc)
        kavg = kms * 1e-3 / max(launches, 1)
        upl_eff = steps / max(launches, 1)
        out[name] = {"workload": "%d agents, %s nets, replay %d/agent, %d updates per launch" % (G, size, replay, u),
                     "value": G * steps / dt, "unit": "updates/s", "ms_per_step": dt / steps * 1e3,
                     "hbm_frac": byts * G * upl_eff / kavg / HBM_PEAK, "mfma_f32_frac": flops * G * upl_eff / kavg / MFMA_F32_PEAK}

    try:
        ddpg_entry("agents128", 128, "mid", 96, 16, args.replay)
        ddpg_entry("small", 1024, "small", 96, 16, 2000)
        ddpg_entry("large", 1024, "large", 40, 8, 2000)
        if left() > 12:
            class _A:                                   # the NFAC sweep's arguments
                agents, steps, warmup, no_cpu, cpu_threads, cpu_seconds = 1024, 10, 2, True, 0, 0.0
            n = nfac_measure(_A, 0, 0, 1, cpu_leg=False)
            out["nfac"] = {"workload": n["config"]["workload"], "value": n["value"], "unit": n["unit"], "ms_per_update": n["ms_per_step"],
                           "mfma_f32_frac_gradient_launch": n["roofline"]["frac"]}
        else:
            out["nfac"] = {"skipped": "time budget"}
        cm = {}
        for N in (151, 631):
            for dev in (False, True):
                key = "N%d_%s" % (N, "throughput" if dev else "parity")
                if left() < (6 if N == 631 else 3):
                    cm[key] = {"skipped": "time budget"}
                    continue
                cm[key] = cmaes_measure(N, 4096, dev, 16 if dev else 4)      # (median per generation: the first decompositions start cold)
        out["cmaes"] = dict(cm, workload="config 4: CMA-ES lambda=4096 mu=2048 on one GPU, synthetic quadratic fitness")
    except Exception as e:                               # the headline line must not be lost to a secondary sweep
        out["error"] = "%s: %s" % (type(e).__name__, e)
    out["seconds"] = time.perf_counter() - t_start
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ddpg", choices=("ddpg", "nfac"),
                    help="ddpg: BASELINE.json's metric (config 5); nfac: SURVEY.md 8(d) config 3, a second line of the same shape")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--agents", type=int, default=1024, help="agents per GPU of the weak-scaling sweep")
    ap.add_argument("--total-agents", type=int, default=1024, help="agents of the strong-scaling sweep (N > 1): the metric's 1024")
    ap.add_argument("--size", default="mid", choices=sorted(SIZES))
    ap.add_argument("--no-ewc", action="store_true")
    ap.add_argument("--replay", type=int, default=10000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling sweep")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: skip the secondary configurations (`extra` in the JSON line)")
    ap.add_argument("--pair", type=int, default=0, help="0: one agent per workgroup always (default); 1: agent pairs for launches of >= 4 updates; 2: pairs whenever they fit")
    ap.add_argument("--loop", type=int, default=1, help="multi-update launches keep an agent's updates in one workgroup (ddpg_update_loop_kernel): 1 where it pays (mid nets), 2 always, 0: one workgroup per (update, agent)")
    ap.add_argument("--w8", type=int, default=0, help="eight waves per agent (kernels_ddpg_w8.hip, opt-in: measured slower): 0 never (default), 1 for populations of up to 256 agents per GPU, 2 always")
    ap.add_argument("--resident", type=int, default=0, help="1: small nets keep an agent's state in LDS over a multi-update launch")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / clock reduction only (gloo, no GPU, no library)")
    ap.add_argument("--updates-per-launch", type=int, default=8,
                    help="consecutive updates per agent fused into one launch (ddrl_pop_ddpg_update_n): the reference's "
                         "fupd loop, 1 + force_more_update = 8 in its ddpg.ini; bit-identical to separate launches")
    args = ap.parse_args()
    global PAIR_MODE, RESIDENT, LOOP, W8
    PAIR_MODE = args.pair
    RESIDENT = args.resident
    LOOP = args.loop
    W8 = args.w8

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)        # nothing above has touched the GPU (or imported torch)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: reporting the world that ran\n" % (args.gpus, world))
    if args.workload == "nfac" and not args.dry_run:
        if args.steps == 200 and args.warmup == 24:      # the DDPG defaults: an NFAC step is 220 full-batch Adam iterations
            args.steps, args.warmup = 10, 2
        return nfac_main(args, rank, local_rank, world)
    S, A, B = 18, 6, 64
    ha, hq = SIZES[args.size]
    ewc = not args.no_ewc
    G = args.agents

    # CPU baseline first: before anything initialises the GPU in this process
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not args.dry_run:
        ncpu = os.cpu_count() or 1
        threads = args.cpu_threads or min(16, ncpu)      # the GPU box gives one GPU's share: 16 cores
        ups, n, dt = cpu_baseline(S, A, B, ha, hq, ewc, args.cpu_seconds, threads, args.replay)
        ups1, n1, dt1 = cpu_baseline(S, A, B, ha, hq, ewc, args.cpu_seconds / 3, 1, args.replay)
        cpu = {"value": ups, "unit": "updates/s", "cores": threads, "kind": "port",
               "single_core": ups1, "host_cpu_count": ncpu,
               "sample": "%d updates of the same per-agent workload (%s nets, B=64, EWC %s, replay %d) by the fp64 oracle, "
                         "one agent per thread on %d threads, %.1f s; single core: %d updates in %.1f s"
                         % (n, args.size, "on" if ewc else "off", args.replay, threads, dt, n1, dt1)}

    import torch
    import torch.distributed as dist
    if args.dry_run:
        if os.environ.get("DDRL_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies before the rendezvous
            sys.exit(3)
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_dt": float(t.item()),
                              "launched_by": "bench.py" if os.environ.get("DDRL_BENCH_CHILD") else "torch.distributed.run"}))
        return
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier(pop):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pop.sync()

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return dt

    upl = max(1, min(args.updates_per_launch, 64))
    if max(ha + hq) > 128:
        upl = 1                      # the layer-by-layer path launches per update anyway

    # ---- the weak-scaling sweep: --agents per GPU
    pop = build_population(G, rank, rank * G, S, A, B, ha, hq, ewc, args.replay, local_rank)
    dt, kernel_ms, launches = timed_sweep(pop, G, args.steps, args.warmup, upl, ewc, barrier)
    dt = max_over_ranks(dt)
    del pop

    # ---- N > 1: the metric's own configuration, 1024 agents over the node (1024 / N per GPU)
    strong = None
    if world > 1 and not args.no_strong:
        from ddrl_amd.shard import block_range
        lo, hi = block_range(args.total_agents, rank, world)
        pop = build_population(hi - lo, rank, lo, S, A, B, ha, hq, ewc, args.replay, local_rank)
        sdt, skms, sl = timed_sweep(pop, hi - lo, args.steps, args.warmup, upl, ewc, barrier)
        sdt = max_over_ranks(sdt)
        strong = {"scaling": "strong", "agents_total": args.total_agents, "agents_per_gpu": hi - lo,
                  "value": args.total_agents * args.steps / sdt, "unit": "updates/s", "ms_per_step": sdt / args.steps * 1e3,
                  "kernel_avg_ms_rank0": skms / max(sl, 1)}
        del pop

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    byts, flops, Na, Nc = algorithmic_work(S, A, B, ha, hq, ewc)
    value = world * G * args.steps / dt
    kavg = kernel_ms * 1e-3 / max(launches, 1)          # per launch, as rocprofv3 reports it
    upl_eff = args.steps / max(launches, 1)             # updates per launch actually run
    achieved = byts * G * upl_eff / kavg
    traffic, traffic_source = canned_traffic(G, args.size, upl_eff)
    out = {
        "metric": "actor+critic minibatch updates/sec (node), 1024-agent HalfCheetah DDPG sweep",
        "value": value, "unit": "updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 5: DDPG%s + inverting gradients, HalfCheetah dims S=18 A=6, %s nets a=%s q=%s, B=64, "
                               "%d agents per GPU, replay %d/agent, device mt19937 replay sampling; %d consecutive updates per launch "
                               "(the fupd loop, force_more_update=%d)"
                               % ("+EWC" if ewc else "", args.size, "x".join(map(str, ha)), "x".join(map(str, hq)), G, args.replay, upl, upl - 1),
                   "agents_per_gpu": G, "agents_total": world * G, "parallelism": "agents sharded, no collective",
                   "setup_updates": max(upl, SETUP_UPDATES // upl * upl)},
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": ("ddpg_update_pair_kernel" if (PAIR_MODE == 2 or (upl >= 4 and PAIR_MODE and args.size == "mid" and G > 512)) else
                                "ddpg_update_loop8_kernel" if (args.size == "mid" and (W8 == 2 or (W8 == 1 and LOOP > 0 and G <= 256)) and PAIR_MODE != 2) else
                                "ddpg_update_loop_kernel" if (upl >= 2 and ((LOOP == 1 and args.size == "mid") or LOOP == 2)) else "ddpg_update_kernel") if max(ha + hq) <= 128 else "wide_* layer kernels, optimiser inside the dW kernels (one update = 45 launches)",
                     "kernel_avg_ms": kavg * 1e3, "launches": launches,
                     "updates_per_launch": upl_eff, "algorithmic_bytes_per_launch": byts * G * upl_eff,
                     "algorithmic_bytes_per_agent_update": byts,
                     "mfma_f32_frac": flops * G * upl_eff / kavg / MFMA_F32_PEAK, "flops_per_agent_update": flops},
        "cpu_baseline": cpu,
    }
    if strong is not None:
        out["strong"] = strong
    if world == 1 and not args.no_extra and args.size == "mid" and G == 1024:
        out["extra"] = extra_measurements(args, S, A, B, ewc, upl, barrier)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
