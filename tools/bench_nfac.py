#!/usr/bin/env python3
"""SURVEY.md 8(d) config 3 on one GPU: NFAC(lambda)-V, S=45, A=17, G=256 agents, 64x64 nets, T=500-step
trajectories, lambda=0.9, gae, 10 fitted iterations x 10 critic steps + 10 actor steps per update, EWC on
both nets (penalty live).  Metric: full-batch Adam iterations / s (all agents)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import ddrl_amd

G, S, A, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 45, 17, 500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = ddrl_amd.NfacConfig(n_agents=G, nb_sensors=S, nb_motors=A, hidden_unit_a=(64, 64), hidden_unit_v=(64, 64),
                          max_trajectory=512, ewc=100.0, ewc_fisher_beta=0.999, ewc_for_critic=1, ewc_best_param_method=1)
cfg.set("lambda", 0.9); cfg.set("gae", 1)
pop = ddrl_amd.NfacPopulation(cfg, init_seed=1)
if len(sys.argv) > 3 and sys.argv[3] == "generic":
    pop.set_option("force_generic", 1)
rng = np.random.RandomState(0)


def load():
    for g in range(G):
        s = np.clip(np.cumsum(rng.normal(0, 0.1, (T + 1, S)), axis=0), -1, 1)
        pa = np.tanh(rng.normal(0, 0.7, (T, A))); a = np.clip(pa + rng.normal(0, 0.2, (T, A)), -1, 1)
        rows = np.concatenate([s[:-1], pa, a, s[1:], rng.normal(0, 1, (T, 1)), np.zeros((T, 1))], axis=1).astype(np.float32)
        pop.traj_load(g, rows, [125, 250, 375, 500])


load(); pop.update()
pop.update_best_param(np.ones(G)); pop.ewc_setup()
times = []
for _ in range(steps):
    load()
    pop.sync()
    t0 = time.perf_counter()
    pop.update()
    pop.sync()
    times.append(time.perf_counter() - t0)
dt = float(np.median(times))
iters = G * (10 * 10 + 10)
na, nv = pop.num_params(0), pop.num_params(1)
flops = G * 6.0 * T * (100 * (nv - 64 - 64 - 1) + 10 * (na - 64 - 64 - 17))      # fwd + dX + dW, 2 flop / MAC, weights only
print(json.dumps({"workload": "config 3: NFAC(lambda)-V G=%d S=45 A=17 64x64 T=500 EWC on" % G, "ms_per_update": dt * 1e3,
                  "adam_iterations_per_s": iters / dt, "agent_updates_per_s": G / dt, "approx_tflops": flops / dt / 1e12}))
