#!/usr/bin/env python3
"""Dev tool (GPU box): update rate of the BatchNorm path (kernels_ddpg_bn.hip).  Default: the nets and tower layouts of
agent/deepqn/data/best-ddpg-halfcheetah.ini (hidden_unit_q=20:5:2, hidden_unit_a=5:5:2, batch_norm_actor=14, batch_norm_critic=11);
`mid`: 100x100 nets with the same layouts.  usage: bench_bn.py [agents] [best|mid] [updates]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddrl_amd
from bench import synth_rows

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
size = sys.argv[2] if len(sys.argv) > 2 else "best"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ha, hq = ((5, 5, 2), (20, 5, 2)) if size == "best" else ((100, 100), (100, 100))
cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=2000, inverting_grad=0, hidden_unit_a=ha, hidden_unit_q=hq, actor_output_layer_type=0,
                          batch_norm_actor=14, batch_norm_critic=11, alpha_a=1e-3, alpha_v=1e-3)
pop = ddrl_amd.DdpgPopulation(cfg, init_seed=1)
for g in range(G):
    pop.replay_load(g, synth_rows(g % 16, 2000, 18, 6))
for _ in range(5):
    pop.update()
pop.sync()
pop.set_profiling(True); pop.profile_read()
t0 = time.perf_counter()
for _ in range(K):
    pop.update()
pop.sync()
dt = time.perf_counter() - t0
ms, n = pop.profile_read()
print(json.dumps({"workload": "DDPG with BatchNorm towers (actor 14, critic 11), %s nets a=%s q=%s, B=64, %d agents" % (size, ha, hq, G),
                  "updates_per_s": G * K / dt, "kernel_avg_us": ms / n * 1e3, "ms_per_population_update": dt / K * 1e3}))
