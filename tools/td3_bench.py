"""Dev tool (GPU box): TD3 update timing on the bench workload (1024 agents, mid nets, B=64)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import ddrl_amd
from bench import synth_rows, algorithmic_work
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ewc = len(sys.argv) < 3 or sys.argv[2] == "ewc"
hidden = tuple(int(x) for x in sys.argv[3].split("x")) if len(sys.argv) > 3 else (100, 100)     # e.g. 400x300
S, A = 18, 6
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=2000, inverting_grad=0, td3=1, policy_freq=2, mini_batch_size=B,
                          hidden_unit_a=hidden, hidden_unit_q=hidden, ewc=100.0 if ewc else -1.0, ewc_for_critic=1)
pop = ddrl_amd.Td3Population(cfg, init_seed=1)
rows = synth_rows(0, 2000, S, A)
for g in range(G):
    pop.replay_load(g, rows)
pop.update()
if ewc:
    pop.update_best_param(np.ones(G)); pop.ewc_setup()
byts, flops, Na, Nc = algorithmic_work(S, A, B, hidden, hidden, ewc)
print("TD3, %d agents, nets %s, B=%d, EWC %s" % (G, "x".join(map(str, hidden)), B, "on" if ewc else "off"))
per = 36 + (16 if ewc else 0)
for name, astep, nbytes in (("policy step", 1, per * (Na + 2 * Nc)), ("critic-only step", 0, per * 2 * Nc), ("alternating (policy_freq=2)", None, per * (Na / 2 + 2 * Nc))):
    for _ in range(4):
        pop.update(actor_step=astep if astep is not None else 1)
    pop.set_profiling(True); pop.profile_read()
    for i in range(20):
        pop.update(actor_step=(i % 2 == 0) if astep is None else astep)
    ms, n = pop.profile_read()
    us = ms / n * 1e3
    print("%-28s kernel avg %.1f us -> %.2f M updates/s, %.2f TB/s of algorithmic state traffic (%.0f B/agent-update)"
          % (name, us, G / us, nbytes * G / us / 1e6, nbytes))
