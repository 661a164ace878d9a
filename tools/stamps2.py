"""Dev tool (GPU box): per-quarter cycle breakdown of the fused DDPG update inside a MULTI-update launch
(s_memtime stamps of every agent's last update of the launch), one-agent kernel or agent pairs.
usage: python tools/stamps2.py [G] [pair 0|1|2] [nupd]"""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import ddrl_amd
from ddrl_amd import lib as _l
from bench import synth_rows
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pair = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NU = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=2000, ewc=100.0, ewc_for_critic=1, inverting_grad=1)
pop = ddrl_amd.DdpgPopulation(cfg, init_seed=1)
pop.set_option("pair", pair)
if len(sys.argv) > 4:
    pop.set_option("w8", int(sys.argv[4]))
rows = synth_rows(0, 2000, 18, 6)
for g in range(G):
    pop.replay_load(g, rows)
pop.update()
pop.update_best_param(np.ones(G)); pop.ewc_setup()
for _ in range(3):
    pop.update_n(NU) if NU > 1 else pop.update()
L = _l.load()
_l.check(L.ddrl_pop_debug_stamps(pop.h, 1, None))
pop.update_n(NU) if NU > 1 else pop.update()
pop.sync()
out = np.zeros((G, 16), np.int64)
_l.check(L.ddrl_pop_debug_stamps(pop.h, 0, out.ctypes.data_as(C.c_void_p)))
q = np.stack([out[:, 4] - out[:, 0], out[:, 5] - out[:, 4], out[:, 9] - out[:, 5], out[:, 10] - out[:, 9]], axis=1)
tot = out[:, 10] - out[:, 0]
print("G=%d pair=%d nupd=%d: last update of the launch, cycles per quarter (median | p10 | p90)" % (G, pair, NU))
for i, n in enumerate(["q0 targets+critic fwd", "q1 critic bwd+opt", "q2 actor/critic fwd, dQ/da, recompute", "q3 actor bwd+opt"]):
    print("  %-40s %8d | %8d | %8d   %5.1f %%" % (n, np.median(q[:, i]), np.percentile(q[:, i], 10), np.percentile(q[:, i], 90), 100 * np.median(q[:, i]) / np.median(tot)))
print("  total %d" % np.median(tot))
d = np.diff(out[:, :11], axis=1)
names = ["gather+actor_t fwd", "critic_t fwd", "gather2+targets", "critic fwd", "critic bwd+adam", "actor fwd", "critic fwd2", "critic bwd dx", "actor hidden recompute", "actor bwd+adam"]
for i, n in enumerate(names):
    print("    %-26s median %8d" % (n, np.median(d[:, i])))
bw = out[:, 11:15]
print("    critic bwd detail: dX3 %d | dW3+opt %d | dZ2 write+dX2 %d | dW2+opt %d | dZ1 write + dW1+opt %d" % tuple(np.median(x) for x in (bw[:, 0] - out[:, 4], bw[:, 1] - bw[:, 0], bw[:, 2] - bw[:, 1], bw[:, 3] - bw[:, 2], out[:, 5] - bw[:, 3])))
pop.set_profiling(True); pop.profile_read()
for _ in range(10):
    pop.update_n(NU) if NU > 1 else pop.update()
ms, n = pop.profile_read()
print("kernel avg %.1f us per update (%d updates per launch, %d launches) -> %.0f updates/s" % (ms / n / NU * 1e3, NU, n, G * n * NU / ms * 1e3))
