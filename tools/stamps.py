"""Dev tool (GPU box): per-phase cycle breakdown of the fused DDPG update (s_memtime stamps)."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import ddrl_amd
from ddrl_amd import lib as _l
from bench import synth_rows
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ewc = 100.0 if (len(sys.argv) < 3 or sys.argv[2] == "ewc") else -1.0
cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=2000, ewc=ewc, ewc_for_critic=1)
pop = ddrl_amd.DdpgPopulation(cfg, init_seed=1)
if os.environ.get("DDRL_CHUNKED"): pop.set_option("force_chunked", 1)
rows = synth_rows(0, 2000, 18, 6)
for g in range(G):
    pop.replay_load(g, rows)
pop.update()
if ewc >= 0:
    pop.update_best_param(np.ones(G)); pop.ewc_setup()
for _ in range(3):
    pop.update()
L = _l.load()
_l.check(L.ddrl_pop_debug_stamps(pop.h, 1, None))
pop.update(); pop.sync()
out = np.zeros((G, 16), np.int64)
_l.check(L.ddrl_pop_debug_stamps(pop.h, 0, out.ctypes.data_as(C.c_void_p)))
d = np.diff(out[:, :11], axis=1)
names = ["gather+actor_t fwd", "critic_t fwd", "gather2+targets", "critic fwd", "critic bwd+adam", "actor fwd", "critic fwd2", "critic bwd dx", "actor hidden recompute", "actor bwd+adam"]
tot = (out[:, 10] - out[:, 0])
print("G=%d ewc=%s  per-WG total cycles median %d (%.1f us @2.1GHz?)" % (G, ewc >= 0, np.median(tot), np.median(tot) / 2100.0))
for i, n in enumerate(names):
    print("%-26s median %8d cycles  %5.1f %%" % (n, np.median(d[:, i]), 100 * np.median(d[:, i]) / np.median(tot)))
pop.set_profiling(True); pop.profile_read()
NU = int(os.environ.get("DDRL_NUPD", "1"))
for _ in range(20):
    pop.update_n(NU) if NU > 1 else pop.update()
ms, n = pop.profile_read()
print("kernel avg %.1f us per update (%d updates per launch, %d launches) -> %.0f updates/s" % (ms / n / NU * 1e3, NU, n, G * n * NU / ms * 1e3))
bw = out[:, 11:15]
print("critic bwd: dX3 %d | dW3+opt %d | dZ2 write+dX2 %d | dW2+opt %d | dZ1 write + dW1+opt %d" % tuple(np.median(x) for x in (bw[:, 0] - out[:, 4], bw[:, 1] - bw[:, 0], bw[:, 2] - bw[:, 1], bw[:, 3] - bw[:, 2], out[:, 5] - bw[:, 3])))
for nm, sl in (("round 1 (agents < G/2)", slice(0, G // 2)), ("round 2", slice(G // 2, G))):
    o = out[sl]
    print(nm, "total %d | dW2+opt critic %d | fwd critic_t %d | start spread (p95-p5) %d" % (np.median(o[:, 10] - o[:, 0]), np.median(o[:, 14] - o[:, 13]), np.median(o[:, 2] - o[:, 1]), np.percentile(o[:, 0], 95) - np.percentile(o[:, 0], 5)))
print("span first start -> last end: %d cycles" % (out[:, 10].max() - out[:, 0].min()))
