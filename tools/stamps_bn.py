"""Dev tool (GPU box): per-phase cycle breakdown of the BatchNorm update kernel (clock64 stamps of workgroup thread 0).
usage: stamps_bn.py [agents] [best|mid]"""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import ddrl_amd
from ddrl_amd import lib as _l
from bench import synth_rows
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
size = sys.argv[2] if len(sys.argv) > 2 else "mid"
ha, hq = ((5, 5, 2), (20, 5, 2)) if size == "best" else ((100, 100), (100, 100))
cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=2000, inverting_grad=0, hidden_unit_a=ha, hidden_unit_q=hq, actor_output_layer_type=0,
                          batch_norm_actor=14, batch_norm_critic=11, alpha_a=1e-3, alpha_v=1e-3)
pop = ddrl_amd.DdpgPopulation(cfg, init_seed=1)
rows = synth_rows(0, 2000, 18, 6)
for g in range(G):
    pop.replay_load(g, rows)
for _ in range(3):
    pop.update()
L = _l.load()
_l.check(L.ddrl_pop_debug_stamps(pop.h, 1, None))
pop.update(); pop.sync()
out = np.zeros((G, 16), np.int64)
_l.check(L.ddrl_pop_debug_stamps(pop.h, 0, out.ctypes.data_as(C.c_void_p)))
names = ["gather + actor_t fwd", "critic_t fwd", "gather + targets", "critic fwd", "critic bwd (dW stored)", "critic apply sweep", "gather + actor fwd",
         "critic fwd 2", "critic bwd dX", "actor hidden recompute", "actor bwd (dW stored)", "actor apply sweep", "soft statistics"]
d = np.diff(out[:, :14], axis=1)
tot = out[:, 13] - out[:, 0]
print("G=%d %s nets: per-workgroup total %d ticks (clock64) median" % (G, size, np.median(tot)))
for i, n in enumerate(names):
    print("%-26s median %9d  %5.1f %%" % (n, np.median(d[:, i]), 100 * np.median(d[:, i]) / np.median(tot)))
