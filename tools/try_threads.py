"""Two handles on one device driven from two host threads (distinct handles may live on distinct threads): wide
updates (two-stream split, shared per-device auxiliary streams) and one-tile updates, against the same populations run one
after the other.  Prints OK or the first difference."""
import sys, threading
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import ddrl_amd
from ddpg_common import synth_transitions


def build(ha, hq, G, seed):
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=18, nb_motors=6, mini_batch_size=64, replay_memory=600, hidden_unit_a=ha, hidden_unit_q=hq)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=seed)
    rows = synth_transitions(np.random.RandomState(seed), 500, 18, 6)
    for g in range(G):
        pop.replay_load(g, rows)
    return pop


def run(pop, n, out, key):
    for u in range(n):
        idx = np.random.RandomState(100 + u).randint(0, 500, (pop.G if hasattr(pop, "G") else pop.cfg.n_agents, 64)).astype(np.int32)
        pop.update(idx)
    out[key] = [pop.weights(k) for k in range(4)]


bad = 0
for name, ha, hq, G in (("wide", (400, 300), (400, 300), 130), ("tile", (100, 100), (100, 100), 256)):
    ref = {}
    for k in (0, 1):
        p = build(ha, hq, G, 7 + k)
        run(p, 6, ref, k)
        p.close()
    got = {}
    pops = [build(ha, hq, G, 7 + k) for k in (0, 1)]
    ths = [threading.Thread(target=run, args=(pops[k], 6, got, k)) for k in (0, 1)]
    for t in ths: t.start()
    for t in ths: t.join()
    for p in pops: p.close()
    for k in (0, 1):
        for a, b in zip(ref[k], got[k]):
            if not np.array_equal(a, b):
                bad += 1
                print(name, "handle", k, "differs: max", np.max(np.abs(a - b)))
                break
print("OK" if not bad else "DIFFERENT")
