"""Runs in its own process with DDRL_POP_SO pointing at the fp64 debug library (tests/test_f64_debug_gpu.py): a DDPG
population set up from the .npz the parent wrote, n free-running updates, final state back as .npz."""
import sys

import numpy as np

import ddrl_amd
from ddrl_amd import lib as _l


def main():
    src, dst = sys.argv[1], sys.argv[2]
    d = np.load(src)
    G, n = int(d["G"]), int(d["n"])
    kw = dict(n_agents=G, nb_sensors=int(d["S"]), nb_motors=int(d["A"]), mini_batch_size=int(d["B"]),
              replay_memory=int(d["cap"]), hidden_unit_a=tuple(d["ha"]), hidden_unit_q=tuple(d["hq"]), inverting_grad=1,
              momentum=int(d["momentum"]))
    if float(d["ewc"]) >= 0:
        kw.update(ewc=float(d["ewc"]), ewc_decay=-1.0, ewc_fisher_beta=0.999, ewc_for_critic=1)
    pop = ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(**kw), init_seed=1)
    assert _l.REAL is np.float64 and b"+f64" in _l.load().ddrl_version()
    for key in ("alpha_a", "alpha_v", "tau_soft_update", "gamma"):
        pop.set_hyper(key, d[key])
    for net in range(4):
        pop.set_weights(net, d["w%d" % net])
    for g in range(G):
        pop.replay_load(g, d["rows"][g])
    idx = d["idx"]                                    # [n, G, B]
    losses = np.zeros((n, G))
    for u in range(n):
        pop.update(idx=idx[u])
        if u == 0 and float(d["ewc"]) >= 0:
            pop.update_best_param(np.ones(G)); pop.ewc_setup()
        losses[u] = pop.last()["loss"]
    out = {"w%d" % net: pop.weights(net) for net in range(4)}
    out["loss"] = losses
    for net in (0, 1):
        m, v, it = pop.adam(net)
        out["m%d" % net], out["v%d" % net], out["it%d" % net] = m, v, it
    np.savez(dst, **out)


if __name__ == "__main__":
    main()
