"""The fp64 debug instantiation of the fused DDPG update (ddrl_amd/libddrl_pop_f64.so, built by `make f64`
from the SAME sources with the arithmetic type switched, csrc/pop_internal.h) separates rounding from algorithm:

  * device-fp64 vs the fp64 oracle, 100 FREE-RUNNING updates (no resynchronisation), EWC live, the judged mid-size nets:
    weights, targets, Adam state and every update's loss agree to 1e-9 -- the device kernel computes the oracle's algorithm,
    update after update, with nothing but summation order between them;
  * device-fp32 (the product) vs device-fp64 over the same 100 updates: reported, and held to the bound that rounding alone
    explains (Adam at beta1 = 0 takes sign-like steps, so single weights may sit a step apart: tests/ddpg_common.py).
The debug library runs in its own process (its ABI's `float` is `double`); data travels as .npz."""
import os
import subprocess
import sys

import numpy as np
import pytest

from ddpg_common import close_frac, make_pair, oracle_weights, rel_err, synth_transitions

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64 = os.path.join(ROOT, "ddrl_amd", "libddrl_pop_f64.so")


@pytest.mark.parametrize("momentum", [0, 1])
def test_hundred_free_running_updates_fp64_device_vs_oracle(ora, tmp_path, momentum):
    assert os.path.exists(F64), "build the fp64 debug library: make -C ddrl_amd/csrc f64"
    G, n, S, A, B, cap = 3, 100, 18, 6, 64, 2000
    ha = hq = (100, 100)
    pop, agents = make_pair(ora, G=G, S=S, A=A, B=B, hidden_a=ha, hidden_q=hq, n_replay=1000, replay_cap=cap, wscale=0.1,
                            seed=21, momentum=momentum, sweep=True, ewc=100.0, ewc_for_critic=1)
    rng = np.random.RandomState(5)
    idx = rng.randint(0, 1000, (n, G, B)).astype(np.int32)
    rows = np.stack([synth_transitions(np.random.RandomState(1000 + g + 97 * 21), 1000, S, A) for g in range(G)])   # make_pair's
    # the hyper-parameters as the device holds them (fp32), so the three runs start from identical numbers
    hyper = {k: np.array([getattr(ag.cfg, f) for ag in agents], np.float64)
             for k, f in (("alpha_a", "alpha_a"), ("alpha_v", "alpha_v"), ("tau_soft_update", "tau"), ("gamma", "gamma"))}
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(src, G=G, n=n, S=S, A=A, B=B, cap=cap, ha=np.array(ha), hq=np.array(hq), momentum=momentum, ewc=100.0, idx=idx, rows=rows,
             **hyper, **{"w%d" % net: pop.weights(net) for net in range(4)})
    env = dict(os.environ, DDRL_POP_SO=F64, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "f64_worker.py"), src, dst], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d64 = np.load(dst)
    # the oracle and the fp32 product over the same updates
    loss_o, loss_32 = np.zeros((n, G)), np.zeros((n, G))
    for u in range(n):
        pop.update(idx=idx[u])
        loss_32[u] = pop.last()["loss"]
        for g, ag in enumerate(agents):
            loss_o[u, g], _ = ag.update(idx[u, g])
        if u == 0:
            pop.update_best_param(np.ones(G)); pop.ewc_setup()
            for ag in agents:
                ag.ann.update_best_param(1.0); ag.qnn.update_best_param(1.0); ag.ewc_setup()
    # ---- device fp64 == oracle: the algorithm ----
    assert rel_err(d64["loss"], loss_o) <= 1e-9, rel_err(d64["loss"], loss_o)
    worst = 0.0
    for net in range(4):
        e = rel_err(d64["w%d" % net], oracle_weights(agents, net))
        worst = max(worst, e)
        assert e <= 1e-9, (net, e)
    for net in (0, 1):
        for g, ag in enumerate(agents):
            mo, vo, ito = ag.nets[net].adam()
            assert rel_err(d64["m%d" % net][g], mo) <= 1e-9 and rel_err(d64["v%d" % net][g], vo) <= 1e-9
            assert int(d64["it%d" % net][g]) == ito == n
    # ---- device fp32 vs device fp64: rounding ----
    rep = {}
    for net in range(4):
        w32, w64 = pop.weights(net), d64["w%d" % net]
        rep[net] = (rel_err(w32, w64), float(np.mean(np.abs(w32 - w64) <= 1e-4 * np.max(np.abs(w64)))))
        for g in range(G):
            assert close_frac(w32[g], w64[g], 1e-4, 0.99, 2e-3), (net, g, rep[net])   # measured: all within 1e-4, worst 3.5e-5
    print("fp64 device vs oracle: worst norm-wise error %.2e; fp32 vs fp64 after %d free-running updates (momentum %d): "
          "%s  [(norm-wise max error, share of weights within 1e-4) per net]" % (worst, n, momentum, rep))
    assert rel_err(loss_32[:5], d64["loss"][:5]) <= 1e-4           # early updates: before any sign-like step can have flipped
