"""Developmental gates (DODevMLP.hpp:376-400, semantics SURVEY.md 8h) on the device: per-dimension
multiplicative gates on state inputs, actor outputs and the critic's action inputs, shared by an
agent's nets.  Oracle: oracle/mlp_ref.c with set_gates on every net.  The fork's Developmental layer
source is absent from the tree, so this is parity against the restatement of its call sites and
tests only (parity unpinned); binary gates (probabilist=1 / heuristic switches) and a fractional
gate are both covered."""
import numpy as np
import pytest

from ddpg_common import check_step, make_pair, rel_err
import test_nfac_gpu as nf

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _gates(rng, G, S, A, binary):
    if binary:
        sg = (rng.uniform(0, 1, (G, S)) < 0.6).astype(np.float64)
        ag = (rng.uniform(0, 1, (G, A)) < 0.6).astype(np.float64)
        ag[:, 0] = 1.0
    else:
        sg = np.round(rng.uniform(0.2, 1, (G, S)), 2).astype(np.float32).astype(np.float64)
        ag = np.round(rng.uniform(0.2, 1, (G, A)), 2).astype(np.float32).astype(np.float64)
    return sg, ag


@pytest.mark.parametrize("binary,generic", [(True, False), (False, False), (True, True)])
def test_ddpg_update_with_gates(ora, binary, generic):
    G, S, A = 3, 18, 6
    pop, agents = make_pair(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, seed=6, force_generic=generic)
    sg, ag = _gates(np.random.RandomState(1), G, S, A, binary)
    pop.set_gates(sg, ag)
    for g, a in enumerate(agents):
        a.ann.set_gates(sg[g], ag[g]); a.ann_target.set_gates(sg[g], ag[g])
        cin = np.concatenate([sg[g], ag[g]])
        a.qnn.set_gates(cin, None); a.qnn_target.set_gates(cin, None)
    # inference sees the same gates (ann_testing shares them)
    st = np.random.RandomState(2).uniform(-1, 1, (G, S)).astype(np.float32)
    act = pop.infer(st)
    for g, a in enumerate(agents):
        assert rel_err(act[g], a.ann.forward(st[g].astype(np.float64))[0]) <= TOL
        assert np.all(act[g][ag[g] == 0] == 0)
    ws_prev = [pop.weights(n) for n in range(4)]
    idx = np.random.RandomState(5).randint(0, 1000, (G, 64)).astype(np.int32)
    pop.update(idx)
    last = pop.last()
    for g, a in enumerate(agents):
        loss, _ = a.update(idx[g])
        qt, dact = a.last()
        assert rel_err(last["q_targets"][g], qt) <= TOL
        assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
        assert rel_err(last["dact"][g], dact) <= TOL
    hyper = (np.array([a.cfg.alpha_a for a in agents]), np.array([a.cfg.alpha_v for a in agents]))
    check_step(pop, agents, ws_prev, hyper, 0, TOL)
    # gates off again: back to the ungated result for inference
    pop.set_gates(None, None)
    for a in agents:
        a.ann.set_gates(None, None)
    act = pop.infer(st)
    for g, a in enumerate(agents):
        assert rel_err(act[g], a.ann.forward(st[g].astype(np.float64))[0]) <= TOL


@pytest.mark.parametrize("binary", [True, False])
def test_nfac_update_with_gates(ora, binary):
    S, A, G, maxT = 18, 6, 3, 200
    pop, agents, lrs = nf.make_pair(ora, G, S, A, (32, 32), (32, 32), maxT, seed=4, update_critic_first=0, lambda_=0.9, gae=1,
                                    number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1)
    sg, ag = _gates(np.random.RandomState(3), G, S, A, binary)
    pop.set_gates(sg, ag)
    for g, a in enumerate(agents):
        a.ann.set_gates(sg[g], ag[g])
        a.vnn.set_gates(sg[g], None)
    rng = np.random.RandomState(8)
    for upd in range(2):
        lens = [200, 77, 130]
        trajs = nf.load_trajs(pop, rng, lens, S, A)
        ws_prev, ad_prev = nf.sync_oracle(pop, agents)
        it_prev = [ad_prev[0][2].copy(), ad_prev[1][2].copy()]
        pop.update()
        last = pop.last()
        ns = []
        for g, a in enumerate(agents):
            n, deltas = nf.oracle_update(a, *trajs[g], S, A)
            ns.append(n)
            assert int(last["n_pos"][g]) == n
            assert rel_err(last["deltas"][g, :lens[g]], deltas) <= TOL
        nf.check_nets(pop, agents, ws_prev, it_prev, lrs, 0, [[n > 0 for n in ns], [True] * G])
