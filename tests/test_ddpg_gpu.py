"""GPU parity: the fused gfx950 DDPG update (through the C ABI) vs the fp64 CPU oracle
(oracle/ddpg_ref.c, restating DeepQNAg.hpp:336-473) on identical seeds and inputs.

Tolerance (north_star): replay indices bit-exact; losses, gradients and weights within 1e-5
relative (fp32 vs the fp64 reference arithmetic), norm-wise (max |diff| / max |ref|)."""
import numpy as np
import pytest

from ddpg_common import make_pair, rel_err, oracle_weights

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check_nets(pop, agents, tol, what=""):
    for net in range(4):
        e = rel_err(pop.weights(net), oracle_weights(agents, net))
        assert e <= tol, "%s net %d rel err %.3g" % (what, net, e)


def test_replay_indices_bit_exact(ora):
    pop, agents = make_pair(ora, G=8, hidden_a=(20,), hidden_q=(50, 7), n_replay=777, replay_cap=1000)
    for it in range(12):    # 12 * 64 draws crosses the 624-word twist boundary
        pop.update()
        got = pop.last()["idx"]
        for g, ag in enumerate(agents):
            _, want = ag.update()
            assert np.array_equal(got[g], want), (it, g)


@pytest.mark.parametrize("hidden_a,hidden_q", [((100, 100), (100, 100)), ((20,), (50, 7))])
def test_one_update_matches_oracle(ora, hidden_a, hidden_q):
    pop, agents = make_pair(ora, G=4, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.1)
    rng = np.random.RandomState(5)
    idx = rng.randint(0, 1000, (4, 64)).astype(np.int32)
    pop.update(idx)
    last = pop.last()
    for g, ag in enumerate(agents):
        loss, used = ag.update(idx[g])
        qt, dact = ag.last()
        assert np.array_equal(used, last["idx"][g])
        assert rel_err(last["q_targets"][g], qt) <= TOL
        assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
        assert rel_err(last["dact"][g], dact) <= TOL
    _check_nets(pop, agents, TOL, "1 update")
    for net in (0, 1):
        m, v, it = pop.adam(net)
        assert np.all(it == 1)
        mo = np.stack([ag.nets[net].adam()[0] for ag in agents]); vo = np.stack([ag.nets[net].adam()[1] for ag in agents])
        assert rel_err(m, mo) <= TOL and rel_err(v, vo) <= 2 * TOL   # v = g^2: twice the relative error of g


@pytest.mark.parametrize("momentum,hidden_act,actor_out,inverting", [(0, 1, 2, True), (1, 2, 2, False), (3, 3, 0, False), (2, 1, 2, True)])
def test_many_updates_track_oracle(ora, momentum, hidden_act, actor_out, inverting):
    pop, agents = make_pair(ora, G=3, hidden_a=(64, 32), hidden_q=(48, 40), momentum=momentum, hidden_act=hidden_act,
                            actor_out_act=actor_out, inverting_grad=inverting, decay_v=1e-3, seed=2)
    for it in range(10):
        pop.update()
        for ag in agents:
            ag.update()
    # the fp32/fp64 gap grows with the number of Adam steps (sign-like updates amplify rounding of
    # near-zero gradients): 1e-5 per update budgeted linearly
    _check_nets(pop, agents, 10 * TOL, "10 updates")


def test_small_batch_and_mask_and_guard(ora):
    pop, agents = make_pair(ora, G=4, B=32, hidden_a=(16,), hidden_q=(16, 16), n_replay=300, replay_cap=300)
    w0 = [pop.weights(n) for n in range(4)]
    mask = np.array([1, 0, 1, 0], np.uint8)
    pop.update(mask=mask)
    for g, ag in enumerate(agents):
        if mask[g]:
            ag.update()
    _check_nets(pop, agents, TOL, "masked")
    for n in range(4):
        assert np.array_equal(pop.weights(n)[1], w0[n][1])       # masked agents untouched
    # guard of DeepQNAg.hpp:338: fewer than 250 samples => no update, no draw consumed
    pop2, agents2 = make_pair(ora, G=2, B=32, hidden_a=(16,), hidden_q=(16, 16), n_replay=249, replay_cap=300)
    w0 = pop2.weights(0)
    pop2.update()
    assert np.array_equal(pop2.weights(0), w0)
    assert np.all(pop2.adam(0)[2] == 0)


def test_replay_ring_append_wraps_like_deque(ora):
    S, A = 5, 2
    pop, agents = make_pair(ora, G=2, S=S, A=A, B=16, hidden_a=(8,), hidden_q=(8,), n_replay=0, replay_cap=260)
    rng = np.random.RandomState(9)
    for t in range(300):     # 40 past capacity: pop_front semantics
        s = rng.uniform(-1, 1, (2, S)).astype(np.float32); a = rng.uniform(-1, 1, (2, A)).astype(np.float32)
        s2 = rng.uniform(-1, 1, (2, S)).astype(np.float32); r = rng.normal(0, 1, 2).astype(np.float32)
        goal = (rng.uniform(0, 1, 2) < 0.1).astype(np.uint8)
        pop.replay_append(s, a, r, s2, goal)
        for g, ag in enumerate(agents):
            ag.insert(s[g].astype(np.float64), a[g].astype(np.float64), float(r[g]), s2[g].astype(np.float64), bool(goal[g]))
    assert list(pop.replay_len()) == [260, 260]
    for _ in range(3):
        pop.update()
        for ag in agents:
            ag.update()
    _check_nets(pop, agents, 3 * TOL, "ring")


def test_infer_matches_oracle_forward(ora):
    pop, agents = make_pair(ora, G=5, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.2)
    rng = np.random.RandomState(4)
    s = rng.uniform(-1, 1, (5, 18)).astype(np.float32)
    got = pop.infer(s)
    for g, ag in enumerate(agents):
        want = ag.ann.forward(s[g].astype(np.float64))[0]
        assert rel_err(got[g], want) <= TOL


def test_weights_roundtrip_reference_order():
    import ddrl_amd
    cfg = ddrl_amd.DdpgConfig(n_agents=3, nb_sensors=7, nb_motors=3, hidden_unit_a=(5, 9), hidden_unit_q=(6,), mini_batch_size=16, replay_memory=300)
    pop = ddrl_amd.DdpgPopulation(cfg)
    assert pop.num_params(0) == 7 * 5 + 5 + 5 * 9 + 9 + 9 * 3 + 3
    assert pop.num_params(1) == 10 * 6 + 6 + 6 + 1
    for net in range(4):
        w = np.random.RandomState(net).normal(0, 1, (3, pop.num_params(net))).astype(np.float32).astype(np.float64)
        pop.set_weights(net, w)
        assert np.array_equal(pop.weights(net), w)
    w = pop.weights(0)
    assert abs(w.std() - 1.0) < 0.1


def test_default_init_is_gaussian_001():
    import ddrl_amd
    pop = ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(n_agents=8), init_seed=11)
    w = pop.weights(1)
    assert abs(w.std() - 0.01) < 5e-4 and abs(w.mean()) < 5e-4       # MLP.hpp:936-944
    assert np.array_equal(pop.weights(1), pop.weights(3)) and np.array_equal(pop.weights(0), pop.weights(2))


def test_ewc_lifecycle_matches_oracle(ora):
    kw = dict(G=3, hidden_a=(32, 32), hidden_q=(32, 32), ewc=100.0, ewc_decay=0.9, ewc_fisher_beta=0.999,
              ewc_for_critic=1, ewc_best_method=0, wscale=0.1, seed=3)
    pop, agents = make_pair(ora, **kw)
    for it in range(4):                       # task 1: accumulate Fisher
        pop.update()
        for ag in agents:
            ag.update()
    scores = np.array([1.0, 2.0, 3.0])
    pop.update_best_param(scores)
    for g, ag in enumerate(agents):
        ag.ann.update_best_param(scores[g]); ag.qnn.update_best_param(scores[g])
    for it in range(2):
        pop.update()
        for ag in agents:
            ag.update()
    pop.ewc_setup()                           # development switch
    for ag in agents:
        ag.ewc_setup()
    for net in (0, 1):
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            so = ag.nets[net].ewc_state()
            assert so["active"] == 1
            assert rel_err(st["fisher_hat"][g], so["fisher_hat"]) <= 4 * TOL
            assert rel_err(st["wstar"][g], so["wstar"]) <= 4 * TOL
    _check_nets(pop, agents, 6 * TOL, "after ewc_setup")
    for it in range(3):                       # task 2: penalty live + Fisher restarts
        pop.update()
        for ag in agents:
            ag.update()
        pop.ewc_decay_update()
        for ag in agents:
            import oracle.oracle as o
            o.lib().ora_mlp_ewc_decay_update(ag.ann.h); o.lib().ora_mlp_ewc_decay_update(ag.qnn.h)
    _check_nets(pop, agents, 10 * TOL, "ewc task 2")
    for net in (0, 1):
        cost = pop.ewc_cost(net)
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            assert abs(cost[g] - ag.nets[net].ewc_cost()) <= 1e-4 * abs(ag.nets[net].ewc_cost()) + 1e-12
            so = ag.nets[net].ewc_state()
            assert rel_err(st["fisher"][g], so["fisher"]) <= 1e-4
            assert abs(st["fisher_nbr"][g] - so["fisher_nbr"]) <= 1e-5 * so["fisher_nbr"]
