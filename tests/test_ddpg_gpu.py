"""GPU parity: the fused gfx950 DDPG update (through the C ABI) vs the fp64 CPU oracle
(oracle/ddpg_ref.c, restating DeepQNAg.hpp:336-473) on identical seeds and inputs.

Tolerance (north_star): replay indices bit-exact; losses, gradients and weights within 1e-5
relative (fp32 vs the fp64 reference arithmetic), norm-wise (max |diff| / max |ref|)."""
import numpy as np
import pytest

from ddpg_common import close_frac, make_pair, rel_err, oracle_weights, sync_oracle_from_gpu, check_step

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check_nets(pop, agents, tol, what="", frac=1.0):
    """direct weight comparison: fraction `frac` of each net's parameters within tol (norm-wise)."""
    for net in range(4):
        w, wo = pop.weights(net), oracle_weights(agents, net)
        for g in range(len(agents)):
            wmax = np.max(np.abs(wo[g]))
            ok = np.mean(np.abs(w[g] - wo[g]) <= tol * wmax)
            assert ok >= frac, "%s net %d agent %d: %.5f of the parameters within %.1e" % (what, net, g, ok, tol)


def _hyper(agents):
    return (np.array([a.cfg.alpha_a for a in agents]), np.array([a.cfg.alpha_v for a in agents]))


def test_replay_indices_bit_exact(ora):
    pop, agents = make_pair(ora, G=8, hidden_a=(20,), hidden_q=(50, 7), n_replay=777, replay_cap=1000)
    for it in range(12):    # 12 * 64 draws crosses the 624-word twist boundary
        pop.update()
        got = pop.last()["idx"]
        for g, ag in enumerate(agents):
            _, want = ag.update()
            assert np.array_equal(got[g], want), (it, g)


# generic=False: the compile-time-shape instantiation is used where one matches ("mid", "small");
# generic=True forces the run-time-shape kernel on the same inputs.
@pytest.mark.parametrize("hidden_a,hidden_q,wscale,sweep,generic", [
    ((100, 100), (100, 100), 0.01, False, False),      # BASELINE.md section 3 "mid", reference init scale
    ((100, 100), (100, 100), 0.1, True, False),        # hyper-parameter sweep
    ((100, 100), (100, 100), 0.1, True, True),
    ((20,), (50, 7), 0.1, True, False),                # "small": agent/deepqn/data/ddpg-cp.ini:15-16
    ((20,), (50, 7), 0.1, True, True),
    ((128, 96, 33), (17, 128, 5), 0.1, True, False),   # ragged widths, 3 hidden layers (generic kernel)
])
def test_one_update_matches_oracle(ora, hidden_a, hidden_q, wscale, sweep, generic):
    G = 4
    pop, agents = make_pair(ora, G=G, hidden_a=hidden_a, hidden_q=hidden_q, wscale=wscale, sweep=sweep, force_generic=generic)
    ws_prev = [pop.weights(n) for n in range(4)]
    rng = np.random.RandomState(5)
    idx = rng.randint(0, 1000, (G, 64)).astype(np.int32)
    pop.update(idx)
    last = pop.last()
    for g, ag in enumerate(agents):
        loss, used = ag.update(idx[g])
        qt, dact = ag.last()
        assert np.array_equal(used, last["idx"][g])
        assert rel_err(last["q_targets"][g], qt) <= TOL
        assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
        assert rel_err(last["dact"][g], dact) <= TOL
    assert np.all(pop.adam(0)[2] == 1) and np.all(pop.adam(1)[2] == 1)
    check_step(pop, agents, ws_prev, _hyper(agents), 0, TOL)


@pytest.mark.parametrize("nstep,generic", [(1, False), (3, False), (5, True)])
def test_nstep_return_targets(ora, nstep, generic):
    """agent.nstep_return > 0 (DeepQNAg.hpp:360-373, 394-402): horizon cut at goal transitions and at
    the end of the replay (indices near the end are forced into the batch), reference's exponent quirks."""
    G = 3
    pop, agents = make_pair(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, nstep_return=nstep,
                            goal_prob=0.15, force_generic=generic, seed=4)
    ws_prev = [pop.weights(n) for n in range(4)]
    rng = np.random.RandomState(9)
    idx = rng.randint(0, 1000, (G, 64)).astype(np.int32)
    idx[:, :6] = [999, 998, 997, 996, 0, 1]
    pop.update(idx)
    last = pop.last()
    for g, ag in enumerate(agents):
        loss, used = ag.update(idx[g])
        qt, dact = ag.last()
        assert rel_err(last["q_targets"][g], qt) <= TOL
        assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
        assert rel_err(last["dact"][g], dact) <= TOL
    check_step(pop, agents, ws_prev, _hyper(agents), 0, TOL)


# momentum=3 (beta1=0.9, beta2=0: step = lr * m / |g|, unbounded as g -> 0, MLP.hpp:170-173) makes the
# weights grow by orders of magnitude within a few updates in the reference arithmetic itself; it is
# held to the bar for its first update only.
@pytest.mark.parametrize("momentum,hidden_act,actor_out,inverting,n_updates", [
    (0, 1, 2, True, 20), (1, 3, 2, False, 20), (3, 2, 0, False, 1), (2, 1, 2, True, 20)])
def test_twenty_updates_each_within_bar(ora, momentum, hidden_act, actor_out, inverting, n_updates):
    """N = 20 consecutive device updates; before each one the oracle is re-synchronised to the
    device's fp32 state, so every update is held to the 1e-5 bar (see ddpg_common.check_step)."""
    pop, agents = make_pair(ora, G=3, hidden_a=(64, 32), hidden_q=(48, 40), momentum=momentum, hidden_act=hidden_act,
                            actor_out_act=actor_out, inverting_grad=inverting, decay_v=1e-3, seed=2, wscale=0.05)
    for it in range(n_updates):
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        pop.update()
        last = pop.last()
        for g, ag in enumerate(agents):
            loss, used = ag.update()
            assert np.array_equal(used, last["idx"][g])
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            assert rel_err(last["q_targets"][g], ag.last()[0]) <= TOL
            e = rel_err(last["dact"][g], ag.last()[1])
            assert e <= TOL or (hidden_act != 2 and close_frac(last["dact"][g], ag.last()[1], TOL, frac=0.99)), e
        check_step(pop, agents, ws_prev, _hyper(agents), momentum, TOL, allow_kink=True)
    assert np.all(pop.adam(0)[2] == n_updates)


def test_free_running_ten_updates(ora):
    """No re-synchronisation: 10 updates free-running.  Rounding differences are amplified by
    Adam's sign-like step (chaotic), so the bar here is loose and stated: 99 % of the parameters
    within 1e-4."""
    pop, agents = make_pair(ora, G=4, wscale=0.1, sweep=True)
    for it in range(10):
        pop.update()
        for ag in agents:
            ag.update()
    _check_nets(pop, agents, 1e-4, "10 free-running updates", frac=0.99)


def test_small_batch_and_mask_and_guard(ora):
    pop, agents = make_pair(ora, G=4, B=32, hidden_a=(16,), hidden_q=(16, 16), n_replay=300, replay_cap=300, wscale=0.1)
    w0 = [pop.weights(n) for n in range(4)]
    mask = np.array([1, 0, 1, 0], np.uint8)
    pop.update(mask=mask)
    for g, ag in enumerate(agents):
        if mask[g]:
            ag.update()
    _check_nets(pop, agents, TOL, "masked", frac=0.999)
    for n in range(4):
        assert np.array_equal(pop.weights(n)[1], w0[n][1])       # masked agents untouched
    assert list(pop.adam(0)[2]) == [1, 0, 1, 0]
    # guard of DeepQNAg.hpp:338: fewer than 250 samples => no update, no draw consumed
    pop2, agents2 = make_pair(ora, G=2, B=32, hidden_a=(16,), hidden_q=(16, 16), n_replay=249, replay_cap=300)
    w0 = pop2.weights(0)
    pop2.update()
    assert np.array_equal(pop2.weights(0), w0)
    assert np.all(pop2.adam(0)[2] == 0)


def test_replay_ring_append_wraps_like_deque(ora):
    S, A = 5, 2
    pop, agents = make_pair(ora, G=2, S=S, A=A, B=16, hidden_a=(8,), hidden_q=(8,), n_replay=0, replay_cap=260, wscale=0.1)
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
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        pop.update()
        last = pop.last()
        for g, ag in enumerate(agents):
            _, used = ag.update()
            assert np.array_equal(used, last["idx"][g])
            assert rel_err(last["q_targets"][g], ag.last()[0]) <= TOL      # same rows gathered
        check_step(pop, agents, ws_prev, _hyper(agents), 0, TOL)


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


def test_default_init_is_gaussian_001():
    import ddrl_amd
    pop = ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(n_agents=8), init_seed=11)
    w = pop.weights(1)
    assert abs(w.std() - 0.01) < 5e-4 and abs(w.mean()) < 5e-4       # MLP.hpp:936-944
    assert np.array_equal(pop.weights(1), pop.weights(3)) and np.array_equal(pop.weights(0), pop.weights(2))


@pytest.mark.parametrize("hidden,for_critic", [((32, 32), 1), ((100, 100), 1), ((100, 100), 0)])
def test_ewc_lifecycle_matches_oracle(ora, hidden, for_critic):
    import oracle.oracle as o
    kw = dict(G=3, hidden_a=hidden, hidden_q=hidden, ewc=100.0, ewc_decay=0.9, ewc_fisher_beta=0.999,
              ewc_for_critic=for_critic, ewc_best_method=0, wscale=0.1, seed=3)
    pop, agents = make_pair(ora, **kw)
    ewc_nets = (0, 1) if for_critic else (0,)
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
    for net in ewc_nets:
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            so = ag.nets[net].ewc_state()
            assert so["active"] == 1
            fh, fo = st["fisher_hat"][g], so["fisher_hat"]
            assert np.mean(np.abs(fh - fo) <= 1e-4 * np.max(fo)) >= 0.99
            assert np.mean(np.abs(st["wstar"][g] - so["wstar"]) <= 1e-4 * np.max(np.abs(so["wstar"]))) >= 0.99
    # ewc_setup loads w* back into the live net (DODevMLP.hpp:616)
    assert np.array_equal(pop.weights(0), pop.ewc_state(0)["wstar"])
    # from here on hold each update to the bar from synchronised state, penalty live
    for ag_net in ewc_nets:
        st = pop.ewc_state(ag_net)
        for g, ag in enumerate(agents):
            n = ag.nets[ag_net]
            # re-synchronise the oracle's EWC state to the device's (fp32) values
            so = n.ewc_state()
            so["fisher_hat"][:] = st["fisher_hat"][g]; so["wstar"][:] = st["wstar"][g]
            o.lib().ora_mlp_set_ewc(n.h, st["fisher_hat"][g].ctypes.data, st["wstar"][g].ctypes.data)
    for it in range(3):                       # task 2: penalty live + Fisher restarts
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        pop.update()
        for ag in agents:
            ag.update()
        check_step(pop, agents, ws_prev, _hyper(agents), 0, TOL)
        pop.ewc_decay_update()
        for ag in agents:
            o.lib().ora_mlp_ewc_decay_update(ag.ann.h); o.lib().ora_mlp_ewc_decay_update(ag.qnn.h)
    for net in ewc_nets:
        sync_oracle_from_gpu(pop, agents)
        cost = pop.ewc_cost(net)
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            assert abs(cost[g] - ag.nets[net].ewc_cost()) <= 1e-5 * abs(ag.nets[net].ewc_cost()) + 1e-12
            so = ag.nets[net].ewc_state()
            assert np.mean(np.abs(st["fisher"][g] - so["fisher"]) <= 1e-4 * np.max(so["fisher"])) >= 0.99
            assert abs(st["fisher_nbr"][g] - so["fisher_nbr"]) <= 1e-5 * so["fisher_nbr"]
            assert st["decay_mult"][g] == so["decay_mult"]


@pytest.mark.parametrize("hidden_a,hidden_q", [((64, 33), (100, 100)), ((400, 300), (160, 129))])   # the second: wider than the LDS-resident passes
def test_forward_batch_all_nets(ora, hidden_a, hidden_q):
    """MLP::computeOutBatch / computeOutVFBatch (MLP.hpp:531-579): n rows per agent, n not a multiple of 64."""
    G, S, A = 3, 18, 6
    pop, agents = make_pair(ora, G=G, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.05 if max(hidden_a) > 128 else 0.2, seed=3)
    rng = np.random.RandomState(7)
    for n in (1, 64, 150):
        xs = rng.uniform(-1, 1, (G, n, S)).astype(np.float32)
        xq = rng.uniform(-1, 1, (G, n, S + A)).astype(np.float32)
        for net, x in ((0, xs), (1, xq), (2, xs), (3, xq)):
            out = pop.forward_batch(net, x)
            for g, ag in enumerate(agents):
                assert rel_err(out[g], ag.nets[net].forward(x[g].astype(np.float64))) <= TOL, (n, net, g)


# minibatches of more than one 64-row tile (kernels_ddpg_big.hip): the reference's own ddpg.ini / ddpg-cp.ini
# use mini_batch_size = 512 with the "small" nets; 96 = one and a half tiles; 200 with EWC, gates off
@pytest.mark.parametrize("B,hidden_a,hidden_q,kw", [
    (512, (20,), (50, 7), dict()),
    (96, (100, 100), (100, 100), dict(nstep_return=3, goal_prob=0.1)),
    (200, (33, 17), (40, 21, 9), dict(momentum=1, decay_v=1e-3, inverting_grad=False)),
])
def test_large_minibatch_updates(ora, B, hidden_a, hidden_q, kw):
    G = 3
    pop, agents = make_pair(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.1, sweep=True, seed=6, **kw)
    rng = np.random.RandomState(8)
    for it in range(3):
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        idx = rng.randint(0, 1000, (G, B)).astype(np.int32) if it == 0 else None     # host indices, then the device's own draws
        if idx is not None:
            idx[:, :4] = [999, 998, 0, 1]
        mask = np.array([1, 1, 0], np.uint8) if it == 2 else None
        pop.update(idx, mask=mask)
        last = pop.last()
        for g, ag in enumerate(agents):
            if mask is not None and not mask[g]:
                assert np.array_equal(pop.weights(1)[g], ws_prev[1][g])
                continue
            loss, used = ag.update(None if idx is None else idx[g])
            qt, dact = ag.last()
            assert np.array_equal(used, last["idx"][g]), (it, g)
            assert rel_err(last["q_targets"][g], qt) <= TOL
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            e = rel_err(last["dact"][g], dact)
            assert e <= TOL or close_frac(last["dact"][g], dact, TOL, frac=0.99), e
        if mask is None:
            check_step(pop, agents, ws_prev, _hyper(agents), kw.get("momentum", 0), TOL, allow_kink=True)
    assert list(pop.adam(0)[2]) == [3, 3, 2] and list(pop.adam(1)[2]) == [3, 3, 2]


# the layer-by-layer kernels (kernels_ddpg_wide.hip): BASELINE.md "large" (400 x 300), a ragged wide pair with a
# 1.5-tile minibatch and n-step targets, and -- forced -- the "mid" nets the fused kernel normally takes
@pytest.mark.parametrize("B,hidden_a,hidden_q,kw", [
    (64, (400, 300), (400, 300), dict()),
    (96, (160, 200), (300,), dict(nstep_return=2, goal_prob=0.1, momentum=1, decay_v=1e-3)),
    (64, (100, 100), (100, 100), dict(force_wide=True)),
    (32, (20,), (50, 7), dict(force_wide=True, inverting_grad=False)),
    # EWC on both nets, Fisher accumulating: one chunk and two (the dW kernels sum the chunks and run the optimiser)
    (64, (160, 130), (140, 129), dict(ewc=100.0, ewc_fisher_beta=0.999, ewc_for_critic=1)),
    (100, (160, 130), (140, 129), dict(ewc=100.0, ewc_fisher_beta=0.999, ewc_for_critic=1)),
])
def test_wide_nets_layer_by_layer(ora, B, hidden_a, hidden_q, kw):
    G = 2
    pop, agents = make_pair(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.05, sweep=True, seed=7, **kw)
    rng = np.random.RandomState(3)
    for it in range(3):
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        idx = rng.randint(0, 1000, (G, B)).astype(np.int32) if it == 0 else None
        pop.update(idx)
        last = pop.last()
        for g, ag in enumerate(agents):
            loss, used = ag.update(None if idx is None else idx[g])
            qt, dact = ag.last()
            assert np.array_equal(used, last["idx"][g]), (it, g)
            assert rel_err(last["q_targets"][g], qt) <= TOL
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            e = rel_err(last["dact"][g], dact)
            assert e <= TOL or close_frac(last["dact"][g], dact, TOL, frac=0.99), e
        check_step(pop, agents, ws_prev, _hyper(agents), kw.get("momentum", 0), TOL, allow_kink=True)
    assert list(pop.adam(0)[2]) == [3, 3]
    if kw.get("ewc", -1.0) >= 0:              # the running Fisher of three updates (the oracle's gradients were resynced each step)
        for net in (0, 1):
            st = pop.ewc_state(net)
            for g, ag in enumerate(agents):
                so = ag.nets[net].ewc_state()
                assert close_frac(st["fisher"][g], so["fisher"], 1e-4, 0.99, 1e-1), (net, g)
                assert abs(st["fisher_nbr"][g] - so["fisher_nbr"]) <= 1e-5 * so["fisher_nbr"]
    # inference through the wide nets
    s = rng.uniform(-1, 1, (G, 18)).astype(np.float32)
    a = pop.infer(s)
    for g, ag in enumerate(agents):
        assert np.max(np.abs(a[g] - ag.ann.forward(s[g].astype(np.float64))[0])) <= 1e-5


@pytest.mark.parametrize("G,B,hidden_a,hidden_q", [(19, 100, (160, 130), (140, 129)), (24, 64, (160, 130), (140, 129)),
                                                   (11, 160, (32, 16), (24, 16))])
def test_xcd_order_many_agents(ora, G, B, hidden_a, hidden_q):
    """the XCD-aware workgroup order (xcd_decode) of the layer-by-layer forward / input-gradient launches and of the
    chunk launches of large minibatches (third case) with more groups than XCDs, counts that are not a multiple
    of 8 (19 x 2 chunks, 11 agents) and one that is: every agent against the oracle"""
    pop, agents = make_pair(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.05 if max(hidden_a) > 128 else 0.1, sweep=True, seed=11)
    for it in range(2):
        ws_prev, _ = sync_oracle_from_gpu(pop, agents)
        pop.update()
        last = pop.last()
        for g, ag in enumerate(agents):
            loss, used = ag.update(None)
            qt, dact = ag.last()
            assert np.array_equal(used, last["idx"][g]), (it, g)
            assert rel_err(last["q_targets"][g], qt) <= TOL, (it, g)
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss), (it, g)
        check_step(pop, agents, ws_prev, _hyper(agents), 0, TOL, allow_kink=True)


def _multi_update_mode(pop, mode):
    """which kernel runs a multi-update launch of the compile-time shapes: the in-workgroup loop (the default for mid nets),
    agent pairs, or one workgroup per (update, agent) with the sc1 hand-off"""
    pop.set_option("loop", {"loop": 2, "pair": 0, "handoff": 0, "w8": 2}[mode])
    pop.set_option("pair", {"loop": 0, "pair": 2, "handoff": 0, "w8": 0}[mode])
    pop.set_option("w8", 2 if mode == "w8" else 0)      # "w8": eight waves per agent (kernels_ddpg_w8.hip; mid nets only, else the loop kernel)


@pytest.mark.parametrize("pair", ["loop", "pair", "handoff", "w8"])
@pytest.mark.parametrize("B,hidden,kw", [(64, (100, 100), dict()), (32, (20,), dict(nstep_return=2, goal_prob=0.1)), (96, (24, 16), dict())])
def test_fused_consecutive_updates(ora, B, hidden, kw, pair):
    """ddrl_pop_ddpg_update_n: n passes of the fupd loop in one launch (one-tile minibatches) == n calls of
    ddrl_pop_ddpg_update, bit for bit, including the replay-index stream; and against the oracle.  pair = 1: launches
    of the compile-time shapes run as agent pairs (kernels_ddpg_pair.hip; G = 3 leaves one team without an agent, the
    mask another one idle); "handoff": one workgroup per (update, agent) with the sc1 hand-off; "loop": one workgroup walks its
    agent through the launch (ddpg_update_loop_kernel, the default for mid nets)."""
    G, n = 3, 4
    popa, agents = make_pair(ora, G=G, B=B, hidden_a=hidden, hidden_q=hidden, wscale=0.1, sweep=True, seed=5, **kw)
    _multi_update_mode(popa, pair)
    popb, _ = make_pair(ora, G=G, B=B, hidden_a=hidden, hidden_q=hidden, wscale=0.1, sweep=True, seed=5, **kw)
    popb.set_option("w8", 0)                  # the single-update reference: always the four-wave kernel
    mask = np.array([1, 0, 1], np.uint8)
    popa.update_n(n, mask=mask)
    used = []
    for _ in range(n):
        popb.update(mask=mask)
        used.append(popb.last()["idx"].copy())
    for net in range(4):
        assert np.array_equal(popa.weights(net), popb.weights(net)), net
    for net in (0, 1):
        for x, y in zip(popa.adam(net), popb.adam(net)):
            assert np.array_equal(x, y)
    la, lb = popa.last(), popb.last()
    for g in range(G):
        if mask[g]:
            assert np.array_equal(la["idx"][g], lb["idx"][g]) and la["loss"][g] == lb["loss"][g]
    assert list(popa.adam(0)[2]) == [n, 0, n]
    # host indices: [G x n x B]
    idx = np.stack(used, axis=1)
    popc, _ = make_pair(ora, G=G, B=B, hidden_a=hidden, hidden_q=hidden, wscale=0.1, sweep=True, seed=5, **kw)
    popc.update_n(n, idx=idx, mask=mask)
    for net in range(4):
        assert np.array_equal(popc.weights(net), popb.weights(net)), net
    # and the oracle, free-running over the same n updates (1e-4 on 99 %: see test_free_running_ten_updates)
    for g, ag in enumerate(agents):
        if mask[g]:
            for u in range(n):
                ag.update(used[u][g])
    for net in range(4):
        w, wo = popa.weights(net), oracle_weights(agents, net)
        for g in range(G):
            assert close_frac(w[g], wo[g], 1e-4, 0.99, 1e-2), (net, g)


@pytest.mark.parametrize("ewc", [True, False])
def test_resident_small_nets(ora, ewc):
    """BASELINE's small nets (a = 20, q = 50 x 7): with the "resident" option launches of two or more updates keep the
    agent's whole state in LDS (ddpg_update_resident_kernel) -- bit-identical to separate launches of the one-agent kernel (weights, targets, Adam
    state, Fisher, iteration counters, diagnostics), with and without EWC live, masked agents untouched."""
    G, n = 5, 3
    kw = dict(ewc=100.0, ewc_for_critic=1) if ewc else {}
    popa, _ = make_pair(ora, G=G, B=64, hidden_a=(20,), hidden_q=(50, 7), wscale=0.1, sweep=True, seed=9, **kw)
    popb, _ = make_pair(ora, G=G, B=64, hidden_a=(20,), hidden_q=(50, 7), wscale=0.1, sweep=True, seed=9, **kw)
    mask = np.array([1, 1, 0, 1, 1], np.uint8)
    for p in (popa, popb):
        p.update()
        if ewc:
            p.update_best_param(np.ones(G)); p.ewc_setup()
    popa.set_option("resident", 1)
    popb.set_option("pair", 0)
    for rnd in range(2):
        popa.update_n(n, mask=mask)
        for _ in range(n):
            popb.update(mask=mask)
    for net in range(4):
        assert np.array_equal(popa.weights(net), popb.weights(net)), net
    for net in (0, 1):
        for x, y in zip(popa.adam(net), popb.adam(net)):
            assert np.array_equal(x, y)
        if ewc:
            sa, sb = popa.ewc_state(net), popb.ewc_state(net)
            assert np.array_equal(sa["fisher"], sb["fisher"]) and np.array_equal(sa["fisher_nbr"], sb["fisher_nbr"])
    la, lb = popa.last(), popb.last()
    for g in range(G):
        if mask[g]:
            for k in la:
                assert np.array_equal(la[k][g], lb[k][g]), (k, g)
    assert list(popa.adam(0)[2]) == [1 + 2 * n, 1 + 2 * n, 1, 1 + 2 * n, 1 + 2 * n]


@pytest.mark.parametrize("pair", ["loop", "pair", "handoff", "w8"])
def test_fused_consecutive_updates_full_chip(pair):
    """the same identity at the bench's scale, EWC live.  pair = 0: 8 x 1100 workgroups against 512 resident slots, so
    an agent's consecutive updates run in different dispatch rounds, on different CUs and XCDs (the sc1 hand-off);
    pair = 1: 551 workgroups of two agents each, one quarter of an update apart, an odd agent count (the last
    workgroup's second team has no agent).  The diagnostics are those of the LAST update"""
    import ddrl_amd
    from bench import synth_rows
    G, n = 1101, 8
    pops = []
    for _ in range(2):
        cfg = ddrl_amd.DdpgConfig(n_agents=G, replay_memory=600, ewc=100.0, ewc_decay=-1.0, ewc_for_critic=1)
        p = ddrl_amd.DdpgPopulation(cfg, init_seed=3)
        rng = np.random.RandomState(1)
        p.set_hyper("alpha_a", 10 ** rng.uniform(-5, -2, G)); p.set_hyper("alpha_v", 10 ** rng.uniform(-5, -2, G))
        p.seed_replay((77 + np.arange(G)).astype(np.uint32))
        rows = [synth_rows(k, 600, 18, 6) for k in range(8)]
        for g in range(G):
            p.replay_load(g, rows[g % 8])
        p.update()
        p.update_best_param(np.ones(G)); p.ewc_setup()
        pops.append(p)
    a, b = pops
    _multi_update_mode(a, pair)
    b.set_option("w8", 0)
    for _ in range(2):
        a.update_n(n)
        for _ in range(n):
            b.update()
    a.sync()
    for net in range(4):
        assert np.array_equal(a.weights(net), b.weights(net)), net
    for net in (0, 1):
        for x, y in zip(a.adam(net), b.adam(net)):
            assert np.array_equal(x, y)
        sa, sb = a.ewc_state(net), b.ewc_state(net)
        assert np.array_equal(sa["fisher"], sb["fisher"]) and np.array_equal(sa["fisher_nbr"], sb["fisher_nbr"])
    la, lb = a.last(), b.last()
    for k in la:
        assert np.array_equal(la[k], lb[k]), k
    assert np.all(a.adam(1)[2] == 1 + 2 * n)


@pytest.mark.parametrize("method", [0, 1, 2, 3, 4])
def test_best_param_methods_pick_the_reference_snapshot(ora, method):
    """DODevMLP::update_best_param_previous_task (DODevMLP.hpp:722-758): which weights become the anchor of the next ewc_setup.
    0 / 4: the best score so far; 1: the latest; 2: the best score, or anything once the snapshot is more than 50 calls old;
    3: the first call after a setup.  The device snapshot is a copy: the anchor must equal the chosen weights bit for bit,
    per agent (agents see different score sequences)."""
    G = 2
    pop, _ = make_pair(ora, G=G, hidden_a=(16,), hidden_q=(16,), ewc=10.0, ewc_fisher_beta=0.99, ewc_for_critic=1,
                       ewc_best_method=method, wscale=0.1, seed=method)
    rng = np.random.RandomState(method)
    n_calls = 60 if method == 2 else 6
    scores = np.stack([np.r_[5.0, rng.uniform(0, 4, n_calls - 1)], np.r_[1.0, 2.0, 3.0, rng.uniform(0, 2.5, n_calls - 3)]], 1)   # [call][agent]
    snaps = []
    best = [-np.inf] * G; age = [0] * G; chosen = [None] * G
    for k in range(n_calls):
        pop.update()
        snaps.append([pop.weights(0), pop.weights(1)])
        pop.update_best_param(scores[k])
        for g in range(G):
            s = scores[k, g]
            if method in (0, 4):
                go = s > best[g]
            elif method == 1:
                go = True
            elif method == 2:
                too_old = age[g] > 50
                age[g] += 1
                go = s > best[g] or too_old
                if go:
                    age[g] = 0
            else:
                age[g] += 1
                go = age[g] == 1
            if go:
                best[g] = s; chosen[g] = k
    pop.ewc_setup()
    for net in (0, 1):
        ws = pop.ewc_state(net)["wstar"]
        for g in range(G):
            assert np.array_equal(ws[g], snaps[chosen[g]][net][g]), (method, net, g, chosen[g])
    if method == 2:
        assert chosen[0] > 0          # agent 0's first score is never beaten: only the age rule can have replaced its snapshot


def test_error_word_blocks_every_reader_of_state(ora, tmp_path):
    """a multi-update launch that skipped updates (an expired hand-off or team-barrier wait raises the launch's error word)
    must not be read back as healthy state: weights, Adam moments, EWC arrays and run files all fail with DDRL_ESTATE, once;
    the word is raised here through the test hook "debug_raise_update_error" (what the kernels' bounded waits leave behind)"""
    from ddrl_amd.lib import DdrlError
    readers = [lambda p: p.weights(0), lambda p: p.adam(1), lambda p: p.ewc_state(0), lambda p: p.save_run(str(tmp_path / "x.run")),
               lambda p: p.set_weights(0, np.zeros((2, p.num_params(0))))]
    pop, _ = make_pair(ora, G=2, B=64, hidden_a=(20,), hidden_q=(50, 7), ewc=10.0, seed=3)
    pop.update_n(2)
    pop.sync()
    for rd in readers:
        pop.set_option("debug_raise_update_error", 1)
        with pytest.raises(DdrlError) as e:
            rd(pop)
        assert e.value.code == -4 and "skipped" in str(e.value)          # DDRL_ESTATE
        rd(pop)                                                            # reported once, then the handle answers again


@pytest.mark.parametrize("kw", [dict(), dict(ewc=100.0, ewc_for_critic=1), dict(B=40, inverting_grad=1), dict(nstep_return=3, goal_prob=0.1),
                                dict(gates="det"), dict(gates="bern", ewc=10.0)])
def test_eight_waves_per_agent_bit_identical(ora, kw):
    """kernels_ddpg_w8.hip (512-thread workgroups, two waves per 16-row tile; the default for populations of up to 256 agents of
    the mid nets) against the four-wave kernels: single updates and multi-update launches, bit for bit -- weights, targets, Adam
    state, Fisher, diagnostics -- with EWC live, a short minibatch, n-step targets, deterministic and Bernoulli gates, a mask."""
    kw = dict(kw)
    B = kw.pop("B", 64)
    gates = kw.pop("gates", None)
    ewc = "ewc" in kw
    G, n = 5, 3
    pops = []
    for w8 in (2, 0):
        p, _ = make_pair(ora, G=G, B=B, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, sweep=True, seed=13, **kw)
        p.set_option("w8", w8)
        if gates:
            rng = np.random.RandomState(2)
            if gates == "det":
                p.set_gates((rng.uniform(0, 1, (G, 18)) < 0.7).astype(np.float64), (rng.uniform(0, 1, (G, 6)) < 0.7).astype(np.float64))
            else:
                p.set_gates(rng.uniform(0.3, 1, (G, 18)), rng.uniform(0.3, 1, (G, 6)))
                p.set_gate_mode(1, 1, 1, 0, seed=21)
        pops.append(p)
    a, b = pops
    mask = np.array([1, 1, 0, 1, 1], np.uint8)
    for p in pops:
        p.update()
        if ewc:
            p.update_best_param(np.ones(G)); p.ewc_setup()
        p.update(mask=mask)
        p.update_n(n, mask=mask)
        p.update_n(2)
    for net in range(4):
        assert np.array_equal(a.weights(net), b.weights(net)), net
    for net in (0, 1):
        for x, y in zip(a.adam(net), b.adam(net)):
            assert np.array_equal(x, y)
        if ewc and (net == 0 or kw.get("ewc_for_critic")):
            sa, sb = a.ewc_state(net), b.ewc_state(net)
            assert np.array_equal(sa["fisher"], sb["fisher"]) and np.array_equal(sa["fisher_nbr"], sb["fisher_nbr"])
    la, lb = a.last(), b.last()
    for k in la:
        assert np.array_equal(la[k], lb[k]), k
    assert list(a.adam(0)[2]) == [4 + n, 4 + n, 3, 4 + n, 4 + n]
    a.close(); b.close()


@pytest.mark.parametrize("td3", [False, True])
def test_wide_update_over_two_streams_is_the_same_update(td3):
    """Layers wider than 128 (the layer-by-layer kernels): from 128 agents on, the launch chain is cut into two halves of the
    population on two streams (kernels_ddpg_wide.hip, wide_launch).  Agents are independent, so an agent's update must not
    depend on how many agents the handle holds: agents 0, 64 (first half) and 129 (second half) of a 130-agent population
    against a 3-agent population holding the same three agents -- weights, targets and Adam moments bit for bit, three
    updates, DDPG and TD3 (delayed policy step included)."""
    import ddrl_amd
    from ddpg_common import synth_transitions
    S, A, B, Gbig, pick = 18, 6, 64, 130, (0, 64, 129)
    kw = dict(nb_sensors=S, nb_motors=A, hidden_unit_a=(160, 130), hidden_unit_q=(140, 129), replay_memory=600, mini_batch_size=B,
              inverting_grad=1, ewc=100.0, ewc_for_critic=1)
    cls = ddrl_amd.DdpgPopulation
    if td3:
        kw.update(td3=1, inverting_grad=0, policy_noise=0.2, noise_clip=0.5, policy_freq=2)
        cls = ddrl_amd.Td3Population
    big = cls(ddrl_amd.DdpgConfig(n_agents=Gbig, **kw), init_seed=11)
    small = cls(ddrl_amd.DdpgConfig(n_agents=len(pick), **kw), init_seed=12)
    nets = range(6 if td3 else 4)
    for n in nets:
        small.set_weights(n, big.weights(n)[list(pick)])
    rows = {g: synth_transitions(np.random.RandomState(100 + g), 500, S, A) for g in pick}
    filler = synth_transitions(np.random.RandomState(7), 500, S, A)
    for g in range(Gbig):
        big.replay_load(g, rows.get(g, filler))
    for k, g in enumerate(pick):
        small.replay_load(k, rows[g])
    for upd in range(3):
        idx = np.random.RandomState(50 + upd).randint(0, 500, (Gbig, B)).astype(np.int32)
        if td3:
            noise = np.random.RandomState(60 + upd).normal(0, 0.2, (Gbig, B, A)).astype(np.float32)
            big.update(idx, noise=noise, actor_step=upd % 2)
            small.update(np.ascontiguousarray(idx[list(pick)]), noise=np.ascontiguousarray(noise[list(pick)]), actor_step=upd % 2)
        else:
            big.update(idx)
            small.update(np.ascontiguousarray(idx[list(pick)]))
    for n in nets:
        assert np.array_equal(big.weights(n)[list(pick)], small.weights(n)), n
    for n in ((0, 1, 4) if td3 else (0, 1)):
        mb, vb, ib = big.adam(n)
        ms, vs, is_ = small.adam(n)
        assert np.array_equal(mb[list(pick)], ms) and np.array_equal(vb[list(pick)], vs) and np.array_equal(ib[list(pick)], is_)
    big.close(); small.close()
