"""BatchNorm / Scale towers on the device (kernels_ddpg_bn.hip, bn_device.h) against the CPU oracle (oracle/mlp_ref.c, pinned
by tests/test_oracle_bn.py on the reference's own consistency properties, CaffeUtest.cpp:246-386, 1343-1447).

fp32 against fp64 through a normalisation: the towers divide by sqrt(var + 1e-5), which amplifies rounding where a channel
barely varies over the minibatch, so the bar here is 1e-4 norm-wise (stated per assertion) instead of the 1e-5 of the plain
nets; the update is teacher-forced (the oracle restarts every step from the device's state)."""
import os

import numpy as np
import pytest

from ddpg_common import BETAS, make_pair, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
CODES = [4, 6, 7, 8, 10, 11, 12, 14, 15]


def structured_weights(rng, mlp, wscale, G):
    """[G x np]: InnerProducts ~ N(0, wscale), tower statistics of a net that has seen a few batches, gamma ~ 1, beta ~ 0"""
    lay = mlp.bn_layout()
    w = rng.normal(0, wscale, (G, mlp.np))
    frozen = np.zeros(mlp.np, bool)
    for r in range(len(mlp.dims)):
        if not lay["rank_on"][r]:
            continue
        C, o = int(lay["chan"][r]), int(lay["offbn"][r])
        factor = 4.5
        w[:, o:o + C] = factor * rng.normal(0, 0.3, (G, C))
        w[:, o + C:o + 2 * C] = factor * rng.uniform(0.2, 1.5, (G, C))
        w[:, o + 2 * C] = factor
        frozen[o:o + 2 * C + 1] = True
        if lay["scale"]:
            s = int(lay["offsc"][r])
            w[:, s:s + C] = 1.0 + rng.normal(0, 0.2, (G, C))
            if lay["bias"]:
                w[:, s + C:s + 2 * C] = rng.normal(0, 0.2, (G, C))
    return w.astype(np.float32).astype(np.float64), frozen


def make_bn_pair(ora, bna, bnq, G=3, hidden_a=(32, 16), hidden_q=(24, 16), wscale=0.3, seed=0, **kw):
    pop, agents = make_pair(ora, G=G, hidden_a=hidden_a, hidden_q=hidden_q, seed=seed, batch_norm_actor=bna, batch_norm_critic=bnq,
                            inverting_grad=kw.pop("inverting_grad", False), **kw)
    rng = np.random.RandomState(seed + 50)
    frozen = {}
    for net in (0, 1):
        assert pop.num_params(net) == agents[0].nets[net].np
        w, frozen[net] = structured_weights(rng, agents[0].nets[net], wscale, G)
        wt = w + (~frozen[net]) * rng.normal(0, 0.02, w.shape)
        wt = wt.astype(np.float32).astype(np.float64)
        pop.set_weights(net, w); pop.set_weights(net + 2, wt)
        for g, a in enumerate(agents):
            a.nets[net].set_weights(w[g]); a.nets[net + 2].set_weights(wt[g])
            if net == 0:
                a.sync_testing()
    return pop, agents, frozen


def sync(pop, agents, testing=None):
    ws = [pop.weights(n) for n in range(4)]
    ad = [pop.adam(n) for n in (0, 1)]
    for g, a in enumerate(agents):
        for n in range(4):
            a.nets[n].set_weights(ws[n][g])
        for n in (0, 1):
            a.nets[n].set_adam(ad[n][0][g], ad[n][1][g], int(ad[n][2][g]))
        a.ann_testing.set_weights(ws[0][g] if testing is None else testing[g])
    return ws


def critic_kills_action_gradient(bnq):
    """A tower directly ahead of the critic's last InnerProduct ("all" / "very_all" layouts): dQ/d(input of that layer) is the
    same row of W for every sample, and a TRAIN-phase BatchNorm backward maps a per-channel constant to exactly 0
    (d - mean(d) - mean(d y) y with mean(y) = 0).  The reference's actor then steps on rounding noise; so do we, and
    neither side's noise can be compared."""
    return bool(bnq & 4) and bool(bnq & (8 | 16))


def check_update(pop, agents, frozen, ws_prev, momentum=0, tol=TOL, actor_noise=False):
    b1, b2 = BETAS[momentum]
    for net in (0, 1):
        m, v, it = pop.adam(net)
        w = pop.weights(net)
        for g, a in enumerate(agents):
            mo, vo, _ = a.nets[net].adam()
            wo = a.nets[net].weights()
            fz = frozen[net]
            if net == 0 and actor_noise:
                assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= tol, ("statistics", net, g)
                continue
            assert rel_err(m[g], mo) <= tol, ("m", net, g, rel_err(m[g], mo))
            assert rel_err(v[g], vo) <= 2 * tol, ("v", net, g, rel_err(v[g], vo))
            assert np.all(m[g][fz] == 0) and np.all(v[g][fz] == 0)                    # lr_mult 0: the statistics never step
            assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= tol, ("statistics", net, g, rel_err(w[g][fz], wo[fz]))
            t = int(it[g])
            lr = a.cfg.alpha_a if net == 0 else a.cfg.alpha_v
            lr_t = float(np.float32(lr)) * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            expect = ws_prev[net][g] - lr_t * m[g] / (np.sqrt(v[g]) + 1e-8)
            wmax = np.max(np.abs(wo[~fz]))
            assert np.max(np.abs(w[g][~fz] - expect[~fz])) <= 1e-5 * wmax, ("adam-consistency", net, g)
            err = np.abs(w[g][~fz] - wo[~fz])
            assert np.mean(err <= tol * wmax) >= 0.995, ("weights", net, g, np.mean(err <= tol * wmax))
            # Adam's step is at most lr sqrt((1 - b2^t) / (1 - b2)) <= lr sqrt(t) per entry whatever the gradient (|m| / sqrt(v) <=
            # 1 / sqrt(1 - b2)): entries whose gradient is rounding-level against the norm may differ by up to twice that
            assert np.max(err) <= 2 * float(np.float32(lr)) * np.sqrt(t) * 1.001 + tol * wmax, ("step bound", net, g)
    for net in (2, 3):
        w = pop.weights(net)
        for g, a in enumerate(agents):
            wo = a.nets[net].weights()
            fz = frozen[net - 2]
            if not (net == 2 and actor_noise):
                assert np.mean(np.abs(w[g] - wo) <= tol * np.max(np.abs(wo))) >= 0.995, ("target", net, g)
            assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= tol, ("target statistics", net, g)


@pytest.mark.parametrize("bna,bnq", [(14, 11), (4, 8), (7, 12), (15, 6), (22, 19), (23, 16), (10, 23), (0, 7), (6, 0)])
def test_ddpg_update_with_towers(ora, bna, bnq):
    """three teacher-forced updates; 14 / 11 are agent/deepqn/data/best-ddpg-halfcheetah.ini:63-64's, 16+ the first_except_action
    (critic) and very_all layouts"""
    G, B = 3, 64
    pop, agents, frozen = make_bn_pair(ora, bna, bnq, G=G, seed=bna + bnq)
    noise = critic_kills_action_gradient(bnq)
    for upd in range(3):
        ws_prev = sync(pop, agents)
        idx = np.random.RandomState(5 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL, (upd, g, rel_err(last["q_targets"][g], qt))
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            if noise:      # exactly 0 in exact arithmetic: both are rounding noise far below the critic's weights
                assert np.max(np.abs(dact)) <= 1e-12 and np.max(np.abs(last["dact"][g])) <= 1e-4
            else:
                assert rel_err(last["dact"][g], dact) <= TOL, (upd, g, rel_err(last["dact"][g], dact))
        check_update(pop, agents, frozen, ws_prev, actor_noise=noise)
    pop.close()


@pytest.mark.parametrize("kw", [dict(), dict(bn_adapt=True, inverting_grad=True), dict(test_net=True, momentum=1), dict(ewc=100.0, ewc_for_critic=1)])
def test_ddpg_update_with_towers_tuned_ini_nets(ora, kw):
    """agent/deepqn/data/best-ddpg-halfcheetah.ini's nets (hidden_unit_a = 5:5:2, hidden_unit_q = 20:5:2, batch_norm 14 / 11): small
    enough for the kernel to run them from LDS copies of the agent's parameter blocks (kernels_ddpg_bn.hip, bn_resident) -- weights,
    running statistics, targets, gradient images; three teacher-forced updates, every blob against the oracle"""
    G, B = 3, 64
    pop, agents, frozen = make_bn_pair(ora, 14, 11, G=G, hidden_a=(5, 5, 2), hidden_q=(20, 5, 2), seed=77, **kw)
    momentum = kw.get("momentum", 0)
    for upd in range(3):
        ws_prev = sync(pop, agents)
        idx = np.random.RandomState(25 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL, (upd, g, rel_err(last["q_targets"][g], qt))
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            assert rel_err(last["dact"][g], dact) <= TOL, (upd, g, rel_err(last["dact"][g], dact))
        check_update(pop, agents, frozen, ws_prev, momentum=momentum)
    st = np.random.RandomState(2).uniform(-1, 1, (G, 18)).astype(np.float32)
    act = pop.infer(st)
    sync(pop, agents)
    for g, a in enumerate(agents):
        assert rel_err(act[g], a.ann_testing.forward(st[g].astype(np.float64))[0]) <= TOL
    pop.close()


@pytest.mark.parametrize("kw", [dict(test_net=True), dict(bn_adapt=True, inverting_grad=True), dict(B=40, nstep_return=3, momentum=1),
                                dict(hidden_act=2, actor_out_act=2, decay_v=0.01), dict(hidden_act=3, actor_out_act=0)])
def test_ddpg_update_variants(ora, kw):
    """test_net (targets in TEST phase), bn_adapt (the critic scores ann_testing's actions), a ragged minibatch with n-step
    returns and momentum, tanh / ReLU hidden layers, weight decay on the critic (Scale blobs included)"""
    G = 2
    kw = dict(kw)
    B = kw.get("B", 64)
    momentum = kw.get("momentum", 0)
    pop, agents, frozen = make_bn_pair(ora, 14, 11, G=G, seed=3, **kw)
    testing = pop.weights(0)
    for upd in range(3):
        ws_prev = sync(pop, agents, testing)
        idx = np.random.RandomState(9 + upd).randint(0, 990, (G, B)).astype(np.int32)
        pop.update(idx)
        testing = pop.weights(0)                    # ann_testing <- ann after the update call (DeepQNAg.hpp:469-472)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            assert rel_err(last["dact"][g], dact) <= TOL
        check_update(pop, agents, frozen, ws_prev, momentum)
    pop.close()


@pytest.mark.parametrize("bna", CODES + [20, 22, 23])
def test_inference_is_the_test_phase_copy(ora, bna):
    """ddrl_pop_infer = ann_testing->computeOut (DeepQNAg.hpp:126), a TEST-phase copy holding the actor's parameters;
    forward_batch evaluates any net the same way.  Also the fresh net: factor 0 => mean 0, var 0."""
    import ddrl_amd
    G, S, A = 3, 18, 6
    pop, agents, _ = make_bn_pair(ora, bna, 11, G=G, seed=bna)
    rng = np.random.RandomState(1)
    st = rng.uniform(-1, 1, (G, S)).astype(np.float32)
    act = pop.infer(st)
    x = rng.uniform(-1, 1, (G, 70, S)).astype(np.float32)
    out = pop.forward_batch(0, x)
    xq = rng.uniform(-1, 1, (G, 70, S + A)).astype(np.float32)
    outq = pop.forward_batch(1, xq)
    for g, a in enumerate(agents):
        assert rel_err(act[g], a.ann_testing.forward(st[g].astype(np.float64))[0]) <= 1e-5
        assert rel_err(out[g], a.ann_testing.forward(x[g].astype(np.float64))) <= 1e-5
        assert rel_err(outq[g], a.qnn.clone(test=True).forward(xq[g].astype(np.float64))) <= 1e-5
    pop.close()
    cfg = ddrl_amd.DdpgConfig(n_agents=2, nb_sensors=S, nb_motors=A, hidden_unit_a=(8,), hidden_unit_q=(8,), batch_norm_actor=bna,
                              actor_output_layer_type=0)
    fresh = ddrl_amd.DdpgPopulation(cfg, init_seed=1)
    w = fresh.weights(0)
    net = ora.Mlp([S, 8, A], 1, 0, 1e-3, batch_norm=bna)
    assert w.shape[1] == net.np
    net.set_weights(w[0]); net.set_phase(True)
    assert rel_err(fresh.infer(st[:2])[0], net.forward(st[0].astype(np.float64))[0]) <= 1e-5
    lay = net.bn_layout()
    o, C = int(lay["offbn"][0]), int(lay["chan"][0])
    assert np.all(w[:, o:o + 2 * C + 1] == 0)
    if lay["scale"]:
        assert np.all(w[:, lay["offsc"][0]:lay["offsc"][0] + C] == 1)
    fresh.close()


def test_free_running_and_run_files(ora, tmp_path):
    """ten free-running updates with device-drawn minibatches stay finite; a saved run restores every blob, statistics and
    the acting copy included, and both continue identically"""
    import ddrl_amd
    G = 2
    pop, agents, _ = make_bn_pair(ora, 14, 11, G=G, seed=5, bn_adapt=True)
    for _ in range(5):
        pop.update()
    path = os.path.join(str(tmp_path), "run")
    pop.save_run(path)
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=18, nb_motors=6, hidden_unit_a=(32, 16), hidden_unit_q=(24, 16), replay_memory=2000,
                              batch_norm_actor=14, batch_norm_critic=11, bn_adapt=1, inverting_grad=0)
    other = ddrl_amd.DdpgPopulation(cfg, init_seed=99)
    other.load_previous_run(path)
    st = np.random.RandomState(2).uniform(-1, 1, (G, 18)).astype(np.float32)
    assert np.array_equal(pop.infer(st), other.infer(st))
    for _ in range(5):
        pop.update(); other.update()
    for n in range(4):
        a, b = pop.weights(n), other.weights(n)
        assert np.all(np.isfinite(a)) and np.array_equal(a, b)
    pop.close(); other.close()


def test_reference_ini_loads_and_updates(tmp_path):
    """agent/deepqn/data/best-ddpg-halfcheetah.ini:45-72's [agent] section (hidden 20:5:2 / 5:5:2, batch_norm_actor=14,
    batch_norm_critic=11, alpha 0.3 / 0.1, linear actor output) plus the keys the current DeepQNAg reads that the file
    predates (momentum, nstep_return, bn_adapt): loads, acts and updates"""
    import ddrl_amd
    ini = os.path.join(str(tmp_path), "best.ini")
    with open(ini, "w") as f:
        f.write("[agent]\ngamma=0.99\ndecision_each=1\nnoise=0.05\ngaussian_policy=true\nhidden_unit_q=20:5:2\nhidden_unit_a=5:5:2\n"
                "mini_batch_size=64\nreplay_memory=1000000\ninverting_grad=false\nforce_more_update=0\ntau_soft_update=0.001\n"
                "alpha_a=0.3\nalpha_v=0.1\ndecay_v=-1\nbatch_norm_actor=14\nbatch_norm_critic=11\ncount_last=true\n"
                "actor_output_layer_type=0\nhidden_layer_type=1\ntest_net=false\nmomentum=0\nnstep_return=0\n")
    cfg = ddrl_amd.DdpgConfig.from_ini(ini, n_agents=4, nb_sensors=18, nb_motors=6)
    assert cfg.get("batch_norm_actor") == 14 and cfg.get("batch_norm_critic") == 11 and cfg.get("bn_adapt") == 0
    cfg.set("replay_memory", 5000)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=3)
    # actor: towers on 18, 5, 5, 2 channels with gamma: 3 * 30 + 1 ... ; critic 11: all_except_last with gamma and beta
    assert pop.num_params(0) == (18 * 5 + 5 + 5 * 5 + 5 + 5 * 2 + 2 + 2 * 6 + 6) + sum(3 * c + 1 for c in (18, 5, 5, 2))
    assert pop.num_params(1) == (24 * 20 + 20 + 20 * 5 + 5 + 5 * 2 + 2 + 2 + 1) + sum(4 * c + 1 for c in (24, 20, 5))
    from ddpg_common import synth_transitions
    for g in range(4):
        pop.replay_load(g, synth_transitions(np.random.RandomState(g), 600, 18, 6))
    w0 = [pop.weights(n) for n in range(4)]
    for _ in range(20):
        pop.update()
    for n in range(4):
        w = pop.weights(n)
        assert np.all(np.isfinite(w)) and not np.array_equal(w, w0[n])
    act = pop.infer(np.random.RandomState(1).uniform(-1, 1, (4, 18)).astype(np.float32))
    assert np.all(np.isfinite(act))
    pop.set_gates(np.ones((4, 18)), np.ones((4, 6)))                     # per-dimension gates are served on tower nets ...
    pop.set_gate_mode(1, 0, 1, 0, 7)                                       # ... and so is the per-sample Bernoulli mode: inference,
    pop.set_gates(np.full((4, 18), 0.5), np.full((4, 6), 0.5))              #     forward_batch (tests/test_dev_layer_fixtures_gpu.py)
    out = pop.forward_batch(0, np.random.RandomState(2).uniform(-1, 1, (4, 200, 18)).astype(np.float32))
    assert 0.4 < (out == 0).mean() < 0.6
    pop.update()                                                            #     and the update (test_ddpg_update_with_towers_given_masks)
    assert all(np.all(np.isfinite(pop.weights(n))) for n in range(4))
    pop.close()


@pytest.mark.parametrize("bna,bnq,binary,adapt,inverting", [(14, 11, True, 0, False), (14, 11, False, 0, True), (7, 12, False, 1, True),
                                                             (22, 19, True, 1, False), (0, 7, False, 0, False), (6, 0, True, 0, True)])
def test_ddpg_update_with_towers_and_gates(ora, bna, bnq, binary, adapt, inverting):
    """Developmental gates (DODevMLP.hpp:246-316) on nets with towers, the layouts CaffeUtest.cpp:619-624 runs its Developmental
    tests over: the state gate ahead of the input tower, the action gate behind the actor's output tower and ahead of the
    critic's input tower.  Per-dimension factors (binary: the deterministic layer's 0 / 1; fractional: any w), with and
    without bn_adapt (the TEST-phase copy's actions are gated like the actor's) and inverting gradients; three teacher-forced
    updates, forward_batch on all four nets and the acting copy's inference against the oracle with the same gates."""
    G, B, S, A = 3, 64, 18, 6
    pop, agents, frozen = make_bn_pair(ora, bna, bnq, G=G, seed=bna + bnq + 3, bn_adapt=adapt, inverting_grad=inverting)
    rng = np.random.RandomState(bna * 31 + bnq)
    if binary:
        sg = (rng.uniform(0, 1, (G, S)) < 0.6).astype(np.float64)
        ag = (rng.uniform(0, 1, (G, A)) < 0.6).astype(np.float64)
        ag[:, 0] = 1.0
    else:
        sg = np.round(rng.uniform(0.2, 1, (G, S)), 2).astype(np.float32).astype(np.float64)
        ag = np.round(rng.uniform(0.2, 1, (G, A)), 2).astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    for g, a in enumerate(agents):
        for net in (a.ann, a.ann_target, a.ann_testing):
            net.set_gates(sg[g], ag[g])
        cin = np.concatenate([sg[g], ag[g]])
        a.qnn.set_gates(cin, None); a.qnn_target.set_gates(cin, None)
    noise = critic_kills_action_gradient(bnq)
    for upd in range(3):
        ws_prev = sync(pop, agents)
        idx = np.random.RandomState(15 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL, (upd, g, rel_err(last["q_targets"][g], qt))
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            if noise:
                assert np.max(np.abs(dact)) <= 1e-12 and np.max(np.abs(last["dact"][g])) <= 1e-4
            else:
                assert rel_err(last["dact"][g], dact) <= TOL, (upd, g, rel_err(last["dact"][g], dact))
                assert np.all(last["dact"][g][:, ag[g] == 0] == 0)
        check_update(pop, agents, frozen, ws_prev, actor_noise=noise)
    sync(pop, agents)
    x = np.random.RandomState(4).uniform(-1, 1, (G, 40, S + A)).astype(np.float32)
    for net in range(4):
        xin = x if net in (1, 3) else np.ascontiguousarray(x[:, :, :S])
        out = pop.forward_batch(net, xin)
        for g, a in enumerate(agents):
            ref = a.nets[net].clone(test=True).forward(xin[g].astype(np.float64))       # (a clone keeps the gates)
            assert rel_err(out[g], ref) <= TOL, ("forward_batch", net, g, rel_err(out[g], ref))
            if net in (0, 2):
                assert np.all(out[g][:, ag[g] == 0] == 0)
    st = np.ascontiguousarray(x[:, 0, :S])
    act = pop.infer(st)
    for g, a in enumerate(agents):
        assert rel_err(act[g], a.ann_testing.forward(st[g].astype(np.float64))[0]) <= TOL
    pop.close()


@pytest.mark.parametrize("bna,bnq,scale,adapt,inverting", [(14, 11, 1, 0, True), (7, 12, 0, 1, True), (22, 19, 1, 1, False), (0, 7, 0, 0, False),
                                                            (6, 0, 1, 0, True)])
def test_ddpg_update_with_towers_given_masks(ora, bna, bnq, scale, adapt, inverting):
    """The probabilistic Developmental layer (DODevMLP.hpp:276-304, probabilist = 0) in the update of nets with towers: per-sample
    masks, one per layer instance, the backward passes under their forward pass's mask, the masks ahead of the input towers and
    behind the actor's output tower.  Against the oracle fed the masks the device's generator produces (restated in
    test_gates_prob_gpu.gate_factors; the fork's own generator is absent: parity unpinned beyond the pass rates)."""
    from test_gates_prob_gpu import (ACTOR_OUT, ACTOR_S, ACTOR_T_OUT, ACTOR_T_S, CRITIC_A, CRITIC_A_REPLAY, CRITIC_S, CRITIC_T_A, CRITIC_T_S,
                                     gate_factors)
    G, B, S, A = 3, 64, 18, 6
    pop, agents, frozen = make_bn_pair(ora, bna, bnq, G=G, seed=bna + bnq + 5, bn_adapt=adapt, inverting_grad=inverting)
    rng = np.random.RandomState(bna * 13 + bnq)
    sg = np.round(rng.uniform(0.4, 1, (G, S)), 2); sg[:, ::3] = 1.0
    ag = np.round(rng.uniform(0.4, 1, (G, A)), 2); ag[:, 0] = 1.0
    sg = sg.astype(np.float32).astype(np.float64); ag = ag.astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    noise = critic_kills_action_gradient(bnq)
    seed = 300 + bna
    for upd in range(3):
        pop.set_gate_mode(1, scale, 1, scale, seed=seed + upd)          # draw counter 0 for the coming update
        ws_prev = sync(pop, agents)
        idx = np.random.RandomState(25 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        for g, a in enumerate(agents):
            F = lambda wv, site: gate_factors(wv, 1, scale, seed + upd, 0, g, site, B)
            a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag[g], ACTOR_T_OUT), B)
            a.qnn_target.push_row_gates(np.concatenate([F(sg[g], CRITIC_T_S), F(ag[g], CRITIC_T_A)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A_REPLAY)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A)], 1), None, B)
            a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
            if adapt and bna != 0:                                # the TEST-phase copy's pass (DeepQNAg.hpp:415-420): the actor's masks
                a.ann_testing.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            assert all(n.row_gates_pending() == 0 for n in (a.ann, a.qnn, a.ann_target, a.qnn_target, a.ann_testing))
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL, (upd, g, rel_err(last["q_targets"][g], qt))
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            if noise:
                assert np.max(np.abs(dact)) <= 1e-12 and np.max(np.abs(last["dact"][g])) <= 1e-4
            else:
                assert rel_err(last["dact"][g], dact) <= TOL, (upd, g, rel_err(last["dact"][g], dact))
        check_update(pop, agents, frozen, ws_prev, actor_noise=noise)
    # w in {0, 1}: Bernoulli(w) passes never / always, so the update is the deterministic gates' bit for bit
    res = []
    for bern in (0, 1):
        p2, _, _ = make_bn_pair(ora, bna, bnq, G=G, seed=bna + bnq + 5, bn_adapt=adapt, inverting_grad=inverting)
        p2.set_gates((sg > 0.7).astype(np.float64), (ag > 0.7).astype(np.float64))
        p2.set_gate_mode(bern, bern, bern, bern, seed=4)
        for upd in range(2):
            p2.update(np.random.RandomState(40 + upd).randint(0, 1000, (G, B)).astype(np.int32))
        res.append([p2.weights(n) for n in range(4)])
        p2.close()
    for x, y in zip(*res):
        assert np.array_equal(x, y)
    pop.close()


@pytest.mark.parametrize("bna,bnq,kw", [(14, 11, dict()), (14, 11, dict(bn_adapt=True)), (7, 12, dict(test_net=True)),
                                        (22, 19, dict(ewc=100.0, ewc_for_critic=1)), (0, 7, dict()), (6, 0, dict()),
                                        (14, 11, dict(bern=1, scale=1)), (7, 12, dict(bern=1, scale=0, bn_adapt=True)), (6, 0, dict(bern=1, scale=1)),
                                        (0, 7, dict(bern=1, scale=0)), (7, 12, dict(bn_adapt=True)), (22, 19, dict(bern=1, scale=1, bn_adapt=True))])
def test_td3_update_with_towers(ora, bna, bnq, kw):
    """TD3Ag reads the same batch_norm_actor / _critic, test_net and bn_adapt keys as DeepQNAg (TD3Ag.hpp:176-234): six nets with
    towers.  Four teacher-forced updates alternating critic-only and policy steps (policy_freq = 2), given replay indices and
    smoothing noise: targets (min of the two target critics on the smoothed action), both losses, dQ/da, every blob of the six
    nets (running statistics included; the targets and the acting copy move on policy steps only).  bern: the probabilistic
    Developmental layers on top (per-sample masks, a layer instance per net: the second critic and its target draw their own),
    the oracle fed the masks of the device's generator."""
    import ddrl_amd
    from test_gates_prob_gpu import (ACTOR_OUT, ACTOR_S, ACTOR_T_OUT, ACTOR_T_S, CRITIC_A, CRITIC_A_REPLAY, CRITIC_S, CRITIC_T_A, CRITIC_T_S,
                                     gate_factors)
    C2_T_S, C2_T_A, C2_S, C2_A_REPLAY = 17, 18, 19, 20
    from ddpg_common import synth_transitions
    G, S, A, B = 3, 18, 6, 64
    ha, hq = (32, 16), (24, 16)
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=B, replay_memory=2000, hidden_unit_a=ha, hidden_unit_q=hq,
                              inverting_grad=0, td3=1, policy_noise=0.2, noise_clip=0.5, policy_freq=2, batch_norm_actor=bna, batch_norm_critic=bnq,
                              test_net=int(kw.get("test_net", False)), bn_adapt=int(kw.get("bn_adapt", False)), ewc=kw.get("ewc", -1.0),
                              ewc_for_critic=kw.get("ewc_for_critic", 0))
    pop = ddrl_amd.Td3Population(cfg, init_seed=bna + bnq)
    rng = np.random.RandomState(bna * 7 + bnq)
    alpha_a, alpha_v = 10 ** rng.uniform(-4, -2, G), 10 ** rng.uniform(-4, -2, G)
    tau, gamma = 10 ** rng.uniform(-3, -1, G), rng.uniform(0.9, 0.999, G)
    pop.set_hyper("alpha_a", alpha_a); pop.set_hyper("alpha_v", alpha_v); pop.set_hyper("tau_soft_update", tau); pop.set_hyper("gamma", gamma)
    agents = []
    for g in range(G):
        ag = ora.Ddpg(S, A, B, ha, hq, replay_cap=2000, inverting_grad=False, gamma=float(np.float32(gamma[g])), tau=float(np.float32(tau[g])),
                      alpha_a=float(np.float32(alpha_a[g])), alpha_v=float(np.float32(alpha_v[g])), seed=g, td3=True, batch_norm_actor=bna,
                      batch_norm_critic=bnq, test_net=bool(kw.get("test_net", False)), bn_adapt=bool(kw.get("bn_adapt", False)),
                      ewc=kw.get("ewc", -1.0), ewc_for_critic=kw.get("ewc_for_critic", 0))
        rows = synth_transitions(np.random.RandomState(1000 + g), 1000, S, A)
        pop.replay_load(g, rows)
        for row in rows.astype(np.float64):
            ag.insert(row[:S], row[S:S + A], row[2 * S + A], row[S + A:2 * S + A], row[2 * S + A + 1] != 0)
        agents.append(ag)
    # structured weights: [actor, critic, actor target, critic target, critic 2, critic 2 target]
    frozen = {}
    for net, twin in ((0, 2), (1, 3), (4, 5)):
        assert pop.num_params(net) == agents[0].nets[net].np
        w, fz = structured_weights(rng, agents[0].nets[net], 0.3, G)
        wt = (w + (~fz) * rng.normal(0, 0.02, w.shape)).astype(np.float32).astype(np.float64)
        pop.set_weights(net, w); pop.set_weights(twin, wt)
        frozen[net] = frozen[twin] = fz

    def sync_all():
        ws = {n: pop.weights(n) for n in range(6)}
        ad = {n: pop.adam(n) for n in (0, 1, 4)}
        for g, a in enumerate(agents):
            for n in range(6):
                a.nets[n].set_weights(ws[n][g])
            for n in (0, 1, 4):
                a.nets[n].set_adam(ad[n][0][g], ad[n][1][g], int(ad[n][2][g]))
        return ws, ad

    ws, _ = sync_all()
    for g, a in enumerate(agents):
        a.ann_testing.set_weights(ws[0][g])                                   # the acting copy as of the last policy step
    noise_kills = critic_kills_action_gradient(bnq)
    bern, scale = kw.get("bern", 0), kw.get("scale", 0)
    if bern:
        sg = np.round(rng.uniform(0.4, 1, (G, S)), 2); sg[:, ::3] = 1.0
        ag_ = np.round(rng.uniform(0.4, 1, (G, A)), 2); ag_[:, 0] = 1.0
        sg = sg.astype(np.float32).astype(np.float64); ag_ = ag_.astype(np.float32).astype(np.float64)
        pop.set_gates(sg, ag_)
    for upd in range(4):
        astep = np.full(G, upd % 2, np.uint8)
        ws_prev, ad_prev = sync_all()
        idx = np.random.RandomState(80 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        nz = np.clip(np.random.RandomState(90 + upd).normal(0, 0.2, (G, B, A)), -0.5, 0.5).astype(np.float32)
        if bern:
            pop.set_gate_mode(1, scale, 1, scale, seed=500 + upd)       # draw counter 0 for the coming update
            for g, a in enumerate(agents):
                F = lambda wv, site: gate_factors(wv, 1, scale, 500 + upd, 0, g, site, B)
                cat = lambda s_site, a_site: np.concatenate([F(sg[g], s_site), F(ag_[g], a_site)], 1)
                a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag_[g], ACTOR_T_OUT), B)
                a.qnn_target.push_row_gates(cat(CRITIC_T_S, CRITIC_T_A), None, B)
                a.qnn_target2.push_row_gates(cat(C2_T_S, C2_T_A), None, B)
                a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A_REPLAY), None, B)
                a.qnn2.push_row_gates(cat(C2_S, C2_A_REPLAY), None, B)
                if astep[g]:
                    a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag_[g], ACTOR_OUT), B)
                    if kw.get("bn_adapt") and bna != 0:
                        a.ann_testing.push_row_gates(F(sg[g], ACTOR_S), F(ag_[g], ACTOR_OUT), B)
                    a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A), None, B)
        pop.update(idx=idx, noise=nz, actor_step=astep)
        last = pop.last()
        for g, a in enumerate(agents):
            loss1, loss2, _ = a.update_td3(nz[g], actor_step=bool(astep[g]), idx=idx[g])
            assert all(n.row_gates_pending() == 0 for n in a.nets) and a.ann_testing.row_gates_pending() == 0
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL, (upd, g, rel_err(last["q_targets"][g], qt))
            assert abs(last["loss"][g] - loss1) <= TOL * abs(loss1) and abs(last["loss2"][g] - loss2) <= TOL * abs(loss2), (upd, g)
            if astep[g] and not noise_kills:
                assert rel_err(last["dact"][g], dact) <= TOL, (upd, g, rel_err(last["dact"][g], dact))
        for net in range(6):
            w = pop.weights(net)
            fz = frozen[net]
            for g, a in enumerate(agents):
                wo = a.nets[net].weights()
                if not astep[g] and net == 0:                                 # off the policy step the actor stands still ...
                    assert np.array_equal(w[g], ws_prev[net][g]), (upd, net, g)
                    continue
                if not astep[g] and net in (2, 3, 5):                         # ... and so do the targets' parameters; their TRAIN-phase
                    assert np.array_equal(w[g][~fz], ws_prev[net][g][~fz]), (upd, net, g)     # forwards still move their statistics
                    assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= TOL, ("target statistics", upd, net, g)
                    continue
                if noise_kills and net in (0, 2):
                    assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= TOL
                    continue
                assert not fz.any() or rel_err(w[g][fz], wo[fz]) <= TOL, ("statistics", upd, net, g, rel_err(w[g][fz], wo[fz]))
                wmax = np.max(np.abs(wo[~fz]))
                assert np.mean(np.abs(w[g][~fz] - wo[~fz]) <= TOL * wmax) >= 0.995, ("weights", upd, net, g)
        for net in (0, 1, 4):
            it = pop.adam(net)[2]
            for g in range(G):
                assert it[g] == ad_prev[net][2][g] + (1 if (net != 0 or astep[g]) else 0)
        if noise_kills and astep.all():       # the two actors stepped on different rounding noise: the oracle's acting copy (bn_adapt
            wa = pop.weights(0)               # reads it at the next policy step) follows the device's, like the six nets do
            for g, a in enumerate(agents):
                a.ann_testing.set_weights(wa[g])
    if bern:
        pop.close()
        return
    st = np.random.RandomState(2).uniform(-1, 1, (G, S)).astype(np.float32)
    act = pop.infer(st)                                                       # the last update was a policy step: the acting copy followed
    wa = pop.weights(0)
    for g, a in enumerate(agents):
        if noise_kills:                                                       # (the two actors stepped on different rounding noise)
            a.ann_testing.set_weights(wa[g])
        assert rel_err(act[g], a.ann_testing.forward(st[g].astype(np.float64))[0]) <= TOL
    pop.close()


@pytest.mark.parametrize("adapt", [True, False])
def test_fupd_loop_with_towers(ora, adapt):
    """force_more_update > 0 on tower nets (DeepQNAg.hpp:341, 415-420, 469-472): the passes of one end_episode run back to back,
    and ann_testing -- whose actions the critic scores under bn_adapt -- is the actor as of the END OF THE PREVIOUS end_episode in
    every one of them; it follows only after the loop.  Three passes per call, two calls, free-running inside a call (fp32 against
    fp64 through normalisations: 3e-4 norm-wise), teacher-forced between the calls."""
    G, B, n = 3, 64, 3
    import ddrl_amd
    ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(n_agents=1, batch_norm_actor=14, batch_norm_critic=11, bn_adapt=int(adapt), force_more_update=n - 1,
                                                replay_memory=500), init_seed=0).close()          # (refused with bn_adapt until round 3)
    pop, agents, frozen = make_bn_pair(ora, 14, 11, G=G, seed=31, bn_adapt=adapt, inverting_grad=True)
    st = np.random.RandomState(3).uniform(-1, 1, (G, 18)).astype(np.float32)
    for call in range(2):
        testing_before = [a.ann_testing.weights().copy() for a in agents]
        sync(pop, agents, testing=testing_before)
        idx = np.random.RandomState(70 + call).randint(0, 1000, (G, n, B)).astype(np.int32)
        pop.update_n(n, idx)
        for g, a in enumerate(agents):
            for k in range(n):
                a.update(idx[g, k], sync=False)
                assert np.array_equal(a.ann_testing.weights(), testing_before[g])        # untouched inside the loop
            a.sync_testing()
        for net in range(4):
            w = pop.weights(net)
            for g, a in enumerate(agents):
                assert rel_err(w[g], a.nets[net].weights()) <= 3e-4, (call, net, g, rel_err(w[g], a.nets[net].weights()))
        act = pop.infer(st)                                                               # the acting copy followed the actor
        for g, a in enumerate(agents):
            assert rel_err(act[g], a.ann_testing.forward(st[g].astype(np.float64))[0]) <= 3e-4
    pop.close()


@pytest.mark.parametrize("bna,bnq,for_critic", [(14, 11, 1), (7, 12, 0), (22, 19, 1)])
def test_ewc_lifecycle_with_towers(ora, bna, bnq, for_critic):
    """EWC (DODevMLP.hpp:604-758) on nets with towers: the EWC vectors hold the parameters with lr_mult != 0 only
    (number_of_parameters(true), :641), so a tower's running statistics have no Fisher entry, are not restored by ewc_setup
    (copyWeightsFrom(best, true), :616) and are not counted by ewc_cost's average (:679).  Teacher-forced like the other tower
    tests: Fisher accumulation over a first task, the development switch, three updates with the penalty live, the cost."""
    import oracle.oracle as o
    G, B = 3, 64
    pop, agents, frozen = make_bn_pair(ora, bna, bnq, G=G, seed=bna + bnq + 7, ewc=100.0, ewc_decay=0.9, ewc_fisher_beta=0.999,
                                       ewc_for_critic=for_critic, ewc_best_method=0)
    ewc_nets = (0, 1) if for_critic else (0,)
    noise = critic_kills_action_gradient(bnq)
    step = [0]

    def one_update(check):
        ws_prev = sync(pop, agents)
        idx = np.random.RandomState(40 + step[0]).randint(0, 1000, (G, B)).astype(np.int32)
        step[0] += 1
        pop.update(idx)
        for g, a in enumerate(agents):
            a.update(idx[g])
        if check:
            check_update(pop, agents, frozen, ws_prev, actor_noise=noise)

    for it in range(4):                                   # task 1: the running Fisher
        one_update(False)
    scores = np.array([1.0, 2.0, 3.0])
    pop.update_best_param(scores)
    for g, a in enumerate(agents):
        sync(pop, agents)
        a.ann.update_best_param(scores[g]); a.qnn.update_best_param(scores[g])
    for it in range(2):
        one_update(False)
    for net in ewc_nets:
        if net == 0 and noise:
            continue
        st = pop.ewc_state(net)
        for g, a in enumerate(agents):
            so = a.nets[net].ewc_state()
            assert np.all(st["fisher"][g][frozen[net]] == 0) and np.all(so["fisher"][frozen[net]] == 0)
            assert np.mean(np.abs(st["fisher"][g] - so["fisher"]) <= 1e-3 * np.max(so["fisher"])) >= 0.99, (net, g)
            assert abs(st["fisher_nbr"][g] - so["fisher_nbr"]) <= 1e-5 * so["fisher_nbr"]
    stats_before = [pop.weights(n)[:, frozen[n]] for n in (0, 1)]
    w_before = [pop.weights(n) for n in (0, 1)]
    pop.ewc_setup()                                       # the development switch
    for a in agents:
        a.ewc_setup()
    for net in ewc_nets:
        w = pop.weights(net)
        st = pop.ewc_state(net)
        assert np.array_equal(w[:, frozen[net]], stats_before[net])                    # the statistics stay ...
        assert np.array_equal(w[:, ~frozen[net]], st["wstar"][:, ~frozen[net]])         # ... the parameters are the best ones again
        assert not np.array_equal(w[:, ~frozen[net]], w_before[net][:, ~frozen[net]])
        for g, a in enumerate(agents):
            so = a.nets[net].ewc_state()
            assert so["active"] == 1
            assert np.all(st["fisher_hat"][g][frozen[net]] == 0)
            if not (net == 0 and noise):
                assert np.mean(np.abs(st["fisher_hat"][g] - so["fisher_hat"]) <= 1e-3 * np.max(so["fisher_hat"])) >= 0.99
            # from here on the oracle carries the device's (fp32) EWC state
            o.lib().ora_mlp_set_ewc(a.nets[net].h, st["fisher_hat"][g].ctypes.data, st["wstar"][g].ctypes.data)
    if not for_critic:
        assert np.array_equal(pop.weights(1), w_before[1])
    for it in range(3):                                   # task 2: the penalty live, the Fisher average starts again
        one_update(True)
        pop.ewc_decay_update()
        for a in agents:
            o.lib().ora_mlp_ewc_decay_update(a.ann.h); o.lib().ora_mlp_ewc_decay_update(a.qnn.h)
    sync(pop, agents)
    for net in ewc_nets:
        cost = pop.ewc_cost(net)
        for g, a in enumerate(agents):
            ref = a.nets[net].ewc_cost()
            assert ref > 0 and abs(cost[g] - ref) <= 1e-4 * abs(ref), (net, g, cost[g], ref)
    pop.close()


@pytest.mark.parametrize("kw,msg", [(dict(mini_batch_size=128), "mini_batch_size"), (dict(batch_norm_actor=1), "without scaling"),
                                    (dict(batch_norm_actor=3), "without bn"), (dict(batch_norm_actor=18), "critic layout"),
                                    (dict(hidden_unit_a=(200,)), "128")])
def test_refusals(kw, msg):
    import ddrl_amd
    base = dict(n_agents=1, batch_norm_actor=14, batch_norm_critic=11, replay_memory=500)
    base.update(kw)
    with pytest.raises(ddrl_amd.DdrlError, match=msg):
        ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(**base), init_seed=0)


def test_load_flat_dev_refuses_towers_and_follows_the_actor(ora):
    """ddrl_pop_load_flat_dev unpacks InnerProduct layers only: a net with towers is refused (its flat order carries the towers'
    statistics and parameters, which set_weights handles); a tower-free actor beside a critic with towers loads, and
    ann_testing follows it (DeepQNAg.hpp:469-472)"""
    import torch
    from ddrl_amd.lib import DdrlError
    pop, agents, _ = make_bn_pair(ora, 0, 7, G=2)
    npa = pop.num_params(0)
    w = np.random.RandomState(3).normal(0, 0.3, (2, npa)).astype(np.float32)
    d = torch.from_numpy(w).cuda()
    pop.load_flat_dev(d.data_ptr(), npa, net=0)
    assert np.array_equal(pop.weights(0).astype(np.float32), w)
    s = np.random.RandomState(4).uniform(-1, 1, (2, 18)).astype(np.float32)
    ref = pop.infer(s)
    pop.set_weights(0, w.astype(np.float64))                              # the host path refreshes ann_testing too
    assert np.array_equal(pop.infer(s), ref)
    with pytest.raises(DdrlError) as e:
        pop.load_flat_dev(d.data_ptr(), pop.num_params(1), net=1)
    assert "towers" in str(e.value)
