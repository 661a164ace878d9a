"""Probabilistic developmental gates (the fork's Developmental layer with probabilist = 0 and `scale`; DODevMLP.hpp:276-304,
heuristic 2 :444-467) on the device.

What the tree pins is the layer's pass rates and scaling (CaffeUtest.cpp:1048-1158): test_scaling_control restates that test
through the C ABI.  The layer's generator is the fork's (absent): which samples pass cannot be matched -- parity unpinned
beyond the rates.  Given the masks the device drew (its counter-based generator is restated here in numpy), the update
itself is checked against the CPU oracle with the same per-row masks: test_ddpg_update_given_masks, test_nfac_update_given_masks."""
import os

import numpy as np
import pytest

from ddpg_common import check_step, make_pair, rel_err
import test_nfac_gpu as nf

pytestmark = pytest.mark.gpu
TOL = 1e-5

# site ids of pop_internal.h (GS_*)
(ACTOR_T_S, CRITIC_T_S, ACTOR_T_OUT, CRITIC_T_A, CRITIC_S, CRITIC_A_REPLAY, ACTOR_S, ACTOR_OUT, CRITIC_A, INFER_S, INFER_OUT,
 BATCH_S, BATCH_A, NFAC_V_S, NFAC_V_S2, NFAC_A_S, NFAC_A_OUT) = range(17)
M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = np.asarray(x, np.uint64) & M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85ebca6b)) & M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xc2b2ae35)) & M32
    x ^= x >> np.uint64(16)
    return x


def gate_factors(w, bern, scale, seed, draw, agent, site, nrows):
    """ddrl_gate_factor of pop_internal.h for rows 0..nrows-1 and every dimension of the fp32 gate vector w."""
    w = np.asarray(w, np.float32)
    if not bern:
        return np.tile(w.astype(np.float64), (nrows, 1))
    h = mix32(np.uint64(seed) ^ np.uint64(0x9e3779b9))
    h = mix32(h ^ np.uint64(draw))
    h = mix32(h ^ ((np.uint64(agent) * np.uint64(0x632be5ab) + np.uint64(site)) & M32))
    rows = np.arange(nrows, dtype=np.uint64)[:, None]
    dims = np.arange(w.size, dtype=np.uint64)[None, :]
    k = (rows * np.uint64(0x27d4eb2f) + dims * np.uint64(0x165667b1) + np.uint64(1)) & M32
    h = mix32(h ^ k)
    u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    passed = u < w[None, :]
    val = (np.float32(1.0) / w if scale else np.ones_like(w)).astype(np.float64)
    return np.where(passed, val[None, :], 0.0)


@pytest.mark.parametrize("hidden_layers", [1, 2, 3])
def test_scaling_control(hidden_layers):
    """CaffeUtest.cpp:1048-1158 (DevelopmentalLayerScalingControl): 343 rows, actor 3 -> 8.. -> 4, gates on state dims 1, 2 and
    action dims 1, 2, scale on.  All weights 1: outputs identical to the ungated net, five times over.  Action weights 0.4 / 0.6:
    the output equals ungated / w for a fraction w +- 0.09 of the rows; ungated dimensions always equal."""
    import ddrl_amd
    G, S, A, B = 2, 3, 4, 343
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=16, replay_memory=500,
                              hidden_unit_a=(8,) * hidden_layers, hidden_unit_q=(8,), hidden_layer_type=1, actor_output_layer_type=0)
    plain = ddrl_amd.DdpgPopulation(cfg, init_seed=5)
    gated = ddrl_amd.DdpgPopulation(cfg, init_seed=5)
    rng = np.random.RandomState(hidden_layers)
    w = rng.normal(0, 0.5, (G, plain.num_params(0)))
    plain.set_weights(0, w); gated.set_weights(0, w)
    grid = np.stack(np.meshgrid(*[np.linspace(-1, 1, 7)] * 3, indexing="ij"), -1).reshape(B, 3)      # Combinaison::continuous, 7^3 points
    x = np.where(rng.uniform(0, 1, (G, B, 3)) < 0.4, rng.uniform(-1, 1, (G, B, 3)), grid[None]).astype(np.float32)
    ref = plain.forward_batch(0, x)
    sg, ag = np.ones((G, S)), np.ones((G, A))
    gated.set_gates(sg, ag)
    gated.set_gate_mode(st_bernoulli=True, st_scale=True, ac_bernoulli=True, ac_scale=True, seed=11)
    for forced in range(5):
        assert np.array_equal(gated.forward_batch(0, x), ref)
    ag[:, 1] = 0.4; ag[:, 2] = 0.6
    gated.set_gates(sg, ag)
    seen = []
    for forced in range(5):
        out = gated.forward_batch(0, x)
        seen.append(out)
        for g in range(G):
            assert np.array_equal(out[g, :, 0], ref[g, :, 0]) and np.array_equal(out[g, :, 3], ref[g, :, 3])
            for j, wj in ((1, 0.4), (2, 0.6)):
                want = ref[g, :, j].astype(np.float64) * float(np.float32(1) / np.float32(wj))
                fit = np.abs(want - out[g, :, j]) <= 1e-6 * np.maximum(1.0, np.abs(want))
                zero = out[g, :, j] == 0
                assert np.all(fit | zero)
                assert wj - 0.09 <= fit.mean() <= wj + 0.09, (forced, g, j, fit.mean())
    assert not np.array_equal(seen[0], seen[1])                    # a fresh draw per pass
    assert not np.array_equal(seen[0][0, :, 1] == 0, seen[0][1, :, 1] == 0)   # and per agent
    # state gates: w = 0.5 on dims 1, 2 -> per row each passes (x 2) or is zeroed; compare with the plain net on the masked input
    sg[:, 1:] = 0.5
    gated.set_gates(sg, np.ones((G, A)))
    gated.set_gate_mode(st_bernoulli=True, st_scale=True, ac_bernoulli=True, ac_scale=True, seed=3)    # draw counter back to 0
    out = gated.forward_batch(0, x)
    xm = x.copy()
    rate = []
    for g in range(G):
        f = gate_factors(sg[g], 1, 1, 3, 0, g, BATCH_S, B)
        xm[g] = (x[g].astype(np.float64) * f).astype(np.float32)
        rate.append((f[:, 1:] != 0).mean())
        assert set(np.unique(f[:, 1:])) <= {0.0, 2.0} and np.all(f[:, 0] == 1.0)
    assert rel_err(out, plain.forward_batch(0, xm)) <= TOL
    assert all(0.41 <= r <= 0.59 for r in rate), rate
    # batch-1 inference draws per call as well
    st = np.ones((G, S), np.float32)
    acts = np.stack([gated.infer(st) for _ in range(40)])
    assert len({a.tobytes() for a in acts}) > 1
    plain.close(); gated.close()


def test_bernoulli_with_binary_weights_is_the_deterministic_gate(ora):
    """w in {0, 1}: Bernoulli(w) passes never / always (scale 1 / 1), so the update is bit-identical to the deterministic gates."""
    G, S, A = 3, 18, 6
    rng = np.random.RandomState(1)
    sg = (rng.uniform(0, 1, (G, S)) < 0.6).astype(np.float64)
    ag = (rng.uniform(0, 1, (G, A)) < 0.6).astype(np.float64)
    idx = np.random.RandomState(5).randint(0, 1000, (G, 64)).astype(np.int32)
    res = []
    for bern in (0, 1):
        pop, _ = make_pair(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, seed=6)
        pop.set_gates(sg, ag)
        pop.set_gate_mode(bern, bern, bern, bern, seed=9)
        pop.update(idx)
        pop.update(idx)
        res.append([pop.weights(n) for n in range(4)] + [pop.last()["loss"]])
        pop.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("scale", [0, 1])
def test_ddpg_update_given_masks(ora, scale):
    """One DDPG update with Bernoulli gates against the oracle fed the per-row masks the device's generator produces: every
    layer instance (actor / critic, target / trained, replay action / policy action) has its own mask, the backward passes
    see their forward pass's mask."""
    G, S, A, B = 3, 18, 6, 64
    pop, agents = make_pair(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, seed=7)
    rng = np.random.RandomState(2)
    sg = np.round(rng.uniform(0.3, 1, (G, S)), 2); sg[:, ::3] = 1.0
    ag = np.round(rng.uniform(0.3, 1, (G, A)), 2); ag[:, 0] = 1.0
    sg = sg.astype(np.float32).astype(np.float64); ag = ag.astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    seed = 77
    for upd in range(2):
        pop.set_gate_mode(1, scale, 1, scale, seed=seed + upd)          # draw counter 0 for the coming update
        ws_prev = [pop.weights(n) for n in range(4)]
        idx = np.random.RandomState(5 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        for g, a in enumerate(agents):
            F = lambda wv, site: gate_factors(wv, 1, scale, seed + upd, 0, g, site, B)
            a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag[g], ACTOR_T_OUT), B)
            a.qnn_target.push_row_gates(np.concatenate([F(sg[g], CRITIC_T_S), F(ag[g], CRITIC_T_A)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A_REPLAY)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A)], 1), None, B)
            a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            assert all(n.row_gates_pending() == 0 for n in (a.ann, a.qnn, a.ann_target, a.qnn_target))
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            assert rel_err(last["dact"][g], dact) <= TOL
        hyper = (np.array([a.cfg.alpha_a for a in agents]), np.array([a.cfg.alpha_v for a in agents]))
        check_step(pop, agents, ws_prev, hyper, 0, TOL)
    pop.close()


def test_ddpg_update_n_draws_per_update(ora):
    """a launch of several updates: update u draws with counter draw + u, the next launch carries on"""
    G, S, A, B = 2, 18, 6, 64
    rng = np.random.RandomState(3)
    sg = np.full((G, S), 0.5); ag = np.full((G, A), 0.75)
    idx = rng.randint(0, 1000, (G, 4, B)).astype(np.int32)
    res = []
    for split in (False, True):
        pop, _ = make_pair(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), wscale=0.1, seed=8)
        pop.set_gates(sg, ag)
        pop.set_gate_mode(1, 1, 1, 0, seed=4)
        if split:
            for u in range(4):
                pop.update(np.ascontiguousarray(idx[:, u]))
        else:
            pop.update_n(4, idx)
        res.append([pop.weights(n) for n in range(4)])
        pop.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_nfac_update_given_masks(ora):
    S, A, G, maxT = 18, 6, 3, 200
    pop, agents, lrs = nf.make_pair(ora, G, S, A, (32, 32), (32, 32), maxT, seed=4, update_critic_first=0, lambda_=0.9, gae=1,
                                    number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1)
    rng = np.random.RandomState(3)
    sg = np.round(rng.uniform(0.3, 1, (G, S)), 2).astype(np.float32).astype(np.float64); sg[:, 1::4] = 1.0
    ag = np.round(rng.uniform(0.3, 1, (G, A)), 2).astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    rng = np.random.RandomState(8)
    for upd in range(2):
        seed = 21 + upd
        pop.set_gate_mode(1, 1, 1, 0, seed=seed)
        lens = [200, 77, 130]
        trajs = nf.load_trajs(pop, rng, lens, S, A)
        ws_prev, ad_prev = nf.sync_oracle(pop, agents)
        it_prev = [ad_prev[0][2].copy(), ad_prev[1][2].copy()]
        for g, a in enumerate(agents):
            T = lens[g]
            F = lambda wv, sc, draw, site: gate_factors(wv, 1, sc, seed, draw, g, site, T)
            # end_episode: V(s'), V(s) [launch 0]; actor step [1]; update_critic: V(s'), V(s) [2]; critic step [3]
            a.vnn.push_row_gates(F(sg[g], 1, 0, NFAC_V_S2), None, T)
            a.vnn.push_row_gates(F(sg[g], 1, 0, NFAC_V_S), None, T)
            a.ann.push_row_gates(F(sg[g], 1, 1, NFAC_A_S), F(ag[g], 0, 1, NFAC_A_OUT), T)
            a.vnn.push_row_gates(F(sg[g], 1, 2, NFAC_V_S2), None, T)
            a.vnn.push_row_gates(F(sg[g], 1, 2, NFAC_V_S), None, T)
            a.vnn.push_row_gates(F(sg[g], 1, 3, NFAC_V_S), None, T)
        pop.update()
        last = pop.last()
        ns = []
        for g, a in enumerate(agents):
            n, deltas = nf.oracle_update(a, *trajs[g], S, A)
            ns.append(n)
            assert a.vnn.row_gates_pending() == 0 and a.ann.row_gates_pending() == (0 if n > 0 else 1)
            assert int(last["n_pos"][g]) == n
            assert rel_err(last["deltas"][g, :lens[g]], deltas) <= TOL
        assert all(n > 0 for n in ns)
        nf.check_nets(pop, agents, ws_prev, it_prev, lrs, 0, [[n > 0 for n in ns], [True] * G])
    pop.close()


def test_served_on_nets_with_towers(ora):
    """nets with BatchNorm towers take the Bernoulli mode too: inference, forward_batch and the update (parity against the oracle
    given the masks: tests/test_bn_gpu.py::test_ddpg_update_with_towers_given_masks, test_td3_update_with_towers[bern])"""
    pop, _ = make_pair(ora, G=2, hidden_a=(20, 10), hidden_q=(20, 10), n_replay=300, batch_norm_actor=14, batch_norm_critic=11)
    pop.set_gate_mode(1, 0, 1, 0, seed=1)
    pop.set_gates(np.full((2, 18), 0.7), np.full((2, 6), 0.7))
    w0 = pop.weights(0)
    pop.update()
    w1 = pop.weights(0)
    assert np.all(np.isfinite(w1)) and not np.array_equal(w0, w1)
    pop.close()


@pytest.mark.parametrize("B,hidden_a,hidden_q,scale", [(128, (64, 32), (48, 40), 1), (160, (64, 32), (48, 40), 0),
                                                       (64, (200, 100), (150, 130), 1), (96, (40, 160), (140, 20), 0)])
def test_ddpg_chunked_and_wide_updates_given_masks(ora, B, hidden_a, hidden_q, scale):
    """Minibatches of more than 64 rows (chunk + apply kernels, kernels_ddpg_big.hip) and layers wider than 128 (layer-by-layer
    launches, kernels_ddpg_wide.hip) with Bernoulli gates: the masks are drawn per MINIBATCH row, so the oracle fed the
    generator's masks for rows 0 .. B-1 must agree whatever the chunking."""
    G, S, A = 2, 18, 6
    pop, agents = make_pair(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, wscale=0.1 if max(hidden_a + hidden_q) <= 128 else 0.05, seed=11)
    rng = np.random.RandomState(6)
    sg = np.round(rng.uniform(0.3, 1, (G, S)), 2); sg[:, ::3] = 1.0
    ag = np.round(rng.uniform(0.3, 1, (G, A)), 2); ag[:, 0] = 1.0
    sg = sg.astype(np.float32).astype(np.float64); ag = ag.astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    seed = 91
    for upd in range(2):
        pop.set_gate_mode(1, scale, 1, scale, seed=seed + upd)
        ws_prev = [pop.weights(n) for n in range(4)]
        idx = np.random.RandomState(15 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        for g, a in enumerate(agents):
            F = lambda wv, site: gate_factors(wv, 1, scale, seed + upd, 0, g, site, B)
            a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag[g], ACTOR_T_OUT), B)
            a.qnn_target.push_row_gates(np.concatenate([F(sg[g], CRITIC_T_S), F(ag[g], CRITIC_T_A)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A_REPLAY)], 1), None, B)
            a.qnn.push_row_gates(np.concatenate([F(sg[g], CRITIC_S), F(ag[g], CRITIC_A)], 1), None, B)
            a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
        pop.update(idx)
        last = pop.last()
        for g, a in enumerate(agents):
            loss, _ = a.update(idx[g])
            assert all(n.row_gates_pending() == 0 for n in (a.ann, a.qnn, a.ann_target, a.qnn_target))
            qt, dact = a.last()
            assert rel_err(last["q_targets"][g], qt) <= TOL
            assert abs(last["loss"][g] - loss) <= TOL * abs(loss)
            assert rel_err(last["dact"][g], dact) <= TOL
        hyper = (np.array([a.cfg.alpha_a for a in agents]), np.array([a.cfg.alpha_v for a in agents]))
        check_step(pop, agents, ws_prev, hyper, 0, TOL)
    pop.close()


HEUR2 = """[devnn]
st_scale={scale}
ac_scale=false
st_probabilistic=0
ac_probabilistic=0
st_control=1:3
ac_control=2
heuristic=2
heuristic_linearcoef=0.25:1.0:0.1
ewc=-1
"""


def test_heuristic2_linear_ramp(tmp_path):
    """DODevMLP::develop, heuristic 2 (DODevMLP.hpp:444-467): at every learning episode gate i = episode x coef_i (1 when
    coef_i >= 1); Bernoulli gates with those pass probabilities (capped at 1 here)."""
    import ddrl_amd
    import test_dev_schedule_gpu as ds
    p = os.path.join(str(tmp_path), "h2.ini")
    with open(p, "w") as f:
        f.write(ds.AGENT + HEUR2.format(scale="true"))
    G, S, A = 2, 5, 3
    ag = ddrl_amd.DdpgAgents(p, G, S, A, seed=1)
    rng = np.random.RandomState(0)
    sg, agt = ag.pop.gates()
    assert np.array_equal(sg, np.tile([1, 0, 1, 0, 1], (G, 1))) and np.array_equal(agt, np.tile([1, 1, 0], (G, 1)))
    learn = np.ones(G, np.uint8)
    for e in range(6):
        ds.episode(ag, rng, learn, steps=130)
        sg, agt = ag.pop.gates()
        exp_s = [1, min(1.0, 0.25 * e), 1, 1.0, 1]
        exp_a = [1, 1, min(1.0, 0.1 * e)]
        assert np.allclose(sg, np.tile(exp_s, (G, 1)), atol=1e-7) and np.allclose(agt, np.tile(exp_a, (G, 1)), atol=1e-7), e
        ag.end_episode(learn)
        assert all(np.all(np.isfinite(ag.pop.weights(n))) for n in range(4))
    ag.close()
    # consistency checks of DODevMLP::exploit (:94-97, 192-200)
    for text, msg in ((HEUR2.format(scale="true").replace("st_probabilistic=0", "st_probabilistic=1"), "probabilistic should be 0"),
                      (HEUR2.format(scale="false") + "reset_learning_algo=true\n", "progressive"),
                      (HEUR2.format(scale="false").replace("0.25:1.0:0.1", "0.25:1.0"), "heuristic_linearcoef")):
        with open(p, "w") as f:
            f.write(ds.AGENT + text)
        with pytest.raises(ddrl_amd.DdrlError, match=msg):
            ddrl_amd.DdpgAgents(p, G, S, A)


@pytest.mark.parametrize("scale", [0, 1])
def test_td3_update_given_masks(ora, scale):
    """The one-tile TD3 update with Bernoulli gates against the oracle fed the device's per-row masks: TD3's second critic and
    its target have Developmental layers of their own (sites GS_CRITIC2_*), a policy step runs the first critic once more, the
    delayed (critic-only) steps never touch the actor's layers."""
    import test_td3_gpu as t3
    G, S, A, B = 3, 18, 6, 64
    C2_T_S, C2_T_A, C2_S, C2_A_REPLAY = 17, 18, 19, 20
    pop, agents = t3.make_td3(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), seed=9)
    rng = np.random.RandomState(4)
    sg = np.round(rng.uniform(0.3, 1, (G, S)), 2); sg[:, ::3] = 1.0
    ag = np.round(rng.uniform(0.3, 1, (G, A)), 2); ag[:, 0] = 1.0
    sg = sg.astype(np.float32).astype(np.float64); ag = ag.astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    seed = 31
    for upd, astep in enumerate([np.array([1, 1, 1], np.uint8), np.array([0, 1, 0], np.uint8)]):
        pop.set_gate_mode(1, scale, 1, scale, seed=seed + upd)          # draw counter 0 for the coming update
        idx = np.random.RandomState(50 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        noise = np.random.RandomState(60 + upd).normal(0, 0.1, (G, B, A)).astype(np.float32)
        for g, a in enumerate(agents):
            F = lambda wv, site: gate_factors(wv, 1, scale, seed + upd, 0, g, site, B)
            cat = lambda s_site, a_site: np.concatenate([F(sg[g], s_site), F(ag[g], a_site)], 1)
            a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag[g], ACTOR_T_OUT), B)
            a.qnn_target.push_row_gates(cat(CRITIC_T_S, CRITIC_T_A), None, B)
            a.qnn_target2.push_row_gates(cat(C2_T_S, C2_T_A), None, B)
            a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A_REPLAY), None, B)
            a.qnn2.push_row_gates(cat(C2_S, C2_A_REPLAY), None, B)
            if astep[g]:
                a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
                a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A), None, B)
        t3.step_and_check(pop, agents, idx, noise, astep)
        for a in agents:
            assert all(n.row_gates_pending() == 0 for n in a.nets)
    # w in {0, 1}: the Bernoulli gates are the deterministic ones, bit for bit
    res = []
    for bern in (0, 1):
        p, _ = t3.make_td3(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), seed=9)
        p.set_gates((sg > 0.6).astype(np.float64), (ag > 0.6).astype(np.float64))
        p.set_gate_mode(bern, bern, bern, bern, seed=5)
        for upd in range(2):
            p.update(idx=idx, noise=noise, actor_step=np.array([1, upd, 1], np.uint8))
        res.append([p.weights(n) for n in (0, 1, 2, 3, 4, 5)])
        p.close()
    for x, y in zip(*res):
        assert np.array_equal(x, y)
    pop.close()


@pytest.mark.parametrize("B,hidden_a,hidden_q", [(128, (64, 32), (48, 40)), (160, (64, 32), (48, 40)), (96, (200, 100), (150, 130))])
def test_td3_chunked_and_wide_updates_given_masks(ora, B, hidden_a, hidden_q):
    """TD3 over several minibatch tiles (ddpg_chunk_kernel<.., TD3>) and over layers wider than 128 (layer-by-layer launches) with
    Bernoulli gates, against the oracle fed the device's masks"""
    import test_td3_gpu as t3
    G, S, A, scale = 2, 18, 6, 1
    C2_T_S, C2_T_A, C2_S, C2_A_REPLAY = 17, 18, 19, 20
    pop, agents = t3.make_td3(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, seed=6, wscale=0.1 if max(hidden_a + hidden_q) <= 128 else 0.05)
    rng = np.random.RandomState(8)
    sg = np.round(rng.uniform(0.3, 1, (G, S)), 2); sg[:, ::3] = 1.0
    ag = np.round(rng.uniform(0.3, 1, (G, A)), 2); ag[:, 0] = 1.0
    sg = sg.astype(np.float32).astype(np.float64); ag = ag.astype(np.float32).astype(np.float64)
    pop.set_gates(sg, ag)
    seed = 131
    for upd, astep in enumerate([np.array([1, 1], np.uint8), np.array([0, 1], np.uint8)]):
        pop.set_gate_mode(1, scale, 1, scale, seed=seed + upd)
        idx = np.random.RandomState(70 + upd).randint(0, 1000, (G, B)).astype(np.int32)
        noise = np.random.RandomState(80 + upd).normal(0, 0.1, (G, B, A)).astype(np.float32)
        for g, a in enumerate(agents):
            F = lambda wv, site: gate_factors(wv, 1, scale, seed + upd, 0, g, site, B)
            cat = lambda s_site, a_site: np.concatenate([F(sg[g], s_site), F(ag[g], a_site)], 1)
            a.ann_target.push_row_gates(F(sg[g], ACTOR_T_S), F(ag[g], ACTOR_T_OUT), B)
            a.qnn_target.push_row_gates(cat(CRITIC_T_S, CRITIC_T_A), None, B)
            a.qnn_target2.push_row_gates(cat(C2_T_S, C2_T_A), None, B)
            a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A_REPLAY), None, B)
            a.qnn2.push_row_gates(cat(C2_S, C2_A_REPLAY), None, B)
            if astep[g]:
                a.ann.push_row_gates(F(sg[g], ACTOR_S), F(ag[g], ACTOR_OUT), B)
                a.qnn.push_row_gates(cat(CRITIC_S, CRITIC_A), None, B)
        t3.step_and_check(pop, agents, idx, noise, astep)
        for a in agents:
            assert all(n.row_gates_pending() == 0 for n in a.nets)
    pop.close()
