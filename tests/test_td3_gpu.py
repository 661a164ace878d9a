"""GPU parity: the fused gfx950 TD3 update (ddrl_pop_td3_update, through the C ABI) vs the fp64 CPU
oracle (oracle/td3_ref.c, restating TD3Ag.hpp:284-380) on identical seeds, indices and smoothing noise.

Tolerance as for DDPG (tests/ddpg_common.py): replay indices bit-exact; per update from identical
state, targets / losses / action gradients / Adam moments within 1e-5 norm-wise; weights consistent
with the device's own moments and within the bar on >= 99.9 % of the parameters.  The reference has
no test or fixture for TD3Ag: parity unpinned beyond the MLP known-answer tests the oracle is held to.
The smoothing noise is the reference's clock-seeded host stream: the tests feed both sides the same
draws (the device's own, read back, or host-made ones)."""
import numpy as np
import pytest

from ddpg_common import BETAS, close_frac, oracle_weights, rel_err, synth_transitions

pytestmark = pytest.mark.gpu
TOL = 1e-5
TRAIN = (0, 1, 4)           # ann, qnn, qnn2
ALL = range(6)


def make_td3(ora, G=4, S=18, A=6, B=64, hidden_a=(100, 100), hidden_q=(100, 100), n_replay=1000, replay_cap=2000,
             wscale=0.1, seed=0, force_generic=False, policy_noise=0.2, noise_clip=0.5, **ewc):
    import ddrl_amd
    rng = np.random.RandomState(seed)
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=B, replay_memory=replay_cap,
                              hidden_unit_a=hidden_a, hidden_unit_q=hidden_q, inverting_grad=0, td3=1,
                              policy_noise=policy_noise, noise_clip=noise_clip, policy_freq=2,
                              **{("ewc_best_param_method" if k == "ewc_best_method" else k): v for k, v in ewc.items()})
    pop = ddrl_amd.Td3Population(cfg, init_seed=seed)
    pop.set_option("force_generic", int(force_generic))
    alpha_a = 10 ** rng.uniform(-5, -2, G); alpha_v = 10 ** rng.uniform(-5, -2, G)
    tau = 10 ** rng.uniform(-4, -1, G); gamma = rng.uniform(0.9, 0.999, G)
    pop.set_hyper("alpha_a", alpha_a); pop.set_hyper("alpha_v", alpha_v)
    pop.set_hyper("tau_soft_update", tau); pop.set_hyper("gamma", gamma)
    seeds = (3000 + np.arange(G)).astype(np.uint32)
    pop.seed_replay(seeds)
    na, nq = pop.num_params(0), pop.num_params(1)
    assert pop.num_params(4) == nq and pop.num_params(5) == nq
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    wa, wq, wq2 = f32(rng.normal(0, wscale, (G, na))), f32(rng.normal(0, wscale, (G, nq))), f32(rng.normal(0, wscale, (G, nq)))
    for net, w in ((0, wa), (1, wq), (2, wa), (3, wq), (4, wq2), (5, wq2)):
        pop.set_weights(net, w)
    agents = []
    for g in range(G):
        ag = ora.Ddpg(S, A, B, hidden_a, hidden_q, replay_cap=replay_cap, inverting_grad=False,
                      gamma=float(np.float32(gamma[g])), tau=float(np.float32(tau[g])),
                      alpha_a=float(np.float32(alpha_a[g])), alpha_v=float(np.float32(alpha_v[g])),
                      seed=int(seeds[g]), td3=True, **ewc)
        for n, w in ((0, wa), (1, wq), (2, wa), (3, wq), (4, wq2), (5, wq2)):
            ag.nets[n].set_weights(w[g])
        rows = synth_transitions(np.random.RandomState(1000 + g + 97 * seed), n_replay, S, A)
        pop.replay_load(g, rows)
        for row in rows.astype(np.float64):
            ag.insert(row[:S], row[S:S + A], row[2 * S + A], row[S + A:2 * S + A], row[2 * S + A + 1] != 0)
        agents.append(ag)
    return pop, agents


def sync(pop, agents):
    ws = {n: pop.weights(n) for n in ALL}
    ad = {n: pop.adam(n) for n in TRAIN}
    for g, ag in enumerate(agents):
        for n in ALL:
            ag.nets[n].set_weights(ws[n][g])
        for n in TRAIN:
            ag.nets[n].set_adam(ad[n][0][g], ad[n][1][g], int(ad[n][2][g]))
    return ws, ad


def step_and_check(pop, agents, idx, noise, astep, tol=TOL, frac=0.999):
    """one teacher-forced update on both sides from synchronised state, held to the bar"""
    G = len(agents)
    ws_prev, ad_prev = sync(pop, agents)
    pop.update(idx=idx, noise=noise, actor_step=astep)
    last = pop.last()
    nz = last["noise"]
    astep = np.ones(G, bool) if astep is None else np.asarray(astep, bool)
    for g, ag in enumerate(agents):
        loss1, loss2, used = ag.update_td3(nz[g], actor_step=bool(astep[g]), idx=None if idx is None else idx[g])
        assert np.array_equal(used, last["idx"][g]), g                       # bit-exact replay indices
        qt, dact = ag.last()
        assert rel_err(last["q_targets"][g], qt) <= tol, ("targets", g)
        assert abs(last["loss"][g] - loss1) <= tol * abs(loss1) and abs(last["loss2"][g] - loss2) <= tol * abs(loss2), g
        if astep[g]:
            assert rel_err(last["dact"][g], dact) <= tol, ("dact", g)
    b1, b2 = BETAS[0]
    lrs = {0: [a.cfg.alpha_a for a in agents], 1: [a.cfg.alpha_v for a in agents], 4: [a.cfg.alpha_v for a in agents]}
    for net in TRAIN:
        m, v, it = pop.adam(net)
        w, wo = pop.weights(net), oracle_weights(agents, net)
        for g, ag in enumerate(agents):
            if net == 0 and not astep[g]:                                    # delayed policy update: nothing moves
                assert it[g] == ad_prev[0][2][g] and np.array_equal(w[g], ws_prev[0][g]) and np.array_equal(m[g], ad_prev[0][0][g])
                continue
            assert it[g] == ad_prev[net][2][g] + 1
            mo, vo = ag.nets[net].adam()[:2]
            assert rel_err(m[g], mo) <= tol or close_frac(m[g], mo, tol), ("m", net, g, rel_err(m[g], mo))
            assert rel_err(v[g], vo) <= 2 * tol or close_frac(v[g], vo, 2 * tol), ("v", net, g)
            t = int(it[g])
            lr_t = float(np.float32(lrs[net][g])) * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            expect = ws_prev[net][g] - lr_t * m[g] / (np.sqrt(v[g]) + 1e-8)
            wmax = np.max(np.abs(wo[g]))
            assert np.max(np.abs(w[g] - expect)) <= tol * wmax, ("adam-consistency", net, g)
            err = np.abs(w[g] - wo[g])
            assert np.mean(err <= tol * wmax) >= frac, ("weights", net, g, np.mean(err <= tol * wmax))
            assert np.max(err) <= 2 * lr_t * 1.001 + tol * wmax, ("step bound", net, g)
    for net in (2, 3, 5):
        w, wo = pop.weights(net), oracle_weights(agents, net)
        for g in range(G):
            if not astep[g]:                                                 # targets move with the policy step only (:368-370)
                assert np.array_equal(w[g], ws_prev[net][g]), ("target moved", net, g)
            else:
                wmax = np.max(np.abs(wo[g]))
                assert np.mean(np.abs(w[g] - wo[g]) <= tol * wmax) >= frac, ("target", net, g)
                assert not np.array_equal(w[g], ws_prev[net][g])
    return last


@pytest.mark.parametrize("hidden_a,hidden_q,generic", [
    ((100, 100), (100, 100), False),     # the compile-time "mid" instantiation
    ((100, 100), (100, 100), True),      # the run-time-shape kernel on the same inputs
    ((20,), (50, 7), False),
    ((128, 96, 33), (17, 128, 5), False),
])
def test_td3_update_matches_oracle(ora, hidden_a, hidden_q, generic):
    G = 4
    pop, agents = make_td3(ora, G=G, hidden_a=hidden_a, hidden_q=hidden_q, force_generic=generic)
    rng = np.random.RandomState(11)
    idx = rng.randint(0, 1000, (G, 64)).astype(np.int32)
    noise = np.clip(rng.normal(0, 0.2, (G, 64, 6)), -0.5, 0.5).astype(np.float32)
    last = step_and_check(pop, agents, idx, noise, [1, 0, 1, 0])
    assert np.array_equal(last["noise"], noise)
    # the smoothed target actions saturate at +-1 somewhere with this noise (the clip of :309 is exercised)
    step_and_check(pop, agents, idx[:, ::-1].copy(), 4 * noise, [0, 1, 1, 0])


@pytest.mark.parametrize("B,forced", [(160, False), (512, False), (48, True)])
def test_td3_large_minibatch(ora, B, forced):
    """minibatches above 64 rows (the reference's inis use 512): critic chunks for both critics, three applies,
    actor chunks for the agents on a policy step; `forced` runs the same kernels on a one-tile minibatch"""
    G = 3
    pop, agents = make_td3(ora, G=G, B=B, hidden_a=(64, 32), hidden_q=(48, 40), seed=6)
    if forced:
        pop.set_option("force_chunked", 1)
    rng = np.random.RandomState(3)
    for it, astep in enumerate(([1, 0, 1], [0, 0, 0], None)):
        idx = rng.randint(0, 1000, (G, B)).astype(np.int32) if it < 2 else None
        noise = np.clip(rng.normal(0, 0.3, (G, B, 6)), -0.5, 0.5).astype(np.float32) if it < 2 else None
        step_and_check(pop, agents, idx, noise, astep)


@pytest.mark.parametrize("B,hidden_a,hidden_q,forced", [
    (64, (160, 96), (144, 130), False),          # wider than the LDS-resident kernels: layer-by-layer launches
    (100, (400, 300), (400, 300), False),        # the TD3 paper's nets (BASELINE.md "large"), two chunks
    (48, (24, 16), (20, 12), True),              # the same launches forced on small nets
])
def test_td3_wide_nets_layer_by_layer(ora, B, hidden_a, hidden_q, forced):
    G = 2
    pop, agents = make_td3(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, seed=8, wscale=0.05)
    if forced:
        pop.set_option("force_wide", 1)
    rng = np.random.RandomState(4)
    for it, astep in enumerate(([1, 0], [0, 0], None)):
        idx = rng.randint(0, 1000, (G, B)).astype(np.int32) if it < 2 else None
        noise = np.clip(rng.normal(0, 0.3, (G, B, 6)), -0.5, 0.5).astype(np.float32) if it < 2 else None
        step_and_check(pop, agents, idx, noise, astep)


@pytest.mark.parametrize("G,B,hidden_a,hidden_q", [(13, 100, (160, 96), (144, 130)), (11, 160, (32, 16), (24, 16))])
def test_td3_xcd_order_many_agents(ora, G, B, hidden_a, hidden_q):
    """more workgroup groups than XCDs (xcd_decode: layer-by-layer launches, chunk launches), agents on and off a
    policy step interleaved: every agent against the oracle"""
    pop, agents = make_td3(ora, G=G, B=B, hidden_a=hidden_a, hidden_q=hidden_q, seed=12, wscale=0.05 if max(hidden_a) > 128 else 0.1)
    for astep in ((np.arange(G) % 3 != 1).astype(int).tolist(), None):
        step_and_check(pop, agents, None, None, astep)


def test_td3_free_indices_and_device_noise(ora):
    """device-drawn indices (bit-exact with the oracle's mt19937) and device-drawn smoothing noise,
    six updates with the policy step every second one"""
    G = 3
    pop, agents = make_td3(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), seed=2)
    seen = []
    for it in range(6):
        last = step_and_check(pop, agents, None, None, [it % 2 == 0] * G)
        nz = last["noise"]
        assert np.all(np.abs(nz) <= 0.5) and nz.std() > 0.1
        seen.append(nz.copy())
    assert not np.array_equal(seen[0], seen[1])                              # a fresh draw per update
    assert not np.array_equal(seen[0][0], seen[0][1])                        # and per agent


def test_td3_noise_distribution_and_seeding():
    """N(0, sigma) truncated to +-clip: bounds, moments against the closed forms, reproducible from the seed"""
    import ddrl_amd
    from scipy import stats
    G, B, A = 64, 64, 6
    outs = []
    for trial in range(2):
        cfg = ddrl_amd.DdpgConfig(n_agents=G, mini_batch_size=B, replay_memory=300, hidden_unit_a=(8,), hidden_unit_q=(8,),
                                  inverting_grad=0, td3=1, policy_noise=0.2, noise_clip=0.5)
        pop = ddrl_amd.Td3Population(cfg, init_seed=7)
        pop.seed_noise(np.arange(G, dtype=np.uint64) + 99)
        rows = synth_transitions(np.random.RandomState(1), 260, 18, 6)
        for g in range(G):
            pop.replay_load(g, rows)
        draws = []
        for _ in range(4):
            pop.update()
            draws.append(pop.last()["noise"].copy())
        outs.append(np.stack(draws))
        pop.close()
    assert np.array_equal(outs[0], outs[1])
    z = outs[0].ravel().astype(np.float64)
    assert np.all(np.abs(z) <= 0.5)
    a = 0.5 / 0.2
    want_std = 0.2 * np.sqrt(1 - 2 * a * stats.norm.pdf(a) / (2 * stats.norm.cdf(a) - 1))
    assert abs(z.mean()) < 4 * want_std / np.sqrt(z.size)
    assert abs(z.std() - want_std) < 0.01 * want_std
    # against the truncated normal's CDF
    d = stats.kstest(z[:20000], stats.truncnorm(-a, a, scale=0.2).cdf).statistic
    assert d < 0.02, d


def test_td3_ewc_penalty_live(ora):
    """EWC on the actor and both critics (DODevMLP behind TD3Ag<DODevMLP>): Fisher accumulation, the switch,
    then updates with the penalty live held to the bar"""
    import oracle.oracle as o
    G = 3
    kw = dict(ewc=100.0, ewc_decay=-1.0, ewc_fisher_beta=0.999, ewc_for_critic=1, ewc_best_method=0)
    pop, agents = make_td3(ora, G=G, hidden_a=(100, 100), hidden_q=(100, 100), seed=5, **kw)
    for it in range(3):
        step_and_check(pop, agents, None, None, None)
    scores = np.array([1.0, 2.0, 3.0])
    pop.update_best_param(scores)
    for g, ag in enumerate(agents):
        for net in TRAIN:
            ag.nets[net].update_best_param(scores[g])
    step_and_check(pop, agents, None, None, None)
    pop.ewc_setup()
    for net in TRAIN:
        st = pop.ewc_state(net)
        assert np.array_equal(pop.weights(net), st["wstar"])
        for g, ag in enumerate(agents):                                      # the oracle takes the device's anchors
            ag.nets[net].ewc_setup()
            assert ag.nets[net].ewc_state()["active"] == 1
            o.lib().ora_mlp_set_ewc(ag.nets[net].h, st["fisher_hat"][g].ctypes.data, st["wstar"][g].ctypes.data)
            assert st["fisher_hat"][g].max() == pytest.approx(1.0, rel=1e-5)
    for it in range(3):
        step_and_check(pop, agents, None, None, [it != 1] * G)
    for net in TRAIN:
        cost = pop.ewc_cost(net)
        sync(pop, agents)
        for g, ag in enumerate(agents):
            want = ag.nets[net].ewc_cost()
            assert want > 0 and abs(cost[g] - want) <= 1e-5 * want


def test_td3_entry_point_guards():
    import ddrl_amd
    cfg = ddrl_amd.DdpgConfig(n_agents=2, mini_batch_size=16, replay_memory=300, hidden_unit_a=(8,), hidden_unit_q=(8,))
    ddpg = ddrl_amd.DdpgPopulation(cfg)
    with pytest.raises(ddrl_amd.DdrlError):                                  # not a TD3 population
        ddrl_amd.Td3Population.update(ddpg)
    with pytest.raises(ddrl_amd.DdrlError):
        ddpg.weights(4)
    cfg = ddrl_amd.DdpgConfig(n_agents=2, mini_batch_size=16, replay_memory=300, hidden_unit_a=(8,), hidden_unit_q=(8,),
                              inverting_grad=0, td3=1)
    td3 = ddrl_amd.Td3Population(cfg)
    with pytest.raises(ddrl_amd.DdrlError):                                  # the DDPG update does not apply
        ddrl_amd.DdpgPopulation.update(td3)
    rows = synth_transitions(np.random.RandomState(1), 100, 18, 6)           # < 250 samples: the guard of TD3Ag.hpp:286
    td3.replay_load(0, rows); td3.replay_load(1, rows)
    w = td3.weights(1)
    td3.update()
    assert np.array_equal(td3.weights(1), w) and np.all(td3.adam(4)[2] == 0)
    for bad in (dict(hidden_unit_q=(2000, 300)), dict(nstep_return=3), dict(policy_freq=0)):
        c = ddrl_amd.DdpgConfig(n_agents=2, replay_memory=300, inverting_grad=0, td3=1, **bad)
        with pytest.raises(ddrl_amd.DdrlError):
            ddrl_amd.Td3Population(c)
