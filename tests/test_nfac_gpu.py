"""GPU parity of the NFAC(lambda)-V path (ddrl_nfac_*, kernels_nfac.hip) against oracle/nfac_ref.c,
through the C ABI.  No reference test pins this arithmetic: parity unpinned beyond the
MLP known-answer tests; the oracle is the fp64 restatement of OfflineCaclaAg.hpp:242-595.

Bar (same reasoning as tests/ddpg_common.py): with one Adam step per fit the device and the oracle
start every update from identical state (the oracle is resynced to the device's fp32 state), and
advantages, gradients (Adam m, v) and weights must agree to 1e-5 norm-wise; Adam-consistency of
the device weights with its own m, v; integer outputs (n, Solver::iter, trajectory bookkeeping)
exact.  The multi-iteration configuration (10 x 10 critic steps, 10 actor steps per update) is a
free-running fp32-vs-fp64 comparison and is held to 1e-3 on >= 99 % of the weights.
"""
import numpy as np
import pytest

from ddpg_common import BETAS, close_frac, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def synth_traj(rng, T, S, A, n_ep):
    """SURVEY 8(d) config 3 style rollouts: s_t a random walk in [-1,1]^S, r ~ N(0,1)."""
    s = np.clip(np.cumsum(rng.normal(0, 0.1, (T + 1, S)), axis=0), -1, 1)
    pure_a = np.tanh(rng.normal(0, 0.7, (T, A)))
    a = np.clip(pure_a + rng.normal(0, 0.2, (T, A)), -1, 1)
    r = rng.normal(0, 1, T)
    goal = (rng.uniform(0, 1, T) < 0.02)
    cuts = sorted(set(rng.choice(np.arange(1, T), size=min(n_ep - 1, T - 1), replace=False).tolist())) if n_ep > 1 and T > 1 else []
    ends = np.asarray(cuts + [T], np.int32)
    rows = np.concatenate([s[:-1], pure_a, a, s[1:], r[:, None], goal[:, None].astype(np.float64)], axis=1).astype(np.float32)
    return rows, ends


def make_pair(ora, G, S, A, hidden_a, hidden_v, maxT, seed=0, wscale=0.1, **kw):
    import ddrl_amd
    rng = np.random.RandomState(seed)
    cfg = ddrl_amd.NfacConfig(n_agents=G, nb_sensors=S, nb_motors=A, hidden_unit_a=hidden_a, hidden_unit_v=hidden_v,
                              max_trajectory=maxT, **{("lambda" if k == "lambda_" else k): v for k, v in kw.items()})
    pop = ddrl_amd.NfacPopulation(cfg, init_seed=seed)
    alpha_a = 10 ** rng.uniform(-4, -2.5, G); alpha_v = 10 ** rng.uniform(-4, -2.5, G)
    gamma = rng.uniform(0.9, 0.999, G)
    pop.set_hyper("alpha_a", alpha_a); pop.set_hyper("alpha_v", alpha_v); pop.set_hyper("gamma", gamma)
    na, nv = pop.num_params(0), pop.num_params(1)
    wa = rng.normal(0, wscale, (G, na)).astype(np.float32).astype(np.float64)
    wv = rng.normal(0, wscale, (G, nv)).astype(np.float32).astype(np.float64)
    pop.set_weights(0, wa); pop.set_weights(1, wv)
    okw = dict(hidden_act=cfg.get("hidden_layer_type"), actor_out_act=cfg.get("actor_output_layer_type"),
               momentum=cfg.get("momentum"), update_delta_neg=cfg.get("update_delta_neg"),
               update_critic_first=cfg.get("update_critic_first"), gae=cfg.get("gae"),
               inverting_gradient=cfg.get("inverting_gradient"),
               number_fitted_iteration=cfg.get("number_fitted_iteration"),
               stoch_iter_actor=cfg.get("stoch_iter_actor"), stoch_iter_critic=cfg.get("stoch_iter_critic"),
               lambda_=cfg.get("lambda"), ewc=cfg.get("ewc"), ewc_decay=cfg.get("ewc_decay"),
               ewc_fisher_beta=cfg.get("ewc_fisher_beta"), ewc_for_critic=cfg.get("ewc_for_critic"),
               ewc_best_method=cfg.get("ewc_best_param_method"),
               vnn_from_scratch=cfg.get("vnn_from_scratch"), corrected_update_ac=cfg.get("corrected_update_ac"),
               corrected_update_ac_factor=cfg.get("corrected_update_ac_factor"))
    agents = []
    for g in range(G):
        ag = ora.Nfac(S, A, hidden_a, hidden_v, gamma=float(np.float32(gamma[g])),
                      alpha_a=float(np.float32(alpha_a[g])), alpha_v=float(np.float32(alpha_v[g])), **okw)
        ag.ann.set_weights(wa[g]); ag.vnn.set_weights(wv[g])
        agents.append(ag)
    return pop, agents, (alpha_a, alpha_v)


def load_trajs(pop, rng, lens, S, A, n_ep=3):
    trajs = []
    for g, T in enumerate(lens):
        rows, ends = synth_traj(rng, T, S, A, n_ep) if T > 0 else (np.zeros((0, 2 * S + 2 * A + 2), np.float32), np.asarray([0], np.int32))
        pop.traj_load(g, rows, ends)
        trajs.append((rows.astype(np.float64), ends))
    return trajs


def oracle_update(ag, rows, ends, S, A):
    s, pa, a = rows[:, :S], rows[:, S:S + A], rows[:, S + A:S + 2 * A]
    s2, r, goal = rows[:, S + 2 * A:2 * S + 2 * A], rows[:, 2 * S + 2 * A], rows[:, 2 * S + 2 * A + 1] != 0
    return ag.update(s, pa, a, s2, r, goal, ends)


def sync_oracle(pop, agents):
    ws = [pop.weights(n) for n in (0, 1)]
    ad = [pop.adam(n) for n in (0, 1)]
    for g, ag in enumerate(agents):
        for n in (0, 1):
            ag.nets[n].set_weights(ws[n][g])
            ag.nets[n].set_adam(ad[n][0][g], ad[n][1][g], int(ad[n][2][g]))
    return ws, ad


def check_nets(pop, agents, ws_prev, it_prev, lrs, momentum, stepped, tol=TOL, frac=0.999):
    """one Adam step on net `n` for the agents with stepped[n][g]; the others must be untouched"""
    b1, b2 = BETAS[momentum]
    for net in (0, 1):
        m, v, it = pop.adam(net)
        w = pop.weights(net)
        for g, ag in enumerate(agents):
            assert int(it[g]) == ag.nets[net].iter() == it_prev[net][g] + (1 if stepped[net][g] else 0), ("iter", net, g)
            if not stepped[net][g]:
                assert np.array_equal(w[g], ws_prev[net][g]), ("untouched", net, g)
                continue
            mo, vo = ag.nets[net].adam()[:2]
            wo = ag.nets[net].weights()
            assert rel_err(m[g], mo) <= tol or close_frac(m[g], mo, tol), ("m", net, g, rel_err(m[g], mo))
            assert rel_err(v[g], vo) <= 2 * tol or close_frac(v[g], vo, 2 * tol), ("v", net, g, rel_err(v[g], vo))
            t = int(it[g])
            lr_t = float(np.float32(lrs[net][g])) * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            expect = ws_prev[net][g] - lr_t * m[g] / (np.sqrt(v[g]) + 1e-8)
            wmax = np.max(np.abs(wo))
            assert np.max(np.abs(w[g] - expect)) <= tol * wmax, ("adam-consistency", net, g)
            err = np.abs(w[g] - wo)
            assert np.mean(err <= tol * wmax) >= frac, ("weights", net, g, np.mean(err <= tol * wmax))
            assert np.max(err) <= 2 * lr_t * 1.001 + tol * wmax, ("step bound", net, g, np.max(err))


CASES = {
    # name: (S, A, hidden_a, hidden_v, kwargs)
    "config3": (45, 17, (64, 64), (64, 64), dict(lambda_=0.9, gae=1)),
    "small_td0": (4, 1, (20,), (50, 7), dict(lambda_=-1.0, gae=0, update_delta_neg=1)),
    "ragged3": (18, 6, (33, 17, 9), (40, 21), dict(lambda_=0.6, gae=0, inverting_gradient=1, hidden_layer_type=2)),
    "momentum1": (18, 6, (32, 32), (32, 32), dict(lambda_=0.95, gae=1, momentum=1, actor_output_layer_type=0, inverting_gradient=1)),
    # agent.corrected_update_ac (OfflineCaclaAg.hpp:164-173, 561-563): diff = -(target - pi) * factor, the inverting gradient is skipped
    "corrected": (18, 6, (32, 32), (32, 32), dict(lambda_=0.9, gae=1, corrected_update_ac=1, corrected_update_ac_factor=0.35, inverting_gradient=1)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_single_step_updates_teacher_forced(ora, name):
    S, A, ha, hv, kw = CASES[name]
    G, maxT = 5, 400
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=3, update_critic_first=0,
                                 number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1, **kw)
    rng = np.random.RandomState(11)
    momentum = kw.get("momentum", 0)
    for upd in range(4):
        lens = [400, 64, 1, 130, 0] if upd == 0 else list(rng.randint(1, maxT + 1, G))   # ragged, empty, full
        trajs = load_trajs(pop, rng, lens, S, A, n_ep=1 + upd)
        assert list(pop.traj_len()) == lens
        ws_prev, ad_prev = sync_oracle(pop, agents)
        it_prev = [ad_prev[0][2].copy(), ad_prev[1][2].copy()]
        pop.update()
        last = pop.last()
        stepped = [[False] * G, [False] * G]
        for g, ag in enumerate(agents):
            rows, ends = trajs[g]
            n, deltas = oracle_update(ag, rows, ends, S, A)
            T = lens[g]
            assert int(last["n_pos"][g]) == n, (upd, g)
            if T:
                assert rel_err(last["deltas"][g, :T], deltas) <= TOL, (upd, g, rel_err(last["deltas"][g, :T], deltas))
            stepped[0][g] = T > 0 and n > 0
            stepped[1][g] = T > 0
        check_nets(pop, agents, ws_prev, it_prev, lrs, momentum, stepped)
        assert list(pop.traj_len()) == [0] * G           # trajectory.clear()


def test_long_trajectory_scan_with_flags_in_memory(ora):
    """max_trajectory at its limit: beyond 32 358 rows the scan kernel's LDS holds the deltas only and the backward walk tests the
    episode-end flags in memory (nfac_scan_launch); shorter launches keep the flags beside the deltas.  Same deltas either way."""
    S, A, ha, hv = 6, 2, (16,), (16,)
    G, maxT = 2, 32768
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=9, update_critic_first=0, lambda_=0.9, gae=1,
                                 number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1)
    rng = np.random.RandomState(4)
    for lens in ([32500, 300], [900, 32000]):             # flags in memory, flags in LDS
        trajs = load_trajs(pop, rng, lens, S, A, n_ep=7)
        sync_oracle(pop, agents)
        pop.update()
        last = pop.last()
        for g, ag in enumerate(agents):
            n, deltas = oracle_update(ag, *trajs[g], S, A)
            assert rel_err(last["deltas"][g, :lens[g]], deltas) <= TOL, (lens, g, rel_err(last["deltas"][g, :lens[g]], deltas))
            assert abs(int(last["n_pos"][g]) - n) <= max(1, lens[g] // 1000)      # deltas within rounding of 0 may fall either side
    pop.close()


def test_critic_first_single_steps(ora):
    """update_critic_first = 1: the advantages are taken from the critic AFTER its step."""
    S, A, ha, hv = 18, 6, (32, 32), (32, 32)
    G, maxT = 3, 256
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=5, update_critic_first=1, lambda_=0.9, gae=1,
                                 number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1)
    rng = np.random.RandomState(2)
    lens = [256, 100, 77]
    trajs = load_trajs(pop, rng, lens, S, A)
    ws_prev, ad_prev = sync_oracle(pop, agents)
    pop.update()
    last = pop.last()
    for g, ag in enumerate(agents):
        n, deltas = oracle_update(ag, *trajs[g], S, A)
        # the critic moved by +-lr per weight first; a sign flip on a ~0 gradient shifts V by O(lr)
        assert rel_err(last["deltas"][g, :lens[g]], deltas) <= 1e-3
        assert abs(int(last["n_pos"][g]) - n) <= max(1, lens[g] // 100)
    for net in (0, 1):
        w = pop.weights(net)
        for g, ag in enumerate(agents):
            wo = ag.nets[net].weights()
            assert close_frac(w[g], wo, 1e-4, 0.99, 1e-1), (net, g)


def test_masked_agents_keep_everything(ora):
    S, A = 8, 2
    pop, agents, _ = make_pair(ora, 4, S, A, (16,), (16,), 128, seed=1, number_fitted_iteration=1,
                               stoch_iter_actor=1, stoch_iter_critic=1)
    rng = np.random.RandomState(0)
    load_trajs(pop, rng, [100, 50, 128, 10], S, A)
    w0 = [pop.weights(n) for n in (0, 1)]
    mask = np.array([1, 0, 1, 0], np.uint8)
    pop.update(mask)
    for n in (0, 1):
        w = pop.weights(n)
        assert np.array_equal(w[1], w0[n][1]) and np.array_equal(w[3], w0[n][3])
        assert not np.array_equal(w[2], w0[n][2])
    assert list(pop.traj_len()) == [0, 50, 0, 10]
    assert list(pop.adam(1)[2]) == [1, 0, 1, 0]


def test_append_matches_load(ora):
    """trajectory.push_back per step + trajectory_end_points per episode == bulk load"""
    S, A, G = 6, 2, 2
    pops = []
    rng = np.random.RandomState(4)
    rows = [synth_traj(rng, 70, S, A, 1)[0] for _ in range(G)]
    for mode in ("append", "load"):
        pop, agents, _ = make_pair(ora, G, S, A, (12,), (12,), 128, seed=9, number_fitted_iteration=2,
                                   stoch_iter_actor=2, stoch_iter_critic=2)
        if mode == "load":
            for g in range(G):
                pop.traj_load(g, rows[g], [30, 70])
        else:
            for t in range(70):
                r = np.stack([rows[g][t] for g in range(G)])
                pop.traj_append(r[:, :S], r[:, S:S + A], r[:, S + A:S + 2 * A], r[:, S + 2 * A:2 * S + 2 * A],
                                r[:, 2 * S + 2 * A], r[:, 2 * S + 2 * A + 1] != 0)
                if t in (29, 69):
                    pop.episode_end()
        pop.update()
        pops.append([pop.weights(n) for n in (0, 1)])
    for n in (0, 1):
        assert np.array_equal(pops[0][n], pops[1][n])        # same kernels, same inputs: bit-identical


def test_default_iteration_counts_free_running(ora):
    """10 fitted iterations x 10 critic steps + 10 actor steps per update, two updates, EWC off."""
    S, A, ha, hv = 45, 17, (64, 64), (64, 64)
    G, maxT = 3, 512
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=8, lambda_=0.9, gae=1)
    rng = np.random.RandomState(21)
    for upd in range(2):
        lens = [500, 333, 64]
        trajs = load_trajs(pop, rng, lens, S, A)
        pop.update()
        for g, ag in enumerate(agents):
            oracle_update(ag, *trajs[g], S, A)
    for net, steps in ((0, 20), (1, 200)):
        w = pop.weights(net)
        it = pop.adam(net)[2]
        for g, ag in enumerate(agents):
            assert int(it[g]) == ag.nets[net].iter() == steps
            assert close_frac(w[g], ag.nets[net].weights(), 1e-3, 0.99, 1e-1), (net, g)


def test_ewc_lifecycle(ora):
    """Fisher accumulation over full-batch steps (updateFisher(T) / updateFisher(n)), ewc_setup, live
    penalty and ewc_decay on both nets (DODevMLP.hpp:604-758)."""
    S, A, ha, hv = 18, 6, (32,), (32,)
    G, maxT = 3, 200
    kw = dict(ewc=50.0, ewc_decay=0.9, ewc_fisher_beta=0.99, ewc_for_critic=1, ewc_best_param_method=1,
              lambda_=0.9, gae=1, update_critic_first=0, number_fitted_iteration=1, stoch_iter_actor=1, stoch_iter_critic=1)
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=13, **kw)
    rng = np.random.RandomState(5)

    def one_update(check=True):
        lens = list(rng.randint(20, maxT + 1, G))
        trajs = load_trajs(pop, rng, lens, S, A)
        ws_prev, ad_prev = sync_oracle(pop, agents)
        it_prev = [ad_prev[0][2].copy(), ad_prev[1][2].copy()]
        pop.update()
        ns = []
        for g, ag in enumerate(agents):
            n, _ = oracle_update(ag, *trajs[g], S, A)
            ns.append(n)
        if check:
            check_nets(pop, agents, ws_prev, it_prev, lrs, 0, [[n > 0 for n in ns], [True] * G])

    for _ in range(3):
        one_update()
    for net in (0, 1):
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            so = ag.nets[net].ewc_state()
            assert rel_err(st["fisher"][g], so["fisher"]) <= 1e-4, ("fisher", net, g, rel_err(st["fisher"][g], so["fisher"]))
            assert abs(st["fisher_nbr"][g] / so["fisher_nbr"] - 1) <= 1e-6
    # task switch
    pop.update_best_param(np.ones(G))
    pop.ewc_setup()
    from oracle import oracle as o
    for net in (0, 1):
        st = pop.ewc_state(net)
        for g, ag in enumerate(agents):
            ag.nets[net].update_best_param(1.0)
            ag.nets[net].ewc_setup()
            so = ag.nets[net].ewc_state()
            assert rel_err(st["fisher_hat"][g], so["fisher_hat"]) <= 1e-4, (net, g)
            # teacher forcing covers the EWC anchor too: a weight that took the other +-lr step
            # before the switch would otherwise carry a different (w - w*) into every later penalty
            fh = np.ascontiguousarray(st["fisher_hat"][g]); ws = np.ascontiguousarray(st["wstar"][g])
            o.lib().ora_mlp_set_ewc(ag.nets[net].h, fh.ctypes.data, ws.ctypes.data)
    for _ in range(3):
        one_update()
    for net in (0, 1):
        cost = pop.ewc_cost(net)
        for g, ag in enumerate(agents):
            co = ag.nets[net].ewc_cost()
            assert abs(cost[g] - co) <= 1e-4 * max(abs(co), 1e-12), (net, g, cost[g], co)
            assert co > 0


def test_vnn_from_scratch(ora):
    """agent.vnn_from_scratch (OfflineCaclaAg.hpp:129, 281-284): every fitted iteration of update_critic replaces the
    critic by a new net -- weights ~N(0, 0.01^2), zero Adam moments, iter 0 -- AFTER the targets were taken from the
    old one.  The fresh weights come from the handle's own stream (Caffe's cannot be matched): they are harvested
    from a twin handle that takes no step after the re-creation, then handed to the oracle (set_fresh)."""
    S, A, ha, hv = 18, 6, (32, 32), (32, 32)
    G, maxT, sic = 4, 200, 3
    kw = dict(update_critic_first=1, lambda_=0.9, gae=1, number_fitted_iteration=1, stoch_iter_actor=1, vnn_from_scratch=1)
    twin, _, _ = make_pair(ora, G, S, A, ha, hv, maxT, seed=8, stoch_iter_critic=0, **kw)
    pop, agents, lrs = make_pair(ora, G, S, A, ha, hv, maxT, seed=8, stoch_iter_critic=sic, **kw)
    lens = [200, 77, 0, 130]
    mask = np.array([1, 1, 1, 0], np.uint8)
    load_trajs(twin, np.random.RandomState(4), lens, S, A)
    trajs = load_trajs(pop, np.random.RandomState(4), lens, S, A)
    wv_before = pop.weights(1).copy()
    twin.update(mask=mask)
    fresh = twin.weights(1)
    m, v, it = twin.adam(1)
    for g in (0, 1):                                   # re-created: new draw, new solver
        assert abs(np.std(fresh[g]) - 0.01) < 0.002 and abs(np.mean(fresh[g])) < 0.002
        assert not m[g].any() and not v[g].any() and it[g] == 0
    for g in (2, 3):                                   # no trajectory / masked: the critic stays
        assert np.array_equal(fresh[g], wv_before[g])
    sync_oracle(pop, agents)
    pop.update(mask=mask)
    last = pop.last()
    w, (m, v, it) = pop.weights(1), pop.adam(1)
    for g in (0, 1):
        agents[g].set_fresh(fresh[g])
        rows, ends = trajs[g]
        n, deltas = oracle_update(agents[g], rows, ends, S, A)
        assert int(it[g]) == sic == agents[g].vnn.iter()
        wo = agents[g].vnn.weights()
        assert close_frac(w[g], wo, 1e-4, 0.99, 1e-2), g            # sic free-running Adam steps from the same fresh net
        assert rel_err(last["deltas"][g, :lens[g]], deltas) <= 1e-3, g
        assert int(last["n_pos"][g]) == n
    for g in (2, 3):
        assert np.array_equal(w[g], wv_before[g]) and it[g] == 0
    # three fitted iterations: three re-creations, the last one's solver counts the steps
    pop3, _, _ = make_pair(ora, G, S, A, ha, hv, maxT, seed=8, stoch_iter_critic=2, **dict(kw, number_fitted_iteration=3))
    load_trajs(pop3, np.random.RandomState(4), lens, S, A)
    pop3.update()
    assert list(pop3.adam(1)[2]) == [2, 2, 0, 2] and np.isfinite(pop3.weights(1)).all()


def test_from_scratch_refused_with_critic_ewc():
    import ddrl_amd
    from ddrl_amd.lib import DdrlError
    cfg = ddrl_amd.NfacConfig(n_agents=1, nb_sensors=4, nb_motors=1, vnn_from_scratch=1, ewc=10.0, ewc_for_critic=1)
    with pytest.raises(DdrlError, match="vnn_from_scratch"):
        ddrl_amd.NfacPopulation(cfg, init_seed=1)


@pytest.mark.parametrize("shape", ["config3", "other"])
def test_step_inside_the_gradient_launch_is_bit_identical(ora, shape):
    """option nfac_fuse = 1: the last chunk workgroup of an agent reduces the partials and steps inside the gradient
    launch; 0 (default: faster): a separate nfac_apply_kernel launch per step.  Same reduction order and arithmetic: identical bits, for the
    compile-time-shape kernels (config 3's nets) and the run-time-shape ones, with EWC live and ragged trajectories."""
    S, A = (45, 17) if shape == "config3" else (18, 6)
    ha = hv = (64, 64) if shape == "config3" else (32, 16)
    G, maxT = 5, 512
    res = []
    for fuse in (1, 0):
        pop, agents, _ = make_pair(ora, G, S, A, ha, hv, maxT, seed=12, lambda_=0.9, gae=1, ewc=10.0, ewc_for_critic=1,
                                   number_fitted_iteration=3, stoch_iter_critic=4, stoch_iter_actor=3)
        pop.set_option("nfac_fuse", fuse)
        rng = np.random.RandomState(4)
        for upd in range(3):
            load_trajs(pop, rng, [500, 64, 65, 1, 333], S, A)
            pop.update(mask=np.array([1, 1, 1, 1, 0], np.uint8) if upd == 1 else None)
            if upd == 0:
                pop.update_best_param(np.ones(G)); pop.ewc_setup()
        last = pop.last()
        res.append([pop.weights(0), pop.weights(1)] + list(pop.adam(0)) + list(pop.adam(1)) + [last["deltas"], last["critic_loss"], last["n_pos"]]
                   + [pop.ewc_state(n)["fisher"] for n in (0, 1)])
        pop.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
