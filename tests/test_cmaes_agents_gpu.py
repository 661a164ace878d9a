"""CMAESAg on the GPU (ddrl_agents_create_cmaes, ddrl_policy_create, ddrl_pop_load_flat_dev) against the
reference's loop (CMAESAg.hpp:38-47, 144-253) restated here over the oracle's c-cmaes and fp64 MLP.
Bit-exact: the weights each policy holds (= the fp32 population row), the fitness handed to
cmaes_UpdateDistribution (float accumulator), the ranking, fbestever / xbestever, the generation
bookkeeping.  Actions: fp32 vs fp64 forward, 1e-5.  The distribution state is compared teacher-forced
like tests/test_cmaes_gpu.py.  No reference test pins this layer: parity unpinned."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INI = """[agent]
gamma=0.95
decision_each={de}
hidden_unit_a=8:5
actor_hidden_layer_type=1
actor_output_layer_type=2
batch_norm=0
population={lam}
initial_deviation=0.3
check_feasible=true
"""


def write_ini(tmp_path, de=1, lam=12):
    p = os.path.join(str(tmp_path), "c.ini")
    with open(p, "w") as f:
        f.write(INI.format(de=de, lam=lam))
    return p


def nerr(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_policies_from_device_rows():
    import ddrl_amd
    G, S, A, hid = 6, 7, 3, (10, 6)
    pol = ddrl_amd.PolicyPopulation(G, S, A, hid, init_seed=2)
    n = pol.num_params(0)
    assert n == (S + 1) * 10 + 11 * 6 + 7 * 3
    w0 = pol.weights(0)
    es = ddrl_amd.Cmaes(np.linspace(-1, 1, n), np.full(n, 0.5), 3, G)
    X = es.sample()
    ptr, ld = es.population_dev()
    mask = np.array([1, 0, 1, 1, 0, 1], np.uint8)
    pol.load_flat_dev(ptr, ld, mask)
    w = pol.weights(0)
    for g in range(G):
        assert np.array_equal(w[g], X[g] if mask[g] else w0[g]), g        # X is the fp32 population, exactly
    pol.load_flat_dev(ptr + 4 * ld * 2, 0)                                 # broadcast row 2
    assert np.array_equal(pol.weights(0), np.tile(X[2], (G, 1)))
    # forward of the loaded policies = the ordinary inference path
    s = np.random.RandomState(0).uniform(-1, 1, (G, S)).astype(np.float32)
    a = pol.infer(s)
    assert a.shape == (G, A) and np.all(np.abs(a) <= 1) and np.ptp(a) > 0
    with pytest.raises(ddrl_amd.DdrlError):
        pol.set_hyper("alpha_a", np.ones(G))
    with pytest.raises(ddrl_amd.DdrlError):
        pol.weights(1)
    with pytest.raises(ddrl_amd.DdrlError):
        pol.load_flat_dev(ptr, n - 1)
    es.close(); pol.close()


def test_generation_loop_matches_reference_loop(ora, tmp_path):
    import ddrl_amd
    S, A, de, lam, seed = 4, 2, 2, 12, 7
    ag = ddrl_amd.CmaesAgents(write_ini(tmp_path, de=de, lam=lam), S, A, seed=seed)
    assert ag.G == lam == ag.es.lam
    N = ag.pop.num_params(0)
    # the distribution starts at the net's initial weights (CMAESAg.hpp:89-98); new_population() has sampled
    # already but no individual is loaded before start_instance
    startx = ag.pop.weights(0)[0]
    assert nerr(ag.es.vec("xmean"), startx) == 0.0
    ref = ora.Cmaes(startx, [0.3] * N, seed % 2147483646 + 1, lam)
    nets = [ora.Mlp([S, 8, 5, A], hidden_act=1, out_act=2) for _ in range(lam)]
    rng = np.random.RandomState(3)
    best_ever = np.inf
    for gi in range(4):
        Xd, Xr = ag.es.population(), ref.sample()
        xm = ref.vec("xmean")
        assert nerr(Xd - xm, Xr - xm) <= 1e-5, gi
        ref.view("X", lam * N)[:] = Xd.ravel()
        sequential = gi == 2                    # one generation in the reference's order: one individual per episode
        groups = [[g] for g in range(lam)] if sequential else [list(range(lam))]
        swr = np.zeros(lam)
        for grp in groups:
            mask = np.zeros(lam, np.uint8); mask[grp] = 1
            s0 = rng.uniform(-1, 1, (lam, S)).astype(np.float32)
            ag.start_episode(s0, True, mask)
            W = ag.pop.weights(0)
            for g in grp:
                assert np.array_equal(W[g], Xd[g]), (gi, g)
                nets[g].set_weights(W[g])
            T = 9
            inter = {g: [] for g in grp}; tfa = {g: 1 for g in grp}; gp = {g: 1.0 for g in grp}
            ret = {g: None for g in grp}
            for t in range(T):
                r = rng.normal(0, 1, lam).astype(np.float32)
                s = rng.uniform(-1, 1, (lam, S)).astype(np.float32)
                last = np.full(lam, t == T - 1)
                act = ag.run(r, s, True, False, last, mask)
                for g in grp:
                    inter[g].append(float(r[g])); tfa[g] -= 1
                    if tfa[g] == 0 or last[g]:
                        rew = max(inter[g])
                        ret[g] = nets[g].forward(s[g].astype(np.float64))[0]
                        swr[g] += rew * gp[g]; gp[g] *= 0.95
                        tfa[g] = de; inter[g] = []
                    assert np.max(np.abs(act[g] - ret[g])) <= 1e-5, (gi, g, t)      # no action noise (CMAESAg.hpp:43)
            st = ag.state()
            assert st["generations"] == gi and st["n_evaluated"] == (grp[0] if sequential else 0)
            ag.end_episode(True, mask)
            assert np.array_equal(ag.stats()["sum_weighted_reward"][grp], swr[grp])
        fit = (-swr).astype(np.float32).astype(np.float64)          # accumulate(scores, 0.f) / size (:239)
        ref.update(fit)
        st = ag.state()
        assert st["generations"] == gi + 1 and st["n_evaluated"] == 0 and st["error_count"] == 0
        assert list(ag.es.index()) == list(ref.index()), gi
        best_ever = min(best_ever, fit.min())
        assert ag.es.get("fbestever") == ref.get("fbestever") == best_ever
        assert np.array_equal(ag.es.vec("xbestever"), ref.vec("xbestever"))
        assert abs(ag.es.get("sigma") / ref.get("sigma") - 1.0) <= 1e-5
        for key in ("xmean", "ps", "pc"):
            assert nerr(ag.es.vec(key), ref.vec(key)) <= 1e-5, (gi, key)
        Cd = ag.es.vec("C")
        ref.view("C", N * N)[:] = Cd.ravel()
        for key in ("xmean", "ps", "pc"):
            ref.view(key)[:] = ag.es.vec(key)
        ref.set_sigma(ag.es.get("sigma"))
        assert list(ag.stats()["episode"]) == [gi + 1] * lam
    # testing after the first update runs the best-ever solution (CMAESAg.hpp:215-218), learning slots the new individuals
    learn = np.zeros(lam, np.uint8); learn[:3] = 1
    ag.start_episode(np.zeros((lam, S), np.float32), learn)
    W, X = ag.pop.weights(0), ag.es.population()
    best = ag.es.vec("xbestever").astype(np.float32)
    for g in range(lam):
        assert np.array_equal(W[g], X[g] if learn[g] else best), g
    ag.close()


def test_config_errors(tmp_path):
    import ddrl_amd
    p = write_ini(tmp_path)
    with open(p, "a") as f:
        f.write("[devnn]\newc=0.5\n")
    with pytest.raises(ddrl_amd.DdrlError, match="gradient"):
        ddrl_amd.CmaesAgents(p, 4, 2)
    with pytest.raises(ddrl_amd.DdrlError):
        ddrl_amd.CmaesAgents(os.path.join(str(tmp_path), "missing.ini"), 4, 2)
