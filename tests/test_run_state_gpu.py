"""save_run / load_previous_run (DeepQNAg.hpp:504-537, CMAESAg.hpp:279-308, cmaes.c:426-545)
through the C ABI: a run that is saved, loaded into a FRESH handle of the same shapes and continued must be
bit-identical to the uninterrupted run -- for every algorithm and for the agent front ends, including the
random streams (replay sampling, smoothing noise, exploration noise, c-cmaes' generator).  Files of another
shape, truncated files and foreign files are refused."""
import os

import numpy as np
import pytest

from ddpg_common import synth_transitions

pytestmark = pytest.mark.gpu


def _ddpg(td3, seed, G=5):
    import ddrl_amd
    kw = dict(n_agents=G, mini_batch_size=32, replay_memory=400, hidden_unit_a=(24, 16), hidden_unit_q=(20, 12),
              ewc=50.0, ewc_decay=0.9, ewc_for_critic=1)
    if td3:
        cfg = ddrl_amd.DdpgConfig(inverting_grad=0, td3=1, policy_noise=0.2, noise_clip=0.5, **kw)
        return ddrl_amd.Td3Population(cfg, init_seed=seed)
    return ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(nstep_return=2, **kw), init_seed=seed)


def _state(pop, nets):
    out = [pop.weights(n) for n in nets]
    for n in [x for x in nets if x in (0, 1, 4)]:
        out += list(pop.adam(n))
        st = pop.ewc_state(n)
        out += [st[k] for k in ("fisher", "fisher_hat", "wstar", "fisher_nbr", "decay_mult")]
    out.append(pop.replay_len())
    out += list(pop.gates())
    return out


@pytest.mark.parametrize("td3", [False, True])
def test_population_resumes_bit_identically(tmp_path, td3):
    G = 5
    nets = range(6) if td3 else range(4)
    a = _ddpg(td3, seed=1)
    rng = np.random.RandomState(0)
    a.set_hyper("alpha_a", 10 ** rng.uniform(-4, -2, G)); a.set_hyper("tau_soft_update", 10 ** rng.uniform(-3, -1, G))
    a.set_gates(rng.uniform(0, 1, (G, 18)) > 0.2, None)
    for g in range(G):
        a.replay_load(g, synth_transitions(np.random.RandomState(g), 300 + 10 * g, 18, 6))
    step = (lambda p, i: p.update(actor_step=i % 2 == 0)) if td3 else (lambda p, i: p.update_n(2))
    for i in range(3):
        step(a, i)
    a.update_best_param(np.arange(G, dtype=float))
    step(a, 3)
    a.ewc_setup()
    a.ewc_decay_update()
    step(a, 4)
    path = str(tmp_path / "continue.ddrl_pop")
    a.save_run(path)
    b = _ddpg(td3, seed=99)                                 # another initialisation, default hyper-parameters, no gates
    assert not np.array_equal(b.weights(0), a.weights(0))
    b.load_previous_run(path)
    for x, y in zip(_state(a, nets), _state(b, nets)):
        assert np.array_equal(x, y)
    s = np.random.RandomState(5).uniform(-1, 1, (G, 18)).astype(np.float32)
    tr = synth_transitions(np.random.RandomState(77), G, 18, 6)
    for p in (a, b):                                        # the continuation: new samples, updates, a task switch
        p.replay_append(tr[:, :18], tr[:, 18:24], tr[:, 42], tr[:, 24:42], tr[:, 43])
        for i in range(5, 8):
            step(p, i)
        p.update_best_param(np.arange(G, dtype=float) + 10)
        p.ewc_setup()
        step(p, 8)
    for x, y in zip(_state(a, nets), _state(b, nets)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.infer(s), b.infer(s))
    la, lb = a.last(), b.last()
    assert np.array_equal(la["idx"], lb["idx"]) and np.array_equal(la["q_targets"], lb["q_targets"])
    if td3:
        assert np.array_equal(la["noise"], lb["noise"]) and np.abs(la["noise"]).max() > 0
    a.close(); b.close()


def test_run_files_of_another_shape_are_refused(tmp_path):
    import ddrl_amd
    a = _ddpg(False, seed=1)
    path = str(tmp_path / "run")
    a.save_run(path)
    w = a.weights(1)
    other = ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(n_agents=5, mini_batch_size=32, replay_memory=400, hidden_unit_a=(24, 16),
                                                        hidden_unit_q=(20, 13), ewc=50.0, ewc_for_critic=1), init_seed=1)
    w_other = other.weights(0)
    with pytest.raises(ddrl_amd.DdrlError, match="different configuration"):
        other.load_previous_run(path)
    assert np.array_equal(other.weights(0), w_other)       # refused before anything was overwritten
    with pytest.raises(ddrl_amd.DdrlError):
        _ddpg(True, seed=1).load_previous_run(path)         # a TD3 population is not a DDPG one
    blob = open(path, "rb").read()
    open(path + ".cut", "wb").write(blob[: len(blob) // 2])
    with pytest.raises(ddrl_amd.DdrlError):
        a.load_previous_run(path + ".cut")
    open(path + ".txt", "w").write("not a run file at all")
    with pytest.raises(ddrl_amd.DdrlError, match="not a ddrl_pop run file"):
        a.load_previous_run(path + ".txt")
    with pytest.raises(ddrl_amd.DdrlError):
        a.load_previous_run(str(tmp_path / "missing"))
    a.load_previous_run(path)
    assert np.array_equal(a.weights(1), w)


def test_large_population_file_moves_in_pieces(tmp_path):
    """replay rings beyond the 64-MiB staging buffer of the record stream"""
    import ddrl_amd
    G = 256
    mk = lambda seed: ddrl_amd.DdpgPopulation(ddrl_amd.DdpgConfig(n_agents=G, replay_memory=3000, hidden_unit_a=(20,), hidden_unit_q=(50, 7)),
                                              init_seed=seed)
    a = mk(1)
    rows = [synth_transitions(np.random.RandomState(k), 3000, 18, 6) for k in range(4)]
    for g in range(G):
        a.replay_load(g, rows[g % 4][: 2500 + g])
    a.update_n(3)
    path = str(tmp_path / "big.run")
    a.save_run(path)
    assert os.path.getsize(path) > 128 << 20
    b = mk(2)
    b.load_previous_run(path)
    for p in (a, b):
        p.update_n(4)
    for net in range(4):
        assert np.array_equal(a.weights(net), b.weights(net)), net
    assert np.array_equal(a.last()["idx"], b.last()["idx"]) and np.array_equal(a.replay_len(), b.replay_len())
    a.close(); b.close()
    os.remove(path)


def test_nfac_population_resumes_mid_batch(tmp_path):
    """trajectories gathered but not yet learned from travel with the file"""
    import ddrl_amd
    from test_nfac_gpu import synth_traj
    G, S, A, maxT = 3, 7, 3, 128
    mk = lambda seed: ddrl_amd.NfacPopulation(ddrl_amd.NfacConfig(n_agents=G, nb_sensors=S, nb_motors=A, hidden_unit_a=(16,), hidden_unit_v=(12, 8),
                                                                 max_trajectory=maxT, ewc=20.0, ewc_for_critic=1), init_seed=seed)
    a = mk(1)
    rng = np.random.RandomState(3)
    for g in range(G):
        rows, ends = synth_traj(rng, 60 + 5 * g, S, A, 2)
        a.traj_load(g, rows, ends)
    a.update()
    for g in range(G):
        rows, ends = synth_traj(rng, 40 + 3 * g, S, A, 1)
        a.traj_load(g, rows, ends)
    path = str(tmp_path / "nfac.run")
    a.save_run(path)
    b = mk(2)
    b.load_previous_run(path)
    assert list(a.traj_len()) == list(b.traj_len()) and a.traj_len()[0] == 40
    for p in (a, b):
        p.update()
    for net in (0, 1):
        assert np.array_equal(a.weights(net), b.weights(net))
        for x, y in zip(a.adam(net), b.adam(net)):
            assert np.array_equal(x, y)
    la, lb = a.last(), b.last()
    for g in range(G):                                       # (rows beyond a trajectory keep the previous update's values)
        assert np.array_equal(la["deltas"][g, :40 + 3 * g], lb["deltas"][g, :40 + 3 * g])
    for k in la:
        if k != "deltas":
            assert np.array_equal(la[k], lb[k]), k
    assert list(b.traj_len()) == [0] * G


def test_cmaes_resumes_bit_identically(tmp_path):
    import ddrl_amd
    N, lam = 23, 16
    rng = np.random.RandomState(1)
    x0, sd = rng.normal(0, 1, N), np.full(N, 0.4)
    fit = lambda X: np.sum((X - 0.3) ** 2, axis=1) + 0.1 * np.sum(np.abs(X), axis=1)
    a = ddrl_amd.Cmaes(x0, sd, 7, lam)
    for _ in range(6):
        a.update(fit(a.sample()))
    Xa = a.sample()                                          # saved between sampling and the update
    path = str(tmp_path / "continue.cmaes.data")
    a.save_run(path)
    b = ddrl_amd.Cmaes(rng.normal(0, 1, N), sd * 2, 12345, lam)
    b.load_previous_run(path)
    assert np.array_equal(b.population(), Xa)
    for es in (a, b):
        es.update(fit(es.population()))
        for _ in range(5):
            es.update(fit(es.sample()))
    assert np.array_equal(a.sample(), b.sample())
    for key in ("xmean", "xbestever", "ps", "pc", "D", "C", "B"):
        assert np.array_equal(a.vec(key), b.vec(key)), key
    for key in ("sigma", "fbestever", "gen", "countevals", "eigen_uptodate"):
        assert a.get(key) == b.get(key), key
    assert a.test_for_termination() == b.test_for_termination()
    with pytest.raises(ddrl_amd.DdrlError, match="different configuration"):
        ddrl_amd.Cmaes(x0, sd, 7, lam + 4).load_previous_run(path)


INI = """[agent]
gamma=0.97
decision_each=1
noise=0.3
gaussian_policy={gp}
noise2=0.15
noise3=0.01
hidden_unit_q=24:12
hidden_unit_a=16
mini_batch_size=32
replay_memory=400
inverting_grad=true
force_more_update=1
tau_soft_update=0.01
alpha_a=0.001
alpha_v=0.002
decay_v=-1
batch_norm_critic=0
batch_norm_actor=0
actor_output_layer_type=2
hidden_layer_type=1
test_net=false
bn_adapt=false
momentum=0
nstep_return=0
policy_noise=0.2
noise_clip=0.5
policy_freq=2
"""


@pytest.mark.parametrize("kind,gp", [("ddpg", 1), ("ddpg", 2), ("td3", 1)])
def test_agent_front_end_resumes_mid_episode(tmp_path, kind, gp):
    """counters, returns, the pending transition and the exploration-noise engines: identical actions after the resume"""
    import ddrl_amd
    ini = os.path.join(str(tmp_path), "a.ini")
    open(ini, "w").write(INI.format(gp=gp))
    G, S, A = 4, 5, 2
    cls = ddrl_amd.DdpgAgents if kind == "ddpg" else ddrl_amd.Td3Agents
    rng = np.random.RandomState(2)
    T = 340
    R = rng.normal(0, 1, (T, G)).astype(np.float32); X = rng.uniform(-1, 1, (T, G, S)).astype(np.float32)
    goal = rng.uniform(0, 1, (T, G)) < 0.01
    a = cls(ini, G, S, A, seed=4)
    a.start_episode(X[0], True)
    for t in range(1, 300):
        a.run(R[t], X[t], True, goal[t], False)
    a.end_episode(True)
    a.start_episode(X[300], True)
    for t in range(301, 320):
        a.run(R[t], X[t], True, goal[t], False)
    path = str(tmp_path / "agents.run")
    a.save_run(path)
    b = cls(ini, G, S, A, seed=1234)
    b.load_previous_run(path)
    for t in range(320, T):
        ya, yb = a.run(R[t], X[t], True, goal[t], t == T - 1), b.run(R[t], X[t], True, goal[t], t == T - 1)
        assert np.array_equal(ya, yb), t
    for p in (a, b):
        p.end_episode(True)
    sa, sb = a.stats(), b.stats()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert sa["episode"][0] == 2 and a.pop.adam(1)[2][0] > 50
    for net in range(4 if kind == "ddpg" else 6):
        assert np.array_equal(a.pop.weights(net), b.pop.weights(net)), net
    a.close(); b.close()


def test_cmaes_front_end_resumes_mid_generation(tmp_path):
    import ddrl_amd
    from test_cmaes_agents_gpu import write_ini
    ini = write_ini(tmp_path, lam=8)
    S, A = 4, 2
    a = ddrl_amd.CmaesAgents(ini, S, A, seed=3)
    G = a.G
    rng = np.random.RandomState(0)
    X = rng.uniform(-1, 1, (40, G, S)).astype(np.float32)

    def episode(ag, t0, mask=None):
        ag.start_episode(X[t0], True, mask)
        for t in range(t0 + 1, t0 + 5):
            act = ag.run(np.sum(X[t], axis=1) * 0.1, X[t], True, False, t == t0 + 4, mask)
        ag.end_episode(True, mask)
        return act

    episode(a, 0)                                            # one full generation
    half = np.arange(G) < G // 2
    episode(a, 5, half.astype(np.uint8))                     # half of the next one
    path = str(tmp_path / "cmaes_agents.run")
    a.save_run(path)
    b = ddrl_amd.CmaesAgents(ini, S, A, seed=77)
    b.load_previous_run(path)
    assert a.state() == b.state() and a.state()["n_evaluated"] == G // 2 and a.state()["generations"] == 1
    for ag in (a, b):
        episode(ag, 10, (~half).astype(np.uint8))            # completes the generation: update + new population
        last = episode(ag, 15)
        ag._last = last
    assert a.state() == b.state() and a.state()["generations"] == 3
    assert np.array_equal(a._last, b._last)
    assert np.array_equal(a.pop.weights(0), b.pop.weights(0))
    for key in ("xmean", "C"):
        assert np.array_equal(a.es.vec(key), b.es.vec(key))
    assert a.es.get("sigma") == b.es.get("sigma")
    a.close(); b.close()
