"""The NFAC(lambda)-V agent front end (ddrl_agents_create_nfac, OfflineCaclaAg_*) on the GPU against a
Python restatement of ARLAgent::runf (ARLAgent.hpp:37-137) + OfflineCaclaAg::_run / end_episode /
end_instance (OfflineCaclaAg.hpp:79-119, 403-600) driving the fp64 oracle (oracle/nfac_ref.c).
Bookkeeping (decision_each, reward max-pooling, reward_scale, trajectory length, update every
update_each_episode episodes, Solver::iter, episode counters) must be exact; actions and weights are a
compared per update (1e-3 on >= 99 % of the weights, then the oracle restarts from the device state).  No reference test pins this layer: parity
unpinned."""
import os

import numpy as np
import pytest

from ddpg_common import close_frac

pytestmark = pytest.mark.gpu

INI = """[agent]
gamma=0.98
decision_each={de}
reward_scale={rs}
noise={noise}
gaussian_policy={gp}
noise2={n2}
noise3={n3}
hidden_unit_v=24:12
hidden_unit_a=16
update_delta_neg=false
vnn_from_scratch=false
update_critic_first=true
number_fitted_iteration=2
stoch_iter_actor=3
stoch_iter_critic=2
batch_norm_actor=0
batch_norm_critic=0
actor_output_layer_type=2
hidden_layer_type=1
alpha_a=0.001
alpha_v=0.002
lambda=0.8
gae=true
momentum=0
update_each_episode={uee}
"""


def write_ini(tmp_path, de=1, rs=1.0, noise=0.0, gp=1, n2=0.5, n3=4.0, uee=1, name="n.ini"):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(INI.format(de=de, rs=rs, noise=noise, gp=gp, n2=n2, n3=n3, uee=uee))
    return p


def test_population_run_matches_reference_loop(ora, tmp_path):
    import ddrl_amd
    G, S, A, seed = 3, 5, 2, 4
    de, rs, uee = 2, 0.5, 2
    ag = ddrl_amd.NfacAgents(write_ini(tmp_path, de=de, rs=rs, uee=uee, noise=0.25, gp=1), G, S, A, max_trajectory=256, seed=seed)
    assert ag.pop.maxT == 256
    rng = np.random.RandomState(0)
    na, nv = ag.pop.num_params(0), ag.pop.num_params(1)
    wa = rng.normal(0, 0.3, (G, na)).astype(np.float32).astype(np.float64)
    wv = rng.normal(0, 0.3, (G, nv)).astype(np.float32).astype(np.float64)
    ag.pop.set_weights(0, wa); ag.pop.set_weights(1, wv)
    refs = []
    for g in range(G):
        o = ora.Nfac(S, A, (16,), (24, 12), gamma=0.98, alpha_a=float(np.float32(0.001)), alpha_v=float(np.float32(0.002)),
                     hidden_act=1, actor_out_act=2, momentum=0, update_delta_neg=0, update_critic_first=1, gae=1,
                     inverting_gradient=0, number_fitted_iteration=2, stoch_iter_actor=3, stoch_iter_critic=2, lambda_=0.8)
        o.ann.set_weights(wa[g]); o.vnn.set_weights(wv[g])
        refs.append(dict(o=o, rows=[], ends=[], ep=0))
    worst, moved, decisions = 0.0, 0, 0
    learn = np.array([1, 1, 1], np.uint8)
    for episode in range(5):
        if episode == 3:
            learn = np.array([1, 0, 1], np.uint8)          # agent 1 is tested, not trained, for one episode
        if episode == 4:
            learn = np.array([1, 1, 1], np.uint8)
        steps = [37, 22, 40, 9, 31][episode]
        s0 = rng.uniform(-1, 1, (G, S)).astype(np.float32)
        ag.start_episode(s0, learn)
        for g, st in enumerate(refs):
            st.update(inter=[], tfa=1, gpow=1.0, swr=0.0, sr=0.0, last_a=None, last_pa=None, last_s=s0[g].astype(np.float64),
                      ret=np.zeros(A))
        for t in range(steps):
            r = rng.normal(0, 1, G).astype(np.float32)
            s = rng.uniform(-1, 1, (G, S)).astype(np.float32)
            goal = rng.uniform(0, 1, G) < 0.03
            last = np.full(G, t == steps - 1)
            act = ag.run(r, s, learn, goal, last)
            pure = ag.pop.infer(s)
            for g, st in enumerate(refs):
                st["inter"].append(float(r[g]))
                st["tfa"] -= 1
                if st["tfa"] == 0 or goal[g] or last[g]:
                    rew = max(st["inter"])
                    if goal[g]:
                        rew = float(r[g])
                    nxt = st["o"].ann.forward(s[g].astype(np.float64))[0]
                    if st["last_a"] is not None and learn[g]:
                        st["rows"].append(np.concatenate([st["last_s"], st["last_pa"], st["last_a"], s[g].astype(np.float64),
                                                          [float(np.float32(rew * rs))], [float(goal[g])]]))
                    # the explored action is the device's (its own noise stream, fp32 bits); the pure action is
                    # pi(s), compared below
                    st["last_pa"] = nxt
                    st["last_a"] = act[g].astype(np.float64)
                    worst = max(worst, float(np.max(np.abs(pure[g] - nxt))))
                    moved += int(np.any(act[g] != pure[g])); decisions += 1
                    if not learn[g]:
                        assert np.array_equal(act[g], pure[g])
                    st["last_s"] = s[g].astype(np.float64)
                    st["ret"] = nxt
                    st["tfa"] = de
                    st["inter"] = []
                    st["swr"] += rew * st["gpow"]; st["sr"] += rew; st["gpow"] *= 0.98
        lens_before = ag.pop.traj_len()
        assert list(lens_before) == [len(st["rows"]) for st in refs]
        ag.end_episode(learn)
        for g, st in enumerate(refs):
            if not learn[g]:
                continue
            st["ends"].append(len(st["rows"]))
            if st["ep"] % uee == 0:
                rows = np.asarray(st["rows"], np.float64).reshape(-1, 2 * S + 2 * A + 2)
                o = st["o"]
                o.update(rows[:, :S], rows[:, S:S + A], rows[:, S + A:S + 2 * A], rows[:, S + 2 * A:2 * S + 2 * A],
                         rows[:, 2 * S + 2 * A], rows[:, 2 * S + 2 * A + 1] != 0, np.asarray(st["ends"], np.int32))
                st["rows"], st["ends"] = [], []
            st["ep"] += 1
        # every update is checked on its own, then the oracle restarts from the device's fp32 state: the
        # positive-advantage selection (OfflineCaclaAg.hpp:457-470) makes a free run discontinuous
        for net in (0, 1):
            w, (m, v, it) = ag.pop.weights(net), ag.pop.adam(net)
            for g, st in enumerate(refs):
                assert int(it[g]) == st["o"].nets[net].iter(), (episode, net, g)
                assert close_frac(w[g], st["o"].nets[net].weights(), 1e-3, 0.99, 1e-1), (episode, net, g)
                st["o"].nets[net].set_weights(w[g]); st["o"].nets[net].set_adam(m[g], v[g], int(it[g]))
        stats = ag.stats()
        for g, st in enumerate(refs):
            assert stats["sum_weighted_reward"][g] == st["swr"], (episode, g)
            assert stats["sum_reward"][g] == st["sr"]
            assert stats["episode"][g] == st["ep"]
        assert list(ag.pop.traj_len()) == [len(st["rows"]) for st in refs]
    assert worst <= 1e-4, worst
    assert moved > 0.8 * decisions
    assert ag.pop.adam(0)[2][0] > 0
    ag.close()


@pytest.mark.parametrize("gp", [3, 4])
def test_nfac_only_exploration_policies(tmp_path, gp):
    """gaussian_policy 3: truncated Gaussian with probability noise2; 4: sigma = noise * noise2^(noise3 - step)
    (OfflineCaclaAg.hpp:95-102)."""
    import ddrl_amd
    from scipy.stats import truncnorm
    G, S, A, noise, n2, n3 = 256, 3, 4, 0.4, (0.3 if gp == 3 else 0.9), 5.0
    ag = ddrl_amd.NfacAgents(write_ini(tmp_path, noise=noise, gp=gp, n2=n2, n3=n3), G, S, A, max_trajectory=64, seed=9)
    ag.pop.set_weights(0, np.zeros((G, ag.pop.num_params(0))))      # pi(s) = tanh(0) = 0
    ag.start_episode(np.zeros((G, S)), True)
    acts = np.stack([ag.run(0.0, np.zeros((G, S)), True) for _ in range(12)])      # [T, G, A]
    assert np.all(np.abs(acts) <= 1.0)
    if gp == 3:
        moved = np.any(acts != 0.0, axis=2)
        assert abs(moved.mean() - n2) < 0.03
        ref = truncnorm(-1 / noise, 1 / noise, loc=0, scale=noise)
        assert abs(acts[moved].var() - ref.var()) < 0.01
    else:
        for t in (0, 5, 11):
            sig = noise * n2 ** (n3 - t)
            ref = truncnorm(-1 / sig, 1 / sig, loc=0, scale=sig)
            assert abs(acts[t].std() - ref.std()) < 0.05 * ref.std() + 0.005, t
    # the pure action is what the trajectory stores beside the noisy one: nothing to learn from => n_pos >= 0 only
    ag.end_episode(True)
    assert np.all(ag.pop.traj_len() == 0)           # consumed by the update
    ag.close()


def test_reference_c_flavour(tmp_path):
    """OfflineCaclaAg_* exactly as gym/agent.py:17-62 drives libddrl-nfac.so."""
    import ddrl_amd
    S, A = 4, 2
    ini = write_ini(tmp_path, noise=0.2)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        a = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "3"], algo="nfac")
        assert a.name() == "NFAC(lambda)-V"
        rng = np.random.RandomState(1)
        probe = rng.uniform(-1, 1, S)
        a.start_ep(probe, False)
        first = a.run(0.0, probe, False, False, False)
        assert first.shape == (A,) and np.all(np.abs(first) <= 1)
        for ep in range(3):
            s = rng.uniform(-1, 1, S)
            a.start_ep(s, True)
            for t in range(60):
                a.run(float(rng.normal()), rng.uniform(-1, 1, S), True, False, t == 59)
            a.end_ep(True)
        a.dump(True, 2, 60, 0.5)
        assert open("0.learning.data").read().startswith("2 60 0.5")
        a.start_ep(probe, False)
        after = a.run(0.0, probe, False, False, False)
        assert not np.array_equal(after, first)          # three updates moved the actor
        a.save(2)
        b = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "50"], algo="nfac")
        b.load(2)
        b.start_ep(probe, False)
        assert np.array_equal(b.run(0.0, probe, False, False, False), after)
        a.close(); b.close()
    finally:
        os.chdir(cwd)
