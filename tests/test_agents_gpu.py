"""The agent-level front end (ddrl_agents_*, DDPGAg_*) on the GPU against a Python restatement of
ARLAgent::runf / DeepQNAg::_run (ARLAgent.hpp:37-137, DeepQNAg.hpp:122-165, 290-339) driving the
fp64 oracle nets.  Bookkeeping (decision_each, reward max-pooling, reward_scale, discounted return,
replay length, replay indices, episode counters) must be exact; actions and weights are a free-
running fp32-vs-fp64 comparison over a few dozen updates (1e-3 on >= 99 %).  No reference test
pins this layer: parity unpinned.  Noise: distribution checks (the reference's engine is clock-
seeded)."""
import os

import numpy as np
import pytest

from ddpg_common import close_frac, oracle_weights

pytestmark = pytest.mark.gpu

INI = """[agent]
gamma=0.97
decision_each={de}
reward_scale={rs}
noise={noise}
gaussian_policy={gp}
noise2=0.15
noise3=0.01
hidden_unit_q=24:12
hidden_unit_a=16
mini_batch_size=32
replay_memory=400
inverting_grad=true
force_more_update={fmu}
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
"""


def write_ini(tmp_path, de=1, rs=1.0, noise=0.0, gp=0, fmu=0, name="a.ini"):
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(INI.format(de=de, rs=rs, noise=noise, gp=gp, fmu=fmu))
    return p


def replay_seed(seed, g):
    return (seed * 40503 + 7919 * g + 17) & 0xFFFFFFFF


def test_population_run_matches_reference_loop(ora, tmp_path):
    import ddrl_amd
    G, S, A, B, seed = 3, 5, 2, 32, 11
    de, rs, fmu = 2, 0.5, 1
    ag = ddrl_amd.DdpgAgents(write_ini(tmp_path, de=de, rs=rs, fmu=fmu), G, S, A, seed=seed)
    rng = np.random.RandomState(0)
    na, nq = ag.pop.num_params(0), ag.pop.num_params(1)
    wa = rng.normal(0, 0.3, (G, na)).astype(np.float32).astype(np.float64)
    wq = rng.normal(0, 0.3, (G, nq)).astype(np.float32).astype(np.float64)
    for net, w in ((0, wa), (1, wq), (2, wa), (3, wq)):
        ag.pop.set_weights(net, w)
    refs = []
    for g in range(G):
        o = ora.Ddpg(S, A, B, (16,), (24, 12), replay_cap=400, gamma=0.97, tau=float(np.float32(0.01)),
                     alpha_a=float(np.float32(0.001)), alpha_v=float(np.float32(0.002)), seed=replay_seed(seed, g))
        o.ann.set_weights(wa[g]); o.ann_target.set_weights(wa[g]); o.qnn.set_weights(wq[g]); o.qnn_target.set_weights(wq[g])
        refs.append(dict(o=o, inter=[], tfa=1, gpow=1.0, swr=0.0, sr=0.0, last_s=None, last_a=None, ret=np.zeros(A), ep=0))

    def ref_update(st):
        o = st["o"]
        for _ in range(1 + fmu):
            if o.replay_len() >= 250 and o.replay_len() >= B:
                o.update()

    worst_act = 0.0
    for episode in range(2):
        steps = 520 if episode == 0 else 21
        s0 = rng.uniform(-1, 1, (G, S)).astype(np.float32)
        ag.start_episode(s0, True)
        for st in refs:
            st.update(inter=[], tfa=1, gpow=1.0, swr=0.0, sr=0.0, last_a=None)
        for g in range(G):
            refs[g]["last_s"] = s0[g].astype(np.float64)
        for t in range(steps):
            r = rng.normal(0, 1, G).astype(np.float32)
            s = rng.uniform(-1, 1, (G, S)).astype(np.float32)
            goal = rng.uniform(0, 1, G) < 0.02
            last = np.full(G, t == steps - 1)
            act = ag.run(r, s, True, goal, last)
            for g, st in enumerate(refs):
                st["inter"].append(float(r[g]))
                st["tfa"] -= 1
                if st["tfa"] == 0 or goal[g] or last[g]:
                    rew = max(st["inter"])
                    if goal[g]:
                        rew = float(r[g])
                    nxt = st["o"].ann.forward(s[g].astype(np.float64))[0]
                    if st["last_a"] is not None:
                        st["o"].insert(st["last_s"], st["last_a"], float(np.float32(rew * rs)), s[g].astype(np.float64), bool(goal[g]))
                        ref_update(st)
                    # the stored action is what the device returned (fp32), so both replays hold the same bits
                    st["last_a"] = act[g].astype(np.float64)
                    st["last_s"] = s[g].astype(np.float64)
                    st["ret"] = nxt
                    st["tfa"] = de
                    st["inter"] = []
                    st["swr"] += rew * st["gpow"]; st["sr"] += rew; st["gpow"] *= 0.97
                worst_act = max(worst_act, float(np.max(np.abs(act[g] - st["ret"]))))
        ag.end_episode(True)
        for st in refs:
            ref_update(st)
            st["ep"] += 1
        stats = ag.stats()
        for g, st in enumerate(refs):
            assert stats["sum_weighted_reward"][g] == st["swr"], (episode, g)
            assert stats["sum_reward"][g] == st["sr"]
            assert stats["episode"][g] == st["ep"]
    assert worst_act <= 2e-3, worst_act
    L = ddrl_amd.lib.load()
    ln = np.empty(G, np.int32)
    assert L.ddrl_pop_replay_len(ag.pop.h, ln.ctypes.data) == 0
    assert list(ln) == [st["o"].replay_len() for st in refs]
    it = ag.pop.adam(1)[2]
    assert list(it) == [st["o"].qnn.iter() for st in refs] and it[0] > 20
    for net in range(4):
        w = ag.pop.weights(net)
        wo = oracle_weights([st["o"] for st in refs], net)
        for g in range(G):
            assert close_frac(w[g], wo[g], 1e-3, 0.99, 1e-1), (net, g)
    ag.close()


@pytest.mark.parametrize("gp,noise", [(0, 0.3), (1, 0.5), (2, 0.2)])
def test_exploration_noise_distributions(tmp_path, gp, noise):
    import ddrl_amd
    G, S, A = 64, 3, 4
    ag = ddrl_amd.DdpgAgents(write_ini(tmp_path, noise=noise, gp=gp), G, S, A, seed=5)
    for net in (0, 2):
        ag.pop.set_weights(net, np.zeros((G, ag.pop.num_params(net))))      # pi(s) = tanh(0) = 0
    ag.start_episode(np.zeros((G, S)), True)
    acts = np.stack([ag.run(0.0, np.zeros((G, S)), True) for _ in range(60)])      # [T, G, A]
    assert np.all(np.abs(acts) <= 1.0)
    test = ag.run(0.0, np.zeros((G, S)), False)
    assert np.all(test == 0.0)                                                      # no noise when not learning
    if gp == 0:      # e-greedy: with prob. noise the whole action is uniform in [-1, 1]
        moved = np.any(acts != 0.0, axis=2)
        assert abs(moved.mean() - noise) < 0.03
        u = acts[moved]
        assert abs(u.mean()) < 0.03 and abs(u.var() - 1.0 / 3.0) < 0.03
    elif gp == 1:    # N(0, sigma) truncated to [-1, 1]
        from scipy.stats import truncnorm
        ref = truncnorm(-1 / noise, 1 / noise, loc=0, scale=noise)
        assert abs(acts.mean()) < 0.01 and abs(acts.var() - ref.var()) < 0.01
        assert abs(np.mean(np.abs(acts) > 0.8) - 2 * ref.sf(0.8)) < 0.01
    else:            # OU: x_{t+1} = x_t - theta dt x_t + sigma sqrt(dt) N(0,1), action = clip(pi + x)
        d = np.diff(acts, axis=0)                   # far from the clip: increments ~ N(-theta dt x, sigma^2 dt)
        assert abs(d.std() - noise * np.sqrt(0.01)) < 0.002
        assert abs(acts[0].std() - noise * np.sqrt(0.01)) < 0.004
    ag.close()


def test_reference_c_flavour(tmp_path):
    """DDPGAg_* exactly as gym/agent.py:113-157 drives libddrl-ddpg.so."""
    import ddrl_amd
    S, A = 4, 2
    ini = write_ini(tmp_path, noise=0.0)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        a = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "3"])
        assert a.name() == "DDPG"
        rng = np.random.RandomState(1)
        s = rng.uniform(-1, 1, S)
        a.start_ep(s, True)
        first = a.run(0.0, s, True, False, False)
        assert first.shape == (A,) and first.dtype == np.float64 and np.all(np.abs(first) <= 1)
        for t in range(300):
            out = a.run(float(rng.normal()), rng.uniform(-1, 1, S), True, False, t == 299)
        a.end_ep(True)
        a.dump(True, 0, 300, 1.5)
        a.display(True, 0, 300, 1.5)
        assert open("0.learning.data").read().startswith("0 300 1.5")
        probe = rng.uniform(-1, 1, S)
        a.start_ep(probe, False)
        before = a.run(0.0, probe, False, False, False)
        assert not np.array_equal(before, first)          # it learned something (weights moved)
        a.save(7)
        b = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "99"])
        b.start_ep(probe, False)
        assert not np.array_equal(b.run(0.0, probe, False, False, False), before)
        b.load(7)
        b.start_ep(probe, False)
        assert np.array_equal(b.run(0.0, probe, False, False, False), before)
        # a missing config is reported, not fatal (the reference exits)
        c = ddrl_amd.DDRLAg(A, S, ["--config", os.path.join(str(tmp_path), "nope.ini")])
        with pytest.raises(ddrl_amd.DdrlError):
            c.run(0.0, probe, False, False, False)
        for x in (a, b, c):
            x.close()
    finally:
        os.chdir(cwd)


def test_config1_cartpole_plumbing(ora, tmp_path):
    """cartpole dims (S=4, A=1), one agent, the nets / minibatch / force_more_update of the
    reference's agent/deepqn/data/ddpg-cp.ini (q=50,7 a=20, B=512, 8 passes of the fupd loop per stored transition),
    synthetic transitions s ~ U(-1,1)^4.  The first update round is held to the oracle (indices bit-exact); then the
    run goes on to 1500 steps: update counts exact, weights finite."""
    import ddrl_amd
    ini = os.path.join(str(tmp_path), "cp.ini")
    open(ini, "w").write("""[agent]
gamma=0.99
decision_each=1
noise=0.1
gaussian_policy=0
hidden_unit_q=50,7
hidden_unit_a=20
mini_batch_size=512
replay_memory=100000
force_more_update=7
tau_soft_update=0.001
alpha_a=0.1
alpha_v=0.01
inverting_grad=true
decay_v=-1
batch_norm_critic=0
batch_norm_actor=0
actor_output_layer_type=2
hidden_layer_type=1
test_net=false
bn_adapt=false
momentum=0
nstep_return=0
""")
    S, A, B, seed = 4, 1, 512, 21
    ag = ddrl_amd.DdpgAgents(ini, 1, S, A, seed=seed)
    rng = np.random.RandomState(0)
    wa = rng.normal(0, 0.1, (1, ag.pop.num_params(0))).astype(np.float32).astype(np.float64)
    wq = rng.normal(0, 0.1, (1, ag.pop.num_params(1))).astype(np.float32).astype(np.float64)
    for net, w in ((0, wa), (1, wq), (2, wa), (3, wq)):
        ag.pop.set_weights(net, w)
    o = ora.Ddpg(S, A, B, (20,), (50, 7), replay_cap=100000, gamma=0.99, tau=float(np.float32(0.001)), alpha_a=float(np.float32(0.1)),
                 alpha_v=float(np.float32(0.01)), seed=replay_seed(seed, 0))
    o.ann.set_weights(wa[0]); o.ann_target.set_weights(wa[0]); o.qnn.set_weights(wq[0]); o.qnn_target.set_weights(wq[0])
    s = rng.uniform(-1, 1, (1, S)).astype(np.float32)
    ag.start_episode(s, True)
    last_s, last_a, rounds = None, None, 0
    for t in range(1500):
        r = float(rng.normal())
        s = rng.uniform(-1, 1, (1, S)).astype(np.float32)
        act = ag.run(r, s, True, False, False)
        if last_a is not None and rounds == 0:
            o.insert(last_s, last_a, float(np.float32(r)), s[0].astype(np.float64), False)
            if o.replay_len() >= B:                          # the first update round (DeepQNAg.hpp:338-341): 8 passes
                used = [o.update()[1] for _ in range(8)]
                assert np.array_equal(ag.pop.last()["idx"][0], used[-1])          # same engine, same 8 x 512 draws
                for net in range(4):
                    w, wo = ag.pop.weights(net)[0], o.nets[net].weights()
                    assert close_frac(w, wo, 1e-3, 0.99, 0.5), net              # alpha_a = 0.1: Adam's sign steps are large
                rounds = 1
        last_s, last_a = s[0].astype(np.float64), act[0].astype(np.float64)
    n_rounds = 1499 - (B - 1)                                # transitions 512 .. 1499 each trigger a round
    assert ag.pop.replay_len()[0] == 1499
    assert ag.pop.adam(1)[2][0] == 8 * n_rounds and ag.pop.adam(0)[2][0] == 8 * n_rounds
    for net in range(4):
        assert np.all(np.isfinite(ag.pop.weights(net)))
    ag.close()
