"""The TD3 agent front end (ddrl_agents_create_td3, TD3Ag_*) on the GPU against a Python restatement of
TD3Ag's _run / insertSample / end_episode / end_instance (TD3Ag.hpp:121-163, 233-244, 278-380) driving
the fp64 oracle: one update per stored transition, the actor and the targets on every policy_freq-th
step of the episode, one more update at the episode's end.  policy_noise = 0 makes the target-policy
smoothing a no-op on both sides (the reference's stream is clock-seeded).  Update counts, replay
length and episode counters must be exact; weights are a free-running fp32-vs-fp64 comparison
(1e-3 on >= 99 %).  No reference test pins this layer: parity unpinned."""
import os

import numpy as np
import pytest

from ddpg_common import close_frac, oracle_weights

pytestmark = pytest.mark.gpu

INI = """[agent]
gamma=0.97
decision_each=1
noise=0.0
gaussian_policy=0
hidden_unit_q=24:12
hidden_unit_a=16
mini_batch_size=32
replay_memory=400
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
policy_noise={pn}
noise_clip=0.5
policy_freq={pf}
"""


def write_ini(tmp_path, pn=0.0, pf=3):
    p = os.path.join(str(tmp_path), "td3.ini")
    with open(p, "w") as f:
        f.write(INI.format(pn=pn, pf=pf))
    return p


def test_td3_agents_follow_reference_schedule(ora, tmp_path):
    import ddrl_amd
    G, S, A, B, seed, pf = 3, 5, 2, 32, 11, 3
    ag = ddrl_amd.Td3Agents(write_ini(tmp_path, pf=pf), G, S, A, seed=seed)
    rng = np.random.RandomState(0)
    na, nq = ag.pop.num_params(0), ag.pop.num_params(1)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    wa, wq, wq2 = f32(rng.normal(0, 0.3, (G, na))), f32(rng.normal(0, 0.3, (G, nq))), f32(rng.normal(0, 0.3, (G, nq)))
    W = ((0, wa), (1, wq), (2, wa), (3, wq), (4, wq2), (5, wq2))
    for net, w in W:
        ag.pop.set_weights(net, w)
    refs = []
    for g in range(G):
        o = ora.Ddpg(S, A, B, (16,), (24, 12), replay_cap=400, gamma=0.97, tau=float(np.float32(0.01)), inverting_grad=False,
                     alpha_a=float(np.float32(0.001)), alpha_v=float(np.float32(0.002)),
                     seed=(seed * 40503 + 7919 * g + 17) & 0xFFFFFFFF, td3=True)
        for net, w in W:
            o.nets[net].set_weights(w[g])
        refs.append(dict(o=o, counter=0, last_s=None, last_a=None, ep=0))
    zero = np.zeros((B, A))

    def ref_update(st):                                   # TD3Ag::end_episode(true)
        o = st["o"]
        if o.replay_len() >= 250 and o.replay_len() >= B:
            o.update_td3(zero, actor_step=st["counter"] % pf == 0)

    for episode in range(2):
        steps = 300 if episode == 0 else 41
        s0 = rng.uniform(-1, 1, (G, S)).astype(np.float32)
        ag.start_episode(s0, True)
        for g, st in enumerate(refs):
            st.update(counter=0, last_a=None, last_s=s0[g].astype(np.float64))
        for t in range(steps):
            r = rng.normal(0, 1, G).astype(np.float32)
            s = rng.uniform(-1, 1, (G, S)).astype(np.float32)
            goal = rng.uniform(0, 1, G) < 0.02
            act = ag.run(r, s, True, goal, np.full(G, t == steps - 1))
            for g, st in enumerate(refs):
                if st["last_a"] is not None:
                    st["o"].insert(st["last_s"], st["last_a"], float(r[g]), s[g].astype(np.float64), bool(goal[g]))
                    st["counter"] += 1                    # insertSample (:156-163)
                    ref_update(st)
                st["last_a"] = act[g].astype(np.float64)
                st["last_s"] = s[g].astype(np.float64)
        ag.end_episode(True)
        for st in refs:
            ref_update(st)
            st["ep"] += 1
        assert list(ag.stats()["episode"]) == [st["ep"] for st in refs]
    assert list(ag.pop.replay_len()) == [st["o"].replay_len() for st in refs]
    for net in (0, 1, 4):                                 # exact update counts: critics every step, the actor every pf-th
        it = ag.pop.adam(net)[2]
        assert list(it) == [st["o"].nets[net].iter() for st in refs], net
    assert ag.pop.adam(1)[2][0] > 60 and ag.pop.adam(0)[2][0] * 2 < ag.pop.adam(1)[2][0]
    for net in range(6):
        w = ag.pop.weights(net)
        wo = oracle_weights([st["o"] for st in refs], net)
        for g in range(G):
            assert close_frac(w[g], wo[g], 1e-3, 0.99, 1e-1), (net, g)
    ag.close()


def test_td3_agents_with_towers(tmp_path):
    """the front end over an ini with TD3Ag's batch_norm keys (TD3Ag.hpp:176-181): loads, acts through the TEST-phase acting copy,
    keeps the reference's update schedule (both critics every stored transition, the actor every policy_freq-th) and stays
    finite; the arithmetic is held to the oracle in tests/test_bn_gpu.py::test_td3_update_with_towers"""
    import ddrl_amd
    G, S, A, pf = 2, 5, 2, 2
    ini = os.path.join(str(tmp_path), "td3bn.ini")
    with open(ini, "w") as f:
        f.write(INI.format(pn=0.2, pf=pf).replace("batch_norm_critic=0", "batch_norm_critic=11").replace("batch_norm_actor=0", "batch_norm_actor=14")
                .replace("bn_adapt=false", "bn_adapt=true"))
    ag = ddrl_amd.Td3Agents(ini, G, S, A, seed=5)
    rng = np.random.RandomState(1)
    ag.start_episode(rng.uniform(-1, 1, (G, S)).astype(np.float32), True)
    steps = 300
    for t in range(steps):
        act = ag.run(rng.normal(0, 1, G).astype(np.float32), rng.uniform(-1, 1, (G, S)).astype(np.float32), True,
                     np.zeros(G, bool), np.full(G, t == steps - 1))
        assert np.all(np.isfinite(act)) and np.all(np.abs(act) <= 1.0)
    ag.end_episode(True)
    itq, itq2, ita = ag.pop.adam(1)[2], ag.pop.adam(4)[2], ag.pop.adam(0)[2]
    assert list(itq) == list(itq2) and itq[0] >= 40 and abs(int(ita[0]) * pf - int(itq[0])) <= pf
    for net in range(6):
        assert np.all(np.isfinite(ag.pop.weights(net)))
    ag.close()


def test_td3_c_flavour(tmp_path):
    """the reference's libddrl-td3 calls (ddpg.cpp built with WANT_TD3), here under the TD3Ag_ prefix"""
    import ddrl_amd
    S, A = 4, 2
    ini = write_ini(tmp_path, pn=0.2, pf=2)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        a = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "3"], algo="td3")
        assert a.name() == "TD3"
        rng = np.random.RandomState(1)
        s = rng.uniform(-1, 1, S)
        a.start_ep(s, True)
        first = a.run(0.0, s, True, False, False)
        assert first.shape == (A,) and np.all(np.abs(first) <= 1)
        for t in range(300):
            a.run(float(rng.normal()), rng.uniform(-1, 1, S), True, False, t == 299)
        a.end_ep(True)
        probe = rng.uniform(-1, 1, S)
        a.start_ep(probe, False)
        before = a.run(0.0, probe, False, False, False)
        assert not np.array_equal(before, first)
        a.save(3)
        b = ddrl_amd.DDRLAg(A, S, ["--config", ini, "--seed", "99"], algo="td3")
        b.load(3)
        b.start_ep(probe, False)
        assert np.array_equal(b.run(0.0, probe, False, False, False), before)
        # a DDPG ini lacks TD3Ag's keys: reported, not fatal
        c = ddrl_amd.DDRLAg(A, S, ["--config", os.path.join(str(tmp_path), "nope.ini")], algo="td3")
        with pytest.raises(ddrl_amd.DdrlError):
            c.run(0.0, probe, False, False, False)
        for x in (a, b, c):
            x.close()
    finally:
        os.chdir(cwd)
