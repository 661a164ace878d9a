"""DODevMLP's developmental schedule in the agent front end (ddrl_agents_*, [devnn] keys) on the GPU:
heuristic_devpoints switches, the intrinsic-motivation variant, reset_learning_algo, ewc_setup at a
switch, neutral_action (DODevMLP.hpp:403-574, 764-775; DeepQNAg.hpp:250-272).  The schedule is host
logic; expectations are written out by hand for the devpoints case and restated from the reference's
developIM for the other.  The one reference test that drives inform() (CaffeUtest.cpp:1257-1341, devpoints 0,0,30,30) is restated
in tests/test_dev_layer_fixtures_gpu.py; the schedules beyond it (later devpoints, developIM, resets) are parity unpinned."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AGENT = """[agent]
gamma=0.9
decision_each=1
noise=0.0
gaussian_policy=1
hidden_unit_q=12
hidden_unit_a=10
mini_batch_size=16
replay_memory=1000
inverting_grad=false
force_more_update=0
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

DEVNN = """[devnn]
st_scale=false
ac_scale=false
st_probabilistic=1
ac_probabilistic=1
st_control={st}
ac_control={ac}
heuristic={heur}
heuristic_devpoints={dp}
reset_learning_algo={reset}
informed_sensorimotor_space={informed}
pd_controller_factor=0.1:-0.2:0.3
intrasec_motivation={im}
im_smooth=2
im_window=1
ewc={ewc}
ewc_decay=-1
ewc_fisher_beta=0.99
ewc_best_param_method=1
ewc_for_critic=2
"""


def write(tmp_path, **kw):
    d = dict(st="1:3", ac="2", heur=1, dp="2:4:3", reset="true", informed="true", im="false", ewc=-1)
    d.update(kw)
    p = os.path.join(str(tmp_path), "d.ini")
    with open(p, "w") as f:
        f.write(AGENT + DEVNN.format(**d))
    return p


def episode(ag, rng, learn, steps=4, rewards=None):
    G, S = ag.G, ag.S
    ag.start_episode(rng.uniform(-1, 1, (G, S)), learn)
    acts = []
    for t in range(steps):
        r = rng.normal(0, 1, G) if rewards is None else rewards[t]
        acts.append(ag.run(r, rng.uniform(-1, 1, (G, S)), learn, False, t == steps - 1))
    return np.stack(acts)


def test_devpoints_switches_reset_and_neutral_action(tmp_path):
    import ddrl_amd
    G, S, A = 3, 5, 3
    ag = ddrl_amd.DdpgAgents(write(tmp_path, ewc=0.5), G, S, A, seed=1)
    rng = np.random.RandomState(0)
    ag.pop.set_weights(0, rng.normal(0, 0.5, (G, ag.pop.num_params(0))))
    ag.pop.set_weights(2, ag.pop.weights(0))
    sg, agt = ag.pop.gates()
    assert np.array_equal(sg, np.tile([1, 0, 1, 0, 1], (G, 1))) and np.array_equal(agt, np.tile([1, 1, 0], (G, 1)))
    # agent 2 only tests during the first two episodes: its own episode counter lags by two
    learn_seq = [np.array([1, 1, 0], np.uint8)] * 2 + [np.array([1, 1, 1], np.uint8)] * 6
    ep = np.zeros(G, int)
    for e, learn in enumerate(learn_seq):
        acts = episode(ag, rng, learn, steps=130)
        sg, agt = ag.pop.gates()
        for g in range(G):
            k = ep[g] if learn[g] else ep[g] - 1            # inform() only runs on learning episodes
            # state dim 1 opens at episode 2, action dim 2 at 3, state dim 3 at 4 (devpoints 2:4:3)
            if learn[g]:
                exp_s = [1, float(ep[g] >= 2), 1, float(ep[g] >= 4), 1]
                exp_a = [1, 1, float(ep[g] >= 3)]
                assert list(sg[g]) == exp_s and list(agt[g]) == exp_a, (e, g, k)
                if ep[g] < 3:                                  # closed action dimension: the PD set point, exactly
                    assert np.all(acts[:, g, 2] == np.float32(0.3)), (e, g)
                else:
                    assert np.all(acts[:, g, 2] != np.float32(0.3)), (e, g)
                assert np.all(np.abs(acts[:, g, :2]) <= 1)
        ag.end_episode(learn)
        ep += learn
    # reset_learning_algo: the replay restarts at a switch => after the switch at episode 4 agents 0 and 1
    # hold the samples of episodes 4..7 only (129 transitions per 130-step episode), agent 2 those of its 4, 5
    L = ddrl_amd.lib.load()
    ln = np.empty(G, np.int32)
    assert L.ddrl_pop_replay_len(ag.pop.h, ln.ctypes.data) == 0
    assert list(ln) == [516, 516, 258]
    # ewc_setup ran for the actor and (ewc_for_critic = 2) the critic at the later switches: anchors are live
    for net in (0, 1):
        st = ag.pop.ewc_state(net)
        assert np.all(np.abs(st["wstar"]).sum(axis=1) > 0), net
        assert np.all(np.isfinite(st["fisher_hat"])) and np.all(st["fisher_hat"].max(axis=1) == 1.0), net
        assert np.all(np.isfinite(ag.pop.weights(net)))
    ag.close()


def test_intrinsic_motivation_schedule(tmp_path):
    import ddrl_amd
    G, S, A = 2, 5, 3
    # groups: devpoints value = group index: state dim 1 in group 0, action dim 2 in group 1, state dim 3 in group 2
    ag = ddrl_amd.NfacAgents(write_nfac(tmp_path), G, S, A, max_trajectory=64, seed=2)
    rng = np.random.RandomState(1)
    st_w, ac_w = np.zeros((G, 2)), np.zeros((G, 1))
    scores, last_changed, im_index = [[] for _ in range(G)], [0] * G, [0] * G
    dp, smooth, window = [0, 2, 1], 2, 1
    last_swr = np.zeros(G)

    def update_dl(g):
        for i, k in enumerate(dp):
            if k <= im_index[g]:
                w = st_w[g] if i < 2 else ac_w[g]
                j = i if i < 2 else i - 2
                if w[j] < 0.5:
                    w[j] = 1.0
        im_index[g] += 1

    def still(g):
        for i, k in enumerate(dp):
            if k <= im_index[g] + 1:
                if (st_w[g][i] if i < 2 else ac_w[g][i - 2]) == 0.0:
                    return True
        return False

    def inform(g, episode, score):                              # DODevMLP::developIM (DODevMLP.hpp:472-515)
        if episode == 0:
            update_dl(g)
            return True
        ep = episode - last_changed[g] - 1
        scores[g].append(score)
        ne = sum(-1e8 if ep - i < 0 else scores[g][ep - i] for i in range(smooth)) / smooth
        nw = sum(-1e8 if ep - i - window < 0 else scores[g][ep - i - window] for i in range(smooth)) / smooth
        ch = ne - nw <= 0 and ep >= smooth + window and still(g)
        if ch:
            last_changed[g] += len(scores[g])
            update_dl(g)
            scores[g] = []
        return ch

    # returns rise for a while, then stall => a switch; agent 1 sees a different sequence
    plan = [[1, 2, 3, 4, 4, 3, 5, 6, 7, 7, 6, 8, 9, 9, 9, 9], [5, 4, 3, 2, 1, 2, 3, 4, 5, 6, 7, 7, 7, 7, 8, 9]]
    n_changes = [0, 0]
    for e in range(len(plan[0])):
        for g in range(G):
            if inform(g, e, last_swr[g]):
                n_changes[g] += 1
        r = np.array([plan[0][e], plan[1][e]], np.float64)
        acts = episode(ag, rng, True, steps=2, rewards=[r, np.zeros(G)])
        sg, agt = ag.pop.gates()
        for g in range(G):
            assert list(sg[g]) == [1, float(st_w[g][0] >= 0.5), 1, float(st_w[g][1] >= 0.5), 1], (e, g)
            assert list(agt[g]) == [1, 1, float(ac_w[g][0] >= 0.5)], (e, g)
        ag.end_episode(True)
        last_swr = ag.stats()["sum_weighted_reward"].copy()
        assert list(last_swr) == list(r)
    assert min(n_changes) >= 2 and n_changes[0] != 0      # the plan exercises switches beyond episode 0
    assert np.all(st_w[0] == 1) and np.all(ac_w[0] == 1)
    ag.close()


NFAC_AGENT = """[agent]
gamma=0.9
decision_each=1
noise=0.1
gaussian_policy=1
hidden_unit_v=12
hidden_unit_a=10
update_delta_neg=false
vnn_from_scratch=false
update_critic_first=true
number_fitted_iteration=1
stoch_iter_actor=1
stoch_iter_critic=1
batch_norm_actor=0
batch_norm_critic=0
actor_output_layer_type=2
hidden_layer_type=1
alpha_a=0.001
alpha_v=0.002
lambda=0.8
gae=true
momentum=0
"""


def write_nfac(tmp_path):
    p = os.path.join(str(tmp_path), "n.ini")
    with open(p, "w") as f:
        f.write(NFAC_AGENT + DEVNN.format(st="1:3", ac="2", heur=1, dp="0:2:1", reset="false", informed="false", im="true", ewc=-1))
    return p


@pytest.mark.parametrize("kw,msg", [
    (dict(heur=0), "learning alone"), (dict(heur=0, reset="false"), None), (dict(heur=2), "probabilistic"), (dict(dp="1:2"), "heuristic_devpoints"),
    (dict(st="1:9"), "st_control"), (dict(im="true", heur=1, dp="0:1:2", reset="true"), None)])
def test_config_checks(tmp_path, kw, msg):
    import ddrl_amd
    p = write(tmp_path, **kw)
    if msg is None:
        ddrl_amd.DdpgAgents(p, 2, 5, 3).close()             # a valid intrinsic-motivation config for DDPG
        return
    with pytest.raises(ddrl_amd.DdrlError, match=msg):
        ddrl_amd.DdpgAgents(p, 2, 5, 3)
