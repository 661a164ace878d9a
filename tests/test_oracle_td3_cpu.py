"""CPU checks of the TD3 restatement (oracle/td3_ref.c, TD3Ag.hpp:284-380).  The reference holds no
test, fixture or golden vector for TD3Ag (parity unpinned beyond the MLP known-answer tests of
tests/test_oracle_kat.py); what can be pinned is its relation to the DDPG restatement and the
structural properties of TD3Ag::end_episode."""
import numpy as np

from ddpg_common import synth_transitions


def _pair(ora, seed=0, **kw):
    S, A, B = 6, 3, 16
    rng = np.random.RandomState(seed)
    mk = lambda td3: ora.Ddpg(S, A, B, (12,), (10, 7), replay_cap=400, inverting_grad=False, tau=0.05, alpha_a=1e-3,
                              alpha_v=2e-3, seed=5, td3=td3, **kw)
    d, t = mk(False), mk(True)
    wa = rng.normal(0, 0.3, d.ann.weights().shape); wq = rng.normal(0, 0.3, d.qnn.weights().shape)
    wq2 = rng.normal(0, 0.3, wq.shape)
    for o in (d, t):
        o.ann.set_weights(wa); o.ann_target.set_weights(wa); o.qnn.set_weights(wq); o.qnn_target.set_weights(wq)
    t.qnn2.set_weights(wq2); t.qnn_target2.set_weights(wq2)
    rows = synth_transitions(np.random.RandomState(3), 300, S, A, goal_prob=0.1).astype(np.float64)
    for row in rows:
        for o in (d, t):
            o.insert(row[:S], row[S:S + A], row[2 * S + A], row[S + A:2 * S + A], row[2 * S + A + 1] != 0)
    return d, t, (S, A, B)


def test_td3_with_equal_critics_and_no_noise_is_ddpg(ora):
    """min(Q', Q') = Q' and eps = 0: a policy step of TD3Ag is DeepQNAg's update without inverting gradients"""
    d, t, (S, A, B) = _pair(ora)
    t.qnn2.set_weights(t.qnn.weights()); t.qnn_target2.set_weights(t.qnn_target.weights())
    idx = np.random.RandomState(1).randint(0, 300, B).astype(np.int32)
    loss_d, _ = d.update(idx)
    loss1, loss2, used = t.update_td3(np.zeros((B, A)), actor_step=True, idx=idx)
    assert np.array_equal(used, idx) and loss1 == loss_d and loss2 == loss_d
    assert np.array_equal(t.last()[0], d.last()[0]) and np.array_equal(t.last()[1], d.last()[1])
    for n in range(4):
        assert np.array_equal(t.nets[n].weights(), d.nets[n].weights()), n
    assert np.array_equal(t.qnn2.weights(), t.qnn.weights()) and np.array_equal(t.qnn_target2.weights(), t.qnn_target.weights())


def test_td3_delayed_policy_step_and_sampling_stream(ora):
    d, t, (S, A, B) = _pair(ora, seed=2)
    before = [n.weights().copy() for n in t.nets]
    _, _, used = t.update_td3(np.zeros((B, A)), actor_step=False)
    _, want = d.update()                                   # same engine seed: the same B draws (TD3Ag.hpp:271-276)
    assert np.array_equal(used, want)
    after = [n.weights() for n in t.nets]
    for n in (0, 2, 3, 5):                                 # actor and all targets wait for the policy step (:333, :368-370)
        assert np.array_equal(after[n], before[n]), n
    for n in (1, 4):                                       # both critics learn every step (:329-330)
        assert not np.array_equal(after[n], before[n]), n
    assert t.ann.iter() == 0 and t.qnn.iter() == 1 and t.qnn2.iter() == 1
    t.update_td3(np.zeros((B, A)), actor_step=True)
    for n in (0, 2, 3, 5):
        assert not np.array_equal(t.nets[n].weights(), before[n]), n
    assert t.ann.iter() == 1 and t.qnn.iter() == 2


def test_td3_targets_use_the_smaller_critic_and_the_clipped_action(ora):
    d, t, (S, A, B) = _pair(ora, seed=4)
    idx = np.arange(B, dtype=np.int32)
    # y = r + gamma min(Q1', Q2') recomputed by hand from the nets' own forward passes
    noise = np.random.RandomState(9).normal(0, 2.0, (B, A))       # large: a' saturates at +-1 often (:309)
    w = {n: t.nets[n].weights().copy() for n in range(6)}
    t.update_td3(noise, actor_step=False, idx=idx)
    y = t.last()[0]
    probe = ora.Ddpg(S, A, B, (12,), (10, 7), replay_cap=400, td3=True)
    for n in range(6):
        probe.nets[n].set_weights(w[n])
    rep = synth_transitions(np.random.RandomState(3), 300, S, A, goal_prob=0.1).astype(np.float64)[:B]
    a2 = np.clip(probe.ann_target.forward(rep[:, S + A:2 * S + A]) + noise, -1.0, 1.0)
    x = np.concatenate([rep[:, S + A:2 * S + A], a2], axis=1)
    q = np.minimum(probe.qnn_target.forward(x)[:, 0], probe.qnn_target2.forward(x)[:, 0])
    want = np.where(rep[:, 2 * S + A + 1] != 0, rep[:, 2 * S + A], rep[:, 2 * S + A] + 0.99 * q)
    assert np.any(np.abs(a2) == 1.0) and np.any(rep[:, 2 * S + A + 1] != 0)
    np.testing.assert_allclose(y, want, rtol=1e-13, atol=0)
