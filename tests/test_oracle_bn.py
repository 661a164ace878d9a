"""BatchNorm / Scale towers and TRAIN / TEST twins of the CPU oracle (oracle/mlp_ref.c), pinned by the consistency
properties the reference's own tests state: CaffeUtest.cpp:246-386 (CaffeSaveLoad, CaffeCopyActor, CaffeCopyWeightsActor),
:388-467 (BackwardActor / BackwardCritic run), :1343-1447 (CaffeBNTesting).  The layer arithmetic is BVLC Caffe 1.0's
BatchNormLayer / ScaleLayer (published algorithm; the fork is absent), checked here by finite differences where the
reference's in-place layering leaves the analytic gradient exact (towers that follow no activation)."""
import numpy as np
import pytest

BNS = [0, 4, 6, 7, 8, 10, 11, 12, 14, 15]


def grid1(rng, n):
    x = np.linspace(-1, 1, n)
    r = rng.uniform(0, 1, n) < 0.4
    return np.where(r, rng.uniform(-1, 1, n), x)


def expected_params(dims, code, critic_sensors=0):
    """learnable_params() count: InnerProducts + per tower 2C + 1 (+ C gamma (+ C beta))"""
    L = len(dims) - 1
    H = L - 1
    n = sum(dims[l] * dims[l + 1] + dims[l + 1] for l in range(L))
    b0, b1, b2, b3, b4 = [(code >> k) & 1 for k in range(5)]
    ranks = []
    if b4 and not b2: ranks = [(0, critic_sensors)]
    elif b4: ranks = [(r, dims[r]) for r in range(H + 1)] + ([(L, dims[L])] if critic_sensors == 0 else [])
    elif b2 and b3: ranks = [(r, dims[r]) for r in range(H + 1)]
    elif b2: ranks = [(0, dims[0])]
    elif b3: ranks = [(r, dims[r]) for r in range(H)]
    for _, c in ranks:
        n += 2 * c + 1 + (c if b1 else 0) + (c if b0 else 0)
    return n


@pytest.mark.parametrize("bn", BNS)
def test_parameter_count(ora, bn):
    """CaffeUtest.cpp:372-386: 1 -> 8 -> 1 has 25 parameters without towers and more with any"""
    for hidden_act in (1, 2, 3):
        a = ora.Mlp([1, 8, 1], hidden_act, 0, 0.01, batch_norm=bn)
        assert a.np == expected_params([1, 8, 1], bn)
        assert (a.np == 25) == (bn == 0)
    q = ora.Mlp([2, 20, 10, 1], 1, 0, 0.005, batch_norm=bn, critic_sensors=1)
    assert q.np == expected_params([2, 20, 10, 1], bn, 1)
    for code in (16, 18, 19, 20, 22, 23):          # first_except_action / very_all
        assert ora.Mlp([5, 7, 3, 1], 1, 0, 0.01, batch_norm=code, critic_sensors=3).np == expected_params([5, 7, 3, 1], code, 3)
        if code & 4:
            assert ora.Mlp([5, 7, 3, 2], 1, 2, 0.01, batch_norm=code).np == expected_params([5, 7, 3, 2], code, 0)


@pytest.mark.parametrize("bn", BNS)
def test_copy_actor_twins(ora, bn):
    """CaffeUtest.cpp:322-370: TRAIN copies agree with the net, TEST copies agree with each other, over five forwards
    during which the TRAIN nets' statistics learn; without towers everything agrees."""
    rng = np.random.RandomState(bn)
    for hidden_act in (1, 2, 3):
        actor = ora.Mlp([1, 8, 1], hidden_act, 0, 0.01, batch_norm=bn)
        w = actor.weights()
        lay = actor.bn_layout()
        for l in range(2):
            n = actor.dims[l] * actor.dims[l + 1]
            w[lay["offw"][l]:lay["offw"][l] + n] = rng.normal(0, 0.5, n)
            w[lay["offb"][l]:lay["offb"][l] + actor.dims[l + 1]] = rng.normal(0, 0.5, actor.dims[l + 1])
        actor.set_weights(w)
        t1 = actor.clone()                 # MLP(actor, true): train
        t2 = actor.clone(test=True)        # MLP(actor, false, TEST)
        t3 = actor.clone(test=True)        # MLP(actor, true, TEST)
        t4 = t2.clone(test=True)
        x = grid1(rng, 400).reshape(400, 1)
        for forced in range(5):
            o, o1, o2, o3, o4 = [n.forward(x) for n in (actor, t1, t2, t3, t4)]
            assert np.array_equal(o, o1) and np.array_equal(o3, o2) and np.array_equal(o3, o4)
            if bn == 0:
                assert np.array_equal(o1, o2)
            else:
                assert not np.array_equal(o1, o2)      # fresh TEST copies: factor 0 => mean 0, var 0 (x / sqrt(eps))


@pytest.mark.parametrize("bn", BNS)
def test_save_load_twins_critic(ora, bn):
    """CaffeUtest.cpp:246-320: after 5 learning steps a net rebuilt from the saved parameters, and copies of it, compute what
    the original and its copies compute -- TRAIN nets compared with TRAIN nets, TEST with TEST (batch of one, as computeOutVF
    pads: here a batch of 4 rows, identical for every net)."""
    rng = np.random.RandomState(100 + bn)
    for hidden_act in (1, 2, 3):
        nn = ora.Mlp([2, 20, 10, 1], hidden_act, 0, 0.005, batch_norm=bn, critic_sensors=1)
        w = nn.weights(); lay = nn.bn_layout()
        for l in range(3):
            n = nn.dims[l] * nn.dims[l + 1]
            w[lay["offw"][l]:lay["offw"][l] + n] = rng.normal(0, 0.3, n)
        nn.set_weights(w)
        sa = np.stack([grid1(rng, 400), grid1(rng, 400)], 1)
        q = -(sa[:, 1] ** 2 - sa[:, 0] ** 2)
        nn.learn_batch(sa, q, 5)
        assert np.all(np.isfinite(nn.weights()))
        nn_test = nn.clone(test=True)
        nn2 = ora.Mlp([2, 20, 10, 1], hidden_act, 0, 0.005, batch_norm=bn, critic_sensors=1)
        nn2.set_weights(nn.weights())                        # save / load = every parameter incl. the statistics
        nn3 = nn2.clone(test=True)
        nn4 = nn2.clone()
        for n in range(20):
            x = rng.uniform(-1, 1, (4, 2))
            a, b, c = nn.forward(x), nn2.forward(x), nn4.forward(x)
            assert np.array_equal(a, b) and np.array_equal(b, c)
            assert np.array_equal(nn_test.forward(x), nn3.forward(x))


@pytest.mark.parametrize("bn", BNS[1:])
def test_bn_testing(ora, bn):
    """CaffeUtest.cpp:1343-1447: the statistics of a TRAIN net move with every forward (sum |dw| > 2 after seven), a TEST copy
    made afterwards equals an earlier TEST copy that received copyWeightsFrom(copyWeightsTo) (diff4 == 0), and the TEST
    copies' parameters differ from what they were."""
    rng = np.random.RandomState(bn)
    actor = ora.Mlp([1, 50, 25, 1], 1, 0, 0.01, batch_norm=bn)
    w = actor.weights(); lay = actor.bn_layout()
    for l in range(3):
        n = actor.dims[l] * actor.dims[l + 1]
        w[lay["offw"][l]:lay["offw"][l] + n] = rng.normal(0, 0.3, n)
    actor.set_weights(w)
    dontlearn = actor.clone(test=True)
    assert dontlearn.np == actor.np
    x1 = grid1(rng, 400).reshape(400, 1)
    x2 = 5.0 + x1
    o_first = actor.forward(x1)
    for x in (x2, x1, x2, x1, x2):
        actor.forward(x)
    o_again = actor.forward(x1)
    assert np.array_equal(o_first, o_again)                 # "BN output is the same over every same batch" (TRAIN: batch statistics)
    assert np.abs(actor.weights() - dontlearn.weights()).sum() > 2.0
    actor.forward(x1)
    after = actor.clone(test=True)
    stale = dontlearn.forward(x1)
    dontlearn.set_weights(actor.weights())
    o2, o5 = dontlearn.forward(x1), after.forward(x1)
    assert np.array_equal(o2, o5)                            # diff4 == 0
    assert not np.array_equal(stale, o2)
    assert not np.array_equal(actor.forward(x1), o2)         # "BN is not the same in learning and testing"


def _loss(net, x, dout):
    return float((net.forward(x) * dout).sum())


@pytest.mark.parametrize("bn,critic_s,test", [(4, 0, False), (6, 0, False), (7, 0, False), (7, 0, True), (19, 2, False), (18, 2, False), (16, 2, True)])
def test_backward_against_finite_differences(ora, bn, critic_s, test):
    """towers that follow no in-place activation (rank 0): the analytic backward is the exact gradient.  (Behind an activation
    the reference's in-place layering feeds the activation's backward the tower's output, MLP.hpp:756-803 -- restated as is,
    not a true gradient, so not checked this way.)"""
    rng = np.random.RandomState(bn)
    dims = [5, 6, 4, 3 if critic_s == 0 else 1]
    net = ora.Mlp(dims, 2, 2 if critic_s == 0 else 0, 0.01, batch_norm=bn, critic_sensors=critic_s)
    lay = net.bn_layout()
    w = rng.normal(0, 0.5, net.np)
    w[lay["offbn"][0]:lay["offbn"][0] + 2 * lay["chan"][0] + 1] = np.abs(w[lay["offbn"][0]:lay["offbn"][0] + 2 * lay["chan"][0] + 1]) + 0.5
    net.set_weights(w)
    net.set_phase(test)
    B = 9
    x = rng.normal(0, 1, (B, dims[0])); dout = rng.normal(0, 1, (B, dims[-1]))
    net.set_weights(w); net.zero_grad(); net.forward(x)
    din = net.backward(dout, want_din=True)
    g = net.grads()
    learn = [i for i in range(net.np) if not (lay["offbn"][0] <= i < lay["offbn"][0] + 2 * lay["chan"][0] + 1)]
    eps = 1e-6
    for i in learn[::3]:
        wp, wm = w.copy(), w.copy(); wp[i] += eps; wm[i] -= eps
        net.set_weights(wp); lp = _loss(net, x, dout)
        net.set_weights(wm); lm = _loss(net, x, dout)
        assert abs((lp - lm) / (2 * eps) - g[i]) <= 1e-6 * max(1.0, abs(g[i])), (i, g[i])
    for i in range(0, B * dims[0], 4):
        xp, xm = x.copy().ravel(), x.copy().ravel(); xp[i] += eps; xm[i] -= eps
        net.set_weights(w); lp = _loss(net, xp.reshape(x.shape), dout)
        net.set_weights(w); lm = _loss(net, xm.reshape(x.shape), dout)
        assert abs((lp - lm) / (2 * eps) - din.ravel()[i]) <= 1e-6 * max(1.0, abs(din.ravel()[i])), i


@pytest.mark.parametrize("bn", BNS)
def test_backward_runs_and_frozen_statistics(ora, bn):
    """CaffeUtest.cpp:388-467 (valgrind-only in the reference): forward / backward / ApplyUpdate run for every code; the
    statistics blobs have lr_mult 0 -- only forwards move them."""
    rng = np.random.RandomState(bn)
    for hidden_act in (1, 2, 3):
        actor = ora.Mlp([1, 8, 1], hidden_act, 2, 0.01, batch_norm=bn)
        actor.set_weights(actor.weights() + rng.normal(0, 0.3, actor.np) * (actor.weights() == 0))
        lay = actor.bn_layout()
        for forced in range(5):
            x = grid1(rng, 400).reshape(400, 1)
            actor.forward(x)
            before = actor.weights()
            actor.zero_grad()
            actor.backward(rng.normal(0, 1, (400, 1)))
            actor.apply_update()
            after = actor.weights()
            assert np.all(np.isfinite(after)) and not np.array_equal(before, after)
            for r in range(3):
                if lay["rank_on"][r]:
                    sl = slice(lay["offbn"][r], lay["offbn"][r] + 2 * lay["chan"][r] + 1)
                    assert np.array_equal(before[sl], after[sl])
