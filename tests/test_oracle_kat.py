"""Pins the CPU oracle against the reference's own known-answer tests.

common/src/test/CaffeUtest.cpp:24-55  (MLP.ConsistentActivationFunction): flat parameter order
{W1,b1,W2,b2}, tanh / leaky-ReLU(0.01) / linear output semantics.
common/src/test/CaffeUtest.cpp:372-386 (MLP.CaffeCopyWeightsActor): a 1-8-1 actor has 25 params.
common/src/test/CaffeUtest.cpp:57-100 (MLP.LearnOpposite): exact outputs before learning and
convergence after 1000 online Adam steps.
"""
import math

import numpy as np


def test_consistent_activation_function(ora):
    # MLP nn(1, 1, {1}, 0.01f, 1, -1, 2, 0): critic ctor, input=1, sensors=1 (no actions), hidden {1}, tanh
    nn = ora.Mlp([1, 1, 1], hidden_act=2, out_act=0, lr=0.01)
    assert nn.np == 4
    nn.set_weights([1, 0, 1, 0])
    assert nn.forward([[1.0]])[0, 0] == math.tanh(1.0)
    nn2 = ora.Mlp([1, 1, 1], hidden_act=1, out_act=0, lr=0.01)
    nn2.set_weights([1, 0, 1, 0])
    assert nn2.forward([[1.0]])[0, 0] == 1.0
    v = nn2.forward([[-1.0]])[0, 0]
    assert -0.01 - 0.001 <= v <= -0.01 + 0.001
    nn2.set_weights([0, 0, 0, 5])
    assert nn2.forward([[-1.0]])[0, 0] == 5.0
    nn.set_weights([0.2, 0.4, 0.6, 0.8])
    ref = math.tanh(-1.0 * np.float32(0.2) + np.float32(0.4)) * np.float32(0.6) + np.float32(0.8)
    assert abs(nn.forward([[-1.0]])[0, 0] - ref) <= 1e-4


def test_actor_param_count(ora):
    # MLP actor(1, {8}, 1, ...) -> 25 parameters (CaffeUtest.cpp:380-383)
    assert ora.Mlp([1, 8, 1], hidden_act=1, out_act=2).np == 25


def test_learn_opposite(ora):
    nn = ora.Mlp([1, 1, 1], hidden_act=2, out_act=0, lr=0.01)
    nn.set_weights([-1, 0.0, 1.0 / math.tanh(1.0), 0])
    for x in (1.0, -1.0):
        # EXPECT_DOUBLE_EQ = within 4 ulps
        assert abs(nn.forward([[x]])[0, 0] - (-x)) <= 4 * np.finfo(np.float64).eps
    rng = np.random.RandomState(0)
    for _ in range(1000):
        x = 1.0 if rng.rand() < 0.5 else -1.0
        nn.learn_batch([[x]], [[-x]], 1)
    for x in (1.0, -1.0):
        assert abs(nn.forward([[x]])[0, 0] - (-x)) < 1e-6


def test_learn_and_or(ora):
    # CaffeUtest.cpp:102-139: 3-10-1 tanh critic, 10000 online Adam steps lr 0.01, +-0.16
    nn = ora.Mlp([3, 10, 1], hidden_act=2, out_act=0, lr=0.01)
    rng = np.random.RandomState(1)
    nn.set_weights(rng.normal(0, 0.01, nn.np))
    f = lambda a, b, c: 1.0 if ((a == 1 or b == 1) and c == 1) else -1.0
    for _ in range(10000):
        x = np.where(rng.rand(3) < 0.5, 1.0, -1.0)
        nn.learn_batch([x], [[f(*x)]], 1)
    for a in (-1.0, 1.0):
        for b in (-1.0, 1.0):
            for c in (-1.0, 1.0):
                assert abs(nn.forward([[a, b, c]])[0, 0] - f(a, b, c)) < 0.16


def test_gradients_match_finite_differences(ora):
    rng = np.random.RandomState(2)
    for hid, out in ((1, 0), (2, 2), (3, 0)):
        nn = ora.Mlp([5, 7, 6, 3], hidden_act=hid, out_act=out)
        w = rng.normal(0, 0.5, nn.np)
        nn.set_weights(w)
        x = rng.uniform(-1, 1, (4, 5))
        c = rng.normal(0, 1, (4, 3))
        nn.zero_grad()
        nn.forward(x)
        din = nn.backward(c)
        g = nn.grads()
        loss = lambda: float((nn.forward(x) * c).sum())
        eps = 1e-6
        for k in rng.choice(nn.np, 25, replace=False):
            w2 = w.copy(); w2[k] += eps; nn.set_weights(w2); lp = loss()
            w2[k] -= 2 * eps; nn.set_weights(w2); lm = loss()
            assert abs((lp - lm) / (2 * eps) - g[k]) < 1e-6 * max(1.0, abs(g[k]))
        nn.set_weights(w)
        xp = x.copy(); xp[1, 2] += eps; lp = float((nn.forward(xp) * c).sum())
        xp[1, 2] -= 2 * eps; lm = float((nn.forward(xp) * c).sum())
        assert abs((lp - lm) / (2 * eps) - din[1, 2]) < 1e-6


def test_adam_matches_hand_written_numpy(ora):
    """Caffe AdamSolver: lr_t = lr*sqrt(1-b2^t)/(1-b1^t), eps added to sqrt(v)."""
    rng = np.random.RandomState(3)
    for momentum, (b1, b2) in enumerate([(0, .999), (.9, .999), (0, 0), (.9, 0)]):
        nn = ora.Mlp([4, 6, 2], hidden_act=1, out_act=0, lr=1e-3, momentum=momentum, decay=0.01)
        w = rng.normal(0, 0.3, nn.np); nn.set_weights(w)
        m = np.zeros(nn.np); v = np.zeros(nn.np)
        x = rng.uniform(-1, 1, (8, 4)); y = rng.normal(0, 1, (8, 2))
        for t in range(1, 6):
            probe = ora.Mlp([4, 6, 2], hidden_act=1, out_act=0)
            probe.set_weights(w)
            out = probe.forward(x)
            probe.backward((out - y) / 8.0, want_din=False)
            g = probe.grads() + 0.01 * w
            m = b1 * m + (1 - b1) * g
            v = b2 * v + (1 - b2) * g * g
            w = w - 1e-3 * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + 1e-8)
            nn.learn_batch(x, y, 1)
            np.testing.assert_allclose(nn.weights(), w, rtol=1e-12, atol=1e-15)
