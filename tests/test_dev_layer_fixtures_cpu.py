"""The reference's Developmental-layer unit tests restated on the CPU oracle (oracle/devmlp.py): CaffeUtest.cpp:619-690, 692-802,
916-1046 (counts, order, shared blobs, pass rates) and :1257-1341 (heuristic 1, devpoints 0,0,30,30, inform(0)).  The GPU side of
the same tests is tests/test_dev_layer_fixtures_gpu.py."""
import numpy as np
import pytest

HIDDEN_TYPES = [1, 2, 3]


def grid_inputs(rng, dims, points):
    grid = np.stack(np.meshgrid(*[np.linspace(-1, 1, points)] * dims, indexing="ij"), -1).reshape(-1, dims)
    return np.where(rng.uniform(0, 1, grid.shape) < 0.4, rng.uniform(-1, 1, grid.shape), grid)


@pytest.mark.parametrize("hidden_type", HIDDEN_TYPES)
def test_developmental_layer(ora, hidden_type):
    """:619-690"""
    from oracle.devmlp import DevMlp
    rng = np.random.RandomState(hidden_type)
    actor = ora.Mlp([1, 8, 1], hidden_act=hidden_type, out_act=0)
    actor.set_weights(rng.normal(0, 0.5, actor.np))
    actor2 = DevMlp([1, 8, 1], hidden_act=hidden_type, out_act=0, st_probabilistic=0, ac_probabilistic=0, seed=hidden_type)
    assert actor2.number_of_parameters(False) > actor.np
    assert actor2.number_of_parameters(True) - 2 == actor.np
    w2 = actor2.copy_weights_to(True)
    w2[1:-1] = actor.weights(); w2[0] = 1.0; w2[-1] = 1.0
    actor2.copy_weights_from(w2, True)
    x = grid_inputs(rng, 1, 400)
    ref = actor.forward(x)
    for forced in range(5):
        assert np.array_equal(actor2.forward(x), ref)
    w2[-1] = np.float32(0.6)
    actor2.copy_weights_from(w2, True)
    for forced in range(5):
        fit = (actor2.forward(x) == ref).mean()
        assert 0.52 <= fit <= 0.68, fit


@pytest.mark.parametrize("hidden_type", HIDDEN_TYPES)
def test_more_dimension(ora, hidden_type):
    """:692-802"""
    from oracle.devmlp import DevMlp
    rng = np.random.RandomState(10 + hidden_type)
    actor = ora.Mlp([3, 8, 4], hidden_act=hidden_type, out_act=0)
    actor.set_weights(rng.normal(0, 0.5, actor.np))
    actor2 = DevMlp([3, 8, 4], hidden_act=hidden_type, out_act=0, seed=hidden_type)
    assert actor2.number_of_parameters(True) - 3 - 4 == actor.np
    w2 = np.concatenate([np.ones(3), actor.weights(), np.ones(4)])
    actor2.copy_weights_from(w2, True)
    x = grid_inputs(rng, 3, 7)
    ref = actor.forward(x)
    for forced in range(5):
        assert np.array_equal(actor2.forward(x), ref)
    w2[-4:] = [0.1, 0.8, 0.4, 0.6]
    actor2.copy_weights_from(w2, True)
    for forced in range(5):
        fit = (actor2.forward(x) == ref).mean(axis=0)
        assert np.all(np.abs(fit - [0.1, 0.8, 0.4, 0.6]) <= 0.1), fit


@pytest.mark.parametrize("hidden_type", HIDDEN_TYPES)
def test_shared_parameters(ora, hidden_type):
    """:916-1046"""
    from oracle.devmlp import DevMlp
    actor = DevMlp([3, 8, 4], hidden_act=hidden_type, out_act=0, st_control=[1, 2], ac_control=[1, 2])
    critic = ora.Mlp([7, 10, 5, 1], hidden_act=hidden_type, out_act=0)
    critic2 = DevMlp([7, 10, 5, 1], hidden_act=hidden_type, out_act=0, actor=actor, nb_sensors=3)
    assert critic.np == critic2.number_of_parameters(True) - 4
    rng = np.random.RandomState(hidden_type)
    wa = actor.copy_weights_to(True)
    wa[:2] = rng.uniform(0.2, 0.9, 2); wa[-2:] = rng.uniform(0.2, 0.9, 2)
    actor.copy_weights_from(wa, True)
    wq = critic2.copy_weights_to(True)
    assert np.array_equal(wq[2:4], wa[:2]) and np.array_equal(wq[:2], wa[-2:])       # learnable_params()[1] / [0] of the critic
    wq[0] = 0.25
    critic2.copy_weights_from(wq, True)
    assert actor.copy_weights_to(True)[-2] == 0.25
    x = grid_inputs(rng, 3, 7)
    q = critic2.forward(np.concatenate([x, actor.forward(x)], 1))
    assert q.shape == (343, 1) and np.all(np.isfinite(q))


@pytest.mark.parametrize("hidden_type", HIDDEN_TYPES)
def test_heuristic1(ora, hidden_type):
    """:1257-1341"""
    from oracle.devmlp import DevMlp
    rng = np.random.RandomState(20 + hidden_type)
    kw = dict(hidden_act=hidden_type, out_act=0, heuristic=1, st_probabilistic=1, ac_probabilistic=1)
    actor = DevMlp([3, 8, 4], st_control=[1, 2], ac_control=[1, 2], devpoints=[0, 0, 30, 30], **kw)
    actor2 = DevMlp([3, 8, 4], st_control=None, ac_control=[1, 2], devpoints=[30, 30], **kw)
    actor.net.set_weights(rng.normal(0, 0.5, actor.net.np))
    actor2.copy_weights_from(actor.copy_weights_to(True), True)
    assert actor.number_of_parameters(True) == actor2.number_of_parameters(True)
    assert actor.number_of_parameters(False) > actor2.number_of_parameters(False)
    x = grid_inputs(rng, 3, 7)
    out, out2 = actor.forward(x), actor2.forward(x)
    m = (out != 0) & (x[:, 1] > 0.0001)[:, None] & (x[:, 2] > 0.0001)[:, None]
    assert m.any() and np.all(out[m] != out2[m])
    assert actor.inform(0) and not actor2.inform(0)
    assert np.array_equal(actor.forward(x), actor2.forward(x))
    assert not actor.inform(1) and actor.inform(30) and np.all(actor.ac.w == 1)
