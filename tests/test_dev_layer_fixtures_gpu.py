"""The reference's own Developmental-layer unit tests, restated through the C ABI on the GPU:

  CaffeUtest.cpp:619-690    DevelopmentalLayer              heuristic 0: +2 parameters, state gate first / action gate last, w = 1 is the
                                                            plain net exactly, w = 0.6 passes 52-68 % of the rows
  CaffeUtest.cpp:919-1046   DevelopmentalLayerSharedParameters   critic view = [action gates][state gates][InnerProducts], +4 parameters, the
                                                            gate blobs are shared between actor, critic and their copies
  CaffeUtest.cpp:1257-1341  DevelopmentalLayerHeuristic1    devpoints 0,0,30,30: outputs differ from the state-ungated twin before
                                                            inform(0) and equal it exactly afterwards; parameter counts with and
                                                            without the lr_mult-0 blobs

The layer's forward, the counts and the order are pinned by these tests.  What is NOT in the tree is the fork's backward for the gate
weights under heuristic 0: the library serves the forward and the view and leaves the gate weights where the caller puts them
(parity unpinned for that gradient).  Which rows a Bernoulli gate passes depends on the fork's generator (absent): rates only."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BN0_HIDDEN_TYPES = [1, 2, 3]          # the reference loops hidden_layer (the hidden layer TYPE) over 1..3 ...
BATCH_NORMS = [0, 4, 6, 7, 8, 10, 11, 12, 14, 15]      # ... and batch_norm over these codes (CaffeUtest.cpp:620, 693, 805, 917).  With towers the
# library evaluates forward_batch as a TEST-phase copy would (running statistics, here a fresh net's), the reference's test its TRAIN-phase
# nets on the batch statistics: the properties held -- the gated twin equals the plain net at w = 1, an action passes at its rate and is
# exactly 0 otherwise, the parameter views and their order -- do not depend on which statistics normalise.  The views are
# number_of_parameters(true) / copyWeightsTo(w, true), as in the reference's tests: a tower's lr_mult-0 statistics are not in them.


def grid_inputs(rng, G, dims, points):
    """bib::Combinaison::continuous over [-1, 1]^dims with `points` values per axis, 40 % of the entries redrawn uniformly (:638-645)"""
    grid = np.stack(np.meshgrid(*[np.linspace(-1, 1, points)] * dims, indexing="ij"), -1).reshape(-1, dims)
    B = grid.shape[0]
    return np.where(rng.uniform(0, 1, (G, B, dims)) < 0.4, rng.uniform(-1, 1, (G, B, dims)), grid[None]).astype(np.float32)


@pytest.mark.parametrize("batch_norm", BATCH_NORMS)
@pytest.mark.parametrize("hidden_type", BN0_HIDDEN_TYPES)
def test_developmental_layer(hidden_type, batch_norm):
    """CaffeUtest.cpp:619-690: MLP(1, {8}, 1) against DODevMLP(1, {8}, 1) with empty control lists (= every dimension), both gates
    probabilistic without scaling, heuristic 0 (the default)."""
    import ddrl_amd
    G, B = 2, 400
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=1, nb_motors=1, mini_batch_size=16, replay_memory=500, hidden_unit_a=(8,),
                              hidden_unit_q=(8,), hidden_layer_type=hidden_type, actor_output_layer_type=0, batch_norm_actor=batch_norm)
    actor = ddrl_amd.DdpgPopulation(cfg, init_seed=3)
    actor2 = ddrl_amd.DdpgPopulation(cfg, init_seed=4)
    actor2.set_gate_mode(st_bernoulli=True, st_scale=False, ac_bernoulli=True, ac_scale=False, seed=hidden_type)
    actor2.set_gate_layout([], [], learnable=True)
    n = actor.dev_num_params(0, ignore_null_lr=True)                              # number_of_parameters(true): a plain MLP has no gate blobs
    assert (actor.num_params(0) == n) == (batch_norm == 0)                        # ... and leaves a tower's statistics out
    assert actor2.dev_num_params(0, ignore_null_lr=False) > actor.dev_num_params(0, ignore_null_lr=False)      # :635
    assert actor2.dev_num_params(0, ignore_null_lr=True) - 2 == n                 # :636
    rng = np.random.RandomState(hidden_type)
    x = grid_inputs(rng, G, 1, B)
    w = actor.dev_weights(0)
    w2 = actor2.dev_weights(0)
    w2[:, 1:-1] = w                                                               # :652-655
    w2[:, 0] = 1.0
    w2[:, -1] = 1.0
    actor2.set_dev_weights(0, w2)
    assert np.array_equal(actor2.dev_weights(0), w2) and np.array_equal(actor2.weights(0), actor.weights(0))      # (fresh statistics on both)
    ref = actor.forward_batch(0, x)
    for forced in range(5):                                                       # :659-667
        assert np.array_equal(actor2.forward_batch(0, x), ref)
    w2[:, -1] = np.float32(0.6)                                                   # :669
    actor2.set_dev_weights(0, w2)
    seen = []
    for forced in range(5):                                                       # :671-684
        out = actor2.forward_batch(0, x)
        seen.append(out)
        for g in range(G):
            fit = (out[g] == ref[g]).mean()
            assert 0.52 <= fit <= 0.68, (forced, g, fit)
            assert np.all((out[g] == ref[g]) | (out[g] == 0))                     # a dropped action is exactly 0 (no scaling)
    assert not np.array_equal(seen[0], seen[1])
    actor.close(); actor2.close()


@pytest.mark.parametrize("batch_norm", BATCH_NORMS)
@pytest.mark.parametrize("hidden_type", BN0_HIDDEN_TYPES)
def test_shared_parameters_and_critic_order(hidden_type, batch_norm):
    """CaffeUtest.cpp:919-1046: actor 3 -> 8 -> 4 and critic (3 + 4) -> 10 -> 5 -> 1 on st_control = 1:2, ac_control = 1:2.
    learnable_params()[0] of the actor and [1] of the critic are the SAME state-gate blob; the critic's [0] is the action-gate blob."""
    import ddrl_amd
    G, S, A = 2, 3, 4
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=16, replay_memory=500, hidden_unit_a=(8,),
                              hidden_unit_q=(10, 5), hidden_layer_type=hidden_type, actor_output_layer_type=0, batch_norm_actor=batch_norm,
                              batch_norm_critic=batch_norm)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=7)
    na, nq = pop.dev_num_params(0), pop.dev_num_params(1)                         # (no gate layout yet: the nets' own number_of_parameters(true))
    pop.set_gate_mode(st_bernoulli=True, st_scale=False, ac_bernoulli=True, ac_scale=False, seed=1)
    pop.set_gate_layout([1, 2], [1, 2], learnable=True)
    assert pop.dev_num_params(1) - 4 == nq                                        # :937
    assert pop.dev_num_params(3) == pop.dev_num_params(1)                         # :938 the copy (here: the target net) has the same view
    assert pop.dev_num_params(0) - 4 == na
    rng = np.random.RandomState(10 + hidden_type)
    wa = pop.dev_weights(0)
    wa[:, :2] = rng.uniform(0.2, 0.9, (G, 2))                                     # state gates of dims 1, 2
    wa[:, -2:] = rng.uniform(0.2, 0.9, (G, 2))                                    # action gates of dims 1, 2
    pop.set_dev_weights(0, wa)
    wq, wq_t = pop.dev_weights(1), pop.dev_weights(3)
    for i in range(2):                                                            # :988-993
        assert np.array_equal(wa[:, i], wq[:, i + 2])                             # actor [0] == critic [1]: state gates
        assert np.array_equal(wa[:, -2 + i], wq[:, i])                            # critic [0]: action gates
        assert np.array_equal(wq[:, i], wq_t[:, i]) and np.array_equal(wq[:, i + 2], wq_t[:, i + 2])
    if batch_norm == 0:
        assert np.array_equal(wq[:, 4:], pop.weights(1)) and np.array_equal(wa[:, 2:-2], pop.weights(0))
    assert wq.shape[1] == nq + 4 and wa.shape[1] == na + 4
    # the other way round: written through the critic's view, read through the actor's
    wq[:, 0] = 0.25; wq[:, 3] = 0.75
    pop.set_dev_weights(1, wq)
    wa2 = pop.dev_weights(0)
    assert np.all(wa2[:, -2] == 0.25) and np.all(wa2[:, 1] == 0.75)
    assert np.array_equal(wa2[:, 2:-2], wa[:, 2:-2])
    # the device holds the pass probabilities of the controlled dimensions, 1 elsewhere
    sg, ag = pop.gates()
    assert np.all(sg[:, 0] == 1) and np.all(ag[:, 0] == 1) and np.all(ag[:, 3] == 1)
    assert np.array_equal(sg[:, 1:].astype(np.float32), wa2[:, :2].astype(np.float32))
    assert np.array_equal(ag[:, 1:3].astype(np.float32), wa2[:, -2:].astype(np.float32))
    # :1013-1043 five forced passes of actor and critic run (343 rows), the blobs stay shared
    x = grid_inputs(rng, G, 3, 7)
    for forced in range(5):
        acts = pop.forward_batch(0, x)
        q = pop.forward_batch(1, np.concatenate([x, acts], -1))
        assert np.all(np.isfinite(acts)) and np.all(np.isfinite(q)) and q.shape == (G, 343, 1)
        assert np.array_equal(pop.dev_weights(0), wa2) and np.array_equal(pop.dev_weights(1)[:, :4], wq[:, :4])
    pop.close()


@pytest.mark.parametrize("batch_norm", BATCH_NORMS)
@pytest.mark.parametrize("hidden_type", BN0_HIDDEN_TYPES)
@pytest.mark.parametrize("case", ["more_dimension", "control_restriction"])
def test_more_dimension_and_control_restriction(hidden_type, case, batch_norm):
    """CaffeUtest.cpp:692-802 (every state and action dimension gated: +3 +4 parameters, action weights 0.1 / 0.8 / 0.4 / 0.6 pass at
    those rates +- 0.1) and :804-913 (st_control = 1:2, ac_control = 1:2: +2 +2, weights 0.4 / 0.6 on actions 1, 2 +- 0.11,
    actions 0 and 3 always equal); actor 3 -> 8 -> 4, 343 rows."""
    import ddrl_amd
    G, S, A = 2, 3, 4
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=S, nb_motors=A, mini_batch_size=16, replay_memory=500, hidden_unit_a=(8,),
                              hidden_unit_q=(8,), hidden_layer_type=hidden_type, actor_output_layer_type=0, batch_norm_actor=batch_norm)
    actor = ddrl_amd.DdpgPopulation(cfg, init_seed=3)
    actor2 = ddrl_amd.DdpgPopulation(cfg, init_seed=9)
    actor2.set_gate_mode(st_bernoulli=True, st_scale=False, ac_bernoulli=True, ac_scale=False, seed=17 + hidden_type)
    if case == "more_dimension":
        actor2.set_gate_layout([], [], learnable=True)
        ns, rates, tol = 3, {0: 0.1, 1: 0.8, 2: 0.4, 3: 0.6}, 0.1
    else:
        actor2.set_gate_layout([1, 2], [1, 2], learnable=True)
        ns, rates, tol = 2, {1: 0.4, 2: 0.6}, 0.11
    na = len(rates)
    n = actor.dev_num_params(0, True)
    assert actor2.dev_num_params(0, True) - ns - na == n                          # :708, :820
    w2 = actor2.dev_weights(0)
    w2[:, :ns] = 1.0
    w2[:, ns:ns + n] = actor.dev_weights(0)
    w2[:, ns + n:] = 1.0
    actor2.set_dev_weights(0, w2)
    rng = np.random.RandomState(3 * hidden_type + len(case))
    x = grid_inputs(rng, G, 3, 7)
    ref = actor.forward_batch(0, x)
    for forced in range(5):
        assert np.array_equal(actor2.forward_batch(0, x), ref)
    w2[:, ns + n:] = np.float32([rates[j] for j in sorted(rates)])
    actor2.set_dev_weights(0, w2)
    for forced in range(5):
        out = actor2.forward_batch(0, x)
        for g in range(G):
            for j in range(A):
                fit = (out[g, :, j] == ref[g, :, j]).mean()
                want = rates.get(j, 1.0)
                assert want - tol <= fit <= want + tol, (forced, g, j, fit)
    actor.close(); actor2.close()


AGENT = """[agent]
gamma=0.9
decision_each=1
noise=0.0
gaussian_policy=1
hidden_unit_q=10:5
hidden_unit_a=8
mini_batch_size=16
replay_memory=1000
inverting_grad=false
force_more_update=0
tau_soft_update=0.01
alpha_a=0.01
alpha_v=0.005
decay_v=-1
batch_norm_critic=0
batch_norm_actor=0
actor_output_layer_type=0
hidden_layer_type={hidden_type}
test_net=false
bn_adapt=false
momentum=0
nstep_return=0
[devnn]
heuristic={heuristic}
st_scale=false
st_probabilistic={prob}
ac_scale=false
ac_probabilistic={prob}
st_control={st}
ac_control=1:2
heuristic_devpoints={dp}
"""


def agents(tmp_path, name, hidden_type, st, dp, heuristic=1, prob=1, G=2):
    import ddrl_amd
    p = os.path.join(str(tmp_path), name)
    with open(p, "w") as f:
        f.write(AGENT.format(hidden_type=hidden_type, st=st, dp=dp, heuristic=heuristic, prob=prob))
    return ddrl_amd.DdpgAgents(p, G, 3, 4, seed=1)


@pytest.mark.parametrize("hidden_type", BN0_HIDDEN_TYPES)
def test_heuristic1_devpoints(tmp_path, hidden_type):
    """CaffeUtest.cpp:1257-1341 through the agent front end: `actor` gates state dims 1, 2 (opened by inform(0): devpoints 0, 0) and
    action dims 1, 2 (devpoints 30, 30); `actor2` has no state gate layer and the same action gates."""
    G, B = 2, 343
    ag1 = agents(tmp_path, "a.ini", hidden_type, "1:2", "0:0:30:30")
    ag2 = agents(tmp_path, "b.ini", hidden_type, "None", "30:30")
    a1, a2 = ag1.pop, ag2.pop
    assert a1.dev_num_params(0, True) == a2.dev_num_params(0, True) == a1.num_params(0)     # :1313 the blobs have lr_mult 0
    assert a1.dev_num_params(0, False) == a2.dev_num_params(0, False) + 2                   # :1314
    assert a2.dev_num_params(0, False) == a2.num_params(0) + 2
    a2.set_dev_weights(0, a1.dev_weights(0, True), True)                                    # :1308-1310
    rng = np.random.RandomState(hidden_type)
    x = grid_inputs(rng, G, 3, 7)
    out1, out2 = a1.forward_batch(0, x), a2.forward_batch(0, x)
    for g in range(G):                                                                     # :1320-1324
        m = (out1[g] != 0) & (x[g, :, 1] > 0.0001)[:, None] & (x[g, :, 2] > 0.0001)[:, None]
        assert m.any() and np.all(out1[g][m] != out2[g][m])
        assert np.all(out1[g][:, 1:3] == 0) and np.all(out2[g][:, 1:3] == 0)               # the action gates are closed in both
    # inform(0, 0.f): the first learning episode starts
    learn = np.ones(G, np.uint8)
    ag1.start_episode(rng.uniform(-1, 1, (G, 3)), learn)
    sg, agt = a1.gates()
    assert np.array_equal(sg, np.ones((G, 3))) and np.array_equal(agt, np.tile([1, 0, 0, 1], (G, 1)))
    out1 = a1.forward_batch(0, x)
    assert np.array_equal(out1, a2.forward_batch(0, x))                                     # :1334-1335, every entry
    # the view that counts the lr_mult-0 blobs shows them in the layer order: [state gates][IP][action gates]
    full = a1.dev_weights(0, False)
    assert np.all(full[:, :2] == 1) and np.all(full[:, -2:] == 0) and np.array_equal(full[:, 2:-2], a1.weights(0))
    ag1.close(); ag2.close()


def test_heuristic0_front_end(tmp_path):
    """devnn.heuristic = 0 in the agent front end (DODevMLP.hpp:84, 202-205): the gate weights are parameters of the view, nothing is
    scheduled, reset_learning_algo is refused; probabilistic = 0 gives Bernoulli gates, 1 the deterministic threshold at 0.5."""
    import ddrl_amd
    G = 2
    ag = agents(tmp_path, "h0.ini", 1, "1:2", "", heuristic=0, prob=1)
    pop = ag.pop
    n = pop.num_params(0)
    assert pop.dev_num_params(0, True) == n + 4 and pop.dev_num_params(1, True) == pop.num_params(1) + 4
    w = pop.dev_weights(0)
    assert np.all(w[:, :2] == 1) and np.all(w[:, -2:] == 1)                 # pass-through until somebody says otherwise
    rng = np.random.RandomState(0)
    x = grid_inputs(rng, G, 3, 7)
    ref = pop.forward_batch(0, x)
    w[:, 0] = 0.6; w[:, 1] = 0.4; w[:, -1] = 0.49
    pop.set_dev_weights(0, w)
    sg, agt = pop.gates()                                                   # probabilist = 1: a weight below 0.5 closes its dimension
    assert np.array_equal(sg, np.tile([1, 1, 0], (G, 1))) and np.array_equal(agt, np.tile([1, 1, 0, 1], (G, 1)))
    assert np.array_equal(pop.dev_weights(0), w)                            # the view keeps the weights themselves
    xm = x.copy(); xm[:, :, 2] = 0
    want = pop_forward_plain(pop, xm, ref)
    pop.set_dev_weights(0, w)                                               # (set_gates above replaced the view's weights by factors)
    out = pop.forward_batch(0, x)
    assert np.array_equal(out[:, :, [0, 1, 3]], want[:, :, [0, 1, 3]]) and np.all(out[:, :, 2] == 0)
    for e in range(3):                                                      # episodes change nothing
        learn = np.ones(G, np.uint8)
        ag.start_episode(rng.uniform(-1, 1, (G, 3)), learn)
        ag.run(np.zeros(G), rng.uniform(-1, 1, (G, 3)), learn, False, True)
        ag.end_episode(learn)
        assert np.array_equal(pop.dev_weights(0), w)
    ag.close()
    p = os.path.join(str(tmp_path), "h0r.ini")
    with open(p, "w") as f:
        f.write(AGENT.format(hidden_type=1, st="1:2", dp="", heuristic=0, prob=1) + "reset_learning_algo=true\n")
    with pytest.raises(ddrl_amd.DdrlError, match="learning alone"):
        ddrl_amd.DdpgAgents(p, G, 3, 4)
    bern = agents(tmp_path, "h0b.ini", 1, "", "", heuristic=0, prob=0)      # empty st_control: every state dimension
    assert bern.pop.dev_num_params(0) == bern.pop.num_params(0) + 3 + 2
    bern.close()
    p = os.path.join(str(tmp_path), "h0n.ini")                              # CaffeUtest.cpp:702: ac_control = None still gates every action
    with open(p, "w") as f:
        f.write(AGENT.format(hidden_type=1, st="", dp="", heuristic=0, prob=0).replace("ac_control=1:2", "ac_control=None"))
    none = ddrl_amd.DdpgAgents(p, G, 3, 4)
    assert none.pop.dev_num_params(0) == none.pop.num_params(0) + 3 + 4
    none.close()


def pop_forward_plain(pop, xm, ref):
    """the same net with the gates taken off, on an input that was masked by hand"""
    sg, agt = pop.gates()
    pop.set_gates(np.ones_like(sg), np.ones_like(agt))
    out = pop.forward_batch(0, xm)
    pop.set_gates(sg, agt)
    return out
