"""a handle is thread-compatible, distinct handles may live on distinct threads.  Two handles on one GPU driven
from two host threads at once (ctypes releases the GIL during the calls) must give the bits of the same runs done one after
the other: the layer-by-layer path shares per-device auxiliary streams and events between handles (kernels_ddpg_wide.hip,
taken in turn), the one-tile path shares nothing."""
import threading

import numpy as np
import pytest

from ddpg_common import synth_transitions

pytestmark = pytest.mark.gpu


def build(ha, hq, G, seed):
    import ddrl_amd
    cfg = ddrl_amd.DdpgConfig(n_agents=G, nb_sensors=18, nb_motors=6, mini_batch_size=64, replay_memory=600, hidden_unit_a=ha, hidden_unit_q=hq)
    pop = ddrl_amd.DdpgPopulation(cfg, init_seed=seed)
    rows = synth_transitions(np.random.RandomState(seed), 500, 18, 6)
    for g in range(G):
        pop.replay_load(g, rows)
    return pop


def run(pop, G, n, out, key):
    try:
        for u in range(n):
            pop.update(np.random.RandomState(100 + u).randint(0, 500, (G, 64)).astype(np.int32))
        out[key] = [pop.weights(k) for k in range(4)]
    except Exception as e:          # surfaces in the main thread's assertions
        out[key] = e


@pytest.mark.parametrize("name,ha,hq,G", [("wide", (200, 150), (150, 200), 130), ("tile", (100, 100), (100, 100), 64)])
def test_two_handles_two_threads(name, ha, hq, G):
    ref, got = {}, {}
    for k in (0, 1):
        p = build(ha, hq, G, 7 + k)
        run(p, G, 5, ref, k)
        p.close()
    pops = [build(ha, hq, G, 7 + k) for k in (0, 1)]
    ths = [threading.Thread(target=run, args=(pops[k], G, 5, got, k)) for k in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths)
    for p in pops:
        p.close()
    for k in (0, 1):
        assert not isinstance(ref[k], Exception) and not isinstance(got[k], Exception), (ref[k], got[k])
        for a, b in zip(ref[k], got[k]):
            assert np.array_equal(a, b), (name, k)


def build_nfac(G, seed):
    import ddrl_amd
    cfg = ddrl_amd.NfacConfig(n_agents=G, nb_sensors=18, nb_motors=6, hidden_unit_a=(32, 32), hidden_unit_v=(32, 32), max_trajectory=256,
                              number_fitted_iteration=2, stoch_iter_actor=3, stoch_iter_critic=3)
    cfg.set("lambda", 0.9); cfg.set("gae", 1)
    return ddrl_amd.NfacPopulation(cfg, init_seed=seed)


def run_nfac(pop, G, n, out, key):
    from test_nfac_gpu import synth_traj
    try:
        for u in range(n):
            for g in range(G):
                rows, ends = synth_traj(np.random.RandomState(1000 * u + g), 100 + 7 * g, 18, 6, 3)
                pop.traj_load(g, rows, ends)
            pop.update()
        out[key] = [pop.weights(k) for k in range(2)]
    except Exception as e:
        out[key] = e


def test_two_nfac_handles_two_threads():
    G = 12
    ref, got = {}, {}
    for k in (0, 1):
        p = build_nfac(G, 3 + k)
        run_nfac(p, G, 3, ref, k)
        p.close()
    pops = [build_nfac(G, 3 + k) for k in (0, 1)]
    ths = [threading.Thread(target=run_nfac, args=(pops[k], G, 3, got, k)) for k in (0, 1)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths)
    for p in pops:
        p.close()
    for k in (0, 1):
        assert not isinstance(ref[k], Exception) and not isinstance(got[k], Exception), (ref[k], got[k])
        for a, b in zip(ref[k], got[k]):
            assert np.array_equal(a, b), k
