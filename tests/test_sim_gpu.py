"""The batched environment front end (ddrl_sim_*; arch::Simulator, Simulator.hpp:76-248) on the GPU: G synthetic
environments with different episode lengths stepped by a thread pool must drive the agents exactly as the
reference's per-agent run_episode loop does (restated here in Python over the same agent front end), write the
reference's log lines, follow its learning / testing schedule and resume with --continue."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INI = """[simulation]
max_episode={me}
test_episode_per_episode={tpe}
test_episode_at_end={tae}
dump_log_each=1
display_log_each=2
save_agent_each=100
continue_save_each=2

[agent]
gamma=0.97
decision_each=1
noise=0.2
gaussian_policy=1
hidden_unit_q=24:12
hidden_unit_a=16
mini_batch_size=32
replay_memory=400
inverting_grad=true
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


class PointEnv:
    """a point pushed around in [-1, 1]^S by the first S action dimensions; reward -|x|; stops after `horizon` steps
    or when the point is near the origin (an absorbing goal state)"""

    def __init__(self, g, S, A):
        self.g, self.S, self.A = g, S, A
        self.horizon = 90 + 7 * g
        self.rng = np.random.RandomState(100 + g)
        self.log = []

    def reset_episode(self, learning):
        self.x = self.rng.uniform(-1, 1, self.S)
        self.t = 0
        self.log.append(("reset", bool(learning)))

    def perceptions(self):
        return self.x

    def performance(self):
        return -float(np.abs(self.x).sum())

    def apply(self, a):
        assert a.shape == (self.A,) and np.all(np.abs(a) <= 1)
        self.x = np.clip(self.x + 0.1 * a[: self.S], -1, 1)
        self.t += 1

    def final_state(self):
        return bool(np.abs(self.x).max() < 0.05)

    def running(self):
        return self.t < self.horizon and not self.final_state()

    def dump(self):
        return "%.6f %d" % (self.performance(), self.t)


def reference_loop(agents, envs, learning, G):
    """Simulator::run_episode (:139-208), one (agent, environment) pair after the other is not possible with a population
    front end, so the same calls are made for all pairs in lock step with masks -- by hand"""
    for e in envs:
        e.reset_episode(learning)
    agents.start_episode(np.stack([e.perceptions() for e in envs]), learning)
    alive = np.ones(G, bool)
    steps = np.zeros(G, int)
    while alive.any():
        run = np.array([alive[g] and envs[g].running() for g in range(G)])
        s = np.stack([e.perceptions() for e in envs])
        r = np.array([e.performance() for e in envs])
        goal = np.array([alive[g] and not run[g] and envs[g].final_state() for g in range(G)])
        last = alive & ~run
        act = agents.run(r, s, learning, goal, last, mask=alive.astype(np.uint8))
        for g in range(G):
            if run[g]:
                envs[g].apply(act[g].astype(np.float64))
                steps[g] += 1
        alive = run
    agents.end_episode(learning)
    return steps


@pytest.mark.parametrize("threads", [1, 4])
def test_simulator_drives_agents_like_the_reference_loop(tmp_path, threads, capfd):
    import ddrl_amd
    G, S, A = 5, 3, 4
    ini = os.path.join(str(tmp_path), "sim.ini")
    open(ini, "w").write(INI.format(me=4, tpe=1, tae=2))
    a = ddrl_amd.DdpgAgents(ini, G, S, A, seed=9)
    b = ddrl_amd.DdpgAgents(ini, G, S, A, seed=9)
    ea, eb = [PointEnv(g, S, A) for g in range(G)], [PointEnv(g, S, A) for g in range(G)]
    out = os.path.join(str(tmp_path), "out%d" % threads)
    sim = ddrl_amd.Simulator(a, ea, ini, out, n_threads=threads)
    sim.run()
    want_steps = []
    for episode in range(4):                                  # run_loop (:86-110): learn, then test_episode_per_episode tests
        want_steps.append(("L", episode, reference_loop(b, eb, True, G)))
        want_steps.append(("T", episode, reference_loop(b, eb, False, G)))
    for t in range(2):
        want_steps.append(("T", 4, reference_loop(b, eb, False, G)))
    for net in range(4):
        assert np.array_equal(a.pop.weights(net), b.pop.weights(net)), net
    sa, sb = a.stats(), b.stats()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert a.pop.adam(1)[2][0] > 50 and sa["episode"][0] == 4
    for g in range(G):
        assert ea[g].log == eb[g].log and np.array_equal(ea[g].x, eb[g].x)
        lines = open(os.path.join(out, str(g), "0.learning.data")).read().splitlines()
        assert len(lines) == 4
        for ep, line in enumerate(lines):                     # episode steps sum_weighted_reward trajectory.size() <env fields>
            f = line.split()
            assert int(f[0]) == ep and int(f[1]) == [w for w in want_steps if w[0] == "L"][ep][2][g] and len(f) == 6
            assert int(f[5]) == int(f[1])
        assert int(lines[-1].split()[3]) == a.pop.replay_len()[g]
        assert len(open(os.path.join(out, str(g), "0.0.testing.data")).read().splitlines()) == 4 + 1     # 4 during learning + the first at the end
        assert len(open(os.path.join(out, str(g), "0.1.testing.data")).read().splitlines()) == 1
    assert float(open(os.path.join(out, "time_elapsed")).read()) >= 0
    shown = [l for l in capfd.readouterr().out.splitlines() if l[:2] in ("L ", "T ")]
    assert len([l for l in shown if l.startswith("L ")]) == 2 * G          # display_log_each = 2: episodes 0 and 2
    sim.close(); a.close(); b.close()


def test_simulator_continue(tmp_path):
    """--continue (Simulator.hpp:76-84, 196-205): a run killed after episode 2 and started again ends where the
    uninterrupted run does"""
    import ddrl_amd
    G, S, A = 3, 3, 4
    ini = os.path.join(str(tmp_path), "sim.ini")
    open(ini, "w").write(INI.format(me=5, tpe=0, tae=0))
    mk = lambda seed: ddrl_amd.DdpgAgents(ini, G, S, A, seed=seed)
    full, fe = mk(3), [PointEnv(g, S, A) for g in range(G)]
    s0 = ddrl_amd.Simulator(full, fe, ini, os.path.join(str(tmp_path), "full"), n_threads=2, can_continue=True)
    s0.run()
    part, pe = mk(3), [PointEnv(g, S, A) for g in range(G)]
    out = os.path.join(str(tmp_path), "part")
    s1 = ddrl_amd.Simulator(part, pe, ini, out, n_threads=2, can_continue=True)
    for ep in range(3):                                       # episodes 0, 1, 2: the state is saved after episode 2
        s1.run_episode(True, ep)
    assert open(os.path.join(out, "continue.simu.data")).read().split() == ["3"]
    s1.close(); part.close()
    again = mk(77)                                            # a new process: other seeds, fresh nets
    s2 = ddrl_amd.Simulator(again, pe, ini, out, n_threads=2, can_continue=True)     # (the environments' own state is the caller's)
    s2.run()                                                  # episodes 3 and 4
    for net in range(4):
        assert np.array_equal(again.pop.weights(net), full.pop.weights(net)), net
    assert np.array_equal(again.stats()["episode"], full.stats()["episode"])
    assert len(open(os.path.join(out, "0", "0.learning.data")).read().splitlines()) == 5
    with pytest.raises(ddrl_amd.DdrlError, match="max_episode"):
        bad = os.path.join(str(tmp_path), "bad.ini")
        open(bad, "w").write(INI.replace("max_episode={me}\n", "").format(tpe=0, tae=0))
        ddrl_amd.Simulator(again, pe, bad, out)
    s0.close(); s2.close(); full.close(); again.close()


def test_simulator_with_cmaes_and_td3_agents(tmp_path):
    """the same loop over the other front ends: CMAESAg (one individual per environment, one generation per
    learning episode; its own dump line) and TD3Ag"""
    import ddrl_amd
    from test_cmaes_agents_gpu import write_ini as cmaes_ini
    S, A = 3, 4
    ini = cmaes_ini(tmp_path, lam=8)
    open(ini, "a").write("\n[simulation]\nmax_episode=3\ntest_episode_per_episode=0\ntest_episode_at_end=1\ndump_log_each=1\n"
                         "display_log_each=10\nsave_agent_each=100\n")
    ag = ddrl_amd.CmaesAgents(ini, S, A, seed=2)
    envs = [PointEnv(g, S, A) for g in range(ag.G)]
    out = os.path.join(str(tmp_path), "cm")
    sim = ddrl_amd.Simulator(ag, envs, ini, out, n_threads=3)
    sim.run()
    assert ag.state()["generations"] == 3
    for g in range(ag.G):
        lines = open(os.path.join(out, str(g), "0.learning.data")).read().splitlines()
        assert len(lines) == 3 and len(lines[0].split()) == 5          # episode steps sum_weighted_reward <env fields>
        assert len(open(os.path.join(out, str(g), "0.0.testing.data")).read().splitlines()) == 1
    sim.close(); ag.close()

    from test_td3_agents_gpu import write_ini as td3_ini
    ini = td3_ini(tmp_path, pn=0.1, pf=2)
    open(ini, "a").write("\n[simulation]\nmax_episode=4\ntest_episode_per_episode=0\ntest_episode_at_end=0\ndump_log_each=2\n"
                         "display_log_each=10\nsave_agent_each=100\n")
    G = 4
    a, b = ddrl_amd.Td3Agents(ini, G, S, A, seed=5), ddrl_amd.Td3Agents(ini, G, S, A, seed=5)
    ea, eb = [PointEnv(g, S, A) for g in range(G)], [PointEnv(g, S, A) for g in range(G)]
    out = os.path.join(str(tmp_path), "td3")
    sim = ddrl_amd.Simulator(a, ea, ini, out, n_threads=2)
    sim.run()
    for episode in range(4):
        reference_loop(b, eb, True, G)
    for net in range(6):
        assert np.array_equal(a.pop.weights(net), b.pop.weights(net)), net
    assert a.pop.adam(1)[2][0] > 100 and a.pop.adam(0)[2][0] < a.pop.adam(1)[2][0]
    assert len(open(os.path.join(out, "0", "0.learning.data")).read().splitlines()) == 2      # dump_log_each = 2: episodes 0 and 2
    sim.close(); a.close(); b.close()
