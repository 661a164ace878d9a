"""Dev tool (GPU box): the PCIe-inclusive rate of the env-step path -- ddrl_agents_run for 1024 DDPG agents (host
states in, host actions out; per step: H2D states, batch-1 inference, D2H actions, replay append, one update round)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import ddrl_amd
INI = """[agent]
gamma=0.99
decision_each=1
noise=0.05
gaussian_policy=1
hidden_unit_q=100:100
hidden_unit_a=100:100
mini_batch_size=64
replay_memory=10000
inverting_grad=true
force_more_update={fmu}
tau_soft_update=0.001
alpha_a=0.0001
alpha_v=0.001
decay_v=-1
batch_norm_critic=0
batch_norm_actor=0
actor_output_layer_type=2
hidden_layer_type=1
test_net=false
bn_adapt=false
momentum=0
nstep_return=0
[devnn]
ewc=100
ewc_decay=-1
ewc_fisher_beta=0.999
ewc_best_param_method=0
ewc_for_critic=1
"""
G, S, A = 1024, 18, 6
for fmu in (0, 7):
    open("/tmp/step.ini", "w").write(INI.format(fmu=fmu))
    ag = ddrl_amd.DdpgAgents("/tmp/step.ini", G, S, A, seed=1)
    rng = np.random.RandomState(0)
    X = rng.uniform(-1, 1, (64, G, S)).astype(np.float32); Rw = rng.normal(0, 1, (64, G)).astype(np.float32)
    ag.start_episode(X[0], True)
    for t in range(260):
        ag.run(Rw[t % 64], X[t % 64], True)
    ag.pop.sync()
    n = 200
    t0 = time.perf_counter()
    for t in range(n):
        ag.run(Rw[t % 64], X[t % 64], True)
    ag.pop.sync()
    dt = time.perf_counter() - t0
    print("force_more_update=%d: %.3f ms per population step -> %.0f agent-steps/s, %.2f M updates/s (host buffers in and out every step)"
          % (fmu, dt / n * 1e3, G * n / dt, G * n * (1 + fmu) / dt / 1e6))
    ag.close()
