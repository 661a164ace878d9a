"""include/ddrl_agent.hpp: the G = 1 C++ adaptor with the reference's AAgent / ARLAgent method names and signatures
(AAgent.hpp:25-27, 39-123; ARLAgent.hpp:37-74, 109-137) over the C ABI.  Compiled with plain g++ (C++11, no boost)
and linked against the in-tree libddrl_pop.so; on a box without GPUs construction fails with the library's error,
on the GPU box a DDPG agent runs three episodes, learns, saves and resumes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INI = """[agent]
gamma=0.95
decision_each=1
noise=0.1
gaussian_policy=1
hidden_unit_q=50:7
hidden_unit_a=20
mini_batch_size=32
replay_memory=2000
inverting_grad=true
force_more_update=1
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


def _build(tmp_path):
    exe = str(tmp_path / "cpp_adaptor")
    lib = os.path.join(ROOT, "ddrl_amd")
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp_adaptor_main.cpp"),
           "-o", exe, "-L", lib, "-lddrl_pop", "-Wl,-rpath," + lib]
    subprocess.run(cmd, check=True, timeout=300)
    ini = tmp_path / "ddpg.ini"
    ini.write_text(INI)
    return exe, str(ini)


def test_adaptor_compiles_and_fails_cleanly_without_gpu(tmp_path):
    import torch
    exe, ini = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    p = subprocess.run([exe, ini, "nogpu"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "nogpu ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_adaptor_runs_a_ddpg_agent(tmp_path):
    exe, ini = _build(tmp_path)
    p = subprocess.run([exe, ini, "run"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "run ok" in p.stdout, (p.returncode, p.stdout, p.stderr)
