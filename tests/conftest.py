import os
import sys

import pytest

# torch ships its own copy of the HIP runtime; a process that uses both torch.cuda and libddrl_pop.so (bench.py, the
# device-resident CMA-ES loop of ddrl_amd/shard.py) must let torch load its copy FIRST: initialising torch.cuda after
# the system runtime has taken the GPU fails with "No HIP GPUs are available".  The test session fixes the order here.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (checker only)."""
    from oracle import oracle as o
    o.lib()
    return o
