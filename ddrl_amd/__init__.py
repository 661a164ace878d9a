"""ddrl_amd -- MI355X-native (gfx950) implementation of ddrl's learning hot path.

The product is ddrl_amd/libddrl_pop.so (hand-written HIP kernels behind the C ABI declared in
include/ddrl_pop.h).  This package is the thin Python mirror of that ABI used by tests and bench.py;
it has no CPU path and raises if the library is missing.
"""
from .lib import load, LibraryMissing, DdrlError  # noqa: F401
from .population import DdpgConfig, DdpgPopulation, Td3Population, NfacConfig, NfacPopulation, PolicyPopulation  # noqa: F401
from .cmaes import Cmaes  # noqa: F401
from .sim import Simulator  # noqa: F401
from .agents import DdpgAgents, Td3Agents, NfacAgents, CmaesAgents, DDRLAg  # noqa: F401
