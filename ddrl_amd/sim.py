"""Python mirror of the batched environment front end (ddrl_sim_*, include/ddrl_pop.h): the population
analogue of arch::Simulator (common/include/arch/Simulator.hpp:76-248).  Environments are objects with
the reference's Environment methods; they are handed to the library as C callbacks."""
import ctypes as C

import numpy as np

from . import lib as _l


class Simulator:
    """envs: one object per agent with reset_episode(learning), perceptions() -> [S], performance() -> float,
    apply(actuators [A]), running() -> bool, final_state() -> bool and optionally next_instance(learning),
    dump() -> str."""

    def __init__(self, agents, envs, ini_path, out_dir=".", n_threads=1, can_continue=False):
        self.L = _l.load()
        self.agents, self.envs = agents, list(envs)
        if len(self.envs) != agents.G:
            raise ValueError("one environment per agent")
        S, A, E = agents.S, agents.A, self.envs
        V = _l.EnvVtable

        def perceptions(_, g, out):
            p = np.asarray(E[g].perceptions(), np.float64)
            for i in range(S):
                out[i] = p[i]

        def dump(_, g, buf, n):
            txt = E[g].dump().encode()[: n - 1] + b"\0"
            C.memmove(buf, txt, len(txt))

        has = lambda name: all(hasattr(e, name) for e in E)
        self._cb = dict(
            reset_episode=V.RESET(lambda _, g, l: E[g].reset_episode(bool(l))),
            perceptions=V.PERCEPTIONS(perceptions),
            performance=V.PERFORMANCE(lambda _, g: float(E[g].performance())),
            apply=V.APPLY(lambda _, g, a: E[g].apply(np.ctypeslib.as_array(a, shape=(A,)).copy())),
            running=V.FLAG(lambda _, g: int(bool(E[g].running()))),
            final_state=V.FLAG(lambda _, g: int(bool(E[g].final_state()))),
            next_instance=V.NEXT(lambda _, g, l: E[g].next_instance(bool(l))) if has("next_instance") else V.NEXT(),
            dump=V.DUMP(dump) if has("dump") else V.DUMP())
        self._vt = V(None, *[self._cb[k] for k in ("reset_episode", "perceptions", "performance", "apply", "running",
                                                      "final_state", "next_instance", "dump")])
        h = C.c_void_p()
        _l.check(self.L.ddrl_sim_create(agents.h, C.byref(self._vt), str(ini_path).encode(), str(out_dir).encode(),
                                        int(n_threads), int(can_continue), C.byref(h)))
        self.h = h

    def run(self):
        _l.check(self.L.ddrl_sim_run(self.h))

    def run_episode(self, learning, lepisode, tepisode=0):
        _l.check(self.L.ddrl_sim_run_episode(self.h, int(learning), lepisode, tepisode))

    def close(self):
        if getattr(self, "h", None):
            self.L.ddrl_sim_destroy(self.h)
            self.h = None

    __del__ = close
