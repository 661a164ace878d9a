"""Python mirror of the agent-level front end (ddrl_agents_* and the reference's own DDPGAg_* /
OfflineCaclaAg_* names, include/ddrl_pop.h).  DDRLAg follows gym/agent.py:17-62 (NFAC) and :120-157
(DDPG) of the reference call for call."""
import ctypes as C

import numpy as np

from . import lib as _l
from .population import DdpgPopulation, NfacPopulation, PolicyPopulation, Population, Td3Population, _p


class _Borrowed(Population):
    """the population owned by a front end (not destroyed with this view)"""

    def __init__(self, L, handle, G, S, A):
        self.L, self.h, self.G, self.S, self.A = L, handle, G, S, A
        b = C.c_int()
        _l.check(L.ddrl_pop_shape(handle, None, None, None, None, C.byref(b)))
        self.B = b.value

    def close(self):
        self.h = None

    __del__ = close


class _Agents:
    def _adopt(self, h, n_agents, nb_sensors, nb_motors):
        self.h, self.G, self.S, self.A = h, n_agents, nb_sensors, nb_motors
        self.pop = self._view(self.L, C.c_void_p(self.L.ddrl_agents_population(h)), n_agents, nb_sensors, nb_motors)

    def close(self):
        if getattr(self, "h", None):
            self.pop.close()
            self.L.ddrl_agents_destroy(self.h)
            self.h = None

    __del__ = close

    def _flags(self, x):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.uint8), (self.G,)))

    def _mask(self, m):
        return None if m is None else np.ascontiguousarray(m, np.uint8).reshape(self.G)

    def start_episode(self, states, learning=True, mask=None):
        s = np.ascontiguousarray(states, np.float32).reshape(self.G, self.S)
        _l.check(self.L.ddrl_agents_start_episode(self.h, _p(s), _p(self._flags(learning)), _p(self._mask(mask))))

    def run(self, reward, states, learning=True, goal=False, last=False, mask=None):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(reward, np.float32), (self.G,)))
        s = np.ascontiguousarray(states, np.float32).reshape(self.G, self.S)
        out = np.empty((self.G, self.A), np.float32)
        _l.check(self.L.ddrl_agents_run(self.h, _p(r), _p(s), _p(self._flags(learning)), _p(self._flags(goal)),
                                        _p(self._flags(last)), _p(self._mask(mask)), _p(out)))
        return out

    def end_episode(self, learning=True, mask=None):
        _l.check(self.L.ddrl_agents_end_episode(self.h, _p(self._flags(learning)), _p(self._mask(mask))))

    def save_run(self, path):
        _l.check(self.L.ddrl_agents_save_run(self.h, str(path).encode()))

    def load_previous_run(self, path):
        _l.check(self.L.ddrl_agents_load_previous_run(self.h, str(path).encode()))

    def stats(self):
        swr, sr, ep = np.empty(self.G), np.empty(self.G), np.empty(self.G, np.int32)
        _l.check(self.L.ddrl_agents_stats(self.h, _p(swr), _p(sr), _p(ep)))
        return dict(sum_weighted_reward=swr, sum_reward=sr, episode=ep)


class _BorrowedDdpg(_Borrowed, DdpgPopulation):
    pass


class DdpgAgents(_Agents):
    """G DDPG agents driven step by step: ARLAgent::runf + DeepQNAg::_run for every agent per call."""
    _view = _BorrowedDdpg

    def __init__(self, ini_path, n_agents, nb_sensors, nb_motors, seed=0, device=-1):
        self.L = _l.load()
        h = C.c_void_p()
        _l.check(self.L.ddrl_agents_create_ddpg(ini_path.encode(), n_agents, nb_sensors, nb_motors, device, seed, C.byref(h)))
        self._adopt(h, n_agents, nb_sensors, nb_motors)


class _BorrowedTd3(_Borrowed, Td3Population):
    pass


class Td3Agents(_Agents):
    """G TD3 agents (TD3Ag.hpp): DeepQNAg's front end, one update per stored transition, the actor and
    the targets every policy_freq-th step of an episode."""
    _view = _BorrowedTd3

    def __init__(self, ini_path, n_agents, nb_sensors, nb_motors, seed=0, device=-1):
        self.L = _l.load()
        h = C.c_void_p()
        _l.check(self.L.ddrl_agents_create_td3(ini_path.encode(), n_agents, nb_sensors, nb_motors, device, seed, C.byref(h)))
        self._adopt(h, n_agents, nb_sensors, nb_motors)


class _BorrowedNfac(_Borrowed, NfacPopulation):
    pass


class NfacAgents(_Agents):
    """G NFAC(lambda)-V agents: ARLAgent::runf + OfflineCaclaAg::_run / end_episode per call."""
    _view = _BorrowedNfac

    def __init__(self, ini_path, n_agents, nb_sensors, nb_motors, max_trajectory=0, seed=0, device=-1):
        self.L = _l.load()
        h = C.c_void_p()
        _l.check(self.L.ddrl_agents_create_nfac(ini_path.encode(), n_agents, nb_sensors, nb_motors, max_trajectory, device,
                                                seed, C.byref(h)))
        self._adopt(h, n_agents, nb_sensors, nb_motors)
        self.pop.maxT = self.L.ddrl_nfac_max_trajectory(self.pop.h)
        if self.pop.maxT < 0:
            _l.check(self.pop.maxT)


class _BorrowedPolicy(_Borrowed, PolicyPopulation):
    pass


class CmaesAgents(_Agents):
    """CMAESAg with one agent slot per individual of the generation (ddrl_agents_create_cmaes):
    start_episode = start_instance, end_episode = end_episode + end_instance."""
    _view = _BorrowedPolicy

    def __init__(self, ini_path, nb_sensors, nb_motors, seed=0, device=-1):
        from .cmaes import Cmaes
        self.L = _l.load()
        h = C.c_void_p()
        _l.check(self.L.ddrl_agents_create_cmaes(ini_path.encode(), nb_sensors, nb_motors, device, seed, C.byref(h)))
        lam = C.c_int()
        _l.check(self.L.ddrl_agents_cmaes_state(h, C.byref(lam), None, None, None))
        self._adopt(h, lam.value, nb_sensors, nb_motors)
        self.es = Cmaes.borrow(self.L.ddrl_agents_cmaes(h), self.pop.num_params(0))

    def state(self):
        v = [C.c_int() for _ in range(4)]
        _l.check(self.L.ddrl_agents_cmaes_state(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("lambda", "n_evaluated", "generations", "error_count"), [x.value for x in v]))

    def close(self):
        if getattr(self, "es", None):
            self.es.close()
        _Agents.close(self)

    __del__ = close


class DDRLAg:
    """gym/agent.py's DDRLAg against libddrl_pop.so: same methods, same DDPGAg_* (algo="ddpg") or
    OfflineCaclaAg_* (algo="nfac") calls; algo="td3" = the DDPGAg_* calls of the reference's libddrl-td3,
    here under the TD3Ag_ prefix."""

    def __init__(self, nb_motors, nb_sensors, argv=(), algo="ddpg"):
        self.L = _l.load()
        self.nb_motors, self.nb_sensors = nb_motors, nb_sensors
        self._pre = {"ddpg": "DDPGAg_", "nfac": "OfflineCaclaAg_", "td3": "TD3Ag_"}[algo]
        self._name = {"ddpg": "DDPG", "nfac": "NFAC(lambda)-V", "td3": "TD3"}[algo]
        self.obj = self._f("new")(nb_motors, nb_sensors)
        argv = list(argv) + [""]
        arr = (C.c_char_p * len(argv))(*[a.encode("utf-8") for a in argv])
        self._f("unique_invoke")(self.obj, len(argv), arr)

    def _f(self, name):
        return getattr(self.L, self._pre + name)

    def run(self, reward, state, learning, goal, last):
        s = np.ascontiguousarray(state, np.float64)
        p = self._f("run")(self.obj, reward, s.ctypes.data, learning, goal, last)
        if not p:
            raise _l.DdrlError(-1, self.L.ddrl_last_error().decode(errors="replace"))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.nb_motors,)).copy()

    def start_ep(self, state, learning):
        s = np.ascontiguousarray(state, np.float64)
        self._f("start_episode")(self.obj, s.ctypes.data, learning)

    def end_ep(self, learning):
        self._f("end_episode")(self.obj, learning)

    def dump(self, learning, episode, step, treward):
        self._f("dump")(self.obj, learning, episode, step, treward)

    def display(self, learning, episode, step, treward):
        self._f("display")(self.obj, learning, episode, step, treward)

    def save(self, episode):
        self._f("save")(self.obj, episode)

    def load(self, episode):
        self._f("load")(self.obj, episode)

    def name(self):
        return self._name

    def close(self):
        if getattr(self, "obj", None):
            self._f("delete")(self.obj)
            self.obj = None

    __del__ = close
