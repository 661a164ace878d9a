"""Python mirror of the ddrl_cmaes_* C ABI (include/ddrl_pop.h): the c-cmaes call sequence CMAESAg
drives (/root/reference/agent/cmaes/include/CMAESAg.hpp:97-98, 135-171, 239-247) on one MI355X."""
import ctypes as C

import numpy as np

from . import lib as _l


class Cmaes:
    """cmaes_init / SamplePopulation / ReSampleSingle / UpdateDistribution / Get."""

    def __init__(self, xstart, stds, seed, lam=0, device=-1):
        xstart = np.ascontiguousarray(xstart, np.float64)
        stds = np.ascontiguousarray(stds, np.float64)
        if xstart.shape != stds.shape or xstart.ndim != 1:
            raise ValueError("xstart and stds must be vectors of the same length")
        self._L = _l.load()
        self.N = int(xstart.size)
        h = C.c_void_p()
        _l.check(self._L.ddrl_cmaes_create(self.N, xstart.ctypes.data, stds.ctypes.data, int(seed), int(lam),
                                           int(device), C.byref(h)))
        self._h = h
        self.lam = int(self.get("lambda"))
        self.mu = int(self.get("mu"))

    @classmethod
    def borrow(cls, handle, N):
        """view of a handle owned elsewhere (ddrl_agents_cmaes); close() does not destroy it"""
        self = cls.__new__(cls)
        self._L, self._h, self.N, self._borrowed = _l.load(), C.c_void_p(handle), int(N), True
        self.lam, self.mu = int(self.get("lambda")), int(self.get("mu"))
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._L.ddrl_cmaes_destroy(self._h)
            self._h = None

    __del__ = close

    def sample(self):
        _l.check(self._L.ddrl_cmaes_sample(self._h))
        return self.population()

    def resample(self, k):
        _l.check(self._L.ddrl_cmaes_resample(self._h, int(k)))

    def make_feasible(self, bound=1.0):
        n = C.c_int(0)
        _l.check(self._L.ddrl_cmaes_make_feasible(self._h, float(bound), C.byref(n)))
        return n.value

    def population(self):
        X = np.empty((self.lam, self.N), np.float64)
        _l.check(self._L.ddrl_cmaes_get_population(self._h, X.ctypes.data))
        return X

    def population_dev(self):
        p, ld = C.c_void_p(), C.c_int()
        _l.check(self._L.ddrl_cmaes_population_dev(self._h, C.byref(p), C.byref(ld)))
        return p.value, ld.value

    def sample_dev(self):
        """cmaes_SamplePopulation without the host copy: returns (device pointer of X [lambda x ld] fp32, ld)"""
        _l.check(self._L.ddrl_cmaes_sample(self._h))
        return self.population_dev()

    def population_torch(self):
        """the sampled population as a torch view [lambda x N] (fp32) of the library's device buffer: no copy"""
        import torch
        ptr, ld = self.population_dev()

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"data": (ptr, False), "shape": (self.lam, self.N), "typestr": "<f4",
                                      "strides": (4 * ld, 4), "version": 2}
        return torch.as_tensor(v, device="cuda")

    def update(self, fitness):
        f = np.ascontiguousarray(fitness, np.float64)
        if f.shape != (self.lam,):
            raise ValueError("fitness must have lambda == %d entries" % self.lam)
        _l.check(self._L.ddrl_cmaes_update(self._h, f.ctypes.data))

    def update_dev(self, d_fitness_ptr):
        _l.check(self._L.ddrl_cmaes_update_dev(self._h, d_fitness_ptr))

    def rank(self, fitness):
        f = np.ascontiguousarray(fitness, np.float64)
        idx = np.empty(f.size, np.int32)
        _l.check(self._L.ddrl_cmaes_rank(self._h, f.ctypes.data, f.size, idx.ctypes.data))
        return idx

    def get(self, key):
        v = C.c_double()
        _l.check(self._L.ddrl_cmaes_get(self._h, key.encode(), C.byref(v)))
        return v.value

    def vec(self, key):
        n = {"weights": self.mu, "C": self.N * self.N, "B": self.N * self.N}.get(key, self.N)
        out = np.empty(n, np.float64)
        _l.check(self._L.ddrl_cmaes_get_vec(self._h, key.encode(), out.ctypes.data))
        return out.reshape(self.N, self.N) if key in ("C", "B") else out

    # cmaes_WriteToFile(evo, "resume", ...) / cmaes_resume_distribution (CMAESAg.hpp:279-308), bit-identical continuation
    def save_run(self, path):
        _l.check(self._L.ddrl_cmaes_save_run(self._h, str(path).encode()))

    def load_previous_run(self, path):
        _l.check(self._L.ddrl_cmaes_load_previous_run(self._h, str(path).encode()))

    def set_option(self, key, value):
        _l.check(self._L.ddrl_cmaes_set_option(self._h, key.encode(), int(value)))

    def test_for_termination(self):
        """'|'-separated names of the stopping criteria met ('' = keep going)"""
        buf = C.create_string_buffer(256)
        n = self._L.ddrl_cmaes_test_for_termination(self._h, buf, 256)
        if n < 0:
            _l.check(n)
        return buf.value.decode()

    def index(self):
        idx = np.empty(self.lam, np.int32)
        _l.check(self._L.ddrl_cmaes_get_index(self._h, idx.ctypes.data))
        return idx
