"""The entry point that allocates nothing (ddrl_cmaes_rank_noalloc, include/ddrl_pop.h): through the C ABI,
against the existing ddrl_cmaes_rank, bit for bit, on two calls in a row, the second with a smaller input.
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rank_ref(es, f):
    from ddrl_amd import lib as _l
    out = np.empty(len(f), np.int32)
    _l.check(_l.load().ddrl_cmaes_rank(es._h, f.ctypes.data_as(C.c_void_p), len(f), out.ctypes.data_as(C.c_void_p)))
    return out


def _rank_noalloc(f):
    from ddrl_amd import lib as _l
    out = np.full(len(f), -1, np.int32)
    _l.check(_l.load().ddrl_cmaes_rank_noalloc(f.ctypes.data_as(C.c_void_p), len(f), out.ctypes.data_as(C.c_void_p), None))
    return out


def test_noalloc_rank_matches_rank_twice_second_smaller():
    import ddrl_amd
    from ddrl_amd import lib as _l
    lam = 4096
    es = ddrl_amd.Cmaes(np.zeros(8), np.full(8, 0.5), seed=3, lam=lam)
    rng = np.random.RandomState(7)
    # first call: the full 4096 values, with ties (fp32-rounded fitness as CMAESAg.hpp:239 produces) and signed zeros
    f1 = rng.normal(size=lam).astype(np.float32).astype(np.float64)
    f1[rng.randint(0, lam, 600)] = f1[rng.randint(0, lam, 600)]
    f1[:4] = [0.0, -0.0, 0.0, -0.0]
    a = _rank_noalloc(f1)
    assert np.array_equal(a, _rank_ref(es, f1))
    assert sorted(a.tolist()) == list(range(lam))
    # second call right after, smaller input: nothing of the first call's scratch contents may leak into it
    f2 = np.round(rng.normal(size=1000), 1)
    b = _rank_noalloc(f2)
    assert np.array_equal(b, _rank_ref(es, f2))
    assert sorted(b.tolist()) == list(range(1000))
    # a third, tiny one, and the limits
    f3 = np.array([3.0, 1.0, 2.0, 1.0, 3.0, 1.0])
    assert _rank_noalloc(f3).tolist() == [5, 3, 1, 2, 4, 0]          # SURVEY 8(c'): equal values => later index first
    L = _l.load()
    big = np.zeros(4097)
    out = np.zeros(4097, np.int32)
    assert L.ddrl_cmaes_rank_noalloc(big.ctypes.data_as(C.c_void_p), 4097, out.ctypes.data_as(C.c_void_p), None) == -1
    assert b"4096" in L.ddrl_last_error()
    assert L.ddrl_cmaes_rank_noalloc(big.ctypes.data_as(C.c_void_p), 0, out.ctypes.data_as(C.c_void_p), None) == -1
