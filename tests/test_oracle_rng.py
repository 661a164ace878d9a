"""mt19937 is standard-exact; the index distribution is pinned to this image's libstdc++ by
oracle/pin_stdlib.cpp (compiled here against the real <random>)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_mt19937_10000th(ora):
    g = ora.Mt19937(5489)
    v = 0
    for _ in range(10000):
        v = g.next()
    assert v == 4123659995


def test_index_in_range_and_deterministic(ora):
    a, b = ora.Mt19937(1000), ora.Mt19937(1000)
    xs = [a.index(10000) for _ in range(1000)]
    assert xs == [b.index(10000) for _ in range(1000)]
    assert 0 <= min(xs) and max(xs) < 10000


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_pinned_against_libstdcxx():
    d = os.path.join(os.path.dirname(HERE), "oracle")
    subprocess.check_call(["make", "-s", "-C", d, "pin_stdlib"])
    out = subprocess.check_output([os.path.join(d, "pin_stdlib")]).decode()
    assert out.startswith("OK"), out
