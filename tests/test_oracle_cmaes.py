"""Pins oracle/cmaes_ref.c bit-for-bit against golden vectors produced by the reference's own
c-cmaes (agent/cmaes/src/cmaes.c), see oracle/gen_cmaes_golden.c + oracle/Makefile `golden`,
and against SURVEY.md 8(c') (known-answer vector captured by the survey from the same source)."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(G, "cmaes_%s.json" % name)) as f:
        return json.load(f)


FIT = {
    "sphere": lambda x: float(np.sum(x * x)),
    "float_steps": None,
    "rosen": None,
    "elli": None,
}


def _fitness(name, X, golden_fit):
    # fitness values are taken from the fixture (the generator's C arithmetic), so the test pins the
    # CMA-ES state machine and not a Python re-derivation of pow()/floor()
    return np.asarray(golden_fit, dtype=np.float64)


def test_kat(ora):
    k = load("kat")
    c = ora.Cmaes([.1, .2, .3, .4, .5], [.5] * 5, 1, 8)
    got = [c.gauss() for _ in range(32)]
    assert got == k["gauss_seed1_head"]
    assert list(ora.sorted_index(k["tie_input"])) == k["tie_index"] == [5, 3, 1, 2, 4, 0]


def test_survey_known_answer(ora):
    """SURVEY.md 8(c'): lambda=8 mu=4, N=5, x0=.1...5, sigma0=.5, seed=1, sphere."""
    c = ora.Cmaes([.1, .2, .3, .4, .5], [.5] * 5, 1, 8)
    assert c.mu == 4
    assert c.get("mueff") == 2.8406104297170534
    assert c.get("cs") == 0.44652563258316402
    assert c.get("damps") == 1.4458311881388162
    assert c.get("ccumcov") == 0.44444444444444442
    assert c.get("ccov") == 0.075624512429424723
    assert c.get("chiN") == 2.1285237557247996
    w = c.vec("weights", 4)
    assert list(w) == [0.49373837748434096, 0.28109683248064332, 0.15670950255807004, 0.068455287476945695]
    X = c.sample()
    assert list(X[0]) == [-0.31842690129640305, 0.11386003796338778, 0.39355872412703724, 1.2077199245032197, 0.41161298051140544]
    assert list(X[7]) == [0.08359317542843113, 0.78255833040609701, -0.2239999809440128, 0.80890733714536978, -0.20364738655640857]
    c.update((X * X).sum(1))
    assert list(c.index()) == [4, 5, 3, 7, 6, 0, 2, 1]
    assert c.get("sigma") == 0.4906387142471581
    assert c.vec("xmean")[0] == 0.14887943324175787
    C = c.vec("C", 25).reshape(5, 5)
    assert C[0, 0] == 0.93034624341520489 and C[1, 0] == 0.0057828650290830058
    assert c.vec("ps")[0] == 0.13722626374428595 and c.vec("pc")[0] == 0.13699777064836419


@pytest.mark.parametrize("name", ["n5", "n12_ties", "n40", "n151", "n151_big", "n631_big", "term_flat", "term_noeffect", "term_sphere"])
def test_golden_runs(ora, name):
    g = load(name)
    N, lam = g["N"], g["lambda"]
    x0 = g["x0scale"] * (np.arange(N) + 1) / float(N)
    c = ora.Cmaes(x0, [g["sigma0"]] * N, g["seed"], lam)
    assert c.mu == g["mu"]
    for key in ("mueff", "cs", "damps", "ccumcov", "ccov", "mucov", "chiN", "modulo"):
        assert c.get(key) == g[key], key
    for gi, gen in enumerate(g["gens"]):
        X = c.sample()
        assert c.get_int("eigen_uptodate") == gen["eigen_uptodate_at_sample"], gi
        assert list(X[0]) == gen["x_first"], gi
        assert list(X[-1]) == gen["x_last"], gi
        if "X" in gen:
            assert list(X.ravel()) == gen["X"]
        assert list(c.vec("D")) == gen["D_at_sample"], gi
        c.update(np.asarray(gen["fitness"]))
        assert list(c.index()) == gen["index"], gi
        assert c.get("sigma") == gen["sigma"], gi
        assert list(c.vec("xmean")) == gen["xmean"], gi
        assert list(c.vec("ps")) == gen["ps"], gi
        assert list(c.vec("pc")) == gen["pc"], gi
        Cm = c.vec("C", N * N).reshape(N, N)
        assert list(np.diag(Cm)) == gen["diagC"], gi
        if "C_lower" in gen:
            assert list(Cm[np.tril_indices(N)]) == gen["C_lower"], gi
        if "term" in gen:     # cmaes_TestForTermination's criteria (cmaes.c:1536-1667), by name
            assert c.test_for_termination() == gen["term"], gi


def test_ties_present_in_fixture():
    g = load("n12_ties")
    assert any(len(set(gen["fitness"])) < len(gen["fitness"]) for gen in g["gens"])
