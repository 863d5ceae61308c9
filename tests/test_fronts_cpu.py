"""The symbolic phase of the FRONTAL launch shape (csrc/fronts.cpp) on the CPU: the plan blob is executed in numpy
(tests/front_ref.py, operation for operation what front_kernel.hip.hpp does) and its step must equal the dense solve of
(JtJ + lambda I) d = -Jt r built from the oracle's Jacobian rows (reference: ezpz/src/solver/newton.rs:73-102).  No GPU needed:
this is host logic.  The GPU tests of the kernel itself are tests/test_gpu_fronts.py."""
import numpy as np
import pytest

import gen
import front_ref as FR
from oracle import oracle as O


def check(recs, g, wgs, lam=1e-3, lds_bytes=160 * 1024, rel=1e-9):
    n = len(g)
    p = FR.Plan(recs, n, wgs=wgs, lds_bytes=lds_bytes)
    assert p.ok
    d, bad = FR.linear_step(p, g, lam)
    want = FR.dense_step(recs, n, g, lam)
    assert not bad and not np.any(np.isnan(d))
    assert np.max(np.abs(d - want)) <= rel * max(1.0, np.max(np.abs(want))), float(np.max(np.abs(d - want)))
    return p


@pytest.mark.parametrize("npts,wgs", [(3, 1), (8, 1), (25, 1), (75, 1), (75, 2), (150, 1), (150, 3), (150, 8), (400, 0)])
def test_connected_sketch_plan_equals_the_dense_solve(npts, wgs):
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    p = check(recs, g + 0.01, wgs)
    assert p.max_rows <= 63 and p.max_pivots <= 16
    if wgs > 1:
        assert 1 < p.n_wgs <= wgs and p.n_chunks > 0
    if wgs == 0:
        assert p.n_wgs > 1  # 800 variables: several workgroups by the planner's own choice


@pytest.mark.parametrize("family", ["tree", "band", "hub", "comb"])
@pytest.mark.parametrize("wgs", [1, 4])
def test_graph_families_plan_equals_the_dense_solve(family, wgs):
    rng = np.random.default_rng(11)
    recs, true = gen.graph_sketch(family, 60, rng)
    check(recs, true + rng.uniform(-0.02, 0.02, len(true)), wgs)


def test_all_kinds_weights_duplicate_columns_and_free_variables():
    """Random block of all 25 kinds (duplicate ids inside a row, two-row kinds whose rows touch different variables), weights,
    and variables no constraint touches: the plan keeps one front per loose variable and the step is the dense solve's."""
    rng = np.random.default_rng(5)
    n = 40
    cons = []
    for kind in range(25):
        c = gen.arb_constraint(rng, kind, hi=32)
        c["weight"] = float(rng.uniform(0.5, 2.0))
        cons.append(c)
    recs = O.stack(cons)
    x = rng.uniform(-5.0, 5.0, n)
    p = check(recs, x, 1, lam=0.5, rel=1e-8)
    assert p.n_fronts >= 8  # variables 32..39 are touched by nothing: one front each
    check(recs, x, 3, lam=0.5, rel=1e-8)


def test_a_front_too_large_for_a_wavefront_is_refused():
    """A clique of 40 points (every pair at a distance): one front of 80 rows -- more than a wavefront's 63: the shape does not
    apply and the other shapes serve."""
    pts = 40
    cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
    for i in range(pts):
        for j in range(i):
            cons.append(O.distance((2 * i, 2 * i + 1), (2 * j, 2 * j + 1), 1.0))
    p = FR.Plan(O.stack(cons), 2 * pts)
    assert not p.ok


def test_workgroup_shares_fit_the_lds_they_are_given():
    recs, g = gen.connected_sketch(400, 1400)
    p = FR.Plan(recs, len(g), wgs=0, lds_bytes=64 * 1024)
    assert p.ok and p.lds_bytes <= 64 * 1024 and p.n_wgs >= 4
    assert not FR.Plan(recs, len(g), wgs=1, lds_bytes=64 * 1024).ok  # 800 variables do not fit 64 KB on one workgroup
