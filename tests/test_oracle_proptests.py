"""Pins the CPU oracle to the reference's solve-level property tests (tests/proptest_cases.py: transcriptions of
/root/reference/ezpz/src/tests/proptests.rs:294-707, the fixed cases :1253-1282 and the shrunk failures kept in
ezpz/proptest-regressions/tests/proptests.txt).  proptest runs 256 cases per property; so does this, from a fixed seed."""
import math

import numpy as np
import pytest

import proptest_cases as P
from adapters import OracleAdapter
from oracle import oracle as O

CASES = 256


@pytest.mark.parametrize("linsolve", [O.LINSOLVE_DENSE, O.LINSOLVE_SPARSE], ids=["dense", "sparse"])
@pytest.mark.parametrize("prop", P.PROPERTIES, ids=[p.name for p in P.PROPERTIES])
def test_reference_property(prop, linsolve):
    A = OracleAdapter(linsolve)
    for d in prop.draws(CASES, seed=20260 + len(prop.name)):
        prop.run(A, d)


def test_square_property():
    """proptests.rs:294-330"""
    A = OracleAdapter()
    for d in P.square_draws(CASES, seed=4):
        P.square_property(A, d)


@pytest.mark.parametrize("case", P.FIXED_CASES, ids=[c[0] for c in P.FIXED_CASES])
def test_fixed_cases_and_regression_seeds(case):
    name, prop, d, holds = case
    p = P.BY_NAME[prop]
    reqs, guesses = p.build(d)
    out = OracleAdapter().solve(reqs, guesses)
    assert out.error == 0
    if holds:
        p.check(out, d)
    # either way both linear solvers of the oracle walk the same LM path
    out2 = OracleAdapter(O.LINSOLVE_SPARSE).solve(reqs, guesses)
    assert out2.iterations == out.iterations and out2.converged == out.converged and out2.unsatisfied == out.unsatisfied
    assert np.allclose(out2.final_values, out.final_values, rtol=0, atol=1e-9)


# ---- DistanceVar, proptests.rs:612-707 (constraint level: residual / jacobian_rows) --------------------------------------------
def _distance_var(px, py, qx, qy, d):  # make_distance_var_constraint, proptests.rs:771-803
    return O.distance_var((0, 1), (2, 3), 4), np.array([px, py, qx, qy, d])


def _pd(row, var):
    return next((pd for i, pd in row if i == var), None)


def test_distance_var_jacobian_entries_stay_finite():
    """proptests.rs:612-640: exact coincidence, near-coincidence and general positions."""
    rng = np.random.default_rng(612)
    for _ in range(CASES):
        px, py, qx_any, qy_any, d = rng.uniform(-100, 100, 5)
        mode = int(rng.integers(0, 3))
        qx, qy = (px, py) if mode == 0 else (px + P.EPSILON * 0.5, py - P.EPSILON * 0.5) if mode == 1 else (qx_any, qy_any)
        c, x = _distance_var(px, py, qx, qy, d)
        rows, _ = O.jacobian_rows(c, x)
        assert all(math.isfinite(pd) for _, pd in rows[0])
        df_dd = _pd(rows[0], 4)
        assert df_dd is None or math.isfinite(df_dd)


def test_distance_var_analytic_jacobian_matches_finite_difference():
    """proptests.rs:642-670"""
    rng = np.random.default_rng(642)
    n = 0
    while n < CASES:
        px, py, qx, qy, d = rng.uniform(-100, 100, 5)
        if not math.hypot(px - qx, py - qy) > 1e-2:
            continue
        n += 1
        c, x = _distance_var(px, py, qx, qy, d)
        rows, degenerate = O.jacobian_rows(c, x)
        assert not degenerate
        for var in range(5):
            analytic = _pd(rows[0], var)
            assert analytic is not None
            step = 1e-6 * (1.0 + abs(x[var]))
            xp, xm = x.copy(), x.copy()
            xp[var] += step
            xm[var] -= step
            (rp,), dp = O.residual(c, xp)
            (rm,), dm = O.residual(c, xm)
            assert not dp and not dm
            numeric = (rp - rm) / (2.0 * step)
            assert abs(analytic - numeric) <= 1e-6 + 1e-4 * max(abs(analytic), abs(numeric))


def test_distance_var_is_symmetric_under_point_swap():
    """proptests.rs:672-707"""
    rng = np.random.default_rng(672)
    for _ in range(CASES):
        px, py, qx, qy, d = rng.uniform(-100, 100, 5)
        c, x = _distance_var(px, py, qx, qy, d)
        swapped = O.distance_var((2, 3), (0, 1), 4)
        (r,), _ = O.residual(c, x)
        (rs,), _ = O.residual(swapped, x)
        assert abs(r - rs) <= 1e-12
        rows, deg = O.jacobian_rows(c, x)
        rows_s, deg_s = O.jacobian_rows(swapped, x)
        assert deg == deg_s
        for var in range(5):
            assert abs((_pd(rows[0], var) or 0.0) - (_pd(rows_s[0], var) or 0.0)) <= 1e-12
        if not deg:
            assert abs((_pd(rows[0], 0) or 0.0) + (_pd(rows[0], 2) or 0.0)) <= 1e-12
            assert abs((_pd(rows[0], 1) or 0.0) + (_pd(rows[0], 3) or 0.0)) <= 1e-12
