"""The reference's solve-level property tests (tests/proptest_cases.py: proptests.rs:294-707, the fixed cases and the
shrunk regression inputs) on the HIP path through the C ABI (-m gpu): every property must hold on ezpz_solve, and every
draw must agree with the oracle -- iteration count, flags, warnings, coordinates (1e-6 on what the constraints
determine; the bar of tests/sensitivity.py where the oracle itself is that sensitive, e.g. a point free to slide on an arc)."""
import numpy as np
import pytest

import proptest_cases as P
from adapters import GpuAdapter, OracleAdapter
from oracle import oracle as O
from sensitivity import assert_batch_matches_oracle

pytestmark = pytest.mark.gpu
CASES = 96


@pytest.fixture(scope="module")
def A():
    a = GpuAdapter()
    if a.E.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return a


def agree(A, reqs, guesses, got, name):
    want = OracleAdapter().solve(reqs, guesses)
    assert got.error == want.error and got.unsatisfied == want.unsatisfied and got.warnings == want.warnings, name
    recs = O.stack([O.set_from_initial_values(c, np.array([v for _, v in sorted(guesses)])) for c in reqs])
    x0 = np.array([v for _, v in sorted(guesses)])[None, :]
    assert_batch_matches_oracle(recs, x0, np.asarray(got.final_values)[None, :], [got.iterations], [got.converged],
                                linsolve=O.LINSOLVE_DENSE, oracle_result=(np.asarray(want.final_values)[None, :], [want.iterations], [want.converged]),
                                what=name)


@pytest.mark.parametrize("prop", P.PROPERTIES, ids=[p.name for p in P.PROPERTIES])
def test_reference_property_holds_and_agrees_with_the_oracle(A, prop):
    for d in prop.draws(CASES, seed=20260 + len(prop.name)):
        reqs, guesses = prop.build(d)
        got = A.solve(reqs, guesses)
        prop.check(got, d)
        agree(A, reqs, guesses, got, (prop.name, d))


def test_square_property(A):
    """proptests.rs:294-330 through the product's own text front end."""
    for d in P.square_draws(CASES, seed=4):
        got = P.square_property(A, d)
        want = P.square_property(OracleAdapter(), d)
        assert got.iterations == want.iterations and got.converged == want.converged
        assert np.all(np.abs(got.final_values - want.final_values) <= 1e-6 * np.maximum(1.0, np.abs(want.final_values)))


@pytest.mark.parametrize("case", P.FIXED_CASES, ids=[c[0] for c in P.FIXED_CASES])
def test_fixed_cases_and_regression_seeds(A, case):
    name, prop, d, holds = case
    p = P.BY_NAME[prop]
    reqs, guesses = p.build(d)
    got = A.solve(reqs, guesses)
    assert got.error == 0
    if holds:
        p.check(got, d)
    agree(A, reqs, guesses, got, name)


def test_distance_var_properties_on_the_evaluation_kernel(A):
    """proptests.rs:612-707 on ezpz_system_eval_batch: partials finite at (near-)coincident points, equal to central
    differences of the kernel's own residual, invariant under swapping the points."""
    E = A.E
    rng = np.random.default_rng(612)
    c = O.distance_var((0, 1), (2, 3), 4)
    swapped = O.distance_var((2, 3), (0, 1), 4)
    sys_c, sys_s = E.System(O.stack([c]), 5), E.System(O.stack([swapped]), 5)
    x = rng.uniform(-100, 100, (3 * 256, 5))
    x[0:256, 2:4] = x[0:256, 0:2]                                                # exact coincidence
    x[256:512, 2] = x[256:512, 0] + P.EPSILON * 0.5                                # near-coincidence
    x[256:512, 3] = x[256:512, 1] - P.EPSILON * 0.5
    r, J, deg = sys_c.eval_batch(x)
    rs, Js, degs = sys_s.eval_batch(x)
    assert np.all(np.isfinite(J))
    assert np.all(np.abs(r - rs) <= 1e-12) and np.all(np.abs(J - Js) <= 1e-12) and np.array_equal(deg, degs)
    general = np.hypot(x[:, 0] - x[:, 2], x[:, 1] - x[:, 3]) > 1e-2
    assert np.all(deg[general] == 0)
    g = x[general]
    assert np.all(np.abs(J[general][:, 0, 0] + J[general][:, 0, 2]) <= 1e-12) and np.all(np.abs(J[general][:, 0, 1] + J[general][:, 0, 3]) <= 1e-12)
    for var in range(5):
        step = 1e-6 * (1.0 + np.abs(g[:, var]))
        xp, xm = g.copy(), g.copy()
        xp[:, var] += step
        xm[:, var] -= step
        numeric = (sys_c.eval_batch(xp)[0][:, 0] - sys_c.eval_batch(xm)[0][:, 0]) / (2.0 * step)
        analytic = J[general][:, 0, var]
        assert np.all(np.abs(analytic - numeric) <= 1e-6 + 1e-4 * np.maximum(np.abs(analytic), np.abs(numeric)))
