"""Per-constraint math of the oracle: the reference's property tests and unit known-answers.

  * analytic Jacobian == finite difference of the residual for all 25 kinds
    (proptests.rs:187-234, FD helper :743-770, tolerance 1e-6 + 1e-4*max)
  * Jacobian scale invariance (proptests.rs:243-292)
  * sympy-derived known answers (constraints.rs:2741-2783, :2859-2956)
"""
import math

import numpy as np
import pytest

import gen
from oracle import oracle as O

SCALED_PARAM_KINDS = {O.DISTANCE, O.VERTICAL_DISTANCE, O.HORIZONTAL_DISTANCE, O.FIXED, O.CIRCLE_RADIUS,
                      O.ARC_RADIUS, O.ARC_LENGTH, O.POINT_LINE_DISTANCE, O.VERTICAL_POINT_LINE_DISTANCE,
                      O.HORIZONTAL_POINT_LINE_DISTANCE}


def _fd(c, vals, var, comp):
    step = 1e-6 * (1.0 + abs(vals[var]))

    def ev(delta):
        v = vals.copy()
        v[var] += delta
        r, deg = O.residual(c, v)
        return None if deg else r[comp]

    fp, fm, f0 = ev(step), ev(-step), ev(0.0)
    if fp is None or fm is None or f0 is None:
        return None
    central = (fp - fm) / (2.0 * step)
    forward = (fp - f0) / step
    backward = (f0 - fm) / step
    smooth = abs(forward - backward) <= 1e-3 * (1.0 + abs(central)) + 1e-6
    return central if smooth else None


@pytest.mark.parametrize("kind", range(O.NUM_KINDS), ids=O.KIND_NAMES)
def test_analytic_jacobian_matches_finite_difference(kind):
    rng = np.random.default_rng(1000 + kind)
    checked = 0
    for _ in range(150):
        c = gen.arb_constraint(rng, kind)
        vals = rng.uniform(-8.0, 8.0, size=32)
        rows, deg = O.jacobian_rows(c, vals)
        if deg:
            continue
        ids = sorted({i for row in O.nonzeroes(c) for i in row})
        for comp, row in enumerate(rows):
            for var in ids:
                numeric = _fd(c, vals, var, comp)
                if numeric is None:
                    continue
                analytic = sum(pd for (i, pd) in row if i == var)
                tol = 1e-6 + 1e-4 * max(abs(analytic), abs(numeric))
                assert abs(analytic - numeric) <= tol, (O.KIND_NAMES[kind], comp, var, analytic, numeric)
                checked += 1
    assert checked > 50


@pytest.mark.parametrize("kind", range(O.NUM_KINDS), ids=O.KIND_NAMES)
def test_residual_jacobian_is_scale_invariant(kind):
    rng = np.random.default_rng(2000 + kind)
    k = 8.0
    for _ in range(100):
        c = gen.arb_constraint(rng, kind)
        vals = rng.uniform(-8.0, 8.0, size=32)
        a, da = O.jacobian_rows(c, vals)
        if da:
            continue
        cs = c.copy()
        if kind in SCALED_PARAM_KINDS:
            cs["param"] = c["param"] * k
        b, db = O.jacobian_rows(cs, vals * k)
        if db:
            continue
        for ra, rb in zip(a, b):
            assert len(ra) == len(rb)
            for (ia, pa), (ib, pb) in zip(ra, rb):
                assert ia == ib
                if not (math.isfinite(pa) and math.isfinite(pb)):
                    continue
                assert abs(pa - pb) <= 1e-6 * (1.0 + max(abs(pa), abs(pb))), (O.KIND_NAMES[kind], ia, pa, pb)


def test_pds_of_symmetric_sympy_values():
    """constraints.rs:2741-2783"""
    c = O.symmetric((0, 1), (2, 3), (4, 5), (6, 7))
    x = [1.0, 2.0, 0.5, -1.0, 3.0, 4.0, 0.0, 0.0]
    rows, deg = O.jacobian_rows(c, x)
    assert not deg
    exp = {0: [3.59386413440468, 0.482103725346969], 1: [-0.598977355734112, -0.0803506208911613],
           2: [-1.64791818845873, -0.806428049671293], 3: [0.274653031409788, 0.134404674945215],
           4: [-0.945945945945946, 0.324324324324324], 5: [0.324324324324324, 0.945945945945946],
           6: [-1.0, 0.0], 7: [0.0, -1.0]}
    for r in range(2):
        for i, pd in rows[r]:
            assert abs(pd - exp[i][r]) <= 1e-5


def test_pds_for_point_line_known_answers():
    """constraints.rs:2859-2956"""
    s2 = math.sqrt(2.0)
    tests = [
        ([0.0, 1.0, 0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, -1.0, 0.0, 0.0]),
        ([2.0, 0.0, 0.0, 0.0, 2.0, 2.0], [-s2 / 2, s2 / 2, s2 / 4, -s2 / 4, s2 / 4, -s2 / 4]),
        ([5.0, 1.0, 2.0, -1.0, 2.0, 3.0], [-1.0, 0.0, 0.5, 0.0, 0.5, 0.0]),
    ]
    c = O.point_line_distance((0, 1), (2, 3), (4, 5), 0.0)
    for x, exp in tests:
        rows, _ = O.jacobian_rows(c, x)
        assert [i for i, _ in rows[0]] == [0, 1, 2, 3, 4, 5]
        for (i, pd), e in zip(rows[0], exp):
            assert abs(pd - e) < 1e-9


def test_equation_of_line_through_residual():
    """constraints.rs:2785-2826: (A,B,C) checked through the signed distance (A px + B py + C)/hypot(A,B)."""
    for (px, py, qx, qy), (a, b, cc) in [((1.0, 2.0, 3.0, 3.0), (-1.0, 2.0, -3.0)), ((0.0, 0.0, 5.0, 0.0), (0.0, 5.0, 0.0)),
                                         ((2.0, 1.0, 2.0, 4.0), (-3.0, 0.0, 6.0)), ((-2.0, 3.0, 1.0, -1.0), (4.0, 3.0, -1.0))]:
        c = O.point_line_distance((0, 1), (2, 3), (4, 5), 0.0)
        x = [0.7, -1.3, px, py, qx, qy]
        r, deg = O.residual(c, x)
        assert not deg
        assert abs(r[0] - (a * 0.7 + b * -1.3 + cc) / math.hypot(a, b)) < 1e-12


def test_residual_dim_and_nonzeroes_shape():
    """constraints.rs:378-491, :954-993 and datatypes.rs:126-138 (arc.all_variables = start, end, center)."""
    two_row = {O.POINTS_COINCIDENT, O.ARC_RADIUS, O.MIDPOINT, O.SYMMETRIC, O.POINT_ARC_COINCIDENT, O.ARC_LENGTH,
               O.POINTS_AT_ANGLE}
    rng = np.random.default_rng(7)
    for kind in range(O.NUM_KINDS):
        c = gen.arb_constraint(rng, kind)
        assert O.residual_dim(c) == (2 if kind in two_row else 1)
        assert len(O.nonzeroes(c)) == O.residual_dim(c)
    arc = O.arc((0, 1), (2, 3), (6, 7))
    assert O.nonzeroes(arc) == [[2, 3, 6, 7, 0, 1]]
    assert O.nonzeroes(O.arc_angle((0, 1), (2, 3), (4, 5), ("deg", 30.0))) == [[0, 1, 2, 3, 0, 1, 4, 5]]
    assert O.nonzeroes(O.points_coincident((0, 1), (2, 3))) == [[0, 2], [1, 3]]
    assert O.nonzeroes(O.arc_radius((0, 1), (2, 3), (4, 5), 1.0)) == [[0, 1, 2, 3], [0, 1, 4, 5]]
    assert O.nonzeroes(O.midpoint((0, 1), (2, 3), (4, 5))) == [[0, 2, 4], [1, 3, 5]]
    assert O.nonzeroes(O.vertical_point_line_distance((0, 1), (2, 3), (4, 5), 1.0)) == [[2, 3, 4, 5, 0, 1]]


def test_side_inference():
    """constraints.rs:146-193"""
    c = O.line_tangent_to_circle((0, 1), (2, 3), (4, 5), 6)
    assert O.set_from_initial_values(c, [0, 3, 5, 3, 2, 4.5, 1.5])["tag"] == O.LINE_LEFT
    assert O.set_from_initial_values(c, [0, 3, 5, 3, 2, 1.5, 1.5])["tag"] == O.LINE_RIGHT
    assert O.set_from_initial_values(c, [0, 3, 5, 3, 2, 3.0, 1.5])["tag"] == O.LINE_LEFT  # cross == 0 -> Left
    cc = O.circle_tangent_to_circle((0, 1), 2, (3, 4), 5)
    assert O.set_from_initial_values(cc, [0, 0, 2, 4, 0, 3])["tag"] == O.CIRCLE_EXTERIOR
    assert O.set_from_initial_values(cc, [0, 0, 5, 1, 0, 2])["tag"] == O.CIRCLE_INTERIOR


# ---- FreedomAnalysis (solver/find_dof.rs) -----------------------------------------------------------------------------
def _projector_participation(J, tol=1e-8):
    """diag of the orthogonal projector onto null(J), by SVD (independent of any QR)."""
    u, sv, vt = np.linalg.svd(J)
    rank = int(np.sum(sv > tol * max(sv.max(), 1e-300))) if sv.size else 0
    N = vt[rank:].T
    return (N * N).sum(axis=1), J.shape[1] - rank


def test_freedom_analysis_dense_matches_svd_projector():
    """find_dof.rs:31-103: the participation is the diagonal of the projector onto null(J), whatever basis is used."""
    rng = np.random.default_rng(77)
    for trial in range(60):
        m, n = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        J = rng.uniform(-2, 2, (m, n)) * (rng.uniform(0, 1, (m, n)) < 0.6)
        if trial % 3 == 0 and m > 1:
            J[-1] = J[0] * 2.0  # dependent row
        if trial % 4 == 0 and n > 1:
            J[:, -1] = 0.0  # a variable nothing constrains
        under, part = O.freedom_analysis_dense(J)
        want, nullity = _projector_participation(J)
        assert np.allclose(part, want, atol=1e-9), (trial, part, want)
        thr = (1e-3 * want.max()) ** 2
        assert under == [j for j in range(n) if want[j] > thr and nullity > 0]


def test_freedom_analysis_known_structures():
    # x0 fixed, x1 - x2 = 0: x1 and x2 move together, x0 does not
    J = np.array([[1.0, 0, 0], [0, 1.0, -1.0]])
    under, part = O.freedom_analysis_dense(J)
    assert under == [1, 2] and np.allclose(part, [0, 0.5, 0.5])
    # square full rank: nothing is free
    assert O.freedom_analysis_dense(np.eye(4))[0] == []
    # wide zero matrix: everything is free
    assert O.freedom_analysis_dense(np.zeros((2, 3)))[0] == [0, 1, 2]
    # no rows or no columns: EmptySystemNotAllowed (find_dof.rs:43-44)
    with pytest.raises(ValueError):
        O.freedom_analysis_dense(np.zeros((0, 3)))


def test_solve_analysis_follows_the_returned_tier():
    """lib.rs:139-146 + :215-246: the analysis belongs to the tier whose outcome is returned."""
    reqs = [O.fixed(0, 1.0, priority=0), O.fixed(1, 2.0, priority=1), O.fixed(1, 3.0, priority=1)]
    out = O.solve(reqs, [(0, 0.0), (1, 0.0)], analysis=True)
    assert out.priority_solved == 0 and out.underconstrained == [1]  # tier 1 is contradictory, tier 0 leaves x1 free
    out = O.solve(reqs[:2], [(0, 0.0), (1, 0.0)], analysis=True)
    assert out.priority_solved == 1 and out.underconstrained == []
    assert O.solve([], [(0, 0.5)], analysis=True).underconstrained == []  # A::no_constraints()
