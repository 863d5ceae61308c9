"""The reference's own known-answer tests, restated against a solver adapter.

Every function here is a transcription of a test in /root/reference/ezpz/src/tests.rs (file:line in
each docstring): same inputs, same asserted values, same tolerance (EPSILON = 1e-4, lib.rs:43).
They are run twice: against the CPU oracle (tests/test_oracle_pins.py, pins the oracle to the
reference) and against the HIP path through the C ABI (tests/test_gpu_parity.py, -m gpu).

Adapter protocol (see tests/adapters.py):
    A.solve(reqs, guesses, config=None) -> result   (reqs: list of 56-byte constraint records)
    A.run(case, filename="problem.md", config=None) -> (result, system)
    result: error, err_constraint_id, err_variable, final_values, iterations, converged, unsatisfied,
            warnings [(about_constraint, content)], priority_solved, num_vars, num_eqs,
            underconstrained (FreedomAnalysis; the adapters always run solve_analysis like tests.rs `run`)
    system: point(values, label), circle(values, label), arc(values, label)
"""
import math

from oracle import oracle as O  # constraint-record constructors + tag constants only

EPSILON = 1e-4
PI = math.pi


def nearly_eq(l, r):
    assert abs(l - r) < EPSILON, f"LHS was {l}, RHS was {r}, difference was {abs(l - r)}"


def points_eq(l, r):
    d = math.hypot(l[0] - r[0], l[1] - r[1])
    assert d < EPSILON, f"LHS was {l}, RHS was {r}, dist was {d}"


def freedom(out, expected):
    """`solved.analysis` of tests.rs `run` (= solve_with_config_analysis): expected None means only
    `is_underconstrained()` is asserted, a list is `underconstrained()` exactly ([] = not underconstrained)."""
    assert out.underconstrained is not None, "adapter did not run the freedom analysis"
    if expected is None:
        assert out.underconstrained
    else:
        assert list(out.underconstrained) == expected, out.underconstrained


def dist(a, b):
    return math.hypot(a[0] - b[0], a[1] - b[1])


def pt(i):
    """DatumPoint::new for the i-th generated point: ids (2i, 2i+1)."""
    return (2 * i, 2 * i + 1)


# ---- API-level tests ----------------------------------------------------------------------------
def empty(A):
    """tests.rs:38-47"""
    out = A.solve([O.fixed(0, 0.0)], [])
    assert out.error != 0


def it_returns_best_satisfied_solution(A):
    """tests.rs:49-68"""
    reqs = [O.fixed(0, 0.0, priority=0), O.fixed(0, 1.0, priority=1), O.fixed(0, 2.0, priority=1)]
    out = A.solve(reqs, [(0, 0.5)])
    assert out.error == 0 and not out.unsatisfied
    assert out.priority_solved == 0


def initials_become_finals_if_no_constraints(A):
    """tests.rs:70-84"""
    out = A.solve([], [(0, 0.5)])
    assert out.error == 0 and not out.unsatisfied
    assert list(out.final_values) == [0.5]
    assert out.iterations == 0 and out.converged


def priority_solver_reports_original_indices(A):
    """tests.rs:86-106"""
    reqs = [O.fixed(0, 0.0, priority=1), O.fixed(0, 1.0, priority=0), O.fixed(0, 2.0, priority=0)]
    out = A.solve(reqs, [(0, 0.5)])
    assert out.unsatisfied == [1, 2]
    assert out.priority_solved == 0


def too_many_variables(A):
    """tests.rs:108-128"""
    out = A.solve([O.fixed(0, 0.0)], [])
    assert out.error == O.ERR_MISSING_GUESS
    assert out.err_constraint_id == 0 and out.err_variable == 0


def reports_missing_guess_for_second_row_ids(A):
    """solver.rs:448-478 (through the public boundary: id 1 has no guess)."""
    out = A.solve([O.points_coincident((0, 1), (2, 3))], [(0, 0.0), (2, 0.0)])
    assert out.error != 0


def weight_biases_inconsistent_solution(A):
    """tests.rs:255-284"""
    out = A.solve([O.fixed(0, 0.0), O.fixed(0, 100.0, weight=100.0)], [(0, 50.0)])
    assert out.final_values[0] > 99.0
    base = A.solve([O.fixed(0, 0.0), O.fixed(0, 100.0)], [(0, 50.0)])
    nearly_eq(base.final_values[0], 50.0)


def _tangent_case(A, side, center_y, expect_y):
    p0, p1, center, radius = pt(0), pt(1), pt(2), 6
    reqs = [
        O.fixed(p0[1], 3.0),
        O.fixed(p1[1], 3.0),
        O.circle_radius(center, radius, 1.5),
        O.line_tangent_to_circle(p0, p1, center, radius, side),
    ]
    guesses = [(0, 0.0), (1, 3.0), (2, 5.0), (3, 3.0), (4, 2.0), (5, center_y), (6, 1.5)]
    out = A.solve(reqs, guesses)
    assert out.error == 0 and not out.unsatisfied
    nearly_eq(out.final_values[5], expect_y)
    nearly_eq(out.final_values[6], 1.5)


def line_tangent_left_explicit(A):
    """tests.rs:341-376"""
    _tangent_case(A, O.LINE_LEFT, 1.5, 4.5)


def line_tangent_right_explicit(A):
    """tests.rs:378-413"""
    _tangent_case(A, O.LINE_RIGHT, 4.5, 1.5)


def line_tangent_left_inferred(A):
    """tests.rs:415-450"""
    _tangent_case(A, O.SIDE_UNDEFINED, 4.5, 4.5)


def line_tangent_right_inferred(A):
    """tests.rs:452-487"""
    _tangent_case(A, O.SIDE_UNDEFINED, 1.5, 1.5)


def _circle_tangent_case(A, ra, bx, rb, expect):
    ca, ida, cb, idb = (0, 1), 2, (3, 4), 5
    guesses = [(0, 0.0), (1, 0.0), (2, ra), (3, bx), (4, 0.0), (5, rb)]
    reqs = [O.fixed(ida, ra), O.fixed(idb, rb), O.circle_tangent_to_circle(ca, ida, cb, idb, O.SIDE_UNDEFINED)]
    out = A.solve(reqs, guesses)
    assert out.error == 0 and not out.unsatisfied
    v = out.final_values
    nearly_eq(dist((v[0], v[1]), (v[3], v[4])), expect)


def circle_tangent_external_inferred(A):
    """tests.rs:489-524"""
    _circle_tangent_case(A, 2.0, 4.0, 3.0, 5.0)


def circle_tangent_internal_inferred(A):
    """tests.rs:526-561"""
    _circle_tangent_case(A, 5.0, 1.0, 2.0, 3.0)


def test_trim_arc2_left_side_arc1_should_remain_fixed(A):
    """tests.rs:763-897"""
    a1c, a1s, a1e, a2c, a2s, a2e = (pt(i) for i in range(6))
    guesses = [(0, 30.0), (1, 0.0), (2, 0.0), (3, 5.0), (4, 0.0), (5, -5.0),
               (6, 0.0), (7, -30.0), (8, 5.0), (9, 0.0), (10, -5.0), (11, 0.0)]
    reqs = [
        O.arc(a1c, a1s, a1e),
        O.arc(a2c, a2s, a2e),
        O.fixed(a1c[0], 30.0), O.fixed(a1c[1], 0.0),
        O.fixed(a1s[0], 0.0), O.fixed(a1s[1], 5.0),
        O.fixed(a1e[0], 0.0), O.fixed(a1e[1], -5.0),
        O.fixed(a2c[0], 0.0), O.fixed(a2c[1], -30.0),
        O.fixed(a2s[0], 5.0), O.fixed(a2s[1], 0.0),
        O.point_arc_coincident(a2c, a2s, a2e, a2e),
        O.point_arc_coincident(a1c, a1s, a1e, a2e),
    ]
    out = A.solve(reqs, guesses)
    assert out.error == 0 and not out.unsatisfied
    for i, e in enumerate([30.0, 0.0, 0.0, 5.0, 0.0, -5.0]):
        nearly_eq(out.final_values[i], e)


def _arc_length_case(A, cx, cy, radius, start_rad, desired, end_guess):
    """tests.rs:899-943"""
    center, start, end = pt(0), pt(1), pt(2)
    sx = cx + math.cos(start_rad) * radius
    sy = cy + math.sin(start_rad) * radius
    guesses = [(0, cx), (1, cy), (2, sx), (3, sy), (4, end_guess[0]), (5, end_guess[1])]
    reqs = [O.arc(center, start, end), O.fixed(0, cx), O.fixed(1, cy), O.fixed(2, sx), O.fixed(3, sy),
            O.arc_length(center, start, end, desired)]
    out = A.solve(reqs, guesses)
    assert out.error == 0
    return out


def _ccw_len(out, cx, cy, radius, start_rad):
    ex, ey = out.final_values[4], out.final_values[5]
    two_pi = 2.0 * PI
    end_rad = math.atan2(ey - cy, ex - cx) % two_pi
    return radius * ((end_rad - start_rad) % two_pi)


def arc_length_ccw_over_pi(A):
    """tests.rs:945-984"""
    out = _arc_length_case(A, 0.0, 0.0, 1.0, 0.0, 1.5 * PI, (0.0, -1.0))
    assert not out.unsatisfied
    nearly_eq(dist((out.final_values[4], out.final_values[5]), (0.0, 0.0)), 1.0)
    nearly_eq(_ccw_len(out, 0.0, 0.0, 1.0, 0.0), 1.5 * PI)


def arc_length_near_zero(A):
    """tests.rs:986-1016"""
    cx, cy, r, s0 = -2.0, 3.0, 5.0, 0.25 * PI
    eg = (cx + math.cos(s0 + 1.0e-2) * r, cy + math.sin(s0 + 1.0e-2) * r)
    out = _arc_length_case(A, cx, cy, r, s0, 1.0e-3, eg)
    assert not out.unsatisfied
    nearly_eq(_ccw_len(out, cx, cy, r, s0), 1.0e-3)


def arc_length_near_full_circle(A):
    """tests.rs:1018-1048"""
    cx, cy, r, s0 = 1.0, -1.0, 2.5, 0.0
    desired = 2.0 * PI * r - 1.0e-3
    eg = (cx + math.cos(-1.0e-2) * r, cy + math.sin(-1.0e-2) * r)
    out = _arc_length_case(A, cx, cy, r, s0, desired, eg)
    assert not out.unsatisfied
    nearly_eq(_ccw_len(out, cx, cy, r, s0), desired)


def arc_length_degenerate_warns(A):
    """tests.rs:1050-1087"""
    center, start, end = pt(0), pt(1), pt(2)
    guesses = [(0, 0.0), (1, 0.0), (2, 0.0), (3, 0.0), (4, 1.0), (5, 0.0)]
    reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.fixed(2, 0.0), O.fixed(3, 0.0),
            O.arc_length(center, start, end, 1.0)]
    out = A.solve(reqs, guesses)
    assert out.error == 0
    assert any(w[1] == O.WARN_DEGENERATE for w in out.warnings)


def strange_nonconvergence(A):
    """tests.rs:1089-1127: iterations == 2"""
    p, q, r, s, t = (pt(i) for i in range(5))
    reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.points_coincident(r, s), O.points_coincident(q, p),
            O.lines_equal_length(q, r, s, t)]
    guesses = [(0, 0.0), (1, -0.02), (2, -3.39), (3, -0.38), (4, -2.76), (5, 4.83), (6, -1.54), (7, 5.21),
               (8, -1.15), (9, 2.75)]
    out = A.solve(reqs, guesses, config=dict(max_iterations=31))
    assert out.error == 0
    assert out.iterations == 2


def _pac_fixed_arc(A, center, start, end, initial):
    """tests.rs:1204-1245"""
    c, s, e, p = pt(0), pt(1), pt(2), pt(3)
    reqs = [O.point_arc_coincident(c, s, e, p),
            O.fixed(0, center[0]), O.fixed(1, center[1]), O.fixed(2, start[0]), O.fixed(3, start[1]),
            O.fixed(4, end[0]), O.fixed(5, end[1])]
    guesses = [(0, center[0]), (1, center[1]), (2, start[0]), (3, start[1]), (4, end[0]), (5, end[1]),
               (6, initial[0]), (7, initial[1])]
    out = A.solve(reqs, guesses)
    assert out.error == 0 and not out.unsatisfied, "constraint should be satisfied"
    return (out.final_values[6], out.final_values[7])


def _signed_angle(a, b):
    return math.atan2(a[0] * b[1] - a[1] * b[0], a[0] * b[0] + a[1] * b[1])


def _assert_point_on_arc_ccw(point, center, start, end):
    """tests.rs:1175-1202"""
    radius = dist(start, center)
    nearly_eq(dist(point, center), radius)
    s = (start[0] - center[0], start[1] - center[1])
    e = (end[0] - center[0], end[1] - center[1])
    p = (point[0] - center[0], point[1] - center[1])
    two_pi = 2.0 * PI
    a_sp = _signed_angle(s, p) % two_pi
    a_se = _signed_angle(s, e) % two_pi
    if a_sp > two_pi - EPSILON:
        a_sp = 0.0
    assert a_sp <= a_se + EPSILON


def point_arc_coincident_old_incorrect_convergence_1(A):
    """tests.rs:1253-1262"""
    c, s, e = (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)
    sp = _pac_fixed_arc(A, c, s, e, (0.0, -1.0))
    _assert_point_on_arc_ccw(sp, c, s, e)
    points_eq(sp, s)


def point_arc_coincident_old_incorrect_convergence_2(A):
    """tests.rs:1270-1279"""
    c, s, e = (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)
    sp = _pac_fixed_arc(A, c, s, e, (-3.0, -3.0))
    _assert_point_on_arc_ccw(sp, c, s, e)
    points_eq(sp, s)


def lines_at_angle_isolated(A):
    """tests.rs:1505-1607: exact iteration counts 0,0,0,0,0,0,4,4,4,4"""
    cases = [
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 2.0]], 0.5 * PI, 0),
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 2.0]], -0.5 * PI, 0),
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [2.0, 0.0]], 0.0, 0),
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [2.0, 0.0]], PI, 0),
        ([[0.0, 0.0], [-1.0, 0.0], [0.0, 0.0], [2.0, 0.0]], 0.0, 0),
        ([[0.0, 0.0], [-1.0, 0.0], [0.0, 0.0], [2.0, 0.0]], PI, 0),
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 2.0]], 0.0, 4),
        ([[0.0, 0.0], [1.0, 0.0], [0.0, 0.0], [0.0, 2.0]], PI, 4),
        ([[0.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 2.0]], 0.5 * PI, 4),
        ([[0.0, 0.0], [0.0, 1.0], [0.0, 0.0], [0.0, 2.0]], -0.5 * PI, 4),
    ]
    for points, angle, expected_iters in cases:
        reqs = [O.lines_at_angle(pt(0), pt(1), pt(2), pt(3), ("rad", angle))]
        guesses = []
        for i, (x, y) in enumerate(points):
            guesses += [(2 * i, x), (2 * i + 1, y)]
        out = A.solve(reqs, guesses, config=dict(max_iterations=100))
        assert out.error == 0 and not out.unsatisfied
        assert out.iterations == expected_iters, f"unexpected iteration count for angle {angle}"


def lines_angle_sign_check(A):
    """tests.rs:1609-1684: iterations 3 and 4 + final signed angle"""
    for angle, expected_iters in [(0.1 * PI, 3), (-0.1 * PI, 4)]:
        p0, p1, p2 = pt(0), pt(1), pt(2)
        reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.fixed(2, 1.0), O.fixed(3, 0.0),
                O.lines_at_angle(p0, p1, p1, p2, ("rad", angle))]
        guesses = [(0, 0.0), (1, 0.0), (2, 1.0), (3, 0.0), (4, 2.0), (5, 1.0)]
        out = A.solve(reqs, guesses, config=dict(max_iterations=100))
        assert out.error == 0 and not out.unsatisfied
        assert out.iterations == expected_iters, f"unexpected iteration count for angle {angle}"
        v = out.final_values
        u = (v[2] - v[0], v[3] - v[1])
        w = (v[4] - v[2], v[5] - v[3])
        nearly_eq(_signed_angle(u, w), angle)


def _paa_angle(v):
    u = (v[2] - v[0], v[3] - v[1])
    w = (v[4] - v[0], v[5] - v[1])
    return _signed_angle(u, w)


def points_at_angle_already_satisfied(A):
    """tests.rs:1694-1766: 0 iterations x5"""
    for p1, p2, angle in [([1.0, 0.0], [0.0, 2.0], 0.5 * PI), ([1.0, 0.0], [0.0, -2.0], -0.5 * PI),
                          ([1.0, 0.0], [3.0, 0.0], 0.0), ([1.0, 0.0], [-2.0, 0.0], PI),
                          ([2.0, 0.0], [1.0, 1.0], 0.25 * PI)]:
        reqs = [O.points_at_angle(pt(0), pt(1), pt(2), ("rad", angle))]
        guesses = [(0, 0.0), (1, 0.0), (2, p1[0]), (3, p1[1]), (4, p2[0]), (5, p2[1])]
        out = A.solve(reqs, guesses, config=dict(max_iterations=100))
        assert out.error == 0 and not out.unsatisfied
        assert out.iterations == 0, f"angle {angle} should already be satisfied (0 iterations)"


def points_at_angle_degenerate(A):
    """tests.rs:1768-1792"""
    reqs = [O.points_at_angle(pt(0), pt(1), pt(2), ("deg", 180.0))]
    guesses = [(0, 0.0), (1, 0.0), (2, 13.0), (3, 13.0), (4, 13.0), (5, 13.0)]
    out = A.solve(reqs, guesses, config=dict(max_iterations=100))
    assert out.error == 0
    assert out.warnings[0][1] == O.WARN_DEGENERATE


def points_at_angle_unique_solution(A):
    """tests.rs:1794-1858"""
    target = 0.25 * PI
    reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.fixed(2, 1.0), O.fixed(3, 0.0),
            O.points_at_angle(pt(0), pt(1), pt(2), ("rad", target))]
    for g in ([(0, 0.0), (1, 0.0), (2, 1.0), (3, 0.0), (4, 1.0), (5, 1.0)],
              [(0, 0.0), (1, 0.0), (2, 1.0), (3, 0.0), (4, -1.0), (5, -1.0)]):
        out = A.solve(reqs, g, config=dict(max_iterations=100))
        assert out.error == 0 and not out.unsatisfied
        nearly_eq(_paa_angle(out.final_values), target)


def points_at_angle_sign_distinguishable(A):
    """tests.rs:1860-1917"""
    theta = 0.25 * PI
    for target, init in [(theta, [1.0, 0.0]), (-theta, [1.0, 0.0]), (theta, [0.0, 1.0]), (-theta, [0.0, 1.0]),
                         (theta, [-1.0, 0.0]), (-theta, [-1.0, 0.0]), (theta, [0.0, -1.0]), (-theta, [0.0, -1.0])]:
        reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.fixed(2, 1.0), O.fixed(3, 0.0),
                O.distance(pt(0), pt(2), 1.0), O.points_at_angle(pt(0), pt(1), pt(2), ("rad", target))]
        g = [(0, 0.0), (1, 0.0), (2, 1.0), (3, 0.0), (4, init[0]), (5, init[1])]
        out = A.solve(reqs, g, config=dict(max_iterations=100))
        assert out.error == 0 and not out.unsatisfied
        nearly_eq(_paa_angle(out.final_values), target)


# ---- fixture (.md) tests ---------------------------------------------------------------------------
def coincident(A):
    """tests.rs:130-138"""
    out, s = A.run("coincident")
    freedom(out, [])  # tests.rs:134
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (3.0, 3.0))
    points_eq(s.point(out.final_values, "q"), (3.0, 3.0))


def symmetric(A):
    """tests.rs:147-160"""
    out, s = A.run("symmetric")
    freedom(out, [])  # :151
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p"), (0.0, 0.0))
    points_eq(s.point(v, "q"), (2.0, 2.0))
    points_eq(s.point(v, "a"), (0.5, 0.4))
    points_eq(s.point(v, "b"), (0.4, 0.5))


def perpdist(A):
    """tests.rs:162-184"""
    out, s = A.run("perpdist")
    freedom(out, [4, 5])  # :178-183
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p"), (0.0, 0.0))
    points_eq(s.point(v, "q"), (2.0, 3.0))
    points_eq(s.point(v, "a"), (0.10055560181546289, 1.9536090405127489))


def perpdist_negative(A):
    """tests.rs:186-207"""
    out, s = A.run("perpdist_negative")
    freedom(out, [4, 5])  # :192-197
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p"), (0.0, 0.0))
    points_eq(s.point(v, "q"), (2.0, 3.0))
    points_eq(s.point(v, "a"), (1.5192717280306194, 0.476131954511605))


def midpoint(A):
    """tests.rs:209-218"""
    out, s = A.run("midpoint")
    freedom(out, [])  # :213
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p"), (0.0, 0.0))
    points_eq(s.point(v, "q"), (2.0, 3.0))
    points_eq(s.point(v, "m"), (1.0, 1.5))


def underconstrained(A):
    """tests.rs:220-230"""
    out, s = A.run("underconstrained")
    freedom(out, [0, 1])  # :223-225
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (1.0, 1.0))
    points_eq(s.point(out.final_values, "q"), (0.0, 0.0))


def tiny(A):
    """tests.rs:232-239; CLI size pin ezpz-cli/src/main.rs:277 (4 rows, 4 vars)"""
    out, s = A.run("tiny")
    freedom(out, [])  # :236
    assert not out.unsatisfied
    assert (out.num_eqs, out.num_vars) == (4, 4)
    points_eq(s.point(out.final_values, "p"), (0.0, 0.0))
    points_eq(s.point(out.final_values, "q"), (0.0, 0.0))


def inconsistent(A):
    """tests.rs:241-253"""
    out, s = A.run("inconsistent")
    freedom(out, [])  # :249
    assert out.unsatisfied
    points_eq(s.point(out.final_values, "o"), (0.0, 0.0))
    points_eq(s.point(out.final_values, "p"), (2.5, 2.5))


def circle(A):
    """tests.rs:286-299"""
    out, s = A.run("circle")
    freedom(out, [])  # :290
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (5.0, 5.0))
    c = s.circle(out.final_values, "a")
    nearly_eq(c["radius"], 3.4)
    points_eq(c["center"], (0.1, 0.2))


def circle_center(A):
    """tests.rs:301-311"""
    out, s = A.run("circle_center")
    freedom(out, [])  # :306
    assert not out.unsatisfied
    c = s.circle(out.final_values, "a")
    nearly_eq(c["radius"], 1.0)
    points_eq(c["center"], (0.0, 0.0))


def circle_tangent(A):
    """tests.rs:313-325"""
    out, s = A.run("circle_tangent")
    freedom(out, [])  # :319
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (0.0, 3.0))
    points_eq(s.point(out.final_values, "q"), (5.0, 3.0))
    c = s.circle(out.final_values, "a")
    nearly_eq(c["center"][1], 1.5)
    nearly_eq(c["radius"], 1.5)


def circle_tangent_other_dir(A):
    """tests.rs:327-339"""
    out, s = A.run("circle_tangent_other_dir")
    freedom(out, [])  # :333
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (0.0, 3.0))
    points_eq(s.point(out.final_values, "q"), (5.0, 3.0))
    c = s.circle(out.final_values, "a")
    nearly_eq(c["center"][1], 1.5)
    nearly_eq(c["radius"], 1.5)


def two_rectangles(A):
    """tests.rs:563-578"""
    out, s = A.run("two_rectangles")
    freedom(out, [])  # :567
    assert not out.unsatisfied
    exp = [(1.0, 1.0), (5.0, 1.0), (5.0, 4.0), (1.0, 4.0), (2.0, 2.0), (6.0, 2.0), (6.0, 6.0), (2.0, 6.0)]
    for i, e in enumerate(exp):
        points_eq(s.point(out.final_values, f"p{i}"), e)


def angle_constraints(A):
    """tests.rs:580-591"""
    for f in ("angle_parallel", "angle_parallel_manual"):
        out, s = A.run(f)
        assert not out.unsatisfied
        freedom(out, [])  # :585
        v = out.final_values
        points_eq(s.point(v, "p0"), (0.0, 0.0))
        points_eq(s.point(v, "p1"), (4.0, 4.0))
        points_eq(s.point(v, "p2"), (0.0, 0.0))
        points_eq(s.point(v, "p3"), (4.0, 4.0))


def perpendicular(A):
    """tests.rs:593-602"""
    out, s = A.run("perpendicular")
    freedom(out, [])  # :597
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p0"), (0.0, 0.0))
    points_eq(s.point(v, "p1"), (0.0, 4.0))
    points_eq(s.point(v, "p2"), (0.0, 0.0))
    points_eq(s.point(v, "p3"), (4.0, 0.0))


def nonsquare(A):
    """tests.rs:604-611"""
    out, s = A.run("nonsquare")
    freedom(out, [])  # :608
    assert not out.unsatisfied
    points_eq(s.point(out.final_values, "p"), (0.0, 0.0))
    points_eq(s.point(out.final_values, "q"), (0.0, 0.0))


def square(A):
    """tests.rs:613-626"""
    out, s = A.run("square")
    freedom(out, [])  # :617
    assert not out.unsatisfied
    v = out.final_values
    a, b, c, d = (s.point(v, l) for l in "abcd")
    nearly_eq(a[1] - c[1], b[1] - d[1])
    nearly_eq(a[0] - c[0], d[0] - b[0])


def parallelogram(A):
    """tests.rs:628-646"""
    out, s = A.run("parallelogram")
    freedom(out, [4, 5, 6, 7])  # :633-637
    v = out.final_values
    a, b, c, d = (s.point(v, l) for l in "abcd")
    nearly_eq(a[1] - c[1], b[1] - d[1])
    nearly_eq(a[0] - c[0], b[0] - d[0])


def underdetermined_lines(A):
    """tests.rs:648-665"""
    out, s = A.run("underdetermined_lines")
    freedom(out, [5])  # :655-660
    assert not out.unsatisfied
    v = out.final_values
    points_eq(s.point(v, "p0"), (0.0, 0.0))
    points_eq(s.point(v, "p1"), (4.0, 0.0))
    points_eq(s.point(v, "p2"), (4.0, 4.0))


def arc_radius(A):
    """tests.rs:667-688; CLI size pin ezpz-cli/src/main.rs:298 (4 rows, 8 vars)"""
    out, s = A.run("arc_radius")
    freedom(out, [0, 1, 2, 3, 4, 5])  # :671-683
    assert not out.unsatisfied
    assert (out.num_eqs, out.num_vars) == (4, 8)
    arc = s.arc(out.final_values, "a")
    points_eq(arc["center"], (0.0, 0.0))
    nearly_eq(5.0, dist(arc["a"], (0.0, 0.0)))
    nearly_eq(5.0, dist(arc["b"], (0.0, 0.0)))


def parc_coincident(A):
    """tests.rs:690-703"""
    out, s = A.run("parc_coincident")
    freedom(out, None)  # :695
    assert not out.unsatisfied
    arc = s.arc(out.final_values, "a")
    points_eq(arc["center"], (0.0, 0.0))
    nearly_eq(5.0, dist(arc["a"], (0.0, 0.0)))
    nearly_eq(5.0, dist(arc["b"], (0.0, 0.0)))
    nearly_eq(5.0, dist(arc["center"], s.point(out.final_values, "p")))


def arc_equidistant(A):
    """tests.rs:705-728"""
    out, s = A.run("arc_equidistant")
    freedom(out, [0, 1, 2, 3, 4, 5])  # :709-720
    assert not out.unsatisfied
    arc = s.arc(out.final_values, "a")
    points_eq(arc["center"], (0.0, 0.0))
    nearly_eq(dist(arc["a"], arc["center"]), dist(arc["b"], arc["center"]))


def chamfer_square(A):
    """tests.rs:730-740"""
    out, s = A.run("chamfer_square")
    freedom(out, [])  # :734
    assert not out.unsatisfied
    v = out.final_values
    for l, e in zip("abcde", [(0.0, 40.0), (30.0, 40.0), (40.0, 30.0), (40.0, 0.0), (0.0, 0.0)]):
        points_eq(s.point(v, l), e)


def arc_length(A):
    """tests.rs:742-746"""
    out, s = A.run("arc_length")
    assert not out.unsatisfied


def point_basically_already_on_arc_should_not_cause_much_change_in_sketch(A):
    """tests.rs:1293-1383"""
    out_without, _ = A.run("arc_line_coincident_bug", "problem_without_arc_constraint.md")
    assert out_without.error == 0
    out, s = A.run("arc_line_coincident_bug")
    initial_line4_start = (-2.32, -2.96)
    initial_arc_center = (1.06, -3.26)
    initial_arc_a = (-1.44, -0.99)
    r0 = dist(initial_arc_center, initial_arc_a)
    d0 = abs(dist(initial_line4_start, initial_arc_center) - r0)
    assert d0 < 0.5
    change = dist(s.point(out.final_values, "line4start"), initial_line4_start)
    assert change <= d0 * 10.0


def arc_center_point_coincident(A):
    """tests.rs:1398-1503"""
    out, s = A.run("arc_center_point_coincident")
    v = out.final_values
    initial_line4_start = (-1.16, -2.63)
    c0, a0, b0 = (0.55, -3.31), (2.25, -3.99), (1.43, -1.71)
    px, py = initial_line4_start
    start_cross0 = (a0[0] - c0[0]) * (c0[1] - py) - (a0[1] - c0[1]) * (c0[0] - px)
    end_cross0 = (b0[0] - c0[0]) * (c0[1] - py) - (b0[1] - c0[1]) * (c0[0] - px)
    assert not (start_cross0 <= 0.0 and end_cross0 < 0.0)
    p = s.point(v, "line4start")
    arc = s.arc(v, "arc1")
    movement = dist(p, initial_line4_start)
    radius = dist(arc["center"], arc["a"])
    assert abs(dist(p, arc["center"]) - radius) < 0.01
    if start_cross0 > 0.1:
        assert movement > radius * 0.3
    cx, cy = arc["center"]
    start_cross = (arc["a"][0] - cx) * (cy - p[1]) - (arc["a"][1] - cy) * (cx - p[0])
    end_cross = (arc["b"][0] - cx) * (cy - p[1]) - (arc["b"][1] - cy) * (cx - p[0])
    assert start_cross < 0.01
    assert end_cross < 1e-6


def warnings_lint(A):
    """tests.rs:1129-1158"""
    txt = """# constraints
point p
point q
p.x = 0
p.y = 0
q.y = 0
vertical(p, q)
point r
point s
r.x = 0
s.x = 0
s.y = 0
lines_at_angle(p, q, r, s, 0rad)

# guesses
p roughly (3, 4)
q roughly (5, 6)
r roughly (3, 4)
s roughly (5, 6)
"""
    out, s = A.run_text(txt)
    assert out.warnings
    assert (7, O.WARN_SHOULD_BE_PARALLEL) in out.warnings


def massive_readme(A):
    """README.md:36-38: the 2000 x 2000 parallel-line system needs 2 iterations."""
    from oracle import textual as T

    out, s = A.run_text(T.gen_big_problem(500))
    assert (out.num_eqs, out.num_vars) == (2000, 2000)
    assert out.iterations == 2 and out.converged and not out.unsatisfied
    v = out.final_values
    for line in (0, 1, 250, 499):
        points_eq((v[4 * line], v[4 * line + 1]), (float(line), 0.0))
        points_eq((v[4 * line + 2], v[4 * line + 3]), (float(line), 4.0))


API_CASES = [
    empty, it_returns_best_satisfied_solution, initials_become_finals_if_no_constraints,
    priority_solver_reports_original_indices, too_many_variables, reports_missing_guess_for_second_row_ids,
    weight_biases_inconsistent_solution, line_tangent_left_explicit, line_tangent_right_explicit,
    line_tangent_left_inferred, line_tangent_right_inferred, circle_tangent_external_inferred,
    circle_tangent_internal_inferred, test_trim_arc2_left_side_arc1_should_remain_fixed,
    arc_length_ccw_over_pi, arc_length_near_zero, arc_length_near_full_circle, arc_length_degenerate_warns,
    strange_nonconvergence, point_arc_coincident_old_incorrect_convergence_1,
    point_arc_coincident_old_incorrect_convergence_2, lines_at_angle_isolated, lines_angle_sign_check,
    points_at_angle_already_satisfied, points_at_angle_degenerate, points_at_angle_unique_solution,
    points_at_angle_sign_distinguishable,
]
FIXTURE_CASES = [
    coincident, symmetric, perpdist, perpdist_negative, midpoint, underconstrained, tiny, inconsistent, circle,
    circle_center, circle_tangent, circle_tangent_other_dir, two_rectangles, angle_constraints, perpendicular,
    nonsquare, square, parallelogram, underdetermined_lines, arc_radius, parc_coincident, arc_equidistant,
    chamfer_square, arc_length, point_basically_already_on_arc_should_not_cause_much_change_in_sketch,
    arc_center_point_coincident, warnings_lint, massive_readme,
]
ALL_CASES = API_CASES + FIXTURE_CASES
