"""ctypes front for the CPU oracle (oracle/libezpz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under ezpz_amd/ imports this module.

The constraint record is the 56-byte POD both the oracle and the product accept
(`CONSTRAINT_DTYPE`); the helper constructors mirror the variants of
`ezpz::Constraint` (reference ezpz/src/constraints.rs:37-93).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libezpz_oracle.so")

# kind tags: enum order of constraints.rs:37-93
LINE_TANGENT_TO_CIRCLE = 0
CIRCLE_TANGENT_TO_CIRCLE = 1
DISTANCE = 2
DISTANCE_VAR = 3
VERTICAL_DISTANCE = 4
HORIZONTAL_DISTANCE = 5
VERTICAL = 6
HORIZONTAL = 7
LINES_AT_ANGLE = 8
FIXED = 9
SCALAR_EQUAL = 10
POINTS_COINCIDENT = 11
CIRCLE_RADIUS = 12
LINES_EQUAL_LENGTH = 13
ARC_RADIUS = 14
ARC = 15
MIDPOINT = 16
POINT_LINE_DISTANCE = 17
VERTICAL_POINT_LINE_DISTANCE = 18
HORIZONTAL_POINT_LINE_DISTANCE = 19
SYMMETRIC = 20
POINT_ARC_COINCIDENT = 21
ARC_LENGTH = 22
ARC_ANGLE = 23
POINTS_AT_ANGLE = 24
NUM_KINDS = 25

KIND_NAMES = [
    "LineTangentToCircle", "CircleTangentToCircle", "Distance", "DistanceVar", "VerticalDistance",
    "HorizontalDistance", "Vertical", "Horizontal", "LinesAtAngle", "Fixed", "ScalarEqual",
    "PointsCoincident", "CircleRadius", "LinesEqualLength", "ArcRadius", "Arc", "Midpoint",
    "PointLineDistance", "VerticalPointLineDistance", "HorizontalPointLineDistance", "Symmetric",
    "PointArcCoincident", "ArcLength", "ArcAngle", "PointsAtAngle",
]
# number of ids used by each kind
KIND_NUM_IDS = [7, 6, 4, 5, 4, 4, 4, 4, 8, 1, 2, 4, 3, 8, 6, 6, 6, 6, 6, 6, 8, 8, 6, 6, 6]

SIDE_UNDEFINED, LINE_LEFT, LINE_RIGHT = 0, 1, 2
CIRCLE_EXTERIOR, CIRCLE_INTERIOR = 1, 2
ANGLE_PARALLEL, ANGLE_PERPENDICULAR, ANGLE_OTHER_DEG, ANGLE_OTHER_RAD = 0, 1, 2, 3

WARN_DEGENERATE, WARN_SHOULD_BE_PARALLEL, WARN_SHOULD_BE_PERPENDICULAR = 0, 1, 2

ERR_WRONG_NUMBER_GUESSES = -2
ERR_MISSING_GUESS = -3
ERR_EMPTY_SYSTEM = -8

LINSOLVE_DENSE, LINSOLVE_SPARSE = 0, 1

CONSTRAINT_DTYPE = np.dtype(
    [
        ("kind", "<u2"),
        ("tag", "u1"),
        ("flags", "u1"),
        ("priority", "<u4"),
        ("ids", "<u4", (8,)),
        ("param", "<f8"),
        ("weight", "<f8"),
    ]
)
assert CONSTRAINT_DTYPE.itemsize == 56


class _Config(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_uint64),
        ("residual_tolerance", C.c_double),
        ("step_tolerance", C.c_double),
        ("initial_lambda", C.c_double),
    ]


class _Warning(C.Structure):
    _fields_ = [("about_constraint", C.c_int32), ("content", C.c_int32)]


class _Outcome(C.Structure):
    _fields_ = [
        ("error", C.c_int32),
        ("err_constraint_id", C.c_int32),
        ("err_variable", C.c_int64),
        ("iterations", C.c_uint64),
        ("converged", C.c_int32),
        ("priority_solved", C.c_uint32),
        ("n_unsatisfied", C.c_uint64),
        ("n_warnings", C.c_uint64),
        ("num_vars", C.c_uint64),
        ("num_eqs", C.c_uint64),
        ("final_lambda", C.c_double),
        ("final_residual_inf", C.c_double),
    ]


@dataclass
class Config:
    """ezpz::Config, solver.rs:31-81 (defaults :72-81)."""

    max_iterations: int = 35
    residual_tolerance: float = 1e-8
    step_tolerance: float = 1e-12
    initial_lambda: float = 1e-9

    def _c(self) -> _Config:
        return _Config(self.max_iterations, self.residual_tolerance, self.step_tolerance, self.initial_lambda)


@dataclass
class Outcome:
    """SolveOutcome (solve_outcome.rs:12-26) or FailureOutcome (:126-136) when error != 0."""

    error: int
    err_constraint_id: int
    err_variable: int
    final_values: np.ndarray
    iterations: int
    converged: bool
    unsatisfied: List[int]
    warnings: List[Tuple[int, int]]
    priority_solved: int
    num_vars: int
    num_eqs: int
    final_lambda: float = 0.0
    final_residual_inf: float = 0.0
    underconstrained: Optional[List[int]] = None  # FreedomAnalysis (analysis.rs:24-31); None = not requested

    def is_satisfied(self) -> bool:
        return not self.unsatisfied


_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("ezpz_oracle.c", "ezpz_oracle_solve.c", "ezpz_oracle.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.orc_residual_dim.restype = C.c_int
        L.orc_residual_dim.argtypes = [vp]
        L.orc_nonzeroes.restype = C.c_int
        L.orc_nonzeroes.argtypes = [vp, vp, C.POINTER(C.c_int), vp, C.POINTER(C.c_int)]
        L.orc_residual.restype = None
        L.orc_residual.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
        L.orc_jacobian_rows.restype = None
        L.orc_jacobian_rows.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int), vp, vp, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
        L.orc_set_from_initial_values.restype = None
        L.orc_set_from_initial_values.argtypes = [vp, vp]
        L.orc_solve.restype = C.c_int
        L.orc_solve.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(_Config), C.c_int, vp, vp, vp,
                                C.c_size_t, C.POINTER(_Outcome)]
        L.orc_solve_analysis.restype = C.c_int
        L.orc_solve_analysis.argtypes = list(L.orc_solve.argtypes) + [vp, C.POINTER(C.c_uint64)]
        L.orc_freedom_analysis_dense.restype = C.c_int
        L.orc_freedom_analysis_dense.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.POINTER(C.c_uint64), vp]
        L.orc_solve_inner.restype = C.c_int
        L.orc_solve_inner.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(_Config), C.c_int, vp,
                                      vp, vp, C.c_size_t, C.POINTER(_Outcome)]
        L.orc_time_solves_analysis.restype = C.c_double
        L.orc_time_solves_analysis.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(_Config), C.c_int, C.c_int,
                                               C.POINTER(C.c_uint64)]
        L.orc_time_solves.restype = C.c_double
        L.orc_time_solves.argtypes = [vp, C.c_size_t, vp, vp, C.c_size_t, C.POINTER(_Config), C.c_int, C.c_int,
                                      C.POINTER(C.c_uint64)]
        L.orc_solve_batch.restype = C.c_int
        L.orc_solve_batch.argtypes = [vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.POINTER(_Config), C.c_int,
                                      C.c_int, vp, vp, vp, vp]
        _lib = L
    return _lib


# ---- constraint constructors (ids in datum field order, see ezpz_oracle.h) -------------------------
Point = Tuple[int, int]  # (x_id, y_id)


def _mk(kind: int, ids: Sequence[int], param: float = 0.0, tag: int = 0, priority: int = 0, weight: float = 1.0):
    rec = np.zeros((), dtype=CONSTRAINT_DTYPE)
    rec["kind"] = kind
    rec["tag"] = tag
    rec["priority"] = priority
    rec["param"] = param
    rec["weight"] = weight
    arr = np.zeros(8, dtype=np.uint32)
    arr[: len(ids)] = np.asarray(list(ids), dtype=np.uint32)
    rec["ids"] = arr
    return rec


def line_tangent_to_circle(p0: Point, p1: Point, center: Point, radius: int, side=SIDE_UNDEFINED, **kw):
    return _mk(LINE_TANGENT_TO_CIRCLE, [*p0, *p1, *center, radius], tag=side, **kw)


def circle_tangent_to_circle(ca: Point, ra: int, cb: Point, rb: int, side=SIDE_UNDEFINED, **kw):
    return _mk(CIRCLE_TANGENT_TO_CIRCLE, [*ca, ra, *cb, rb], tag=side, **kw)


def distance(p0: Point, p1: Point, d: float, **kw):
    return _mk(DISTANCE, [*p0, *p1], d, **kw)


def distance_var(p: Point, q: Point, d_id: int, **kw):
    return _mk(DISTANCE_VAR, [*p, *q, d_id], **kw)


def vertical_distance(p0: Point, p1: Point, d: float, **kw):
    return _mk(VERTICAL_DISTANCE, [*p0, *p1], d, **kw)


def horizontal_distance(p0: Point, p1: Point, d: float, **kw):
    return _mk(HORIZONTAL_DISTANCE, [*p0, *p1], d, **kw)


def vertical(p0: Point, p1: Point, **kw):
    return _mk(VERTICAL, [*p0, *p1], **kw)


def horizontal(p0: Point, p1: Point, **kw):
    return _mk(HORIZONTAL, [*p0, *p1], **kw)


def _angle(angle_kind):
    """angle_kind: 'parallel' | 'perpendicular' | ('deg', v) | ('rad', v)"""
    if angle_kind == "parallel":
        return ANGLE_PARALLEL, 0.0
    if angle_kind == "perpendicular":
        return ANGLE_PERPENDICULAR, 0.0
    unit, val = angle_kind
    return (ANGLE_OTHER_DEG if unit == "deg" else ANGLE_OTHER_RAD), float(val)


def lines_at_angle(l0p0: Point, l0p1: Point, l1p0: Point, l1p1: Point, angle_kind, **kw):
    tag, val = _angle(angle_kind)
    return _mk(LINES_AT_ANGLE, [*l0p0, *l0p1, *l1p0, *l1p1], val, tag=tag, **kw)


def fixed(var: int, value: float, **kw):
    return _mk(FIXED, [var], value, **kw)


def scalar_equal(a: int, b: int, **kw):
    return _mk(SCALAR_EQUAL, [a, b], **kw)


def points_coincident(p0: Point, p1: Point, **kw):
    return _mk(POINTS_COINCIDENT, [*p0, *p1], **kw)


def circle_radius(center: Point, radius: int, r: float, **kw):
    return _mk(CIRCLE_RADIUS, [*center, radius], r, **kw)


def lines_equal_length(l0p0: Point, l0p1: Point, l1p0: Point, l1p1: Point, **kw):
    return _mk(LINES_EQUAL_LENGTH, [*l0p0, *l0p1, *l1p0, *l1p1], **kw)


def arc_radius(center: Point, start: Point, end: Point, r: float, **kw):
    return _mk(ARC_RADIUS, [*center, *start, *end], r, **kw)


def arc(center: Point, start: Point, end: Point, **kw):
    return _mk(ARC, [*center, *start, *end], **kw)


def midpoint(p0: Point, p1: Point, mp: Point, **kw):
    return _mk(MIDPOINT, [*p0, *p1, *mp], **kw)


def point_line_distance(pt: Point, p0: Point, p1: Point, d: float, **kw):
    return _mk(POINT_LINE_DISTANCE, [*pt, *p0, *p1], d, **kw)


def vertical_point_line_distance(pt: Point, p0: Point, p1: Point, d: float, **kw):
    return _mk(VERTICAL_POINT_LINE_DISTANCE, [*pt, *p0, *p1], d, **kw)


def horizontal_point_line_distance(pt: Point, p0: Point, p1: Point, d: float, **kw):
    return _mk(HORIZONTAL_POINT_LINE_DISTANCE, [*pt, *p0, *p1], d, **kw)


def symmetric(lp: Point, lq: Point, a: Point, b: Point, **kw):
    return _mk(SYMMETRIC, [*lp, *lq, *a, *b], **kw)


def point_arc_coincident(center: Point, start: Point, end: Point, pt: Point, **kw):
    return _mk(POINT_ARC_COINCIDENT, [*center, *start, *end, *pt], **kw)


def arc_length(center: Point, start: Point, end: Point, d: float, **kw):
    return _mk(ARC_LENGTH, [*center, *start, *end], d, **kw)


def arc_angle(center: Point, start: Point, end: Point, angle, **kw):
    tag, val = _angle(angle)
    return _mk(ARC_ANGLE, [*center, *start, *end], val, tag=tag, **kw)


def points_at_angle(p0: Point, p1: Point, p2: Point, angle_kind, **kw):
    tag, val = _angle(angle_kind)
    return _mk(POINTS_AT_ANGLE, [*p0, *p1, *p2], val, tag=tag, **kw)


def stack(constraints) -> np.ndarray:
    if isinstance(constraints, np.ndarray) and constraints.dtype == CONSTRAINT_DTYPE:
        return np.ascontiguousarray(constraints).reshape(-1)
    out = np.zeros(len(constraints), dtype=CONSTRAINT_DTYPE)
    for i, c in enumerate(constraints):
        out[i] = c
    return out


# ---- per-constraint evaluation -----------------------------------------------------------------------
def residual_dim(c) -> int:
    a = stack([c])
    return lib().orc_residual_dim(a.ctypes.data)


def nonzeroes(c):
    a = stack([c])
    r0 = np.zeros(8, np.uint32)
    r1 = np.zeros(8, np.uint32)
    n0, n1 = C.c_int(0), C.c_int(0)
    dim = lib().orc_nonzeroes(a.ctypes.data, r0.ctypes.data, C.byref(n0), r1.ctypes.data, C.byref(n1))
    rows = [r0[: n0.value].tolist(), r1[: n1.value].tolist()]
    return rows[:dim]


def residual(c, x) -> Tuple[List[float], bool]:
    a = stack([c])
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = np.zeros(3)
    deg = C.c_int(0)
    lib().orc_residual(a.ctypes.data, x.ctypes.data, r.ctypes.data, C.byref(deg))
    return r[: residual_dim(c)].tolist(), bool(deg.value)


def jacobian_rows(c, x):
    """Returns ([(id, pd), ...] per row, degenerate)."""
    a = stack([c])
    x = np.ascontiguousarray(x, dtype=np.float64)
    i0, i1 = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    p0, p1 = np.zeros(8), np.zeros(8)
    n0, n1, deg = C.c_int(0), C.c_int(0), C.c_int(0)
    lib().orc_jacobian_rows(a.ctypes.data, x.ctypes.data, i0.ctypes.data, p0.ctypes.data, C.byref(n0),
                            i1.ctypes.data, p1.ctypes.data, C.byref(n1), C.byref(deg))
    rows = [list(zip(i0[: n0.value].tolist(), p0[: n0.value].tolist())),
            list(zip(i1[: n1.value].tolist(), p1[: n1.value].tolist()))]
    return rows[: residual_dim(c)], bool(deg.value)


def set_from_initial_values(c, initial_values):
    a = stack([c]).copy()
    x = np.ascontiguousarray(initial_values, dtype=np.float64)
    lib().orc_set_from_initial_values(a.ctypes.data, x.ctypes.data)
    return a[0]


# ---- solve -------------------------------------------------------------------------------------------
def _split_guesses(guesses):
    if isinstance(guesses, np.ndarray) and guesses.ndim == 1 and guesses.dtype.kind == "f":
        ids = np.arange(len(guesses), dtype=np.uint32)
        vals = np.ascontiguousarray(guesses, dtype=np.float64)
    else:
        ids = np.asarray([g[0] for g in guesses], dtype=np.uint32)
        vals = np.asarray([g[1] for g in guesses], dtype=np.float64)
    return np.ascontiguousarray(ids), np.ascontiguousarray(vals)


def solve(reqs, guesses, config: Optional[Config] = None, linsolve: int = LINSOLVE_DENSE,
          warn_cap: int = 4096, analysis: bool = False) -> Outcome:
    """`ezpz::solve` (lib.rs:80-87) or, with analysis=True, `ezpz::solve_analysis` (lib.rs:134-146) on the CPU
    oracle.  guesses: [(id, value), ...] or a float array."""
    cfg = (config or Config())._c()
    a = stack(reqs)
    ids, vals = _split_guesses(guesses)
    n = len(vals)
    x_out = np.zeros(max(n, 1))
    unsat = np.zeros(max(len(a), 1), dtype=np.uint64)
    warns = (_Warning * max(warn_cap, 1))()
    out = _Outcome()
    under = np.zeros(max(n, 1), dtype=np.uint32)
    n_under = C.c_uint64(0)
    args = (a.ctypes.data if len(a) else None, len(a), ids.ctypes.data if n else None,
            vals.ctypes.data if n else None, n, C.byref(cfg), linsolve, x_out.ctypes.data,
            unsat.ctypes.data, C.cast(warns, C.c_void_p), warn_cap, C.byref(out))
    if analysis:
        lib().orc_solve_analysis(*args, under.ctypes.data, C.byref(n_under))
    else:
        lib().orc_solve(*args)
    nw = min(int(out.n_warnings), warn_cap)
    return Outcome(
        underconstrained=under[: n_under.value].astype(int).tolist() if analysis else None,
        error=out.error,
        err_constraint_id=out.err_constraint_id,
        err_variable=out.err_variable,
        final_values=x_out[:n].copy(),
        iterations=int(out.iterations),
        converged=bool(out.converged),
        unsatisfied=unsat[: int(out.n_unsatisfied)].astype(int).tolist(),
        warnings=[(warns[i].about_constraint, warns[i].content) for i in range(nw)],
        priority_solved=int(out.priority_solved),
        num_vars=int(out.num_vars),
        num_eqs=int(out.num_eqs),
        final_lambda=out.final_lambda,
        final_residual_inf=out.final_residual_inf,
    )


def freedom_analysis_dense(jac: np.ndarray):
    """find_dof.rs:31-103 on a dense m x n Jacobian.  Returns (underconstrained indices, participation [n])."""
    jac = np.asfortranarray(jac, dtype=np.float64)
    m, n = jac.shape
    under = np.zeros(max(n, 1), dtype=np.uint32)
    part = np.zeros(max(n, 1))
    cnt = C.c_uint64(0)
    rc = lib().orc_freedom_analysis_dense(jac.ctypes.data, m, n, under.ctypes.data, C.byref(cnt), part.ctypes.data)
    if rc != 0:
        raise ValueError(rc)
    return under[: cnt.value].astype(int).tolist(), part[:n]


def time_solves(reqs, guesses, repeats: int = 100, config: Optional[Config] = None,
                linsolve: int = LINSOLVE_SPARSE, analysis: bool = False) -> Tuple[float, int]:
    """CLI timing protocol (ezpz-cli/src/main.rs:86-100); analysis=True times solve_analysis instead (the reference's
    `*_analysis` benchmarks).  Returns (seconds for `repeats` solves, iterations)."""
    cfg = (config or Config())._c()
    a = stack(reqs)
    ids, vals = _split_guesses(guesses)
    it = C.c_uint64(0)
    fn = lib().orc_time_solves_analysis if analysis else lib().orc_time_solves
    secs = fn(a.ctypes.data, len(a), ids.ctypes.data, vals.ctypes.data, len(vals),
                                 C.byref(cfg), linsolve, repeats, C.byref(it))
    return secs, int(it.value)


def solve_batch(reqs, guesses_aos: np.ndarray, config: Optional[Config] = None, linsolve: int = LINSOLVE_DENSE,
                nthreads: int = 0):
    """Independent systems sharing one request list; guesses_aos [batch, n_vars]."""
    cfg = (config or Config())._c()
    a = stack(reqs)
    g = np.ascontiguousarray(guesses_aos, dtype=np.float64)
    batch, n = g.shape
    x_out = np.zeros_like(g)
    iters = np.zeros(batch, np.uint32)
    conv = np.zeros(batch, np.uint8)
    nun = np.zeros(batch, np.uint32)
    rc = lib().orc_solve_batch(a.ctypes.data, len(a), n, g.ctypes.data, batch, C.byref(cfg), linsolve, nthreads,
                               x_out.ctypes.data, iters.ctypes.data, conv.ctypes.data, nun.ctypes.data)
    return rc, x_out, iters, conv, nun
