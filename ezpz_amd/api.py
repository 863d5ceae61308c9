"""Host-side mirror of the `ezpz` crate's public solve API, on top of the C ABI (include/ezpz_amd.h).

Same names, argument meaning and error behaviour as the reference for the hot path:
    solve(reqs, initial_guesses, config)        reference ezpz/src/lib.rs:80-87
    Constraint / ConstraintRequest / Config     reference constraints.rs:37-93, constraint_request.rs, solver.rs:31-81
    DatumPoint / DatumLineSegment / ...         reference datatypes/inputs.rs
    SolveOutcome / FailureOutcome / Warning     reference solve_outcome.rs, error.rs, warnings.rs
`Result<SolveOutcome, FailureOutcome>` becomes "return SolveOutcome or raise FailureOutcome".
Everything numeric happens in the HIP kernel behind `ezpz_solve`; this module only packs records.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from ._lib import CONSTRAINT_DTYPE, STATUS_DTYPE, CConfig, COutcome, CSystemInfo, CWarning, lib

Id = int

# EzpzKind
(LINE_TANGENT_TO_CIRCLE, CIRCLE_TANGENT_TO_CIRCLE, DISTANCE, DISTANCE_VAR, VERTICAL_DISTANCE, HORIZONTAL_DISTANCE,
 VERTICAL, HORIZONTAL, LINES_AT_ANGLE, FIXED, SCALAR_EQUAL, POINTS_COINCIDENT, CIRCLE_RADIUS, LINES_EQUAL_LENGTH,
 ARC_RADIUS, ARC, MIDPOINT, POINT_LINE_DISTANCE, VERTICAL_POINT_LINE_DISTANCE, HORIZONTAL_POINT_LINE_DISTANCE,
 SYMMETRIC, POINT_ARC_COINCIDENT, ARC_LENGTH, ARC_ANGLE, POINTS_AT_ANGLE) = range(25)

KIND_NAMES = [
    "LineTangentToCircle", "CircleTangentToCircle", "Distance", "DistanceVar", "VerticalDistance",
    "HorizontalDistance", "Vertical", "Horizontal", "LinesAtAngle", "Fixed", "ScalarEqual", "PointsCoincident",
    "CircleRadius", "LinesEqualLength", "ArcRadius", "Arc", "Midpoint", "PointLineDistance",
    "VerticalPointLineDistance", "HorizontalPointLineDistance", "Symmetric", "PointArcCoincident", "ArcLength",
    "ArcAngle", "PointsAtAngle",
]

# error codes (EzpzError)
ERR_WRONG_NUMBER_GUESSES = -2
ERR_MISSING_GUESS = -3
ERR_MATRIX = -4
ERR_EMPTY_SYSTEM = -8
ERR_NO_DEVICE = -100


class IdGenerator:
    """id.rs:5-30"""

    def __init__(self):
        self._next = 0

    def next_id(self) -> Id:
        out = self._next
        self._next += 1
        return out


@dataclass(frozen=True)
class DatumDistance:
    id: Id

    @staticmethod
    def new(id: Id) -> "DatumDistance":
        return DatumDistance(id)


@dataclass(frozen=True)
class DatumPoint:
    x_id: Id
    y_id: Id

    @staticmethod
    def new(ids: IdGenerator) -> "DatumPoint":
        return DatumPoint(ids.next_id(), ids.next_id())

    @staticmethod
    def new_xy(x: Id, y: Id) -> "DatumPoint":
        return DatumPoint(x, y)

    def id_x(self) -> Id:
        return self.x_id

    def id_y(self) -> Id:
        return self.y_id

    def _ids(self):
        return [self.x_id, self.y_id]


@dataclass(frozen=True)
class DatumLineSegment:
    p0: DatumPoint
    p1: DatumPoint

    @staticmethod
    def new(p0: DatumPoint, p1: DatumPoint) -> "DatumLineSegment":
        return DatumLineSegment(p0, p1)

    def _ids(self):
        return self.p0._ids() + self.p1._ids()


@dataclass(frozen=True)
class DatumCircle:
    center: DatumPoint
    radius: DatumDistance

    def _ids(self):
        return self.center._ids() + [self.radius.id]


@dataclass(frozen=True)
class DatumCircularArc:
    center: DatumPoint
    start: DatumPoint
    end: DatumPoint

    def _ids(self):
        return self.center._ids() + self.start._ids() + self.end._ids()


@dataclass(frozen=True)
class Angle:
    """datatypes.rs:18-73"""

    val: float
    degrees: bool

    @staticmethod
    def from_degrees(v: float) -> "Angle":
        return Angle(float(v), True)

    @staticmethod
    def from_radians(v: float) -> "Angle":
        return Angle(float(v), False)

    def to_degrees(self) -> float:
        return self.val if self.degrees else self.val * (180.0 / math.pi)

    def to_radians(self) -> float:
        return self.val * (math.pi / 180.0) if self.degrees else self.val

    def __str__(self):
        return f"{self.val}deg" if self.degrees else f"{self.val}rad"


class AngleKind:
    """datatypes.rs:9-16"""

    Parallel = "Parallel"
    Perpendicular = "Perpendicular"

    @staticmethod
    def Other(angle: Angle):
        return ("Other", angle)

    @staticmethod
    def _encode(kind) -> Tuple[int, float]:
        if kind == AngleKind.Parallel:
            return 0, 0.0
        if kind == AngleKind.Perpendicular:
            return 1, 0.0
        _, angle = kind
        return (2 if angle.degrees else 3), angle.val


class LineSide:
    Undefined, Left, Right = 0, 1, 2


class CircleSide:
    Undefined, Exterior, Interior = 0, 1, 2


def _rec(kind: int, ids: Sequence[int], param: float = 0.0, tag: int = 0):
    r = np.zeros((), dtype=CONSTRAINT_DTYPE)
    r["kind"] = kind
    r["tag"] = tag
    r["param"] = param
    r["weight"] = 1.0
    a = np.zeros(8, np.uint32)
    a[: len(ids)] = np.asarray(list(ids), dtype=np.uint32)
    r["ids"] = a
    return r


class Constraint:
    """constraints.rs:37-93.  One classmethod per enum variant, same field order."""

    __slots__ = ("record",)

    def __init__(self, record):
        self.record = record

    @property
    def kind(self) -> int:
        return int(self.record["kind"])

    def constraint_kind(self) -> str:
        return KIND_NAMES[self.kind]

    def __repr__(self):
        n = [7, 6, 4, 5, 4, 4, 4, 4, 8, 1, 2, 4, 3, 8, 6, 6, 6, 6, 6, 6, 8, 8, 6, 6, 6][self.kind]
        return f"{self.constraint_kind()}(ids={self.record['ids'][:n].tolist()}, param={float(self.record['param'])}, tag={int(self.record['tag'])})"

    @staticmethod
    def LineTangentToCircle(line: DatumLineSegment, circle: DatumCircle, side=LineSide.Undefined):
        return Constraint(_rec(LINE_TANGENT_TO_CIRCLE, line._ids() + circle._ids(), tag=side))

    @staticmethod
    def CircleTangentToCircle(a: DatumCircle, b: DatumCircle, side=CircleSide.Undefined):
        return Constraint(_rec(CIRCLE_TANGENT_TO_CIRCLE, a._ids() + b._ids(), tag=side))

    @staticmethod
    def Distance(p0: DatumPoint, p1: DatumPoint, d: float):
        return Constraint(_rec(DISTANCE, p0._ids() + p1._ids(), d))

    @staticmethod
    def DistanceVar(p0: DatumPoint, p1: DatumPoint, d: DatumDistance):
        return Constraint(_rec(DISTANCE_VAR, p0._ids() + p1._ids() + [d.id]))

    @staticmethod
    def VerticalDistance(p0: DatumPoint, p1: DatumPoint, d: float):
        return Constraint(_rec(VERTICAL_DISTANCE, p0._ids() + p1._ids(), d))

    @staticmethod
    def HorizontalDistance(p0: DatumPoint, p1: DatumPoint, d: float):
        return Constraint(_rec(HORIZONTAL_DISTANCE, p0._ids() + p1._ids(), d))

    @staticmethod
    def Vertical(line: DatumLineSegment):
        return Constraint(_rec(VERTICAL, line._ids()))

    @staticmethod
    def Horizontal(line: DatumLineSegment):
        return Constraint(_rec(HORIZONTAL, line._ids()))

    @staticmethod
    def LinesAtAngle(l0: DatumLineSegment, l1: DatumLineSegment, kind):
        tag, val = AngleKind._encode(kind)
        return Constraint(_rec(LINES_AT_ANGLE, l0._ids() + l1._ids(), val, tag))

    @staticmethod
    def Fixed(id: Id, value: float):
        return Constraint(_rec(FIXED, [id], value))

    @staticmethod
    def ScalarEqual(a: Id, b: Id):
        return Constraint(_rec(SCALAR_EQUAL, [a, b]))

    @staticmethod
    def PointsCoincident(p0: DatumPoint, p1: DatumPoint):
        return Constraint(_rec(POINTS_COINCIDENT, p0._ids() + p1._ids()))

    @staticmethod
    def CircleRadius(circle: DatumCircle, r: float):
        return Constraint(_rec(CIRCLE_RADIUS, circle._ids(), r))

    @staticmethod
    def LinesEqualLength(l0: DatumLineSegment, l1: DatumLineSegment):
        return Constraint(_rec(LINES_EQUAL_LENGTH, l0._ids() + l1._ids()))

    @staticmethod
    def ArcRadius(arc: DatumCircularArc, r: float):
        return Constraint(_rec(ARC_RADIUS, arc._ids(), r))

    @staticmethod
    def Arc(arc: DatumCircularArc):
        return Constraint(_rec(ARC, arc._ids()))

    @staticmethod
    def Midpoint(line: DatumLineSegment, point: DatumPoint):
        return Constraint(_rec(MIDPOINT, line._ids() + point._ids()))

    @staticmethod
    def PointLineDistance(point: DatumPoint, line: DatumLineSegment, d: float):
        return Constraint(_rec(POINT_LINE_DISTANCE, point._ids() + line._ids(), d))

    @staticmethod
    def VerticalPointLineDistance(point: DatumPoint, line: DatumLineSegment, d: float):
        return Constraint(_rec(VERTICAL_POINT_LINE_DISTANCE, point._ids() + line._ids(), d))

    @staticmethod
    def HorizontalPointLineDistance(point: DatumPoint, line: DatumLineSegment, d: float):
        return Constraint(_rec(HORIZONTAL_POINT_LINE_DISTANCE, point._ids() + line._ids(), d))

    @staticmethod
    def Symmetric(line: DatumLineSegment, a: DatumPoint, b: DatumPoint):
        return Constraint(_rec(SYMMETRIC, line._ids() + a._ids() + b._ids()))

    @staticmethod
    def PointArcCoincident(arc: DatumCircularArc, point: DatumPoint):
        return Constraint(_rec(POINT_ARC_COINCIDENT, arc._ids() + point._ids()))

    @staticmethod
    def ArcLength(arc: DatumCircularArc, d: float):
        return Constraint(_rec(ARC_LENGTH, arc._ids(), d))

    @staticmethod
    def ArcAngle(arc: DatumCircularArc, angle: Angle):
        return Constraint(_rec(ARC_ANGLE, arc._ids(), angle.val, 2 if angle.degrees else 3))

    @staticmethod
    def PointsAtAngle(p0: DatumPoint, p1: DatumPoint, p2: DatumPoint, kind):
        tag, val = AngleKind._encode(kind)
        return Constraint(_rec(POINTS_AT_ANGLE, p0._ids() + p1._ids() + p2._ids(), val, tag))

    # constraints/composite.rs:9-62
    @staticmethod
    def lines_parallel(lines):
        return Constraint.LinesAtAngle(lines[0], lines[1], AngleKind.Parallel)

    @staticmethod
    def lines_perpendicular(lines):
        return Constraint.LinesAtAngle(lines[0], lines[1], AngleKind.Perpendicular)

    @staticmethod
    def point_bisects_arc(arc: DatumCircularArc, point: DatumPoint):
        return [Constraint.PointArcCoincident(arc, point),
                Constraint.Symmetric(DatumLineSegment(arc.center, point), arc.start, arc.end)]

    @staticmethod
    def parallel_lines_distance(lines, distance: float):
        return [Constraint.lines_parallel(lines), Constraint.PointLineDistance(lines[0].p0, lines[1], distance)]

    @staticmethod
    def circle_arc_coincident(circle: DatumCircle, arc: DatumCircularArc):
        return [Constraint.PointsCoincident(circle.center, arc.center),
                Constraint.LinesEqualLength(DatumLineSegment(arc.center, arc.start),
                                            DatumLineSegment(arc.center, arc.end))]


class ConstraintRequest:
    """constraint_request.rs:5-60"""

    __slots__ = ("_constraint", "_priority", "_weight")

    def __init__(self, constraint: Constraint, priority: int, weight: float = 1.0):
        self._constraint, self._priority, self._weight = constraint, int(priority), float(weight)

    @staticmethod
    def new(constraint: Constraint, priority: int) -> "ConstraintRequest":
        return ConstraintRequest(constraint, priority)

    @staticmethod
    def highest_priority(constraint: Constraint) -> "ConstraintRequest":
        return ConstraintRequest(constraint, 0)

    def with_weight(self, weight: float) -> "ConstraintRequest":
        return ConstraintRequest(self._constraint, self._priority, weight)

    def constraint(self) -> Constraint:
        return self._constraint

    def priority(self) -> int:
        return self._priority

    def weight(self) -> float:
        return self._weight

    def record(self):
        r = self._constraint.record.copy()
        r["priority"] = self._priority
        r["weight"] = self._weight
        return r


@dataclass(frozen=True)
class Config:
    """solver.rs:31-81"""

    max_iterations: int = 35
    residual_tolerance: float = 1e-8
    step_tolerance: float = 1e-12
    initial_lambda: float = 1e-9

    def with_max_iterations(self, v: int) -> "Config":
        return Config(int(v), self.residual_tolerance, self.step_tolerance, self.initial_lambda)

    def with_convergence_tolerance(self, v: float) -> "Config":
        return Config(self.max_iterations, float(v), self.step_tolerance, self.initial_lambda)

    def with_step_tolerance(self, v: float) -> "Config":
        return Config(self.max_iterations, self.residual_tolerance, float(v), self.initial_lambda)

    def with_initial_lambda(self, v: float) -> "Config":
        return Config(self.max_iterations, self.residual_tolerance, self.step_tolerance, float(v))

    def _c(self) -> CConfig:
        return CConfig(self.max_iterations, self.residual_tolerance, self.step_tolerance, self.initial_lambda)


class WarningContent:
    Degenerate, ShouldBeParallel, ShouldBePerpendicular = 0, 1, 2


@dataclass(frozen=True)
class Warning:
    about_constraint: Optional[int]
    content: int


@dataclass
class RawResult:
    """Flat result of ezpz_solve (EzpzOutcome + buffers); what tests/cases.py consumes."""

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
    final_lambda: float
    final_residual_inf: float
    underconstrained: Optional[List[int]] = None  # FreedomAnalysis; None = not requested


class NonLinearSystemError(Exception):
    """error.rs:35-86"""

    def __init__(self, code: int, constraint_id: int = -1, variable: int = -1):
        self.code, self.constraint_id, self.variable = code, constraint_id, variable
        msg = lib().ezpz_error_string(code).decode()
        if code == ERR_MISSING_GUESS:
            msg = (f"Constraint {constraint_id} references variable {variable} but no such variable appears in "
                   "your initial guesses.")
        super().__init__(msg)


class FailureOutcome(Exception):
    """solve_outcome.rs:126-136"""

    def __init__(self, error: NonLinearSystemError, warnings, num_vars, num_eqs):
        super().__init__(str(error))
        self.error, self.warnings, self.num_vars, self.num_eqs = error, warnings, num_vars, num_eqs


class SolveOutcome:
    """solve_outcome.rs:12-100"""

    def __init__(self, raw: RawResult):
        self._raw = raw

    def unsatisfied(self) -> List[int]:
        return self._raw.unsatisfied

    def converged(self) -> bool:
        return self._raw.converged

    def final_values(self) -> np.ndarray:
        return self._raw.final_values

    def iterations(self) -> int:
        return self._raw.iterations

    def warnings(self) -> List[Warning]:
        return [Warning(a, c) for a, c in self._raw.warnings]

    def priority_solved(self) -> int:
        return self._raw.priority_solved

    def final_value_distance(self, d: DatumDistance) -> float:
        return float(self._raw.final_values[d.id])

    def final_value_point(self, p: DatumPoint):
        return (float(self._raw.final_values[p.x_id]), float(self._raw.final_values[p.y_id]))

    def final_value_arc(self, arc: DatumCircularArc):
        return {"a": self.final_value_point(arc.start), "b": self.final_value_point(arc.end),
                "center": self.final_value_point(arc.center)}

    def final_value_circle(self, c: DatumCircle):
        return {"center": self.final_value_point(c.center), "radius": self.final_value_distance(c.radius)}

    def is_satisfied(self) -> bool:
        return not self._raw.unsatisfied

    def is_unsatisfied(self) -> bool:
        return bool(self._raw.unsatisfied)


def stack_records(records) -> np.ndarray:
    if isinstance(records, np.ndarray) and records.dtype == CONSTRAINT_DTYPE:
        return np.ascontiguousarray(records).reshape(-1)
    out = np.zeros(len(records), dtype=CONSTRAINT_DTYPE)
    for i, r in enumerate(records):
        out[i] = r
    return out


def _split_guesses(guesses):
    if isinstance(guesses, np.ndarray) and guesses.ndim == 1 and guesses.dtype.kind == "f":
        ids = np.arange(len(guesses), dtype=np.uint32)
        vals = np.ascontiguousarray(guesses, dtype=np.float64)
    else:
        ids = np.ascontiguousarray([g[0] for g in guesses], dtype=np.uint32)
        vals = np.ascontiguousarray([g[1] for g in guesses], dtype=np.float64)
    return ids, vals


def solve_records(records, guesses, config: Optional[Config] = None, warn_cap: int = 4096,
                  analysis: bool = False) -> RawResult:
    """`ezpz_solve` (or `ezpz_solve_analysis`) on a record array; never raises for solver errors (error code in the
    result)."""
    a = stack_records(records)
    ids, vals = _split_guesses(guesses)
    n = len(vals)
    cfg = (config or Config())._c()
    x_out = np.zeros(max(n, 1))
    unsat = np.zeros(max(len(a), 1), dtype=np.uint64)
    warns = (CWarning * max(warn_cap, 1))()
    out = COutcome()
    args = (a.ctypes.data if len(a) else None, len(a), ids.ctypes.data if n else None,
            vals.ctypes.data if n else None, n, C.byref(cfg), x_out.ctypes.data, unsat.ctypes.data,
            C.cast(warns, C.c_void_p), warn_cap, C.byref(out))
    under = np.zeros(max(n, 1), dtype=np.uint32)
    n_under = C.c_uint64(0)
    if analysis:
        lib().ezpz_solve_analysis(*args, under.ctypes.data, C.byref(n_under))
    else:
        lib().ezpz_solve(*args)
    nw = min(int(out.n_warnings), warn_cap)
    return RawResult(
        underconstrained=under[: n_under.value].astype(int).tolist() if analysis else None,
        error=out.error, err_constraint_id=out.err_constraint_id, err_variable=out.err_variable,
        final_values=x_out[:n].copy(), iterations=int(out.iterations), converged=bool(out.converged),
        unsatisfied=unsat[: int(out.n_unsatisfied)].astype(int).tolist(),
        warnings=[(warns[i].about_constraint, warns[i].content) for i in range(nw)],
        priority_solved=int(out.priority_solved), num_vars=int(out.num_vars), num_eqs=int(out.num_eqs),
        final_lambda=out.final_lambda, final_residual_inf=out.final_residual_inf)


def solve(reqs: Iterable[ConstraintRequest], initial_guesses: Sequence[Tuple[Id, float]],
          config: Optional[Config] = None) -> SolveOutcome:
    """`ezpz::solve` (lib.rs:80-87): returns a SolveOutcome or raises FailureOutcome."""
    raw = solve_records([r.record() for r in reqs], list(initial_guesses), config)
    if raw.error != 0:
        err = NonLinearSystemError(raw.error, raw.err_constraint_id, raw.err_variable)
        raise FailureOutcome(err, [Warning(a, c) for a, c in raw.warnings], raw.num_vars, raw.num_eqs)
    return SolveOutcome(raw)


class FreedomAnalysis:
    """analysis.rs:24-77"""

    def __init__(self, underconstrained: List[int]):
        self._under = list(underconstrained)

    def is_underconstrained(self) -> bool:
        return bool(self._under)

    def underconstrained(self) -> List[int]:
        return self._under

    def into_underconstrained(self) -> List[int]:
        return self._under


class SolveOutcomeFreedomAnalysis:
    """solve_outcome.rs: `analysis` + `outcome`."""

    def __init__(self, raw: RawResult):
        self.analysis = FreedomAnalysis(raw.underconstrained or [])
        self.outcome = SolveOutcome(raw)


def solve_analysis(reqs: Iterable[ConstraintRequest], initial_guesses: Sequence[Tuple[Id, float]],
                   config: Optional[Config] = None) -> SolveOutcomeFreedomAnalysis:
    """`ezpz::solve_analysis` (lib.rs:134-146)."""
    raw = solve_records([r.record() for r in reqs], list(initial_guesses), config, analysis=True)
    if raw.error != 0:
        err = NonLinearSystemError(raw.error, raw.err_constraint_id, raw.err_variable)
        raise FailureOutcome(err, [Warning(a, c) for a, c in raw.warnings], raw.num_vars, raw.num_eqs)
    return SolveOutcomeFreedomAnalysis(raw)


def solve_batch(reqs, x0: np.ndarray, config: Optional[Config] = None, want_mask: bool = False):
    """`ezpz::solve` semantics (side inference, priority tiers) for a batch of guess vectors sharing one request list.

    reqs: ConstraintRequests or 56-byte records; x0: [batch, n_vars].  Returns (x, status, priority_solved, mask|None)."""
    recs = stack_records([r.record() if isinstance(r, ConstraintRequest) else r for r in reqs]
                         if not isinstance(reqs, np.ndarray) else reqs)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    batch, n = x0.shape
    cfg = (config or Config())._c()
    x = np.empty_like(x0)
    st = np.zeros(batch, dtype=STATUS_DTYPE)
    prio = np.zeros(batch, dtype=np.uint32)
    mask = np.zeros((batch, max(len(recs), 1)), dtype=np.uint8) if want_mask else None
    ec, ev = C.c_int32(-1), C.c_int64(-1)
    rc = lib().ezpz_solve_batch(recs.ctypes.data if len(recs) else None, len(recs), n, x0.ctypes.data, batch,
                                C.byref(cfg), x.ctypes.data, st.ctypes.data, prio.ctypes.data,
                                mask.ctypes.data if want_mask else None, C.byref(ec), C.byref(ev))
    if rc != 0:
        raise NonLinearSystemError(rc, ec.value, ev.value)
    return x, st, prio, mask


def specialized_source(records, n_vars: int, compile=False, wave=False) -> str:
    """`ezpz_specialized_source`: the generated source of the request's class-specialised kernel ('' = none); with
    compile=True it is also compiled for gfx950 (hiprtc; no device needed) and a failure raises with the log;
    compile="cached": through the on-disk cache of code objects, like the solve entry points."""
    recs = stack_records(records)
    buf = C.create_string_buffer(1 << 22)
    rc = lib().ezpz_specialized_source(recs.ctypes.data if len(recs) else None, len(recs), int(n_vars),
                                       (2 if compile == "cached" else 1 if compile else 0) | (4 if wave else 0), buf, len(buf))
    if rc < 0:
        raise RuntimeError("run-time compilation failed:\n" + buf.value.decode(errors="replace")[-4000:])
    return buf.value.decode() if rc > 0 else ""


def host_register(array: np.ndarray) -> None:
    """`ezpz_host_register`: page-locks a (contiguous) array the caller will pass to batch calls again and again."""
    rc = lib().ezpz_host_register(array.ctypes.data, array.nbytes)
    if rc != 0:
        raise NonLinearSystemError(rc)


def host_unregister(array: np.ndarray) -> None:
    rc = lib().ezpz_host_unregister(array.ctypes.data)
    if rc != 0:
        raise NonLinearSystemError(rc)


def resolve_sides(records, values) -> np.ndarray:
    """`Constraint::set_from_initial_values` (constraints.rs:146-193) over a request list: a copy of `records` in which
    every undefined LineSide / CircleSide is the one `values` (by id) imply.  `System` takes side-resolved records."""
    recs = stack_records(records).copy()
    vals = np.ascontiguousarray(values, dtype=np.float64)
    rc = lib().ezpz_resolve_sides(recs.ctypes.data if len(recs) else None, len(recs),
                                  vals.ctypes.data if len(vals) else None, len(vals))
    if rc != 0:
        raise NonLinearSystemError(rc)
    return recs


TEAM_AUTO_LISTS = 0xFFFFFFFE  # `team_size`: automatic, list-walk shapes only (never component-resident)
TEAM_BATCH_LANES = 0xFFFFFFFD  # `team_size`: automatic, connected sketches one lane per system at every batch size
TEAM_LATENCY_PHASES = 0xFFFFFFFC  # `team_size`: TEAM_AUTO_LATENCY without the record walk (dense phases instead)
TEAM_LATENCY_WAVE = 0xFFFFFFFB  # `team_size`: TEAM_AUTO_LATENCY, a small system always on one wavefront per system where that form exists
TEAM_LATENCY_RECORDS = 0xFFFFFFF9  # `team_size`: TEAM_AUTO_LATENCY without the frontal shape (the record walk, as before round 5)
TEAM_FRONTS = 0xFFFFFFFA  # `team_size`: the frontal shape (team_mode 5) whatever the size of the system, TEAM_AUTO_LATENCY behind it
TEAM_AUTO_LATENCY = 0xFFFFFFFF  # `team_size`: choose for the latency of one solve instead of batch throughput (ezpz_amd.h)


class System:
    """One analysed topology resident on a device: `ezpz_system_create` + batched solves.

    This is the batch mode that has no counterpart in the reference (it solves one system per call);
    every system of a batch shares the constraint list and differs only in its initial guesses.
    """

    def __init__(self, records, n_vars: int, device: int = 0, team_size: int = 0):
        self.records = stack_records(records)
        self.n_vars = int(n_vars)
        self.device = device
        h = C.c_void_p()
        ec, ev = C.c_int32(-1), C.c_int64(-1)
        rc = lib().ezpz_system_create(self.records.ctypes.data if len(self.records) else None, len(self.records),
                                      self.n_vars, device, team_size, C.byref(h), C.byref(ec), C.byref(ev))
        if rc != 0:
            raise NonLinearSystemError(rc, ec.value, ev.value)
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # (at interpreter shutdown the module's globals may be gone already)
            lib().ezpz_system_destroy(h)
            self._h = None

    def specialize(self, wait: bool = True) -> int:
        """`ezpz_system_specialize`: 2 = the class-specialised kernel is ready, 1 = compiling, 0 = none for this system."""
        rc = lib().ezpz_system_specialize(self._h, 1 if wait else 0)
        if rc < 0:
            raise NonLinearSystemError(rc)
        return rc

    def info(self) -> dict:
        i = CSystemInfo()
        lib().ezpz_system_info(self._h, C.byref(i))
        return {f: getattr(i, f) for f, _ in CSystemInfo._fields_}

    def solve_batch_logged(self, x0: np.ndarray, config: Optional[Config] = None, warn_cap: int = 1024):
        """solve_batch plus the Degenerate warnings of every system: a list per system of (pass, constraint position)
        in the reference's chronological order (the kernel's log entries sorted), truncated to warn_cap."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, max(self.n_vars, 1))
        batch = x0.shape[0]
        cfg = (config or Config())._c()
        x = np.empty_like(x0)
        st = np.zeros(batch, dtype=STATUS_DTYPE)
        log = np.zeros((batch, warn_cap), dtype=np.uint64)
        rc = lib().ezpz_system_solve_batch(self._h, x0.ctypes.data, batch, C.byref(cfg), x.ctypes.data, st.ctypes.data,
                                           None, log.ctypes.data, warn_cap)
        if rc != 0:
            raise NonLinearSystemError(rc)
        logs = []
        for b in range(batch):
            e = np.sort(log[b, : min(int(st["n_warnings"][b]), warn_cap)])
            logs.append([(int(v >> np.uint64(32)), int(v & np.uint64(0xFFFFFFFF))) for v in e])
        return x, st, logs

    def solve_batch(self, x0: np.ndarray, config: Optional[Config] = None, want_mask: bool = False):
        """Host arrays in/out.  x0 [batch, n_vars] -> (x [batch, n_vars], status (STATUS_DTYPE), mask or None)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, max(self.n_vars, 1))
        batch = x0.shape[0]
        cfg = (config or Config())._c()
        x = np.empty_like(x0)
        st = np.zeros(batch, dtype=STATUS_DTYPE)
        mask = np.zeros((batch, max(len(self.records), 1)), dtype=np.uint8) if want_mask else None
        rc = lib().ezpz_system_solve_batch(self._h, x0.ctypes.data, batch, C.byref(cfg), x.ctypes.data,
                                           st.ctypes.data, mask.ctypes.data if want_mask else None, None, 0)
        if rc != 0:
            raise NonLinearSystemError(rc)
        return x, st, mask

    def eval_batch(self, x: np.ndarray):
        """Residual + Jacobian sweep only.  Returns (r [batch, m], dense J [batch, m, n], degenerate counts)."""
        info = self.info()
        m, zj, n = info["n_rows"], info["nnz_j"], self.n_vars
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, max(n, 1))
        batch = x.shape[0]
        r = np.zeros((batch, max(m, 1)))
        jv = np.zeros((batch, max(zj, 1)))
        deg = np.zeros(batch, np.uint32)
        rc = lib().ezpz_system_eval_batch(self._h, x.ctypes.data, batch, r.ctypes.data, jv.ctypes.data,
                                          deg.ctypes.data)
        if rc != 0:
            raise NonLinearSystemError(rc)
        rows, cols = np.zeros(max(zj, 1), np.uint32), np.zeros(max(zj, 1), np.uint32)
        lib().ezpz_system_jacobian_pattern(self._h, rows.ctypes.data, cols.ctypes.data)
        J = np.zeros((batch, m, n))
        J[:, rows[:zj], cols[:zj]] = jv[:, :zj]
        return r[:, :m], J, deg

    def freedom_batch(self, x: np.ndarray):
        """FreedomAnalysis (find_dof.rs) of each value vector.  Returns (mask [batch, n] uint8, participation)."""
        n = self.n_vars
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, max(n, 1))
        batch = x.shape[0]
        mask = np.zeros((batch, max(n, 1)), np.uint8)
        part = np.zeros((batch, max(n, 1)))
        rc = lib().ezpz_system_freedom_batch(self._h, x.ctypes.data, batch, mask.ctypes.data, part.ctypes.data)
        if rc != 0:
            raise NonLinearSystemError(rc)
        return mask, part

    def freedom_batch_device(self, x_ptr: int, batch: int, mask_ptr: int, part_ptr: int = 0, count_ptr: int = 0,
                             stream: int = 0) -> None:
        rc = lib().ezpz_system_freedom_batch_device(self._h, x_ptr, batch, mask_ptr, part_ptr or None,
                                                    count_ptr or None, stream or None)
        if rc != 0:
            raise NonLinearSystemError(rc)

    def solve_batch_device(self, x0_ptr: int, batch: int, x_out_ptr: int, status_ptr: int, mask_ptr: int = 0,
                           stream: int = 0, config: Optional[Config] = None) -> None:
        """Device pointers (e.g. torch tensors' data_ptr()) and a hipStream_t handle; enqueue only."""
        cfg = (config or Config())._c()
        rc = lib().ezpz_system_solve_batch_device(self._h, x0_ptr, batch, C.byref(cfg), x_out_ptr, status_ptr,
                                                  mask_ptr or None, None, 0, stream or None)
        if rc != 0:
            raise NonLinearSystemError(rc)


class MultiSystem:
    """One analysed topology resident on several devices of the node (`ezpz_multi_*`): a batch is sharded contiguously
    over them, one host worker thread per device, every shard over its own device's host link.  `device_mask` bit d =
    HIP device d, 0 = all."""

    def __init__(self, records, n_vars: int, device_mask: int = 0, team_size: int = 0):
        self.records = stack_records(records)
        self.n_vars = int(n_vars)
        h = C.c_void_p()
        ec, ev = C.c_int32(-1), C.c_int64(-1)
        rc = lib().ezpz_multi_create(self.records.ctypes.data if len(self.records) else None, len(self.records), self.n_vars,
                                     device_mask, team_size, C.byref(h), C.byref(ec), C.byref(ev))
        if rc != 0:
            raise NonLinearSystemError(rc, ec.value, ev.value)
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib().ezpz_multi_destroy(h)
            self._h = None

    def devices(self):
        return [lib().ezpz_multi_device(self._h, i) for i in range(lib().ezpz_multi_device_count(self._h))]

    def shard(self, batch: int, index: int):
        first, count = C.c_size_t(0), C.c_size_t(0)
        lib().ezpz_multi_shard(self._h, batch, index, C.byref(first), C.byref(count))
        return first.value, count.value

    def specialize(self, wait: bool = True) -> int:
        rc = lib().ezpz_multi_specialize(self._h, 1 if wait else 0)
        if rc < 0:
            raise NonLinearSystemError(rc)
        return rc

    def solve_batch(self, x0: np.ndarray, config: Optional[Config] = None, want_mask: bool = False, out=None):
        """x0 [batch, n_vars] -> (x, status, mask or None).  `out` = (x, status) arrays to fill (e.g. registered ones)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, max(self.n_vars, 1))
        batch = x0.shape[0]
        cfg = (config or Config())._c()
        x, st = out if out is not None else (np.empty_like(x0), np.zeros(batch, dtype=STATUS_DTYPE))
        mask = np.zeros((batch, max(len(self.records), 1)), dtype=np.uint8) if want_mask else None
        rc = lib().ezpz_multi_solve_batch(self._h, x0.ctypes.data, batch, C.byref(cfg), x.ctypes.data, st.ctypes.data,
                                          mask.ctypes.data if want_mask else None)
        if rc != 0:
            raise NonLinearSystemError(rc)
        return x, st, mask


class MixedBatch:
    """One batch of systems of DIFFERENT topologies (`ezpz_mixed_*`): system b has the topology
    `systems[topology_of_system[b]]`; x0 / x are RAGGED -- the systems' rows one after the other in batch order,
    `offsets[b]` = where system b's values start (`offsets[-1]` = the total).  Reusable for any number of solves."""

    def __init__(self, systems, topology_of_system):
        self.systems = list(systems)  # (kept alive: the handle refers to them)
        self.topology = np.ascontiguousarray(topology_of_system, dtype=np.uint32)
        self.batch = len(self.topology)
        handles = (C.c_void_p * len(self.systems))(*[s._h for s in self.systems])
        h = C.c_void_p()
        rc = lib().ezpz_mixed_create(handles, len(self.systems), self.topology.ctypes.data, self.batch, C.byref(h))
        if rc != 0:
            raise NonLinearSystemError(rc)
        self._h = h
        self.offsets = np.zeros(self.batch + 1, dtype=np.uint64)
        lib().ezpz_mixed_offsets(self._h, self.offsets.ctypes.data)
        self.total = int(lib().ezpz_mixed_total_values(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib().ezpz_mixed_destroy(h)
            self._h = None

    def solve(self, x0: np.ndarray, config: Optional[Config] = None):
        """x0: flat ragged array of `total` values -> (x flat, status [batch])."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
        assert x0.size == self.total, (x0.size, self.total)
        cfg = (config or Config())._c()
        x, st = np.empty_like(x0), np.zeros(self.batch, dtype=STATUS_DTYPE)
        rc = lib().ezpz_mixed_solve(self._h, x0.ctypes.data, C.byref(cfg), x.ctypes.data, st.ctypes.data)
        if rc != 0:
            raise NonLinearSystemError(rc)
        return x, st

    def solve_device(self, x0_ptr: int, x_out_ptr: int, status_ptr: int, stream: int = 0, config: Optional[Config] = None):
        """Device pointers (e.g. torch tensors' data_ptr()); only enqueues on `stream`."""
        cfg = (config or Config())._c()
        rc = lib().ezpz_mixed_solve_device(self._h, x0_ptr, C.byref(cfg), x_out_ptr, status_ptr, stream)
        if rc != 0:
            raise NonLinearSystemError(rc)


def solve_batch_mixed(systems, topology_of_system, x0: np.ndarray, config: Optional[Config] = None):
    """`ezpz_system_solve_batch_mixed`: the one-call form."""
    topo = np.ascontiguousarray(topology_of_system, dtype=np.uint32)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
    handles = (C.c_void_p * len(systems))(*[s._h for s in systems])
    cfg = (config or Config())._c()
    x, st = np.empty_like(x0), np.zeros(len(topo), dtype=STATUS_DTYPE)
    rc = lib().ezpz_system_solve_batch_mixed(handles, len(systems), topo.ctypes.data, x0.ctypes.data, len(topo), C.byref(cfg),
                                             x.ctypes.data, st.ctypes.data)
    if rc != 0:
        raise NonLinearSystemError(rc)
    return x, st


def solve_batch_mixed_multi(multis, topology_of_system, x0: np.ndarray, config: Optional[Config] = None):
    """`ezpz_multi_solve_batch_mixed`: the ragged batch sharded over the devices the MultiSystems share."""
    topo = np.ascontiguousarray(topology_of_system, dtype=np.uint32)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1)
    handles = (C.c_void_p * len(multis))(*[m._h for m in multis])
    cfg = (config or Config())._c()
    x, st = np.empty_like(x0), np.zeros(len(topo), dtype=STATUS_DTYPE)
    rc = lib().ezpz_multi_solve_batch_mixed(handles, len(multis), topo.ctypes.data, x0.ctypes.data, len(topo), C.byref(cfg),
                                            x.ctypes.data, st.ctypes.data)
    if rc != 0:
        raise NonLinearSystemError(rc)
    return x, st


def solve_batch_multi(records, n_vars: int, x0: np.ndarray, device_mask: int = 0, config: Optional[Config] = None):
    """`ezpz_system_solve_batch_multi`: the one-call form (handles cached by request bytes and mask)."""
    recs = stack_records(records)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, max(int(n_vars), 1))
    cfg = (config or Config())._c()
    x = np.empty_like(x0)
    st = np.zeros(x0.shape[0], dtype=STATUS_DTYPE)
    rc = lib().ezpz_system_solve_batch_multi(recs.ctypes.data if len(recs) else None, len(recs), int(n_vars), device_mask,
                                             x0.ctypes.data, x0.shape[0], C.byref(cfg), x.ctypes.data, st.ctypes.data)
    if rc != 0:
        raise NonLinearSystemError(rc)
    return x, st


def analyze(records, n_vars: int) -> dict:
    """Host-only symbolic analysis (`ezpz_analyze`): sizes of the topology program, no device needed."""
    a = stack_records(records)
    i = CSystemInfo()
    ec, ev = C.c_int32(-1), C.c_int64(-1)
    rc = lib().ezpz_analyze(a.ctypes.data if len(a) else None, len(a), int(n_vars), C.byref(i), C.byref(ec),
                            C.byref(ev))
    if rc != 0:
        raise NonLinearSystemError(rc, ec.value, ev.value)
    return {f: getattr(i, f) for f, _ in CSystemInfo._fields_}


def device_count() -> int:
    return lib().ezpz_device_count()


def launch_policy(compute_units: int = 0):
    """The table of launch-shape thresholds (include/ezpz_amd.h: EzpzLaunchPolicy) for a device of `compute_units` CUs
    (0 = the full 256-CU MI355X, as the header says -- NOT the current device: a caller on a CPX/DPX partition or a CU-masked
    device passes its own count, e.g. torch.cuda.get_device_properties(i).multi_processor_count)."""
    from ._lib import CLaunchPolicy

    p = CLaunchPolicy()
    rc = lib().ezpz_launch_policy(int(compute_units), C.byref(p))
    if rc != 0:
        raise NonLinearSystemError(rc, -1, -1)
    return p
