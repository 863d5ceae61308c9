"""`ezpz::textual` mirror (reference ezpz/src/textual.rs:43-49, textual/executor.rs:448-613): Problem.from_str,
Problem.to_constraint_system, ConstraintSystem.solve*.  Parsing and lowering run in the C++ front end
(ezpz_amd/csrc/textual.cpp) through `ezpz_problem_parse`."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import api
from ._lib import CONSTRAINT_DTYPE, lib


class TextualError(ValueError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


@dataclass
class Outcome:
    """textual/executor.rs:588-613"""

    unsatisfied: List[int]
    iterations: int
    warnings: list
    points: Dict[str, Tuple[float, float]]
    circles: Dict[str, dict]
    arcs: Dict[str, dict]
    num_vars: int
    num_eqs: int
    priority_solved: int
    converged: bool

    def get_point(self, label):
        return self.points.get(label)

    def get_circle(self, label):
        return self.circles.get(label)

    def get_arc(self, label):
        return self.arcs.get(label)


@dataclass
class ConstraintSystem:
    """textual/executor.rs:448-459"""

    records: np.ndarray
    guesses: np.ndarray
    inner_points: List[str]
    inner_circles: List[str]
    inner_arcs: List[str]

    @property
    def num_vars(self) -> int:
        return len(self.guesses)

    @property
    def constraints(self) -> np.ndarray:
        return self.records

    def variables(self) -> List[Tuple[int, float]]:
        return [(i, float(v)) for i, v in enumerate(self.guesses)]

    # label lookups, executor.rs:521-566
    def point(self, values, label):
        i = self.inner_points.index(label)
        return (float(values[2 * i]), float(values[2 * i + 1]))

    def circle(self, values, label):
        i = self.inner_circles.index(label)
        s = 2 * len(self.inner_points) + 3 * i
        return {"center": (float(values[s]), float(values[s + 1])), "radius": float(values[s + 2])}

    def arc(self, values, label):
        i = self.inner_arcs.index(label)
        s = 2 * len(self.inner_points) + 3 * len(self.inner_circles) + 6 * i
        return {"a": (float(values[s]), float(values[s + 1])), "b": (float(values[s + 2]), float(values[s + 3])),
                "center": (float(values[s + 4]), float(values[s + 5]))}

    def solve_no_metadata(self, config: Optional[api.Config] = None) -> api.SolveOutcome:
        raw = api.solve_records(self.records, self.variables(), config)
        if raw.error != 0:
            err = api.NonLinearSystemError(raw.error, raw.err_constraint_id, raw.err_variable)
            raise api.FailureOutcome(err, [api.Warning(a, c) for a, c in raw.warnings], raw.num_vars, raw.num_eqs)
        return api.SolveOutcome(raw)

    def solve_with_config(self, config: Optional[api.Config] = None) -> Outcome:
        o = self.solve_no_metadata(config)
        v = o.final_values()
        return Outcome(
            unsatisfied=o.unsatisfied(), iterations=o.iterations(), warnings=o.warnings(),
            points={l: self.point(v, l) for l in self.inner_points},
            circles={l: self.circle(v, l) for l in self.inner_circles},
            arcs={l: self.arc(v, l) for l in self.inner_arcs},
            num_vars=self.num_vars, num_eqs=o._raw.num_eqs, priority_solved=o.priority_solved(),
            converged=o.converged())

    def solve(self) -> Outcome:
        return self.solve_with_config(None)


class Problem:
    """textual.rs:30-49"""

    def __init__(self, system: ConstraintSystem):
        self._system = system

    @staticmethod
    def from_str(text: str) -> "Problem":
        data = text.encode("utf-8")
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().ezpz_problem_parse(data, len(data), C.byref(h), err, len(err))
        if rc != 0:
            raise TextualError(rc, err.value.decode() or lib().ezpz_error_string(rc).decode())
        try:
            L = lib()
            n_cs, n_vars = L.ezpz_problem_num_constraints(h), L.ezpz_problem_num_vars(h)
            recs = np.zeros(n_cs, dtype=CONSTRAINT_DTYPE)
            if n_cs:
                C.memmove(recs.ctypes.data, L.ezpz_problem_constraints(h), n_cs * CONSTRAINT_DTYPE.itemsize)
            guesses = np.zeros(n_vars)
            if n_vars:
                C.memmove(guesses.ctypes.data, L.ezpz_problem_guesses(h), n_vars * 8)
            labels = [[L.ezpz_problem_label(h, k, i).decode() for i in range(L.ezpz_problem_num_labels(h, k))]
                      for k in range(3)]
        finally:
            lib().ezpz_problem_destroy(h)
        return Problem(ConstraintSystem(recs, guesses, labels[0], labels[1], labels[2]))

    def to_constraint_system(self) -> ConstraintSystem:
        return self._system


def gen_big_problem(total_lines: int, overconstrain: bool = False) -> str:
    """The synthetic parallel-line problem of BASELINE configs[1] / [3]: the text
    test_cases/massive_parallel_system/gen_big_problem.py:16-35 prints for `total_lines` (and `true` for the
    over-constrained variant): per line two points, `vertical`, x of the first fixed to the line number, y fixed to 0
    and 4 (plus `distance(.., 4)` when over-constrained); guesses p_k roughly (k, k)."""
    out = ["# constraints"]
    for line in range(total_lines):
        a, b = 2 * line, 2 * line + 1
        out += [f"point p{a}", f"point p{b}", f"vertical(p{a}, p{b})", f"p{a}.x={line}", f"p{a}.y=0", f"p{b}.y=4"]
        if overconstrain:
            out.append(f"distance(p{a}, p{b}, 4)")
    out += ["", "# guesses"]
    for k in range(2 * total_lines):
        out.append(f"p{k} roughly ({k},{k})")
    return "\n".join(out) + "\n"
