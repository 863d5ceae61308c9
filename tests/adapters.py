"""Solver adapters for tests/cases.py: one for the CPU oracle, one for the HIP path (C ABI)."""
import os

from oracle import oracle as O
from oracle import textual as T

from conftest import read_case


class OracleAdapter:
    name = "oracle"

    def __init__(self, linsolve=O.LINSOLVE_DENSE):
        self.linsolve = linsolve

    def solve(self, reqs, guesses, config=None):
        cfg = O.Config(**(config or {}))
        return O.solve(list(reqs), guesses, cfg, linsolve=self.linsolve, analysis=True)

    def run_text(self, text, config=None):
        cs = T.load(text)
        return self.solve(cs.constraints, cs.variables(), config), cs

    def run(self, case, filename="problem.md", config=None):
        return self.run_text(read_case(case, filename), config)


class GpuAdapter:
    """HIP path through the C ABI (libezpz_amd.so); text goes through the product's own C++ parser."""

    name = "gpu"

    def __init__(self):
        import ezpz_amd

        self.E = ezpz_amd

    def solve(self, reqs, guesses, config=None):
        E = self.E
        cfg = E.Config(**(config or {}))
        return E.solve_records(O.stack(list(reqs)), guesses, cfg, analysis=True)

    def run_text(self, text, config=None):
        E = self.E
        system = E.textual.Problem.from_str(text).to_constraint_system()
        cfg = E.Config(**(config or {}))
        return E.solve_records(system.records, system.variables(), cfg, analysis=True), system

    def run(self, case, filename="problem.md", config=None):
        return self.run_text(read_case(case, filename), config)
