"""Pins the CPU oracle to the reference's own known answers: every test of
/root/reference/ezpz/src/tests.rs that asserts a value at the solve() boundary, restated in
tests/cases.py, must pass on the oracle with BOTH of its linear solvers."""
import pytest

import cases
from adapters import OracleAdapter
from oracle import oracle as O


@pytest.mark.parametrize("linsolve", [O.LINSOLVE_DENSE, O.LINSOLVE_SPARSE], ids=["dense", "sparse"])
@pytest.mark.parametrize("case", cases.ALL_CASES, ids=[c.__name__ for c in cases.ALL_CASES])
def test_reference_known_answer(case, linsolve):
    case(OracleAdapter(linsolve))
