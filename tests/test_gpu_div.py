"""The class-specialised kernels divide by a column's diagonal through its refined reciprocal (jit_kernel.hip.hpp:
recip_of / div_by) and fall back to plain divisions where the guard says an operand is out of range.  That is only
legitimate if the short form gives the correctly rounded quotient -- the bits of `n / D`, which is what the
interpreters, the list-walk kernels and the CPU oracle compute -- whenever the guard passes.  tests/div_exact.hip
sweeps 1.6e9 operand pairs per run (every exponent, denormals, zeros of both signs, infinities, NaN, exact quotients,
the guard's edges) and counts disagreements."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_short_division_is_the_correctly_rounded_quotient(tmp_path):
    exe = str(tmp_path / "div_exact")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-I", os.path.join(ROOT, "ezpz_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "div_exact.hip"), "-o", exe])
    rows = [list(map(int, line.split())) for line in subprocess.check_output([exe], text=True, timeout=600).splitlines()]
    assert [r[0] for r in rows] == [0, 1, 2]
    for mode, pairs, ok, mismatches, ok_zero, not_ok in rows:
        assert pairs == 8 * 4096 * 256 * 64 and ok + not_ok == pairs
        assert mismatches == 0, (mode, mismatches)
    # the sweeps exercise both sides of the guard: the whole-range sweep rejects the tiny / huge / infinite numerators and
    # the out-of-range denominators but keeps zeros; the everyday range never leaves the short path except for its
    # denominators outside [2^-40, 2^40]
    assert rows[0][2] > rows[0][1] // 4 and rows[0][5] > rows[0][1] // 4 and rows[0][4] > 0
    assert rows[1][2] > 0.75 * rows[1][1]
    assert rows[2][2] > 0 and rows[2][5] > 0
