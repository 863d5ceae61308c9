"""GPU test (-m gpu) of the on-disk code-object cache (jit.cpp): a second process finds the kernels the first one
compiled and runs on them from its first solves -- a CLI-style process (1 + 100 solves, ezpz-cli/src/main.rs:86-100)
never reaches the 256-solve threshold that starts a compilation."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
    import numpy as np
    import ezpz_amd as E
    from conftest import read_case
    mode = sys.argv[1]
    square = E.textual.Problem.from_str(read_case("square")).to_constraint_system()
    block = E.textual.Problem.from_str(E.textual.gen_big_problem(64)).to_constraint_system()
    for cs in (square, block):
        recs = E.resolve_sides(cs.records, cs.guesses)
        s = E.System(recs, cs.num_vars)
        x0 = np.tile(cs.guesses, (8, 1))
        if mode == "first":   # compiles (and stores) both kernels
            assert s.specialize(wait=True) == 2
            want = s.solve_batch(x0)[0]
        else:                 # nothing asks for a compilation here: 8 systems per call, a handful of calls
            t0 = time.perf_counter()
            want = s.solve_batch(x0)[0]          # the first launch asks the cache in the background
            for _ in range(200):
                if s.specialize(wait=False) == 2:
                    break
                time.sleep(0.001)
            ready_after = time.perf_counter() - t0
            assert s.specialize(wait=False) == 2 and ready_after < 0.25, ready_after   # (a compilation takes > 0.5 s)
            got = s.solve_batch(x0)[0]           # ... and this one runs on the specialised kernel
            assert np.all(np.abs(got - want) <= 1e-9 * np.maximum(1.0, np.abs(want)))
    print(mode, "ok")
""")


def test_second_process_starts_on_the_cached_kernels(tmp_path):
    env = dict(os.environ, EZPZ_JIT_CACHE_DIR=str(tmp_path / "jit"))
    for mode in ("first", "second"):
        r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}, mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and f"{mode} ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert len(os.listdir(tmp_path / "jit")) == 2
