"""The reference's own criterion benchmarks (ezpz/benches/solver_bench.rs) as a latency table: one `solve()` /
`solve_analysis()` call per iteration, exactly what each benchmark body times, on the HIP path (first 256 solves of a
process = the interpreting kernels; steady state = what criterion measures, the topology's specialised kernel once it is
compiled or found in the on-disk cache; cold = cache cleared before every call: symbolic phase included) beside the 1-core CPU port (the oracle, per-call setup like the reference), iteration counts compared.

usage (GPU box): python tools/reference_benches.py > profiles/r03_reference_benches.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

import ezpz_amd as E  # noqa: E402
from ezpz_amd._lib import COutcome  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import textual as T  # noqa: E402


def case(name):
    text = open(os.path.join(ROOT, "tests", "golden", "test_cases", name, "problem.md")).read()
    ref = T.load(text)
    return ref.constraints, ref.variables()


def two_rectangles_dependent():
    """solver_bench.rs:61-143"""
    pt = lambda i: (2 * i, 2 * i + 1)
    p0, p1, p2, p3, p5, p6, p7 = (pt(i) for i in range(7))
    cons = [O.fixed(p0[0], 1.0), O.fixed(p0[1], 1.0), O.horizontal(p0, p1), O.horizontal(p2, p3), O.vertical(p3, p0),
            O.vertical(p1, p2), O.distance(p0, p1, 4.0), O.distance(p0, p3, 3.0),
            O.horizontal(p2, p5), O.horizontal(p6, p7), O.vertical(p7, p2), O.vertical(p5, p6), O.distance(p2, p5, 4.0),
            O.distance(p2, p7, 4.0)]
    vals = [1.0, 1.0, 4.5, 1.5, 4.0, 3.5, 1.5, 3.0, 5.5, 3.5, 5.0, 4.5, 2.5, 4.0]
    return cons, list(enumerate(vals))


def massive(lines):
    ref = T.load(T.gen_big_problem(lines))
    return ref.constraints, ref.variables()


BENCHES = [  # (criterion id, solver_bench.rs lines, builder, analysis?)
    ("solve_inconsistent", "43-45", lambda: case("inconsistent"), False),
    ("solve_two_rectangles", "47-49", lambda: case("two_rectangles"), False),
    ("solve_nonsquare", "51-53", lambda: case("nonsquare"), False),
    ("solve_nonsquare_analysis", "55-57", lambda: case("nonsquare"), True),
    ("solve two rectangles dependent", "61-143", two_rectangles_dependent, False),
    ("massively_parallel/200", "175-201", lambda: massive(200), False),
    ("massively_parallel/600", "175-201", lambda: massive(600), False),
    ("massively_parallel_analysis/200", "149-173", lambda: massive(200), True),
]


def main():
    print("# python tools/reference_benches.py  -- ezpz/benches/solver_bench.rs, one solve() per iteration")
    print("# benchmark | rows x vars | iterations (HIP = CPU port?) | HIP us, first 256 solves (interpreting kernels) | HIP us, steady state (specialised kernel where the topology has one) | HIP cold us | CPU port 1 core us | steady-state speed-up")
    for name, where, build, analysis in BENCHES:
        reqs, guesses = build()
        recs = O.stack(reqs)
        got = E.solve_records(recs, guesses, analysis=analysis)
        # the timed call is the C ABI itself on buffers prepared once (what a host-language caller does; the Python
        # object layer around it costs more than a small solve)
        ids = np.ascontiguousarray([g[0] for g in guesses], dtype=np.uint32)
        vals = np.ascontiguousarray([g[1] for g in guesses], dtype=np.float64)
        n = len(vals)
        cfg, out = E.Config()._c(), COutcome()
        x_out, unsat, under, n_under = np.zeros(n), np.zeros(len(recs) + 1, dtype=np.uint64), np.zeros(n, dtype=np.uint32), C.c_uint64(0)
        args = [recs.ctypes.data, len(recs), ids.ctypes.data, vals.ctypes.data, n, C.byref(cfg), x_out.ctypes.data, unsat.ctypes.data,
                None, 0, C.byref(out)]
        fn = E.lib().ezpz_solve_analysis if analysis else E.lib().ezpz_solve
        if analysis:
            args += [under.ctypes.data, C.byref(n_under)]
        call = lambda: fn(*args)
        assert call() == 0 and out.iterations == got.iterations
        want = O.solve(reqs, guesses, linsolve=O.LINSOLVE_SPARSE, analysis=analysis)
        assert got.error == 0 and want.error == 0
        same = got.iterations == want.iterations and got.converged == want.converged and got.unsatisfied == want.unsatisfied and \
            (not analysis or list(got.underconstrained) == list(want.underconstrained))
        reps = 200 if len(guesses) < 100 else 100
        E.lib().ezpz_cache_clear()
        for _ in range(20):
            call()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        early = (time.perf_counter() - t0) / reps * 1e6  # a process's first 256 solves of a topology: the interpreting kernels
        for _ in range(300):  # ... past 256 solves the topology's kernel is compiled in the background (or comes from the on-disk cache)
            call()
        time.sleep(1.5)
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        warm = (time.perf_counter() - t0) / reps * 1e6
        cold = 0.0
        for _ in range(5):
            E.lib().ezpz_cache_clear()
            t0 = time.perf_counter()
            call()
            cold += (time.perf_counter() - t0) / 5 * 1e6
        secs, _ = O.time_solves(reqs, guesses, repeats=reps, linsolve=O.LINSOLVE_SPARSE, analysis=analysis)
        cpu = secs / reps * 1e6
        print(f"{name} (solver_bench.rs:{where}) | {got.num_eqs} x {got.num_vars} | {got.iterations} ({'equal' if same else 'DIFFERENT: ' + str(want.iterations)}) | "
              f"{early:.1f} | {warm:.1f} | {cold:.1f} | {cpu:.1f} | {cpu / warm:.2f}x")


if __name__ == "__main__":
    main()
