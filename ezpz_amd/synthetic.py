"""Synthetic workloads of the LM path: the configurations BASELINE.json names, built with the product's own front end
(ezpz_amd.textual, ezpz_amd.api constructors), and the keyed PRNG that jitters their guesses.  Used by bench.py, the
tools and the tests; product code only (no test infrastructure is imported here)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def keyed_uniform(seed, n_systems, n_vars, lo, hi, integer=False):
    """Counter-based PRNG keyed (seed, system, var) -> uniform [lo, hi)."""
    sys_idx = np.arange(n_systems, dtype=np.uint64)[:, None]
    var_idx = np.arange(n_vars, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = np.asarray([seed], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        x = key ^ (sys_idx * np.uint64(0xD1B54A32D192ED03)) ^ (var_idx * np.uint64(0x8CB92BA72F3D8DD7))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    if integer:
        return np.floor(lo + u * (hi - lo))
    return lo + u * (hi - lo)


def make_workload(name: str):
    """Returns (description, side-resolved constraint records, file guesses, jitter amplitude, expected iterations or
    None).  Built with the product's own front end (ezpz_amd.textual)."""
    import ezpz_amd as E

    if name.startswith("massive"):
        over = name.endswith("o")  # gen_big_problem.py <lines> true: one distance per line on top (5 rows per line, non-linear)
        lines = int(name[len("massive"):].rstrip("o") or 500)
        cs = E.textual.Problem.from_str(E.textual.gen_big_problem(lines, over)).to_constraint_system()
        rows = (5 if over else 4) * lines
        return (f"massive_parallel_system gen_big_problem.py {lines}{' true' if over else ''} ({rows} rows x {4 * lines} vars)",
                cs.records, cs.guesses, 0.25, None if over else 2)
    if name.startswith("sketch"):
        # one connected, fully determined sketch of mixed kinds: every point tied to its predecessors by two scalar
        # conditions consistent with a hidden layout (the generator of tests/gen.py:connected_sketch, same random stream,
        # on the product's own constructors)
        from ezpz_amd.api import DISTANCE, FIXED, HORIZONTAL_DISTANCE, VERTICAL_DISTANCE, _rec, stack_records

        npts = int(name[len("sketch"):] or 150)
        rng = np.random.default_rng(1000 + npts)
        pt = lambda i: [2 * i, 2 * i + 1]
        dist = lambda i, j: _rec(DISTANCE, pt(i) + pt(j), float(np.hypot(*(true[i] - true[j]))))
        hd = lambda i, j: _rec(HORIZONTAL_DISTANCE, pt(i) + pt(j), float(true[i][0] - true[j][0]))
        vd = lambda i, j: _rec(VERTICAL_DISTANCE, pt(i) + pt(j), float(true[i][1] - true[j][1]))
        cons, true = [_rec(FIXED, [0], 0.0), _rec(FIXED, [1], 0.0)], [np.zeros(2)]
        for i in range(1, npts):
            true.append(true[-1] + rng.uniform(0.5, 2.0, 2) * rng.choice([-1.0, 1.0], 2))
            a, b = i - 1, max(0, i - int(rng.integers(2, 4)))
            choice = int(rng.integers(0, 5))
            if choice == 0:
                cons += [hd(i, a), vd(i, a)]
            elif choice == 1:
                cons += [dist(i, a), dist(i, b) if b != a else hd(i, a)]
            elif choice == 2:
                cons += [dist(i, a), vd(i, a)]
            elif choice == 3:
                cons += [_rec(FIXED, [2 * i], float(true[i][0])), dist(i, a)]
            else:
                cons += [hd(i, b), dist(i, a)]
        guesses = np.concatenate(true) + rng.uniform(-0.05, 0.05, 2 * npts)
        return f"one connected sketch of {npts} points ({2 * npts} rows x {2 * npts} vars)", stack_records(cons), guesses, 0.02, None
    path = os.path.join(ROOT, "tests", "golden", "test_cases", name, "problem.md")
    cs = E.textual.Problem.from_str(open(path).read()).to_constraint_system()
    return f"test_cases/{name} ({cs.num_vars} vars)", E.resolve_sides(cs.records, cs.guesses), cs.guesses, 0.1, None
