"""GPU parity tests (-m gpu) of the component-resident launch shape (comp_kernel.hip.hpp: one lane per connected
component, class programs read through the scalar unit), through the C ABI, against the CPU oracle and against the
list-walk kernels.  Same tolerances as test_gpu_parity.py; block systems are additionally bitwise equal to the oracle
(every component is factorised in the same order with the same operations)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gen
from conftest import ROOT, read_case
from oracle import oracle as O
from oracle import textual as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return ezpz_amd


def assert_x_close(got, want, rel=1e-6):
    got, want = np.asarray(got), np.asarray(want)
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and (not np.any(~np.isnan(err)) or np.nanmax(err) <= rel), float(np.nanmax(err))


def replicate(constraints, guesses, copies, seed=0, jitter=0.0):
    """`copies` independent replicas of a system (variable ids shifted): a block system of isomorphic components."""
    n = len(guesses)
    rng = np.random.default_rng(seed)
    recs, gs = [], []
    for r in range(copies):
        for c in constraints:
            c = c.copy()
            c["ids"] = c["ids"] + r * n
            recs.append(c)
        gs.append(np.asarray(guesses) + (rng.uniform(-jitter, jitter, n) if jitter else 0.0))
    return O.stack(recs), np.concatenate(gs)


def solve_both(E, recs, x0, cfg=None, linsolve=O.LINSOLVE_SPARSE, expect_mode=3):
    """Component-resident solve of a batch + the oracle per system; checks everything the status carries."""
    cfg = cfg or {}
    n = x0.shape[1]
    sysobj = E.System(recs, n)
    info = sysobj.info()
    assert info["team_mode"] == expect_mode, info
    x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=linsolve)
    assert rc == 0
    assert np.array_equal(st["iterations"], it), (st["iterations"][:8], it[:8])
    assert np.array_equal(st["converged"], conv)
    assert np.array_equal(st["n_unsatisfied"], nun)
    assert np.array_equal(mask.sum(axis=1), nun)
    # the class-specialised kernel (run-time compiled straight-line code, state in registers): same bits as the
    # interpreter in every output, including the warning log
    xl, stl, logs = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=256)
    assert sysobj.specialize(wait=True) == 2
    x2, st2, mask2 = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
    assert np.array_equal(x2, x, equal_nan=True) and np.array_equal(mask2, mask)
    for f in st.dtype.names:
        assert np.array_equal(st2[f], st[f], equal_nan=True), f
    x3, st3, logs3 = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=256)
    assert np.array_equal(x3, xl, equal_nan=True) and logs3 == logs
    return sysobj, x, st, mask, xo


def test_massive_parallel_system_is_component_resident_and_bitwise(E):
    """BASELINE configs[1]: 2000 x 2000 = 1500 components in 2 classes; bitwise equal to the oracle and to the
    wavefront-partitioned list-walk kernel, 2 iterations (README.md:36-38)."""
    ref = T.load(T.gen_big_problem(500))
    x0 = ref.guesses[None, :] + gen.keyed_uniform(11, 40, ref.num_vars, -0.25, 0.25)
    x0[0] = ref.guesses
    sysobj, x, st, mask, xo = solve_both(E, ref.constraints, x0)
    info = sysobj.info()
    assert (info["n_components"], info["n_partitions"], info["team_size"]) == (1500, 24, 512)
    assert (info["nnz_j"], info["nnz_a"], info["nnz_l"], info["n_levels"]) == (2500, 2500, 2500, 2)
    assert np.array_equal(x, xo)
    assert np.all(st["iterations"] == 2) and np.all(st["n_unsatisfied"] == 0) and np.all(st["final_residual_inf"] <= 1e-9)
    walk = E.System(ref.constraints, ref.num_vars, team_size=512)
    assert walk.info()["team_mode"] == 1
    xw, stw, _ = walk.solve_batch(x0)
    assert np.array_equal(x, xw)
    for f in ("iterations", "converged", "n_unsatisfied", "n_warnings", "final_residual_inf", "final_lambda"):
        assert np.array_equal(st[f], stw[f]), f
    # one call of the reference's solve() lands on the same shape
    got = E.solve_records(ref.constraints, ref.guesses)
    assert (got.error, got.iterations, got.converged, got.unsatisfied) == (0, 2, True, [])
    assert np.array_equal(got.final_values, xo[0])


@pytest.mark.parametrize("lines", [43, 64, 130, 600])
def test_ragged_chunk_sizes_and_free_variables(E, lines):
    """Instance counts that are not multiples of 64 (partial chunks) and variables no constraint mentions (components of
    their own with no constraint: the reference's step for them is exactly zero)."""
    ref = T.load(T.gen_big_problem(lines))
    n = ref.num_vars
    x0 = np.concatenate([ref.guesses[None, :] + gen.keyed_uniform(5, 7, n, -0.25, 0.25), np.arange(35.0).reshape(7, 5) - 17.0], axis=1)
    x0[3, n + 2] = -0.0
    sysobj, x, st, mask, xo = solve_both(E, ref.constraints, x0)
    assert sysobj.info()["n_components"] == 3 * lines + 5
    assert np.array_equal(x, xo) and np.array_equal(np.signbit(x), np.signbit(xo))
    assert np.all(st["iterations"] == 2)


def test_overconstrained_lines_general_build(E):
    """gen_big_problem.py <lines> true: one non-linear `distance` per line on top -- 4-variable components of one class
    on the general (all kinds) build, Jacobian values in LDS, 3-4 iterations."""
    ref = T.load(T.gen_big_problem(300, True))
    x0 = ref.guesses[None, :] + gen.keyed_uniform(23, 12, ref.num_vars, -0.25, 0.25)
    sysobj, x, st, mask, xo = solve_both(E, ref.constraints, x0)
    assert sysobj.info()["n_components"] == 300
    assert_x_close(x, xo)
    assert np.all(st["n_unsatisfied"] == 0) and np.all(st["iterations"] >= 3)
    walk = E.System(ref.constraints, ref.num_vars, team_size=256)
    xw, stw, _ = walk.solve_batch(x0)
    assert np.array_equal(x, xw) and np.array_equal(st["iterations"], stw["iterations"])


@pytest.mark.parametrize("case,copies", [("square", 150), ("two_rectangles", 130), ("circle_tangent", 200), ("arc_radius", 140),
                                         ("chamfer_square", 128), ("perpendicular", 129), ("symmetric", 131)])
def test_replicated_fixtures_match_the_oracle_per_system(E, case, copies):
    """Block systems made of one reference fixture replicated: dense little components (square: 8 variables, lists
    longer than one record), circles, arcs, rejected steps (perpendicular).  LM accepts or rejects a step for the whole
    system, so a system of jittered replicas is its own test of the shared control."""
    ref = T.load(read_case(case))
    base = [O.set_from_initial_values(c, ref.guesses) for c in ref.constraints]
    recs, g = replicate(base, ref.guesses, copies, seed=7, jitter=0.02)
    x0 = np.stack([g, g + 0.004, g - 0.003])
    cfg = dict(max_iterations=120)
    sysobj, x, st, mask, xo = solve_both(E, recs, x0, cfg)
    assert_x_close(x, xo)
    for b in range(x0.shape[0]):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied
        assert int(st["n_warnings"][b]) == len(want.warnings)
        assert abs(float(st["final_residual_inf"][b]) - want.final_residual_inf) <= 1e-9 * max(1.0, want.final_residual_inf)
    # run-to-run determinism, and the same bits for every system of a batch of identical inputs
    x2, st2, _ = sysobj.solve_batch(np.tile(x0[:1], (9, 1)), E.Config(**cfg))
    assert np.all(x2 == x[0]) and np.all(st2["iterations"] == st["iterations"][0])


def test_weights_and_unsatisfiable_components(E):
    """Non-unit weights (class constants of the linear build when exactly representable, else the general build) and
    components that cannot be satisfied: the unsatisfied mask and counts, with r re-evaluated unweighted."""
    for wts, mode_linear in (((1.0, 1.0), True), ((2.5, 0.5), True), ((1.0 / 3.0, 7.1), False)):
        recs = []
        for k in range(160):
            a, b = 2 * k, 2 * k + 1
            recs.append(O.fixed(a, float(k), weight=wts[0]))
            recs.append(O.fixed(a, float(k) + (0.5 if k % 7 == 0 else 0.0), weight=wts[1]))  # conflicting on every 7th
            recs.append(O.scalar_equal(a, b))
            recs.append(O.fixed(b, float(k) + (2.0 if k % 5 == 0 else 0.0)))  # conflicting on every 5th
        recs = O.stack(recs)
        x0 = gen.keyed_uniform(3, 6, 320, -3.0, 3.0)
        sysobj, x, st, mask, xo = solve_both(E, recs, x0)
        assert np.all(st["n_unsatisfied"] > 0)
        rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
        assert_x_close(x, xo, 1e-9)
        want = O.solve(recs, x0[0], linsolve=O.LINSOLVE_SPARSE)
        assert np.nonzero(mask[0])[0].tolist() == want.unsatisfied


def test_degenerate_warnings_and_failed_pivots(E):
    """Degenerate evaluations (coincident points under `distance`): one warning per evaluation that officially happened,
    in the reference's order, including the sweeps of rejected steps and none for the speculative sweep of an iteration
    whose factorisation failed (newton.rs:93-99); NaN guesses in one component stall the whole system like the
    reference's global LM control does."""
    recs, guesses = [], []
    for k in range(140):
        o = 4 * k
        recs.append(O.distance((o, o + 1), (o + 2, o + 3), 1.0 + 0.01 * k))
        recs.append(O.fixed(o, float(k)))
        recs.append(O.fixed(o + 1, 0.0))
        recs.append(O.horizontal((o, o + 1), (o + 2, o + 3)))
        guesses += [float(k), 0.0, float(k) + (0.0 if k % 9 == 0 else 0.7), 0.0 if k % 9 == 0 else 0.2]
    recs = O.stack(recs)
    g = np.asarray(guesses)
    for cfg, gg in ((dict(), g), (dict(initial_lambda=1e-30), g), (dict(max_iterations=7), np.where(np.arange(len(g)) == 6, np.nan, g))):
        got = E.solve_records(recs, gg, E.Config(**cfg), warn_cap=1 << 16)
        want = O.solve(recs, gg, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert (got.error, got.iterations, got.converged, got.unsatisfied) == (want.error, want.iterations, want.converged, want.unsatisfied)
        assert got.warnings == want.warnings and len(want.warnings) > 0
        assert np.array_equal(np.isnan(got.final_values), np.isnan(want.final_values))
        assert_x_close(got.final_values, want.final_values)
        # the same through the interpreter and the specialised kernel (solve_both compares their logs entry by entry)
        sysobj, x, st, mask, xo = solve_both(E, recs, np.stack([gg, gg]), cfg)
        assert int(st["n_warnings"][0]) == len(want.warnings)


def test_iteration_limits_and_tolerances(E):
    ref = T.load(T.gen_big_problem(200))
    x0 = ref.guesses[None, :] + gen.keyed_uniform(2, 3, ref.num_vars, -0.25, 0.25)
    for cfg in (dict(max_iterations=0), dict(max_iterations=1), dict(residual_tolerance=1e3), dict(step_tolerance=1e3),
                dict(initial_lambda=1.0, max_iterations=6)):
        sysobj, x, st, mask, xo = solve_both(E, ref.constraints, x0, cfg)
        assert np.array_equal(x, xo), cfg


def test_mixed_classes_with_nonlinear_members(E):
    """A block system with several classes at once (linear and non-linear components, different sizes)."""
    parts = []
    for name, copies in (("tiny", 50), ("circle_tangent", 40), ("midpoint", 45), ("nonsquare", 33)):
        ref = T.load(read_case(name))
        base = [O.set_from_initial_values(c, ref.guesses) for c in ref.constraints]
        parts.append(replicate(base, ref.guesses, copies, seed=len(name), jitter=0.01))
    recs, gs, off = [], [], 0
    for r, g in parts:
        r = r.copy()
        r["ids"] = r["ids"] + off
        recs.append(r)
        gs.append(g)
        off += len(g)
    recs, g = np.concatenate(recs), np.concatenate(gs)
    x0 = np.stack([g, g + 0.002])
    sysobj, x, st, mask, xo = solve_both(E, recs, x0, dict(max_iterations=60))
    assert_x_close(x, xo)
    assert np.all(st["n_unsatisfied"] == 0)


def test_registered_host_buffers_pipeline_gives_the_same_results(E):
    """ezpz_host_register: batch calls from / to page-locked caller buffers stream through three device slots (copies
    and kernels overlap); results are those of the ordinary host path, bit for bit, for every launch shape that uses it."""
    ref = T.load(T.gen_big_problem(500))
    x0 = np.ascontiguousarray(ref.guesses[None, :] + gen.keyed_uniform(41, 3000, ref.num_vars, -0.25, 0.25))
    sq = T.load(read_case("square"))
    xs = np.ascontiguousarray(gen.keyed_uniform(42, 200000, 8, -100, 100, integer=True))
    for recs, n, xin in ((ref.constraints, ref.num_vars, x0), (sq.constraints, 8, xs)):
        sysobj = E.System(recs, n)
        want_x, want_st, _ = sysobj.solve_batch(xin)
        E.host_register(xin)
        out = np.empty_like(xin)
        E.host_register(out)
        try:
            import ctypes as C

            st = np.zeros(len(xin), dtype=E.STATUS_DTYPE)
            cfg = E.Config()._c()
            for _ in range(2):
                rc = E.lib().ezpz_system_solve_batch(sysobj._h, xin.ctypes.data, len(xin), C.byref(cfg), out.ctypes.data,
                                                     st.ctypes.data, None, None, 0)
                assert rc == 0
                assert np.array_equal(out, want_x)
                for f in st.dtype.names:
                    assert np.array_equal(st[f], want_st[f]), f
        finally:
            E.host_unregister(xin)
            E.host_unregister(out)


def test_registered_buffers_copy_out_kernel_with_odd_rows():
    """The pipeline's copy out as a kernel (EZPZ_H2H_OUT=kernel: what a process on a HIP runtime older than 7.2 takes by itself)
    on a system of an ODD number of variables and an odd number of systems: pieces then start and end on odd doubles, which the
    16-byte copy kernel used to cut short (the last double of a piece stayed unwritten with a status of success).  In a process
    of its own: the switch is read once."""
    code = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import ezpz_amd as E, gen
from oracle import oracle as O
pts = 3
cons = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.distance((0, 1), (2, 3), 2.0), O.horizontal((0, 1), (2, 3)), O.distance((2, 3), (4, 5), 1.5),
        O.vertical((2, 3), (4, 5)), O.fixed(6, 7.0)]
recs, n = O.stack(cons), 7
B = 200001
x0 = np.ascontiguousarray(np.array([0.1, 0.1, 1.9, 0.2, 2.1, 1.4, 6.0])[None, :] + gen.keyed_uniform(7, B, n, -0.05, 0.05))
s = E.System(recs, n)
want_x, want_st, _ = s.solve_batch(x0)
out = np.full_like(x0, np.nan)
E.host_register(x0); E.host_register(out)
st = np.zeros(B, dtype=E.STATUS_DTYPE)
cfg = E.Config()._c()
rc = E.lib().ezpz_system_solve_batch(s._h, x0.ctypes.data, B, C.byref(cfg), out.ctypes.data, st.ctypes.data, None, None, 0)
E.host_unregister(x0); E.host_unregister(out)
assert rc == 0
assert not np.isnan(out).any(), int(np.isnan(out).sum())
assert np.array_equal(out, want_x) and np.array_equal(st["iterations"], want_st["iterations"])
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, EZPZ_H2H_OUT="kernel", EZPZ_H2H_PIECE_MB="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("lines,over", [(12000, False), (12000, True), (3000, True)])
def test_specialised_kernel_on_several_workgroups(E, lines, over):
    """Systems whose state exceeds one CU's LDS have no interpreter form; their specialised kernel shares a system among
    workgroups (self-validating 16-byte chunks for the reductions of the LM control).  Against the list-walk grid team
    and the oracle: linear ladder bitwise; the over-constrained variant (a non-linear `distance` per line, 3-4
    iterations, general evaluators) bitwise against the list-walk kernel, 1e-6 against the oracle."""
    ref = T.load(T.gen_big_problem(lines, over))
    n = ref.num_vars
    x0 = ref.guesses[None, :] + gen.keyed_uniform(77, 5, n, -0.25, 0.25)
    x0[0] = ref.guesses
    sysobj = E.System(ref.constraints, n)
    xw, stw, maskw = sysobj.solve_batch(x0, want_mask=True)   # list-walk (grid team or one workgroup)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.array_equal(stw["iterations"], it)
    spec = sysobj.specialize(wait=True)
    assert spec == 2
    for _ in range(3):
        x, st, mask = sysobj.solve_batch(x0, want_mask=True)
        assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv) and np.array_equal(st["n_unsatisfied"], nun)
        assert np.array_equal(mask, maskw)
        assert np.array_equal(x, xw)
        assert_x_close(x, xo)
        if not over:
            assert np.array_equal(x, xo)
        assert np.array_equal(st["final_lambda"], stw["final_lambda"]) and np.array_equal(st["n_warnings"], stw["n_warnings"])


@pytest.mark.parametrize("lines,over,B", [(12000, False, 33), (3000, True, 33), (500, False, 80), (500, True, 80)])
def test_verdicts_not_waited_for_every_exit(E, lines, over, B):
    """A linear block system does not wait for the verdicts of the LM control (jit_kernel.hip.hpp: solve_kernel_fast for a
    system on one workgroup -- 500 lines: the 2000 x 2000 headline --, solve_kernel_grid_fast on several -- 12 000 lines; calls of
    more than 1 MB through the host entry run out of place, which those kernels need):
    both iterations are taken and the values stored before anybody knows whether the steps stand, and
    whatever the verdicts turn out to be the system must end on the reference's path (newton.rs:47-139) -- converged at the
    start, converged after one iteration, a failed pivot (a NaN guess: every iteration burnt), a rejected first step, the step
    tolerance met by the first step, iteration limits of 0 ... 3 -- with such systems at the start, in the middle and at the
    end of a slot's sequence of ordinary ones (a re-solve, then verdicts waited for, while two later systems' verdicts are
    still pending).  Against the list-walk kernel in every output, and the oracle.  (over: the non-linear variant, which
    always waits -- the same cases on the loop alone.)"""
    ref = T.load(T.gen_big_problem(lines, over))
    n = ref.num_vars
    sysobj = E.System(ref.constraints, n)
    exact = np.zeros(n)
    exact[0::4] = exact[2::4] = np.arange(lines)
    exact[3::4] = 4.0
    x0 = ref.guesses[None, :] + gen.keyed_uniform(78, B, n, -0.25, 0.25)
    cfgs = [E.Config(), E.Config(max_iterations=0), E.Config(max_iterations=1), E.Config(max_iterations=2), E.Config(max_iterations=3),
            E.Config(residual_tolerance=-1.0),                       # exact start: step 0, rejected (0 < 0 is false), step test ends it
            E.Config(residual_tolerance=-1.0, step_tolerance=-1.0, max_iterations=6),  # ... and nothing ends it: six rejected steps
            E.Config(step_tolerance=1e3)]                            # the first step meets the step tolerance
    seen = set()
    for variant in range(2):
        if variant == 0:  # special systems scattered among ordinary ones
            x0[1] = exact                                                    # converged before the first iteration
            x0[10] = exact + gen.keyed_uniform(79, 1, n, -1e-4, 1e-4)[0]     # one iteration is enough
            x0[18, 5] = np.nan                                               # a pivot fails in every iteration
            x0[27] = exact
            x0[27, 2 * (lines // 2)] += 1e-6                                  # one component away from the solution, barely
            x0[B - 1] = exact                                                # the last system of its slot
        else:  # ... and every system special (every verdict a re-solve from the first on)
            x0[:] = exact[None, :] + gen.keyed_uniform(80, B, n, -1e-5, 1e-5)
            x0[::3] = exact
        fresh = E.System(ref.constraints, n)
        want = [fresh.solve_batch(x0, cfg, want_mask=True) for cfg in cfgs]  # list-walk
        assert sysobj.specialize(wait=True) == 2
        for cfg, (xw, stw, maskw) in zip(cfgs, want):
            ocfg = O.Config(cfg.max_iterations, cfg.residual_tolerance, cfg.step_tolerance, cfg.initial_lambda)
            rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, ocfg, linsolve=O.LINSOLVE_SPARSE)
            assert rc == 0
            for rep in range(2):
                x, st, mask = sysobj.solve_batch(x0, cfg, want_mask=rep == 0)
                for f in st.dtype.names:
                    assert np.array_equal(st[f], stw[f], equal_nan=True), (cfg, f, st[f], stw[f])
                assert np.array_equal(x, xw, equal_nan=True) and (rep or np.array_equal(mask, maskw)), cfg
                assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv) and np.array_equal(st["n_unsatisfied"], nun), cfg
                fin = ~np.isnan(xo).any(axis=1)
                assert_x_close(x[fin], xo[fin])
                if not over:
                    assert np.array_equal(x, xo, equal_nan=True), cfg
            seen.update((int(i), int(c)) for i, c in zip(st["iterations"], st["converged"]))
    # the exits were really taken
    assert {(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (6, 0)} <= seen, str(sorted(seen))


@pytest.mark.parametrize("lines,permute", [(333, False), (500, True), (97, False), (1000, True)])
def test_verdicts_not_waited_for_row_layouts(E, lines, permute):
    """How the kernel of a linear block system moves its rows (jit_kernel.hip.hpp: fast_wave): where every wavefront's variables
    are one contiguous piece of the row -- gen_big_problem.py's numbering -- as full 16-byte accesses through an LDS copy of the
    piece, its last wavefront's piece shorter, odd lengths included (333, 97 lines); where they are not -- the same systems with
    their variables renumbered at random -- value by value.  Calls above 1 MB (out of place); bitwise against the oracle."""
    ref = T.load(T.gen_big_problem(lines))
    n = ref.num_vars
    recs = O.stack(ref.constraints).copy()
    guesses = ref.guesses.copy()
    if permute:
        perm = np.random.default_rng(lines).permutation(n).astype(np.uint32)
        for i in range(len(recs)):
            k = O.KIND_NUM_IDS[int(recs["kind"][i])]
            recs["ids"][i, :k] = perm[recs["ids"][i, :k]]
        g2 = np.empty(n)
        g2[perm] = guesses
        guesses = g2
    B = max(40, (1 << 20) // (8 * n) + 8)
    x0 = guesses[None, :] + gen.keyed_uniform(5 + lines, B, n, -0.25, 0.25)
    sysobj = E.System(recs, n)
    assert sysobj.specialize(wait=True) == 2
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.all(it == 2)
    for _ in range(2):
        x, st, mask = sysobj.solve_batch(x0, want_mask=True)
        assert np.array_equal(x, xo) and np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv) and not mask.any()


def test_verdicts_not_waited_for_on_two_streams(E):
    """The `_device` entry may be called on one EzpzSystem from several streams.  The kernels that do not wait for verdicts share two
    redo lists per system between consecutive calls (each call's last launch zeroes the other list's count): calls on another
    stream than the last one are chained behind it (launch.hip: chain_launches).  Twelve calls alternating between two streams, every
    third system of every call one that needs the loop kernel: every call's results equal the host entry's."""
    import torch

    ref = T.load(T.gen_big_problem(500))
    n = ref.num_vars
    sysobj = E.System(ref.constraints, n)
    assert sysobj.specialize(wait=True) == 2
    exact = np.zeros(n)
    exact[0::4] = exact[2::4] = np.arange(500)
    exact[3::4] = 4.0
    B = 96
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    calls = []
    for c in range(12):
        x0 = ref.guesses[None, :] + gen.keyed_uniform(900 + c, B, n, -0.25, 0.25)
        x0[c % 3::3] = exact + gen.keyed_uniform(950 + c, 1, n, -1e-5, 1e-5)[0]  # (one iteration is enough: not the expected verdicts)
        st = streams[c % 2]
        with torch.cuda.stream(st):
            xd = torch.from_numpy(x0).to(dev, non_blocking=False)
            xo = torch.empty_like(xd)
            sd = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
            sysobj.solve_batch_device(xd.data_ptr(), B, xo.data_ptr(), sd.data_ptr(), 0, st.cuda_stream, E.Config())
        calls.append((x0, xd, xo, sd))
    torch.cuda.synchronize(dev)
    for x0, xd, xo, sd in calls:
        x, st, _ = sysobj.solve_batch(x0)
        assert np.array_equal(xo.cpu().numpy(), x)
        got = np.frombuffer(sd.cpu().numpy().tobytes(), dtype=E.STATUS_DTYPE)
        assert np.array_equal(got["iterations"], st["iterations"]) and np.array_equal(got["converged"], st["converged"])
        assert set(int(i) for i in st["iterations"]) == {1, 2}


def test_random_linear_classes_on_the_kernels_that_do_not_wait(E):
    """The generator splits a linear class's `solve` into the factorisation of J^T J + lambda I (lambda only: once per launch, in
    scalar registers) and the substitutions (comp_program.cpp: factor / solve_f) for the kernels that do not wait for the LM
    control's verdicts.  gen_big_problem.py's two classes are a 2 x 2 and a 1 x 1 matrix; here: random little systems of the nine
    LINEAR kinds (3 ... 8 variables, 2 ... 7 constraints: several elimination levels, fill, free variables, duplicated and
    contradictory constraints -- whose systems do not converge in two iterations and go through the redo list), each replicated 130 ...
    400 times with jittered guesses into a block system, calls above 1 MB (out of place: those kernels), twice (the redo lists of
    consecutive calls alternate).  Every output against the component interpreter of a fresh system, bit for bit, and the oracle."""
    rng = np.random.default_rng(2468)
    linear = [O.FIXED, O.SCALAR_EQUAL, O.VERTICAL, O.HORIZONTAL, O.VERTICAL_DISTANCE, O.HORIZONTAL_DISTANCE, O.CIRCLE_RADIUS,
              O.POINTS_COINCIDENT, O.MIDPOINT]
    took_fast = redone = 0
    for trial in range(12):
        nv = int(rng.integers(3, 9))
        cons = [gen.arb_constraint(rng, int(rng.choice(linear)), hi=nv) for _ in range(int(rng.integers(2, 8)))]
        base = rng.uniform(-6.0, 6.0, nv)
        copies = int(rng.integers(130, 401))
        recs, g = replicate(cons, base, copies, seed=trial, jitter=0.05)
        n = len(g)
        B = (1 << 20) // (8 * n) + 9  # (just above the zero-copy limit of the host entry)
        x0 = g[None, :] + gen.keyed_uniform(300 + trial, B, n, -0.5, 0.5)
        x0[B // 2] = g  # (and an unjittered one)
        fresh = E.System(recs, n)
        if fresh.info()["team_mode"] != 3:
            continue
        xw, stw, maskw = fresh.solve_batch(x0, want_mask=True)  # the component interpreter
        sysobj = E.System(recs, n)
        assert sysobj.specialize(wait=True) == 2
        src = E.specialized_source(recs, n)
        took_fast += "ezpz_jit_solve_fast" in src
        for rep in range(2):
            x, st, mask = sysobj.solve_batch(x0, want_mask=True)
            assert np.array_equal(x, xw, equal_nan=True) and np.array_equal(mask, maskw), trial
            for f in st.dtype.names:
                assert np.array_equal(st[f], stw[f], equal_nan=True), (trial, f)
        redone += int(np.any(st["iterations"] != 2))
        want = O.solve(recs, x0[0], linsolve=O.LINSOLVE_SPARSE)
        assert want.error == 0 and np.array_equal(np.isnan(x[0]), np.isnan(want.final_values)), trial
        if want.converged and want.final_residual_inf <= 1e-8:
            # (consistent constraints: the residual test ends the solve.  Contradictory ones end on the step test at the least-squares
            # minimum, after as many iterations as the noise in |d| takes to drop below 1e-12 -- the kernels agree with each other
            # on that count bit for bit, the oracle's summation order gives another)
            assert want.iterations == int(st["iterations"][0]) and bool(st["converged"][0]), trial
    assert took_fast >= 8 and redone >= 2, (took_fast, redone)


def test_random_classes_interpreter_and_specialised_kernel_agree_bitwise(E):
    """Random little systems of all 25 kinds (the fuzz generator's), each replicated 130 times with jittered guesses into
    a block system: the component interpreter and the run-time compiled kernel give the same bits in every output
    (solve_both), and the oracle's answer where it is determined."""
    rng = np.random.default_rng(4321)
    kinds_seen = set()
    for trial in range(14):
        nv = int(rng.integers(3, 9))
        cons = []
        for _ in range(int(rng.integers(1, 6))):
            c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nv)
            c["weight"] = float(rng.choice([1.0, 1.0, 2.5]))
            cons.append(c)
        kinds_seen.update(int(c["kind"]) for c in cons)
        base = rng.uniform(-6.0, 6.0, nv)
        recs, g = replicate(cons, base, 130, seed=trial, jitter=0.05)
        x0 = np.stack([g, g + 0.01])
        sysobj = E.System(recs, len(g))
        if sysobj.info()["team_mode"] != 3:
            continue
        cfg = dict(max_iterations=12)
        x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
        xl, stl, logs = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=16384)
        assert int(stl["n_warnings"].max()) <= 16384  # (a truncated log keeps whichever entries arrived first)
        assert sysobj.specialize(wait=True) == 2
        x2, st2, mask2 = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
        x3, st3, logs3 = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=16384)
        assert np.array_equal(x2, x, equal_nan=True) and np.array_equal(mask2, mask) and logs3 == logs, trial
        for f in st.dtype.names:
            assert np.array_equal(st2[f], st[f], equal_nan=True), (trial, f)
        want = O.solve(recs, x0[0], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0 and np.array_equal(np.isnan(x[0]), np.isnan(want.final_values)), trial
        assert int(st["n_warnings"][0]) == len(want.warnings) or not want.converged, trial
    assert len(kinds_seen) >= 20


@pytest.mark.parametrize("lines,over", [(50, False), (500, True)])
def test_batches_beyond_the_launchs_workgroups_draw_their_systems(E, lines, over):
    """The specialised kernel's workgroups draw their systems from eight counters once a batch exceeds the workgroups the device
    holds (jit_kernel.hip.hpp: JitArgs::ticket; the host keeps the counters' running totals from launch to launch): odd batch
    sizes, one launch after the other, every system of every launch bit for bit what calls that FIT the launch give (fixed
    shares: no counter), statuses included -- a system solved twice, skipped, or handed to two workgroups would show."""
    import torch

    ref = T.load(T.gen_big_problem(lines, over))
    n = ref.num_vars
    sysobj = E.System(ref.constraints, n)
    assert sysobj.specialize(wait=True) == 2
    B = 9001
    x0 = ref.guesses[None, :] + gen.keyed_uniform(5, B, n, -0.25, 0.25)
    want_x = np.empty_like(x0)
    want_st = np.zeros(B, dtype=E.STATUS_DTYPE)
    for lo in range(0, B, 256):  # 256 systems: fewer than the device holds workgroups
        xs, sts, _ = sysobj.solve_batch(x0[lo:lo + 256])
        want_x[lo:lo + 256], want_st[lo:lo + 256] = xs, sts
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0[:64], linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.array_equal(want_st["iterations"][:64], it)
    assert_x_close(want_x[:64], xo)
    xin = torch.from_numpy(x0).cuda()
    for count in (B, 4099, 8191, 300, B, 7777):  # (300: a call that fits between two that do not)
        xd = torch.full((count, n), float("nan"), dtype=torch.float64, device="cuda")
        std = torch.zeros((count, 32), dtype=torch.uint8, device="cuda")
        sysobj.solve_batch_device(xin.data_ptr(), count, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), want_x[:count]), count
        assert np.array_equal(std.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1), want_st[:count]), count


def test_a_batch_launch_recorded_into_a_graph_replays_on_new_guesses(E):
    """A launch that is being recorded into a HIP graph (here: torch.cuda.graph) keeps fixed shares of the batch -- a replay would find
    the counters of `test_batches_beyond_the_launchs_workgroups_draw_their_systems` elsewhere (launch.hip): replays on new guesses, with
    direct calls (which draw) between them, all equal the direct call's results bit for bit."""
    import torch

    ref = T.load(T.gen_big_problem(50))
    n = ref.num_vars
    s = E.System(ref.constraints, n)
    assert s.specialize(wait=True) == 2
    B = 9001
    x0 = ref.guesses[None, :] + gen.keyed_uniform(5, B, n, -0.25, 0.25)
    xin = torch.from_numpy(x0).cuda()
    xo = torch.empty_like(xin)
    st = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):  # warm: everything the call creates on first use exists before the recording
            s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, side.cuda_stream)
    torch.cuda.synchronize()
    want = xo.cpu().numpy().copy()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        flip = rep % 2 == 1
        xin.copy_(torch.from_numpy(x0[::-1].copy() if flip else x0).cuda())
        xo.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(xo.cpu().numpy(), want[::-1] if flip else want), rep
        s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(xo.cpu().numpy(), want[::-1] if flip else want), ("direct", rep)


def test_drawn_systems_on_two_streams_of_one_system_object(E):
    """Launches that draw from one system object's counters must not overlap: on one stream they do not anyway, and when the stream
    changes the new one waits for an event recorded on the old (launch.hip).  Two streams taking turns, and then both loaded before
    either is waited for: every result bit for bit the reference run's."""
    import torch

    ref = T.load(T.gen_big_problem(50))
    n = ref.num_vars
    s = E.System(ref.constraints, n)
    assert s.specialize(wait=True) == 2
    B = 6001
    x0 = ref.guesses[None, :] + gen.keyed_uniform(9, B, n, -0.25, 0.25)
    xin = torch.from_numpy(x0).cuda()
    want = torch.empty_like(xin)
    st = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
    s.solve_batch_device(xin.data_ptr(), B, want.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.full_like(xin, float("nan")) for _ in range(6)]
    sts = [torch.zeros((B, 32), dtype=torch.uint8, device="cuda") for _ in range(6)]
    for i, out in enumerate(outs):  # six launches, alternating streams, nothing waited for in between
        s.solve_batch_device(xin.data_ptr(), B, out.data_ptr(), sts[i].data_ptr(), 0, streams[i % 2].cuda_stream)
    torch.cuda.synchronize()
    for i, out in enumerate(outs):
        assert torch.equal(out, want), i
        assert torch.equal(sts[i], st), i


def test_verdicts_not_waited_for_on_several_workgroups_long_launches(E):
    """The 200 000-variable ladder (98 workgroups per system) has two compilations of the kernel that does not wait for verdicts
    (jit.cpp: comp_jit_launch): launches of up to six rounds of the systems in flight take the one at four wavefronts per SIMD, longer
    ones the one at three, whose values wait in LDS for stores that go out back to back.  Device-resident launches of 24 (the first)
    and 72 systems (the second; a host call would arrive in pieces of ten), with systems whose verdicts are not the expected ones at
    the start, in the middle and at the end -- converged at the start, one iteration enough, a failed pivot, barely off -- under
    three configurations: every output equal to the oracle's bit for bit, and the short launch's to the long launch's."""
    import torch

    ref = T.load(T.gen_big_problem(50000))
    n = ref.num_vars
    lines = 50000
    sysobj = E.System(ref.constraints, n)
    assert sysobj.info()["grid_workgroups"] > 1 and sysobj.specialize(wait=True) == 2
    B = 72
    exact = np.zeros(n)
    exact[0::4] = exact[2::4] = np.arange(lines)
    exact[3::4] = 4.0
    x0 = ref.guesses[None, :] + gen.keyed_uniform(178, B, n, -0.25, 0.25)
    x0[0] = exact
    x0[10] = exact + gen.keyed_uniform(179, 1, n, -1e-4, 1e-4)[0]
    x0[18, 5] = np.nan
    x0[27] = exact
    x0[27, 2 * (lines // 2)] += 1e-6
    x0[40] = exact
    x0[B - 1] = exact
    xin = torch.from_numpy(x0).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    seen = set()
    for cfg in (E.Config(), E.Config(max_iterations=3), E.Config(step_tolerance=1e3)):
        ocfg = O.Config(cfg.max_iterations, cfg.residual_tolerance, cfg.step_tolerance, cfg.initial_lambda)
        rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, ocfg, linsolve=O.LINSOLVE_SPARSE)
        assert rc == 0
        for count in (B, 24, B):
            xd = torch.full((count, n), float("nan"), dtype=torch.float64, device="cuda")
            std = torch.zeros((count, 32), dtype=torch.uint8, device="cuda")
            sysobj.solve_batch_device(xin.data_ptr(), count, xd.data_ptr(), std.data_ptr(), 0, stream, cfg)
            torch.cuda.synchronize()
            st = std.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
            assert np.array_equal(st["iterations"], it[:count]) and np.array_equal(st["converged"], conv[:count]), (cfg, count)
            assert np.array_equal(st["n_unsatisfied"], nun[:count]), (cfg, count)
            assert np.array_equal(xd.cpu().numpy(), xo[:count], equal_nan=True), (cfg, count)
        seen.update((int(i), int(c)) for i, c in zip(st["iterations"], st["converged"]))
    assert {(0, 1), (1, 1), (2, 1), (3, 0)} <= seen, str(sorted(seen))


def test_drawn_systems_survive_register_spills(E, tmp_path):
    """Workgroups draw their systems with one lane's atomic whose answer is wanted a system later (jit_kernel.hip.hpp: tickets).
    Round 5 / 6 issued it in an asm statement and waited for it in another: a compilation under register pressure saved the
    register to scratch memory in between -- before the answer had arrived -- and systems were skipped (found with a forced
    occupancy: statuses untouched).  The draw is the compiler's atomic now, which waits before any read of its own.  Here: the
    2400 x 2400 block system compiled for four wavefronts per SIMD (EZPZ_JIT_FAST_MINWAVES=4: 59 registers in scratch memory),
    8192 systems per launch so that the workgroups draw; every system solved, a sample bitwise against the oracle."""
    import subprocess
    import sys

    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import ezpz_amd as E, gen
from oracle import oracle as O
from oracle import textual as T
lines = 600
ref = T.load(T.gen_big_problem(lines))
n = ref.num_vars
s = E.System(ref.constraints, n)
assert s.specialize(wait=True) == 2
B = 8192
x0 = ref.guesses[None, :] + gen.keyed_uniform(611, B, n, -0.25, 0.25)
xin = torch.from_numpy(x0).cuda()
for rep in range(3):
    xd = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
    std = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
    s.solve_batch_device(xin.data_ptr(), B, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = std.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
    x = xd.cpu().numpy()
    assert np.all(st["iterations"] == 2) and np.all(st["converged"] == 1), (rep, int((st["iterations"] != 2).sum()))
    assert not np.isnan(x).any()
    exact = np.zeros(n); exact[0::4] = exact[2::4] = np.arange(lines); exact[3::4] = 4.0
    assert np.max(np.abs(x - exact[None, :])) <= 1e-9
    sample = np.arange(0, B, B // 64)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0[sample], linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.array_equal(x[sample], xo)
print("ok")
'''
    env = dict(os.environ, EZPZ_JIT_FAST_MINWAVES="4", EZPZ_JIT_CACHE_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_random_linear_blocks_in_launches_whose_workgroups_draw(E):
    """tools/stress_random_blocks.py, eight of its random block systems of the linear kinds: device-resident launches of 3000 ... 9001
    systems -- more than the launch has workgroups, so they draw their systems -- through the kernels that do not wait for verdicts
    and their redo lists.  Every value and status equal to the loop kernel alone (a call in place) bit for bit, whole batch, three
    times over; and to the component interpreter wherever the residual test ends the solve.  (61 systems, the compilations forced
    to other occupancies among them, ran clean at the end of round 6: profiles/r06_stress_blocks.txt.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stress_random_blocks as S

    done, fast, redone, bad = S.run_trials(8, 97531)
    assert bad == 0 and done >= 6 and fast >= 6


def test_sequence_numbers_of_the_scratch_areas_start_again(E, tmp_path):
    """The scratch areas through which a system's workgroups talk carry 32-bit sequence numbers that go on from launch to launch
    (the ladder's ring and reductions, the list-walk grid team's reductions, the fronts' hops): before an upper bound of their use
    reaches 2^31 the launch code zeroes the area behind the last launch that used it and the numbers start again (system.hpp:
    seq_budget_spent) -- a process that solves such a system for days never meets the wrap.  Here with EZPZ_SEQ_BUDGET=2000 (a
    reset every few launches): thirty calls each of the ladder (list walk, then compiled), and of a sketch on 14 workgroups, all
    equal to the first call's results bit for bit."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import ezpz_amd as E, gen
from oracle import textual as T
lad = T.load(T.gen_big_problem(12000))
n = lad.num_vars
x0 = lad.guesses[None, :] + gen.keyed_uniform(71, 9, n, -0.25, 0.25)
x0[4] = 0.0; x0[4, 0::4] = x0[4, 2::4] = np.arange(12000); x0[4, 3::4] = 4.0
for compiled in (False, True):
    s = E.System(lad.constraints, n)
    if compiled:
        assert s.specialize(wait=True) == 2
    assert s.info()["grid_workgroups"] > 1
    ref = s.solve_batch(x0)
    assert np.all(ref[1]["converged"] == 1)
    for rep in range(30):
        x, st, _ = s.solve_batch(x0)
        assert np.array_equal(x, ref[0]) and all(np.array_equal(st[f], ref[1][f]) for f in st.dtype.names), (compiled, rep)
recs, g = gen.connected_sketch(1000, 2000)
f = E.System(recs, len(g), team_size=E.TEAM_AUTO_LATENCY)
assert f.info()["front_workgroups"] > 1
xs = g[None, :] + np.random.default_rng(5).uniform(-0.01, 0.01, (3, len(g)))
cfg = E.Config(max_iterations=40)
ref = f.solve_batch(xs, cfg)
for rep in range(30):
    x, st, _ = f.solve_batch(xs, cfg)
    assert np.array_equal(x, ref[0]) and all(np.array_equal(st[k], ref[1][k]) for k in st.dtype.names), rep
print("ok")
'''
    env = dict(os.environ, EZPZ_SEQ_BUDGET="2000", EZPZ_JIT_CACHE_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_a_system_on_several_workgroups_refuses_stream_capture(E):
    """Launches whose workgroups wait for each other are chained on a process-wide event and zero their scratch on first use; inside a
    stream capture either would invalidate the capture and leave the event unusable.  They refuse up front (EZPZ_ERR_INVALID_ARGUMENT,
    nothing enqueued), on the list walk and on the compiled kernels, and a direct call afterwards gives the results it always gave."""
    import warnings

    import torch

    lad = T.load(T.gen_big_problem(12000))
    n, B = lad.num_vars, 4
    x0 = lad.guesses[None, :] + gen.keyed_uniform(81, B, n, -0.25, 0.25)
    for compiled in (False, True):
        s = E.System(lad.constraints, n)
        if compiled:
            assert s.specialize(wait=True) == 2
        assert s.info()["grid_workgroups"] > 1
        xin = torch.from_numpy(x0).cuda()
        xo = torch.empty_like(xin)
        st = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, side.cuda_stream)
        side.synchronize()
        want = xo.clone()
        g = torch.cuda.CUDAGraph()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # ("The CUDA Graph is empty")
            with pytest.raises(E.NonLinearSystemError) as err:
                with torch.cuda.graph(g, stream=side):
                    s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        assert "invalid argument" in str(err.value)
        torch.cuda.synchronize()
        xo.fill_(float("nan"))
        s.solve_batch_device(xin.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(xo, want), compiled
