"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Tolerances (BASELINE.json north_star): converged coordinates within 1e-6 relative, residual norm within
1e-9 absolute, iteration counts / converged flags / unsatisfied lists identical.  Per-constraint residuals
and Jacobian entries are compared at 1e-11 relative (device libm differs from glibc in the last ulp).
"""
import json
import os

import numpy as np
import pytest

import cases
import gen
from adapters import GpuAdapter, OracleAdapter
from conftest import GOLDEN, read_case
from oracle import oracle as O
from oracle import textual as T

pytestmark = pytest.mark.gpu

REL = 1e-6


@pytest.fixture(scope="module")
def E():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return ezpz_amd


def assert_x_close(got, want, rel=REL):
    got, want = np.asarray(got), np.asarray(want)
    scale = np.maximum(1.0, np.abs(want))
    err = np.abs(got - want) / scale
    assert np.all(err <= rel) or (np.array_equal(np.isnan(got), np.isnan(want)) and np.nanmax(err) <= rel), float(np.nanmax(err))


# ---- the reference's own tests, on the HIP path ------------------------------------------------------
@pytest.mark.parametrize("case", cases.ALL_CASES, ids=[c.__name__ for c in cases.ALL_CASES])
def test_reference_known_answer(case, E):
    case(GpuAdapter())


# ---- kernel K1: residual + Jacobian of every kind ------------------------------------------------------
@pytest.mark.parametrize("kind", range(O.NUM_KINDS), ids=O.KIND_NAMES)
def test_residual_and_jacobian_match_oracle(kind, E):
    rng = np.random.default_rng(3000 + kind)
    n = 32
    cons = [gen.arb_constraint(rng, kind, hi=n) for _ in range(40)]
    for c in cons:
        c["weight"] = float(rng.choice([1.0, 1.0, 2.5, 0.5]))
    sysobj = E.System(O.stack(cons), n)
    X = rng.uniform(-8.0, 8.0, size=(24, n))
    X[0, :] = 0.0  # everything coincident: exercises every degenerate guard
    X[1, :] = 1.0
    r, J, deg = sysobj.eval_batch(X)
    for b in range(X.shape[0]):
        row = 0
        ndeg = 0
        for c in cons:
            rr, d0 = O.residual(c, X[b])
            rows, d1 = O.jacobian_rows(c, X[b])
            ndeg += int(d0) + int(d1)
            for k in range(len(rr)):
                want = c["weight"] * rr[k]
                got = r[b, row + k]
                assert (np.isnan(want) and np.isnan(got)) or abs(got - want) <= 1e-11 * max(1.0, abs(want)), (O.KIND_NAMES[kind], b, got, want)
                dense = np.zeros(n)
                for (i, pd) in rows[k]:
                    dense[i] += c["weight"] * pd
                gj = J[b, row + k]
                both_nan = np.isnan(dense) & np.isnan(gj)
                ok = both_nan | (np.abs(gj - dense) <= 1e-10 * np.maximum(1.0, np.abs(dense)))
                # non-finite values (unguarded divisions in the reference) must be non-finite on both sides
                ok |= ~np.isfinite(dense) & ~np.isfinite(gj)
                assert np.all(ok), (O.KIND_NAMES[kind], b, row + k, gj, dense)
            row += len(rr)
        assert deg[b] == ndeg, (O.KIND_NAMES[kind], b, deg[b], ndeg)


# ---- committed golden vectors -------------------------------------------------------------------------------
def test_golden_vectors(E):
    vectors = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))
    assert len(vectors) == 28
    for case, recs in vectors.items():
        text = open(os.path.join(GOLDEN, "test_cases", case)).read()
        system = E.textual.Problem.from_str(text).to_constraint_system()
        for rec in recs:
            got = E.solve_records(system.records, np.asarray(rec["guesses"]))
            assert got.error == 0, case
            assert got.iterations == rec["iterations"], (case, got.iterations, rec["iterations"])
            assert got.converged == rec["converged"], case
            assert got.unsatisfied == rec["unsatisfied"], case
            assert len(got.warnings) == rec["n_warnings"], case
            # Every coordinate the constraints determine is held to 1e-6 relative (north_star).  Only the variables the
            # oracle's FreedomAnalysis flags as underconstrained (find_dof.rs; recorded per vector by make_vectors.py)
            # may use a wider bar: they are held in place by lambda ~ 1e-9..1e-10 alone, the oracle's own answer moves
            # by `ulp_sensitivity` there under a one-ulp input perturbation, and the bar is 20x that -- never beyond
            # the reference's own test tolerance of 1e-4.
            want = np.asarray(rec["final_values"])
            free = np.zeros(len(want), dtype=bool)
            free[rec["underconstrained"]] = True
            gotv = np.asarray(got.final_values)
            if np.any(~free):
                assert_x_close(gotv[~free], want[~free], REL)
            if np.any(free):
                assert_x_close(gotv[free], want[free], min(1e-4, max(REL, 20.0 * rec["ulp_sensitivity"])))
            assert abs(got.final_residual_inf - rec["final_residual_inf"]) <= 1e-9, case


# ---- batches ------------------------------------------------------------------------------------------------
def _batch_vs_oracle(E, text, x0, team_size=0, linsolve=O.LINSOLVE_DENSE):
    ref = T.load(text)
    sysobj = E.System(ref.constraints, ref.num_vars, team_size=team_size)
    x, st, mask = sysobj.solve_batch(x0, want_mask=True)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=linsolve)
    assert rc == 0
    assert np.array_equal(st["iterations"], it), np.nonzero(st["iterations"] != it)[0][:10]
    assert np.array_equal(st["converged"], conv)
    assert np.array_equal(st["n_unsatisfied"], nun)
    assert np.array_equal(mask.sum(axis=1), nun)
    assert_x_close(x, xo)
    return sysobj, x, st


def test_square_batch_file_guesses_and_random_integer_guesses(E):
    """BASELINE config 3 (reduced to what the oracle checks in seconds) + proptests.rs:294-329 distribution."""
    text = read_case("square")
    ref = T.load(text)
    xa = np.tile(ref.guesses, (512, 1))
    _, x, st = _batch_vs_oracle(E, text, xa)
    assert np.all(st["iterations"] == 7) and np.all(st["n_unsatisfied"] == 0)
    assert np.all(x == x[0])  # identical inputs -> bitwise identical outputs across teams
    xb = gen.keyed_uniform(0x657A707A, 4096, 8, -10000, 10000, integer=True)
    _, x, st = _batch_vs_oracle(E, text, xb)
    assert np.all(st["n_unsatisfied"] == 0)


def test_square_full_size_batch_properties(E):
    """65 536 systems (BASELINE configs[2]): size-independent properties -- every system satisfied, geometry is a
    4x4 axis-aligned square on (0,0),(4,4) -- and every one of them against the oracle (3 ... 21 iterations)."""
    text = read_case("square")
    ref = T.load(text)
    B = 65536
    x0 = gen.keyed_uniform(0x657A707A, B, 8, -10000, 10000, integer=True)
    sysobj = E.System(ref.constraints, ref.num_vars)
    x, st, _ = sysobj.solve_batch(x0)
    assert np.all(st["n_unsatisfied"] == 0) and np.all(st["converged"] == 1)
    a, b, c, d = x[:, 0:2], x[:, 2:4], x[:, 4:6], x[:, 6:8]
    assert np.all(np.abs(a) < 1e-4) and np.all(np.abs(c - 4.0) < 1e-4)
    assert np.all(np.abs(b - np.array([4.0, 0.0])) < 1e-4) and np.all(np.abs(d - np.array([0.0, 4.0])) < 1e-4)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0)
    assert np.array_equal(st["iterations"], it) and len(np.unique(it)) > 10
    assert np.array_equal(st["converged"], conv) and np.array_equal(st["n_unsatisfied"], nun)
    assert_x_close(x, xo)


@pytest.mark.parametrize("team", [1, 2, 4, 8, 16, 32, 64, 128, 256])
def test_team_shapes_agree(E, team):
    """Every team shape runs the same program; results must not depend on it beyond rounding of reductions."""
    text = read_case("two_rectangles")
    ref = T.load(text)
    x0 = ref.guesses[None, :] + gen.keyed_uniform(7, 300, ref.num_vars, -0.2, 0.2)
    sysobj, x, st = _batch_vs_oracle(E, text, x0, team_size=team)
    assert sysobj.info()["team_size"] == max(team, 2)  # 64 workspaces of this system do not fit a wavefront's LDS share
    if team == 1:  # one lane per system on a system that does fit
        text = read_case("arc_radius")
        ref = T.load(text)
        x0 = ref.guesses[None, :] + gen.keyed_uniform(8, 300, ref.num_vars, -0.1, 0.1)
        sysobj, x, st = _batch_vs_oracle(E, text, x0, team_size=1)
        assert sysobj.info()["team_size"] == 1


def test_mixed_topologies_batch(E):
    """BASELINE configs[4] at test size: [circle_tangent, parallelogram, arc_radius][i mod 3], jitter U(-0.1,0.1)."""
    for k, name in enumerate(["circle_tangent", "parallelogram", "arc_radius"]):
        text = read_case(name)
        ref = T.load(text)
        # side inference happens above the batch ABI (lib.rs:183-186): resolve from the file guesses
        recs = ref.constraints.copy()
        for i in range(len(recs)):
            recs[i] = O.set_from_initial_values(recs[i], ref.guesses)
        x0 = ref.guesses[None, :] + gen.keyed_uniform(0x657A707A + k, 30000, ref.num_vars, -0.1, 0.1)
        sysobj = E.System(recs, ref.num_vars)
        x, st, _ = sysobj.solve_batch(x0)
        rc, xo, it, conv, nun = O.solve_batch(recs, x0)
        assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
        assert np.array_equal(st["n_unsatisfied"], nun)
        if name == "parallelogram":
            # C and D are free (tests.rs:630-637: underconstrained 4..7) and held only by lambda ~ 1e-9..1e-12, which
            # amplifies last-bit differences of the evaluators by 1/lambda: the determined points match at 1e-6, the
            # free ones at the oracle's own sensitivity (3e-5 worst over 30 000 systems) and never beyond the reference's own
            # test tolerance (EPSILON = 1e-4, lib.rs:43; tests.rs:1161-1173), and both solutions satisfy every constraint
            # (n_unsatisfied above)
            assert_x_close(x[:, :4], xo[:, :4])
            assert_x_close(x[:, 4:], xo[:, 4:], rel=1e-4)
        else:
            assert_x_close(x, xo)


def test_massive_parallel_system_batch(E):
    """BASELINE configs[1]: 2000 x 2000, 2 iterations (README.md:36-38); replicas with jittered guesses."""
    text = T.gen_big_problem(500)
    ref = T.load(text)
    x0 = ref.guesses[None, :] + gen.keyed_uniform(11, 24, ref.num_vars, -0.25, 0.25)
    x0[0] = ref.guesses
    sysobj, x, st = _batch_vs_oracle(E, text, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.all(st["iterations"] == 2) and np.all(st["n_unsatisfied"] == 0)
    assert np.all(st["final_residual_inf"] <= 1e-9)
    info = sysobj.info()
    assert (info["nnz_j"], info["nnz_a"], info["nnz_l"], info["n_levels"]) == (2500, 2500, 2500, 2)
    assert info["team_mode"] == 3  # component-resident; the list-walk shapes on this system: test_gpu_components.py
    walk = E.System(ref.constraints, ref.num_vars, team_size=E.TEAM_AUTO_LISTS)
    info = walk.info()
    assert info["team_mode"] == 1 and info["n_partitions"] == info["team_size"] // 64  # wavefront-partitioned
    xw, stw, _ = walk.solve_batch(x0)
    assert np.array_equal(xw, x) and np.array_equal(stw["iterations"], st["iterations"])


def _chain_system(n_pts):
    """One connected component: a polyline with fixed first point, segment lengths and alternating directions."""
    cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
    guesses = [0.0, 0.0]
    for k in range(1, n_pts):
        a, b = (2 * (k - 1), 2 * k - 1), (2 * k, 2 * k + 1)
        cons.append(O.distance(a, b, 1.0))
        cons.append(O.horizontal(a, b) if k % 2 else O.vertical(a, b))
        guesses += [0.55 * k + 0.1, 0.45 * k - 0.1]
    return O.stack(cons), np.asarray(guesses)


@pytest.mark.parametrize("team", [0, "latency", 256, 512])
def test_single_large_component_uses_barrier_workgroup(E, team):
    """A system that cannot be partitioned (one connected component) runs with all lanes on one partition: a barrier
    workgroup when asked for (or when one solve's latency is what counts, as in solve()), else -- for batches -- one
    wavefront per system when its state fits."""
    recs, g = _chain_system(120)
    # (one solve of 240 variables takes the frontal shape since round 5, tests/test_gpu_fronts.py; the record walk is what
    # TEAM_LATENCY_RECORDS still asks for)
    assert E.System(recs, len(g), team_size=E.TEAM_AUTO_LATENCY).info()["team_mode"] == 5
    sysobj = E.System(recs, len(g), team_size=E.TEAM_LATENCY_RECORDS if team == "latency" else team)
    info = sysobj.info()
    assert info["n_components"] == 1 and info["n_partitions"] == 1
    # (automatic shapes walk records: team_mode 4 -- 128 lanes for a batch of 240 variables, more for one solve)
    assert info["team_mode"] == (4 if team in (0, "latency") else 2) and (team != 0 or info["team_size"] == 128)
    x0 = g[None, :] + gen.keyed_uniform(5, 6, len(g), -0.05, 0.05)
    x, st, _ = sysobj.solve_batch(x0)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0)
    assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
    assert np.array_equal(st["n_unsatisfied"], nun)
    assert_x_close(x, xo)


def test_long_polyline_gets_a_shallow_elimination_tree(E):
    """One connected component with a band-like graph (a polyline): eliminated in request order its elimination tree
    is a path -- 8000 levels for 4000 points, one or two barriers each.  Nested dissection brings it to ~32 levels
    (75 ms -> under 1 ms per solve) at 1.6x the entries of L; the answers stay the oracle's."""
    recs, g = _chain_system(4000)
    sysobj = E.System(recs, len(g))
    info = sysobj.info()
    assert info["n_components"] == 1 and info["team_mode"] == 4 and not info["workspace_in_lds"]  # (8000 variables: the wide record walk)
    assert info["n_levels"] <= 64 and info["nnz_l"] <= 2 * info["nnz_a"]
    x0 = g[None, :] + gen.keyed_uniform(3, 3, len(g), -0.02, 0.02)
    x, st, _ = sysobj.solve_batch(x0)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
    assert np.array_equal(st["n_unsatisfied"], nun)
    assert_x_close(x, xo)


@pytest.mark.parametrize("team", [128, 512, 1024])
def test_partitioned_workgroup_matches_barrier_workgroup(E, team):
    """two_rectangles has two components: the wavefront-partitioned mode must agree with every other mode."""
    text = read_case("two_rectangles")
    ref = T.load(text)
    x0 = ref.guesses[None, :] + gen.keyed_uniform(9, 64, ref.num_vars, -0.2, 0.2)
    sysobj, x, st = _batch_vs_oracle(E, text, x0, team_size=team)
    assert sysobj.info()["team_mode"] in (1, 2)


def test_linear_only_build_matches_the_general_build(E):
    """Topologies made of the nine linear kinds run a kernel built without the other evaluators (and sweep the
    constant Jacobian once); adding one satisfied non-linear constraint on fresh variables switches the same system to
    the general build.  Both must give the oracle's answer, bit for bit on the shared variables."""
    ref = T.load(T.gen_big_problem(40))
    n = ref.num_vars
    x0 = ref.guesses[None, :] + gen.keyed_uniform(77, 9, n, -0.25, 0.25)
    lin = E.System(ref.constraints, n, team_size=256)
    xa, sta, _ = lin.solve_batch(x0)
    extra = O.stack(list(ref.constraints) + [O.distance((n, n + 1), (n + 2, n + 3), 5.0)])
    x0b = np.concatenate([x0, np.tile([0.0, 0.0, 3.0, 4.0], (len(x0), 1))], axis=1)
    gen_sys = E.System(extra, n + 4, team_size=256)
    xb, stb, _ = gen_sys.solve_batch(x0b)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.array_equal(sta["iterations"], it) and np.array_equal(stb["iterations"], it)
    assert np.array_equal(xa, xo) and np.array_equal(xb[:, :n], xo)
    # weights other than 1 scale the constant Jacobian: still one sweep, still exact
    recs = O.stack(ref.constraints).copy()
    recs["weight"][::3] = 2.5
    xw, stw, _ = E.System(recs, n, team_size=256).solve_batch(x0)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.array_equal(stw["iterations"], it) and np.array_equal(stw["converged"], conv)
    assert_x_close(xw, xo)


@pytest.mark.parametrize("team", [0, 64, 256])
def test_variables_without_constraints_pass_through(E, team):
    """Guesses for variables no constraint mentions come back unchanged (the reference never touches them), in
    every team shape -- partitioned teams load and store per partition, so every variable must belong to one."""
    ref = T.load(T.gen_big_problem(100))
    n = ref.num_vars
    sysobj = E.System(ref.constraints, n + 7, team_size=team)
    x0 = np.concatenate([np.tile(ref.guesses, (5, 1)), np.arange(35.0).reshape(5, 7) + 100.0], axis=1)
    x, st, _ = sysobj.solve_batch(x0)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.array_equal(x, xo) and np.array_equal(x[:, n:], x0[:, n:])
    assert np.array_equal(st["iterations"], it)


def test_committed_massive_fixture_and_overconstrained_variant(E):
    for text in (read_case("massive_parallel_system"), T.gen_big_problem(200, True)):
        ref = T.load(text)
        got = E.solve_records(ref.constraints, ref.guesses)
        want = O.solve(ref.constraints, ref.guesses, linsolve=O.LINSOLVE_SPARSE)
        assert (got.error, got.iterations, got.converged, got.unsatisfied) == (0, want.iterations, want.converged, want.unsatisfied)
        assert_x_close(got.final_values, want.final_values)


def test_large_ladder_grid_team_and_global_workspace(E):
    """BASELINE configs[3] at reduced size: one large sparse system whose state does not fit one CU's LDS.  By default
    it is spread over a grid team (many workgroups, each with its share of the state in LDS, grid-wide reductions);
    with an explicit team size one workgroup solves it out of a global-memory workspace.  Same answer both ways."""
    text = T.gen_big_problem(12000)
    ref = T.load(text)
    want = O.solve(ref.constraints, ref.guesses, linsolve=O.LINSOLVE_SPARSE)
    x0 = np.tile(ref.guesses, (3, 1))
    x0[1] += 0.125
    x0[2] -= 0.25
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=O.LINSOLVE_SPARSE)
    for team, grid in ((0, True), (512, False)):
        sysobj = E.System(ref.constraints, ref.num_vars, team_size=team)
        info = sysobj.info()
        assert (info["grid_workgroups"] > 1) == grid
        assert info["workspace_in_lds"] == (1 if grid else 0)
        x, st, _ = sysobj.solve_batch(x0)
        assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
        assert np.all(st["n_unsatisfied"] == 0) and np.all(st["n_warnings"] == 0)
        assert_x_close(x, xo)
        # several systems per launch go through the same slots one after the other
        x2, st2, _ = sysobj.solve_batch(np.tile(x0, (9, 1)))
        assert np.array_equal(x2, np.tile(x, (9, 1)))
    assert (st["iterations"][0], bool(st["converged"][0])) == (want.iterations, want.converged)
    # weights other than 1 (side array of the packed records) and an unsatisfiable constraint, on the grid team
    recs = O.stack(ref.constraints).copy()
    recs["weight"][::3] = 2.5
    recs["weight"][1::7] = 0.5
    recs = O.stack(list(recs) + [O.fixed(int(recs["ids"][5][0]), 123.0)])  # contradicts that variable's Fixed
    sysobj = E.System(recs, ref.num_vars)
    assert sysobj.info()["grid_workgroups"] > 1
    x, st, mask = sysobj.solve_batch(x0, want_mask=True)
    for b in range(3):
        want = O.solve(recs, x0[b], linsolve=O.LINSOLVE_SPARSE)
        assert (int(st["iterations"][b]), bool(st["converged"][b])) == (want.iterations, want.converged)
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied and len(want.unsatisfied) >= 1
        assert_x_close(x[b], want.final_values)


def test_grid_team_with_nonlinear_constraints_and_warnings(E):
    """The over-constrained ladder (gen_big_problem.py N true: a distance per line, 60 000 rows) on a grid team of the
    general build: iteration counts and values as the oracle, and the Degenerate warnings of a start with every point
    coincident are counted across all workgroups and come back in the reference's order."""
    ref = T.load(T.gen_big_problem(12000, True))
    sysobj = E.System(ref.constraints, ref.num_vars)
    assert sysobj.info()["grid_workgroups"] > 1
    x0 = np.tile(ref.guesses, (3, 1))
    x0[1] += 0.05
    x0[2, :] = 1.0
    x, st, mask = sysobj.solve_batch(x0, want_mask=True)
    for b in range(3):
        want = O.solve(ref.constraints, x0[b], linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert (int(st["iterations"][b]), bool(st["converged"][b])) == (want.iterations, want.converged)
        assert int(st["n_warnings"][b]) == len(want.warnings)
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied
        assert_x_close(x[b], want.final_values)
    assert st["n_warnings"][2] == 12000
    got = E.solve_records(O.stack(ref.constraints), x0[2], warn_cap=1 << 16)
    assert got.warnings == want.warnings


def test_empty_batch_and_ragged_sizes(E):
    text = read_case("tiny")
    ref = T.load(text)
    sysobj = E.System(ref.constraints, ref.num_vars)
    x, st, _ = sysobj.solve_batch(np.zeros((0, ref.num_vars)))
    assert x.shape[0] == 0
    for B in (1, 3, 17, 63, 65, 1025):
        x0 = np.tile(ref.guesses, (B, 1))
        x, st, _ = sysobj.solve_batch(x0)
        assert np.all(st["iterations"] == 1) and np.all(np.abs(x) < 1e-4)


def test_degenerate_warnings_in_reference_order(E):
    """solver.rs:340-346,:385-391: one warning per degenerate evaluation, chronological."""
    center, start, end = (0, 1), (2, 3), (4, 5)
    reqs = [O.fixed(0, 0.0), O.fixed(1, 0.0), O.fixed(2, 0.0), O.fixed(3, 0.0), O.arc_length(center, start, end, 1.0),
            O.points_at_angle((0, 1), (2, 3), (4, 5), ("deg", 180.0))]
    guesses = [(0, 0.0), (1, 0.0), (2, 0.0), (3, 0.0), (4, 1.0), (5, 0.0)]
    got = GpuAdapter().solve(reqs, guesses)
    want = OracleAdapter().solve(reqs, guesses)
    assert got.warnings == want.warnings and len(want.warnings) > 2
    assert (got.iterations, got.converged, got.unsatisfied) == (want.iterations, want.converged, want.unsatisfied)


def test_nonfinite_and_singular_inputs_follow_the_reference_control_flow(E):
    """Cholesky failure => lambda x10 and a burnt iteration (newton.rs:93-99); NaN guesses never converge."""
    reqs = [O.distance((0, 1), (2, 3), 1.0)]
    for g in ([0.0, 0.0, 0.0, 0.0], [1e200, 0.0, -1e200, 0.0], [float("nan"), 0.0, 1.0, 1.0]):
        guesses = list(enumerate(g))
        got = GpuAdapter().solve(reqs, guesses, config=dict(initial_lambda=1e-30))
        want = OracleAdapter().solve(reqs, guesses, config=dict(initial_lambda=1e-30))
        assert (got.error, got.iterations, got.converged, got.unsatisfied) == (want.error, want.iterations, want.converged, want.unsatisfied), g
        assert np.array_equal(np.isnan(got.final_values), np.isnan(want.final_values))


def test_object_api_mirrors_the_crate(E):
    """README / lib.rs:48-78 doc example through the object API."""
    ids = E.IdGenerator()
    p, q = E.DatumPoint.new(ids), E.DatumPoint.new(ids)
    reqs = [E.ConstraintRequest.highest_priority(E.Constraint.Fixed(p.id_x(), 0.0)),
            E.ConstraintRequest.highest_priority(E.Constraint.Fixed(p.id_y(), 0.0)),
            E.ConstraintRequest.highest_priority(E.Constraint.Distance(p, q, 4.0))]
    guesses = [(p.id_x(), 0.0), (p.id_y(), -0.02), (q.id_x(), 4.39), (q.id_y(), 4.38)]
    out = E.solve(reqs, guesses, E.Config())
    assert out.is_satisfied() and out.converged()
    px, py = out.final_value_point(p)
    qx, qy = out.final_value_point(q)
    assert abs(px) < 1e-4 and abs(py) < 1e-4 and abs(np.hypot(qx - px, qy - py) - 4.0) < 1e-4
    with pytest.raises(E.FailureOutcome) as e:
        E.solve([E.ConstraintRequest.highest_priority(E.Constraint.Fixed(0, 0.0))], [], E.Config())
    assert e.value.error.code == -3 and (e.value.error.constraint_id, e.value.error.variable) == (0, 0)


def test_cli_equivalent_driver(E):
    """ezpz-cli/src/main.rs:240-300: stdout carries the problem size, iterations and the timing lines."""
    import subprocess
    from conftest import ROOT

    exe = os.path.join(ROOT, "ezpz_amd", "ezpz-amd")
    assert os.path.exists(exe), "build() must produce the CLI"
    for case, size in (("tiny", "Problem size: 4 rows, 4 vars"), ("arc_radius", "Problem size: 4 rows, 8 vars"),
                       ("circle", "Problem size: 5 rows, 5 vars")):
        path = os.path.join(GOLDEN, "test_cases", case, "problem.md")
        out = subprocess.run([exe, "-f", path, "--show-points"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert size in out.stdout
        assert "Iterations needed: " in out.stdout and "Solved up to priority: 0" in out.stdout
        assert "(mean over 100 iterations)" in out.stdout and "solves per second" in out.stdout
        assert "Points:" in out.stdout
    # stdin form (main.rs:225-238) and a parse error
    text = read_case("tiny")
    out = subprocess.run([exe, "-f", "-"], input=text, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "Iterations needed: 1" in out.stdout
    bad = subprocess.run([exe, "-f", "-"], input="nonsense", capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "Error" in bad.stderr


def test_random_systems_fuzz_parity(E):
    """The reference's fuzz target (fuzz/fuzz_targets/fuzz_target_1.rs) turned into a parity test: arbitrary
    constraint lists of every kind over a few variables (ids may repeat), arbitrary guesses and weights.  Wherever
    the oracle's own answer is stable under a one-ulp input perturbation, the HIP path must reproduce iterations,
    convergence, the unsatisfied list and the coordinates; everywhere the error code and NaN pattern must agree."""
    rng = np.random.default_rng(20240607)
    checked = unstable = determined_checked = 0
    kinds_seen = set()
    for trial in range(500):
        nvars = int(rng.integers(4, 13))
        ncons = int(rng.integers(1, 9))
        cons = []
        for _ in range(ncons):
            c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nvars)
            c["weight"] = float(rng.choice([1.0, 1.0, 1.0, 0.25, 3.0]))
            cons.append(c)
        guesses = rng.uniform(-8.0, 8.0, nvars)
        cfg = dict(max_iterations=int(rng.choice([35, 35, 10, 60])))
        want = OracleAdapter().solve(cons, list(enumerate(guesses)), cfg)
        got = GpuAdapter().solve(cons, list(enumerate(guesses)), cfg)
        assert got.error == want.error == 0
        assert np.array_equal(np.isnan(got.final_values), np.isnan(want.final_values)), trial
        # "stable": the oracle itself gives the same answer (a) from guesses perturbed by one ulp and (b) with the
        # constraints listed in reverse order, which only changes floating-point summation orders.  Stagnating
        # least-squares solves, whose accept/reject decisions hinge on the last bits of sum r^2, fail (b).
        bumped = guesses * (1.0 + rng.uniform(-1.0, 1.0, nvars) * 2.0 ** -52)
        again = OracleAdapter().solve(cons, list(enumerate(bumped)), cfg)
        rev = OracleAdapter().solve(cons[::-1], list(enumerate(guesses)), cfg)
        rev_unsat = sorted(len(cons) - 1 - i for i in rev.unsatisfied)
        scale = np.maximum(1.0, np.abs(want.final_values))
        stable = np.all(np.isfinite(want.final_values))
        for other, unsat in ((again, again.unsatisfied), (rev, rev_unsat)):
            stable = (stable and other.iterations == want.iterations and other.converged == want.converged
                      and unsat == want.unsatisfied
                      and np.max(np.abs(other.final_values - want.final_values) / scale) < 1e-9)
        if not stable:
            unstable += 1
            continue
        assert (got.converged, got.unsatisfied) == (want.converged, want.unsatisfied), trial
        if want.final_residual_inf <= 1e-8:
            # ended on the residual test (newton.rs:50-60): the iteration count is well defined
            assert got.iterations == want.iterations, trial
            assert len(got.warnings) == len(want.warnings), trial
        else:
            # ended on the step-size test (newton.rs:134-139) at a least-squares minimum: when ||d|| first drops
            # below 1e-12 depends on last-bit differences between the device and host libm; allow +-2
            assert abs(got.iterations - want.iterations) <= 2, trial
        if want.final_residual_inf <= 1e-8:
            assert abs(got.final_residual_inf - want.final_residual_inf) <= 1e-9, trial  # residual norm within 1e-9 abs
        else:  # unsatisfiable system: the residual at the (possibly non-unique) least-squares point, relative
            assert abs(got.final_residual_inf - want.final_residual_inf) <= 1e-5 * want.final_residual_inf, trial
        # Coordinates are compared where they are determined: rank J(x*) == n.  In an under-determined system the
        # null-space part of every step is (rounding noise)/lambda with lambda shrinking to 1e-14 after a few accepted
        # steps, so it is not reproducible across libm implementations -- only the residual is.
        J = np.zeros((sum(O.residual_dim(c) for c in cons), nvars))
        row = 0
        for c in cons:
            rows, _ = O.jacobian_rows(c, want.final_values)
            for r in rows:
                for i, pd in r:
                    J[row, i] += c["weight"] * pd
                row += 1
        determined = False
        if J.size and np.all(np.isfinite(J)):
            sv = np.linalg.svd(J, compute_uv=False)
            determined = len(sv) >= nvars and sv[nvars - 1] > 1e-7 * max(sv[0], 1e-300)
        if determined:
            assert_x_close(got.final_values, want.final_values)
            determined_checked += 1
            if want.final_residual_inf <= 1e-8 and trial % 3 == 0:
                # the other team shapes of the sub-wavefront kernel (one lane per system ... one wavefront per
                # system) on the same system, several systems per wavefront
                resolved = O.stack([O.set_from_initial_values(c, guesses) for c in cons])
                for team in (1, 4, 64):
                    xs, sts, _ = E.System(resolved, nvars, team_size=team).solve_batch(
                        np.tile(guesses, (9, 1)), E.Config(**cfg))
                    assert np.all(sts["iterations"] == want.iterations) and np.all(sts["converged"] == 1), (trial, team)
                    assert np.all(xs == xs[0])
                    assert_x_close(xs[0], want.final_values)
        checked += 1
        kinds_seen.update(int(c["kind"]) for c in cons)
    assert checked >= 150 and determined_checked >= 20 and len(kinds_seen) == O.NUM_KINDS, (checked, determined_checked, unstable, sorted(kinds_seen))


def _fixture_mosaic(copies, seed, with_perpendicular=False):
    """One big block-diagonal system: `copies` jittered replicas of each fully determined reference fixture, variable
    ids shifted so that the replicas are independent components (11 kinds between them).  LM accepts a step for the
    whole system or not at all, so the mosaic takes 7 iterations (the slowest member alone: `square`, 7) -- or, with
    `perpendicular` (15 alone) in the mix, 81 iterations of accepted and rejected steps."""
    names = ["coincident", "symmetric", "midpoint", "tiny", "circle", "circle_center", "circle_tangent",
             "two_rectangles", "nonsquare", "square", "chamfer_square", "angle_parallel"]
    if with_perpendicular:
        names.append("perpendicular")
    recs, guesses, off = [], [], 0
    rng = np.random.default_rng(seed)
    for rep in range(copies):
        for name in names:
            ref = T.load(read_case(name))
            for c in ref.constraints:
                c = O.set_from_initial_values(c, ref.guesses).copy()
                c["ids"] = c["ids"] + off  # unused id slots become `off`: a valid variable, never dereferenced
                recs.append(c)
            guesses.append(ref.guesses + rng.uniform(-0.02, 0.02, ref.num_vars))
            off += ref.num_vars
    return O.stack(recs), np.concatenate(guesses)


@pytest.mark.parametrize("copies,team,perp", [(6, 0, False), (6, 256, False), (40, 0, False), (40, 512, False),
                                              (3, 0, True), (40, 0, True)])
def test_mosaic_of_fixtures_in_workgroup_and_grid_teams(E, copies, team, perp):
    """The general (all 25 kinds) build of the workgroup kernels on a system with many kinds per partition: packed
    records with several partial-slot patterns, the staged-lists and global-workspace variants, and a grid team."""
    recs, g = _fixture_mosaic(copies, seed=copies, with_perpendicular=perp)
    assert len(set(int(k) for k in recs["kind"])) >= 11
    # (a dozen classes of components: too many rows of state for the component-resident shape, whose multi-class case
    # is test_gpu_components.py::test_mixed_classes_with_nonlinear_members)
    sysobj = E.System(recs, len(g), team_size=team or E.TEAM_AUTO_LISTS)
    info = sysobj.info()
    assert info["team_mode"] in (1, 2)
    if copies == 40:
        assert (info["grid_workgroups"] > 1) == (team == 0)
    cfg = dict(max_iterations=120)
    x0 = np.stack([g, g + 0.01, g - 0.015])
    x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
    # run-to-run determinism: partitions must own every row of their constraints (a two-row constraint's rows can sit
    # in different blocks of JtJ; when they were split over wavefronts this differed from run to run)
    for _ in range(3):
        x2, st2, _ = sysobj.solve_batch(np.tile(x0, (4, 1)), E.Config(**cfg))
        assert np.array_equal(x2, np.tile(x, (4, 1))) and np.array_equal(st2["iterations"], np.tile(st["iterations"], 4))
    for b in range(3):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        # (40 copies with `perpendicular` do not get there in 120 iterations, on the oracle either: same 120 steps)
        assert want.error == 0 and want.converged == (not (perp and copies == 40))
        assert (int(st["iterations"][b]), bool(st["converged"][b])) == (want.iterations, want.converged), b
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied
        assert int(st["n_warnings"][b]) == len(want.warnings)
        assert abs(float(st["final_residual_inf"][b]) - want.final_residual_inf) <= 1e-9 * max(1.0, want.final_residual_inf)
        assert_x_close(x[b], want.final_values)


@pytest.mark.parametrize("team,ncomp", [(0, 120), (128, 120), (512, 120), (0, 1200), (512, 1200)])
def test_random_block_systems_are_deterministic_and_match_the_oracle(E, team, ncomp):
    """Workgroup teams on big systems made of random little components of all 25 kinds (the fuzz generator's, one
    block of variables each): bitwise the same answer on every run and for every system of a batch (a wavefront that
    read another's half-written data would show here), error-free, and -- the LM path of such a system hinges on
    rank-deficient blocks held by lambda only -- the same residual as the oracle where the oracle converges."""
    rng = np.random.default_rng(4242 + team + ncomp)
    for trial in range(6 if ncomp < 1000 else 2):
        recs, guesses, off = [], [], 0
        for comp in range(int(rng.integers(3 * ncomp // 4, 5 * ncomp // 4))):
            nv = int(rng.integers(4, 11))
            for _ in range(int(rng.integers(1, 7))):
                c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nv)
                c["ids"] = c["ids"] + off
                recs.append(c)
            guesses.append(rng.uniform(-6.0, 6.0, nv))
            off += nv
        recs, g = O.stack(recs), np.concatenate(guesses)
        sysobj = E.System(recs, len(g), team_size=team or E.TEAM_AUTO_LISTS)
        assert sysobj.info()["team_mode"] in (1, 2)
        if ncomp >= 1000:
            assert (sysobj.info()["grid_workgroups"] > 1) == (team == 0)
        cfg = dict(max_iterations=25)
        x0 = np.tile(g, (6, 1))
        x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
        assert np.all(x == x[0]) or np.array_equal(np.isnan(x), np.isnan(np.tile(x[0], (6, 1))))
        assert len(set(st["iterations"].tolist())) == 1 and len(set(st["n_warnings"].tolist())) == 1
        for _ in range(2):
            x2, st2, _ = sysobj.solve_batch(x0, E.Config(**cfg))
            assert np.array_equal(x2, x, equal_nan=True) and np.array_equal(st2["iterations"], st["iterations"])
        want = O.solve(recs, g, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 18)
        assert want.error == 0
        assert np.array_equal(np.isnan(x[0]), np.isnan(want.final_values))
        if want.converged and want.final_residual_inf <= 1e-8:
            assert bool(st["converged"][0]) and float(st["final_residual_inf"][0]) <= 1e-8


LISTS = 0xFFFFFFFE  # E.TEAM_AUTO_LISTS


@pytest.mark.parametrize("npts,team,mode", [(90, LISTS, 0), (150, LISTS, 2), (150, 256, 2), (400, LISTS, 2), (1500, LISTS, 2),
                                            (60, 0, 4), (90, 0, 4), (150, 0, 4), (400, 0, 4), (1500, 0, 4)])
def test_random_connected_sketch_in_one_wavefront_or_barrier_workgroup(E, npts, team, mode):
    """One connected component of mixed kinds (a random polyline-like sketch: every point tied to its predecessors by
    one or two random constraints).  The list-walk shapes (`TEAM_AUTO_LISTS`, what batches ran on before the record walk):
    one wavefront (180 variables), a lean 128-lane workgroup whose lists stay in
    global memory (300), the barrier workgroup with staged lists (300 on 256 lanes), with its workspace in LDS (800)
    and in global memory (3000).  The automatic batch shape: the
    record walk (team_mode 4) on one wavefront (120, 180 variables), on 128 lanes (300), on 512 (800) and -- in its wide
    form, 32-bit addresses into a workspace in global memory -- on 3000 variables.  Deterministic
    from run to run, and -- it is fully determined by construction -- the oracle's answer."""
    recs, g = gen.connected_sketch(npts, 77 + npts + (team if team < 1024 else 0))
    sysobj = E.System(recs, len(g), team_size=team)
    info = sysobj.info()
    assert info["n_components"] == 1 and info["team_mode"] == mode and info["n_partitions"] == 1
    if (npts, team) == (150, LISTS):
        assert info["team_size"] == 128 and not info["program_in_lds"]
    if mode == 4:
        assert info["team_size"] == {60: 64, 90: 128, 150: 128, 400: 256, 1500: 512}[npts]  # (800 variables: J in global memory, two per CU)
        assert info["workspace_in_lds"] == (npts < 1500)
    x0 = np.tile(g, (5, 1))
    cfg = dict(max_iterations=60)
    x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
    assert np.all(x == x[0]) and len(set(st["iterations"].tolist())) == 1
    x2, st2, _ = sysobj.solve_batch(x0, E.Config(**cfg))
    assert np.array_equal(x2, x) and np.array_equal(st2["iterations"], st["iterations"])
    want = O.solve(recs, g, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
    assert want.error == 0 and want.converged and not want.unsatisfied
    assert (int(st["iterations"][0]), bool(st["converged"][0])) == (want.iterations, True)
    assert not mask.any() and int(st["n_warnings"][0]) == len(want.warnings)
    assert_x_close(x[0], want.final_values)


@pytest.mark.parametrize("k,npts", [(4, 40), (7, 25), (3, 150), (40, 8)])
def test_a_few_sketches_in_one_system_walk_records_as_one_partition(E, k, npts):
    """A document of several sketches: k connected components of 50 ... 300 variables in ONE system.  The record walk needs
    levels, not connectivity: the automatic shapes make them one partition (team_mode 4; shape.cpp, analyze_into: up to 127 components)
    where the list-walk shapes give each wavefront a balanced share (`TEAM_AUTO_LISTS`).  Both against the oracle on jittered
    starts, and against each other."""
    parts, guesses, off = [], [], 0
    for c in range(k):
        recs, g = gen.connected_sketch(npts, 7300 + 31 * c + npts)
        r = recs.copy()
        for i in range(len(r)):
            r["ids"][i][:O.KIND_NUM_IDS[int(r["kind"][i])]] += off
        parts.append(r)
        guesses.append(g)
        off += len(g)
    recs, g = np.concatenate(parts), np.concatenate(guesses)
    n = len(g)
    x0 = g[None, :] + gen.keyed_uniform(k * 1000 + npts, 9, n, -0.03, 0.03)
    x0[0] = g
    cfg = dict(max_iterations=50)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    walk = E.System(recs, n)
    lists = E.System(recs, n, team_size=E.TEAM_AUTO_LISTS)
    wi, li = walk.info(), lists.info()
    assert wi["n_components"] == k and wi["team_mode"] == 4 and wi["n_partitions"] == 1, wi
    assert li["team_mode"] in (1, 2), li
    from sensitivity import assert_batch_matches_oracle
    one = E.System(recs, n, team_size=E.TEAM_LATENCY_RECORDS)  # (one solve's shape before the fronts: the same walk on more lanes)
    assert one.info()["team_mode"] == 4 and one.info()["n_partitions"] == 1
    for sysobj in (walk, lists, one):
        x, st, _ = sysobj.solve_batch(x0, E.Config(**cfg))
        # (some of these starts take 40 LM iterations: counts inside the oracle's own spread, tests/sensitivity.py)
        assert_batch_matches_oracle(recs, x0, x, st["iterations"], st["converged"], O.Config(**cfg), oracle_result=(xo, it, conv),
                                    what=(k, npts))
        assert np.array_equal(st["n_unsatisfied"], nun)


@pytest.mark.parametrize("npts,latency_mode,batch_mode", [(10, 0, 0), (16, 4, 0), (25, 4, 0), (32, 4, 4)])
def test_small_connected_sketch_walks_records_where_it_pays(E, npts, latency_mode, batch_mode):
    """One connected sketch of 20 ... 64 variables: one solve walks records from 25 variables (team_mode 4),
    batches from 57 (below that two to four systems share a wavefront on the sub-wavefront teams: shape.cpp, analyze_into).
    Both shapes against the oracle, jittered starts and a NaN start."""
    recs, g = gen.connected_sketch(npts, 4100 + npts)
    n = len(g)
    x0 = g[None, :] + gen.keyed_uniform(npts, 8, n, -0.05, 0.05)
    x0[0] = g
    x0[7, 1] = float("nan")
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    for team, mode in ((E.TEAM_LATENCY_RECORDS, latency_mode), (0, batch_mode)):
        sysobj = E.System(recs, n, team_size=team)
        info = sysobj.info()
        assert info["n_components"] == 1 and info["team_mode"] == mode, (team, info)
        assert mode != 4 or info["team_size"] == (128 if team else 64)  # (one solve: two wavefronts, as fast as one)
        x, st, _ = sysobj.solve_batch(x0)
        assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
        assert np.array_equal(st["n_unsatisfied"], nun)
        assert np.array_equal(np.isnan(x), np.isnan(xo))
        assert_x_close(x[:7], xo[:7])


@pytest.mark.parametrize("npts", [60, 150, 260, 400, 1200])
@pytest.mark.parametrize("shape", ["phases", "records"])
def test_connected_sketch_latency_shape_with_dense_root_block(E, npts, shape):
    """The launch shape of one solve (`TEAM_AUTO_LATENCY`, what `ezpz_solve` asks for) runs the linear solve of a connected
    sketch whose state fits the LDS as a RECORD WALK (records.cpp: build_records; lm_kernel.hip.hpp: REC builds): every level of
    the elimination tree one or more rounds in which a group of lanes owns one entry, the lanes' operand addresses ready in
    records requested a round ahead (`shape` "records": team_mode 4; 2400 variables do not fit the LDS and walk the wide form).
    `TEAM_LATENCY_PHASES` keeps what that shape did before: it ends the elimination
    with dense phases: runs of levels at the top of the elimination tree whose columns fall into independent
    blocks of <= 16 (the last one the root block: the last <= 16 columns), each block's Schur complement gathered by
    all lanes and factorised in one wavefront's registers (records.cpp: make_dense_phases; lm_kernel.hip.hpp: dense
    phases), with the program staged in LDS (120-520 variables), read from global memory (800) and with the workspace
    in global memory too (2400).  Against the oracle (iteration counts,
    flags, masks, coordinates at 1e-6), against the plain level walk (an explicit team size keeps it), from run to
    run, from a NaN start (every pivot fails: lambda grows, the iterations burn) and on an inconsistent system."""
    recs, g = gen.connected_sketch(npts, 500 + npts)
    n = len(g)
    latency = E.TEAM_LATENCY_RECORDS if shape == "records" else E.TEAM_LATENCY_PHASES
    lat = E.System(recs, n, team_size=latency)
    walk = E.System(recs, n, team_size=512)
    li, wi = lat.info(), walk.info()
    assert li["n_components"] == 1
    if shape == "records":  # (2400 variables: the wide form, 32-bit addresses into a workspace in global memory)
        assert li["team_mode"] == 4 and li["workspace_in_lds"] == (npts < 1200) and li["n_levels"] == wi["n_levels"]
    else:
        assert li["team_mode"] == 2
        assert li["n_levels"] + 4 <= wi["n_levels"], (li["n_levels"], wi["n_levels"])  # the top levels became the block
    cfg = dict(max_iterations=40)
    x0 = g[None, :] + gen.keyed_uniform(npts, 6, n, -0.02, 0.02)
    x0[0] = g
    x0[5, 3] = float("nan")
    x, st, mask = lat.solve_batch(x0, E.Config(**cfg), want_mask=True)
    x2, st2, _ = lat.solve_batch(x0, E.Config(**cfg))
    assert np.array_equal(x2, x, equal_nan=True) and np.array_equal(st2["iterations"], st["iterations"])
    xw, stw, maskw = walk.solve_batch(x0, E.Config(**cfg), want_mask=True)
    for b in range(6):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0
        assert bool(st["converged"][b]) == want.converged and np.nonzero(mask[b])[0].tolist() == want.unsatisfied, b
        if b == 5:
            assert not want.converged and int(st["iterations"][b]) == want.iterations == 40
            assert np.array_equal(np.isnan(x[b]), np.isnan(want.final_values))
            continue
        assert want.converged and abs(int(st["iterations"][b]) - want.iterations) <= (0 if want.iterations <= 12 else 2), b
        assert int(st["n_warnings"][b]) == len(want.warnings)
        assert_x_close(x[b], want.final_values)
        assert int(stw["iterations"][b]) == int(st["iterations"][b]) and np.array_equal(maskw[b], mask[b])
        assert_x_close(x[b], xw[b], 1e-9)
    # an inconsistent sketch (the last point pinned away from where its distances put it): least-squares exit, same
    # unsatisfied rows as the oracle's
    bad = O.stack(list(recs) + [O.fixed(n - 2, float(g[n - 2]) + 0.4)])
    lat2 = E.System(bad, n, team_size=latency)
    xb, stb, maskb = lat2.solve_batch(g[None, :], E.Config(max_iterations=60), want_mask=True)
    want = O.solve(bad, g, O.Config(max_iterations=60), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
    assert bool(stb["converged"][0]) == want.converged and np.nonzero(maskb[0])[0].tolist() == want.unsatisfied
    # (a least-squares exit is the step test on a flat valley floor: the count hinges on the last bits)
    assert abs(float(stb["final_residual_inf"][0]) - want.final_residual_inf) <= 1e-6 * max(1.0, want.final_residual_inf)
    assert_x_close(xb[0], want.final_values, 1e-4)


def test_sketch_whose_workspace_fills_the_lds_keeps_its_dense_phases(E):
    """1408 variables: 156 KB of workspace in LDS, the program read from global memory, and the few KB left go to the dense
    panels rather than to the level staging buffer (shape.cpp, analyze_into).  Against the oracle."""
    recs, g = gen.connected_sketch(704, 288)
    s = E.System(recs, len(g), team_size=E.TEAM_LATENCY_PHASES)
    plain = E.System(recs, len(g), team_size=512)
    assert s.info()["workspace_in_lds"] == 1 and s.info()["n_levels"] + 5 <= plain.info()["n_levels"]
    # (the record walk needs the factor's diagonal a second time and its descriptors in LDS: no room here, so one solve's
    # automatic shape is the same one)
    assert E.System(recs, len(g), team_size=E.TEAM_LATENCY_RECORDS).info()["team_mode"] == 2
    x0 = np.stack([g, g + 0.01])
    cfg = dict(max_iterations=40)
    x, st, mask = s.solve_batch(x0, E.Config(**cfg), want_mask=True)
    for b in range(2):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0 and bool(st["converged"][b]) == want.converged
        assert abs(int(st["iterations"][b]) - want.iterations) <= (0 if want.iterations <= 12 else 2)
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied
        assert_x_close(x[b], want.final_values)


def _hub_sketch(npts, seed, hub_last):
    """`npts` points each tied to one hub point (a distance and a horizontal distance): one connected component.  The
    hub's two variables come last or -- the way a sketch dimensioned from its origin is written -- first; the layout
    and the guesses are the same either way."""
    rng = np.random.default_rng(seed)
    place = (lambda k: npts if k == 0 else k - 1) if hub_last else (lambda k: k)  # logical point (hub = 0) -> slot
    pt = lambda k: (2 * place(k), 2 * place(k) + 1)
    true = np.zeros((npts + 1, 2))
    true[0] = (1.0, -2.0)
    true[1:] = true[0] + rng.uniform(1.0, 9.0, (npts, 2)) * rng.choice([-1.0, 1.0], (npts, 2))
    guess = true + rng.uniform(-0.05, 0.05, true.shape)
    cons = [O.fixed(pt(0)[0], 1.0), O.fixed(pt(0)[1], -2.0)]
    for k in range(1, npts + 1):
        cons += [O.distance(pt(k), pt(0), float(np.hypot(*(true[k] - true[0])))),
                 O.horizontal_distance(pt(k), pt(0), float(true[k][0] - true[0][0]))]
    g = np.zeros(2 * (npts + 1))
    for k in range(npts + 1):
        g[2 * place(k):2 * place(k) + 2] = guess[k]
    return O.stack(cons), g


@pytest.mark.parametrize("team", [0, 256])
def test_hub_sketch_with_one_level_wider_than_the_team_and_its_staging_buffer(E, team):
    """3000 points each tied to one hub point: one connected component whose first elimination level holds all 6000
    leaf columns -- wider than any team (so that level runs as the two-phase walk) and larger than the LDS buffer a
    level's lists are staged in (so it is walked from global memory) -- followed by the hub's narrow, long-list levels
    (staged, one phase, lists shared by groups of lanes).  The answer is the oracle's."""
    recs, g = _hub_sketch(3000, 5150 + team, hub_last=True)
    sysobj = E.System(recs, len(g), team_size=team)
    info = sysobj.info()
    assert info["n_components"] == 1 and info["team_mode"] == 2 and info["n_partitions"] == 1
    assert info["n_levels"] <= 4 and not info["program_in_lds"]
    x0 = np.stack([g, g + 0.01])
    x, st, mask = sysobj.solve_batch(x0, want_mask=True)
    x2, st2, _ = sysobj.solve_batch(x0)
    assert np.array_equal(x2, x) and np.array_equal(st2["iterations"], st["iterations"])
    for b in range(2):
        want = O.solve(recs, x0[b], linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0 and want.converged and not want.unsatisfied
        assert (int(st["iterations"][b]), bool(st["converged"][b])) == (want.iterations, True)
        assert not mask[b].any()
        assert_x_close(x[b], want.final_values)


@pytest.mark.parametrize("npts,hub_last,shape", [(40, True, (0, 64)), (100, False, (0, 64)), (300, True, (2, 128)),
                                                 (40, True, (4, 64)), (100, False, (4, 128)), (300, True, (4, 256))])
def test_small_hub_sketch_on_one_wavefront_or_lean_workgroup(E, npts, hub_last, shape):
    """The hub sketch at the sizes the list-walk shapes run on one wavefront per system (82, 202 variables) or on a lean
    128-lane workgroup (602): its first level is wider than the team (two-phase walk) and, from 100 points, larger than the
    team's staging buffer (walked from global memory); the hub's levels are one phase with lists shared by groups of
    lanes.  And on the automatic batch shape (record walk, team_mode 4): the wide level is several rounds."""
    recs, g = _hub_sketch(npts, 77 + npts, hub_last)
    sysobj = E.System(recs, len(g), team_size=0 if shape[0] == 4 else E.TEAM_AUTO_LISTS)
    info = sysobj.info()
    assert info["n_components"] == 1 and (info["team_mode"], info["team_size"]) == shape
    x0 = g[None, :] + gen.keyed_uniform(41, 64, len(g), -0.05, 0.05)
    x, st, mask = sysobj.solve_batch(x0, want_mask=True)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, linsolve=O.LINSOLVE_SPARSE)
    assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv)
    assert np.array_equal(st["n_unsatisfied"], nun) and not mask.any()
    assert_x_close(x, xo)


def test_hub_declared_first_is_eliminated_last(E):
    """The same sketch with the hub's variables numbered first, as a sketch dimensioned from its origin is written: in
    request order the hub is eliminated first and L fills in completely (18 M entries, 3.6e10 multiply-adds: `system
    too large`).  The graph is too compact for nested dissection (everything is two steps from the hub), so this is
    minimum degree's case -- on 6002 vertices.  Same factor size and the same answer as with the hub numbered last."""
    recs_f, g_f = _hub_sketch(3000, 5150, hub_last=False)
    recs_l, g_l = _hub_sketch(3000, 5150, hub_last=True)
    first, last = E.System(recs_f, len(g_f)), E.System(recs_l, len(g_l))
    assert first.info()["nnz_l"] == last.info()["nnz_l"] == first.info()["nnz_a"]  # no fill at all
    xf, stf, _ = first.solve_batch(g_f[None, :])
    xl, stl, _ = last.solve_batch(g_l[None, :])
    assert bool(stf["converged"][0]) and int(stf["iterations"][0]) == int(stl["iterations"][0])
    assert_x_close(xf[0, 2:], xl[0, :-2])   # leaves
    assert_x_close(xf[0, :2], xl[0, -2:])   # hub


def test_batch_solve_with_priorities_and_inferred_sides(E):
    """lib.rs:148-263 per system of a batch: sides inferred from each system's own guesses, cumulative priority tiers
    from the original guesses, last fully satisfied tier wins (tests.rs:49-106 semantics, batched)."""
    p0, p1, center, radius = (0, 1), (2, 3), (4, 5), 6
    reqs = [
        O.fixed(1, 3.0), O.fixed(3, 3.0), O.circle_radius(center, radius, 1.5),
        O.line_tangent_to_circle(p0, p1, center, radius, O.SIDE_UNDEFINED),
        O.fixed(4, 2.0, priority=1),              # satisfiable refinement
        O.fixed(5, 100.0, priority=2),            # contradicts the tangent: tier 2 is unsatisfied -> tier 1 is returned
        O.fixed(0, 0.0, priority=1, weight=2.0),
    ]
    rng = np.random.default_rng(5)
    B = 96
    x0 = np.tile(np.array([0.0, 3.0, 5.0, 3.0, 2.0, 1.5, 1.5]), (B, 1)) + rng.uniform(-0.2, 0.2, (B, 7))
    x0[::2, 5] += 3.0  # half of the circles start above the line (Left), half below (Right)
    x, st, prio, mask = E.solve_batch(O.stack(reqs), x0, want_mask=True)
    sides = set()
    for b in range(B):
        want = OracleAdapter().solve(reqs, list(enumerate(x0[b])))
        assert want.error == 0
        assert (int(st["iterations"][b]), bool(st["converged"][b])) == (want.iterations, want.converged), b
        assert int(prio[b]) == want.priority_solved == 1
        assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied == []
        assert_x_close(x[b], want.final_values)
        sides.add(bool(x[b, 5] > 3.0))
    assert sides == {True, False}
    # a first-tier topology error fails the call like FailureOutcome
    with pytest.raises(E.NonLinearSystemError) as e:
        E.solve_batch(O.stack([O.fixed(0, 1.0), O.fixed(9, 1.0)]), np.zeros((4, 3)))
    assert (e.value.code, e.value.constraint_id, e.value.variable) == (-3, 1, 9)
    # no requests: guesses are echoed
    x, st, prio, _ = E.solve_batch(np.zeros(0, dtype=E.CONSTRAINT_DTYPE), x0[:5])
    assert np.array_equal(x, x0[:5]) and np.all(st["converged"] == 1) and np.all(st["iterations"] == 0)


# ---- FreedomAnalysis (solver/find_dof.rs) on the device ----------------------------------------------------------------
def _freedom_vs_oracle(E, recs, n, X, atol=1e-9):
    """Device analysis of every row of X against the oracle's dense QR of the same (device-evaluated) Jacobian."""
    sysobj = E.System(recs, n)
    mask, part = sysobj.freedom_batch(X)
    _, J, _ = sysobj.eval_batch(X)
    for b in range(X.shape[0]):
        under, want = O.freedom_analysis_dense(J[b])
        assert np.allclose(part[b], want, atol=atol), (b, float(np.abs(part[b] - want).max()))
        assert np.nonzero(mask[b])[0].tolist() == under, b
    return sysobj, mask, part


@pytest.mark.parametrize("case", ["underdetermined_lines", "parallelogram", "arc_radius", "perpdist", "square",
                                  "two_rectangles", "circle_tangent", "chamfer_square", "arc_equidistant"])
def test_freedom_participation_matches_oracle_on_fixtures(E, case):
    ref = T.load(read_case(case))
    want = O.solve(ref.constraints, ref.guesses, analysis=True)
    X = want.final_values[None, :] + gen.keyed_uniform(21, 33, ref.num_vars, -0.05, 0.05)
    X[0] = want.final_values
    # ezpz_system_* take side-resolved constraints
    recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
    _, mask, _ = _freedom_vs_oracle(E, recs, ref.num_vars, X)
    assert np.nonzero(mask[0])[0].tolist() == want.underconstrained


def test_freedom_lane_mode_on_block_systems(E):
    """Many small components: one lane per (system, component).  An under-constrained variant of the
    massive_parallel_system generator (every 7th block loses a constraint) against the oracle's dense QR."""
    ref = T.load(T.gen_big_problem(60))
    keep = [c for i, c in enumerate(ref.constraints) if not (i % 28 == 3)]
    n = ref.num_vars
    X = ref.guesses[None, :] + gen.keyed_uniform(31, 5, n, -0.2, 0.2)
    sysobj, mask, part = _freedom_vs_oracle(E, O.stack(keep), n, X)
    assert mask.any() and not mask.all()
    # full size (BASELINE configs[1]): fully constrained, so nothing is free; and the solved batch agrees
    ref = T.load(T.gen_big_problem(500))
    sysobj = E.System(ref.constraints, ref.num_vars)
    x, st, _ = sysobj.solve_batch(np.tile(ref.guesses, (64, 1)))
    mask, part = sysobj.freedom_batch(x)
    assert not mask.any() and np.all(part == 0.0)


@pytest.fixture
def pivoted_qr(monkeypatch):
    """The tests of the pivoted QR's layouts: systems the fronts serve would be analysed by null-space probes (freedom.hip:
    freedom_by_probes, tested below) and never reach them."""
    monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")


def test_freedom_workgroup_mode_with_global_workspace(E, pivoted_qr):
    """One 240-variable component: the dense workspace (1.4 MB) lives in global memory, lanes work on columns."""
    recs, g = _chain_system(120)
    recs = recs[:-1]  # drop the last direction constraint: the last point may swing on its circle
    X = g[None, :] + gen.keyed_uniform(41, 3, len(g), -0.05, 0.05)
    sysobj = E.System(recs, len(g))
    x, st, _ = sysobj.solve_batch(X)
    assert np.all(st["n_unsatisfied"] == 0)
    _, mask, _ = _freedom_vs_oracle(E, recs, len(g), x, atol=1e-8)
    assert np.all(mask[:, -2:].any(axis=1)) and np.all(mask[:, :-2] == 0)  # only the last point is free


def test_freedom_of_one_large_component_spread_over_the_device(E, pivoted_qr):
    """A 300-variable connected sketch that lost its last three constraints: one component, so its pivoted QR runs over the
    whole device (one cooperative launch for all Householder steps, freedom.hip.hpp:fr_qr_kernel) before the ordinary
    kernel takes rank, null space and participation; three systems side by side.  Against the oracle's dense QR."""
    recs, g = gen.connected_sketch(150, 4242)
    recs = recs[:-3]
    X = g[None, :] + gen.keyed_uniform(43, 3, len(g), -0.02, 0.02)
    sysobj = E.System(recs, len(g))
    x, st, _ = sysobj.solve_batch(X, E.Config(max_iterations=60))
    _, mask, part = _freedom_vs_oracle(E, recs, len(g), x, atol=1e-8)
    assert mask.any() and np.all(mask[:, :200] == 0)  # only points near the loose end are free
    # the same analysis for every system of a batch larger than one launch's worth is consistent
    mask2, part2 = sysobj.freedom_batch(np.repeat(x[:1], 5, axis=0))
    assert np.all(mask2 == mask[0]) and np.allclose(part2, part[0], atol=1e-12)


@pytest.mark.parametrize("points,repeated", [(150, 0), (400, 0), (200, 60)])
def test_freedom_wide_cooperative_launch_and_launch_chain_agree(E, points, repeated, monkeypatch, pivoted_qr):
    """The three routes of the WIDE layout -- the matrix resident in the workgroups' registers for the whole factorisation
    (fr_qrc_kernel, the default up to 2048 rows), the trailing matrix streamed once per step inside one cooperative launch
    (fr_qr_kernel, EZPZ_FREEDOM_CHAIN=2), and the chain of one launch pair per step they fall back to when the device
    cannot hold a system's workgroups at once (EZPZ_FREEDOM_CHAIN=1) -- sum in different fixed orders: the same
    underconstrained set, participation equal to rounding, the default against the oracle's dense QR (300 and 800
    variables), and bitwise repeatable."""
    recs, g = gen.connected_sketch(points, 4242)
    recs = recs[:-3]
    if repeated:  # more rows than variables: some constraints twice (dependent rows; the factorisation runs out of pivots early)
        recs = np.concatenate([recs, recs[:repeated]])
    sysobj = E.System(recs, len(g))
    x, st, _ = sysobj.solve_batch(g[None, :] + gen.keyed_uniform(53, 2, len(g), -0.01, 0.01), E.Config(max_iterations=60))
    monkeypatch.delenv("EZPZ_FREEDOM_CHAIN", raising=False)
    _, mask, part = _freedom_vs_oracle(E, recs, len(g), x, atol=1e-8)
    mask_again, part_again = sysobj.freedom_batch(x)
    assert np.array_equal(part, part_again) and np.array_equal(mask, mask_again)
    for route in ("2", "1"):  # the streaming cooperative kernel (fr_qr_kernel), then the chain of launch pairs
        monkeypatch.setenv("EZPZ_FREEDOM_CHAIN", route)
        mask_other, part_other = sysobj.freedom_batch(x)
        assert np.array_equal(mask_other, mask) and mask.any(), route
        assert np.allclose(part_other, part, atol=1e-10), route


def test_freedom_two_large_components_on_one_workgroup(E, pivoted_qr):
    """Two 120-variable chains in one system: the larger workspace of the two is factorised over the whole device (its QR
    left in the global workspace for the ordinary kernel, which takes that component first), the other in sequence by one
    workgroup out of the same workspace (rows contiguous, norms summed during the update)."""
    recs, g = _chain_system(60)
    recs2 = recs.copy()
    recs2["ids"] = recs2["ids"] + len(g)
    both = np.concatenate([recs[:-1], recs2[:-2]])  # each chain loses its last constraint(s)
    gg = np.concatenate([g, g + 0.3])
    X = gg[None, :] + gen.keyed_uniform(47, 2, len(gg), -0.03, 0.03)
    sysobj = E.System(both, len(gg))
    x, st, _ = sysobj.solve_batch(X)
    _, mask, _ = _freedom_vs_oracle(E, both, len(gg), x, atol=1e-8)
    assert mask[:, len(g) - 2:len(g)].any() and mask[:, -2:].any()


@pytest.mark.parametrize("npts,drop,team", [(40, 0, 0), (40, 3, 0), (150, 0, 0), (150, 1, 0), (150, 3, "latency"), (400, 2, 0), (400, 4, "latency"),
                                             (850, 3, "latency"), (1000, 0, "latency")])
def test_freedom_by_null_space_probes_equals_the_oracle(E, npts, drop, team, monkeypatch):
    """FreedomAnalysis of a system the fronts serve (freedom.hip: freedom_by_probes): no pivoted QR -- the projector onto null(J)
    applied to pseudo-random vectors by the frontal factorisation, the candidates refined by subspace iteration, participation from
    the null vectors found.  Fully constrained sketches and sketches that lost their last 1 ... 4 constraints, 80 ... 2000
    variables on 1 ... 14 workgroups, systems created for batches and for one solve: the underconstrained set equal to the oracle's
    dense pivoted QR of the same Jacobian (find_dof.rs:31-103), participation within 1e-9, and equal to the device's own QR.  (The
    1700-variable sketch has a singular value ~1e-6 of the largest entry: a Ritz value stays undecided at the first lambda and the
    second opinion at lambda / 1000 settles it.)"""
    recs, g = gen.connected_sketch(npts, 4242)
    if drop:
        recs = recs[:-drop]
    n = len(g)
    sysobj = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY if team == "latency" else team)
    assert sysobj.info()["front_workgroups"] >= 1
    X = g[None, :] + gen.keyed_uniform(61, 2, n, -0.01, 0.01)
    x, st, _ = sysobj.solve_batch(X, E.Config(max_iterations=60))
    monkeypatch.delenv("EZPZ_FREEDOM_PROBES", raising=False)
    mask, part = sysobj.freedom_batch(x)
    _, J, _ = sysobj.eval_batch(x)
    for b in range(len(x)):
        under, want = O.freedom_analysis_dense(J[b])
        assert np.nonzero(mask[b])[0].tolist() == under, (b, int(mask[b].sum()), len(under))
        assert np.allclose(part[b], want, atol=1e-9), (b, float(np.abs(part[b] - want).max()))
        assert bool(under) == bool(drop)
    monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")
    mask_qr, part_qr = sysobj.freedom_batch(x)
    assert np.array_equal(mask_qr, mask) and np.allclose(part_qr, part, atol=1e-8)


@pytest.mark.parametrize("drop", [0, 1, 2])
def test_freedom_probes_on_a_linear_only_sketch(E, drop, monkeypatch):
    """A chain of points tied by horizontal and vertical distances only (the frontal kernel's linear-only build, whose one Jacobian
    sweep rides in eval()): fully constrained, and with its last one or two constraints gone (the last point free along y, then
    along both)."""
    npts = 60
    cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
    for i in range(1, npts):
        cons += [O.horizontal_distance((2 * i, 2 * i + 1), (2 * i - 2, 2 * i - 1), 1.0 + 0.01 * i),
                 O.vertical_distance((2 * i, 2 * i + 1), (2 * i - 2, 2 * i - 1), 0.5)]
    recs = O.stack(cons)
    if drop:
        recs = recs[:-drop]
    n = 2 * npts
    sysobj = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
    assert sysobj.info()["team_mode"] == 5
    x, st, _ = sysobj.solve_batch(np.zeros((1, n)), E.Config(max_iterations=20))
    monkeypatch.delenv("EZPZ_FREEDOM_PROBES", raising=False)
    mask, part = sysobj.freedom_batch(x)
    _, J, _ = sysobj.eval_batch(x)
    under, want = O.freedom_analysis_dense(J[0])
    assert np.nonzero(mask[0])[0].tolist() == under and len(under) == drop
    assert np.allclose(part[0], want, atol=1e-9)


def test_freedom_probes_leave_many_degrees_of_freedom_to_the_qr(E, monkeypatch):
    """Eight probes are trusted with up to four candidate directions: a sketch that lost twelve constraints goes to the pivoted QR
    (the same answer as with the probes switched off, equal to the oracle's)."""
    recs, g = gen.connected_sketch(150, 4242)
    recs = recs[:-12]
    n = len(g)
    sysobj = E.System(recs, n)
    x, st, _ = sysobj.solve_batch(g[None, :], E.Config(max_iterations=60))
    monkeypatch.delenv("EZPZ_FREEDOM_PROBES", raising=False)
    mask, part = sysobj.freedom_batch(x)
    monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")
    mask_qr, part_qr = sysobj.freedom_batch(x)
    assert np.array_equal(mask, mask_qr) and np.array_equal(part, part_qr)
    _, J, _ = sysobj.eval_batch(x)
    under, want = O.freedom_analysis_dense(J[0])
    assert np.nonzero(mask[0])[0].tolist() == under and len(under) >= 5


def test_freedom_random_systems(E):
    """The fuzz generator's systems (all 25 kinds, repeated ids, isolated variables) through the analysis."""
    rng = np.random.default_rng(99)
    checked = unstable = 0
    for trial in range(120):
        n = int(rng.integers(4, 20))
        cons = [gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=n) for _ in range(int(rng.integers(1, 10)))]
        recs = O.stack(cons)
        X = rng.uniform(-5, 5, (4, n))
        sysobj = E.System(recs, n)
        mask, part = sysobj.freedom_batch(X)
        _, J, _ = sysobj.eval_batch(X)
        for b in range(4):
            if not np.all(np.isfinite(J[b])):
                continue
            sv = np.linalg.svd(J[b], compute_uv=False)
            big = sv.max() if sv.size else 0.0
            if big == 0.0 or np.any((sv > 1e-10 * big) & (sv < 1e-6 * big)):
                unstable += 1  # a singular value near the 1e-8 rank threshold: the rank itself is ill-defined
                continue
            under, want = O.freedom_analysis_dense(J[b])
            assert np.allclose(part[b], want, atol=1e-7), (trial, b, float(np.abs(part[b] - want).max()))
            near = np.abs(want - (1e-3 * want.max()) ** 2) < 1e-9
            got = np.nonzero(mask[b])[0].tolist()
            assert [v for v in got if not near[v]] == [v for v in under if not near[v]], (trial, b)
            checked += 1
    assert checked >= 300, (checked, unstable)


def test_solve_analysis_object_api_and_priority_tiers(E):
    """lib.rs:134-146 through the crate mirror; the analysis follows the tier that is returned."""
    ids = E.IdGenerator()
    p = E.DatumPoint.new(ids)
    reqs = [E.ConstraintRequest.highest_priority(E.Constraint.Fixed(p.x_id, 1.0)),
            E.ConstraintRequest.new(E.Constraint.Fixed(p.y_id, 2.0), 1),
            E.ConstraintRequest.new(E.Constraint.Fixed(p.y_id, 3.0), 1)]
    out = E.solve_analysis(reqs, [(p.x_id, 0.0), (p.y_id, 0.0)])
    assert out.outcome.priority_solved() == 0 and out.analysis.is_underconstrained()
    assert out.analysis.underconstrained() == [1]
    out = E.solve_analysis(reqs[:2], [(p.x_id, 0.0), (p.y_id, 0.0)])
    assert out.outcome.priority_solved() == 1 and not out.analysis.is_underconstrained()
    assert E.solve_analysis([], [(0, 0.5)]).analysis.underconstrained() == []


def test_solve_is_reentrant_across_threads_and_cache_evictions(E):
    """`solve` is callable concurrently (no global state in the reference).  40 distinct topologies from
    8 threads overflow the 16-entry topology cache, so entries are evicted while other threads still solve on them."""
    import threading

    topologies = []
    for k in range(40):
        npts = 3 + k % 7
        reqs = [O.fixed(0, 0.0), O.fixed(1, float(k))]
        guesses = [(0, 0.1), (1, k + 0.2)]
        for p in range(1, npts):
            reqs.append(O.distance((2 * p - 2, 2 * p - 1), (2 * p, 2 * p + 1), 1.0 + 0.1 * k))
            reqs.append(O.horizontal((2 * p - 2, 2 * p - 1), (2 * p, 2 * p + 1)))
            guesses += [(2 * p, 0.9 * p + 0.3), (2 * p + 1, k + 0.1 * p)]
        want = O.solve(reqs, guesses)
        topologies.append((O.stack(reqs), guesses, want))
    errors = []

    def worker(tid):
        try:
            for rep in range(3):
                for k in range(tid, 40, 4):
                    recs, guesses, want = topologies[k]
                    got = E.solve_records(recs, guesses)
                    assert (got.error, got.iterations, got.converged, got.unsatisfied) == (
                        0, want.iterations, want.converged, want.unsatisfied), (tid, k)
                    assert_x_close(got.final_values, want.final_values)
        except Exception as exc:  # noqa: BLE001 -- reported below, in the main thread
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
