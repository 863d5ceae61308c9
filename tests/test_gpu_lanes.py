"""GPU parity tests (-m gpu) of the lane-per-system specialised kernel (jit_kernel.hip.hpp: lane_kernel): batches of one
small system as run-time compiled straight-line code, every lane running the LM loop of its own system.  Through the C
ABI, against the CPU oracle, the committed golden vectors and the sub-wavefront list-walk kernels."""
import json
import os

import numpy as np
import pytest

import gen
from conftest import GOLDEN, read_case
from oracle import oracle as O
from oracle import textual as T
from sensitivity import assert_batch_matches_oracle

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
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert np.array_equal(np.isnan(got), np.isnan(want)) and (not np.any(~np.isnan(err)) or np.nanmax(err) <= rel), float(np.nanmax(err))


def lanes(E, recs, n):
    sysobj = E.System(recs, n)
    assert sysobj.info()["team_mode"] == 0
    assert sysobj.specialize(wait=True) == 2
    return sysobj


def test_golden_vectors_on_the_lane_kernel(E):
    """Every committed vector (28 fixtures x 4 guess variants) as one batch per fixture: iterations, flags, unsatisfied
    mask, warning count and log order, residual norm; determined coordinates at 1e-6, the variables the oracle's
    FreedomAnalysis flags as underconstrained at 20x the oracle's own one-ulp sensitivity (never beyond 1e-4)."""
    vectors = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))
    done = 0
    for case, recs in vectors.items():
        ref = T.load(open(os.path.join(GOLDEN, "test_cases", case)).read())
        by_sides = {}
        for rec in recs:  # side inference happens above the handle API: group the variants by the sides they infer
            resolved = O.stack([O.set_from_initial_values(c, np.asarray(rec["guesses"])) for c in ref.constraints])
            by_sides.setdefault(resolved.tobytes(), (resolved, []))[1].append(rec)
        for resolved, group in by_sides.values():
            if E.specialized_source(resolved, ref.num_vars) == "":
                continue
            sysobj = lanes(E, resolved, ref.num_vars)
            x0 = np.asarray([rec["guesses"] for rec in group])
            x, st, mask = sysobj.solve_batch(x0, want_mask=True)
            _, _, logs = sysobj.solve_batch_logged(x0, warn_cap=4096)
            for b, rec in enumerate(group):
                assert int(st["iterations"][b]) == rec["iterations"], (case, b)
                assert bool(st["converged"][b]) == rec["converged"], case
                assert np.nonzero(mask[b])[0].tolist() == rec["unsatisfied"], case
                assert int(st["n_warnings"][b]) == rec["n_warnings"], case
                assert logs[b] == sorted(logs[b]), case  # written in the reference's order already
                want = np.asarray(rec["final_values"])
                free = np.zeros(len(want), dtype=bool)
                free[rec["underconstrained"]] = True
                if np.any(~free):
                    assert_x_close(x[b][~free], want[~free], REL)
                if np.any(free):
                    assert_x_close(x[b][free], want[free], min(1e-4, max(REL, 20.0 * rec["ulp_sensitivity"])))
                assert abs(float(st["final_residual_inf"][b]) - rec["final_residual_inf"]) <= 1e-9, case
                done += 1
    assert done >= 100


@pytest.mark.parametrize("case", ["square", "two_rectangles", "circle_tangent", "arc_radius", "chamfer_square", "perpendicular",
                                  "symmetric", "tiny", "nonsquare", "arc_length"])
def test_jittered_batches_match_the_oracle_and_the_list_walk_kernel(E, case):
    ref = T.load(read_case(case))
    recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
    B = 5000
    x0 = ref.guesses[None, :] + gen.keyed_uniform(99, B, ref.num_vars, -0.1, 0.1)
    x0[0] = ref.guesses
    cfg = dict(max_iterations=60)
    walk = E.System(recs, ref.num_vars, team_size=E.TEAM_AUTO_LISTS)
    xw, stw, maskw = walk.solve_batch(x0, E.Config(**cfg), want_mask=True)
    sysobj = lanes(E, recs, ref.num_vars)
    x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg))
    assert rc == 0
    assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv) and np.array_equal(st["n_unsatisfied"], nun)
    assert np.array_equal(mask, maskw) and np.array_equal(st["n_warnings"], stw["n_warnings"])
    free = sysobj.freedom_batch(x[:64])[0].astype(bool).any(axis=0)
    assert_x_close(x[:, ~free], xo[:, ~free])
    assert_x_close(x[:, free], xo[:, free], 1e-4)  # (the reference's own tolerance, lib.rs:43; measured worst 3e-5)
    assert np.max(np.abs(st["final_residual_inf"] - stw["final_residual_inf"])) <= 1e-9
    # ragged batch sizes: fewer systems than lanes, not a multiple of the wavefront
    for b in (1, 3, 65, 257):
        xs, sts, _ = sysobj.solve_batch(x0[:b], E.Config(**cfg))
        assert np.array_equal(xs, x[:b]) and np.array_equal(sts["iterations"], st["iterations"][:b])


def test_square_random_integer_guesses_full_size(E):
    """BASELINE configs[2]: 65 536 x square with the reference's own proptest distribution (proptests.rs:294-329): every
    system against the oracle (3 ... 21 iterations; lanes of a wavefront finish at different times and pick up new
    systems), geometry a 4 x 4 square."""
    ref = T.load(read_case("square"))
    B = 65536
    x0 = gen.keyed_uniform(0x657A707A, B, 8, -10000, 10000, integer=True)
    sysobj = lanes(E, ref.constraints, ref.num_vars)
    x, st, _ = sysobj.solve_batch(x0)
    assert np.all(st["n_unsatisfied"] == 0) and np.all(st["converged"] == 1)
    a, b, c, d = x[:, 0:2], x[:, 2:4], x[:, 4:6], x[:, 6:8]
    assert np.all(np.abs(a) < 1e-4) and np.all(np.abs(c - 4.0) < 1e-4)
    assert np.all(np.abs(b - np.array([4.0, 0.0])) < 1e-4) and np.all(np.abs(d - np.array([0.0, 4.0])) < 1e-4)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0)
    assert np.array_equal(st["iterations"], it) and len(np.unique(it)) > 10
    assert_x_close(x, xo)


def test_weights_warnings_and_failed_pivots_per_lane(E):
    """Per-lane LM control: non-unit weights, degenerate evaluations (warnings in the reference's order, none for the
    sweep of a failed factorisation), NaN guesses in some systems of the batch only.  The system is consistent: a solve
    that stalls at a least-squares minimum takes or rejects its last steps on the last bit of the residual sum, where
    device and host libm differ, so exact warning counts are only defined for solves that converge."""
    recs = O.stack([O.distance((0, 1), (2, 3), 3.0), O.fixed(0, 0.0, weight=2.5), O.fixed(1, 0.0), O.horizontal((0, 1), (2, 3)),
                    O.fixed(2, 3.0, weight=0.5), O.points_at_angle((0, 1), (2, 3), (4, 5), ("deg", 90.0)), O.fixed(5, 2.0)])
    n = 6
    x0 = gen.keyed_uniform(5, 3000, n, -2.0, 2.0)
    x0[:, 2] += 3.0
    x0[::7, 2:4] = x0[::7, 0:2]  # coincident points: Degenerate warnings
    x0[5::11, 4] = np.nan
    checked_warnings = 0
    for cfg in (dict(), dict(initial_lambda=1e-30, max_iterations=12)):
        sysobj = lanes(E, recs, n)
        x, st, logs = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=512)
        _, _, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
        for b in list(range(0, 3000, 97)) + list(range(0, 60)):
            want = O.solve(recs, x0[b], O.Config(**cfg), warn_cap=4096)
            assert np.array_equal(np.isnan(x[b]), np.isnan(want.final_values)), b
            assert bool(st["converged"][b]) == want.converged, b
            if not want.converged and not np.any(np.isnan(x0[b])):
                continue
            assert int(st["iterations"][b]) == want.iterations, b
            assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied, b
            assert int(st["n_warnings"][b]) == len(want.warnings), b
            assert float(st["final_lambda"][b]) == want.final_lambda, b
            assert [p for _, p in logs[b]] == [w[0] for w in want.warnings][: len(logs[b])], b
            checked_warnings += len(want.warnings)
            assert_x_close(x[b], want.final_values, 1e-6)
    assert checked_warnings > 50


def test_random_systems_of_all_kinds_on_the_lane_kernel(E):
    """The reference's fuzz target as a comparison, on run-time compiled lane kernels: 48 random systems of all 25 kinds
    (repeated ids, random weights), 24 random guess vectors each, against the oracle.  Always: error-free, NaN patterns,
    `converged`, the unsatisfied mask; where the oracle's own answer is stable under a one-ulp perturbation of the
    guesses: iteration counts of solves that end on the residual test, warnings, coordinates where rank J = n."""
    rng = np.random.default_rng(777)
    kinds_seen, compared, exact = set(), 0, 0
    for trial in range(48):
        nvars = int(rng.integers(4, 13))
        cons = []
        for _ in range(int(rng.integers(1, 9))):
            c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nvars)
            c["weight"] = float(rng.choice([1.0, 1.0, 1.0, 0.25, 3.0]))
            cons.append(c)
        recs = O.stack(cons)
        kinds_seen.update(int(k) for k in recs["kind"])
        cfg = dict(max_iterations=int(rng.choice([35, 35, 10, 60])))
        x0 = rng.uniform(-8.0, 8.0, (24, nvars))
        sysobj = lanes(E, recs, nvars)
        x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
        for b in range(len(x0)):
            want = O.solve(recs, x0[b], O.Config(**cfg), warn_cap=1 << 14)
            assert want.error == 0
            assert np.array_equal(np.isnan(x[b]), np.isnan(want.final_values)), (trial, b)
            again = O.solve(recs, x0[b] * (1.0 + rng.uniform(-1.0, 1.0, nvars) * 2.0 ** -52), O.Config(**cfg))
            scale = np.maximum(1.0, np.abs(want.final_values))
            stable = (np.all(np.isfinite(want.final_values)) and again.iterations == want.iterations and again.converged == want.converged
                      and again.unsatisfied == want.unsatisfied and np.max(np.abs(again.final_values - want.final_values) / scale) < 1e-9)
            if not stable:
                continue
            compared += 1
            assert bool(st["converged"][b]) == want.converged, (trial, b)
            assert np.nonzero(mask[b])[0].tolist() == want.unsatisfied, (trial, b)
            if want.final_residual_inf <= 1e-8:
                assert int(st["iterations"][b]) == want.iterations, (trial, b)
                assert int(st["n_warnings"][b]) == len(want.warnings), (trial, b)
                assert abs(float(st["final_residual_inf"][b]) - want.final_residual_inf) <= 1e-9
                exact += 1
            else:
                assert abs(int(st["iterations"][b]) - want.iterations) <= 2, (trial, b)
            J = np.zeros((sum(O.residual_dim(c) for c in cons), nvars))
            row = 0
            for c in cons:
                rows, _ = O.jacobian_rows(c, want.final_values)
                for r in rows:
                    for i, pd in r:
                        J[row, i] += c["weight"] * pd
                    row += 1
            if J.size and np.all(np.isfinite(J)):
                sv = np.linalg.svd(J, compute_uv=False)
                if len(sv) >= nvars and sv[nvars - 1] > 1e-7 * max(sv[0], 1e-300):
                    assert_x_close(x[b], want.final_values)
    assert compared >= 400 and exact >= 40 and len(kinds_seen) == O.NUM_KINDS, (compared, exact, sorted(kinds_seen))


def test_one_wavefront_per_system_equals_the_lane_kernel_bitwise(E):
    """The latency shape of a small system (jit_kernel.hip.hpp: wave_kernel -- what one ezpz_solve call of a sketch fixture
    runs on): constraint sweeps and the assembly of the normal equations across the 64 lanes of a wavefront, the lane
    kernel's arithmetic operation for operation.  Every fixture, jittered and wild starts, weights, degenerate
    evaluations, NaN guesses, two configurations; random systems of all 25 kinds: values, statuses, unsatisfied masks
    and warning logs bit for bit the lane kernel's (which the tests above hold against the oracle) -- and ezpz_solve
    itself switches to it after 256 calls without changing an answer."""
    import ctypes as C

    def both(recs, n, x0, cfg):
        lane = E.System(recs, n)
        wave = E.System(recs, n, team_size=E.TEAM_LATENCY_WAVE)
        assert lane.specialize(wait=True) == 2 and wave.specialize(wait=True) == 2
        assert "ezpz_jit_wave" in E.specialized_source(recs, n, wave=True)
        out = []
        for sysobj in (lane, wave):
            x, st, logs = sysobj.solve_batch_logged(x0, E.Config(**cfg), warn_cap=256)
            _, _, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)
            out.append((x, st, logs, mask))
        (xl, stl, logl, ml), (xw, stw, logw, mw) = out
        assert np.array_equal(xl, xw, equal_nan=True)
        for f in stl.dtype.names:
            assert np.array_equal(stl[f], stw[f], equal_nan=True), f
        assert logl == logw and np.array_equal(ml, mw)
        return stw

    checked = 0
    for name in sorted(os.listdir(os.path.join(GOLDEN, "test_cases"))):
        ref = T.load(read_case(name))
        if ref.num_vars > 20 or len(ref.constraints) > 40 or not len(ref.constraints):
            continue
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        if not E.specialized_source(recs, ref.num_vars, wave=True):
            continue
        n = ref.num_vars
        x0 = np.concatenate([ref.guesses[None, :], ref.guesses[None, :] + gen.keyed_uniform(3, 40, n, -0.3, 0.3),
                             gen.keyed_uniform(4, 20, n, -8.0, 8.0)])
        x0[7, 0] = np.nan
        for cfg in (dict(), dict(initial_lambda=1e-30, max_iterations=9)):
            both(recs, n, x0, cfg)  # (a device holds > 61 wavefronts: the batch runs one system per workgroup)
            checked += 1
    assert checked >= 40
    # weights, degenerate starts, failed pivots
    recs = O.stack([O.distance((0, 1), (2, 3), 3.0), O.fixed(0, 0.0, weight=2.5), O.fixed(1, 0.0), O.horizontal((0, 1), (2, 3)),
                    O.fixed(2, 3.0, weight=0.5), O.points_at_angle((0, 1), (2, 3), (4, 5), ("deg", 90.0)), O.fixed(5, 2.0)])
    x0 = gen.keyed_uniform(5, 200, 6, -2.0, 2.0)
    x0[::7, 2:4] = x0[::7, 0:2]
    st = both(recs, 6, x0, dict())
    assert int(st["n_warnings"].sum()) > 20
    # random systems of all kinds
    rng = np.random.default_rng(99)
    kinds = set()
    for _ in range(40):
        nvars = int(rng.integers(4, 13))
        cons = []
        for _ in range(int(rng.integers(1, 9))):
            c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nvars)
            c["weight"] = float(rng.choice([1.0, 1.0, 0.25, 3.0]))
            cons.append(c)
        recs = O.stack(cons)
        if not E.specialized_source(recs, nvars, wave=True):
            continue
        kinds.update(int(k) for k in recs["kind"])
        both(recs, nvars, rng.uniform(-8.0, 8.0, (24, nvars)), dict(max_iterations=int(rng.choice([35, 10, 60]))))
    assert len(kinds) >= 20
    # ezpz_solve: the same answer before and after its topology's kernels take over (and from the resident kernel)
    ref = T.load(read_case("square"))
    first = E.solve_records(ref.constraints, ref.variables())
    for _ in range(400):
        got = E.solve_records(ref.constraints, ref.variables())
    import time
    time.sleep(1.5)
    for _ in range(20):
        got = E.solve_records(ref.constraints, ref.variables())
        assert got.iterations == first.iterations and np.allclose(got.final_values, first.final_values, rtol=0, atol=1e-9)


@pytest.mark.parametrize("npts,seed", [(12, 5), (30, 1), (75, 2), (150, 3)])
def test_connected_sketches_lanes_across_the_batch(E, npts, seed):
    """Device-filling batches (>= 64 x 2 x CUs systems; here forced with TEAM_BATCH_LANES) of one connected sketch of
    mixed kinds (tests/gen.py:connected_sketch: too large for a lane's registers) run one lane per system on the
    uniform-program kernel with its state in global memory (batch_kernel.hip.hpp): every system against the oracle
    (iteration counts, flags, unsatisfied masks, warnings, coordinates at 1e-6) and against the per-system list-walk
    teams that serve smaller batches."""
    recs, g = gen.connected_sketch(npts, 1000 + seed)
    n = len(g)
    B = 2500
    x0 = g[None, :] + gen.keyed_uniform(17 + seed, B, n, -0.03, 0.03)
    x0[0] = g
    x0[7, 3] = x0[2000, n - 1] = np.nan  # (every pivot of these two fails: their lanes burn the iterations, the others do not care)
    cfg = dict(max_iterations=40)
    sysobj = E.System(recs, n, team_size=E.TEAM_BATCH_LANES)
    xs, sts, masks = E.System(recs, n).solve_batch(x0[:300], E.Config(**cfg), want_mask=True)  # the list-walk teams
    x, st, mask = sysobj.solve_batch(x0, E.Config(**cfg), want_mask=True)                      # lanes across the batch
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    assert np.array_equal(st["converged"], conv) and np.array_equal(st["n_unsatisfied"], nun) and np.array_equal(mask.sum(axis=1), nun)
    # every system, none excluded: equal iteration counts and 1e-6, or -- where a stop test is decided in the last bits (the
    # elimination orders differ) -- a count and coordinates inside what the oracle itself does from one-ulp perturbations
    # of the start (tests/sensitivity.py)
    ok = ~np.isnan(x0).any(axis=1)
    needed = assert_batch_matches_oracle(recs, x0[ok], x[ok], st["iterations"][ok], st["converged"][ok], O.Config(**cfg),
                                         oracle_result=(xo[ok], it[ok], conv[ok]), what=("lanes", npts, seed))
    assert needed <= 6  # (measured in round 4: 2 of 2498; profiles/r04_parity_bar.txt)
    assert np.array_equal(st["iterations"][~ok], it[~ok]) and np.array_equal(np.isnan(x[~ok]), np.isnan(xo[~ok]))
    # (the lanes eliminate in the order with the least fill, the teams in the one with few levels)
    same = st["iterations"][:300] == sts["iterations"]
    assert same.mean() >= 0.99 and np.array_equal(mask[:300], masks)
    assert_x_close(x[:300][same], xs[same])
    # warnings and weights travel per lane: a degenerate start (two coincident points under `distance`) for some systems
    xd = x0[:2100].copy()
    xd[::5, 2:4] = xd[::5, 0:2]
    x2, st2, logs = sysobj.solve_batch_logged(xd, E.Config(**cfg), warn_cap=256)
    for b in (0, 5, 10, 1001, 2095):
        want = O.solve(recs, xd[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=4096)
        assert (int(st2["iterations"][b]), bool(st2["converged"][b]), int(st2["n_warnings"][b])) == (want.iterations, want.converged, len(want.warnings)), b
        assert [p for _, p in logs[b]] == [w[0] for w in want.warnings][: len(logs[b])], b
        assert_x_close(x2[b], want.final_values)


@pytest.mark.parametrize("npts", [40, 160])
def test_weighted_inconsistent_sketch_on_the_lanes_and_on_dense_phases(E, npts):
    """Non-unit weights and rows that cannot be satisfied on a connected sketch (every fifth constraint weighted 0.25 or
    3, a few points pinned twice at different places): the weighted residual drives the LM loop, the unsatisfied check
    re-evaluates the rows unweighted (lib.rs:305-327).  Lanes across the batch and the one-solve shape with its dense
    phases against the oracle: flags, unsatisfied rows, residual and coordinates."""
    recs, g = gen.connected_sketch(npts, 900 + npts)
    recs = recs.copy()
    for i in range(0, len(recs), 5):
        recs[i]["weight"] = 0.25 if (i // 5) % 2 else 3.0
    extra = [O.fixed(2 * k, float(g[2 * k]) + 0.3, weight=0.5) for k in (npts // 3, npts // 2)]
    recs = O.stack(list(recs) + extra)
    n = len(g)
    cfg = dict(max_iterations=60)
    x0 = g[None, :] + gen.keyed_uniform(npts, 96, n, -0.02, 0.02)
    lanes = E.System(recs, n, team_size=E.TEAM_BATCH_LANES)
    lat = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
    xl, stl, ml = lanes.solve_batch(x0, E.Config(**cfg), want_mask=True)
    xt, stt, mt = lat.solve_batch(x0[:8], E.Config(**cfg), want_mask=True)
    for b in range(8):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0 and want.unsatisfied
        for st, mask, x in ((stl, ml, xl), (stt, mt, xt)):
            assert bool(st["converged"][b]) == want.converged and np.nonzero(mask[b])[0].tolist() == want.unsatisfied, b
            assert abs(float(st["final_residual_inf"][b]) - want.final_residual_inf) <= 1e-6 * max(1.0, want.final_residual_inf)
            assert_batch_matches_oracle(recs, x0[b:b + 1], x[b:b + 1], [st["iterations"][b]], [st["converged"][b]], O.Config(**cfg),
                                        oracle_result=(np.asarray(want.final_values)[None, :], [want.iterations], [want.converged]),
                                        what=("weighted", npts, b), check_iterations=False)
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.array_equal(stl["converged"], conv) and np.array_equal(stl["n_unsatisfied"], nun)


def test_specialisation_survives_thousands_of_topologies(E):
    """Specialised kernels are never unloaded, so a process has a budget of them (512) -- but a topology that was only
    ever interpreted holds nothing but its source text and must not count: after 2300 distinct little systems have come
    and gone (more than the registry keeps entries for), a new topology still gets its kernel."""
    for k in range(2300):
        s = E.System(O.stack([O.fixed(0, float(k)), O.fixed(1, 1.0)]), 2)
        if k % 500 == 0:
            x, st, _ = s.solve_batch(np.zeros((4, 2)))
            assert np.allclose(x, [float(k), 1.0], rtol=0, atol=1e-6)
        del s
    recs = O.stack([O.distance((0, 1), (2, 3), 2.0), O.fixed(0, 0.0), O.fixed(1, 0.0), O.horizontal((0, 1), (2, 3))])
    sysobj = lanes(E, recs, 4)
    x, st, _ = sysobj.solve_batch(np.tile([0.1, -0.1, 1.7, 0.2], (64, 1)))
    assert st["converged"].all() and np.allclose(x, [0.0, 0.0, 2.0, 0.0], atol=1e-9)


def test_one_system_object_on_two_streams(E):
    """The lanes kernel (and the list walk of a sketch too large for the LDS) works in one global-memory workspace per
    system object: calls on different streams are chained on an event, so interleaved launches on two streams give what
    the same calls give one after the other.  (In a child process: the streams and device buffers are torch's, which must
    initialise the device before the library does.)"""
    import subprocess, sys
    from conftest import ROOT
    code = '''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
import numpy as np, ezpz_amd as E, gen
for npts, team in ((40, E.TEAM_BATCH_LANES), (1100, 0)):
    recs, g = gen.connected_sketch(npts, 4000 + npts)
    n = len(g)
    s = E.System(recs, n, team_size=team)
    B = 2048 if team else 24
    cfg = E.Config(max_iterations=30)
    xs = [torch.from_numpy(g[None, :] + gen.keyed_uniform(k, B, n, -0.02, 0.02)).to(dev) for k in range(4)]
    outs = [torch.empty_like(x) for x in xs]
    sts = [torch.zeros((B, 32), dtype=torch.uint8, device=dev) for _ in xs]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize(dev)
    for rep in range(3):
        for k, x in enumerate(xs):
            s.solve_batch_device(x.data_ptr(), B, outs[k].data_ptr(), sts[k].data_ptr(), 0, streams[k %% 2].cuda_stream, cfg)
    torch.cuda.synchronize(dev)
    for k, x in enumerate(xs):
        ref, st = torch.empty_like(x), torch.zeros((B, 32), dtype=torch.uint8, device=dev)
        s.solve_batch_device(x.data_ptr(), B, ref.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream, cfg)
        torch.cuda.synchronize(dev)
        same = torch.equal(ref, outs[k]) and torch.equal(st, sts[k])
        if not same:
            print("MISMATCH", npts, k, int((ref != outs[k]).sum()), int((st != sts[k]).sum()))
            sys.exit(3)
print("two streams ok")
''' % (ROOT, ROOT + "/tests")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "two streams ok" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]
