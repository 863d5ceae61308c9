"""GPU parity tests of the FRONTAL launch shape (team_mode 5: csrc/fronts.cpp, csrc/front_kernel.hip.hpp) through the C ABI,
against the oracle (reference: Model::solve_levenberg_marquardt, ezpz/src/solver/newton.rs:29-145; faer's LLT, newton.rs:87-102).
The fronts factorise in another elimination order than the oracle's: integer and flag outputs are equal, coordinates are held to
1e-6 relative with the measured bar of tests/sensitivity.py where a system's own LM path amplifies rounding."""
import os

import numpy as np
import pytest

import gen
from oracle import oracle as O
from sensitivity import assert_batch_matches_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return ezpz_amd


class env:
    """Environment switches the symbolic phase reads when a system is created."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def front_system(E, recs, n, wgs=None):
    if wgs is None:
        s = E.System(recs, n, team_size=E.TEAM_FRONTS)
    else:
        with env(EZPZ_FRONT_WGS=wgs):
            s = E.System(recs, n, team_size=E.TEAM_FRONTS)
    info = s.info()
    assert info["team_mode"] == 5, info
    return s, info


@pytest.mark.parametrize("npts,wgs", [(4, 1), (8, 1), (25, 1), (75, 1), (75, 2), (150, 1), (150, 3), (400, None), (400, 4), (1000, None)])
def test_connected_sketch_on_fronts_equals_the_oracle(E, npts, wgs):
    """tests/gen.py:connected_sketch (fully determined, mixed kinds) from its own start and from jittered starts: iterations,
    flags, unsatisfied counts equal to the oracle's, coordinates at 1e-6; the batch is bitwise repeatable."""
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    n = len(g)
    s, info = front_system(E, recs, n, wgs)
    if wgs and wgs > 1:
        assert 1 < info["grid_workgroups"] <= wgs
    if wgs is None and npts >= 400:
        assert info["grid_workgroups"] > 1
    x0 = g[None, :] + gen.keyed_uniform(npts, 6, n, -0.02, 0.02)
    x0[0] = g
    cfg = dict(max_iterations=60)
    x, st, mask = s.solve_batch(x0, E.Config(**cfg), want_mask=True)
    x2, st2, _ = s.solve_batch(x0, E.Config(**cfg))
    assert np.array_equal(x, x2) and np.array_equal(st["iterations"], st2["iterations"])
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    assert_batch_matches_oracle(recs, x0, x, st["iterations"], st["converged"], O.Config(**cfg), oracle_result=(xo, it, conv),
                                what=("fronts", npts, wgs))
    assert np.array_equal(st["n_unsatisfied"], nun)
    assert np.array_equal(mask.sum(axis=1), nun)
    assert not np.any(st["iterations"] == 0xFFFFFFFF)


@pytest.mark.parametrize("family", ["tree", "band", "hub", "comb"])
@pytest.mark.parametrize("wgs", [1, 4])
def test_graph_families_on_fronts(E, family, wgs):
    rng = np.random.default_rng(21)
    recs, true = gen.graph_sketch(family, 50, rng)
    n = len(true)
    s, info = front_system(E, recs, n, wgs)
    x0 = true[None, :] + gen.keyed_uniform(77, 5, n, -0.03, 0.03)
    cfg = dict(max_iterations=80)
    x, st, _ = s.solve_batch(x0, E.Config(**cfg))
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    assert_batch_matches_oracle(recs, x0, x, st["iterations"], st["converged"], O.Config(**cfg), oracle_result=(xo, it, conv),
                                what=("fronts", family, wgs))
    assert np.array_equal(st["n_unsatisfied"], nun)


@pytest.mark.parametrize("wgs", [1, 3])
def test_all_kinds_weights_warnings_and_masks_on_fronts(E, wgs):
    """A random system of all 25 kinds (degenerate guards fire, rows repeat columns, weights differ, some variables are touched
    by nothing): Degenerate warnings in the reference's order, the unsatisfied mask and every status field equal to the oracle's."""
    rng = np.random.default_rng(5)
    n = 40
    cons = []
    for kind in range(25):
        c = gen.arb_constraint(rng, kind, hi=32)
        c["weight"] = float(rng.uniform(0.5, 2.0))
        cons.append(c)
    recs = O.stack(cons)
    x0 = rng.uniform(-5.0, 5.0, (6, n))
    x0[5, :8] = 0.0  # coincident points: guards
    s, info = front_system(E, recs, n, wgs)
    cfg = dict(max_iterations=25)
    x, st, logs = s.solve_batch_logged(x0, E.Config(**cfg), warn_cap=4096)
    _, st2, mask = s.solve_batch(x0, E.Config(**cfg), want_mask=True)
    assert np.array_equal(st, st2)
    needed = assert_batch_matches_oracle(recs, x0, x, st["iterations"], st["converged"], O.Config(**cfg), what=("fronts all kinds", wgs))
    for b in range(len(x0)):
        want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
        assert want.error == 0
        if int(st["iterations"][b]) != want.iterations:
            assert needed  # (a chaotic path, judged by the measured bar above)
            continue
        assert int(st["n_warnings"][b]) == len(want.warnings), b
        assert [p for _, p in logs[b]] == [w[0] for w in want.warnings][: len(logs[b])], b
        assert sorted(np.nonzero(mask[b])[0].tolist()) == sorted(want.unsatisfied), b
        assert int(st["n_unsatisfied"][b]) == len(want.unsatisfied)


@pytest.mark.parametrize("wgs", [1, 3])
def test_failed_pivots_limits_and_nan_starts_on_fronts(E, wgs):
    """The loop's rare branches: a negative lambda makes every factorisation fail (each burns an iteration with lambda x 10,
    newton.rs:93-99) until the pivots turn positive; max_iterations = 0 / 1; a NaN start stays NaN with the oracle's flags."""
    recs, g = gen.connected_sketch(40, 1040)
    n = len(g)
    s, info = front_system(E, recs, n, wgs)
    x0 = np.stack([g, g + 0.01, g])
    x0[2, 5] = np.nan
    for cfg in (dict(max_iterations=40, initial_lambda=-1e-3), dict(max_iterations=0), dict(max_iterations=1), dict(max_iterations=30)):
        x, st, _ = s.solve_batch(x0, E.Config(**cfg))
        for b in range(len(x0)):
            want = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
            assert (int(st["iterations"][b]), bool(st["converged"][b]), int(st["n_unsatisfied"][b])) == \
                   (want.iterations, want.converged, len(want.unsatisfied)), (cfg, b)
            assert np.array_equal(np.isnan(x[b]), np.isnan(want.final_values)), (cfg, b)
            err = np.abs(x[b] - want.final_values) / np.maximum(1.0, np.abs(want.final_values))
            assert np.nanmax(err, initial=0.0) <= 1e-6, (cfg, b, float(np.nanmax(err, initial=0.0)))


@pytest.mark.parametrize("npts", [2500, 10000])
def test_large_connected_sketches_on_many_workgroups(E, npts):
    """5000 and 20 000 variables: the planner's own choice of workgroups (22 and 64: the shares must fit one CU's LDS each), the
    top of the tree on workgroup 0, update matrices and steps between workgroups as chunks.  Two starts against the oracle."""
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    n = len(g)
    s, info = front_system(E, recs, n)
    assert info["grid_workgroups"] >= 16 and info["workspace_in_lds"] == 1, info
    x0 = np.stack([g, g + gen.keyed_uniform(npts, 1, n, -0.01, 0.01)[0]])
    cfg = dict(max_iterations=60)
    x, st, _ = s.solve_batch(x0, E.Config(**cfg))
    x2, st2, _ = s.solve_batch(x0, E.Config(**cfg))
    assert np.array_equal(x, x2) and np.array_equal(st["iterations"], st2["iterations"])
    rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    assert_batch_matches_oracle(recs, x0, x, st["iterations"], st["converged"], O.Config(**cfg), oracle_result=(xo, it, conv),
                                what=("fronts", npts, "auto"))
    assert np.array_equal(st["n_unsatisfied"], nun)


def test_one_solve_call_takes_the_fronts_and_equals_the_oracle(E):
    """ezpz_solve (the reference's protocol, one call per solve: ezpz-cli/src/main.rs:96-98) of a 300-variable connected sketch:
    the automatic latency shape is the frontal one from EzpzLaunchPolicy.front_min_vars_one_solve variables."""
    recs, g = gen.connected_sketch(150, 1150)
    n = len(g)
    assert E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY).info()["team_mode"] == 5
    want = O.solve(recs, g, O.Config(max_iterations=60), linsolve=O.LINSOLVE_SPARSE)
    for _ in range(3):  # cold, then warm calls on the request's plan
        got = E.solve_records(recs, g, E.Config(max_iterations=60))
        assert (got.iterations, got.converged, list(got.unsatisfied)) == (want.iterations, want.converged, list(want.unsatisfied))
        err = np.abs(got.final_values - want.final_values) / np.maximum(1.0, np.abs(want.final_values))
        assert float(err.max()) <= 1e-6


@pytest.mark.parametrize("npts", [150, 1000])
def test_small_calls_of_a_batch_system_take_the_fronts(E, npts):
    """A system created for batches (team_size 0) carries the frontal plan and takes it for calls too small to fill the device at one
    workgroup per system (EzpzSystemInfo.front_max_batch): those calls are bit for bit the frontal system's, one system more is bit
    for bit the record walk's (a system created with EZPZ_FRONTS=0), both equal to the oracle; its info describes the large calls."""
    import torch

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    n = len(g)
    auto = E.System(recs, n)
    info = auto.info()
    fronts, finfo = front_system(E, recs, n)
    with env(EZPZ_FRONTS=0):
        records = E.System(recs, n)
    G = finfo["grid_workgroups"]
    assert info["team_mode"] != 5 and info["front_workgroups"] == G, info
    assert info["front_max_batch"] == max(1, cus // G) * max(1, G // 4), (info, G, cus)
    assert finfo["front_max_batch"] == 0xFFFFFFFF and records.info()["front_workgroups"] == 0
    most = info["front_max_batch"]
    x0 = g[None, :] + gen.keyed_uniform(npts, most + 1, n, -0.01, 0.01)
    x0[0] = g
    cfg = E.Config(max_iterations=60)
    for B, same_as in ((1, fronts), (most, fronts), (most + 1, records)):
        x, st, _ = auto.solve_batch(x0[:B], cfg)
        xw, stw, _ = same_as.solve_batch(x0[:B], cfg)
        assert np.array_equal(x, xw) and np.array_equal(st, stw), B
        # ... and on the device-resident entry, whose call is one piece by construction
        xin = torch.from_numpy(x0[:B]).cuda()
        xd = torch.empty_like(xin)
        std = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
        auto.solve_batch_device(xin.data_ptr(), B, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream, cfg)
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), x), B
    x, st, _ = auto.solve_batch(x0[:4], cfg)
    assert_batch_matches_oracle(recs, x0[:4], x, st["iterations"], st["converged"], O.Config(max_iterations=60), what=("auto small call", npts))


@pytest.mark.parametrize("drop", [0, 2])
def test_solve_analysis_call_of_a_connected_sketch(E, drop):
    """ezpz_solve_analysis (lib.rs:134-146) of a 300-variable connected sketch through the one-call entry: the solve on the fronts,
    the analysis by null-space probes on the same factorisation -- and with a deferred program (the symbolic phase returned with the
    frontal plan alone) nothing else of the system is ever analysed.  Outcome and underconstrained ids equal to the oracle's."""
    import time

    recs, g = gen.connected_sketch(150, 4242)
    if drop:
        recs = recs[:-drop]
    guesses = list(enumerate(g.tolist()))
    want = O.solve(recs, guesses, O.Config(max_iterations=60), linsolve=O.LINSOLVE_SPARSE, analysis=True)
    for _ in range(3):  # cold, then warm calls on the request's plan
        got = E.solve_records(recs, guesses, E.Config(max_iterations=60), analysis=True)
        assert (got.error, got.iterations, got.converged) == (0, want.iterations, want.converged)
        assert sorted(got.underconstrained) == sorted(want.underconstrained) and bool(got.underconstrained) == bool(drop)
        err = np.abs(got.final_values - want.final_values) / np.maximum(1.0, np.abs(want.final_values))
        free = np.zeros(len(g), bool)
        free[list(want.underconstrained)] = True
        assert float(err[~free].max()) <= 1e-6
