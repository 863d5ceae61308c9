"""GPU tests (-m gpu) at BASELINE.json's full sizes for configs[3] and configs[4]: size-independent properties over
the whole batch plus a sample of systems against the CPU oracle (the full batches would take the oracle minutes)."""
import numpy as np
import pytest

import gen
from conftest import read_case
from oracle import oracle as O
from oracle import textual as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return ezpz_amd


def test_200k_variable_ladder_full_size(E):
    """BASELINE configs[3]: gen_big_problem.py 50000 -- one sparse system of 200 000 variables / rows (150 000
    components), on a grid team.  Every solve takes the reference's 2 iterations (README.md:36-38), ends with
    max |r| <= 1e-9 and the exact geometry; 16 jittered systems bitwise against the oracle."""
    cs = E.textual.Problem.from_str(E.textual.gen_big_problem(50000)).to_constraint_system()
    n = cs.num_vars
    assert n == 200000 and len(cs.records) == 200000
    sysobj = E.System(cs.records, n)
    info = sysobj.info()
    assert info["grid_workgroups"] > 1 and (info["n_rows"], info["nnz_j"], info["nnz_l"]) == (200000, 250000, 250000)
    B = 16
    x0 = cs.guesses[None, :] + gen.keyed_uniform(0x657A707A, B, n, -0.25, 0.25)
    x0[0] = cs.guesses
    x, st, _ = sysobj.solve_batch(x0)
    assert np.all(st["iterations"] == 2) and np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0)
    assert np.all(st["final_residual_inf"] <= 1e-9)
    lines = np.arange(50000, dtype=np.float64)
    # line l: p_2l = (l, 0), p_2l+1 = (l, 4)
    assert np.max(np.abs(x[:, 0::4] - lines)) <= 1e-9 and np.max(np.abs(x[:, 2::4] - lines)) <= 1e-9
    assert np.max(np.abs(x[:, 1::4])) <= 1e-9 and np.max(np.abs(x[:, 3::4] - 4.0)) <= 1e-9
    rc, xo, it, conv, nun = O.solve_batch(cs.records, x0, linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.array_equal(it, st["iterations"]) and np.array_equal(x, xo)
    # the class-specialised kernel spreads the system over ~100 workgroups of 4 wavefronts (grid reductions of the LM
    # control across them): the same bits, statuses included
    assert sysobj.specialize(wait=True) == 2
    for _ in range(2):
        x2, st2, mask2 = sysobj.solve_batch(x0, want_mask=True)
        assert np.array_equal(x2, xo) and not mask2.any()
        for f in st.dtype.names:
            assert np.array_equal(st2[f], st[f]), f
    x1, st1, _ = sysobj.solve_batch(x0[3:4])  # one system alone
    assert np.array_equal(x1[0], xo[3])


def test_one_million_mixed_systems_full_size(E):
    """BASELINE configs[4] on one GPU: system i uses [circle_tangent, parallelogram, arc_radius][i mod 3], guesses =
    file guesses + U(-0.1, 0.1) from the keyed PRNG.  Every system converges with every constraint satisfied; 16 systems
    per topology (48 in all) against the oracle: iteration counts equal, determined coordinates at 1e-6."""
    total = 1_000_000
    for k, name in enumerate(["circle_tangent", "parallelogram", "arc_radius"]):
        ref = T.load(read_case(name))
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        B = total // 3 + (1 if k < total % 3 else 0)
        x0 = ref.guesses[None, :] + gen.keyed_uniform(0x657A707A + 101 * k, B, ref.num_vars, -0.1, 0.1)
        sysobj = E.System(recs, ref.num_vars)
        x, st, _ = sysobj.solve_batch(x0)
        assert np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0), name
        assert np.all(st["final_residual_inf"] <= 1e-8)
        sample = np.arange(0, B, B // 16)[:16]
        rc, xo, it, conv, nun = O.solve_batch(recs, x0[sample])
        assert rc == 0 and np.array_equal(st["iterations"][sample], it), name
        free = sysobj.freedom_batch(x[sample])[0].astype(bool)
        rel = np.abs(x[sample] - xo) / np.maximum(1.0, np.abs(xo))
        assert np.max(np.where(free, 0.0, rel)) <= 1e-6, name
        assert np.max(np.where(free, rel, 0.0)) <= 1e-4, name  # held by lambda only (tests.rs:630-637): the reference's EPSILON (lib.rs:43), measured worst 3e-5
        # the specialised lane kernel gives the same answers
        if sysobj.specialize(wait=True) == 2:
            x2, st2, _ = sysobj.solve_batch(x0)
            assert np.array_equal(st2["iterations"], st["iterations"]) and np.all(st2["n_unsatisfied"] == 0), name
            rel2 = np.abs(x2[sample] - xo) / np.maximum(1.0, np.abs(xo))
            assert np.max(np.where(free, 0.0, rel2)) <= 1e-6, name


def test_one_million_mixed_systems_through_the_heterogeneous_entry(E):
    """BASELINE configs[4] as ONE call (SURVEY.md 8b last row: a batch of different topologies): 1 M systems interleaved
    i mod 3 over circle_tangent / parallelogram / arc_radius in a ragged batch through ezpz_mixed_* -- regrouped by
    topology inside, one launch per topology on parallel streams, results back in caller order -- bitwise what the three
    per-topology calls give, from host buffers and from device buffers; and a topology that is one contiguous run of
    the batch (solved in place)."""
    import torch
    total = 1_000_000
    systems, rows, refs = [], [], []
    for k, name in enumerate(["circle_tangent", "parallelogram", "arc_radius"]):
        ref = T.load(read_case(name))
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        B = total // 3 + (1 if k < total % 3 else 0)
        rows.append(ref.guesses[None, :] + gen.keyed_uniform(0x657A707A + 101 * k, B, ref.num_vars, -0.1, 0.1))
        systems.append(E.System(recs, ref.num_vars))
        refs.append(ref)
    want = [s.solve_batch(r) for s, r in zip(systems, rows)]  # the per-topology calls: (x, status, _)
    topo = (np.arange(total) % 3).astype(np.uint32)
    mixed = E.MixedBatch(systems, topo)
    nv = np.array([r.num_vars for r in refs], dtype=np.uint64)
    assert mixed.total == int(nv[topo].sum()) and np.array_equal(mixed.offsets[:-1], np.concatenate([[0], np.cumsum(nv[topo])[:-1]]))
    x0 = np.empty(mixed.total)
    for k in range(3):
        idx = np.nonzero(topo == k)[0]
        pos = mixed.offsets[idx][:, None].astype(np.int64) + np.arange(int(nv[k]))[None, :]
        x0[pos] = rows[k]
    x, st = mixed.solve(x0)
    for k in range(3):
        idx = np.nonzero(topo == k)[0]
        pos = mixed.offsets[idx][:, None].astype(np.int64) + np.arange(int(nv[k]))[None, :]
        assert np.array_equal(x[pos], want[k][0]), k
        for f in st.dtype.names:
            assert np.array_equal(st[f][idx], want[k][1][f]), (k, f)
    assert np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0)
    # device buffers, on torch's stream
    dev = torch.device("cuda", 0)
    dx0 = torch.from_numpy(x0).to(dev)
    dx = torch.empty_like(dx0)
    dst = torch.zeros((total, 32), dtype=torch.uint8, device=dev)
    mixed.solve_device(dx0.data_ptr(), dx.data_ptr(), dst.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    assert np.array_equal(dx.cpu().numpy(), x) and np.array_equal(dst.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1), st)
    # the one-call form on a batch sorted by topology (every topology one contiguous run: solved in place), ragged tail
    sizes = [1000, 7, 333]
    topo2 = np.concatenate([np.full(c, k, dtype=np.uint32) for k, c in enumerate(sizes)])
    x02 = np.concatenate([rows[k][:c].reshape(-1) for k, c in enumerate(sizes)])
    x2, st2 = E.solve_batch_mixed(systems, topo2, x02)
    assert np.array_equal(x2, np.concatenate([want[k][0][:c].reshape(-1) for k, c in enumerate(sizes)]))
    assert np.array_equal(st2["iterations"], np.concatenate([want[k][1]["iterations"][:c] for k, c in enumerate(sizes)]))
    # a topology the batch does not use, and an empty batch
    x3, st3 = E.solve_batch_mixed(systems, np.full(5, 2, dtype=np.uint32), rows[2][:5].reshape(-1))
    assert np.array_equal(x3, want[2][0][:5].reshape(-1))
    x4, st4 = E.solve_batch_mixed(systems, np.zeros(0, dtype=np.uint32), np.zeros(0))
    assert x4.size == 0 and st4.size == 0
    with pytest.raises(E.NonLinearSystemError):
        E.solve_batch_mixed(systems, np.full(2, 3, dtype=np.uint32), np.zeros(16))  # topology index out of range


def test_quarter_million_jittered_sketches_on_the_lanes_full_size(E):
    """The connected-sketch leg of bench.py at its full size: 262 144 jittered starts of one 300-variable sketch -- one
    system per lane of the 4096 wavefronts the device holds, 4 to 19 LM iterations each.  Every system converges with every
    constraint satisfied, and a sample against the oracle with the measured bar of tests/sensitivity.py: spread over the
    batch, plus the systems with the MOST iterations -- the stragglers whose wavefronts handed them to the per-system
    teams (batch_kernel.hip.hpp), i.e. results that came out of the indirect team launch."""
    from ezpz_amd.synthetic import keyed_uniform, make_workload
    from sensitivity import assert_batch_matches_oracle

    desc, recs, g, jitter, _ = make_workload("sketch150")
    n, B = len(g), 262144
    x0 = g[None, :] + keyed_uniform(0x657A707A, B, n, -jitter, jitter)
    sysobj = E.System(recs, n)
    x, st, _ = sysobj.solve_batch(x0)
    assert np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0)
    assert np.all(st["final_residual_inf"] <= 1e-8)
    assert st["iterations"].min() >= 3 and st["iterations"].max() <= 35
    late = np.argsort(st["iterations"])[-48:]  # 9 iterations and more: past the round in which the wavefronts give up their last lanes
    assert st["iterations"][late].min() >= 9
    sample = np.unique(np.concatenate([np.arange(0, B, B // 48)[:48], late]))
    needed = assert_batch_matches_oracle(recs, x0[sample], x[sample], st["iterations"][sample], st["converged"][sample],
                                         what="sketch150 x 262144")
    assert needed <= 2  # (measured in round 4: 0; profiles/r04_parity_bar.txt)
    # the same batch again gives the same bits (the hand-over does not depend on timing: a lane gives up at a fixed point of
    # its own wavefront's progress)
    x2, st2, _ = sysobj.solve_batch(x0)
    assert np.array_equal(st2["iterations"], st["iterations"]) and np.array_equal(x2, x)


def test_thirty_two_thousand_jittered_sketches_on_the_record_walk_full_size(E):
    """The teams' leg of bench.py at its full size: 32 768 jittered starts of the 300-variable sketch -- below the batch from
    which the lanes serve a call, so every system is solved by a 128-lane workgroup walking records (team_mode 4: shape.cpp,
    build_records).  Every system converges with every constraint satisfied, a sample spread over the batch plus the systems
    with the most iterations against the oracle with the measured bar of tests/sensitivity.py, the same bits from run to run,
    and the same iteration counts and flags as the 262 144-system batch gives these starts on the lanes."""
    from ezpz_amd.synthetic import keyed_uniform, make_workload
    from sensitivity import assert_batch_matches_oracle

    desc, recs, g, jitter, _ = make_workload("sketch150")
    n, B = len(g), 32768
    x0 = g[None, :] + keyed_uniform(0x657A707A, B, n, -jitter, jitter)
    sysobj = E.System(recs, n)
    assert sysobj.info()["team_mode"] == 4 and sysobj.info()["team_size"] == 128
    x, st, _ = sysobj.solve_batch(x0)
    assert np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0)
    assert np.all(st["final_residual_inf"] <= 1e-8)
    assert st["iterations"].min() >= 3 and st["iterations"].max() <= 35
    late = np.argsort(st["iterations"])[-32:]
    sample = np.unique(np.concatenate([np.arange(0, B, B // 64)[:64], late]))
    needed = assert_batch_matches_oracle(recs, x0[sample], x[sample], st["iterations"][sample], st["converged"][sample],
                                         what="sketch150 x 32768")
    assert needed <= 2  # (measured in round 4: 0; profiles/r04_parity_bar.txt)
    x2, st2, _ = sysobj.solve_batch(x0)
    assert np.array_equal(st2["iterations"], st["iterations"]) and np.array_equal(x2, x)
    # one lane per system (the shape of larger batches) on the same starts: the same LM paths
    lanes = E.System(recs, n, team_size=E.TEAM_BATCH_LANES)
    xl, stl, _ = lanes.solve_batch(x0[:4096])
    assert np.array_equal(stl["converged"], st["converged"][:4096])
    assert np.mean(stl["iterations"] == st["iterations"][:4096]) >= 0.99
    assert np.max(np.abs(xl - x[:4096]) / np.maximum(1.0, np.abs(x[:4096]))) <= 1e-6
