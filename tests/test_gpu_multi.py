"""GPU tests (-m gpu) of everything that exists for more than one device, on the one device a test box has:

* the multi-device entry of the C ABI (`ezpz_multi_*`, `ezpz_system_solve_batch_multi`; include/ezpz_amd.h) -- with one
  device it must equal the single-device entry bit for bit, and with EZPZ_MULTI_OVERSUBSCRIBE=1 (three workers, three
  EzpzSystems, all on device 0) the sharded path itself runs: uneven shards, idle workers, masks at shard offsets,
  registered caller buffers, concurrent callers;
* `ezpz_amd.distributed.solve_batch_sharded` with the HIP local solve: two gloo ranks sharing GPU 0;
* bench.py's N>1 branches with the RCCL ("nccl") backend at world size 1 (EZPZ_BENCH_FORCE_DIST=1): process group on the
  device, barriers, all-reduces on device tensors, the scatter / gather extra;
* the registered-buffer pipeline on a system that runs lanes across the batch at every batch size (a round-2 host loop
  that never advanced).
"""
import json
import os
import subprocess
import sys
import textwrap

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


def _status_equal(a, b):
    return all(np.array_equal(a[f], b[f], equal_nan=True) if a[f].dtype.kind == "f" else np.array_equal(a[f], b[f])
               for f in a.dtype.names)


def test_one_device_mask_equals_the_single_device_entry_bitwise(E):
    for text, B in ((T.gen_big_problem(64), 300), (read_case("square"), 5000), (read_case("two_rectangles"), 777)):
        ref = T.load(text)
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        n = ref.num_vars
        x0 = ref.guesses[None, :] + gen.keyed_uniform(17, B, n, -0.2, 0.2)
        single = E.System(recs, n)
        xs, sts, ms = single.solve_batch(x0, want_mask=True)
        multi = E.MultiSystem(recs, n, device_mask=1)
        assert multi.devices() == [0] and multi.shard(B, 0) == (0, B)
        xm, stm, mm = multi.solve_batch(x0, want_mask=True)
        assert np.array_equal(xm, xs) and _status_equal(stm, sts) and np.array_equal(mm, ms)
        # mask 0 = every device of the node
        x_all, st_all, _ = E.MultiSystem(recs, n, device_mask=0).solve_batch(x0)
        assert np.array_equal(x_all, xs) and _status_equal(st_all, sts)
        # the one-call form, twice (second call served from the handle cache), and after the cache is dropped
        for _ in range(2):
            x1, st1 = E.solve_batch_multi(recs, n, x0, device_mask=1)
            assert np.array_equal(x1, xs) and _status_equal(st1, sts)
        E.lib().ezpz_cache_clear()
        x1, st1 = E.solve_batch_multi(recs, n, x0, device_mask=1)
        assert np.array_equal(x1, xs) and _status_equal(st1, sts)


def test_multi_entry_errors(E):
    ref = T.load(read_case("square"))
    recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
    ndev = E.device_count()
    with pytest.raises(E.NonLinearSystemError) as e:
        E.MultiSystem(recs, ref.num_vars, device_mask=1 << ndev)  # a device the node does not have
    assert e.value.code == -103
    with pytest.raises(E.NonLinearSystemError) as e:  # MissingGuess comes back with the offending constraint / variable
        E.MultiSystem(recs, ref.num_vars - 1, device_mask=1)
    assert e.value.code == -3 and e.value.variable == ref.num_vars - 1
    multi = E.MultiSystem(recs, ref.num_vars, device_mask=1)
    x, st, _ = multi.solve_batch(np.zeros((0, ref.num_vars)))
    assert x.shape == (0, ref.num_vars) and len(st) == 0


OVERSUBSCRIBED = textwrap.dedent("""
    import os, sys, threading
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import ezpz_amd as E, gen
    from conftest import read_case
    from oracle import oracle as O
    from oracle import textual as T

    def same_status(a, b):
        return all(np.array_equal(a[f], b[f]) for f in a.dtype.names)

    for text, sizes in ((T.gen_big_problem(64), (1, 2, 3, 7, 10, 64, 301, 4000)), (read_case("square"), (5, 1000, 65536 + 11)),
                        (read_case("inconsistent"), (13,))):
        ref = T.load(text)
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        n = ref.num_vars
        single = E.System(recs, n)
        multi = E.MultiSystem(recs, n, device_mask=0b10101)  # workers 0, 2, 4 -> three EzpzSystems on device 0
        assert multi.devices() == [0, 0, 0]
        for B in sizes:
            x0 = ref.guesses[None, :] + gen.keyed_uniform(100 + B, B, n, -0.2, 0.2)
            xs, sts, ms = single.solve_batch(x0, want_mask=True)
            xm, stm, mm = multi.solve_batch(x0, want_mask=True)
            shards = [multi.shard(B, g) for g in range(3)]
            per = -(-B // 3)
            assert shards == [(min(B, g * per), min(B, (g + 1) * per) - min(B, g * per)) for g in range(3)], shards
            assert np.array_equal(xm, xs) and same_status(stm, sts) and np.array_equal(mm, ms), B
        # specialised kernels on every device, results unchanged
        if multi.specialize(wait=True) == 2:
            xm2, stm2, _ = multi.solve_batch(x0)
            single.specialize(wait=True)
            xs2, sts2, _ = single.solve_batch(x0)
            assert np.array_equal(xm2, xs2) and same_status(stm2, sts2)

    # registered caller buffers (each shard pipelined on its own worker), and two threads calling one handle at once
    ref = T.load(T.gen_big_problem(64))
    recs, n = O.stack(ref.constraints), ref.num_vars
    multi = E.MultiSystem(recs, n, device_mask=0b111)
    B = 30000  # 61 MB of guesses
    x0 = np.ascontiguousarray(ref.guesses[None, :] + gen.keyed_uniform(9, B, n, -0.25, 0.25))
    want, wst, _ = E.System(recs, n).solve_batch(x0)
    xo, st = np.empty_like(x0), np.zeros(B, dtype=E.STATUS_DTYPE)
    E.host_register(x0); E.host_register(xo)
    try:
        multi.solve_batch(x0, out=(xo, st))
        assert np.array_equal(xo, want) and same_status(st, wst)
    finally:
        E.host_unregister(x0); E.host_unregister(xo)
    results = [None, None]
    def call(i):
        results[i] = multi.solve_batch(x0[i * 1000:(i + 1) * 1000 + 77])
    ts = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for i in range(2):
        assert np.array_equal(results[i][0], want[i * 1000:(i + 1) * 1000 + 77])
    # a ragged batch of three topologies sharded over the three workers (ezpz_multi_solve_batch_mixed): every device index
    # runs its contiguous shard through the single-device heterogeneous entry; bitwise the per-topology answers
    tops, rows_of, singles, multis = [], [], [], []
    for k, name in enumerate(("circle_tangent", "parallelogram", "arc_radius")):
        r = T.load(read_case(name))
        rc_ = O.stack([O.set_from_initial_values(c, r.guesses) for c in r.constraints])
        singles.append(E.System(rc_, r.num_vars))
        multis.append(E.MultiSystem(rc_, r.num_vars, device_mask=0b111))
        rows_of.append(r.guesses[None, :] + gen.keyed_uniform(70 + k, 4000, r.num_vars, -0.1, 0.1))
    rng = np.random.default_rng(5)
    for B in (1, 2, 10, 9001):
        topo = rng.integers(0, 3, B).astype(np.uint32)
        used = [0, 0, 0]
        parts, want_x, want_it = [], [], []
        per_top = [singles[k].solve_batch(rows_of[k]) for k in range(3)]
        for b in range(B):
            k = int(topo[b]); i = used[k]; used[k] += 1
            parts.append(rows_of[k][i]); want_x.append(per_top[k][0][i]); want_it.append(per_top[k][1]["iterations"][i])
        xm, stm = E.solve_batch_mixed_multi(multis, topo, np.concatenate(parts))
        assert np.array_equal(xm, np.concatenate(want_x)) and np.array_equal(stm["iterations"], np.array(want_it)), B
        x1, st1 = E.solve_batch_mixed(singles, topo, np.concatenate(parts))
        assert np.array_equal(x1, xm) and same_status(st1, stm), B
    # a batch that crosses a launch-shape threshold when it is sharded: one connected sketch, more systems than lanes across
    # the batch start at (64 x 4 x CUs), so the single-device call runs one lane per system while each of the three shards
    # stays below the threshold and runs on the per-system teams -- another elimination order, another rounding: the header
    # promises each shard the single-device entry's answer FOR THAT SHARD (checked bit for bit) and agreement with the
    # whole-batch call to rounding (iteration counts, flags, coordinates at 1e-9 on this well-conditioned sketch)
    recs, g = gen.connected_sketch(12, 1005)
    n = len(g)
    import torch

    pol = E.launch_policy(torch.cuda.get_device_properties(0).multi_processor_count)  # (0 would be the 256-CU table)
    B = int(pol.lanes_min_systems_small) + 300
    x0 = g[None, :] + gen.keyed_uniform(77, B, n, -0.02, 0.02)
    single = E.System(recs, n)
    multi = E.MultiSystem(recs, n, device_mask=0b111)
    xs, sts, _ = single.solve_batch(x0)
    xm, stm, _ = multi.solve_batch(x0)
    for gi in range(3):
        lo, cnt = multi.shard(B, gi)
        assert cnt < pol.lanes_min_systems_small
        xg, stg, _ = single.solve_batch(x0[lo:lo + cnt])
        assert np.array_equal(xm[lo:lo + cnt], xg) and same_status(stm[lo:lo + cnt], stg), gi
    assert np.array_equal(stm["converged"], sts["converged"]) and np.array_equal(stm["n_unsatisfied"], sts["n_unsatisfied"])
    assert np.mean(stm["iterations"] == sts["iterations"]) > 0.999
    assert np.allclose(xm, xs, rtol=0, atol=1e-9)
    print("oversubscribed ok")
""")


def test_sharded_path_with_three_workers_on_one_device(E):
    env = dict(os.environ, EZPZ_MULTI_OVERSUBSCRIBE="1")
    r = subprocess.run([sys.executable, "-c", OVERSUBSCRIBED % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "oversubscribed ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


SHARDED_RANK = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import ezpz_amd as E, gen
    from ezpz_amd.distributed import solve_batch_sharded
    from oracle import oracle as O
    from oracle import textual as T

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)  # both ranks share the box's one GPU; gloo moves host tensors
    ref = T.load(T.gen_big_problem(64))
    recs, n = O.stack(ref.constraints), ref.num_vars
    system = E.System(recs, n, device=0)

    def local_solve(x0_cpu):  # the HIP path of this rank around gloo's host tensors
        x0 = x0_cpu.to(dev)
        x = torch.empty_like(x0)
        st = torch.zeros((x0.shape[0], 32), dtype=torch.uint8, device=dev)
        system.solve_batch_device(x0.data_ptr(), x0.shape[0], x.data_ptr(), st.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        return x.cpu(), st.cpu()

    for B in (1, 7, 1000):
        x0 = torch.from_numpy(ref.guesses[None, :] + gen.keyed_uniform(3, B, n, -0.25, 0.25)) if rank == 0 else None
        x, st = solve_batch_sharded(system, x0, n, root=0, local_solve=local_solve)
        if rank == 0:
            want, wst, _ = system.solve_batch(x0.numpy())
            got = st.numpy().view(E.STATUS_DTYPE).reshape(-1)
            assert np.array_equal(x.numpy(), want) and np.array_equal(got["iterations"], wst["iterations"]), B
            assert np.array_equal(got["converged"], wst["converged"])
    # and the default local solve (device tensors end to end) on a one-rank group of this process: the code path RCCL runs
    groups = [dist.new_group([r], backend="gloo") for r in range(world)]  # (every rank creates every group)
    solo = groups[rank]
    x0 = torch.from_numpy(ref.guesses[None, :] + gen.keyed_uniform(4 + rank, 500, n, -0.25, 0.25))
    x, st = solve_batch_sharded(system, x0.to(dev), n, root=0, group=solo, device=dev)
    torch.cuda.synchronize(dev)
    want, wst, _ = system.solve_batch(x0.numpy())
    assert np.array_equal(x.cpu().numpy(), want)
    assert np.array_equal(st.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)["iterations"], wst["iterations"])
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_two_gloo_ranks_on_one_gpu_run_the_sharded_solve_on_the_hip_path(E, tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(SHARDED_RANK % {"root": ROOT})
    port = 29700 + os.getpid() % 200
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and f"rank {r} ok" in out, (out[-1000:], err[-4000:])


def test_bench_takes_its_multi_gpu_branches_with_rccl_at_world_size_one(E):
    env = dict(os.environ, EZPZ_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--batch", "2048",
                        "--cpu-seconds", "0", "--pmc", "0", "--legs", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["results_ok"] is True
    assert line["warmup_launches_run"] >= 2 + 1 + 2
    extras = line["extras"]
    assert "with_rccl_scatter_gather_error" not in extras, extras
    assert extras["with_rccl_scatter_gather_solves_per_s"] > 0 and extras["rccl_scatter_gather_equals_device_path"] is True


def test_registered_buffers_on_a_system_that_always_runs_lanes_across_the_batch(E):
    """EZPZ_TEAM_BATCH_LANES + both buffers registered + more than 1 MiB: the pipelined path used to clamp its piece to
    lanes_min - 1 = 0 systems and never advance.  Such systems take the chunked path."""
    recs, g = gen.connected_sketch(40, 7)
    n = len(g)
    B = 4000  # 2.5 MB of guesses
    x0 = np.ascontiguousarray(g[None, :] + gen.keyed_uniform(1, B, n, -0.02, 0.02))
    system = E.System(recs, n, team_size=E.TEAM_BATCH_LANES)
    want, wst, _ = system.solve_batch(x0)
    xo, st = np.empty_like(x0), np.zeros(B, dtype=E.STATUS_DTYPE)
    E.host_register(x0)
    E.host_register(xo)
    try:
        import ctypes as C

        cfg = E.Config()._c()
        rc = E.lib().ezpz_system_solve_batch(system._h, x0.ctypes.data, B, C.byref(cfg), xo.ctypes.data, st.ctypes.data, None, None, 0)
        assert rc == 0
    finally:
        E.host_unregister(x0)
        E.host_unregister(xo)
    assert np.array_equal(xo, want) and np.array_equal(st["iterations"], wst["iterations"])
    rc, xr, it, conv, _ = O.solve_batch(recs, x0[:64], linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0 and np.all(np.abs(xo[:64] - xr) <= 1e-6 * np.maximum(1.0, np.abs(xr)))


def test_host_entries_leave_the_callers_device_as_it_was(E):
    """Every host entry that works on a system of device 1 -- create, batch solve between host buffers and on device pointers,
    evaluation, FreedomAnalysis, the heterogeneous batch and its destructor, destroy -- puts the calling thread's current HIP device
    back (the library's own runtime: a caller on an 8-GPU node must not find itself on another device after a call).  Needs two
    devices; the single-GPU boxes of the build skip it."""
    if E.device_count() < 2:
        pytest.skip("needs two HIP devices")
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # (the runtime the library itself is linked against: already loaded)
    recs, g = gen.connected_sketch(30, 5)
    n = len(g)
    assert hip.hipSetDevice(0) == 0 and E.lib().ezpz_current_device() == 0
    s = E.System(recs, n, device=1)
    assert E.lib().ezpz_current_device() == 0
    x0 = np.tile(g, (4, 1))
    x, st, _ = s.solve_batch(x0)
    assert E.lib().ezpz_current_device() == 0 and np.all(st["converged"] == 1)
    s.eval_batch(x)
    assert E.lib().ezpz_current_device() == 0
    s.freedom_batch(x)
    assert E.lib().ezpz_current_device() == 0
    mixed = E.MixedBatch([s], np.zeros(4, np.uint32))
    assert E.lib().ezpz_current_device() == 0
    del mixed
    assert E.lib().ezpz_current_device() == 0
    del s
    assert E.lib().ezpz_current_device() == 0
