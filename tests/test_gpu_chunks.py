"""What the kernels that span workgroups rest on, pinned by dedicated stress tests (round 5's review, "Test robustness"):
* a 16-byte (value, sequence number) chunk moved by one device-coherent 128-bit store is observed WHOLE by a 128-bit load on another
  compute unit (jit_kernel.hip.hpp: grid_store / grid_peek and the ring of solve_kernel_grid_fast, grid_ops.hip.hpp,
  front_kernel.hip.hpp) -- not architecturally promised in so many words: tests/chunk_stress.hip sweeps 1e8 exchanges with a
  torn-value detector on every chunk looked at;
* a system on several workgroups beside a CO-TENANT that holds part of the device (another process, as on a shared node): its
  workgroups wait for each other, so the launch needs every one of them resident -- the solve must come out right, late, not as
  EZPZ_ITERATIONS_TEAM_TIMEOUT, as long as the co-tenant leaves within the wait's bound (~1 s)."""
import os
import subprocess
import time

import numpy as np
import pytest

import gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stress(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("chunks") / "chunk_stress")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tests", "chunk_stress.hip"), "-o", exe])
    return exe


def test_sixteen_byte_chunks_are_observed_whole(stress):
    """256 pairs of workgroups x 8 lanes x 25 000 round trips of two exchanges each = 1.02e8 exchanges; every chunk a lane looks at
    on the way (several per exchange) must be consistent with its own sequence number."""
    pairs, trips = 256, 25000
    out = subprocess.check_output([stress, "pairs", str(pairs), str(trips)], text=True, timeout=600).split()
    exchanges, looked, torn, timeouts = map(int, out)
    assert exchanges == 2 * pairs * 8 * trips and exchanges >= 10**8
    assert looked >= exchanges and torn == 0 and timeouts == 0, (looked, torn, timeouts)


def test_a_system_on_several_workgroups_beside_a_co_tenant(stress):
    """The 48 000-variable ladder (24 workgroups per system, as many systems in flight as the device holds) while another PROCESS
    holds 40 % of the device's LDS-limited places for 0.4 s: the launches that do not fit wait, then finish with the right
    answers and ordinary statuses -- in both kernels (the one that does not wait for verdicts, and the loop: EZPZ_JIT_AHEAD=0 is
    covered by the same path through the redo list of a batch of already-solved systems)."""
    import ezpz_amd as E
    from oracle import oracle as O
    from oracle import textual as T

    ref = T.load(T.gen_big_problem(12000))
    n = ref.num_vars
    sysobj = E.System(ref.constraints, n)
    assert sysobj.specialize(wait=True) == 2
    B = 40
    x0 = ref.guesses[None, :] + gen.keyed_uniform(91, B, n, -0.25, 0.25)
    exact = np.zeros(n)
    exact[0::4] = exact[2::4] = np.arange(12000)
    exact[3::4] = 4.0
    x0[1::4] = exact  # (these go through the loop kernel: converged at the start)
    rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0, linsolve=O.LINSOLVE_SPARSE)
    assert rc == 0
    x, st, _ = sysobj.solve_batch(x0)  # warm
    assert np.array_equal(x, xo)
    import torch

    cus = int(torch.cuda.get_device_properties(0).multi_processor_count)
    # 160 KB of LDS per workgroup: one per compute unit, nothing else fits beside it there
    tenant = subprocess.Popen([stress, "occupy", str(int(0.4 * cus)), str(160 * 1024), "400"], stdout=subprocess.PIPE, text=True)
    try:
        assert tenant.stdout.readline().strip() == "occupying"
        t0 = time.perf_counter()
        for _ in range(3):
            x, st, _ = sysobj.solve_batch(x0)
            assert np.array_equal(st["iterations"], it) and np.array_equal(st["converged"], conv), st["iterations"]
            assert np.array_equal(x, xo)
        waited = time.perf_counter() - t0
    finally:
        tenant.wait(timeout=30)
    assert tenant.returncode == 0
    print(f"three calls beside the co-tenant: {waited * 1e3:.0f} ms")
