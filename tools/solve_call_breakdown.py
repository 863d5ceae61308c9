"""Where the time of ONE ezpz_solve call goes (the protocol of the reference's published figure and of every criterion
benchmark: ezpz-cli/src/main.rs:86-100, ezpz/benches/solver_bench.rs:15-24): per fixture the stages of a warm call
(the library's own stamps, ezpz_debug_call_trace), the kernel's duration on the device (HIP events around launches on
device buffers, the same kernel the call used) and the first call of the process.

usage (GPU box): python tools/solve_call_breakdown.py > profiles/r04_solve_call_breakdown.txt
The kernel durations come from a kernel trace of the same systems:
    rocprofv3 --kernel-trace -d gpurun_out/scb -o scb --output-format csv -- python3 tools/solve_call_breakdown.py --launch-only
    python tools/solve_call_breakdown.py --from-trace gpurun_out/scb/.../scb_kernel_trace.csv   (appends the per-case table)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

t_import0 = time.perf_counter()
import ezpz_amd as E  # noqa: E402
from ezpz_amd._lib import COutcome  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import textual as T  # noqa: E402

STAGES = {1: "enter", 2: "request recognised (compare / hash)", 3: "sides, tier, scratch", 4: "lock + device", 5: "guesses staged",
          6: "kernel enqueued", 7: "completion seen", 8: "results unpacked", 9: "unsatisfied / warnings / outcome", 10: "return",
          20: "cold: plan built", 21: "cold: analysed", 22: "cold: uploaded", 23: "cold: kernel found"}


def case(name):
    text = open(os.path.join(ROOT, "tests", "golden", "test_cases", name, "problem.md")).read()
    ref = T.load(text)
    return ref.constraints, ref.variables()


def massive(lines):
    ref = T.load(T.gen_big_problem(lines))
    return ref.constraints, ref.variables()


def two_rectangles_dependent():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import reference_benches as R
    return R.two_rectangles_dependent()


CASES = [("tiny", lambda: case("tiny")), ("nonsquare", lambda: case("nonsquare")), ("inconsistent", lambda: case("inconsistent")),
         ("square", lambda: case("square")), ("two_rectangles", lambda: case("two_rectangles")),
         ("two rectangles dependent", two_rectangles_dependent), ("arc_radius", lambda: case("arc_radius")),
         ("massive 800 x 800", lambda: massive(200)), ("massive 2000 x 2000", lambda: massive(500)),
         ("massive 2400 x 2400", lambda: massive(600))]


def main():
    L = E.lib()
    first = True
    print("# python tools/solve_call_breakdown.py -- one ezpz_solve call per iteration, stages in microseconds (mean of the warm calls)")
    for name, build in CASES:
        reqs, guesses = build()
        recs = O.stack(reqs)
        ids = np.ascontiguousarray([g[0] for g in guesses], dtype=np.uint32)
        vals = np.ascontiguousarray([g[1] for g in guesses], dtype=np.float64)
        n = len(vals)
        cfg, out = E.Config()._c(), COutcome()
        x_out, unsat = np.zeros(n), np.zeros(len(recs) + 1, dtype=np.uint64)
        args = [recs.ctypes.data, len(recs), ids.ctypes.data, vals.ctypes.data, n, C.byref(cfg), x_out.ctypes.data, unsat.ctypes.data,
                None, 0, C.byref(out)]
        call = lambda: L.ezpz_solve(*args)
        trace = np.zeros(64, dtype=np.uint64)
        if first:  # the process's first solve: runtime initialisation, code object load, symbolic phase, first launch
            L.ezpz_debug_call_trace(trace.ctypes.data, trace.size)
            t0 = time.perf_counter()
            assert call() == 0
            dt = (time.perf_counter() - t0) * 1e3
            k = L.ezpz_debug_call_trace(None, 0)
            st = trace[:k].reshape(-1, 2)
            parts = ", ".join(f"{STAGES[int(b[0])]} +{(int(b[1]) - int(a[1])) / 1e6:.2f} ms" for a, b in zip(st[:-1], st[1:]))
            print(f"first solve of the process ({name}): {dt:.1f} ms  [{parts}]")
            first = False
        assert call() == 0
        want = O.solve(reqs, guesses, linsolve=O.LINSOLVE_SPARSE)
        iters_equal = out.iterations == want.iterations
        # cold: the request cache dropped before every call (symbolic phase + first launch of the new system)
        cold = 0.0
        cold_acc = {}
        for _ in range(5):
            L.ezpz_cache_clear()
            L.ezpz_debug_call_trace(trace.ctypes.data, trace.size)
            t0 = time.perf_counter()
            call()
            cold += (time.perf_counter() - t0) / 5 * 1e6
            k = L.ezpz_debug_call_trace(None, 0)
            st = trace[:k].reshape(-1, 2)
            for a, b in zip(st[:-1], st[1:]):
                cold_acc[int(b[0])] = cold_acc.get(int(b[0]), 0.0) + (int(b[1]) - int(a[1])) / 1e3 / 5
        for _ in range(400):  # past the point where the topology's specialised kernel takes over (256 solves, or the on-disk cache)
            call()
        time.sleep(1.0)
        for _ in range(50):
            call()
        reps = 300
        acc = {}
        total = 0.0
        for _ in range(reps):
            L.ezpz_debug_call_trace(trace.ctypes.data, trace.size)
            call()
            k = L.ezpz_debug_call_trace(None, 0)
            st = trace[:k].reshape(-1, 2)
            for a, b in zip(st[:-1], st[1:]):
                acc[int(b[0])] = acc.get(int(b[0]), 0.0) + (int(b[1]) - int(a[1])) / 1e3
            total += (int(st[-1][1]) - int(st[0][1])) / 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        wall = (time.perf_counter() - t0) / reps * 1e6
        secs, _ = O.time_solves(reqs, guesses, repeats=100, linsolve=O.LINSOLVE_SPARSE)
        cpu = secs / 100 * 1e6
        print(f"## {name}: {out.num_eqs} rows x {n} vars, {out.iterations} iterations ({'equal to' if iters_equal else 'DIFFERENT from'} the CPU port's)")
        print(f"   warm call {wall:6.1f} us (ctypes included; inside the library {total / reps:6.1f} us) | cold {cold:7.1f} us | CPU port, 1 core {cpu:8.1f} us | speed-up {cpu / wall:6.2f}x")
        print("   warm: " + " | ".join(f"{STAGES[k]} {v / reps:.1f}" for k, v in sorted(acc.items())))
        order = [20, 21, 22, 23, 2, 3, 4, 5, 6, 7, 8, 9, 10]
        print("   cold: " + " | ".join(f"{STAGES[k]} {cold_acc[k]:.1f}" for k in order if k in cold_acc))
    print(f"# device: {E.device_count()} x HIP device; stamps cost ~0.03 us each")


KERNEL_LAUNCHES = 200


def kernel_us(recs, n, vals):
    """Back-to-back launch period of one solve's kernel on device buffers (a RATE: launches overlap their own overheads;
    the kernel's duration proper comes from the kernel trace, --launch-only / --from-trace)."""
    import torch
    dev = torch.device("cuda", 0)
    s = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
    s.specialize(wait=True)
    x0 = torch.from_numpy(vals[None, :].copy()).to(dev)
    xo = torch.empty_like(x0)
    st = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(KERNEL_LAUNCHES):
        s.solve_batch_device(x0.data_ptr(), 1, xo.data_ptr(), st.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / KERNEL_LAUNCHES * 1e6


SOLVE_KERNELS = ("ezpz_jit_", "lm_solve_kernel", "comp_solve_kernel", "batch_lane_kernel")


def launch_only():
    """Under rocprofv3 --kernel-trace: exactly KERNEL_LAUNCHES launches of each case's one-solve kernel, in CASES order."""
    for name, build in CASES:
        reqs, guesses = build()
        recs = O.stack(reqs)
        vals = np.ascontiguousarray([g[1] for g in guesses], dtype=np.float64)
        kernel_us(recs, len(vals), vals)


def from_trace(path):
    import csv
    rows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in SOLVE_KERNELS)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == KERNEL_LAUNCHES * len(CASES), (len(rows), KERNEL_LAUNCHES * len(CASES))
    print("# kernel durations of one solve (rocprofv3 --kernel-trace of tools/solve_call_breakdown.py --launch-only), microseconds")
    print("# case | kernel | workgroups x lanes | median | min | max")
    for i, (name, _) in enumerate(CASES):
        part = rows[i * KERNEL_LAUNCHES:(i + 1) * KERNEL_LAUNCHES][20:]
        d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in part])
        r0 = part[0]
        kn = r0["Kernel_Name"].split("(")[0][:60]
        print(f"{name} | {kn} | {int(r0['Grid_Size_X']) // int(r0['Workgroup_Size_X'])} x {r0['Workgroup_Size_X']} | {np.median(d):.2f} | {d.min():.2f} | {d.max():.2f}")


if __name__ == "__main__":
    if "--launch-only" in sys.argv:
        launch_only()
    elif "--from-trace" in sys.argv:
        from_trace(sys.argv[sys.argv.index("--from-trace") + 1])
    else:
        main()
