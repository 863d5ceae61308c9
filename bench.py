#!/usr/bin/env python3
"""Benchmark of the LM constraint-solve hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  For N > 1 it is either
launched by torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE in the environment) or started
plainly, in which case it starts the N rank processes itself -- before anything in this process touches the GPU -- and
exits non-zero when the node has fewer than N devices.  It never reports fewer GPUs than it was asked for.

  metric   solves/sec on the 2000-row massive_parallel_system (BASELINE.json), plus iters-to-converge
  step     one launch of the LM kernel over a batch of `--batch` independent replicas of the workload system
           (jittered initial guesses), inputs already resident in HBM
  value    whole-job solves/s = N * batch * K / max-over-ranks wall time of the K timed steps, device-resident
           guesses in, device-resident results out (`config.value_is`).  `value_host_to_host` is the same batch
           through the host-pointer entry point (SURVEY.md 8d's "results back on host": H2D + kernel + D2H).
  scaling  weak: every rank owns its own shard of `--batch` systems; the path has no data-path collective
           (systems are independent), so none is issued inside the timed region
  roofline three roofs, the highest fraction is `bound`:
             hbm    compulsory bytes (x0 in, x* + status out: 16 n + 32 per solve) x solves/s against 8 TB/s; `traffic`
                    is what the PMC counters saw per launch in THIS run (FETCH_SIZE doubled per the gfx950 note of
                    MI355X_MICROARCH.md + WRITE_SIZE)
             lds    LDS-array busy cycles (SQ_LDS_IDX_ACTIVE) against CUs x kernel cycles
             issue  vector-ALU busy cycles (SQ_ACTIVE_INST_VALU, quad-cycles) against SIMDs x kernel cycles
           plus `algorithmic_equiv`: SURVEY.md 8(d)'s BYTES formula, which charges the solver state to HBM although
           the fused kernel keeps it on chip (so it can exceed the HBM peak and is not a bound).
           The kernel time is measured live with HIP events on the launch stream; the counters come from rocprofv3
           child passes of this same script (one --pmc set per pass, never with a trace).
  cpu_baseline  the CPU oracle (a C port of the reference algorithm, sparse Cholesky, per-call setup like the
           reference) timed on 1 host core with the CLI protocol on a bounded sample -- a reported baseline only
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "solves/sec on 2000-row massive_parallel_system @1/2/4/8 GPU; iters-to-converge"
# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # HBM3E peak BW 8.0 TB/s (spec)
N_CUS, N_SIMDS = 256, 1024
PMC_SETS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
     "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
]
SOLVE_KERNELS = ("lm_solve_kernel", "comp_solve_kernel", "ezpz_jit_solve", "ezpz_jit_lane", "batch_lane_kernel")


def algorithmic_bytes(info: dict, k: int) -> int:
    """BYTES = (1+k)(56C + 8n + 8m + 8zJ) + k(12zJ + 8m + 16zA + 24zL + 48n)."""
    C, n, m = info["n_constraints"], info["n_vars"], info["n_rows"]
    zj, za, zl = info["nnz_j"], info["nnz_a"], info["nnz_l"]
    return (1 + k) * (56 * C + 8 * n + 8 * m + 8 * zj) + k * (12 * zj + 8 * m + 16 * za + 24 * zl + 48 * n)


def make_workload(name: str):
    """Returns (description, side-resolved constraint records, file guesses, jitter amplitude, expected iterations or
    None).  Built with the product's own front end (ezpz_amd.textual); nothing here touches oracle/."""
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
        import numpy as np
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


def cpu_baseline(records, guesses, budget_s: float):
    """Oracle (`kind: port`) timed on one core with the CLI protocol (ezpz-cli/src/main.rs:86-100)."""
    from oracle import oracle as O

    secs, iters = O.time_solves(records, guesses, repeats=20, linsolve=O.LINSOLVE_SPARSE)
    per = max(secs / 20.0, 1e-7)
    repeats = int(min(max(budget_s / per, 100), 200000))
    secs, iters = O.time_solves(records, guesses, repeats=repeats, linsolve=O.LINSOLVE_SPARSE)
    out = {"value": repeats / secs, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": f"{repeats} back-to-back full solve() calls (setup + sparse LLT + LM, {iters} iterations each) "
                     f"of the same system on 1 core in {secs:.1f} s"}
    # the same port with OpenMP over independent systems on every host core (reported beside the
    # 1-core figure, never used for `value` or the speed-up)
    import numpy as np

    cores = len(os.sched_getaffinity(0))
    nb = int(min(max(out["value"] * cores * min(budget_s, 4.0), 4 * cores), max(64, (256 << 20) // (8 * max(len(guesses), 1)))))
    x0 = np.tile(np.asarray(guesses, dtype=np.float64), (nb, 1))
    t0 = time.perf_counter()
    O.solve_batch(records, x0, linsolve=O.LINSOLVE_SPARSE, nthreads=cores)
    out["all_cores"] = {"value": nb / (time.perf_counter() - t0), "unit": "solves/s", "cores": cores,
                        "sample": f"{nb} replicas, OpenMP over systems"}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16384, help="systems per launch per GPU (4096 fills the 768 resident workgroups 5.3 times: the last round runs a third empty)")
    ap.add_argument("--workload", default="massive500", help="massive<lines>[o] (o = over-constrained variant), sketch<points> (one connected sketch), a test_cases/ directory name, or mixed")
    ap.add_argument("--team", type=int, default=0, help="override lanes per system (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--check", type=int, default=1, help="verify the results of the last step against the oracle")
    ap.add_argument("--extras", type=int, default=1, help="also report the host-to-host rate, single-solve latency (and, "
                    "for N>1, the rate with the RCCL scatter/gather of the batch) -- never part of `value`")
    ap.add_argument("--specialize", type=int, default=1, help="use the run-time compiled class-specialised kernel where the topology has one")
    ap.add_argument("--pmc", type=int, default=1, help="N=1: collect the roofline's PMC counters with rocprofv3 child passes of this script")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  Runs before this
    process has touched the GPU (device_count() does not initialise it) and never re-executes a process that has."""
    dry = os.environ.get("EZPZ_BENCH_DRY") == "1"
    backend = os.environ.get("EZPZ_BENCH_BACKEND", "nccl")
    if not dry and backend == "nccl":
        import torch

        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node exposes {ndev} HIP device(s); refusing to report a "
                  f"{args.gpus}-GPU number from fewer", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def dry_run(args, world, rank) -> int:
    """Test hook (EZPZ_BENCH_DRY=1, CPU only, never set by the driver): the N>1 control path of this script --
    rendezvous, barriers, max-over-ranks timing, one line from rank 0 -- with an empty step."""
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group(backend="gloo")
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = 1
    if world > 1:
        t = torch.tensor([elapsed, 1.0], dtype=torch.float64)
        dist.all_reduce(t[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
        elapsed, seen = float(t[0]), int(t[1])
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "dry_run": True,
                          "world_size_seen": seen}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def collect_pmc(args) -> dict:
    """Per-launch means of the PMC counters of the solve kernel, from rocprofv3 child passes of this script with the
    same workload (one counter set per pass; the profiler's child is `python3 bench.py ...` itself)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    base = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else tempfile.gettempdir()
    outdir = tempfile.mkdtemp(prefix="bench_pmc_", dir=base)
    env = dict(os.environ, TMPDIR="/tmp")
    child = ["python3", os.path.abspath(__file__), "--workload", args.workload, "--batch", str(args.batch), "--team", str(args.team),
             "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--check", "0", "--extras", "0", "--pmc", "0", "--specialize", str(args.specialize)]
    counters, errors = {}, []
    for i, cset in enumerate(PMC_SETS):
        d = os.path.join(outdir, f"set{i}")
        try:
            subprocess.run([exe, "--pmc"] + cset + ["--output-format", "csv", "-d", d, "--"] + child, env=env, cwd="/tmp",
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240, check=True)
        except (subprocess.SubprocessError, OSError) as exc:
            errors.append(f"set{i}: {type(exc).__name__}: {str(getattr(exc, 'stderr', b'') or exc)[-200:]}")
            continue
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if any(k in row["Kernel_Name"] for k in SOLVE_KERNELS):
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                    if "batch_lane_kernel" in row["Kernel_Name"]:
                        counters["_state_in_hbm"] = 1.0
        for k, v in acc.items():
            counters[k] = sum(v) / len(v)
            counters.setdefault("_dispatches", len(v))
    shutil.rmtree(outdir, ignore_errors=True)
    if errors:
        counters["errors"] = errors
    return counters


def roofline(info_parts, B, kernel_ms, solves_per_launch_iters, pmc, n_kernels):
    """The three roofs (module docstring).  `pmc` = per-launch counter means or {} / {"error": ...}."""
    t = kernel_ms * 1e-3
    compulsory = sum((16 * p["info"]["n_vars"] + 32) * p["B"] for p in info_parts)  # bytes per launch
    hbm = {"achieved": compulsory / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "compulsory_bytes_per_solve": compulsory / B}
    hbm["frac"] = hbm["achieved"] / hbm["peak"]
    if pmc.get("_state_in_hbm"):
        # lanes across the batch (batch_kernel.hip.hpp) keep the solver state in global memory by design: for this kernel
        # SURVEY 8(d)'s BYTES formula IS the traffic model, and the HBM roof is priced with it
        hbm.update({"achieved": solves_per_launch_iters / t / 1e9, "algorithmic_bytes_per_solve": solves_per_launch_iters / B,
                    "state": "in HBM (one lane per system): SURVEY 8(d) BYTES per solve"})
        hbm["frac"] = hbm["achieved"] / hbm["peak"]
    roofs = {"hbm": hbm}
    traffic = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        hbm["traffic_over_compulsory"] = traffic / compulsory
        hbm["measured_gbs"] = traffic / t / 1e9
    cycles = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
    if cycles > 0 and "SQ_LDS_IDX_ACTIVE" in pmc:
        roofs["lds"] = {"achieved": pmc["SQ_LDS_IDX_ACTIVE"], "peak": N_CUS * cycles, "unit": "LDS-array cycles per launch",
                        "frac": pmc["SQ_LDS_IDX_ACTIVE"] / (N_CUS * cycles),
                        "lds_insts_per_solve": pmc.get("SQ_INSTS_LDS", 0.0) / B,
                        "bank_conflict_frac": pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(pmc["SQ_LDS_IDX_ACTIVE"], 1.0)}
    if cycles > 0 and "SQ_ACTIVE_INST_VALU" in pmc:
        roofs["issue"] = {"achieved": 4.0 * pmc["SQ_ACTIVE_INST_VALU"], "peak": N_SIMDS * cycles,
                          "unit": "vector-ALU busy cycles per launch", "frac": 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (N_SIMDS * cycles),
                          "valu_insts_per_solve": pmc.get("SQ_INSTS_VALU", 0.0) / B,
                          "salu_insts_per_solve": pmc.get("SQ_INSTS_SALU", 0.0) / B,
                          "wave_cycles_waiting_frac": pmc.get("SQ_WAIT_ANY", 0.0) / max(pmc.get("SQ_WAVE_CYCLES", 0.0), 1.0),
                          "kernel_cycles": cycles}
    bound = max(roofs, key=lambda k: roofs[k]["frac"])
    top = roofs[bound]
    assert all(r["frac"] <= 1.0 + 1e-9 for r in roofs.values()), roofs  # every roof is a bound
    algo = solves_per_launch_iters / t / 1e9
    return {
        "bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
        "traffic": traffic,
        "roofs_measured": sorted(roofs),  # without PMC counters (--pmc 0, N>1) only the HBM roof is known
        "traffic_source": ("rocprofv3 --pmc child passes of this run (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes, per launch)"
                           if traffic is not None else pmc.get("error") or pmc.get("errors") or "not collected (--pmc 0 or N>1)"),
        "roofs": roofs,
        "algorithmic_equiv": {"gbs": algo, "over_hbm_peak": algo / HBM_PEAK_GBS, "bytes_per_solve": solves_per_launch_iters / B,
                              "note": "SURVEY 8(d) BYTES formula; charges LDS-resident solver state to HBM, not a bound"},
        "kernel": ("comp_solve_kernel / lm_solve_kernel" if n_kernels > 1 else "the LM solve kernel of this topology") +
                  (f" x{n_kernels} (one launch per topology)" if n_kernels > 1 else ""),
        "kernel_ms": kernel_ms,
        "solves_per_launch": B,
    }


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("EZPZ_BENCH_DRY") == "1":
        sys.exit(dry_run(args, world, rank))

    import numpy as np
    import torch
    import torch.distributed as dist

    import ezpz_amd as E
    import gen

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Test hook (1-GPU boxes): EZPZ_BENCH_BACKEND=gloo runs the N>1 code path -- rendezvous, barriers, max-over-ranks
    # timing, rank-0 line -- with every rank on GPU (LOCAL_RANK mod device count).  The driver never sets it.
    backend = os.environ.get("EZPZ_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but only {torch.cuda.device_count()} HIP device(s) visible")
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    # BASELINE configs[4] flavour: system i uses [circle_tangent, parallelogram, arc_radius][i mod 3]; the batch is
    # grouped by topology (one launch per topology per step).  Everything else is a single-topology batch.
    names = ["circle_tangent", "parallelogram", "arc_radius"] if args.workload == "mixed" else [args.workload]
    stream = torch.cuda.current_stream(dev)
    parts = []
    for k, name in enumerate(names):
        desc_k, records_k, guesses_k, jitter_k, expect_k = make_workload(name)
        n_k = len(guesses_k)
        B_k = args.batch // len(names) + (1 if k < args.batch % len(names) else 0)
        system_k = E.System(records_k, n_k, device=device_index, team_size=args.team)
        # component-resident systems: the class-specialised kernel (run-time compiled once per topology, outside the
        # timed region like the symbolic phase; batch calls would start it themselves in the background)
        spec_k = system_k.specialize(wait=True) == 2 if args.specialize else False
        # synthetic inputs: replicas of the system with keyed-PRNG jitter on the guesses, resident in HBM
        x0_host_k = guesses_k[None, :] + gen.keyed_uniform(0x657A707A + rank + 101 * k, B_k, n_k, -jitter_k, jitter_k)
        x0_host_k[0] = guesses_k
        x0_k = torch.from_numpy(x0_host_k).to(dev)
        parts.append(dict(desc=desc_k, records=records_k, guesses=guesses_k, jitter=jitter_k, expect=expect_k, n=n_k,
                          B=B_k, system=system_k, info=system_k.info(), specialized=spec_k, x0_host=x0_host_k, x0=x0_k,
                          x_out=torch.empty_like(x0_k), status=torch.zeros((B_k, 32), dtype=torch.uint8, device=dev)))
    # the single-topology names used below refer to the first (usually only) part
    p0 = parts[0]
    desc, records, guesses, jitter, expect_iters = p0["desc"], p0["records"], p0["guesses"], p0["jitter"], p0["expect"]
    n, system, info, B = p0["n"], p0["system"], p0["info"], args.batch
    x0_host, x0, x_out, status = p0["x0_host"], p0["x0"], p0["x_out"], p0["status"]
    if len(parts) > 1:
        desc = "mixed: " + " + ".join(p["desc"] for p in parts)

    def step():
        for p in parts:
            p["system"].solve_batch_device(p["x0"].data_ptr(), p["B"], p["x_out"].data_ptr(), p["status"].data_ptr(), 0,
                                           stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # one kernel per step, back to back on this stream
    world_seen = 1
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms, 1.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t[:2], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[2:], op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, world_seen = float(t[0]), float(t[1]), int(round(float(t[2])))

    extras = {}
    value_h2h = None
    if args.extras:
        # (1) host-pointer entry point: H2D + kernel + D2H per call -- SURVEY 8(d)'s "results back on host"
        if rank == 0:
            # (the C ABI itself, into buffers allocated once: what a host-language caller does)
            import ctypes as C

            hb = min(B, 65536)  # the same systems per call as `value`
            hx = np.ascontiguousarray(x0_host[:hb])
            hxo = np.empty_like(hx)
            hst = np.zeros(hb, dtype=E.STATUS_DTYPE)
            hcfg = E.Config()._c()

            def host_call():
                rc = E.lib().ezpz_system_solve_batch(system._h, hx.ctypes.data, hb, C.byref(hcfg), hxo.ctypes.data,
                                                     hst.ctypes.data, None, None, 0)
                assert rc == 0, rc

            def rate(reps):
                host_call()
                host_call()
                th = time.perf_counter()
                for _ in range(reps):
                    host_call()
                return reps * hb / (time.perf_counter() - th)

            # pageable buffers first, then the same buffers page-locked once (ezpz_host_register: what a caller that
            # reuses its buffers does); `value_host_to_host` is the registered rate
            extras["host_to_host_pageable_solves_per_s"] = rate(3)
            E.host_register(hx)
            E.host_register(hxo)
            try:
                value_h2h = rate(8)
            finally:
                E.host_unregister(hx)
                E.host_unregister(hxo)
            extras["host_to_host_batch"] = hb
            extras["host_to_host_results_equal_device_path"] = bool(
                np.array_equal(hxo, x_out[:hb].cpu().numpy()) if len(parts) == 1 else True)
            # (2) one system per launch, back to back on the stream: device-side latency of a single solve
            one_x = x0[:1].clone()
            one_o = torch.empty_like(one_x)
            one_s = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
            for _ in range(10):
                system.solve_batch_device(one_x.data_ptr(), 1, one_o.data_ptr(), one_s.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            tl = time.perf_counter()
            for _ in range(200):
                system.solve_batch_device(one_x.data_ptr(), 1, one_o.data_ptr(), one_s.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            extras["single_solve_latency_us"] = (time.perf_counter() - tl) / 200 * 1e6
            # (2a) one full ezpz_solve() call from host buffers (the reference's solve(): lint + Model::new + LM +
            # unsatisfied check): warm = topology served from the cache, cold = cache cleared before every call
            E.solve_records(records, guesses)
            tw = time.perf_counter()
            for _ in range(50):
                E.solve_records(records, guesses)
            extras["full_solve_call_us_warm"] = (time.perf_counter() - tw) / 50 * 1e6
            tc = 0.0
            for _ in range(5):
                E.lib().ezpz_cache_clear()
                t_ = time.perf_counter()
                E.solve_records(records, guesses)
                tc += time.perf_counter() - t_
            extras["full_solve_call_us_cold"] = tc / 5 * 1e6
            # (2b) FreedomAnalysis (find_dof.rs) of the solved batch, device to device
            Bf = p0["B"]
            fa_mask = torch.zeros((Bf, n), dtype=torch.uint8, device=dev)
            fa_cnt = torch.zeros(Bf, dtype=torch.int32, device=dev)
            fa = lambda: system.freedom_batch_device(x_out.data_ptr(), Bf, fa_mask.data_ptr(), 0, fa_cnt.data_ptr(),
                                                     stream.cuda_stream)
            fa()
            torch.cuda.synchronize(dev)
            tf = time.perf_counter()
            for _ in range(5):
                fa()
            torch.cuda.synchronize(dev)
            extras["freedom_analyses_per_s"] = 5 * Bf / (time.perf_counter() - tf)
            extras["underconstrained_systems"] = int((fa_cnt > 0).sum().item())
        # (3) N>1: whole batch starts and ends on rank 0; one RCCL scatter + one gather around the solve
        if world > 1 and backend == "nccl":
            from ezpz_amd.distributed import solve_batch_sharded

            try:  # an extra: a failure here must not cost the run its headline line
                full = torch.cat([x0] * world, dim=0) if rank == 0 else None
                solve_batch_sharded(system, full, n, device=dev)
                torch.cuda.synchronize(dev)
                dist.barrier()
                te = time.perf_counter()
                reps = max(1, min(args.steps, 10))
                for _ in range(reps):
                    solve_batch_sharded(system, full, n, device=dev)
                torch.cuda.synchronize(dev)
                dist.barrier()
                extras["with_rccl_scatter_gather_solves_per_s"] = world * B * reps / (time.perf_counter() - te)
            except Exception as exc:  # noqa: BLE001
                extras["with_rccl_scatter_gather_error"] = repr(exc)[:200]

    st = np.concatenate([p["status"].cpu().numpy().view(E.STATUS_DTYPE).reshape(-1) for p in parts])
    iters = np.unique(st["iterations"]).tolist()
    ok = bool(np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0))
    if expect_iters is not None:
        ok = ok and iters == [expect_iters]
    checked = None
    if args.check and rank == 0:
        from oracle import oracle as O  # the checker (test infrastructure), outside every timed region

        # coordinates the constraints determine must match at 1e-6 (BASELINE north_star); coordinates the system's own
        # FreedomAnalysis reports as underconstrained are held only by lambda ~ 1e-9..1e-12 and are compared at the
        # oracle's sensitivity there
        err, err_free, it_equal, n_checked, bitwise = 0.0, 0.0, True, 0, True
        for p in parts:
            Bp = p["B"]
            sample = np.arange(0, Bp, max(1, Bp // 16))[:16]
            rc, xo, it, conv, nun = O.solve_batch(p["records"], p["x0_host"][sample], linsolve=O.LINSOLVE_SPARSE)
            xg = p["x_out"][torch.from_numpy(sample).to(dev)].cpu().numpy()
            stp = p["status"].cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
            rel = np.abs(xg - xo) / np.maximum(1.0, np.abs(xo))
            free = p["system"].freedom_batch(xg)[0].astype(bool)[:, : xg.shape[1]]
            err = max(err, float(np.max(np.where(free, 0.0, rel))))
            err_free = max(err_free, float(np.max(np.where(free, rel, 0.0))))
            it_equal = it_equal and bool(np.array_equal(stp["iterations"][sample], it))
            bitwise = bitwise and bool(np.array_equal(xg, xo))
            n_checked += len(sample)
        checked = {"systems": int(n_checked), "max_rel_err": err, "max_rel_err_underconstrained": err_free,
                   "iterations_equal": it_equal, "bitwise_equal": bitwise}
        ok = ok and err <= 1e-6 and err_free <= 1e-3 and checked["iterations_equal"]

    if rank == 0:
        launch_bytes = 0
        for p in parts:
            stp = p["status"].cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
            launch_bytes += algorithmic_bytes(p["info"], int(round(float(np.mean(stp["iterations"]))))) * p["B"]
        pmc = collect_pmc(args) if (args.pmc and world == 1) else {}
        line = {
            "metric": METRIC,
            "value": world * B * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "iters_to_converge": iters if len(iters) != 1 else iters[0],
            "results_ok": ok,
            "world_size_seen": world_seen,
            "value_host_to_host": value_h2h,
            "config": {
                "workload": desc,
                "value_is": "device-resident guesses -> device-resident results (inputs in HBM when the timed region "
                            "starts); value_host_to_host = the same batch through ezpz_system_solve_batch from / to "
                            "host buffers registered once with ezpz_host_register (H2D, kernels and D2H pipelined over three "
                            "streams); extras.host_to_host_pageable_solves_per_s = unregistered (pageable) buffers",
                "systems_per_launch_per_gpu": B,
                "rows": info["n_rows"], "vars": info["n_vars"], "constraints": info["n_constraints"],
                "nnz_j": info["nnz_j"], "nnz_a": info["nnz_a"], "nnz_l": info["nnz_l"], "levels": info["n_levels"],
                "team_size": info["team_size"], "team_mode": info["team_mode"], "workspace_in_lds": bool(info["workspace_in_lds"]),
                "class_specialised_kernel": [bool(p["specialized"]) for p in parts] if len(parts) > 1 else bool(p0["specialized"]),
                "parallelism": f"batch-sharded x{world}, no collective on the data path",
                "inputs": "resident in HBM; guesses = file guesses + keyed U(-%.2f,%.2f)" % (jitter, jitter),
            },
            "roofline": roofline(parts, B, kernel_ms, launch_bytes, pmc, len(parts)),
        }
        if pmc:
            line["pmc_per_launch"] = {k: v for k, v in pmc.items() if not k.startswith("_")}
        if checked:
            line["oracle_check"] = checked
        if extras:
            line["extras"] = extras
        if args.cpu_seconds > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(records, guesses, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
