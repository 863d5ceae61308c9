#!/usr/bin/env python3
"""Benchmark of the LM constraint-solve hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  For N > 1 it is either
launched by torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE in the environment) or started
plainly, in which case it starts the N rank processes itself -- before anything in this process touches the GPU -- and
exits non-zero when the node has fewer than N devices.  It never reports fewer GPUs than it was asked for.

  metric   solves/sec on the 2000-row massive_parallel_system (BASELINE.json), plus iters-to-converge
  step     one launch of the LM kernel over a batch of `--batch` independent replicas of the workload system
           (jittered initial guesses), inputs already resident in HBM
  warm-up  W launches, then launches in windows of 20 until two consecutive windows agree within 2 % (or 0.3 s have
           passed): the first ~30 launches of a process run slower (clocks, first touch), and the timed region must not
           depend on how many the caller asked for.  `warmup_launches_run` says how many ran in all.
  value    whole-job solves/s = N * batch * K / max-over-ranks wall time of the K timed steps, device-resident
           guesses in, device-resident results out (`config.value_is`).  `value_host_to_host` is the same batch
           through the host-pointer entry point (SURVEY.md 8d's "results back on host": H2D + kernel + D2H).
  scaling  weak: every rank owns its own shard of `--batch` systems; the path has no data-path collective
           (systems are independent), so none is issued inside the timed region
  configs  after the headline, short legs (a few launches each, outside `value`) over the other BASELINE.json
           configurations -- `square` x 65 536, the 1 M mixed batch, the 200 000-variable ladder, and a batch of one
           connected 300-variable sketch -- each with its rate, iterations, an oracle check and its roofs
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
import math
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

from ezpz_amd.synthetic import keyed_uniform, make_workload  # noqa: E402  (re-exported: tests build workloads through bench)

METRIC = "solves/sec on 2000-row massive_parallel_system @1/2/4/8 GPU; iters-to-converge"
# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # HBM3E peak BW 8.0 TB/s (spec)
STREAMING_COPY_GBS = 5400.0  # measured, not a datasheet figure: tools/row_copy_bench.hip (full-line accesses, 768-2048 workgroups)
PMC_SETS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
     "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
]
SOLVE_KERNELS = ("lm_solve_kernel", "comp_solve_kernel", "ezpz_jit_solve", "ezpz_jit_lane", "batch_lane_kernel", "front_solve_kernel")
# The other BASELINE.json configurations, as (workload, systems per launch at N = 1): configs[2], [4], [3], and the
# connected-sketch batches of DESIGN.md section 3 (one lane per system at 262 144 systems per launch; the per-system teams'
# record walk at 32 768).  Short legs after the headline; the mixed batch is sharded over the ranks.
# ... and ONE solve of a 2000-variable connected sketch on the frontal shape (team 0xFFFFFFFF = the automatic latency shape:
# a tree of dense fronts on several workgroups): a launch is one solve, its roofs are over the CUs it uses.
LEGS = [("square", 65536, 0), ("mixed", 1 << 20, 0), ("massive50000", 64, 0), ("sketch150", 262144, 0), ("sketch150", 32768, 0),
        ("sketch1000", 1, 0xFFFFFFFF)]


def algorithmic_bytes(info: dict, k: int) -> int:
    """BYTES = (1+k)(56C + 8n + 8m + 8zJ) + k(12zJ + 8m + 16zA + 24zL + 48n)."""
    C, n, m = info["n_constraints"], info["n_vars"], info["n_rows"]
    zj, za, zl = info["nnz_j"], info["nnz_a"], info["nnz_l"]
    return (1 + k) * (56 * C + 8 * n + 8 * m + 8 * zj) + k * (12 * zj + 8 * m + 16 * za + 24 * zl + 48 * n)


def cpu_baseline(records, guesses, budget_s: float, max_iterations: int = 0):
    """Oracle (`kind: port`) timed on one core with the CLI protocol (ezpz-cli/src/main.rs:86-100)."""
    from oracle import oracle as O

    ocfg = O.Config(max_iterations=max_iterations) if max_iterations else None
    secs, iters = O.time_solves(records, guesses, repeats=20, config=ocfg, linsolve=O.LINSOLVE_SPARSE)
    per = max(secs / 20.0, 1e-7)
    repeats = int(min(max(budget_s / per, 100 if per < 1e-3 else 5), 200000))
    secs, iters = O.time_solves(records, guesses, repeats=repeats, config=ocfg, linsolve=O.LINSOLVE_SPARSE)
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
    O.solve_batch(records, x0, ocfg, linsolve=O.LINSOLVE_SPARSE, nthreads=cores)
    out["all_cores"] = {"value": nb / (time.perf_counter() - t0), "unit": "solves/s", "cores": cores,
                        "sample": f"{nb} replicas, OpenMP over systems"}
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536,
                    help="systems per launch per GPU (the gap between launches and the partly empty last round of the 768 resident "
                         "workgroups are 2000 x 2000's 4096: 81.9, 16 384: 88.4, 32 768: 92.0, 65 536: 93.5, 131 072: 92.8 M solves/s)")
    ap.add_argument("--workload", default="massive500", help="massive<lines>[o] (o = over-constrained variant), sketch<points> (one connected sketch), a test_cases/ directory name, or mixed")
    ap.add_argument("--team", type=lambda v: int(v, 0), default=0, help="override lanes per system (0 = auto; 0xFFFFFFFF = the automatic shape for one solve)")
    ap.add_argument("--max-iterations", type=int, default=0, help="Config.max_iterations (0 = the reference's default, 35)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--check", type=int, default=1, help="verify the results of the last step against the oracle")
    ap.add_argument("--extras", type=int, default=1, help="also report the host-to-host rate, single-solve latency (and, "
                    "for N>1, the rate with the RCCL scatter/gather of the batch) -- never part of `value`")
    ap.add_argument("--specialize", type=int, default=1, help="use the run-time compiled class-specialised kernel where the topology has one")
    ap.add_argument("--pmc", type=int, default=1, help="N=1: collect the roofline's PMC counters with rocprofv3 child passes of this script")
    ap.add_argument("--legs", type=int, default=1, help="after the headline, short legs over the other BASELINE.json configurations (`configs` in the line)")
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


def collect_pmc(args, workload=None, batch=None, sets=None, team=None) -> dict:
    """Per-launch means of the PMC counters of the solve kernel, from rocprofv3 child passes of this script with the
    same workload (one counter set per pass; the profiler's child is this interpreter running bench.py itself)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    base = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else tempfile.gettempdir()
    outdir = tempfile.mkdtemp(prefix="bench_pmc_", dir=base)
    env = dict(os.environ, TMPDIR="/tmp")
    # the interpreter that runs this script (a plain ELF placed directly after `--`: no env / shim hop under the profiler)
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__), "--workload", workload or args.workload,
             "--batch", str(batch or args.batch), "--team", str(team if team is not None else args.team), "--max-iterations", str(args.max_iterations), "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
             "--check", "0", "--extras", "0", "--pmc", "0", "--legs", "0", "--specialize", str(args.specialize)]
    counters, errors = {}, []
    for i, cset in enumerate(sets or PMC_SETS):
        d = os.path.join(outdir, f"set{i}")
        try:
            subprocess.run([exe, "--pmc"] + cset + ["--output-format", "csv", "-d", d, "--"] + child, env=env, cwd="/tmp",
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300, check=True)
        except (subprocess.SubprocessError, OSError) as exc:
            errors.append(f"set{i}: {type(exc).__name__}: {str(getattr(exc, 'stderr', b'') or exc)[-200:]}")
            continue
        # a step may be several kernels of DIFFERENT names (a linear system on several workgroups: ezpz_jit_solve_fast, then the
        # loop kernel for what it lists -- usually nothing): a counter's value per step is the sum over the names of its mean per
        # dispatch of that name (kernels of one name launched several times per step -- the mixed batch -- are roofline()'s n_kernels)
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if any(k in row["Kernel_Name"] for k in SOLVE_KERNELS):
                    acc.setdefault(row["Counter_Name"], {}).setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
                    if "batch_lane_kernel" in row["Kernel_Name"]:
                        counters["_state_in_hbm"] = 1.0
        for k, by_name in acc.items():
            counters[k] = sum(sum(v) / len(v) for v in by_name.values())
            counters.setdefault("_dispatches", max(len(v) for v in by_name.values()))
            if len(by_name) > 1:
                counters["_kernels"] = sorted(n.split("(")[0][:48] for n in by_name)
    shutil.rmtree(outdir, ignore_errors=True)
    if errors:
        counters["errors"] = errors
        print("bench.py: PMC pass failed: " + "; ".join(errors), file=sys.stderr)
    return counters


def roofline(info_parts, B, kernel_ms, solves_per_launch_iters, pmc, n_kernels, n_cus):
    """The three roofs (module docstring).  `pmc` = per-launch counter means or {} / {"error": ...}; `n_cus` = compute
    units of the device the kernel ran on (4 SIMDs each)."""
    t = kernel_ms * 1e-3
    n_simds = 4 * n_cus
    if n_kernels > 1 and pmc:
        # a step of several kernels (the mixed batch: one per topology): collect_pmc returns the mean over DISPATCHES, every
        # other figure here is per STEP -- round 4 compared one kernel's traffic with the whole step's compulsory bytes
        # (traffic_over_compulsory 0.69); the counters of a step are the per-dispatch means times the kernels of a step
        pmc = {k: (v * n_kernels if isinstance(v, float) and not k.startswith("_") else v) for k, v in pmc.items()}
    compulsory = sum((16 * p["info"]["n_vars"] + 32) * p["B"] for p in info_parts)  # bytes per launch
    hbm = {"achieved": compulsory / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "compulsory_bytes_per_solve": compulsory / B}
    hbm["frac"] = hbm["achieved"] / hbm["peak"]
    roofs = {"hbm": hbm}
    traffic = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports both in KiB; gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        hbm["traffic_over_compulsory"] = traffic / compulsory
        hbm["measured_gbs"] = traffic / t / 1e9
        # what a kernel that only copies 16 KB rows in and out with whole-line accesses reaches on this device at the solve kernels'
        # occupancy (tools/row_copy_bench.hip, profiles/r06_row_copy_bench.txt: 5.2-5.45 TB/s; 4.9-5.0 with 8-byte strided accesses)
        hbm["streaming_copy_gbs"] = STREAMING_COPY_GBS
        hbm["frac_of_streaming_copy"] = hbm["measured_gbs"] / STREAMING_COPY_GBS
    if pmc.get("_state_in_hbm"):
        # lanes across the batch (batch_kernel.hip.hpp) keep the solver state in global memory by design.  SURVEY 8(d)'s BYTES
        # formula is the MODEL of that traffic (every access priced as if it reached HBM); the roof itself is what the counters
        # saw -- much of the state is served from L2 / MALL -- whenever they were collected
        hbm.update({"model_gbs": solves_per_launch_iters / t / 1e9, "model_frac": solves_per_launch_iters / t / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_solve": solves_per_launch_iters / B,
                    "state": "in HBM (one lane per system); model = SURVEY 8(d) BYTES per solve, roof = measured traffic when counters were collected"})
        if traffic is not None:
            hbm["achieved"] = traffic / t / 1e9
        else:
            hbm["achieved"] = hbm["model_gbs"]
        hbm["frac"] = hbm["achieved"] / hbm["peak"]
    cycles = pmc.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
    if cycles > 0 and "SQ_LDS_IDX_ACTIVE" in pmc:
        roofs["lds"] = {"achieved": pmc["SQ_LDS_IDX_ACTIVE"], "peak": n_cus * cycles, "unit": "LDS-array cycles per launch",
                        "frac": pmc["SQ_LDS_IDX_ACTIVE"] / (n_cus * cycles),
                        "lds_insts_per_solve": pmc.get("SQ_INSTS_LDS", 0.0) / B,
                        "bank_conflict_frac": pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(pmc["SQ_LDS_IDX_ACTIVE"], 1.0)}
    if cycles > 0 and "SQ_ACTIVE_INST_VALU" in pmc:
        roofs["issue"] = {"achieved": 4.0 * pmc["SQ_ACTIVE_INST_VALU"], "peak": n_simds * cycles,
                          "unit": "vector-ALU busy cycles per launch", "frac": 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (n_simds * cycles),
                          "kernel_cycles": cycles}
        if "SQ_INSTS_VALU" in pmc:
            roofs["issue"].update({"valu_insts_per_solve": pmc["SQ_INSTS_VALU"] / B, "salu_insts_per_solve": pmc.get("SQ_INSTS_SALU", 0.0) / B,
                                   "wave_cycles_waiting_frac": pmc.get("SQ_WAIT_ANY", 0.0) / max(pmc.get("SQ_WAVE_CYCLES", 0.0), 1.0)})
    # the bound is the roof the kernel is nearest to -- for memory, nearest to what memory GIVES: the traffic the counters saw against
    # the rate of a pure copy of the rows on this device (frac_of_streaming_copy) where that was measured, else bytes against the
    # datasheet's 8 TB/s.  (Until round 6 every roof was priced against its datasheet peak, and a kernel at 0.9 of the attainable
    # memory rate with 0.65 of the vector issue slots busy read "issue".)  The top-level fields stay the contract's: achieved and
    # peak of the bounding roof -- for hbm the compulsory bytes per second against 8 TB/s.
    nearness = {k: max(v["frac"], v.get("frac_of_streaming_copy", 0.0)) if k == "hbm" else v["frac"] for k, v in roofs.items()}
    bound = max(nearness, key=nearness.get)
    top = roofs[bound]
    algo = solves_per_launch_iters / t / 1e9
    out = {
        "bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
        "traffic": traffic,
        "nearness_to_attainable": nearness,
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
    # every roof is meant to be a bound; one above 1 (a partitioned device whose counters cover another CU count, a
    # traffic model served from cache) is reported, not asserted: the line must come out either way
    over = sorted(k for k, r in roofs.items() if r["frac"] > 1.0 + 1e-9)
    if over:
        out["roof_violation"] = over
    return out


class Workload:
    """One workload resident on this rank's device: systems (one per topology), jittered guesses in HBM, outputs."""

    def __init__(self, E, torch, name, batch, device_index, dev, team, specialize, seed_rank, max_iterations=0):
        # BASELINE configs[4] flavour: system i uses [circle_tangent, parallelogram, arc_radius][i mod 3]; the batch is
        # grouped by topology (one launch per topology per step).  Everything else is a single-topology batch.
        names = ["circle_tangent", "parallelogram", "arc_radius"] if name == "mixed" else [name]
        self.E, self.torch, self.dev, self.batch = E, torch, dev, batch
        self.config = E.Config(max_iterations=max_iterations) if max_iterations else None
        self.stream = torch.cuda.current_stream(dev)
        self.parts = []
        for k, nm in enumerate(names):
            desc, records, guesses, jitter, expect = make_workload(nm)
            n = len(guesses)
            B = batch // len(names) + (1 if k < batch % len(names) else 0)
            system = E.System(records, n, device=device_index, team_size=team)
            # component-resident systems: the class-specialised kernel (run-time compiled once per topology, outside the
            # timed region like the symbolic phase; batch calls would start it themselves in the background)
            spec = system.specialize(wait=True) == 2 if specialize else False
            # synthetic inputs: replicas of the system with keyed-PRNG jitter on the guesses, resident in HBM
            x0_host = guesses[None, :] + keyed_uniform(0x657A707A + seed_rank + 101 * k, B, n, -jitter, jitter)
            x0_host[0] = guesses
            x0 = torch.from_numpy(x0_host).to(dev)
            self.parts.append(dict(desc=desc, records=records, guesses=guesses, jitter=jitter, expect=expect, n=n, B=B,
                                   system=system, info=system.info(), specialized=spec, x0_host=x0_host, x0=x0,
                                   x_out=torch.empty_like(x0), status=torch.zeros((B, 32), dtype=torch.uint8, device=dev)))
        self.desc = self.parts[0]["desc"] if len(self.parts) == 1 else "mixed: " + " + ".join(p["desc"] for p in self.parts)
        # the mixed batch is ONE ragged batch, system i of topology i mod 3, through the heterogeneous entry
        # (ezpz_mixed_solve_device: regrouped by topology inside, one launch per topology on parallel streams, results in
        # caller order); the per-topology tensors above are filled from its results for the checks (sync_parts)
        self.mixed = None
        if len(self.parts) > 1:
            import numpy as np

            topo = (np.arange(batch) % len(self.parts)).astype(np.uint32)
            self.mixed = E.MixedBatch([p["system"] for p in self.parts], topo)
            ragged = np.empty(self.mixed.total)
            for k, p in enumerate(self.parts):
                idx = np.nonzero(topo == k)[0]
                pos = self.mixed.offsets[idx][:, None].astype(np.int64) + np.arange(p["n"])[None, :]
                ragged[pos] = p["x0_host"]
                p["sys_idx"] = torch.from_numpy(idx).to(dev)
                p["pos"] = torch.from_numpy(pos).to(dev)
            self.x0_ragged_host = ragged
            self.x0_ragged = torch.from_numpy(ragged).to(dev)
            self.x_ragged = torch.empty_like(self.x0_ragged)
            self.status_all = torch.zeros((batch, 32), dtype=torch.uint8, device=dev)

    def step(self):
        if self.mixed is not None:
            self.mixed.solve_device(self.x0_ragged.data_ptr(), self.x_ragged.data_ptr(), self.status_all.data_ptr(), self.stream.cuda_stream)
            return
        for p in self.parts:
            p["system"].solve_batch_device(p["x0"].data_ptr(), p["B"], p["x_out"].data_ptr(), p["status"].data_ptr(), 0,
                                           self.stream.cuda_stream, self.config)

    def sync_parts(self):
        """The mixed batch's results, per topology (the checks below read the per-topology tensors)."""
        if self.mixed is None:
            return
        self.torch.cuda.synchronize(self.dev)
        for p in self.parts:
            p["x_out"] = self.x_ragged[p["pos"]]
            p["status"] = self.status_all[p["sys_idx"]]

    def host_to_host_rate(self, reps=4):
        """SURVEY 8(d)'s solve -- results back on host -- for this workload: the host-pointer entry point on buffers registered
        once (ezpz_host_register): H2D, kernels and D2H pipelined.  Returns (solves/s, results equal to the device path's)."""
        import ctypes as C

        import numpy as np

        E = self.E
        cfg = (self.config or E.Config())._c()
        self.sync_parts()
        if self.mixed is not None:
            hx, hxo = np.ascontiguousarray(self.x0_ragged_host), np.empty_like(self.x0_ragged_host)
            hst = np.zeros(self.batch, dtype=E.STATUS_DTYPE)
            call = lambda: E.lib().ezpz_mixed_solve(self.mixed._h, hx.ctypes.data, C.byref(cfg), hxo.ctypes.data, hst.ctypes.data)
            want = lambda: self.x_ragged.cpu().numpy()
            n_sys = self.batch
        else:
            p = self.parts[0]
            n_sys = min(p["B"], 262144)
            hx, hxo = np.ascontiguousarray(p["x0_host"][:n_sys]), np.empty_like(p["x0_host"][:n_sys])
            hst = np.zeros(n_sys, dtype=E.STATUS_DTYPE)
            call = lambda: E.lib().ezpz_system_solve_batch(p["system"]._h, hx.ctypes.data, n_sys, C.byref(cfg), hxo.ctypes.data,
                                                           hst.ctypes.data, None, None, 0)
            want = lambda: p["x_out"][:n_sys].cpu().numpy()
        for a in (hx, hxo, hst):
            E.host_register(a)
        try:
            assert call() == 0 and call() == 0
            t0 = time.perf_counter()
            for _ in range(reps):
                assert call() == 0
            rate = reps * n_sys / (time.perf_counter() - t0)
            # (pieces of a pipelined call may run on another kernel of the topology than the device-filling launch -- the teams
            # below the lanes' threshold: then equal to rounding, not bitwise)
            ref = want()
            with np.errstate(invalid="ignore"):
                rel = float(np.nanmax(np.abs(hxo - ref) / np.maximum(1.0, np.abs(ref)))) if ref.size else 0.0
            return rate, (True if np.array_equal(hxo, ref, equal_nan=True) else f"to rounding: max relative difference {rel:.2e}")
        finally:
            for a in (hx, hxo, hst):
                E.host_unregister(a)

    def _window(self, launches):
        torch = self.torch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(self.stream)
        for _ in range(launches):
            self.step()
        ev1.record(self.stream)
        torch.cuda.synchronize(self.dev)
        return ev0.elapsed_time(ev1) / launches

    def warm_to_steady_state(self, at_least):
        """`at_least` launches, then windows of 20 launches (fewer when a launch takes milliseconds) until two consecutive
        windows agree within 2 % or 0.3 s have passed.  Returns the launches run."""
        for _ in range(at_least):
            self.step()
        self.torch.cuda.synchronize(self.dev)
        ran, t0 = at_least, time.perf_counter()
        first = self._window(1)
        ran += 1
        window = int(min(20, max(2, math.ceil(20.0 / max(first, 1e-3)))))  # ~20 ms of launches, 20 at most
        prev = None
        while True:
            ms = self._window(window)
            ran += window
            if prev is not None and abs(ms - prev) <= 0.02 * prev:
                break
            if time.perf_counter() - t0 >= 0.3:
                break
            prev = ms
        return ran

    def statuses(self):
        import numpy as np

        self.sync_parts()
        return np.concatenate([p["status"].cpu().numpy().view(self.E.STATUS_DTYPE).reshape(-1) for p in self.parts])

    def oracle_check(self, per_part=16):
        """Coordinates the constraints determine must match at 1e-6 (BASELINE north_star); coordinates the system's own
        FreedomAnalysis reports as underconstrained are held only by lambda ~ 1e-9..1e-12 and are compared at the
        oracle's sensitivity there.  The oracle is the checker, outside every timed region."""
        import numpy as np
        from oracle import oracle as O

        torch = self.torch
        self.sync_parts()
        err, err_free, it_equal, n_checked, bitwise = 0.0, 0.0, True, 0, True
        for p in self.parts:
            Bp = p["B"]
            sample = np.arange(0, Bp, max(1, Bp // per_part))[:per_part]
            ocfg = O.Config(max_iterations=self.config.max_iterations) if self.config else None
            rc, xo, it, conv, nun = O.solve_batch(p["records"], p["x0_host"][sample], ocfg, linsolve=O.LINSOLVE_SPARSE)
            xg = p["x_out"][torch.from_numpy(sample).to(self.dev)].cpu().numpy()
            stp = p["status"].cpu().numpy().view(self.E.STATUS_DTYPE).reshape(-1)
            rel = np.abs(xg - xo) / np.maximum(1.0, np.abs(xo))
            free = p["system"].freedom_batch(xg)[0].astype(bool)[:, : xg.shape[1]]
            err = max(err, float(np.max(np.where(free, 0.0, rel))))
            err_free = max(err_free, float(np.max(np.where(free, rel, 0.0))))
            it_equal = it_equal and bool(np.array_equal(stp["iterations"][sample], it))
            bitwise = bitwise and bool(np.array_equal(xg, xo))
            n_checked += len(sample)
        return {"systems": int(n_checked), "max_rel_err": err, "max_rel_err_underconstrained": err_free,
                "iterations_equal": it_equal, "bitwise_equal": bitwise}

    def algorithmic_launch_bytes(self):
        import numpy as np

        self.sync_parts()
        total = 0
        for p in self.parts:
            stp = p["status"].cpu().numpy().view(self.E.STATUS_DTYPE).reshape(-1)
            total += algorithmic_bytes(p["info"], int(round(float(np.mean(stp["iterations"]))))) * p["B"]
        return total


def run_leg(E, torch, dist, args, name, batch, world, rank, device_index, dev, backend, n_cus, team=0):
    """One short leg over another BASELINE configuration: steady-state warm-up, a few timed launches bracketed like the
    headline's (barrier + synchronize, max over ranks), iterations, oracle check, roofs (hbm always; issue from one PMC
    pass at N = 1)."""
    import numpy as np

    # (N > 1: a rank that cannot set the leg up must not leave the others waiting at the leg's barriers -- every rank says
    # whether it is ready, and the leg runs only if all are)
    w, ran, failure = None, 0, None
    try:
        w = Workload(E, torch, name, batch, device_index, dev, team or args.team, args.specialize, rank, args.max_iterations)
        ran = w.warm_to_steady_state(2)
    except Exception as exc:  # noqa: BLE001
        failure = repr(exc)[:300]
    if world > 1:
        ready = torch.tensor([0.0 if failure else 1.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if float(ready[0]) < 1.0 and not failure:
            failure = "another rank could not set this leg up"
    if failure:
        return {"workload": name, "error": failure}
    one = w._window(1)
    steps = int(min(200, max(5, math.ceil(50.0 / max(one, 1e-3)))))  # ~50 ms of launches
    if world > 1:  # the same number of launches on every rank
        st_ = torch.tensor([float(steps)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(st_, op=dist.ReduceOp.MAX)
        steps = int(st_[0])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(w.stream)
    for _ in range(steps):
        w.step()
    ev1.record(w.stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    st = w.statuses()
    iters = np.unique(st["iterations"]).tolist()
    ok = bool(np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0))
    out = {"workload": w.desc, "systems_per_launch_per_gpu": batch, "value": world * batch * steps / elapsed, "unit": "solves/s",
           "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup_launches_run": ran,
           "iters": iters if len(iters) <= 8 else {"min": int(min(iters)), "max": int(max(iters)), "mean": float(np.mean(st["iterations"]))},
           "class_specialised_kernel": [bool(p["specialized"]) for p in w.parts] if len(w.parts) > 1 else bool(w.parts[0]["specialized"])}
    if rank == 0:
        if args.check:
            out["oracle_check"] = w.oracle_check(per_part=-(-64 // len(w.parts)))  # >= 64 systems of every leg
            ok = ok and out["oracle_check"]["max_rel_err"] <= 1e-6 and out["oracle_check"]["iterations_equal"]
        if args.extras:
            try:
                out["value_host_to_host"], out["host_to_host_results_equal_device_path"] = w.host_to_host_rate()
            except Exception as exc:  # noqa: BLE001
                out["value_host_to_host_error"] = repr(exc)[:200]
        # (the kernel that keeps its state in HBM -- lanes across the batch, sketch150 x 262 144 -- gets its traffic measured too:
        # its bound is read from the counters, not from the model)
        sets = [PMC_SETS[1]] + (PMC_SETS[2:] if name.startswith("sketch") and batch >= 65536 else [])
        pmc = collect_pmc(args, workload=name, batch=batch, sets=sets, team=team or None) if (args.pmc and world == 1) else {}
        r = roofline(w.parts, batch, kernel_ms, w.algorithmic_launch_bytes(), pmc, len(w.parts), n_cus)
        out["roofline"] = {"bound": r["bound"], "frac": r["frac"], "kernel_ms": kernel_ms,
                           "roofs": {k: v["frac"] for k, v in r["roofs"].items()}}
        info0 = w.parts[0]["info"]
        if info0.get("team_mode") == 5:
            # the frontal shape: a launch of `batch` systems runs on batch x grid_workgroups workgroups, one per CU -- a single solve
            # uses a few CUs of the chip, and the issue / LDS roofs that mean something are those of the CUs it runs on
            active = min(n_cus, batch * int(info0["grid_workgroups"]))
            out["roofline"]["active_cus"] = active
            out["roofline"]["workgroups_per_system"] = int(info0["grid_workgroups"])
            out["roofline"]["roofs_of_active_cus"] = {k: r["roofs"][k]["frac"] * n_cus / active for k in ("issue", "lds") if k in r["roofs"]}
            out["roofline"]["fronts"] = int(info0["n_partitions"])
        if "wave_cycles_waiting_frac" in r["roofs"].get("issue", {}):
            out["roofline"]["wave_cycles_waiting_frac"] = r["roofs"]["issue"]["wave_cycles_waiting_frac"]
        if r["roofs"]["hbm"].get("state"):
            out["roofline"]["hbm_model"] = {k: r["roofs"]["hbm"][k] for k in ("state", "model_gbs", "model_frac", "measured_gbs") if k in r["roofs"]["hbm"]}
        if "roof_violation" in r:
            out["roofline"]["roof_violation"] = r["roof_violation"]
    out["results_ok"] = ok
    return out


def rank0_extras(E, torch, np, args, extras, w, dev, stream):
    """Rank 0's extras (never part of `value`): the host-pointer entry point, single-solve latencies, FreedomAnalysis.
    Fills `extras`, returns value_host_to_host."""
    parts, B = w.parts, args.batch
    p0 = parts[0]
    records, guesses, n, system = p0["records"], p0["guesses"], p0["n"], p0["system"]
    x0_host, x0, x_out = p0["x0_host"], p0["x0"], p0["x_out"]
    value_h2h = None
    import ctypes as C

    # pageable buffers first, then the same buffers page-locked once (ezpz_host_register: what a caller that
    # reuses its buffers does); `value_host_to_host` is the registered rate
    hb = min(B, 65536)
    hx = np.ascontiguousarray(x0_host[:hb])
    hxo = np.empty_like(hx)
    hst = np.zeros(hb, dtype=E.STATUS_DTYPE)
    hcfg = E.Config()._c()

    def host_call():
        rc = E.lib().ezpz_system_solve_batch(system._h, hx.ctypes.data, hb, C.byref(hcfg), hxo.ctypes.data,
                                             hst.ctypes.data, None, None, 0)
        assert rc == 0, rc

    host_call()
    host_call()
    th = time.perf_counter()
    for _ in range(3):
        host_call()
    extras["host_to_host_pageable_solves_per_s"] = 3 * hb / (time.perf_counter() - th)
    value_h2h, equal = w.host_to_host_rate(reps=8)
    extras["host_to_host_batch"] = min(B, 262144)
    extras["host_to_host_results_equal_device_path"] = equal
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
    # (a RATE: launches on one stream overlap their own overheads -- not the latency of a call, which is the next figure)
    extras["back_to_back_launch_period_us"] = (time.perf_counter() - tl) / 200 * 1e6
    # (2a) ONE full ezpz_solve() call per iteration from host buffers -- the protocol of the reference's published figure
    # and of its criterion benchmarks (main.rs:86-100, solver_bench.rs:15-24): lint + Model::new + LM + unsatisfied check.
    # warm = the request served from the plan cache, cold = ezpz_cache_clear before every call (symbolic phase included).
    # (the C ABI itself on buffers prepared once: what a host-language caller does)
    from ezpz_amd._lib import COutcome

    rec_arr = E.stack_records(records)
    ids = np.arange(n, dtype=np.uint32)
    gv = np.ascontiguousarray(guesses, dtype=np.float64)
    out_x, out_un, out_o = np.zeros(n), np.zeros(len(rec_arr) + 1, dtype=np.uint64), COutcome()
    solve_args = [rec_arr.ctypes.data, len(rec_arr), ids.ctypes.data, gv.ctypes.data, n, C.byref(hcfg), out_x.ctypes.data,
                  out_un.ctypes.data, None, 0, C.byref(out_o)]
    one_call = lambda: E.lib().ezpz_solve(*solve_args)
    for _ in range(300):  # (past the 256 solves after which a topology's specialised kernel takes over)
        assert one_call() == 0
    time.sleep(0.5)
    tw = time.perf_counter()
    for _ in range(200):
        one_call()
    extras["full_solve_call_us_warm"] = (time.perf_counter() - tw) / 200 * 1e6
    extras["full_solve_call_iterations"] = int(out_o.iterations)
    tc = 0.0
    for _ in range(5):
        E.lib().ezpz_cache_clear()
        t_ = time.perf_counter()
        one_call()
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
    return value_h2h


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

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Test hooks (1-GPU boxes; the driver never sets them).  EZPZ_BENCH_BACKEND=gloo runs the N>1 code path --
    # rendezvous, barriers, max-over-ranks timing, rank-0 line -- with every rank on GPU (LOCAL_RANK mod device count).
    # EZPZ_BENCH_FORCE_DIST=1 takes every N>1 branch at world size 1: the RCCL process group on the device, barriers,
    # the all-reduces on device tensors, the scatter / gather extra.
    backend = os.environ.get("EZPZ_BENCH_BACKEND", "nccl")
    distributed = world > 1 or os.environ.get("EZPZ_BENCH_FORCE_DIST") == "1"
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but only {torch.cuda.device_count()} HIP device(s) visible")
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    n_cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (only when something other than this script or torch.distributed.run set RANK / WORLD_SIZE without a port: the ranks
        # cannot agree on a free one without it; both launchers above choose the port themselves)
        os.environ.setdefault("MASTER_PORT", "29577")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    w = Workload(E, torch, args.workload, args.batch, device_index, dev, args.team, args.specialize, rank, args.max_iterations)
    parts, stream, B = w.parts, w.stream, args.batch
    p0 = parts[0]
    desc, records, guesses, jitter, expect_iters = w.desc, p0["records"], p0["guesses"], p0["jitter"], p0["expect"]
    n, system, info = p0["n"], p0["system"], p0["info"]
    x0_host, x0, x_out = p0["x0_host"], p0["x0"], p0["x_out"]

    warmup_ran = w.warm_to_steady_state(args.warmup)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        w.step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)  # one kernel per step, back to back on this stream
    world_seen = 1
    if distributed:
        t = torch.tensor([elapsed, kernel_ms, 1.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t[:2], op=dist.ReduceOp.MAX)
        dist.all_reduce(t[2:], op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, world_seen = float(t[0]), float(t[1]), int(round(float(t[2])))

    extras = {}
    value_h2h = None
    if args.extras:
        # (1) host-pointer entry point: H2D + kernel + D2H per call -- SURVEY 8(d)'s "results back on host"
        # (rank 0 only, and never fatal: at N > 1 the other ranks are waiting for it in the collective extra below)
        try:
            if rank == 0:
                value_h2h = rank0_extras(E, torch, np, args, extras, w, dev, stream)
        except Exception as exc:  # noqa: BLE001
            extras["rank0_extras_error"] = repr(exc)[:300]
        # (2c) N>1: the whole job's batch from rank 0's HOST buffers over every device's own host link (ezpz_multi_solve_batch:
        # one worker thread and one analysed topology per device, contiguous shards, no collective) -- the only path past
        # one link.  The other ranks wait at the barrier.
        if distributed and len(parts) == 1:
            try:
                if rank == 0:
                    mask = (1 << world) - 1 if backend == "nccl" else 1
                    multi = E.MultiSystem(records, n, device_mask=mask)
                    if args.specialize:
                        multi.specialize(wait=True)
                    mb = world * min(B, 16384)  # (16 KB per system: 0.27 GB each way per device)
                    mx = np.ascontiguousarray(np.tile(x0_host[: min(B, 16384)], (world, 1)))
                    mxo, mst = np.empty_like(mx), np.zeros(mb, dtype=E.STATUS_DTYPE)
                    for a_ in (mx, mxo, mst):
                        E.host_register(a_)
                    try:
                        multi.solve_batch(mx, out=(mxo, mst))
                        tm = time.perf_counter()
                        for _ in range(4):
                            multi.solve_batch(mx, out=(mxo, mst))
                        extras["multi_host_to_host_solves_per_s"] = 4 * mb / (time.perf_counter() - tm)
                        extras["multi_host_to_host_devices"] = multi.devices()
                    finally:
                        for a_ in (mx, mxo, mst):
                            E.host_unregister(a_)
                    del multi
            except Exception as exc:  # noqa: BLE001
                extras["multi_host_to_host_error"] = repr(exc)[:200]
            dist.barrier()
        # (3) N>1: whole batch starts and ends on rank 0; one RCCL scatter + one gather around the solve
        if distributed and backend == "nccl" and len(parts) == 1:
            from ezpz_amd.distributed import solve_batch_sharded

            try:  # an extra: a failure here must not cost the run its headline line
                full = torch.cat([x0] * world, dim=0) if rank == 0 else None
                xs, sts = solve_batch_sharded(system, full, n, device=dev)
                torch.cuda.synchronize(dev)
                dist.barrier()
                if rank == 0:  # the sharded results are the device path's, bit for bit (replicas of this rank's batch)
                    extras["rccl_scatter_gather_equals_device_path"] = bool(torch.equal(xs[:B], x_out) and torch.equal(xs[-B:], x_out))
                te = time.perf_counter()
                reps = max(1, min(args.steps, 10))
                for _ in range(reps):
                    solve_batch_sharded(system, full, n, device=dev)
                torch.cuda.synchronize(dev)
                dist.barrier()
                extras["with_rccl_scatter_gather_solves_per_s"] = world * B * reps / (time.perf_counter() - te)
            except Exception as exc:  # noqa: BLE001
                extras["with_rccl_scatter_gather_error"] = repr(exc)[:200]

    st = w.statuses()
    iters = np.unique(st["iterations"]).tolist()
    ok = bool(np.all(st["converged"] == 1) and np.all(st["n_unsatisfied"] == 0))
    if expect_iters is not None:
        ok = ok and iters == [expect_iters]
    checked = None
    if args.check and rank == 0:
        checked = w.oracle_check()
        ok = ok and checked["max_rel_err"] <= 1e-6 and checked["max_rel_err_underconstrained"] <= 1e-4 and checked["iterations_equal"]

    # the other BASELINE configurations (outside `value`): every rank takes part in a leg's barriers; at N > 1 only the
    # mixed batch -- BASELINE configs[4], "batch-sharded across 8 x MI355X" -- runs, split over the ranks
    legs = []
    if args.legs and args.workload == "massive500":
        for name, lb, lteam in LEGS:
            if world > 1 and name != "mixed":
                continue
            try:
                # (the headline's tensors are small; a leg's go away with its Workload)
                legs.append(run_leg(E, torch, dist, args, name, max(1, lb // world), world, rank, device_index, dev, backend, n_cus, team=lteam))
            except Exception as exc:  # noqa: BLE001  (a leg must not cost the run its headline)
                legs.append({"workload": name, "error": repr(exc)[:300]})
            torch.cuda.empty_cache()

    if rank == 0:
        pmc = collect_pmc(args) if (args.pmc and world == 1) else {}
        line = {
            "metric": METRIC,
            "value": world * B * args.steps / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "warmup_launches_run": warmup_ran,
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
                            "host buffers registered once with ezpz_host_register (copies in, kernels and copies out pipelined, one "
                            "stream each); extras.host_to_host_pageable_solves_per_s = unregistered (pageable) buffers",
                "systems_per_launch_per_gpu": B,
                "rows": info["n_rows"], "vars": info["n_vars"], "constraints": info["n_constraints"],
                "nnz_j": info["nnz_j"], "nnz_a": info["nnz_a"], "nnz_l": info["nnz_l"], "levels": info["n_levels"],
                "team_size": info["team_size"], "team_mode": info["team_mode"], "workspace_in_lds": bool(info["workspace_in_lds"]),
                "class_specialised_kernel": [bool(p["specialized"]) for p in parts] if len(parts) > 1 else bool(p0["specialized"]),
                "parallelism": f"batch-sharded x{world}, no collective on the data path",
                "inputs": "resident in HBM; guesses = file guesses + keyed U(-%.2f,%.2f)" % (jitter, jitter),
                "compute_units": n_cus,
            },
            "roofline": roofline(parts, B, kernel_ms, w.algorithmic_launch_bytes(), pmc, len(parts), n_cus),
        }
        if pmc:
            line["pmc_per_launch"] = {k: v for k, v in pmc.items() if not k.startswith("_")}
        if info.get("team_mode") == 5:  # the frontal shape: the roofs of the CUs a launch runs on (see run_leg)
            active = min(n_cus, B * int(info["grid_workgroups"]))
            r_ = line["roofline"]
            r_["active_cus"], r_["workgroups_per_system"], r_["fronts"] = active, int(info["grid_workgroups"]), int(info["n_partitions"])
            r_["roofs_of_active_cus"] = {k: r_["roofs"][k]["frac"] * n_cus / active for k in ("issue", "lds") if k in r_["roofs"]}
        if checked:
            line["oracle_check"] = checked
        if legs:
            line["configs"] = legs
        if extras:
            line["extras"] = extras
        if args.cpu_seconds > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(records, guesses, args.cpu_seconds, args.max_iterations)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        if not ok:
            # a rate measured on wrong results is not a measurement: the line says so and the run fails
            line["invalid"] = "the solves of the timed launches failed validation (statuses of the whole last batch, iteration count, oracle check): see results_ok, iters_to_converge, oracle_check"
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
