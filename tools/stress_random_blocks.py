"""Stress: random block systems of the nine LINEAR kinds (3 ... 8 variables, 2 ... 7 constraints per block, 130 ... 900 blocks) through the
run-time compiled kernels with DEVICE-RESIDENT batches larger than the launch's workgroups -- so that the workgroups draw their
systems (tickets) -- against the component interpreter of a fresh system, bit for bit in every value and status, whole batch; plus
one system against the oracle.  The host-entry tests of tests/test_gpu_components.py stay below the launch's workgroups.
usage (GPU box): python tools/stress_random_blocks.py [trials] [seed]      (EZPZ_JIT_FAST_MINWAVES=5 / 6: compilations under register pressure)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import oracle as O


def replicate(constraints, guesses, copies, seed=0, jitter=0.0):
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


def run(sysobj, xin, B, n, cfg):
    xd = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
    std = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
    sysobj.solve_batch_device(xin.data_ptr(), B, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream, cfg)
    torch.cuda.synchronize()
    return xd.cpu().numpy(), std.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)


def run_trials(trials, seed):
    rng = np.random.default_rng(seed)
    linear = [O.FIXED, O.SCALAR_EQUAL, O.VERTICAL, O.HORIZONTAL, O.VERTICAL_DISTANCE, O.HORIZONTAL_DISTANCE, O.CIRCLE_RADIUS, O.POINTS_COINCIDENT, O.MIDPOINT]
    done = fast = redone = bad = 0
    for trial in range(trials):
        nv = int(rng.integers(3, 9))
        cons = [gen.arb_constraint(rng, int(rng.choice(linear)), hi=nv) for _ in range(int(rng.integers(2, 8)))]
        base = rng.uniform(-6.0, 6.0, nv)
        copies = int(rng.choice([130, 200, 333, 500, 900]))
        recs, g = replicate(cons, base, copies, seed=trial, jitter=0.05)
        n = len(g)
        B = int(rng.choice([3000, 5000, 9001]))
        x0 = g[None, :] + gen.keyed_uniform(700 + trial, B, n, -0.5, 0.5)
        x0[B // 2] = g
        fresh = E.System(recs, n)
        if fresh.info()["team_mode"] != 3:
            continue
        xin = torch.from_numpy(x0).cuda()
        cfg = E.Config()
        xw, stw = run(fresh, xin, B, n, cfg)  # the component interpreter
        sysobj = E.System(recs, n)
        if sysobj.specialize(wait=True) != 2:
            print(f"trial {trial}: no specialised kernel", flush=True)
            continue
        src = E.specialized_source(recs, n)
        is_fast = "ezpz_jit_solve_fast" in src
        ok = True
        # the loop kernel alone (a call in place: the kernels that do not wait need the guesses kept) -- what the redo list must reproduce
        xi = xin.clone()
        sti = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
        sysobj.solve_batch_device(xi.data_ptr(), B, xi.data_ptr(), sti.data_ptr(), 0, torch.cuda.current_stream().cuda_stream, cfg)
        torch.cuda.synchronize()
        xl, stl = xi.cpu().numpy(), sti.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
        # systems the residual test ends: the same bits from every kernel.  Contradictory systems end on the step test at the
        # least-squares minimum after as many iterations as the noise in |d| and in the strict comparison of the sums takes -- the
        # interpreter (another number of wavefronts: another summation tree) may count differently there, the oracle's count is a third
        settled = (stw["converged"] == 1) & (stw["final_residual_inf"] <= 1e-8)
        for rep in range(3):
            x, st = run(sysobj, xin, B, n, cfg)
            same_loop = np.array_equal(x, xl, equal_nan=True) and all(np.array_equal(st[f], stl[f], equal_nan=True) for f in st.dtype.names)
            same_int = np.array_equal(x[settled], xw[settled], equal_nan=True) and all(np.array_equal(st[f][settled], stw[f][settled], equal_nan=True) for f in st.dtype.names)
            if not (same_loop and same_int):
                ok = False
                nbad = int((~np.all((x == xl) | (np.isnan(x) & np.isnan(xl)), axis=1)).sum())
                nbad2 = int((~np.all((x[settled] == xw[settled]) | (np.isnan(x[settled]) & np.isnan(xw[settled])), axis=1)).sum())
                print(f"trial {trial} rep {rep}: MISMATCH: {nbad} of {B} systems differ from the loop kernel alone, {nbad2} of {int(settled.sum())} settled ones from the interpreter; statuses untouched {int((st['iterations'] == 0).sum())}", flush=True)
        want = O.solve(recs, x0[0], linsolve=O.LINSOLVE_SPARSE)
        if want.error != 0 or not np.array_equal(np.isnan(xw[0]), np.isnan(want.final_values)):
            ok = False
            print(f"trial {trial}: the oracle disagrees on system 0", flush=True)
        done += 1
        fast += is_fast
        redone += int(np.any(stw["iterations"] != 2))
        bad += not ok
        print(f"trial {trial}: {nv} variables x {copies} blocks, {B} systems per launch, kernel that does not wait: {is_fast}, iterations {sorted(set(int(i) for i in st['iterations']))}, ended by the residual test {int(settled.sum())}: {'ok' if ok else 'FAILED'}", flush=True)
    print(f"# {done} block systems, {fast} on the kernels that do not wait for verdicts, {redone} with systems on the redo list, {bad} failed")
    return done, fast, redone, bad


def run_trials_all_kinds(trials, seed):
    """Blocks of ALL 25 kinds (weights 1 / 2.5): the loop kernel of the run-time compiled code in one launch whose workgroups draw
    their systems against the same kernel in launches of 256 systems (fewer than it has workgroups: nobody draws), bit for bit."""
    rng = np.random.default_rng(seed)
    done = bad = 0
    kinds = set()
    for trial in range(trials):
        nv = int(rng.integers(3, 9))
        cons = []
        for _ in range(int(rng.integers(1, 6))):
            c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nv)
            c["weight"] = float(rng.choice([1.0, 1.0, 2.5]))
            cons.append(c)
        base = rng.uniform(-6.0, 6.0, nv)
        copies = int(rng.choice([130, 200, 333]))
        recs, g = replicate(cons, base, copies, seed=trial, jitter=0.05)
        n = len(g)
        B = int(rng.choice([3000, 5000]))
        x0 = g[None, :] + gen.keyed_uniform(900 + trial, B, n, -0.3, 0.3)
        sysobj = E.System(recs, n)
        if sysobj.info()["team_mode"] != 3 or sysobj.specialize(wait=True) != 2:
            continue
        kinds.update(int(c["kind"]) for c in cons)
        xin = torch.from_numpy(x0).cuda()
        cfg = E.Config(max_iterations=12)
        x, st = run(sysobj, xin, B, n, cfg)
        xs = np.empty_like(x)
        sts = np.empty_like(st)
        for lo in range(0, B, 256):
            hi = min(B, lo + 256)
            xs[lo:hi], sts[lo:hi] = run(sysobj, xin[lo:hi], hi - lo, n, cfg)
        ok = True
        for rep in range(2):
            if rep:
                x, st = run(sysobj, xin, B, n, cfg)
            same = np.array_equal(x, xs, equal_nan=True) and all(np.array_equal(st[f], sts[f], equal_nan=True) for f in st.dtype.names)
            if not same:
                ok = False
                rows = ~np.all((x == xs) | (np.isnan(x) & np.isnan(xs)), axis=1)
                print(f"trial {trial} rep {rep}: MISMATCH in {int(rows.sum())} of {B} systems; statuses untouched {int((st['iterations'] == 0).sum())} / {int((sts['iterations'] == 0).sum())}; iterations differing {int((st['iterations'] != sts['iterations']).sum())}", flush=True)
        done += 1
        bad += not ok
        print(f"trial {trial}: kinds {sorted(int(c['kind']) for c in cons)}, {nv} variables x {copies} blocks, {B} systems per launch, iterations {sorted(set(int(i) for i in st['iterations']))[:8]}: {'ok' if ok else 'FAILED'}", flush=True)
    print(f"# all kinds: {done} block systems ({len(kinds)} kinds seen), {bad} failed")
    return done, len(kinds), bad


def run_trials_grid(trials, seed):
    """Linear blocks, so many that a system takes SEVERAL workgroups (the ladder's kernels: solve_kernel_grid_fast, both compilations --
    launches of up to six rounds of the systems in flight and longer ones, jit.cpp: comp_jit_launch -- and the redo list through
    solve_kernel_grid): device-resident launches of 12 and of 40 rounds' worth against the loop kernel alone (calls in place), bit for bit."""
    rng = np.random.default_rng(seed)
    linear = [O.FIXED, O.SCALAR_EQUAL, O.VERTICAL, O.HORIZONTAL, O.VERTICAL_DISTANCE, O.HORIZONTAL_DISTANCE, O.CIRCLE_RADIUS, O.POINTS_COINCIDENT, O.MIDPOINT]
    done = bad = redone = 0
    for trial in range(trials):
        nv = int(rng.integers(3, 7))
        cons = [gen.arb_constraint(rng, int(rng.choice(linear)), hi=nv) for _ in range(int(rng.integers(2, 6)))]
        base = rng.uniform(-6.0, 6.0, nv)
        copies = int(rng.choice([9000, 14000, 20000]))
        recs, g = replicate(cons, base, copies, seed=trial, jitter=0.05)
        n = len(g)
        sysobj = E.System(recs, n)
        info = sysobj.info()
        if sysobj.specialize(wait=True) != 2:
            print(f"trial {trial}: no specialised kernel ({info['team_mode']})", flush=True)
            continue
        G = sysobj.info()["grid_workgroups"]
        if G <= 1:
            print(f"trial {trial}: one workgroup per system", flush=True)
            continue
        slots = max(1, 1024 // G)
        ok = True
        for B in (2 * slots + 1, 8 * slots + 3):
            x0 = g[None, :] + gen.keyed_uniform(1200 + trial, B, n, -0.5, 0.5)
            x0[B // 2] = g
            xin = torch.from_numpy(x0).cuda()
            cfg = E.Config()
            xi = xin.clone()
            sti = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
            sysobj.solve_batch_device(xi.data_ptr(), B, xi.data_ptr(), sti.data_ptr(), 0, torch.cuda.current_stream().cuda_stream, cfg)
            torch.cuda.synchronize()
            xl, stl = xi.cpu().numpy(), sti.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
            for rep in range(2):
                x, st = run(sysobj, xin, B, n, cfg)
                same = np.array_equal(x, xl, equal_nan=True) and all(np.array_equal(st[f], stl[f], equal_nan=True) for f in st.dtype.names)
                if not same:
                    ok = False
                    rows = ~np.all((x == xl) | (np.isnan(x) & np.isnan(xl)), axis=1)
                    print(f"trial {trial} B {B} rep {rep}: MISMATCH in {int(rows.sum())} of {B} systems; statuses untouched {int((st['iterations'] == 0).sum())}", flush=True)
            redone += int(np.any(stl["iterations"] != 2))
        done += 1
        bad += not ok
        print(f"trial {trial}: {nv} variables x {copies} blocks on {G} workgroups, launches of {2 * slots + 1} and {8 * slots + 3} systems, iterations {sorted(set(int(i) for i in stl['iterations']))[:8]}: {'ok' if ok else 'FAILED'}", flush=True)
    print(f"# several workgroups per system: {done} block systems, {redone} launches with systems on the redo list, {bad} failed")
    return done, redone, bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grid":
        sys.exit(1 if run_trials_grid(int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 1357)[2] else 0)
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        sys.exit(1 if run_trials_all_kinds(int(sys.argv[2]) if len(sys.argv) > 2 else 24, int(sys.argv[3]) if len(sys.argv) > 3 else 8642)[2] else 0)
    sys.exit(1 if run_trials(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 97531)[3] else 0)
