"""Stress: several host threads drive the SAME systems at once, each on a stream of its own (ctypes releases the GIL: the calls overlap
inside the library) -- a block system on the kernels that do not wait for verdicts (shared redo lists and ticket counters), a system
on several workgroups (shared ring scratch, launches chained on one event), a connected sketch on the fronts and on the record walk,
FreedomAnalysis -- and every result must equal the one the same call gave alone, bit for bit.
usage (GPU box): python tools/stress_threads.py [threads] [calls per thread]"""
import os, sys, threading
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import textual as T


def main():
    n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    cfg = E.Config(max_iterations=40)
    work = []  # (name, system, x0 device tensor, B, n)
    blk = T.load(T.gen_big_problem(500))
    sb = E.System(blk.constraints, blk.num_vars)
    assert sb.specialize(wait=True) == 2
    xb = blk.guesses[None, :] + gen.keyed_uniform(51, 3000, blk.num_vars, -0.25, 0.25)
    exact = np.zeros(blk.num_vars); exact[0::4] = exact[2::4] = np.arange(500); exact[3::4] = 4.0
    xb[7::211] = exact  # (systems for the redo list)
    work.append(("2000 x 2000 blocks x 3000 (kernels that do not wait + redo list)", sb, xb))
    lad = T.load(T.gen_big_problem(12000))
    sl = E.System(lad.constraints, lad.num_vars)
    assert sl.specialize(wait=True) == 2
    xl = lad.guesses[None, :] + gen.keyed_uniform(52, 6, lad.num_vars, -0.25, 0.25)
    work.append((f"12 000 lines x 6 on {sl.info()['grid_workgroups']} workgroups per system", sl, xl))
    recs, g = gen.connected_sketch(150, 1150)
    xs = g[None, :] + np.random.default_rng(3).uniform(-0.01, 0.01, (64, len(g)))
    work.append(("sketch of 150 points x 64, record walk", E.System(recs, len(g)), xs))
    work.append(("sketch of 150 points x 2, fronts", E.System(recs, len(g), team_size=E.TEAM_AUTO_LATENCY), xs[:2]))
    with open(os.path.join(HERE, "..", "tests", "golden", "test_cases", "square", "problem.md")) as f:
        sq = E.textual.Problem.from_str(f.read()).to_constraint_system()
    ss = E.System(sq.records, sq.num_vars)
    assert ss.specialize(wait=True) == 2
    xq = sq.guesses[None, :] + np.random.default_rng(4).uniform(-0.1, 0.1, (20000, sq.num_vars))
    work.append(("square x 20 000, a lane per system", ss, xq))

    def run(sysobj, xin, stream):
        B, n = xin.shape
        with torch.cuda.stream(stream):
            xd = torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
            std = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
        stream.synchronize()
        sysobj.solve_batch_device(xin.data_ptr(), B, xd.data_ptr(), std.data_ptr(), 0, stream.cuda_stream, cfg)
        stream.synchronize()
        return xd.cpu().numpy(), std.cpu().numpy()

    main_stream = torch.cuda.Stream()
    tensors = [(name, s, torch.from_numpy(x).cuda()) for name, s, x in work]
    torch.cuda.synchronize()
    refs = [run(s, xin, main_stream) for _, s, xin in tensors]
    # ... and the host entry of the same systems (pageable buffers: the system's own staging buffers, one call at a time inside)
    hosts = [np.ascontiguousarray(x[: min(len(x), 400)]) for _, _, x in work]
    host_refs = [s.solve_batch(h, cfg) for (_, s, _), h in zip(work, hosts)]
    host_calls = [0]
    errors = []
    counts = [0] * len(tensors)

    def worker(tid):
        stream = torch.cuda.Stream()
        rng = np.random.default_rng(tid)
        try:
            for c in range(calls):
                k = int(rng.integers(0, len(tensors)))
                name, s, xin = tensors[k]
                if c % 4 == 3:
                    xh, sth, _ = s.solve_batch(hosts[k], cfg)
                    host_calls[0] += 1
                    if not (np.array_equal(xh, host_refs[k][0], equal_nan=True) and all(np.array_equal(sth[f], host_refs[k][1][f], equal_nan=True) for f in sth.dtype.names)):
                        errors.append(f"thread {tid} call {c}: {name}, host entry: differs")
                    continue
                x, st = run(s, xin, stream)
                counts[k] += 1
                if not (np.array_equal(x, refs[k][0], equal_nan=True) and np.array_equal(st, refs[k][1])):
                    rows = int((~np.all((x == refs[k][0]) | (np.isnan(x) & np.isnan(refs[k][0])), axis=1)).sum())
                    errors.append(f"thread {tid} call {c}: {name}: {rows} systems differ")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {tid}: {exc!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in threads: t.start()
    for t in threads: t.join()
    for (name, _, _), c in zip(tensors, counts):
        print(f"{name}: {c} concurrent calls")
    print(f"host entries among them: {host_calls[0]} calls")
    print("\n".join(errors[:10]))
    print(f"# {n_threads} threads x {calls} calls: " + ("every result the same bits as alone" if not errors else f"{len(errors)} FAILED"))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
