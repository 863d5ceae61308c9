"""Stress: does any result depend on LDS a kernel reads before it writes?  AMD devices do not clear the LDS between kernels: a
workgroup finds what the last one on its CU left.  tools/liblds_poison.so (tools/lds_poison.hip) fills every CU's LDS with a pattern;
each workload below -- one per kernel family -- runs after zeros, after quiet-NaN bit patterns and after 1.0s, and every output of the
three runs must be the same bits.
usage (GPU box): hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/liblds_poison.so tools/lds_poison.hip && python tools/stress_lds_poison.py"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import oracle as O
from oracle import textual as T

P = ctypes.CDLL(os.path.join(HERE, "liblds_poison.so"))
P.lds_poison.argtypes = [ctypes.c_ulonglong]
P.lds_poison.restype = ctypes.c_int
P.lds_peek.argtypes = [ctypes.c_ulonglong]
P.lds_peek.restype = ctypes.c_double
PATTERNS = [("zeros", 0), ("quiet NaNs", 0x7FF8DEADBEEF0001), ("1.0", 0x3FF0000000000000), ("huge", 0x7FE0000000000001), ("all ones", 0xFFFFFFFFFFFFFFFF)]


def outputs(fn):
    res = []
    for name, pat in PATTERNS:
        torch.cuda.synchronize()
        rc = P.lds_poison(pat)
        assert rc > 0, rc
        res.append(fn())
    return res


def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None or b is None:
        return a is b
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.names:
        return all(np.array_equal(a[f], b[f], equal_nan=True) for f in a.dtype.names)
    return np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


def check(name, fn):
    res = outputs(fn)
    bad = [PATTERNS[i][0] for i in range(1, len(res)) if not same(res[0], res[i])]
    print(f"{name}: {'the same bits after every pattern' if not bad else 'DIFFERS after ' + ', '.join(bad)}", flush=True)
    return not bad


def main():
    rng = np.random.default_rng(0)
    ok = True
    # the control: what a kernel finds in LDS it has not written, after each pattern
    for name, pat in PATTERNS:
        assert P.lds_poison(pat) > 0
        print(f"control: after {name}, a kernel that only reads finds the pattern in {100.0 * P.lds_peek(pat):.1f} % of the LDS words it looks at", flush=True)
    # small non-linear fixture: list walk, then the lane-per-system compiled kernel, one solve() calls (resident wave kernel)
    with open(os.path.join(HERE, "..", "tests", "golden", "test_cases", "square", "problem.md")) as f:
        sq = E.textual.Problem.from_str(f.read()).to_constraint_system()
    x0 = sq.guesses[None, :] + rng.uniform(-0.1, 0.1, size=(4096, sq.num_vars))
    s = E.System(sq.records, sq.num_vars, team_size=E.TEAM_AUTO_LISTS)
    ok &= check("square x 4096, list walk", lambda: s.solve_batch(x0, want_mask=True))
    s2 = E.System(sq.records, sq.num_vars)
    assert s2.specialize(wait=True) == 2
    ok &= check("square x 4096, a lane per system (compiled)", lambda: s2.solve_batch(x0, want_mask=True))
    ok &= check("square, one solve() call", lambda: (lambda r: (r.final_values, r.iterations, r.unsatisfied))(E.solve_records(sq.records, sq.variables())))
    # block system: interpreter, compiled loop (in place), the kernel that does not wait (out of place), non-linear variant
    for lines, over in ((500, False), (500, True), (97, False)):
        ref = T.load(T.gen_big_problem(lines, over))
        n = ref.num_vars
        B = 3000
        xb = ref.guesses[None, :] + gen.keyed_uniform(31, B, n, -0.25, 0.25)
        xin = torch.from_numpy(xb).cuda()

        def dev(sysobj, inplace):
            xd = xin.clone() if inplace else torch.full((B, n), float("nan"), dtype=torch.float64, device="cuda")
            std = torch.zeros((B, 32), dtype=torch.uint8, device="cuda")
            sysobj.solve_batch_device(xin.data_ptr() if not inplace else xd.data_ptr(), B, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            return xd.cpu().numpy(), std.cpu().numpy()
        fresh = E.System(ref.constraints, n)
        ok &= check(f"blocks {lines} over {over} x {B}, interpreter", lambda: dev(fresh, False))
        comp = E.System(ref.constraints, n)
        assert comp.specialize(wait=True) == 2
        ok &= check(f"blocks {lines} over {over} x {B}, compiled, out of place", lambda: dev(comp, False))
        ok &= check(f"blocks {lines} over {over} x {B}, compiled, in place (the loop kernel)", lambda: dev(comp, True))
    # the ladder on several workgroups: list walk grid team, compiled kernels
    lad = T.load(T.gen_big_problem(12000))
    n = lad.num_vars
    xl = lad.guesses[None, :] + gen.keyed_uniform(32, 40, n, -0.25, 0.25)
    g1 = E.System(lad.constraints, n)
    ok &= check("12 000 lines x 40, list walk on several workgroups", lambda: g1.solve_batch(xl))
    g2 = E.System(lad.constraints, n)
    assert g2.specialize(wait=True) == 2
    ok &= check("12 000 lines x 40, compiled on several workgroups", lambda: g2.solve_batch(xl))
    # connected sketches: every shape
    recs, g = gen.connected_sketch(150, 1150)
    xs = g[None, :] + rng.uniform(-0.01, 0.01, (96, len(g)))
    cfg = E.Config(max_iterations=40)
    for shape, nm in ((E.TEAM_AUTO_LATENCY, "fronts"), (E.TEAM_LATENCY_RECORDS, "record walk (latency)"), (0, "record walk (batch)"), (E.TEAM_LATENCY_PHASES, "dense phases"),
                      (E.TEAM_BATCH_LANES, "lanes across the batch"), (E.TEAM_AUTO_LISTS, "list walk")):
        sk = E.System(recs, len(g), team_size=shape)
        ok &= check(f"sketch of 150 points x 96, {nm} (team_mode {sk.info()['team_mode']})", lambda: sk.solve_batch(xs, cfg, want_mask=True))
        ok &= check(f"sketch of 150 points, one system, {nm}", lambda: sk.solve_batch(xs[:1], cfg, want_mask=True))
    big, gb = gen.connected_sketch(1000, 2000)
    fb = E.System(big, len(gb), team_size=E.TEAM_AUTO_LATENCY)
    xbig = gb[None, :] + rng.uniform(-0.01, 0.01, (3, len(gb)))
    ok &= check(f"sketch of 1000 points x 3, fronts on {fb.info()['front_workgroups']} workgroups", lambda: fb.solve_batch(xbig, cfg))
    # FreedomAnalysis: probes, the QR in LDS, the large QR's routes
    loose = E.System(recs[:-2], len(g), team_size=E.TEAM_AUTO_LATENCY)
    xf, _, _ = loose.solve_batch(xs[:3], cfg)
    ok &= check("FreedomAnalysis of the sketch less two constraints, probes", lambda: loose.freedom_batch(xf))
    os.environ["EZPZ_FREEDOM_PROBES"] = "0"
    ok &= check("... the pivoted QR", lambda: loose.freedom_batch(xf))
    lb = E.System(big[:-2], len(gb), team_size=E.TEAM_AUTO_LATENCY)
    xfb, _, _ = lb.solve_batch(xbig, cfg)
    for chain in ("", "2", "1"):
        os.environ["EZPZ_FREEDOM_CHAIN"] = chain
        ok &= check(f"FreedomAnalysis of 2000 variables x 3, the pivoted QR, EZPZ_FREEDOM_CHAIN={chain!r}", lambda: lb.freedom_batch(xfb))
        ok &= check(f"... one system", lambda: lb.freedom_batch(xfb[:1]))
    os.environ.pop("EZPZ_FREEDOM_CHAIN")
    os.environ.pop("EZPZ_FREEDOM_PROBES")
    small = E.System(sq.records[:-1], sq.num_vars)
    xq, _, _ = small.solve_batch(x0[:64])
    ok &= check("FreedomAnalysis of square less one constraint x 64 (a lane per system)", lambda: small.freedom_batch(xq))
    print("# every workload the same bits after every pattern" if ok else "# SOME WORKLOAD DEPENDS ON WHAT THE LDS HELD")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
