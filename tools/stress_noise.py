"""Stress: kernels whose workgroups exchange values inside a launch (self-validating chunks, rings, rendezvous) are deterministic by
design -- so a result that changes when a co-running kernel shifts their timing is a race (round 6 found one in the large QR that way
round: three systems side by side were the noise).  Each workload runs alone, then four times beside tools/libnoise.so's kernel
(64 and 160 single-wavefront workgroups hammering 256 MB on a stream of their own); every output must keep its bits.  Launches are
kept small enough for every workgroup of a system to be resident beside the noise.
usage (GPU box): hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libnoise.so tools/noise.hip && python tools/stress_noise.py"""
import ctypes, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import textual as T
from stress_lds_poison import same

N = ctypes.CDLL(os.path.join(HERE, "libnoise.so"))
N.noise_start.argtypes = [ctypes.c_int]


def check(name, fn, reps=4):
    torch.cuda.synchronize()
    ref = fn()
    bad = []
    for wgs in (64, 160):
        assert N.noise_start(wgs) == 0
        try:
            for r in range(reps):
                if not same(ref, fn()):
                    bad.append(f"{wgs} noise workgroups, run {r}")
        finally:
            assert N.noise_stop() == 0
    print(f"{name}: {'the same bits beside the noise' if not bad else 'CHANGED: ' + '; '.join(bad)}", flush=True)
    return not bad


def main():
    rng = np.random.default_rng(1)
    ok = True
    cfg = E.Config(max_iterations=40)
    lad = T.load(T.gen_big_problem(50000))
    n = lad.num_vars
    xl = lad.guesses[None, :] + gen.keyed_uniform(41, 5, n, -0.25, 0.25)
    xl[2] = 0.0
    xl[2, 0::4] = xl[2, 2::4] = np.arange(50000)
    xl[2, 3::4] = 4.0  # (a system at the solution: the redo list)
    g1 = E.System(lad.constraints, n)
    ok &= check("ladder x 5, list walk on a grid team", lambda: g1.solve_batch(xl))
    g2 = E.System(lad.constraints, n)
    assert g2.specialize(wait=True) == 2
    ok &= check("ladder x 5, the kernels that do not wait + redo list (98 workgroups per system)", lambda: g2.solve_batch(xl))
    xin = torch.from_numpy(xl).cuda()

    def inplace():
        xd = xin.clone()
        std = torch.zeros((5, 32), dtype=torch.uint8, device="cuda")
        g2.solve_batch_device(xd.data_ptr(), 5, xd.data_ptr(), std.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return xd.cpu().numpy(), std.cpu().numpy()
    ok &= check("ladder x 5, the loop kernel on 98 workgroups per system (in place)", inplace)
    mid = T.load(T.gen_big_problem(12000))
    xm = mid.guesses[None, :] + gen.keyed_uniform(42, 12, mid.num_vars, -0.25, 0.25)
    g3 = E.System(mid.constraints, mid.num_vars)
    assert g3.specialize(wait=True) == 2
    ok &= check(f"12 000 lines x 12, compiled on {g3.info()['grid_workgroups']} workgroups per system", lambda: g3.solve_batch(xm))
    for npts, B in ((1000, 3), (2500, 2), (5000, 1)):
        recs, g = gen.connected_sketch(npts, 1000 + npts)
        fs = E.System(recs, len(g), team_size=E.TEAM_AUTO_LATENCY)
        xs = g[None, :] + rng.uniform(-0.01, 0.01, (B, len(g)))
        ok &= check(f"sketch of {npts} points x {B}, fronts on {fs.info()['front_workgroups']} workgroups", lambda: fs.solve_batch(xs, cfg, want_mask=True))
        if npts <= 1000:
            loose = E.System(recs[:-2], len(g), team_size=E.TEAM_AUTO_LATENCY)
            xf, _, _ = loose.solve_batch(xs, cfg)
            ok &= check(f"... FreedomAnalysis less two constraints, probes", lambda: loose.freedom_batch(xf))
            os.environ["EZPZ_FREEDOM_PROBES"] = "0"
            for chain in ("", "2", "1"):
                os.environ["EZPZ_FREEDOM_CHAIN"] = chain
                ok &= check(f"... the pivoted QR x {B}, EZPZ_FREEDOM_CHAIN={chain!r}", lambda: loose.freedom_batch(xf), reps=2)
                ok &= check(f"... the pivoted QR x 1, EZPZ_FREEDOM_CHAIN={chain!r}", lambda: loose.freedom_batch(xf[:1]), reps=2)
            os.environ.pop("EZPZ_FREEDOM_CHAIN")
            os.environ.pop("EZPZ_FREEDOM_PROBES")
    print("# every workload kept its bits beside the noise" if ok else "# SOME RESULT CHANGED BESIDE THE NOISE")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
