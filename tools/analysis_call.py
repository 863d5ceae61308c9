"""Diagnostic: one ezpz_solve / ezpz_solve_analysis call (the reference's protocol, lib.rs:80-87 / :134-146) of a connected sketch, warm
(the request's plan exists) and cold (a new topology: symbolic phase, upload, first launch), against the CPU port."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ezpz_amd as E, gen
from oracle import oracle as O

for npts in [int(a) for a in sys.argv[1:]] or [150, 1000]:
    for drop in (0, 2):
        recs, g = gen.connected_sketch(npts, 4242)
        if drop: recs = recs[:-drop]
        guesses = list(enumerate(g.tolist()))
        cfg = E.Config(max_iterations=60)
        out = {}
        for analysis in (False, True):
            # cold: a topology the process has not seen (one constraint's parameter nudged: another request, the same structure...
            # the plan is keyed by the request's bytes, so a nudged copy is a new one)
            cold = []
            for k in range(3):
                r2 = recs.copy(); r2["param"][5] += 1e-9 * (k + 1) * (2 if analysis else 1)
                t = time.perf_counter(); E.solve_records(r2, guesses, cfg, analysis=analysis); cold.append(time.perf_counter() - t)
            E.solve_records(recs, guesses, cfg, analysis=analysis)
            t = time.perf_counter()
            for _ in range(20): got = E.solve_records(recs, guesses, cfg, analysis=analysis)
            warm = (time.perf_counter() - t) / 20
            t = time.perf_counter(); want = O.solve(recs, guesses, O.Config(max_iterations=60), linsolve=O.LINSOLVE_SPARSE, analysis=analysis); cpu = time.perf_counter() - t
            out[analysis] = (warm, min(cold), cpu, got.iterations)
        print(f"{len(g)} variables, last {drop} constraints dropped ({out[False][3]} iterations): solve warm {out[False][0]*1e3:.3f} ms, cold {out[False][1]*1e3:.2f} ms (CPU port {out[False][2]*1e3:.2f} ms) | "
              f"solve_analysis warm {out[True][0]*1e3:.3f} ms, cold {out[True][1]*1e3:.2f} ms (CPU port {out[True][2]*1e3:.1f} ms)")
