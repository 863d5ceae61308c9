"""One wavefront per system (jit_kernel.hip.hpp: wave_kernel, its elimination across the lanes) against the lane-per-system kernel
on random systems of all 25 kinds: values and statuses bit for bit, NaN / 1e150 / 1e-150 starts included (the permanent test
is tests/test_gpu_lanes.py::test_one_wavefront_per_system_equals_the_lane_kernel_bitwise; this is the wider sweep).
usage (GPU box): python tools/wave_fuzz.py [seed] [systems]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ezpz_amd as E, gen
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
tested = skipped = has = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    nvars = int(rng.integers(4, 21))
    cons = []
    for _ in range(int(rng.integers(2, 16))):
        c = gen.arb_constraint(rng, int(rng.integers(0, O.NUM_KINDS)), hi=nvars)
        c["weight"] = float(rng.choice([1.0, 1.0, 0.25, 3.0]))
        cons.append(c)
    recs = O.stack(cons)
    src = E.specialized_source(recs, nvars, wave=True)
    if not src:
        skipped += 1
        continue
    has += "HAS_TAIL_WAVE = true" in src
    lane = E.System(recs, nvars); wave = E.System(recs, nvars, team_size=E.TEAM_LATENCY_WAVE)
    if lane.specialize(wait=True) != 2 or wave.specialize(wait=True) != 2:
        skipped += 1
        continue
    x0 = rng.uniform(-8.0, 8.0, (32, nvars)); x0[3, 0] = np.nan; x0[5] *= 1e150; x0[6] *= 1e-150
    for cfg in (dict(), dict(max_iterations=12, initial_lambda=1e-20)):
        xl, sl, _ = lane.solve_batch(x0, E.Config(**cfg)); xw, sw, _ = wave.solve_batch(x0, E.Config(**cfg))
        same = ((xl == xw) | (np.isnan(xl) & np.isnan(xw))).all() and all(np.array_equal(sl[f], sw[f], equal_nan=True) if sl[f].dtype.kind == "f" else np.array_equal(sl[f], sw[f]) for f in sl.dtype.names)
        if not same:
            bad = np.nonzero(~((xl == xw) | (np.isnan(xl) & np.isnan(xw))).all(axis=1) | (sl["iterations"] != sw["iterations"]))[0]
            print("MISMATCH trial", trial, "nvars", nvars, "kinds", sorted(set(int(k) for k in recs["kind"])), "systems", bad.tolist()[:8], cfg)
    tested += 1
print("tested", tested, "with the elimination across lanes", has, "skipped", skipped)
