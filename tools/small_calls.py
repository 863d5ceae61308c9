"""Diagnostic: SMALL calls of a system created for batches (team_size 0) of one connected sketch (tests/gen.py:connected_sketch) --
device time per call and solves/s with the guesses resident in HBM, on the frontal plan the system takes for such calls
(EzpzSystemInfo.front_max_batch) against the same system created with EZPZ_FRONTS=0 (the record walk at every call size, the shape
of every call before round 5).  The crossover is what EzpzLaunchPolicy.front_small_call_fill encodes."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
import gen

dev = torch.device('cuda', 0)
sizes = [int(a) for a in sys.argv[1:]] or [75, 150, 400, 1000, 2500]
batches = [int(b) for b in os.environ.get("BATCHES", "1,4,16,32,64,128,256,512").split(",")]
cfg = E.Config(max_iterations=60)


def rate(s, g, B):
    x0 = torch.from_numpy(np.tile(g, (B, 1))).to(dev)
    xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    reps = max(3, min(30, 4096 // B))
    for _ in range(2): s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps


for npts in sizes:
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    auto = E.System(recs, len(g))
    os.environ["EZPZ_FRONTS"] = "0"
    records = E.System(recs, len(g))
    del os.environ["EZPZ_FRONTS"]
    info = auto.info()
    print(f"n={len(g):5d}: large calls team_mode {info['team_mode']} team {info['team_size']}; fronts on {info['front_workgroups']} workgroups "
          f"for calls of <= {info['front_max_batch']} systems")
    for B in batches:
        ta, tr = rate(auto, g, B), rate(records, g, B)
        took = "fronts " if B <= info['front_max_batch'] else "records"
        print(f"    call of {B:4d}: automatic ({took}) {ta*1e6:9.1f} us = {B/ta:10.0f} solves/s | record walk only {tr*1e6:9.1f} us = {B/tr:10.0f} solves/s | x{tr/ta:5.2f}")
