import sys, time, threading; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, ezpz_amd as E
from oracle import oracle as O, textual as T
from conftest import read_case
names=["square","circle_tangent","two_rectangles","parallelogram","arc_radius","chamfer_square","perpendicular","symmetric"]
systems=[]
for nm in names:
    ref=T.load(read_case(nm))
    recs=O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
    systems.append((recs, ref.guesses))
for r,g in systems: E.solve_records(r,g)
def run(nthreads, reps=300):
    def worker(t):
        r,g=systems[t%len(systems)]
        for _ in range(reps): E.solve_records(r,g)
    th=[threading.Thread(target=worker,args=(t,)) for t in range(nthreads)]
    t0=time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return nthreads*reps/(time.perf_counter()-t0)
for nt in (1,2,4,8):
    print(nt, "threads:", round(run(nt)), "solves/s")
