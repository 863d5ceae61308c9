import time, numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import ezpz_amd as E
from oracle import textual as T
from conftest import read_case
for name in ("square",):
    cs=T.load(read_case(name))
    s=E.System(cs.constraints, cs.num_vars)
    x0=cs.guesses[None,:]
    for _ in range(5): s.solve_batch(x0)
    t=time.perf_counter()
    for _ in range(200): s.solve_batch(x0)
    print(name,"System.solve_batch (host ptr, batch 1): %.1f us"%((time.perf_counter()-t)/200*1e6))
    t=time.perf_counter()
    for _ in range(200): s.solve_batch(x0, want_mask=True)
    print(name,"  + mask: %.1f us"%((time.perf_counter()-t)/200*1e6))
    for _ in range(5): E.solve_records(cs.constraints, cs.variables())
    t=time.perf_counter()
    for _ in range(200): E.solve_records(cs.constraints, cs.variables())
    print(name,"ezpz_solve via ctypes: %.1f us"%((time.perf_counter()-t)/200*1e6))
