"""Diagnostic: what a RESIDENT one-call kernel (one workgroup that polls a word of device memory for the caller's next request, DESIGN.md
section 4) costs the kernels of ANOTHER stream on the same device.  The main thread times batch launches of the 2000 x 2000 system
(65 536 systems per launch, its own stream); a second thread keeps one solve() of a small system going every `period` microseconds, so
that its kernel is resident all the time.  usage (GPU box): python tools/resident_cost.py"""
import sys, threading, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
from oracle import textual as T

dev = torch.device('cuda', 0)
cs = T.load(T.gen_big_problem(500)); n = cs.num_vars
big = E.System(cs.constraints, n)
big.specialize(wait=True)
B = 65536
x0 = torch.from_numpy(np.tile(cs.guesses, (B, 1))).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(dev)

def rate(reps=40):
    with torch.cuda.stream(stream):
        for _ in range(5): big.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): big.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream.cuda_stream)
        torch.cuda.synchronize()
    return B * reps / (time.perf_counter() - t)

sq = T.load(open('tests/golden/test_cases/square/problem.md').read())
stop = False
calls = [0]
def caller(period_us):
    while not stop:
        E.solve_records(sq.constraints, sq.variables())
        calls[0] += 1
        if period_us >= 100:
            time.sleep(period_us * 1e-6)  # (releases the interpreter lock: the main thread's launches are not held up)
        else:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < period_us: pass

import os
print(f"# EZPZ_RESIDENT_US={os.environ.get('EZPZ_RESIDENT_US', '(default 200)')}")
base = [rate() for _ in range(3)]
print(f"batches alone: {np.mean(base) / 1e6:.2f} M solves/s ({min(base) / 1e6:.2f} .. {max(base) / 1e6:.2f})")
# any co-tenant: ONE workgroup of another stream that stays on the device (torch.cuda._sleep: one block spinning) -- no host thread involved
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    torch.cuda._sleep(int(2.0e9 * 0.6))
time.sleep(0.01)
r = [rate(20) for _ in range(3)]
torch.cuda.synchronize()
print(f"beside ONE spinning workgroup of another stream (torch.cuda._sleep): {np.mean(r) / 1e6:.2f} M solves/s ({min(r) / 1e6:.2f} .. {max(r) / 1e6:.2f}) = "
      f"{np.mean(r) / np.mean(base):.3f} of the rate alone")
for period in (20, 100, 1000):
    stop = False; calls[0] = 0
    th = threading.Thread(target=caller, args=(period,)); th.start()
    time.sleep(0.2)
    t0 = time.perf_counter(); r = [rate() for _ in range(3)]; dt = time.perf_counter() - t0
    stop = True; th.join()
    print(f"beside a thread calling solve() of `square` every ~{period} us (its kernel resident between calls; {calls[0] / max(dt + 0.2, 1e-9):.0f} calls/s): "
          f"{np.mean(r) / 1e6:.2f} M solves/s ({min(r) / 1e6:.2f} .. {max(r) / 1e6:.2f}) = {np.mean(r) / np.mean(base):.3f} of the rate alone")
