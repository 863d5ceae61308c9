"""Host-to-host rate of ezpz_system_solve_batch on registered buffers (results back on host), 2000 x 2000 by default.
usage: [EZPZ_H2H_PIECE_MB=n] python tools/h2h_rate.py [lines] [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if os.environ.get('TORCH') == '1':  # (the same call inside a process that has initialised torch's HIP runtime, like bench.py)
    import torch
    torch.zeros(1, device='cuda:0')
import ezpz_amd as E
import gen
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cs = E.textual.Problem.from_str(E.textual.gen_big_problem(lines)).to_constraint_system()
n = cs.num_vars
s = E.System(cs.records, n)
s.specialize(wait=True)
hx = np.ascontiguousarray(cs.guesses[None, :] + gen.keyed_uniform(1, B, n, -0.25, 0.25))
hxo = np.empty_like(hx)
import ctypes as C
st = np.zeros(B, dtype=E.STATUS_DTYPE)
cfg = E.Config()._c()
def call():
    rc = E.lib().ezpz_system_solve_batch(s._h, hx.ctypes.data, B, C.byref(cfg), hxo.ctypes.data, st.ctypes.data, None, None, 0)
    assert rc == 0, rc
def rate(reps):
    call(); call()
    t = time.perf_counter()
    for _ in range(reps):
        call()
    return B * reps / (time.perf_counter() - t), st
r0, _ = rate(3)
E.host_register(hx); E.host_register(hxo); E.host_register(st)
r1, st = rate(5)
print(f"{n} variables x {B} systems: pageable {r0/1e6:.2f} M solves/s, registered {r1/1e6:.2f} M solves/s = {r1*n*8/1e9:.1f} GB/s each way; iterations {int(st['iterations'].min())}..{int(st['iterations'].max())}, converged {bool(st['converged'].all())}")
