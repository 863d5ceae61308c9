"""Time the LM kernel on the massive system as a function of max_iterations (0 = eval + final sweep only)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import textual as T

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
team = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cs = T.load(T.gen_big_problem(lines))
n = cs.num_vars
s = E.System(cs.constraints, n, team_size=team)
print(s.info())
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(cs.guesses[None, :] + gen.keyed_uniform(1, B, n, -0.25, 0.25)).to(dev)
xo = torch.empty_like(x0)
st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev)
for it in (0, 1, 2, 35):
    cfg = E.Config(max_iterations=it)
    for _ in range(3):
        s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream.cuda_stream, cfg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream.cuda_stream, cfg)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    per_cu = B / 256.0
    print(f"max_it={it}: {ms:.4f} ms per launch, {ms * 1e3 / per_cu:.2f} us per system per CU")
