"""Diagnostic: one launch of the lanes-across-the-batch kernel on a connected sketch (for rocprofv3 --pmc / --kernel-trace):
python tools/lanes_traffic.py <points> <batch> <launches>"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
import gen
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
recs, g = gen.connected_sketch(npts, 1000 + npts)
s = E.System(recs, len(g), team_size=E.TEAM_BATCH_LANES)
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(np.tile(g, (B, 1))).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
cfg = E.Config(max_iterations=60)
s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps): s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print(f"npts={npts} n={len(g)} batch={B}: {dt*1e3:.2f} ms per launch, {B/dt:.0f} solves/s")
