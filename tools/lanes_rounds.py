"""Diagnostic (GPU box): where the time of a lanes-across-the-batch launch goes -- the same jittered batch of one connected
sketch solved with max_iterations capped at 1, 2, 3 ...: the increments are the cost of each round of LM iterations
(fewer and fewer lanes are still working), the last line is the uncapped launch.
usage: python tools/lanes_rounds.py [points=150] [batch=262144]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ezpz_amd as E  # noqa: E402
from ezpz_amd.synthetic import keyed_uniform, make_workload  # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
desc, recs, g, jitter, _ = make_workload(f"sketch{npts}")
n = len(g)
dev = torch.device("cuda", 0)
x0 = torch.from_numpy(g[None, :] + keyed_uniform(0x657A707A, B, n, -jitter, jitter)).to(dev)
x = torch.empty_like(x0)
st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
s = E.System(recs, n, team_size=E.TEAM_BATCH_LANES)
stream = torch.cuda.current_stream(dev).cuda_stream
prev = 0.0
for cap in list(range(1, 13)) + [16, 20, 35]:
    cfg = E.Config(max_iterations=cap)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.solve_batch_device(x0.data_ptr(), B, x.data_ptr(), st.data_ptr(), 0, stream, cfg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
    sv = st.cpu().numpy().view(E.STATUS_DTYPE).reshape(-1)
    print(f"cap {cap:3d}: {ms:8.2f} ms  (+{ms - prev:6.2f})  converged {100.0 * sv['converged'].mean():6.2f} %")
    prev = ms
