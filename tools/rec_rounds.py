"""Diagnostic: cycles per round of the record walk (lm_kernel.hip.hpp, REC builds) on wavefront 0, second LM iteration of one solve of
tests/gen.py:connected_sketch(npts).  Needs a library built with -DEZPZ_STAMPS -DEZPZ_REC_TIMES (ezpz_amd/build.py: build(extra_flags=...,
lib_path=...)) named by EZPZ_AMD_LIB.

usage (GPU box): EZPZ_AMD_LIB=$PWD/ezpz_amd/libezpz_amd_stamps.so python tools/rec_rounds.py 150"""
import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
import gen
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
recs, g = gen.connected_sketch(npts, 1000 + npts)
s = E.System(recs, len(g), team_size=0xFFFFFFFF)
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(np.asarray(g)[None, :].copy()).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
L = E.lib(); L.ezpz_debug_set_stamps.argtypes = [C.c_void_p]; L.ezpz_debug_set_stamps(buf.data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    buf.zero_(); s.solve_batch_device(x0.data_ptr(), 1, xo.data_ptr(), st.data_ptr(), 0, stream)
torch.cuda.synchronize()
t = buf.cpu().numpy()[4096:4096 + 6 * 126].reshape(-1, 6)
t = t[t[:, 0] > 0]
print("# round: start -> rendezvous' exit (record wait, requests, the other wavefronts) | -> partial sums (operand loads) | -> group sums | -> 1/sqrt and quotient | -> end (stores) | -> next round's start   [backward rounds: no group-sum / sqrt stamps]")
for r in range(len(t)):
    nxt = t[r + 1, 0] - t[r, 5] if r + 1 < len(t) else 0
    if t[r, 2] <= t[r, 3] <= t[r, 4] <= t[r, 5]:  # (backward rounds leave the two middle stamps as they were)
        print(f"round {r:3d}: {t[r, 1] - t[r, 0]:5d} | {t[r, 2] - t[r, 1]:5d} | {t[r, 3] - t[r, 2]:5d} | {t[r, 4] - t[r, 3]:5d} | {t[r, 5] - t[r, 4]:5d} | {nxt:5d}")
    else:
        print(f"round {r:3d}: {t[r, 1] - t[r, 0]:5d} | {t[r, 2] - t[r, 1]:5d} |     - |     - | {t[r, 5] - t[r, 2]:5d} | {nxt:5d}")
print("rounds", len(t), "mean cycles per round", (t[-1, 5] - t[0, 0]) / len(t))
