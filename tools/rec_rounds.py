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
t = buf.cpu().numpy()[4096:4096 + 126]
t = t[t > 0]
print("rounds", len(t), "cycles per round:", list(np.diff(t)))
print("mean", np.diff(t).mean())
