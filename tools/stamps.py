"""Diagnostic: in-kernel s_memtime stamps of block 0 / lane 0 for one launch of the list-walk kernel.  Needs the stamped build:
python -c "import ezpz_amd.build as b; b.build(extra_flags=['-DEZPZ_STAMPS'], lib_path=b.LIB.replace('.so', '_stamps.so'))" """
import ctypes as C, os, sys
os.environ.setdefault("EZPZ_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ezpz_amd", "libezpz_amd_stamps.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E, gen
from oracle import textual as T
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cs = T.load(T.gen_big_problem(lines)); n = cs.num_vars
s = E.System(cs.constraints, n, team_size=E.TEAM_AUTO_LISTS)
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(cs.guesses[None, :] + gen.keyed_uniform(1, B, n, -0.25, 0.25)).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
buf = torch.zeros(2048, dtype=torch.int64, device=dev)
L = E.lib(); L.ezpz_debug_set_stamps.argtypes = [C.c_void_p]; L.ezpz_debug_set_stamps(buf.data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    buf.zero_(); s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream)
torch.cuda.synchronize()
b = buf.cpu().numpy().reshape(-1, 2)
names = {1: "start", 2: "x loaded", 10: "normal eq", 11: "chol+fwd", 12: "bwd", 13: "reduce(bad,dmax)", 14: "x+=d", 20: "R sweep",
         21: "reduce(sq,max)", 22: "J sweep / revert", 40: "  r: loop top", 41: "  r: record loaded", 42: "  r: residual done", 30: "unsat loop", 31: "reduce(unsat)", 32: "x stored"}
prev = None
for i, t in b:
    if i == 0: break
    print(f"{names.get(int(i), i):>18}: +{(t - prev) if prev is not None else 0:7d} ticks")
    prev = t
print("total ticks", b[b[:, 0] > 0][-1, 1] - b[0, 1])

print("--- grid reduction internals (block 0 thread 0) ---")
d = b[300:]
prev = None
for i, t in d:
    if i == 0: break
    print(f"{ {50:'enter',51:'published',52:'all flags seen',53:'folded',54:'scattered'}.get(int(i), i):>18}: +{(t - prev) if prev is not None else 0:7d}")
    prev = t
