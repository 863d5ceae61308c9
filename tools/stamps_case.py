"""Diagnostic: in-kernel cycle stamps of block 0 / lane 0 for one solve of a test_cases fixture (needs libezpz_amd_stamps.so)."""
import ctypes as C, os, sys
os.environ.setdefault("EZPZ_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ezpz_amd", "libezpz_amd_stamps.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
from oracle import oracle as O, textual as T
from conftest import read_case
name = sys.argv[1] if len(sys.argv) > 1 else "square"
team = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # systems per launch (stamps come from block 0 / lane 0 only)
if name.startswith("sketch:"):  # tests/gen.py:connected_sketch
    import gen, types
    recs, g = gen.connected_sketch(int(name[7:]), 1000 + int(name[7:]))
    ref = types.SimpleNamespace(guesses=g, num_vars=len(g))
else:
    ref = T.load(read_case(name))
    recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
s = E.System(recs, ref.num_vars, team_size=team)
print(s.info())
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(np.repeat(ref.guesses[None, :], B, axis=0).copy()).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
buf = torch.zeros(1 << 16, dtype=torch.int64, device=dev)
L = E.lib(); L.ezpz_debug_set_stamps.argtypes = [C.c_void_p]; L.ezpz_debug_set_stamps(buf.data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    buf.zero_(); s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream)
torch.cuda.synchronize()
b = buf.cpu().numpy().reshape(-1, 2)
names = {1: "start", 2: "x loaded", 10: "normal eq", 11: "chol+fwd", 12: "bwd", 13: "reduce(bad,dmax)", 14: "x+=d", 20: "R sweep",
         21: "reduce(sq,max)", 22: "J sweep / revert", 30: "unsat loop", 31: "reduce(unsat)", 32: "x stored"}
prev = None; tot = {}
for i, t in b:
    if i == 0: break
    if int(i) in (40, 41, 42) or int(i) >= 1000: continue
    d = (t - prev) if prev is not None else 0
    tot[names.get(int(i), i)] = tot.get(names.get(int(i), i), 0) + d
    prev = t
lv = [(int(i), int(t - b[k - 1][1])) for k, (i, t) in enumerate(b) if 1000 <= i < 3000]
nl = max(i % 1000 for i, _ in lv) + 1 if lv else 0
if lv:
    print("Cholesky levels of the first iteration (cycles; 2xxx = staged from global):", [(i, d) for i, d in lv[:nl]])
for k, v in tot.items(): print(f"{k:>18}: {v:8d} cycles total")
print("total", b[b[:, 0] > 0][-1, 1] - b[0, 1])
