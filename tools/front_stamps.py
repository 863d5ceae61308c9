"""Diagnostic: cycle stamps of workgroup 0 / thread 0 over one solve on the frontal shape (front_kernel.hip.hpp, FRONT_STAMP): where an LM
iteration's time goes -- inside wavefront 0's fronts (extend-add, pivots, Schur complement; the other wavefronts run their own lists),
the backward substitution, the sweeps and the reductions.  Needs the stamped build:
python -c "import ezpz_amd.build as b; b.build(extra_flags=['-DEZPZ_STAMPS'], lib_path=b.LIB.replace('.so', '_stamps.so'))"
usage (GPU box): python tools/front_stamps.py <points> [workgroups] [iteration to print]"""
import ctypes as C, os, sys
os.environ.setdefault("EZPZ_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ezpz_amd", "libezpz_amd_stamps.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E, gen
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 150
if len(sys.argv) > 2 and int(sys.argv[2]) > 0: os.environ["EZPZ_FRONT_WGS"] = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
recs, g = gen.connected_sketch(npts, 1000 + npts)
s = E.System(recs, len(g), team_size=E.TEAM_FRONTS)
print(s.info())
dev = torch.device('cuda', 0)
x0 = torch.from_numpy(np.asarray(g)[None, :].copy()).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
buf = torch.zeros(8192, dtype=torch.int64, device=dev)
L = E.lib(); L.ezpz_debug_set_stamps.argtypes = [C.c_void_p]; L.ezpz_debug_set_stamps(buf.data_ptr())
stream = torch.cuda.current_stream(dev).cuda_stream
cfg = E.Config(max_iterations=60)
for _ in range(3):
    buf.zero_(); s.solve_batch_device(x0.data_ptr(), 1, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
torch.cuda.synchronize()
b = buf.cpu().numpy().reshape(-1, 2)
b = b[b[:, 0] > 0]
names = {1: "start", 2: "x loaded", 11: "factorised, verdict known", 12: "backward done", 14: "x += d", 20: "residual sweep", 21: "reduction",
         22: "accept: Jacobian sweep / reject", 30: "loop left", 32: "stored", 100: "  front: zeroed", 101: "  front: assembled", 102: "  front: children added",
         103: "  front: pivots", 104: "  front: Schur complement"}
# iterations are delimited by stamp 22
it, prev = 0, None
print(f"total cycles {b[-1, 1] - b[0, 1]} over {int((b[:, 0] == 22).sum())} trips of the loop")
for i, t in b:
    i = int(i)
    if it == which:
        nm = names.get(i, "every wavefront's fronts factorised" if i == 1000 else "... substituted back" if i == 2000 else str(i))
        print(f"{nm:>34}: +{(t - prev) if prev is not None else 0:7d}")
    if i == 22: it += 1
    prev = t
