"""Diagnostic: where a system on several workgroups spends its time -- wall-clock stamps (100 MHz) of thread 0 of every workgroup of the
class-specialised kernel (jit_kernel.hip.hpp: solve_kernel_grid; JitArgs::stamps through ezpz_debug_set_stamps), over one launch of
the 200 000-variable ladder.  Prints, per stamp, the mean / max over workgroups of the time since the previous stamp and of the time
since the system's first workgroup started; and when each system started relative to the launch.
Stamps when verdicts are not waited for (the default): 0 start, 1 guesses loaded and eval() computed, 2 both steps taken and
stores issued, 3 the barrier, 4 partials published, 5 (workgroup 0) verdict on the previous system written; with EZPZ_JIT_AHEAD=0:
none (the stamps are the fast path's).
usage (GPU box): python tools/ladder_stamps.py [lines=50000] [systems=14]"""
import ctypes as C, os, sys
os.environ["EZPZ_JIT_STAMPS"] = "1"  # (the kernels compiled with their time stamps)
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E, gen
lines = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 14
cs = E.textual.Problem.from_str(E.textual.gen_big_problem(lines)).to_constraint_system()
n = cs.num_vars
s = E.System(cs.records, n)
assert s.specialize(wait=True) == 2
G = ((lines + 63) // 64 + 7) // 8  # workgroups per system of the specialised kernel (comp_program.cpp): 4 wavefronts x 2 slots of 64 two-variable components
print("workgroups per system", G)
dev = torch.device('cuda', 0)
x0h = cs.guesses[None, :] + gen.keyed_uniform(5, B, n, -0.25, 0.25)
x0 = torch.from_numpy(x0h).to(dev)
xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
buf = torch.zeros(B * G * 16, dtype=torch.int64, device=dev)
L = E.lib(); L.ezpz_debug_set_stamps.argtypes = [C.c_void_p]
stream = torch.cuda.current_stream(dev).cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(5):
    if rep == 4: L.ezpz_debug_set_stamps(buf.data_ptr())
    buf.zero_()
    ev0.record()
    s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, E.Config())
    ev1.record()
    torch.cuda.synchronize()
    print(f"launch {rep}: {ev0.elapsed_time(ev1) * 1000:.1f} us for {B} systems")
L.ezpz_debug_set_stamps(None)
t = buf.cpu().numpy().reshape(B, G, 16).astype(np.float64) / 100.0  # us
t[t == 0] = np.nan
t0 = np.nanmin(t)
ns = int(np.sum(~np.isnan(t[0, 0])))
print("stamps per (system, workgroup):", ns)
print("system starts (us after the launch's first stamp), first / last workgroup:")
for b in range(B):
    print(f"  system {b:3d}: {np.nanmin(t[b, :, 0]) - t0:8.2f} .. {np.nanmax(t[b, :, 0]) - t0:8.2f}   ends {np.nanmax(t[b, :, ns - 1]) - t0:8.2f}")
# period of one slot: start to start of its consecutive systems (n_slots systems apart), earliest workgroup and workgroup 0
slots = int(os.environ.get("LADDER_SLOTS", "7"))
starts = np.array([np.nanmin(t[b, :, 0]) for b in range(0, B, slots)])
print("slot 0: start-to-start periods (us):", " ".join(f"{d:.1f}" for d in np.diff(starts)))
print("slot 0, per system: last stamp of the slowest workgroup - first stamp of the fastest (us):",
      " ".join(f"{np.nanmax(t[b]) - np.nanmin(t[b]):.1f}" for b in range(0, B, slots)))
for b in sorted(set([0, B // 2, B - 1])):
    print(f"system {b}: stamp: since previous (mean / max over workgroups; workgroup 0) | since the system's start (mean / max)")
    s0 = np.nanmin(t[b, :, 0])
    for k in range(ns):
        d = t[b, :, k] - (t[b, :, k - 1] if k else t[b, :, 0])
        print(f"  {k:2d}: {np.nanmean(d):7.2f} / {np.nanmax(d):7.2f} ; wg0 {d[0]:7.2f} | {np.nanmean(t[b, :, k]) - s0:7.2f} / {np.nanmax(t[b, :, k]) - s0:7.2f}")
