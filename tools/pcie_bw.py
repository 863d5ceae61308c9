"""Diagnostic: host<->device copy rates of this box (pageable / pinned, one direction / both at once on two streams)."""
import time
import torch

dev = torch.device("cuda", 0)
N = 64 << 20
hp = torch.empty(N, dtype=torch.uint8)
hq = torch.empty(N, dtype=torch.uint8)
pin_a = torch.empty(N, dtype=torch.uint8).pin_memory()
pin_b = torch.empty(N, dtype=torch.uint8).pin_memory()
d_a = torch.empty(N, dtype=torch.uint8, device=dev)
d_b = torch.empty(N, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timeit(f, reps=10):
    f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


def both():
    with torch.cuda.stream(s1):
        d_a.copy_(pin_a, non_blocking=True)
    with torch.cuda.stream(s2):
        pin_b.copy_(d_b, non_blocking=True)


def chunks(n):
    c = N // n
    for i in range(n):
        with torch.cuda.stream(s1):
            d_a[i * c:(i + 1) * c].copy_(pin_a[i * c:(i + 1) * c], non_blocking=True)
        with torch.cuda.stream(s2):
            pin_b[i * c:(i + 1) * c].copy_(d_b[i * c:(i + 1) * c], non_blocking=True)


print("pageable H2D GB/s", N / timeit(lambda: d_a.copy_(hp)) / 1e9)
print("pageable D2H GB/s", N / timeit(lambda: hq.copy_(d_a)) / 1e9)
print("pinned   H2D GB/s", N / timeit(lambda: d_a.copy_(pin_a, non_blocking=True)) / 1e9)
print("pinned   D2H GB/s", N / timeit(lambda: pin_b.copy_(d_b, non_blocking=True)) / 1e9)
print("pinned both ways at once, GB/s each", N / timeit(both) / 1e9)
print("pinned both ways, 8 chunks each, GB/s each", N / timeit(lambda: chunks(8)) / 1e9)

# the shape of ezpz_system_solve_batch's pipelined path: three streams, each H2D piece -> D2H piece, pieces interleaved
import numpy as np
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ezpz_amd as E

streams = [torch.cuda.Stream(dev) for _ in range(3)]


def pipeline(src, dst, pieces):
    c = N // pieces
    for i in range(pieces):
        with torch.cuda.stream(streams[i % 3]):
            d_a[i * c:(i + 1) * c].copy_(src[i * c:(i + 1) * c], non_blocking=True)
            dst[i * c:(i + 1) * c].copy_(d_a[i * c:(i + 1) * c], non_blocking=True)


for pieces in (4, 8, 16, 32):
    print("3-stream pipeline, torch pinned,", pieces, "pieces: GB/s each way", N / timeit(lambda: pipeline(pin_a, pin_b, pieces)) / 1e9)
na, nb = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
E.host_register(na)
E.host_register(nb)
ta, tb = torch.from_numpy(na), torch.from_numpy(nb)
for pieces in (8, 16):
    print("3-stream pipeline, hipHostRegister'ed numpy,", pieces, "pieces: GB/s each way", N / timeit(lambda: pipeline(ta, tb, pieces)) / 1e9)
print("registered H2D alone GB/s", N / timeit(lambda: d_a.copy_(ta, non_blocking=True)) / 1e9)
