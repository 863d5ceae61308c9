"""Builds and runs tools/ab_microbench.hip on the GPU box: once plainly (its own timing lines) and once under
`rocprofv3 --kernel-trace --stats`; files <outdir>/ab_microbench.txt and <outdir>/ab_microbench_kernel_stats.csv.
usage: python tools/ab_microbench.py <outdir>"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ab"))
os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "ab_microbench")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I",
                       os.path.join(ROOT, "ezpz_amd", "csrc"), os.path.join(ROOT, "tools", "ab_microbench.hip"), "-o", exe])
text = subprocess.check_output([exe], text=True)
open(os.path.join(out, "ab_microbench.txt"), "w").write(
    "# tools/ab_microbench.hip (hipcc -O3 -ffp-contract=off, gfx950): times from HIP events, 20 launches each\n" + text)
print(text)
prof = os.path.join(out, "prof")
subprocess.call(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof, "--", exe],
                env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL)
for f in glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, "ab_microbench_kernel_stats.csv"))
shutil.rmtree(prof, ignore_errors=True)
os.remove(exe)
