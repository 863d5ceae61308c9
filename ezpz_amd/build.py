"""Builds libezpz_amd.so (host C++ + HIP kernels for gfx950) in-tree with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libezpz_amd.so")
SOURCES = ["api.hip", "solve.cpp", "program.cpp", "textual.cpp"]
HEADERS = ["program.hpp", "kinds.hpp", "constraint_eval.hip.hpp", "lm_kernel.hip.hpp", "freedom.hip.hpp", "../../include/ezpz_amd.h"]
# -ffp-contract=off: the reference (Rust) never fuses a*b+c; see constraint_eval.hip.hpp.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


CLI = os.path.join(HERE, "ezpz-amd")


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    cli_src = os.path.join(CSRC, "cli.cpp")
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(cli_src), os.path.getmtime(LIB)):
        # the reference's CLI (ezpz-cli) restated on the C ABI; a plain host program linked against the library
        cmd = [hipcc(), "-O2", "-std=c++17", "-o", CLI, cli_src, "-L" + HERE, "-lezpz_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
