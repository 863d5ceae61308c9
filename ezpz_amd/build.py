"""Builds libezpz_amd.so (host C++ + HIP kernels for gfx950) in-tree with hipcc.

Every source is compiled to its own object (in parallel, only when it or a header is newer) and the objects are
linked into a temporary file that replaces the library atomically, all under a file lock: N ranks starting on an
unbuilt checkout build once, and nobody ever loads a half-written library.
"""
import fcntl
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libezpz_amd.so")
SOURCES = ["api.hip", "comp.hip", "comp_program.cpp", "solve.cpp", "program.cpp", "textual.cpp"]
HEADERS = ["program.hpp", "kinds.hpp", "constraint_eval.hip.hpp", "lm_kernel.hip.hpp", "freedom.hip.hpp", "wave_ops.hip.hpp",
           "comp_kernel.hip.hpp", "comp_program.hpp", "system.hpp", "../../include/ezpz_amd.h"]
# -ffp-contract=off: the reference (Rust) never fuses a*b+c; see constraint_eval.hip.hpp.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return _deps_mtime() > t or any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


CLI = os.path.join(HERE, "ezpz-amd")


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib_path: str = LIB) -> str:
    """extra_flags / lib_path: diagnostic builds (e.g. -DEZPZ_STAMPS into another file) next to the product library."""
    os.makedirs(OBJ, exist_ok=True)
    obj_dir_tag = "" if lib_path == LIB else "." + os.path.basename(lib_path)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or lib_path != LIB or stale():
                objdir = OBJ + obj_dir_tag
                os.makedirs(objdir, exist_ok=True)

                def one(src):
                    obj = os.path.join(objdir, src + ".o")
                    path = os.path.join(CSRC, src)
                    if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), _deps_mtime()):
                        cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-c", "-o", obj, path]
                        if verbose:
                            print(" ".join(cmd), flush=True)
                        subprocess.check_call(cmd, cwd=CSRC)
                    return obj

                with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                    objs = list(pool.map(one, SOURCES))
                tmp = lib_path + ".tmp.%d" % os.getpid()
                cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, lib_path)
            if lib_path == LIB:
                cli_src = os.path.join(CSRC, "cli.cpp")
                if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(os.path.getmtime(cli_src), os.path.getmtime(LIB)):
                    # the reference's CLI (ezpz-cli) restated on the C ABI; a plain host program linked against the library
                    tmp = CLI + ".tmp.%d" % os.getpid()
                    cmd = [hipcc(), "-O2", "-std=c++17", "-o", tmp, cli_src, "-L" + HERE, "-lezpz_amd", "-Wl,-rpath,$ORIGIN"]
                    if verbose:
                        print(" ".join(cmd), flush=True)
                    subprocess.check_call(cmd, cwd=CSRC)
                    os.replace(tmp, CLI)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
