"""Disassembles the run-time compiled kernel of a block system (no device needed): the code object hiprtc makes for
`gen_big_problem.py N [true]`, through the on-disk cache, as text + a one-line summary per entry (registers, spills,
instructions).  Used to check that an edit of jit_kernel.hip.hpp left a kernel it should not touch instruction for
instruction the same:  python tools/jit_isa.py 500 /tmp/before.s ; <edit> ; python tools/jit_isa.py 500 /tmp/after.s"""
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    n = int(sys.argv[1])
    out = sys.argv[2]
    over = len(sys.argv) > 3 and sys.argv[3] == "true"
    d = tempfile.mkdtemp(prefix="ezpz_isa_")
    os.environ["EZPZ_JIT_CACHE_DIR"] = d
    import ezpz_amd as E

    p = E.textual.Problem.from_str(E.textual.gen_big_problem(n, over)).to_constraint_system()
    src = E.specialized_source(p.records, p.num_vars, compile="cached")
    assert src
    (path,) = glob.glob(d + "/*.co")
    blob = open(path, "rb").read()
    magic, tag_b, src_b, code_b, _ = struct.unpack("<8sQQQQ", blob[:40])
    code = blob[40 + tag_b + src_b:40 + tag_b + src_b + code_b]
    co = os.path.join(d, "k.co")
    open(co, "wb").write(code)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    # hiprtc hands out a fat binary or a plain code object; unbundle if needed
    if code[:4] != b"\x7fELF":
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co + ".elf"])
        co = co + ".elf"
    text = subprocess.check_output([objdump, "-d", "--no-show-raw-insn", co]).decode()
    text = re.sub(r"^\s*([a-z_0-9]+ .*?)\s*//\s*[0-9A-F]+:.*$", r"\1", text, flags=re.M)
    open(out, "w").write(text)
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co]).decode()
    for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
        print("%s: sgpr %s (spilled %s) vgpr %s (spilled %s)" % m.groups())
    print("instructions:", len(re.findall(r"^\s+[sv]_|^\s+(global|ds|buffer|flat|scratch)_", text, re.M)))


main()
