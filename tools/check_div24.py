"""Scans the gfx950 code of the product (metaeuk_amd/lib/obj/*.hip.o) for the compiler's 24-bit integer-division expansion
(AMDGPUCodeGenPrepare::expandDivRem24): v_rcp_iflag_f32 ... v_trunc_f32, v_fma_f32 -q, n, y, v_cmp_ge_f32 |r|, n.  For numerators of a full 24 bits
that sequence returns remainder 0xFFFFFF where the true remainder is n - 1 (tools/micro/urem24.hip reproduces it on the device and models it on
the host; DESIGN.md 4.13, "the lost subset"): a `hash % n` on values the compiler can prove to fit 24 bits must not come back into a kernel.
   python tools/check_div24.py            -> one line per occurrence (object, kernel); exit status 1 when there is one"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj, tmp):
    fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.o")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], stderr=subprocess.DEVNULL)
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", dev]).decode()


def find(text):
    """-> [(kernel, line number)] of every v_trunc_f32 that is followed by the negated fma and the |r| >= n compare"""
    out, kernel = [], "?"
    lines = text.splitlines()
    for i, ln in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            kernel = m.group(1)
        if "v_trunc_f32" in ln:
            window = lines[i + 1:i + 8]
            if any(re.search(r"v_(fma|mad|fmac)_f32.* -v", w) for w in window) and any(re.search(r"v_cmp_ge_f32.*\|v", w) for w in window):
                out.append((kernel, i + 1))
    return out


def scan(objs=None):
    objs = objs or sorted(glob.glob(os.path.join(ROOT, "metaeuk_amd", "lib", "obj", "*.hip.o")))
    found = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            for kernel, line in find(disassemble(o, tmp)):
                found.append((os.path.basename(o), kernel, line))
    return objs, found


if __name__ == "__main__":
    objs, found = scan(sys.argv[1:] or None)
    for o, k, ln in found:
        print("%s: %s (disassembly line %d)" % (o, k, ln))
    print("%d object files, %d occurrences of the 24-bit division expansion" % (len(objs), len(found)))
    sys.exit(1 if found else 0)
