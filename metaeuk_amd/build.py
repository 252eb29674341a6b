"""Builds metaeuk_amd/lib/libmetaeuk_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

The extension is built IN-TREE so that it travels with the repository snapshot to the GPU box.
hipcc cross-compiles for gfx950 without a GPU being present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("mk_host.cpp", "mk_exons.cpp", "mk_indexfile.cpp", "mk_abi.cpp", "mk_sw.hip", "mk_align.hip", "mk_prefilter.hip", "mk_derive.hip", "mk_profile.hip", "mk_kmer7.hip", "mk_orf.hip", "mk_index.hip", "mk_synth.cpp", "mk_cli.cpp")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("mk_host.hpp", "mk_kernels.hpp", "mk_prefilter.hpp", "mk_align.hpp", "mk_dbio.hpp", "mk_enum.hpp", "mk_profile.hpp", "mk_kmer7.hpp", "mk_orf.hpp", "mk_exons.hpp", "mk_indexfile.hpp", "mk_index.hpp", "mk_segsort.hpp")] + [
    os.path.join(HERE, "..", "include", "metaeuk_amd.h"), os.path.join(HERE, "..", "include", "metaeuk_amd_debug.h"), os.path.join(HERE, "data", "matrices.inc")]
LIB = os.path.join(HERE, "lib", "libmetaeuk_amd.so")
BIN = os.path.join(HERE, "lib", "metaeuk-amd")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off",
         "-Wno-unused-value", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [s for s in SRC if os.path.exists(s)]
    lib_srcs = [s for s in srcs if not s.endswith("mk_cli.cpp")]
    objdir = os.path.join(os.path.dirname(LIB), "obj")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in lib_srcs:                      # one object per translation unit, compiled in parallel, rebuilt only when stale
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + HDR):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc] + FLAGS + ["-shared"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli = os.path.join(HERE, "csrc", "mk_cli.cpp")
    if os.path.exists(cli) and (force or _stale(BIN, [cli, LIB] + HDR)):
        cmd = [hipcc] + FLAGS + [cli, "-o", BIN, "-L" + os.path.dirname(LIB), "-lmetaeuk_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
