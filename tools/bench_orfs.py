#!/usr/bin/env python3
"""Throughput of the ORF-fragment producer: mk_extract_orfs on the GPU vs the reference's own
Orf.cpp / TranslateNucl.h (oracle/_ref/ref_harness orfs, one thread -- extractorfs parallelises over contigs) on the same contigs.

  python tools/bench_orfs.py --contigs 10000
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--cpu-contigs", type=int, default=2000, help="contigs in the CPU sample")
    args = ap.parse_args()
    import oracle
    from metaeuk_amd import api, synth
    _, founders = synth.make_targets(2000, args.seed)
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(args.contigs, founders, args.seed)]
    nt = sum(len(c) for c in contigs)
    api.init(0)
    api.Orfs(contigs).close()                                # warm-up (tables, scratch buffers at their final size)
    api.kernel_stats(reset=True)
    t0 = time.time()
    o = api.Orfs(contigs)
    t_gpu = time.time() - t0
    stats = api.kernel_stats()
    out = {"contigs": len(contigs), "nucleotides": nt, "fragments": o.n, "residues": int(o.aa_off[-1]),
           "gpu_wall_s_incl_upload_download": round(t_gpu, 4), "gpu_kernels_ms": round(stats.get("extract_orfs", {"ms": 0})["ms"], 2)}
    if os.path.exists(oracle.REF):
        sample = contigs[:args.cpu_contigs]
        with tempfile.TemporaryDirectory() as tmp:
            cf = os.path.join(tmp, "c.txt")
            open(cf, "w").write("\n".join(sample) + "\n")
            t0 = time.time()
            subprocess.check_call([oracle.REF, "orfs", cf, os.path.join(tmp, "o.txt")], stdout=subprocess.DEVNULL)
            t_cpu = time.time() - t0
        out["cpu_reference_1_thread_s_per_contig"] = t_cpu / len(sample)
        out["cpu_reference_16_threads_s_same_batch_ideal"] = t_cpu / len(sample) * len(contigs) / 16
    print(json.dumps(out))


if __name__ == "__main__":
    main()
