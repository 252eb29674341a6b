"""Pins the config-5-shaped test (tests/test_gpu_scale.py::test_config5_scale) to the REFERENCE: generates the same seeded database and
planted fragments the test generates (native generator, metaeuk_amd/csrc/mk_synth.cpp), runs the reference's own compiled prefilter +
align over them (oracle/_ref/ref_harness, AVX2) and writes the digests of its outputs to tests/golden/config5_digest_<n_targets>.json.

  python tools/config5_digest.py <n_targets> [-k 7] [--queries 20000] [--threads N] [--out tests/golden/...json] [--work /tmp/config5]

The digest is what tests/oracle.py::digest_blocks_file defines (per-query line counts + the lines) -- the test hashes the GPU's result
the same way (digest_arrays).  Run it where the memory is: ~12 bytes per target residue + 10 GB for a k = 7 table."""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FRAGMENTS = dict(seed=5, mutation_rate=0.1, min_len=30, max_len=120, random_every=10)      # shared with the test
TARGET_SEED = 11


def write_lines(path, res, off):
    """one sequence per line (what the harness reads), without a Python loop over the sequences"""
    from metaeuk_amd import api
    data, _, _, _ = api.synth_seqdb(res, off)
    CH = 1 << 28
    with open(path, "wb") as f:
        for a in range(0, data.size, CH):
            piece = data[a:a + CH]
            f.write(piece[piece != 0].tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_targets", type=int)
    ap.add_argument("-k", type=int, default=0)
    ap.add_argument("--queries", type=int, default=20000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--work", default="/tmp/config5_digest")
    a = ap.parse_args()
    import oracle
    from metaeuk_amd import api
    threads = a.threads or api.lib().mk_host_threads()
    os.makedirs(a.work, exist_ok=True)
    t0 = time.time()
    res, off = api.synth_targets(a.n_targets, seed=TARGET_SEED)
    fr, foff, src = api.synth_fragments(a.queries, res, off, **FRAGMENTS)
    write_lines(os.path.join(a.work, "targets.txt"), res, off)
    write_lines(os.path.join(a.work, "queries.txt"), fr, foff)
    t_gen = time.time() - t0
    mat = oracle.REF_MATDIR if os.path.isdir(oracle.REF_MATDIR) else oracle.write_matrix_files(os.path.join(a.work, "mat"))
    cmd = [oracle.REF, "pipeline", mat, os.path.join(a.work, "targets.txt"), os.path.join(a.work, "queries.txt"), os.path.join(a.work, "ref"),
           "--threads", str(threads)] + (["-k", str(a.k)] if a.k else [])
    t0 = time.time()
    line = subprocess.check_output(cmd, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
    t_ref = time.time() - t0
    info = json.loads(line)
    d_pref, n1 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "pref.txt"))
    d_aln, n2 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "aln.txt"))
    assert n1 == n2 == a.queries
    l2 = ctypes.CDLL(None).sysconf(191)
    out = dict(n_targets=a.n_targets, target_seed=TARGET_SEED, target_residues=int(off[-1]), n_queries=a.queries, fragments=FRAGMENTS, kmer_size_forced=a.k,
               reference=dict(k=info["k"], kmer_thr=info["kmer_thr"], pref_hits=info["pref_hits"], alignments=info["alignments"], passed=info["passed"],
                              t_index_s=info["t_index"], t_prefilter_s=info["t_prefilter"], t_align_s=info["t_align"], threads=threads, wall_s=round(t_ref, 1)),
               host_l2_bytes=int(l2 if l2 and l2 > 0 else 262144), sha256_pref=d_pref, sha256_aln=d_aln,
               made_by="tools/config5_digest.py: oracle/_ref/ref_harness (the reference's translation units compiled by oracle/Makefile.ref)",
               t_generate_s=round(t_gen, 1))
    path = a.out or os.path.join(ROOT, "tests", "golden", "config5_digest_%d.json" % a.n_targets)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))
    for name in ("targets.txt", "queries.txt"):
        os.remove(os.path.join(a.work, name))


if __name__ == "__main__":
    main()
