"""Pins the config-5-shaped test (tests/test_gpu_scale.py::test_config5_scale) to the REFERENCE: generates the same seeded database and
planted fragments the test generates (native generator, metaeuk_amd/csrc/mk_synth.cpp), runs the reference's own compiled prefilter +
align over them (oracle/_ref/ref_harness, AVX2) and writes the digests of its outputs to tests/golden/config5_digest_<n_targets>.json.

  python tools/config5_digest.py <n_targets> [-k 7] [--queries 20000] [--threads N] [--out tests/golden/...json] [--work /tmp/config5]
                                 [--split N] [--long M]
--split N: the reference in TARGET_DB_SPLIT mode (Prefiltering.cpp:273-377,352-362, merge :379-496): N residue-balanced target ranges, each masked,
indexed (k from the residues per range) and searched on its own, the lists joined by the reference's own mergeTargetSplits -- how the reference itself
runs a database whose index does not fit the host (60 M proteins unsplit: ~270 GB), and what `metaeuk-amd prefilter --split N --split-mode 0`
reproduces.  --long M: M more fragments of 300 .. 1 500 residues behind the planted ones (the heavy queries of the wide kernel).

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
LONG_FRAGMENTS = dict(seed=6, mutation_rate=0.1, min_len=300, max_len=1500, random_every=0)
TARGET_SEED = 11


def make_fragments(api, res, off, n_short, n_long):
    """the planted fragments (+ n_long long ones behind them): residues, offsets -- the test generates the same"""
    fr, foff, src = api.synth_fragments(n_short, res, off, **FRAGMENTS)
    if n_long:
        fr2, foff2, src2 = api.synth_fragments(n_long, res, off, **LONG_FRAGMENTS)
        fr = np.concatenate([fr[:int(foff[-1])], fr2[:int(foff2[-1])]])
        foff = np.concatenate([foff, foff2[1:] + foff[-1]]).astype(np.uint64)
        src = np.concatenate([src, src2])
    return fr, foff, src


CONTIG_SEED, CONTIG_PICK_SEED = 23, 29


def make_contigs(api, res, off, n_contigs):
    """n_contigs seeded 5-kb contigs, each with a multi-exon gene derived (25 % mutated) from a protein of the database -- picked over the whole
    range of ids -- plus one contig without a gene; as strings.  The test generates the same."""
    from metaeuk_amd import synth
    n_targets = len(off) - 1
    rs = np.random.RandomState(CONTIG_PICK_SEED)
    picked = rs.randint(0, n_targets, size=max(1, n_contigs))
    founders = [np.array(res[int(off[t]):int(off[t + 1])], dtype=np.uint8) for t in picked]
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(n_contigs, founders, seed=CONTIG_SEED)]
    contigs.append("ACGT" * 600)
    return contigs


def exon_sets_digest(per_contig):
    """SHA-256 over '>contig\n' + its prediction lines for contigs 0 .. n-1 (per_contig: list of strings)"""
    import hashlib
    return hashlib.sha256("".join(">%d\n%s" % (c, per_contig[c]) for c in range(len(per_contig))).encode()).hexdigest()


def end_to_end(a, api, oracle, threads):
    """--contigs N: data/predictexons.sh:42-87 by the reference's own code over the config-5 database in TARGET_DB_SPLIT mode: ref_harness orfs |
    pipeline --split N | exons on N seeded contigs -> tests/golden/config5_e2e_<n_targets>_split<N>.json (fragment count, digests of the prefilter
    lists, the alignments and the exon sets)."""
    os.makedirs(a.work, exist_ok=True)
    t0 = time.time()
    res, off = api.synth_targets(a.n_targets, seed=TARGET_SEED)
    contigs = make_contigs(api, res, off, a.contigs)
    if a.n_targets >= 20000000:
        api.synth_write_seqdb(os.path.join(a.work, "T"), res, off, with_lines=True)
        os.rename(os.path.join(a.work, "T.txt"), os.path.join(a.work, "targets.txt"))
        for sfx in ("", ".index", ".dbtype"):
            os.remove(os.path.join(a.work, "T" + sfx))
    else:
        write_lines(os.path.join(a.work, "targets.txt"), res, off)
    n_res = int(off[-1])
    del res
    with open(os.path.join(a.work, "c.txt"), "w") as f:
        f.write("\n".join(contigs) + "\n")
    t_gen = time.time() - t0
    mat = oracle.REF_MATDIR if os.path.isdir(oracle.REF_MATDIR) else oracle.write_matrix_files(os.path.join(a.work, "mat"))
    t0 = time.time()
    subprocess.check_call([oracle.REF, "orfs", os.path.join(a.work, "c.txt"), os.path.join(a.work, "orfs.txt")], stdout=subprocess.DEVNULL)
    n_orf = 0
    with open(os.path.join(a.work, "q.txt"), "w") as f:
        for line in open(os.path.join(a.work, "orfs.txt")):
            if not line.startswith(">"):
                f.write(line.rstrip("\n").rsplit("\t", 1)[1] + "\n")
                n_orf += 1
    cmd = [oracle.REF, "pipeline", mat, os.path.join(a.work, "targets.txt"), os.path.join(a.work, "q.txt"), os.path.join(a.work, "ref"),
           "--threads", str(threads)] + (["-k", str(a.k)] if a.k else []) + (["--split", str(a.split)] if a.split > 1 else [])
    line = subprocess.check_output(cmd, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
    info = json.loads(line)
    t_pipe = time.time() - t0
    t1 = time.time()
    subprocess.check_call([oracle.REF, "exons", os.path.join(a.work, "targets.txt"), os.path.join(a.work, "c.txt"), os.path.join(a.work, "orfs.txt"),
                           os.path.join(a.work, "ref", "aln.txt"), os.path.join(a.work, "exons.txt")], stdout=subprocess.DEVNULL)
    t_exons = time.time() - t1
    exp, cur = {}, None
    for ln in open(os.path.join(a.work, "exons.txt")):
        if ln.startswith(">"):
            cur = int(ln[1:])
            exp[cur] = ""
        else:
            exp[cur] += ln
    per = [exp.get(c, "") for c in range(len(contigs))]
    d_pref, n1 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "pref.txt"))
    d_aln, n2 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "aln.txt"))
    assert n1 == n2 == n_orf
    l2 = ctypes.CDLL(None).sysconf(191)
    out = dict(n_targets=a.n_targets, target_seed=TARGET_SEED, target_residues=n_res, n_contigs=len(contigs), contig_seed=CONTIG_SEED, contig_pick_seed=CONTIG_PICK_SEED,
               fragments=n_orf, target_splits=a.split, kmer_size_forced=a.k,
               reference=dict(k=info["k"], kmer_thr=info["kmer_thr"], pref_hits=info["pref_hits"], alignments=info["alignments"], passed=info["passed"],
                              t_index_s=info["t_index"], t_prefilter_s=info["t_prefilter"], t_align_s=info["t_align"], threads=threads,
                              wall_orfs_and_pipeline_s=round(t_pipe, 1), wall_exons_s=round(t_exons, 1),
                              contigs_with_predictions=sum(1 for x in per if x), prediction_lines=sum(x.count("\n") for x in per)),
               host_l2_bytes=int(l2 if l2 and l2 > 0 else 262144), sha256_pref=d_pref, sha256_aln=d_aln, sha256_exon_sets=exon_sets_digest(per),
               made_by="tools/config5_digest.py --contigs: oracle/_ref/ref_harness orfs | pipeline --split | exons (the reference's translation units compiled by oracle/Makefile.ref)",
               t_generate_s=round(t_gen, 1))
    path = a.out or os.path.join(ROOT, "tests", "golden", "config5_e2e_%d_split%d.json" % (a.n_targets, a.split))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))
    for name in ("targets.txt", "q.txt", "c.txt", "orfs.txt", "exons.txt"):
        if os.path.exists(os.path.join(a.work, name)):
            os.remove(os.path.join(a.work, name))


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
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--long", type=int, default=0)
    ap.add_argument("--contigs", type=int, default=0, help="end-to-end mode: this many seeded contigs through orfs | pipeline | exons")
    a = ap.parse_args()
    import oracle
    from metaeuk_amd import api
    threads = a.threads or api.lib().mk_host_threads()
    if a.contigs > 0:
        return end_to_end(a, api, oracle, threads)
    os.makedirs(a.work, exist_ok=True)
    t0 = time.time()
    res, off = api.synth_targets(a.n_targets, seed=TARGET_SEED)
    fr, foff, src = make_fragments(api, res, off, a.queries, a.long)
    n_all = len(foff) - 1
    if a.n_targets >= 20000000:                                    # the native writer streams to disk (no 23 GB image in memory)
        api.synth_write_seqdb(os.path.join(a.work, "T"), res, off, with_lines=True)
        os.rename(os.path.join(a.work, "T.txt"), os.path.join(a.work, "targets.txt"))
        for sfx in ("", ".index", ".dbtype"):
            os.remove(os.path.join(a.work, "T" + sfx))
    else:
        write_lines(os.path.join(a.work, "targets.txt"), res, off)
    write_lines(os.path.join(a.work, "queries.txt"), fr, foff)
    n_res = int(off[-1])
    del res
    t_gen = time.time() - t0
    mat = oracle.REF_MATDIR if os.path.isdir(oracle.REF_MATDIR) else oracle.write_matrix_files(os.path.join(a.work, "mat"))
    cmd = [oracle.REF, "pipeline", mat, os.path.join(a.work, "targets.txt"), os.path.join(a.work, "queries.txt"), os.path.join(a.work, "ref"),
           "--threads", str(threads)] + (["-k", str(a.k)] if a.k else []) + (["--split", str(a.split)] if a.split > 1 else [])
    t0 = time.time()
    line = subprocess.check_output(cmd, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
    t_ref = time.time() - t0
    info = json.loads(line)
    d_pref, n1 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "pref.txt"))
    d_aln, n2 = oracle.digest_blocks_file(os.path.join(a.work, "ref", "aln.txt"))
    assert n1 == n2 == n_all
    l2 = ctypes.CDLL(None).sysconf(191)
    out = dict(n_targets=a.n_targets, target_seed=TARGET_SEED, target_residues=n_res, n_queries=a.queries, n_long_queries=a.long, fragments=FRAGMENTS,
               long_fragments=LONG_FRAGMENTS if a.long else None, target_splits=a.split, kmer_size_forced=a.k,
               reference=dict(k=info["k"], kmer_thr=info["kmer_thr"], pref_hits=info["pref_hits"], alignments=info["alignments"], passed=info["passed"],
                              t_index_s=info["t_index"], t_prefilter_s=info["t_prefilter"], t_align_s=info["t_align"], threads=threads, wall_s=round(t_ref, 1)),
               host_l2_bytes=int(l2 if l2 and l2 > 0 else 262144), sha256_pref=d_pref, sha256_aln=d_aln,
               made_by="tools/config5_digest.py: oracle/_ref/ref_harness (the reference's translation units compiled by oracle/Makefile.ref)",
               t_generate_s=round(t_gen, 1))
    path = a.out or os.path.join(ROOT, "tests", "golden", ("config5_digest_%d.json" % a.n_targets) if a.split <= 1 else ("config5_digest_%d_split%d.json" % (a.n_targets, a.split)))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out))
    for name in ("targets.txt", "queries.txt"):
        os.remove(os.path.join(a.work, name))


if __name__ == "__main__":
    main()
