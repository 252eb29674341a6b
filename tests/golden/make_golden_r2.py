#!/usr/bin/env python3
"""Round-2 additions to the golden fixtures, generated in the build container from oracle/_ref/ref_harness (the reference's own
sources, AVX2 and SSE4.1 builds must agree):

  sw2_*   Smith-Waterman pairs at the top of the int16 range of the reference's word pass (simdi16_adds,
          StripedSmithWaterman.cpp:1059): identical and near-identical pairs of 2000 .. 5600 residues, with and without flanks
          (scores up to ~32 000).  A pair that actually REACHES 32767 has no defined reference result: the AVX2 and the
          SSE4.1 build return different garbage (no end position, uninitialised e-value) -- sw_saturation() checks that and
          keeps such pairs out of the fixture.

  round 3: the fixtures of the profile-target path (prof_pref / prof_aln / prof_search_res), of k = 7 (e2e_process_pref_k7) and of the
  target splits (e2e_process_pref_split3_maxseqs20) regenerated from `ref_harness profilesearch`, `pipeline -k 7` and `pipeline --split 3`
  -- the reference's translation units compiled by oracle/Makefile.ref (Sequence::mapProfile, KmerGenerator's profile and k = 7 divide
  strategies, Matcher PROFILE_SEQ, result_t::swapResult, Prefiltering::mergeTargetSplits) -- and REQUIRED to equal, byte for byte, the
  files the real binary wrote (make_process_golden.sh / make_profile_golden.sh): harness_regenerated() raises otherwise.  tests/
  test_oracle_golden.py repeats the comparison on every CPU run in the build container.

  python tests/golden/make_golden_r2.py
"""
import gzip
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_AVX2 = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REF_SSE = os.path.join(ROOT, "oracle", "_ref", "sse41", "ref_harness")
MATDIR = "/root/reference/lib/mmseqs/data"
AA = "ACDEFGHIKLMNPQRSTVWY"


def write(name, text):
    with gzip.open(os.path.join(HERE, name), "wt", compresslevel=9) as f:
        f.write(text)


def sw_saturation(tmp):
    rng = random.Random(2024)
    rs = lambda n: "".join(rng.choice(AA) for _ in range(n))
    mut = lambda s, r: "".join(ch if rng.random() > r else rng.choice(AA) for ch in s)
    queries, targets = [], []
    for n, rate, flank in ((3500, 0.0, 0), (3300, 0.0, 40), (4500, 0.1, 25), (3100, 0.0, 0), (2000, 0.0, 10), (4000, 0.02, 0), (3900, 0.3, 60), (3200, 0.0, 300),
                           (5600, 0.0, 0), (5500, 0.0, 30)):
        base = rs(n)
        queries.append(base)
        targets.append(rs(flank) + mut(base, rate) + rs(flank))
    n_defined = len(queries)
    queries.append("W" * 3000); targets.append("W" * 3200)          # one diagonal, 22 per cell: reaches 32767 after ~1500 columns
    base = rs(6500); queries.append(base); targets.append(base)      # identical pair beyond 32767
    tf, qf, pf = os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt"), os.path.join(tmp, "p.txt")
    open(tf, "w").write("\n".join(targets) + "\n"); open(qf, "w").write("\n".join(queries) + "\n")
    open(pf, "w").write("\n".join("%d %d" % (i, i) for i in range(len(queries))) + "\n")
    outs = []
    for b in (REF_AVX2, REF_SSE):
        o = os.path.join(tmp, "sw_" + str(len(outs)))
        subprocess.check_call([b, "sw", MATDIR, tf, qf, pf, o, "--dbres", "7500000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(open(o).read())
    a, b = outs[0].splitlines(), outs[1].splitlines()
    assert a[:n_defined] == b[:n_defined], "AVX2 and SSE4.1 reference builds disagree below the saturation point"
    keep = list(range(n_defined))
    for k in range(n_defined, len(a)):
        print("saturating pair %d: AVX2 %r / SSE4.1 %r -> %s" % (k, a[k], b[k], "agree, kept" if a[k] == b[k] else "DISAGREE (undefined reference behaviour, not in the fixture)"))
        if a[k] == b[k]:
            keep.append(k)
    rows = []
    for n, k in enumerate(keep):                                   # renumber the kept pairs
        f = a[k].split("\t")
        f[0] = f[1] = f[2] = str(n)
        rows.append("\t".join(f))
    return [targets[k] for k in keep], [queries[k] for k in keep], "\n".join(rows) + "\n"


def read_gz(name):
    with gzip.open(os.path.join(HERE, name), "rt") as f:
        return f.read()


def harness_regenerated(tmp, ref=REF_AVX2, threads=8, k7_fragments=None):
    """{fixture name: text} of the profile / k = 7 / target-split fixtures as the reference harness produces them"""
    out = {}
    frags = [l.rsplit("\t", 1)[1] for l in read_gz("e2e_process_orfs.txt.gz").splitlines()]
    order = [int(x) for x in read_gz("prof_frag_order.txt.gz").split()]
    with open(os.path.join(tmp, "prof.bin"), "wb") as f:
        f.write(gzip.open(os.path.join(HERE, "prof_db.bin.gz"), "rb").read())
    with open(os.path.join(tmp, "frags.txt"), "w") as f:
        f.write("\n".join(frags[k] for k in order) + "\n")
    with open(os.path.join(tmp, "keys.txt"), "w") as f:
        f.write("\n".join(str(k) for k in order) + "\n")
    quiet = dict(stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([ref, "profilesearch", MATDIR, os.path.join(tmp, "prof.bin"), os.path.join(HERE, "prof_db.index"), os.path.join(tmp, "frags.txt"),
                           os.path.join(tmp, "prof"), "--keys", os.path.join(tmp, "keys.txt"), "--threads", str(threads)], **quiet)
    for f, name in (("pref.txt", "prof_pref.txt.gz"), ("aln.txt", "prof_aln.txt.gz"), ("swapped.txt", "prof_search_res.txt.gz")):
        out[name] = open(os.path.join(tmp, "prof", f)).read()
    with open(os.path.join(tmp, "t.txt"), "w") as f:
        f.write(read_gz("e2e_targets.txt.gz"))
    with open(os.path.join(tmp, "q.txt"), "w") as f:
        f.write("".join(x + "\n" for x in frags))
    subprocess.check_call([ref, "pipeline", MATDIR, os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt"), os.path.join(tmp, "split"), "-s", "5.7", "--split", "3",
                           "--max-seqs", "20", "--threads", str(threads), "--no-align"], **quiet)
    out["e2e_process_pref_split3_maxseqs20.txt.gz"] = open(os.path.join(tmp, "split", "pref.txt")).read()
    if k7_fragments:                     # (a prefix: the k = 7 index table costs the reference ~1 minute to set up and ~10 ms per fragment)
        with open(os.path.join(tmp, "q.txt"), "w") as f:
            f.write("".join(x + "\n" for x in frags[:k7_fragments]))
    subprocess.check_call([ref, "pipeline", MATDIR, os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt"), os.path.join(tmp, "k7"), "-s", "5.7", "-k", "7",
                           "--threads", str(threads), "--no-align"], **quiet)
    out["e2e_process_pref_k7.txt.gz"] = open(os.path.join(tmp, "k7", "pref.txt")).read()
    return out


def main():
    with tempfile.TemporaryDirectory() as tmp:
        for ref in (REF_AVX2, REF_SSE):
            if not os.path.exists(ref):
                continue
            for name, text in harness_regenerated(tmp, ref).items():
                if text != read_gz(name):
                    raise SystemExit("%s: %s differs from the real binary's file" % (ref, name))
                print("%s == %s" % (name, os.path.relpath(ref, ROOT)))
        t, q, sw = sw_saturation(tmp)
        write("sw2_targets.txt.gz", "\n".join(t) + "\n")
        write("sw2_queries.txt.gz", "\n".join(q) + "\n")
        write("sw2_expected.tsv.gz", sw)
        print(sw)


if __name__ == "__main__":
    main()
