#!/usr/bin/env python3
"""Round-2 additions to the golden fixtures, generated in the build container from oracle/_ref/ref_harness (the reference's own
sources, AVX2 and SSE4.1 builds must agree):

  sw2_*   Smith-Waterman pairs at the top of the int16 range of the reference's word pass (simdi16_adds,
          StripedSmithWaterman.cpp:1059): identical and near-identical pairs of 2000 .. 5600 residues, with and without flanks
          (scores up to ~32 000).  A pair that actually REACHES 32767 has no defined reference result: the AVX2 and the
          SSE4.1 build return different garbage (no end position, uninitialised e-value) -- sw_saturation() checks that and
          keeps such pairs out of the fixture.

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


def main():
    with tempfile.TemporaryDirectory() as tmp:
        t, q, sw = sw_saturation(tmp)
        write("sw2_targets.txt.gz", "\n".join(t) + "\n")
        write("sw2_queries.txt.gz", "\n".join(q) + "\n")
        write("sw2_expected.tsv.gz", sw)
        print(sw)


if __name__ == "__main__":
    main()
