#!/usr/bin/env python3
"""Regenerates the golden fixtures from the REFERENCE's own compiled code (oracle/_ref/ref_harness,
AVX2 build, cross-checked against the SSE4.1 build oracle/_ref/sse41/ref_harness).  Run in the build
container (needs /root/reference); the fixtures it writes are plain data (inputs + expected outputs).

  python tests/golden/make_golden.py
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from metaeuk_amd import synth  # noqa: E402

REF_AVX2 = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
REF_SSE = os.path.join(ROOT, "oracle", "_ref", "sse41", "ref_harness")
MATDIR = "/root/reference/lib/mmseqs/data"
AA = "ACDEFGHIKLMNPQRSTVWY"


def edge_workload():
    rng = random.Random(20260926)
    rs = lambda n, al=AA: "".join(rng.choice(al) for _ in range(n))
    targets = [rs(rng.randrange(40, 400)) for _ in range(120)]
    targets += ["A" * 80, "QQQQQQQQQQPPPPPPPPPPQQQQQQQQQQPPPPPPPPPPQQQQQQQQQQ", rs(9), rs(10), rs(14), "MKV", "",
                rs(60) + "XXXXXXXXXX" + rs(60), "ACDEFGHIKLMNPQRSTVWY" * 12, rs(1500), rs(30) + "BJOUZ*-" + rs(30)]
    fam = rs(220)
    targets += [fam] + ["".join(c if rng.random() > 0.03 * k else rng.choice(AA) for c in fam) for k in range(1, 12)]
    queries = []
    for t in targets[:60]:
        a = rng.randrange(0, max(1, len(t) - 50))
        queries.append("".join(c if rng.random() > 0.15 else rng.choice(AA) for c in t[a:a + rng.randrange(15, 120)]))
    queries += [rs(9), rs(10), rs(11), rs(15), "A" * 40, "X" * 30, rs(20) + "X" + rs(20), "", "M",
                fam, fam[10:150], fam[::-1], rs(700), "ACDEFGHIKLMNPQRSTVWY" * 3, "acdefghiklmnpqrstvwy" * 2 + fam[:30].lower(),
                fam[:100] + "WWWWWWWWWWWWWWWW" + fam[100:], fam[:100] + fam[130:]]
    return targets, queries


def run_ref(binary, targets, queries, tmp, threads=4):
    tf, qf = os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt")
    open(tf, "w").write("\n".join(targets) + "\n")
    open(qf, "w").write("\n".join(queries) + "\n")
    out = os.path.join(tmp, "out_" + os.path.basename(os.path.dirname(binary)))
    subprocess.check_call([binary, "pipeline", MATDIR, tf, qf, out, "--threads", str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(out, "pref.txt")).read(), open(os.path.join(out, "aln.txt")).read()


def sw_pairs_fixture(tmp):
    rng = random.Random(99)
    rs = lambda n: "".join(rng.choice(AA) for _ in range(n))
    queries, targets, pairs = [], [], []
    for it in range(400):
        a, c = rs(rng.randrange(15, 70)), rs(rng.randrange(15, 70))
        x = rng.choice("WCF") * rng.randrange(5, 35)
        y = rng.choice("DGPN") * rng.randrange(5, 35)
        mode = it % 5
        if mode == 0:
            q, t = a + x + c, a + y + c
        elif mode == 1:
            q, t = a + c, a + y + c
        elif mode == 2:
            q, t = a + x + c, a + c
        elif mode == 3:
            q, t = rs(rng.randrange(15, 300)), rs(rng.randrange(15, 500))
        else:
            base = rs(rng.randrange(100, 900))
            q, t = base, "".join(ch if rng.random() > 0.2 else rng.choice(AA) for ch in base)
        queries.append(q); targets.append(t); pairs.append("%d %d" % (len(queries) - 1, len(targets) - 1))
    tf, qf, pf = os.path.join(tmp, "swt.txt"), os.path.join(tmp, "swq.txt"), os.path.join(tmp, "swp.txt")
    open(tf, "w").write("\n".join(targets) + "\n"); open(qf, "w").write("\n".join(queries) + "\n"); open(pf, "w").write("\n".join(pairs) + "\n")
    outs = []
    for b in (REF_AVX2, REF_SSE):
        o = os.path.join(tmp, "sw_" + str(len(outs)))
        subprocess.check_call([b, "sw", MATDIR, tf, qf, pf, o, "--dbres", "7500000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(open(o).read())
    assert outs[0] == outs[1], "AVX2 and SSE4.1 reference builds disagree on the SW fixture"
    return targets, queries, outs[0]


def orf_contigs():
    """contigs for the extractorfs fixture: synthetic metagenome contigs with IUPAC codes, N runs, lower-case stretches,
    U, gaps and junk characters sprinkled in, plus degenerate ones (shorter than a codon, only stops, no stop at all)"""
    import random
    rng = random.Random(3)
    _, founders = synth.make_targets(120, 5)
    contigs = []
    for c in synth.make_contigs(90, founders, 5):
        s = list("".join("ACGT"[x] for x in c))
        for _ in range(rng.randint(0, 6)):
            s[rng.randrange(len(s))] = rng.choice("NRYKMSWBDHVnU-uX*")
        if rng.random() < 0.3:
            a = rng.randrange(len(s) - 50)
            n = rng.randint(1, 40)
            s[a:a + n] = list("N" * n)
        if rng.random() < 0.3:
            a = rng.randrange(len(s) - 200)
            b = a + rng.randint(1, 200)
            s[a:b] = list("".join(s[a:b]).lower())
        contigs.append("".join(s))
    contigs += ["ACG", "AC", "", "TAATAATAA", "ATGTAA" * 40, "N" * 100, "acgt" * 30, "A" * 46, "TTA" * 15 + "C", "C" + "TAA" * 20 + "GG",
                "ATG" + "GCA" * 30 + "TAA" + "C" * 10, "ACGTNNNNacgtRYKM" * 20]
    return contigs


def write(name, text):
    with gzip.open(os.path.join(HERE, name), "wt", compresslevel=9) as f:
        f.write(text)


def main():
    with tempfile.TemporaryDirectory() as tmp:
        for tag, (targets, queries) in (("small", synth.make_workload(12, 200, seed=7)), ("edge", edge_workload())):
            p1, a1 = run_ref(REF_AVX2, targets, queries, tmp)
            p2, a2 = run_ref(REF_SSE, targets, queries, tmp)
            assert p1 == p2 and a1 == a2, "AVX2 and SSE4.1 reference builds disagree on " + tag
            write(tag + "_targets.txt.gz", "\n".join(targets) + "\n")
            write(tag + "_queries.txt.gz", "\n".join(queries) + "\n")
            write(tag + "_pref.txt.gz", p1)
            write(tag + "_aln.txt.gz", a1)
            print(tag, len(targets), "targets", len(queries), "queries", p1.count("\n") - len(queries), "hits", a1.count("\n") - len(queries), "alignments")
        t, q, sw = sw_pairs_fixture(tmp)
        write("sw_targets.txt.gz", "\n".join(t) + "\n")
        write("sw_queries.txt.gz", "\n".join(q) + "\n")
        write("sw_expected.tsv.gz", sw)
        print("sw pairs", len(q))
        contigs = orf_contigs()
        cf, of = os.path.join(tmp, "contigs.txt"), os.path.join(tmp, "orfs.txt")
        open(cf, "w").write("\n".join(contigs) + "\n")
        outs = []
        for binary in (REF_AVX2, REF_SSE):
            subprocess.check_call([binary, "orfs", cf, of], stdout=subprocess.DEVNULL)
            outs.append(open(of).read())
        assert outs[0] == outs[1], "AVX2 and SSE4.1 reference builds disagree on the ORF fixture"
        write("orf_contigs.txt.gz", "\n".join(contigs) + "\n")
        write("orf_expected.txt.gz", outs[0])
        print("orf contigs", len(contigs), "fragments", outs[0].count("\n") - len(contigs))
        # end to end: contigs -> ORF fragments -> prefilter + align -> exon sets (resultspercontig + collectoptimalset)
        tcodes, founders = synth.make_targets(200, 32)
        e2e_t = [synth.codes_to_str(t) for t in tcodes]
        e2e_c = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(120, founders, 32)]
        outs = []
        for binary in (REF_AVX2, REF_SSE):
            d = os.path.join(tmp, "e2e_" + os.path.basename(os.path.dirname(binary)))
            os.makedirs(d, exist_ok=True)
            open(os.path.join(d, "t.txt"), "w").write("\n".join(e2e_t) + "\n")
            open(os.path.join(d, "c.txt"), "w").write("\n".join(e2e_c) + "\n")
            subprocess.check_call([binary, "orfs", os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt")], stdout=subprocess.DEVNULL)
            prots = [l.rstrip("\n").rsplit("\t", 1)[1] for l in open(os.path.join(d, "orfs.txt")) if not l.startswith(">")]
            open(os.path.join(d, "q.txt"), "w").write("\n".join(prots) + "\n")
            subprocess.check_call([binary, "pipeline", MATDIR, os.path.join(d, "t.txt"), os.path.join(d, "q.txt"), os.path.join(d, "out"), "--threads", "4"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            subprocess.check_call([binary, "exons", os.path.join(d, "t.txt"), os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt"),
                                   os.path.join(d, "out", "aln.txt"), os.path.join(d, "exons.txt")], stdout=subprocess.DEVNULL)
            outs.append(open(os.path.join(d, "exons.txt")).read())
        assert outs[0] == outs[1], "AVX2 and SSE4.1 reference builds disagree on the end-to-end fixture"
        write("e2e_targets.txt.gz", "\n".join(e2e_t) + "\n")
        write("e2e_contigs.txt.gz", "\n".join(e2e_c) + "\n")
        write("e2e_exons_expected.txt.gz", outs[0])
        print("end-to-end", len(e2e_c), "contigs", outs[0].count("\n") - len(e2e_c), "exon lines")
        l2 = int(subprocess.check_output(["getconf", "LEVEL2_CACHE_SIZE"]).decode().strip() or 0)
        open(os.path.join(HERE, "PROVENANCE.txt"), "w").write(
            "generated by tests/golden/make_golden.py with oracle/_ref/ref_harness (reference sources compiled with g++ -mavx2;\n"
            "cross-checked byte-for-byte against the -msse4.1 build).  host L2 (Util::getL2CacheSize) = %d bytes -> BINSIZE 2 for these DB sizes.\n"
            "orf_contigs / orf_expected: `ref_harness orfs` = the reference's Orf.cpp + TranslateNucl.h driven like util/extractorfs.cpp (--translate, predictexons defaults).\n"
            "e2e_*: contigs -> `orfs` -> `pipeline` -> `exons` (resultspercontig joining + the reference's collectoptimalset.cpp / PredictionParser.h): the exon sets of predictexons.\n" % l2)


if __name__ == "__main__":
    main()
