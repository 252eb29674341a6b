"""resultspercontig + collectoptimalset in the product (mk_exons.cpp) without a GPU: fed with the ORF fragments and alignments the C
oracle computes for the end-to-end fixture, it must reproduce the exon sets of the reference's own code (tests/golden/e2e_*)."""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def test_product_exon_sets_from_oracle_alignments(tmp_path):
    from metaeuk_amd import api
    oracle.build()
    t, c = _text("e2e_targets.txt.gz"), _text("e2e_contigs.txt.gz")
    (tmp_path / "t.txt").write_text(t)
    (tmp_path / "c.txt").write_text(c)
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt")], stdout=subprocess.DEVNULL)
    orf_rows, prots, contig = [], [], -1
    for line in open(tmp_path / "orfs.txt"):
        if line.startswith(">"):
            contig = int(line[1:])
            continue
        hdr, prot = line.rstrip("\n").rsplit("\t", 1)
        fields = hdr.split("\t")
        pos = fields[1]
        sign = "+" if "+" in pos else "-"
        frm, ln = pos.split(sign)
        frm, ln = int(frm), int(ln)
        orf_rows.append((contig, frm, frm + ln if sign == "+" else frm - ln, 0, 0, 1 if sign == "-" else 0, 0))
        prots.append(prot)
    (tmp_path / "q.txt").write_text("\n".join(prots) + "\n")
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "--l2", "2097152"], stdout=subprocess.DEVNULL)
    blocks = oracle.read_blocks(str(tmp_path / "out" / "aln.txt"))
    assert len(blocks) == len(orf_rows)
    rows, off = [], [0]
    for b in blocks:
        for line in b.splitlines():
            f = line.split("\t")
            rows.append((int(f[0]), int(f[1]), float(f[2]) + 0.0005, float(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]), int(f[9])))
        off.append(len(rows))
    alns = (api.Alignment * max(len(rows), 1))()
    for i, r in enumerate(rows):
        a = alns[i]
        a.db_key, a.bit_score, a.seq_id, a.evalue = r[0], r[1], r[2], r[3]
        a.q_start, a.q_end, a.q_len, a.db_start, a.db_end, a.db_len = r[4:10]
    targets = t.split("\n")[:-1]
    n_contigs = len(c.split("\n")[:-1])
    pred = api.Predictions.from_arrays(np.array(orf_rows, dtype=api.ORF_DTYPE), n_contigs, alns, np.array(off, dtype=np.uint64), sum(len(x) for x in targets))
    got = "".join(">%d\n%s" % (k, pred.lines(k)) for k in range(n_contigs))
    assert pred.n > 100 and got == _text("e2e_exons_expected.txt.gz")
