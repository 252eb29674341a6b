"""CPU test (no GPU): mk_swap_alignments -- swapresults on arrays -- against the real binary's swapresults output
(tests/golden/prof_aln.txt.gz -> prof_search_res.txt.gz, tests/golden/make_profile_golden.sh)."""
import gzip
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def test_swap_alignments_matches_the_real_swapresults():
    from metaeuk_amd import api
    n_frag = len(_text("e2e_process_orfs.txt.gz").splitlines())
    entries = [ln.split() for ln in open(os.path.join(GOLD, "prof_db.index"))]
    residues = sum(int(e[2]) for e in entries) // 25 - len(entries)          # DBReader::getAminoAcidDBSize of the profile DB
    keys, recs, off = [], [], [0]
    for blk in _text("prof_aln.txt.gz").split(">")[1:]:
        head, _, body = blk.partition("\n")
        keys.append(int(head))
        for line in body.splitlines():
            recs.append(line.split("\t"))
        off.append(len(recs))
    alns = (api.Alignment * len(recs))()
    for a, r in zip(alns, recs):
        a.db_key, a.bit_score, a.seq_id, a.evalue = int(r[0]), int(r[1]), float(r[2]), float(r[3])
        a.q_start, a.q_end, a.q_len, a.db_start, a.db_end, a.db_len = (int(x) for x in r[4:10])
    p = api.default_params()
    p.evalue_thr = 1.7976931348623157e308
    sw, soff = api.swap_alignments(alns, np.array(off, dtype=np.uint64), n_frag, residues, query_keys=keys, params=p)
    got = "".join(">%d\n%s" % (t, api.format_alignments(sw, int(soff[t]), int(soff[t + 1]))) for t in range(n_frag))
    assert got == _text("prof_search_res.txt.gz")
    # a threshold drops records and keeps the lists' order
    p.evalue_thr = 1e-10
    sw2, soff2 = api.swap_alignments(alns, np.array(off, dtype=np.uint64), n_frag, residues, query_keys=keys, params=p)
    kept = [l for l in got.splitlines() if not l.startswith(">") and float(l.split("\t")[3]) <= 1e-10]
    assert 0 < int(soff2[-1]) < int(soff[-1])
    assert api.format_alignments(sw2, 0, int(soff2[-1])).splitlines() == kept
