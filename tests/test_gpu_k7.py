"""GPU parity test of k = 7 -- the k-mer size the reference switches to for target databases of 3.35e9 residues or more
(IndexTable.h:439-449), forced with -k 7 on the e2e fixture: 2-mer x 2-mer x 3-mer k-mer lists in the reference's order, the spaced
seed 11010110011, the threshold 186.15 - 11.22 s and a 20^7-cell table, against the REAL binary's `prefilter -k 7`
(tests/golden/e2e_process_pref_k7.txt.gz, make_process_golden.sh) and, for the alignments, against the oracle."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def test_k7_prefilter_matches_the_real_process(gpu_api, tmp_path, monkeypatch):
    api = gpu_api
    # every front end a k = 7 search can take: the sort-based global path and the miniature of the wide per-query kernel first, the
    # production shape of the wide kernel (the default) last -- its result goes on to the alignment checks below
    gold = _text("e2e_process_pref_k7.txt.gz")
    for path, tiers, lists in (("global", "default", "0"), ("wide", "tiny", "0"), ("wide", "default", "1"), ("wide", "tiny", "1"), ("auto", "default", "0")):
        monkeypatch.setenv("MK_PREFILTER_PATH", path)
        monkeypatch.setenv("MK_PREFILTER_TIERS", tiers)
        monkeypatch.setenv("MK_PREFILTER_K7_LISTS", lists)        # 1: the 7-mers as lists in HBM (what profile queries use); 0: enumerated inside the kernel
        _k7_pass(api, tmp_path, gold, check_more=(path == "auto"))


def _k7_pass(api, tmp_path, gold, check_more):
    targets = _text("e2e_targets.txt.gz").splitlines()
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()]
    params = api.default_params()
    params.sensitivity = 5.7
    params.kmer_size = 7
    params.host_l2_bytes = 2097152
    db = api.TargetDB(targets, params)
    q = api.Queries(frags, api.default_params())          # derived for k = 6: the batch follows the database it meets
    (hits, hoff), (alns, aoff) = api.search(db, q, params)
    pref = "".join(">%d\n%s" % (i, api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode()) for i in range(q.n))
    assert pref == gold
    if not check_more:
        return
    st = api.kernel_stats()
    assert "prefilter_query_wide" in st and st["prefilter_query_wide"]["cells"] > 1e9      # the similar k-mers went through the wide kernel
    assert "kmer7_fill" in st and st["kmer7_fill"]["cells"] > 1e9     # ~2 119 similar k-mers per start
    # the alignments of those hits: the oracle on a prefix (its k = 7 run takes seconds per thousand fragments)
    n = 3000
    (tmp_path / "t.txt").write_text("\n".join(targets) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(frags[:n]) + "\n")
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "-s", "5.7", "-k", "7",
                           "--l2", "2097152"], stdout=subprocess.DEVNULL)
    aln = "".join(">%d\n%s" % (i, api.format_alignments_bulk(alns, int(aoff[i]), int(aoff[i + 1])).decode()) for i in range(n))
    assert aln == open(tmp_path / "out" / "aln.txt").read()
    # a k = 6 search of the same batch afterwards: the thresholds are derived back
    p6 = api.default_params()
    p6.sensitivity = 5.7
    p6.host_l2_bytes = 2097152
    db6 = api.TargetDB(targets, p6)
    hits6, hoff6 = api.prefilter(db6, q, p6)
    pref6 = "".join(">%d\n%s" % (i, api.format_hits_bulk(hits6, int(hoff6[i]), int(hoff6[i + 1])).decode()) for i in range(q.n))
    assert pref6 == _text("e2e_process_pref.txt.gz")


def _write_seq_db(base, seqs):
    from metaeuk_amd import api as A
    A.write_seq_db(base, A.seq_db_image(seqs))


def _read_result_db(base):
    data = open(base, "rb").read()
    out = {}
    for line in open(base + ".index"):
        k, o, l = line.split("\t")
        out[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
    return out


def test_cli_target_splits_and_k7_match_the_real_process(gpu_api, tmp_path):
    """`prefilter --split 3 --split-mode 0 --max-seqs 20` (TARGET_DB_SPLIT: per-range index, BINSIZE and reduced --max-seqs, joined lists)
    and `prefilter -k 7` as commands, against the real binary's DBs; where target splits are not restated they are refused"""
    from metaeuk_amd import build
    targets = _text("e2e_targets.txt.gz").splitlines()
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()]
    _write_seq_db(str(tmp_path / "targets"), targets)
    _write_seq_db(str(tmp_path / "frags"), frags)
    blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
    run = lambda *a: subprocess.run([build.BIN] + [str(x) for x in a], stderr=subprocess.PIPE)
    r = run("prefilter", tmp_path / "frags", tmp_path / "targets", tmp_path / "pref_sp3", "--split", "3", "--split-mode", "0", "-s", "5.7", "--max-seqs", "20",
            "--ref-l2-bytes", "2097152")
    assert r.returncode == 0, r.stderr.decode()
    assert blocks(_read_result_db(str(tmp_path / "pref_sp3"))) == _text("e2e_process_pref_split3_maxseqs20.txt.gz")
    # 2 000 fragments through the command with -k 7
    _write_seq_db(str(tmp_path / "frags2k"), frags[:2000])
    r = run("prefilter", tmp_path / "frags2k", tmp_path / "targets", tmp_path / "pref_k7", "-k", "7", "-s", "5.7", "--ref-l2-bytes", "2097152")
    assert r.returncode == 0, r.stderr.decode()
    expected = _text("e2e_process_pref_k7.txt.gz")
    assert blocks(_read_result_db(str(tmp_path / "pref_k7"))) == expected[:expected.index(">2000\n")]
    # target splits inside the one-pass commands: `search --split 3 --split-mode 0` = the split prefilter + the alignment of the joined lists
    # against the whole database (the oracle's pipeline with --split 3 -- pinned to the reference harness's mergeTargetSplits run in
    # tests/test_oracle_golden.py)
    (tmp_path / "t.txt").write_text("\n".join(targets) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(frags[:2000]) + "\n")
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "osplit"), "-s", "5.7", "--split", "3",
                           "--max-seqs", "20", "--l2", "2097152"], stdout=subprocess.DEVNULL)
    r = run("search", tmp_path / "frags2k", tmp_path / "targets", tmp_path / "aln_sp3", tmp_path / "tmp", "--alignment-mode", "2", "--split", "3", "--split-mode", "0",
            "-s", "5.7", "--max-seqs", "20", "-e", "100", "--min-aln-len", "11", "--ref-l2-bytes", "2097152")
    assert r.returncode == 0, r.stderr.decode()
    assert b"3 target splits" in r.stderr
    got = blocks(_read_result_db(str(tmp_path / "aln_sp3")))
    assert got == open(tmp_path / "osplit" / "aln.txt").read() and got.count("\n") > 2100
    # `align` keeps no index: its target side is the residues alone
    r = run("prefilter", tmp_path / "frags2k", tmp_path / "targets", tmp_path / "pref_sp3b", "--split", "3", "--split-mode", "0", "-s", "5.7", "--max-seqs", "20",
            "--ref-l2-bytes", "2097152")
    assert r.returncode == 0
    r = run("align", tmp_path / "frags2k", tmp_path / "targets", tmp_path / "pref_sp3b", tmp_path / "aln_sp3b", "--alignment-mode", "2", "-e", "100", "--min-aln-len", "11")
    assert r.returncode == 0, r.stderr.decode()
    assert blocks(_read_result_db(str(tmp_path / "aln_sp3b"))) == got


def test_cli_split_chosen_from_the_memory_limit(gpu_api, tmp_path):
    """--split 0 (the default) with --split-memory-limit: the number of target splits follows the reference's own estimate
    (Prefiltering::setupSplit / optimizeSplit / estimateMemoryConsumption, Prefiltering.cpp:273-377,1067-1176) -- for the 100 000-protein
    database of config 2 and 16 threads a limit of 1450 M asks for two splits -- and the result is the reference's two-split result;
    `predictexons` takes the same path, and a limit nothing fits is an error, as in the reference"""
    from metaeuk_amd import api as A, build, synth
    targets, queries = synth.make_workload(12, 100000, seed=11)
    targets, queries = list(targets), list(queries)[:1500]
    A.write_seq_db(str(tmp_path / "T"), A.seq_db_image(targets))
    A.write_seq_db(str(tmp_path / "Q"), A.seq_db_image(queries))
    blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
    flags = ["-s", "5.7", "--alignment-mode", "2", "-e", "100", "--min-aln-len", "11", "--threads", "16"]
    r = subprocess.run([build.BIN, "search", str(tmp_path / "Q"), str(tmp_path / "T"), str(tmp_path / "res"), str(tmp_path / "tmp"), "--split-memory-limit", "1450M"] + flags,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert b"2 target splits [chosen from the memory limit]" in r.stderr, r.stderr.decode()[-400:]
    (tmp_path / "t.txt").write_text("\n".join(targets) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(queries) + "\n")
    if os.path.exists(oracle.REF):
        mat = oracle.write_matrix_files(str(tmp_path / "mat"))
        subprocess.check_call([oracle.REF, "pipeline", mat, str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "o"), "-s", "5.7", "--split", "2", "--threads", "16"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "o"), "-s", "5.7", "--split", "2"], stdout=subprocess.DEVNULL)
    got = blocks(_read_result_db(str(tmp_path / "res")))
    assert got == open(tmp_path / "o" / "aln.txt").read() and got.count("\n") > 3000
    # a generous limit: one split, the ordinary result (differs from the split run: --max-seqs is cut per split there)
    r = subprocess.run([build.BIN, "search", str(tmp_path / "Q"), str(tmp_path / "T"), str(tmp_path / "res1"), str(tmp_path / "tmp"), "--split-memory-limit", "100G"] + flags,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"target splits" not in r.stderr
    # nothing fits 800 M (the 20^6 table and the 3-mer matrix alone need a gigabyte in the reference)
    r = subprocess.run([build.BIN, "search", str(tmp_path / "Q"), str(tmp_path / "T"), str(tmp_path / "res2"), str(tmp_path / "tmp"), "--split-memory-limit", "800M"] + flags,
                       stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"Cannot fit databases" in r.stderr
