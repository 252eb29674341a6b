"""CPU tests: the C oracle against the golden fixtures produced by the reference's own code, and
(when the harness is present) against a live run of the reference harness."""
import gzip
import json
import os
import subprocess

import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lines(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read().split("\n")[:-1]


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


@pytest.mark.parametrize("tag", ["small", "edge"])
@pytest.mark.parametrize("lanes", [(32, 16, 4), (16, 8, 2)])
def test_oracle_pipeline_matches_golden(tag, lanes, tmp_path):
    targets, queries = _lines(tag + "_targets.txt.gz"), _lines(tag + "_queries.txt.gz")
    extra = ["--l2", "2097152", "--lanes-byte", str(lanes[0]), "--lanes-word", str(lanes[1]), "--tantan-lanes", str(lanes[2])]
    oracle.run_pipeline(targets, queries, str(tmp_path), extra=extra)
    assert open(tmp_path / "oracle" / "pref.txt").read() == _text(tag + "_pref.txt.gz")
    assert open(tmp_path / "oracle" / "aln.txt").read() == _text(tag + "_aln.txt.gz")


def test_oracle_sw_matches_golden(tmp_path):
    oracle.build()
    t, q = _lines("sw_targets.txt.gz"), _lines("sw_queries.txt.gz")
    (tmp_path / "t.txt").write_text("\n".join(t) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(q) + "\n")
    (tmp_path / "p.txt").write_text("\n".join("%d %d" % (i, i) for i in range(len(q))) + "\n")
    for lanes in ((32, 16), (16, 8), (1, 1)):
        subprocess.check_call([oracle.CLI, "sw", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "p.txt"),
                               str(tmp_path / "out.tsv"), "--dbres", "7500000", "--lanes-byte", str(lanes[0]), "--lanes-word", str(lanes[1])])
        assert (tmp_path / "out.tsv").read_text() == _text("sw_expected.tsv.gz"), lanes


def test_oracle_sw_matches_golden_at_the_top_of_the_int16_range(tmp_path):
    """pairs scoring 10 000 .. 32 767 (the last one saturates the reference's word pass), every stripe length"""
    oracle.build()
    t, q = _lines("sw2_targets.txt.gz"), _lines("sw2_queries.txt.gz")
    (tmp_path / "t.txt").write_text("\n".join(t) + "\n")
    (tmp_path / "q.txt").write_text("\n".join(q) + "\n")
    (tmp_path / "p.txt").write_text("\n".join("%d %d" % (i, i) for i in range(len(q))) + "\n")
    for lanes in ((32, 16), (16, 8)):
        subprocess.check_call([oracle.CLI, "sw", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "p.txt"),
                               str(tmp_path / "out.tsv"), "--dbres", "7500000", "--lanes-byte", str(lanes[0]), "--lanes-word", str(lanes[1])])
        assert (tmp_path / "out.tsv").read_text() == _text("sw2_expected.tsv.gz"), lanes


def test_oracle_orfs_match_golden(tmp_path):
    """extractorfs --translate (Orf::findAll, TranslateNucl, header format) against the reference's own output"""
    oracle.build()
    (tmp_path / "c.txt").write_text(_text("orf_contigs.txt.gz"))
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "o.txt")], stdout=subprocess.DEVNULL)
    assert (tmp_path / "o.txt").read_text() == _text("orf_expected.txt.gz")


def test_oracle_end_to_end_exon_sets_match_golden(tmp_path):
    """contigs -> ORF fragments -> prefilter + align -> exon sets, every stage by the C oracle, against the exon sets the
    reference's own code produces (extractorfs, prefilter, align, resultspercontig, collectoptimalset)"""
    oracle.build()
    t, c = _text("e2e_targets.txt.gz"), _text("e2e_contigs.txt.gz")
    (tmp_path / "t.txt").write_text(t)
    (tmp_path / "c.txt").write_text(c)
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt")], stdout=subprocess.DEVNULL)
    prots = [l.rstrip("\n").rsplit("\t", 1)[1] for l in open(tmp_path / "orfs.txt") if not l.startswith(">")]
    (tmp_path / "q.txt").write_text("\n".join(prots) + "\n")
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "--l2", "2097152"], stdout=subprocess.DEVNULL)
    subprocess.check_call([oracle.CLI, "exons", str(tmp_path / "t.txt"), str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt"),
                           str(tmp_path / "out" / "aln.txt"), str(tmp_path / "exons.txt")], stdout=subprocess.DEVNULL)
    assert (tmp_path / "exons.txt").read_text() == _text("e2e_exons_expected.txt.gz")


def test_oracle_matches_the_real_process(tmp_path):
    """Process-level goldens: the DBs the REAL `metaeuk predictexons -s 5.7` binary left behind (tests/golden/make_process_golden.sh) --
    aa_6f (+ headers), pref_0, search_res -- against the oracle's stages on the same contigs and targets.  This pins the module DRIVERS
    (Prefiltering::runSplit, Alignment::run) that the harness restates, not only the algorithm classes."""
    oracle.build()
    (tmp_path / "t.txt").write_text(_text("e2e_targets.txt.gz"))
    (tmp_path / "c.txt").write_text(_text("e2e_contigs.txt.gz"))
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt")], stdout=subprocess.DEVNULL)
    frags = [l for l in open(tmp_path / "orfs.txt") if not l.startswith(">")]
    assert "".join(frags) == _text("e2e_process_orfs.txt.gz")
    (tmp_path / "q.txt").write_text("".join(l.rsplit("\t", 1)[1] for l in frags))
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "--l2", "2097152"], stdout=subprocess.DEVNULL)
    assert open(tmp_path / "out" / "pref.txt").read() == _text("e2e_process_pref.txt.gz")
    assert open(tmp_path / "out" / "aln.txt").read() == _text("e2e_process_aln.txt.gz")


def test_oracle_k7_matches_the_real_process(tmp_path):
    """k = 7 (the k-mer size of databases from 3.35e9 residues on, IndexTable.h:439-449; forced with -k 7): the 2-mer x 2-mer x 3-mer
    list generator, the spaced seed 11010110011, the threshold 186.15 - 11.22 s and a 20^7-cell index against the real binary's
    `prefilter -k 7` on the e2e fixture (first 5 000 fragments here; the GPU suite compares all of them)"""
    oracle.build()
    (tmp_path / "t.txt").write_text(_text("e2e_targets.txt.gz"))
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()][:5000]
    (tmp_path / "q.txt").write_text("\n".join(frags) + "\n")
    out = subprocess.check_output([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "-s", "5.7", "-k", "7",
                                   "--l2", "2097152"])
    assert json.loads(out)["kmer_thr"] == 122
    expected = _text("e2e_process_pref_k7.txt.gz")
    expected = expected[:expected.index(">5000\n")]
    assert open(tmp_path / "out" / "pref.txt").read() == expected


def test_oracle_target_splits_match_the_real_process(tmp_path):
    """TARGET_DB_SPLIT (--split 3 --split-mode 0 --max-seqs 20): per-range index, BINSIZE and reduced --max-seqs, joined lists"""
    oracle.build()
    (tmp_path / "t.txt").write_text(_text("e2e_targets.txt.gz"))
    (tmp_path / "q.txt").write_text("".join(l.rsplit("\t", 1)[1] + "\n" for l in _text("e2e_process_orfs.txt.gz").splitlines()))
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "-s", "5.7", "--split", "3",
                           "--max-seqs", "20", "--l2", "2097152"], stdout=subprocess.DEVNULL)
    assert open(tmp_path / "out" / "pref.txt").read() == _text("e2e_process_pref_split3_maxseqs20.txt.gz")


@pytest.mark.skipif(not os.path.exists(oracle.REF) or not os.path.isdir("/root/reference"), reason="reference harness not built here")
def test_reference_harness_reproduces_the_real_process_fixtures(tmp_path):
    """The fixtures that only the real binary had written (profile-target path, k = 7, target splits) come out of oracle/_ref/ref_harness
    as well: the reference's own translation units, compiled in place by oracle/Makefile.ref (no cmake, no stand-in files), driven by
    `profilesearch`, `pipeline -k 7` and `pipeline --split 3`.  With this the oracle's restatement of Sequence::mapProfile, the profile and
    k = 7 k-mer lists, PROFILE_SEQ Smith-Waterman, swapResult and mergeTargetSplits is pinned to reference code compiled HERE."""
    import sys
    sys.path.insert(0, GOLD)
    import make_golden_r2 as mg
    got = mg.harness_regenerated(str(tmp_path), threads=4, k7_fragments=2500)      # (make_golden_r2.py itself compares all 24 084 fragments)
    assert sorted(got) == ["e2e_process_pref_k7.txt.gz", "e2e_process_pref_split3_maxseqs20.txt.gz", "prof_aln.txt.gz", "prof_pref.txt.gz", "prof_search_res.txt.gz"]
    for name, text in got.items():
        expected = _text(name)
        if name == "e2e_process_pref_k7.txt.gz":
            expected = expected[:expected.index(">2500\n")]
        assert text == expected, name
        assert text.count("\n") > 1000, name


def _profile_inputs(tmp_path):
    """the fixtures of the profile-target path as files: the profile DB, the fragments in the order of their data offsets in the
    fragment DB (= the prefilter's target numbering) and their DB keys"""
    import gzip
    (tmp_path / "prof.bin").write_bytes(gzip.open(os.path.join(GOLD, "prof_db.bin.gz"), "rb").read())
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()]
    order = [int(x) for x in _text("prof_frag_order.txt.gz").split()]
    (tmp_path / "frags.txt").write_text("\n".join(frags[k] for k in order) + "\n")
    (tmp_path / "keys.txt").write_text("\n".join(str(k) for k in order) + "\n")
    return os.path.join(GOLD, "prof_db.index")


def test_oracle_profile_search_matches_the_real_process(tmp_path):
    """Profile targets (BASELINE config 4): what the REAL `metaeuk predictexons contigsDB profileDB` left behind
    (tests/golden/make_profile_golden.sh: searchslicedtargetprofile.sh = prefilter with profile queries, align, swapresults, then the
    exon stage) against the oracle's restatement of Sequence::mapProfile, the profile k-mer lists, the PROFILE_SEQ Smith-Waterman
    and swapResult -- every stage byte for byte."""
    oracle.build()
    index = _profile_inputs(tmp_path)
    out = subprocess.check_output([oracle.CLI, "profilesearch", str(tmp_path / "prof.bin"), index, str(tmp_path / "frags.txt"),
                                   str(tmp_path / "out"), "--l2", "2097152", "--keys", str(tmp_path / "keys.txt")])
    info = json.loads(out)
    assert info["kmer_thr"] == 109 and info["eval_thr"] == 24084 and info["masked_residues"] == 23422   # the values the real run logged
    assert open(tmp_path / "out" / "pref.txt").read() == _text("prof_pref.txt.gz")
    assert open(tmp_path / "out" / "aln.txt").read() == _text("prof_aln.txt.gz")
    assert open(tmp_path / "out" / "swapped.txt").read() == _text("prof_search_res.txt.gz")
    # the exon stage on the swapped lists; collectoptimalset's e-values use the profile DB's column count
    (tmp_path / "c.txt").write_text(_text("e2e_contigs.txt.gz"))
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt")], stdout=subprocess.DEVNULL)
    (tmp_path / "t.txt").write_text("")
    subprocess.check_call([oracle.CLI, "exons", str(tmp_path / "t.txt"), str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt"),
                           str(tmp_path / "out" / "swapped.txt"), str(tmp_path / "exons.txt"), "--dbres", str(info["profile_db_residues"])],
                          stdout=subprocess.DEVNULL)
    assert (tmp_path / "exons.txt").read_text() == _text("prof_calls.txt.gz")


@pytest.mark.skipif(not os.path.exists(oracle.REF) or not os.path.isdir("/root/reference"), reason="reference harness not built here")
def test_oracle_matches_live_reference(tmp_path):
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(6, 150, seed=3)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--dump"])
    rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["--dump", "--threads", "2"])
    assert opref == rpref and oaln == raln
    for f in ("masked_targets.txt", "index.txt", "stats.txt"):
        assert open(tmp_path / "oracle" / f).read() == open(tmp_path / "ref" / f).read(), f


@pytest.mark.skipif(not os.path.exists(oracle.REF) or not os.path.isdir("/root/reference"), reason="reference harness not built here")
@pytest.mark.parametrize("max_seqs", [100, 300])
def test_oracle_matches_live_reference_when_the_score_threshold_saturates(tmp_path, max_seqs):
    """more than --max-seqs targets score 255 or more on the diagonal: threshold rescale by the self score + tie order at the cut"""
    import ctypes
    targets, queries = oracle.saturated_threshold_workload()
    l2 = ctypes.CDLL(None).sysconf(191)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(l2 if l2 > 0 else 262144), "--max-seqs", str(max_seqs)])
    rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["--threads", "2", "--max-seqs", str(max_seqs)])
    assert opref == rpref and oaln == raln
    n255 = [sum(1 for l in b.splitlines() if int(l.split("\t")[1]) >= 255) for b in rpref]
    assert max(n255) >= max_seqs or max_seqs == 300, n255        # (130 copies per query: the 100-hit cut saturates, the 300-hit cut does not)


@pytest.mark.skipif(not os.path.exists(oracle.REF) or not os.path.isdir("/root/reference"), reason="reference harness not built here")
def test_oracle_matches_live_reference_on_long_sequences(tmp_path):
    """targets of 40 k and 66 k residues, a 35 k-residue query: wrapped 16-bit positions / diagonals, computeLongScore"""
    import ctypes
    targets, queries = oracle.long_sequence_workload()
    l2 = ctypes.CDLL(None).sysconf(191)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(l2 if l2 > 0 else 262144)])
    rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["--threads", "4"])
    assert opref == rpref and oaln == raln
    long_ids = {i for i, t in enumerate(targets) if len(t) >= 32768}
    assert any(int(l.split("\t")[0]) in long_ids for b in rpref for l in b.splitlines()), "a long target should be hit"
    assert len(rpref[3].splitlines()) >= 2, "the long query should have hits"


@pytest.mark.skipif(not os.path.exists(oracle.REF) or not os.path.isdir("/root/reference"), reason="reference harness not built here")
def test_oracle_matches_live_reference_on_the_database_hits_overflow_path(tmp_path):
    """QueryMatcher::match's overflow path (QueryMatcher.cpp:281-334): a query that gathers >= 2 * max(1e6, #targets) index entries is cut
    into segments with a double-diagonal rule each; from the second overflow on the kept diagonals are merged (later one of equal
    neighbours, order reversed), scored and reduced to the best per target.  70 000 near-copies of one protein: three of the four queries
    overflow 4 ... 13 times.  The device path still refuses such a query; this pins the restatement it will be held to."""
    import random
    oracle.build()
    rng = random.Random(3)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    base = "".join(rng.choice(aa) for _ in range(400))
    mut = lambda s, r: "".join(rng.choice(aa) if rng.random() < r else c for c in s)
    targets = [mut(base, 0.02) for _ in range(70000)]
    queries = [base, mut(base, 0.05), base[:150], "".join(rng.choice(aa) for _ in range(300))]
    rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["--threads", "8"])
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path))      # (both take BINSIZE from this host's L2)
    assert opref == rpref and oaln == raln
    assert sum(len(b.splitlines()) for b in opref) >= 900
    # the model of the device algorithm for this path (segments from list sizes, rule per (target, segment), merges replayed per target,
    # the reference's array order rebuilt from segment and arrival number): the same output
    os.environ["MKO_OVERFLOW_MODEL"] = "1"
    try:
        mpref, maln = oracle.run_pipeline(targets, queries, str(tmp_path / "model"))
    finally:
        del os.environ["MKO_OVERFLOW_MODEL"]
    assert mpref == rpref and maln == raln


def test_matrix_tables_reproduce_reference_matrices(tmp_path):
    """the .out text regenerated from the repository's matrix table parses back to identical numbers"""
    d = oracle.write_matrix_files(str(tmp_path / "mat"))
    txt = open(os.path.join(d, "blosum62.out")).read()
    assert "W" in txt and "10.5040" in txt
    m = oracle.submat(0, 2.0, 0.0)
    assert m.sub[18][18] == 11 and m.sub[0][0] == 4 and m.sub[20][20] == -1   # W/W, A/A, X/X of BLOSUM62 in half bits
