"""GPU parity tests of the profile-target path: profiles as the QUERIES of prefilter / align, fragments as the
indexed targets, swapresults -- through the C ABI, against the fixtures the REAL reference binary left behind
(tests/golden/make_profile_golden.sh) and against the oracle on synthetic profiles."""
import ctypes as C
import gzip
import os
import random
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AA = "ACDEFGHIKLMNPQRSTVWY"


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def _golden_inputs():
    data = gzip.open(os.path.join(GOLD, "prof_db.bin.gz"), "rb").read()
    index = sorted((int(k), int(o), int(l)) for k, o, l in (ln.split() for ln in open(os.path.join(GOLD, "prof_db.index"))))
    keys = [k for k, _, _ in index]
    entries = [data[o:o + l] for _, o, l in index]
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()]
    order = [int(x) for x in _text("prof_frag_order.txt.gz").split()]        # fragment keys in the order the prefilter numbers them
    residues = sum(len(e) for e in entries) // 25 - len(entries)             # DBReader::getAminoAcidDBSize of the profile DB
    return keys, entries, frags, order, residues


def _blocks(keys, texts):
    return "".join(">%d\n%s" % (k, t) for k, t in zip(keys, texts))


def _keyed_alignment_text(api, alns, lo, hi, key_of):
    out = []
    buf = C.create_string_buffer(256)
    for i in range(lo, hi):
        a = api.Alignment.from_buffer_copy(alns[i])
        a.db_key = key_of(int(a.db_key))
        n = api.lib().mk_format_alignment(buf, C.byref(a))
        out.append(buf.raw[:n].decode())
    return "".join(out)


def _profile_params(api, n_frag, n_prof, sens=4.0, evalue=100.0):
    p = api.default_params()
    p.sensitivity = sens
    p.profile_search = 1
    p.max_seqs = max(300, n_frag)                                            # Search.cpp:372
    p.evalue_thr = float("%g" % (evalue * (np.float32(n_frag) / np.float32(n_prof))))   # Search.cpp:366-368 + the parameter string round trip
    p.host_l2_bytes = 2097152
    return p


def test_profile_search_matches_the_real_process(gpu_api):
    """prefilter / align / swapresults of 100 profiles x 24084 fragments: byte-identical to the DBs of the real
    `metaeuk predictexons contigsDB profileDB` run"""
    api = gpu_api
    keys, entries, frags, order, prof_residues = _golden_inputs()
    params = _profile_params(api, len(frags), len(entries))
    assert params.evalue_thr == 24084
    db = api.TargetDB([frags[k] for k in order], params)
    frag_keys = np.array(order, dtype=np.uint32)
    api._chk(api.lib().mk_targetdb_set_keys(db.h, api._p(frag_keys), C.c_uint32(len(order))))
    q = api.Profiles(entries, params)
    hits, hoff = api.prefilter(db, q, params)
    pref = _blocks(keys, [api.format_hits(hits, int(hoff[i]), int(hoff[i + 1]), key_of=lambda t: order[t]) for i in range(q.n)])
    assert pref == _text("prof_pref.txt.gz")
    alns, aoff = api.align(db, q, params)
    aln = _blocks(keys, [_keyed_alignment_text(api, alns, int(aoff[i]), int(aoff[i + 1]), lambda t: order[t]) for i in range(q.n)])
    assert aln == _text("prof_aln.txt.gz")
    swap_params = api.default_params()
    swap_params.evalue_thr = 1.7976931348623157e308                          # swapresults -e DBL_MAX (Search.cpp:378-381)
    sw, soff = api.swap_alignments(alns, aoff, len(order), prof_residues, query_keys=keys, params=swap_params)
    by_key = {}
    for t in range(len(order)):
        by_key[order[t]] = api.format_alignments(sw, int(soff[t]), int(soff[t + 1]))
    assert _blocks(sorted(by_key), [by_key[k] for k in sorted(by_key)]) == _text("prof_search_res.txt.gz")
    # the search as one call gives the same two results
    q2 = api.Profiles(entries, params)
    (h2, ho2), (a2, ao2) = api.search(db, q2, params)
    assert np.array_equal(ho2, hoff) and np.array_equal(h2, hits) and np.array_equal(ao2, aoff)
    assert api.format_alignments_bulk(a2, 0, int(ao2[-1])) == api.format_alignments_bulk(alns, 0, int(aoff[-1]))


def test_profile_arrays_match_the_oracle(gpu_api):
    """Sequence::mapProfile on the device: sorted columns (the exchange network's tie order), alignment profile, k-mer thresholds"""
    api = gpu_api
    keys, entries, frags, order, _ = _golden_inputs()
    params = _profile_params(api, len(frags), len(entries))
    q = api.Profiles(entries[:20], params)
    letters, sorted40, aln32, kthr = q.derived()
    L = oracle.lib()
    at = 0
    for e in entries[:20]:
        n = (len(e) - 1) // 25
        scores, idx, aln, query = oracle.profile_arrays(e, n)
        assert np.array_equal(sorted40[at:at + n, :20], scores) and np.array_equal(sorted40[at:at + n, 20:], idx)
        assert np.array_equal(aln32[at:at + n, :21], aln) and not aln32[at:at + n, 21:].any()
        assert np.array_equal(letters[at:at + n], query)
        exp = np.full(n, -1, dtype=np.int16)
        for i in range(0, n - 9):
            if not any(query[i + d] == 20 for d in (0, 1, 3, 5, 8, 9)):
                exp[i] = 109
        assert np.array_equal(kthr[at:at + n], exp)
        at += n
    assert L is not None


def _synthetic_profile(rng, length, x_rate=0.0):
    """a profile entry with random integer scores in the range result2profile produces; some query letters X"""
    out = bytearray()
    for _ in range(length):
        fav = rng.randrange(20)
        col = [rng.randint(-24, 6) for _ in range(20)]
        col[fav] = rng.randint(12, 40)
        for _ in range(rng.randrange(3)):
            col[rng.randrange(20)] = rng.randint(0, 20)
        letter = 20 if rng.random() < x_rate else fav
        out += bytes((v & 0xFF) for v in col) + bytes([letter, fav, 10, 0, 0])
    return bytes(out) + b"\0", "".join(AA[max(range(20), key=lambda a: ((out[25 * i + a] + 128) % 256))] for i in range(length))


@pytest.mark.parametrize("front", ["wide", "wide-tiny", "global"])
def test_profile_search_synthetic_vs_oracle(gpu_api, tmp_path, monkeypatch, front):
    """synthetic profiles (ties inside columns, X query letters, lengths 12 .. 1500: every SW tile shape incl. row tiles) against
    fragments derived from their consensus: the oracle's stages on the same files"""
    api = gpu_api
    monkeypatch.setenv("MK_PREFILTER_PATH", front.split("-")[0])          # profile queries: the wide per-query kernel on the k-mer lists (default), its miniature, the global path
    monkeypatch.setenv("MK_PREFILTER_TIERS", "tiny" if front.endswith("-tiny") else "default")
    rng = random.Random(5)
    entries, cons = [], []
    for L in [12, 31, 64, 100, 129, 200, 260, 390, 520, 800, 1100, 1500] + [rng.randint(20, 400) for _ in range(28)]:
        e, c = _synthetic_profile(rng, L, x_rate=0.01 if L > 100 else 0.0)
        entries.append(e); cons.append(c)
    # edge profiles: no column at all, fewer columns than the seed spans (no k-mer start), exactly one start, every query letter X
    for L in (0, 1, 9, 10):
        e, c = _synthetic_profile(rng, L)
        entries.append(e); cons.append(c)
    e, c = _synthetic_profile(rng, 60, x_rate=1.0)
    entries.append(e); cons.append(c)
    frags = []
    for c in cons:
        if len(c) < 20:
            frags.append(c + _rand(rng, 30))
            continue
        for _ in range(12):
            a = rng.randrange(0, max(1, len(c) - 15)); b = min(len(c), a + rng.randint(15, 160))
            s = "".join(rng.choice(AA) if rng.random() < 0.15 else ch for ch in c[a:b])
            frags.append(_rand(rng, rng.randint(0, 20)) + s + _rand(rng, rng.randint(0, 20)))
    frags += [_rand(rng, rng.randint(15, 120)) for _ in range(600)]
    frags += ["A" * 40, "XXXXXXXXXXXXXXXXXXXX", cons[5][:50] + "X" * 5 + cons[5][55:100]]
    # round 6: fragments around and beyond the 256 rows of the transposed score pass (mk_sw.hip: swt_kernel) -- 250 ... 420 residues of the long
    # profiles' consensus: waves whose longest fragment does not fit take the classic kernel, 8 jobs in two rounds on the 32-lane tiles
    for c in cons:
        if len(c) >= 500:
            for n in (250, 256, 257, 300, 420):
                a = rng.randrange(0, len(c) - n)
                frags.append("".join(rng.choice(AA) if rng.random() < 0.2 else ch for ch in c[a:a + n]))
    rng.shuffle(frags)
    keys = list(range(len(entries)))
    data = b"".join(entries)
    (tmp_path / "prof.bin").write_bytes(data)
    off, lines = 0, []
    for k, e in enumerate(entries):
        lines.append("%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
    (tmp_path / "prof.index").write_text("".join(lines))
    (tmp_path / "frags.txt").write_text("\n".join(frags) + "\n")
    subprocess.check_call([oracle.CLI, "profilesearch", str(tmp_path / "prof.bin"), str(tmp_path / "prof.index"), str(tmp_path / "frags.txt"),
                           str(tmp_path / "out"), "--l2", "2097152", "-s", "5"], stdout=subprocess.DEVNULL)
    params = _profile_params(api, len(frags), len(entries), sens=5.0)
    db = api.TargetDB(frags, params)
    q = api.Profiles(entries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q, params)
    assert int(hoff[-1]) > 500 and int(aoff[-1]) > 300
    pref = _blocks(keys, [api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) for i in range(q.n)])
    assert pref == open(tmp_path / "out" / "pref.txt").read()
    aln = _blocks(keys, [api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) for i in range(q.n)])
    assert aln == open(tmp_path / "out" / "aln.txt").read()
    if front == "wide":
        # the score pass the other way round (MK_SW_NARROW=0: the profile in the lanes' rows for every wave, waves of 4 jobs on the 32-lane tiles): same bytes
        monkeypatch.setenv("MK_SW_NARROW", "0")
        q2 = api.Profiles(entries, params)
        (h2, ho2), (a2, ao2) = api.search(db, q2, params)
        assert _blocks(keys, [api.format_alignments(a2, int(ao2[i]), int(ao2[i + 1])) for i in range(q2.n)]) == aln
        monkeypatch.delenv("MK_SW_NARROW")
        # ... and the transposed POSITION pass off (MK_SW_TPOS=0: the int32 wavefront with the profile in the rows for every job): same bytes
        monkeypatch.setenv("MK_SW_TPOS", "0")
        api.kernel_stats(reset=True)
        q3 = api.Profiles(entries, params)
        (h3, ho3), (a3, ao3) = api.search(db, q3, params)
        assert _blocks(keys, [api.format_alignments(a3, int(ao3[i]), int(ao3[i + 1])) for i in range(q3.n)]) == aln
        assert not any(k.endswith("t") and k.startswith("sw_pos_rows") for k in api.kernel_stats())
        monkeypatch.delenv("MK_SW_TPOS")
        api.kernel_stats(reset=True)
        q4 = api.Profiles(entries, params)
        api.search(db, q4, params)
        assert any(k.endswith("t") and k.startswith("sw_pos_rows") for k in api.kernel_stats()), sorted(api.kernel_stats())
    residues = sum(len(e) for e in entries) // 25 - len(entries)
    swap_params = api.default_params()
    swap_params.evalue_thr = 1.7976931348623157e308
    sw, soff = api.swap_alignments(alns, aoff, len(frags), residues, params=swap_params)
    swapped = _blocks(range(len(frags)), [api.format_alignments(sw, int(soff[t]), int(soff[t + 1])) for t in range(len(frags))])
    assert swapped == open(tmp_path / "out" / "swapped.txt").read()


def _rand(rng, n):
    return "".join(rng.choice(AA) for _ in range(n))


def test_profile_and_sequence_roles_are_checked(gpu_api):
    api = gpu_api
    keys, entries, frags, order, _ = _golden_inputs()
    params = api.default_params()
    db = api.TargetDB(frags[:200], params)                                   # built for sequence queries
    pp = _profile_params(api, 200, 5)
    q = api.Profiles(entries[:5], pp)
    with pytest.raises(api.MkError):
        api.prefilter(db, q, pp)


def _read_result_db(base):
    data = open(base, "rb").read()
    out = {}
    for line in open(base + ".index"):
        k, o, l = line.split("\t")
        out[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
    return out


def test_cli_profile_workflow_equals_the_real_process(gpu_api, tmp_path):
    """the commands of searchslicedtargetprofile.sh -- prefilter, align (key lists), align (the merged lists again), swapresults -- with
    the parameter strings the real workflow printed, on the profile DB it used and a fragment DB laid out in its data order; then
    `predictexons contigsDB profileDB` as one command against the real run's dp_predictions"""
    from metaeuk_amd import api as A, build
    keys, entries, frags, order, _ = _golden_inputs()
    (tmp_path / "profDB").write_bytes(gzip.open(os.path.join(GOLD, "prof_db.bin.gz"), "rb").read())
    (tmp_path / "profDB.index").write_text(open(os.path.join(GOLD, "prof_db.index")).read())
    (tmp_path / "profDB.dbtype").write_bytes((2).to_bytes(4, "little"))
    A.write_seq_db(str(tmp_path / "aa_6f"), A.seq_db_image(frags, order=order))
    run = lambda *a: subprocess.check_call([build.BIN] + [str(x) for x in a], stderr=subprocess.DEVNULL)
    common = ["--threads", "4", "--compressed", "0", "-v", "3"]
    run("prefilter", tmp_path / "profDB", tmp_path / "aa_6f", tmp_path / "pref", "--sub-mat", "aa:blosum62.out,nucl:nucleotide.out",
        "--seed-sub-mat", "aa:VTML80.out,nucl:nucleotide.out", "-s", "4", "-k", "0", "--target-search-mode", "0", "--k-score", "seq:2147483647,prof:2147483647",
        "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535", "--max-seqs", "24084", "--split", "0", "--split-mode", "2", "--split-memory-limit", "0",
        "-c", "0", "--cov-mode", "0", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "1", "--diag-score", "1", "--exact-kmer-matching", "0",
        "--mask", "1", "--mask-prob", "0.9", "--mask-lower-case", "0", "--mask-n-repeat", "0", "--min-ungapped-score", "15", "--add-self-matches", "0",
        "--spaced-kmer-mode", "1", "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800",
        "--ref-l2-bytes", "2097152", *common)
    blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
    assert blocks(_read_result_db(str(tmp_path / "pref"))) == _text("prof_pref.txt.gz")
    aln_flags = ["--sub-mat", "aa:blosum62.out,nucl:nucleotide.out", "-a", "0", "--alignment-mode", "2", "--wrapped-scoring", "0", "-e", "24084", "--min-seq-id", "0",
                 "--min-aln-len", "11", "--seq-id-mode", "0", "--alt-ali", "0", "-c", "0", "--cov-mode", "0", "--max-seq-len", "65535", "--comp-bias-corr", "1",
                 "--comp-bias-corr-scale", "1", "--max-rejected", "2147483647", "--max-accept", "2147483647", "--add-self-matches", "0", "--db-load-mode", "0",
                 "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800", "--score-bias", "0", "--realign", "0",
                 "--realign-score-bias", "-0.2", "--realign-max-seqs", "2147483647", "--corr-score-weight", "0", "--gap-open", "aa:11,nucl:5",
                 "--gap-extend", "aa:1,nucl:2", "--zdrop", "40", *common]
    run("align", tmp_path / "profDB", tmp_path / "aa_6f", tmp_path / "pref", tmp_path / "aln_it", "--alignment-output-mode", "1", *aln_flags)
    it = _read_result_db(str(tmp_path / "aln_it"))
    expected = {}
    for blk in _text("prof_aln.txt.gz").split(">")[1:]:
        head, _, body = blk.partition("\n")
        expected[int(head)] = "".join(l.split("\t")[0] + "\n" for l in body.splitlines())
    assert it == expected                                                    # the key lists = the first column of the final records
    run("align", tmp_path / "profDB", tmp_path / "aa_6f", tmp_path / "aln_it", tmp_path / "aln", "--alignment-output-mode", "0", *aln_flags)
    assert blocks(_read_result_db(str(tmp_path / "aln"))) == _text("prof_aln.txt.gz")
    run("swapresults", tmp_path / "profDB", tmp_path / "aa_6f", tmp_path / "aln", tmp_path / "search_res", "--sub-mat", "aa:blosum62.out,nucl:nucleotide.out",
        "-e", "1.79769e+308", "--split-memory-limit", "0", "--gap-open", "aa:11,nucl:5", "--gap-extend", "aa:1,nucl:2", "--db-load-mode", "0", *common)
    assert blocks(_read_result_db(str(tmp_path / "search_res"))) == _text("prof_search_res.txt.gz")
    # `search <fragmentDB> <profileDB>`: what the search workflow (Search.cpp:357-399) makes of a profile target -- the same swapped lists
    run("search", tmp_path / "aa_6f", tmp_path / "profDB", tmp_path / "search_res_1", tmp_path / "tmp", "--alignment-mode", "2", "-s", "4", "-e", "100",
        "--min-aln-len", "11", "--ref-l2-bytes", "2097152", "--exhaustive-search", "1")
    assert blocks(_read_result_db(str(tmp_path / "search_res_1"))) == _text("prof_search_res.txt.gz")
    # without --exhaustive-search 1 the reference runs the target-side k-mer search (Search.cpp:251-257): refused, not silently replaced
    r = subprocess.run([build.BIN, "search", str(tmp_path / "aa_6f"), str(tmp_path / "profDB"), str(tmp_path / "search_res_x"), str(tmp_path / "tmp"),
                        "--alignment-mode", "2"], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"--exhaustive-search 1" in r.stderr
    # ... and --exhaustive-search 1 against a sequence DB is refused as well
    r = subprocess.run([build.BIN, "search", str(tmp_path / "aa_6f"), str(tmp_path / "aa_6f"), str(tmp_path / "search_res_y"), str(tmp_path / "tmp"),
                        "--alignment-mode", "2", "--exhaustive-search", "1"], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"sequence target" in r.stderr
    # the whole workflow as one command (default -s 4, -e 100 scaled by 24084 fragments / 100 profiles)
    contigs = _text("e2e_contigs.txt.gz").splitlines()
    A.write_seq_db(str(tmp_path / "contigs"), A.seq_db_image(contigs), dbtype=1)
    run("predictexons", tmp_path / "contigs", tmp_path / "profDB", tmp_path / "calls", tmp_path / "tmp", "--threads", "4", "--ref-l2-bytes", "2097152")
    assert blocks(_read_result_db(str(tmp_path / "calls"))) == _text("prof_calls.txt.gz")
    # the same two commands as THREE workers (RANK / WORLD_SIZE of a launcher, all on this GPU): the workers split the profiles, every one
    # indexes all fragments, worker 0 gathers the alignments (full-precision e-values), swaps and writes -- byte-identical results
    def run_workers(cmd, world):
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MK_SHARD_TIMEOUT_S="300")
            procs.append(subprocess.Popen([build.BIN] + [str(x) for x in cmd], env=env, stderr=subprocess.DEVNULL))
        assert all(p.wait() == 0 for p in procs)
    run_workers(["search", tmp_path / "aa_6f", tmp_path / "profDB", tmp_path / "search_res_w3", tmp_path / "tmp", "--alignment-mode", "2", "-s", "4", "-e", "100",
                 "--min-aln-len", "11", "--ref-l2-bytes", "2097152", "--exhaustive-search", "1", "--gpu", "0"], 3)
    assert blocks(_read_result_db(str(tmp_path / "search_res_w3"))) == _text("prof_search_res.txt.gz")
    assert not [f for f in os.listdir(tmp_path) if f.startswith("search_res_w3_")], "shard files left behind"
    run_workers(["predictexons", tmp_path / "contigs", tmp_path / "profDB", tmp_path / "calls_w2", tmp_path / "tmp", "--threads", "4", "--ref-l2-bytes", "2097152", "--gpu", "0"], 2)
    assert blocks(_read_result_db(str(tmp_path / "calls_w2"))) == _text("prof_calls.txt.gz")
    # the contigs translated in batches of at most 50 000 nucleotides (their fragments still form one indexed side), the profiles in slices of
    # at most 5 000 columns
    env = dict(os.environ, MK_CLI_BATCH_NT="50000", MK_CLI_PROFILE_COLS="5000")
    subprocess.check_call([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "profDB"), str(tmp_path / "calls_b"), str(tmp_path / "tmp"),
                           "--ref-l2-bytes", "2097152"], stderr=subprocess.DEVNULL, env=env)
    assert blocks(_read_result_db(str(tmp_path / "calls_b"))) == _text("prof_calls.txt.gz")


@pytest.mark.skipif(not os.path.exists(oracle.REF), reason="reference harness not on this box")
def test_profile_query_fills_the_database_hits_buffer(gpu_api, tmp_path):
    """QueryMatcher::match's overflow path (QueryMatcher.cpp:281-334) with PROFILE queries: profiles of a protein that 70 000 fragments are
    near-copies of gather tens of millions of index entries where the reference's buffer holds two million -- segments, merges and the
    array order as for sequence queries (tests/test_gpu_parity.py::test_database_hits_overflow_path), the diagonals scored with the
    profile's own columns.  Against the reference's own code (oracle/_ref/ref_harness profilesearch)."""
    import ctypes
    from metaeuk_amd import synth
    api = gpu_api
    rng = random.Random(3)
    AA = synth.AA
    base = "".join(rng.choice(AA) for _ in range(400))
    mut = lambda s, r: "".join(rng.choice(AA) if rng.random() < r else c for c in s)
    frags = [mut(base, 0.02) for _ in range(70000)]
    prots = [base, mut(base, 0.05), "".join(rng.choice(AA) for _ in range(300)), base[:150]]
    entries = synth.make_profiles([np.array([AA.index(c) for c in s], dtype=np.uint8) for s in prots], seed=5)
    p = api.default_params()
    p.sensitivity = 4.0
    p.profile_search = 1
    p.max_seqs = max(300, len(frags))
    p.evalue_thr = 1000.0
    l2 = ctypes.CDLL(None).sysconf(191)
    p.host_l2_bytes = l2 if l2 and l2 > 0 else 262144
    db = api.TargetDB(frags, p)
    q = api.Profiles(entries, p)
    api.kernel_stats(reset=True)
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    assert "host_prefilter_overflow" in api.kernel_stats()
    (tmp_path / "p.bin").write_bytes(b"".join(entries))
    off, lines = 0, []
    for k, e in enumerate(entries):
        lines.append("%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
    (tmp_path / "p.index").write_text("".join(lines))
    (tmp_path / "f.txt").write_text("\n".join(frags) + "\n")
    mat = oracle.write_matrix_files(str(tmp_path / "mat"))
    subprocess.check_call([oracle.REF, "profilesearch", mat, str(tmp_path / "p.bin"), str(tmp_path / "p.index"), str(tmp_path / "f.txt"), str(tmp_path / "o"),
                           "-s", "4", "--eval-abs", "1000", "--threads", "16"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    pref = "".join(">%d\n%s" % (i, api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode()) for i in range(q.n))
    aln = "".join(">%d\n%s" % (i, api.format_alignments_bulk(alns, int(aoff[i]), int(aoff[i + 1])).decode()) for i in range(q.n))
    assert pref == open(tmp_path / "o" / "pref.txt").read()
    assert aln == open(tmp_path / "o" / "aln.txt").read()
    assert int(hoff[-1]) > 100000


@pytest.mark.skipif(not os.path.exists(oracle.REF), reason="reference harness not on this box")
def test_profile_queries_with_k7(gpu_api, tmp_path):
    """profile queries against a fragment set searched with k = 7 (what the reference does from 3.35e9 fragment residues on; forced with -k 7):
    seven one-column steps under the seed 11010110011, threshold 149.15 - 6.85 s (Prefiltering.cpp:1041-1043), a 20^7-cell table.  The e2e
    profile fixture against the reference's own code (ref_harness profilesearch -k 7); the batch is derived for k = 6 first and follows the
    database it meets."""
    import ctypes
    api = gpu_api
    _, entries, frags, _, _ = _golden_inputs()
    frags = frags[:6000]
    p = api.default_params()
    p.sensitivity = 4.0
    p.profile_search = 1
    p.max_seqs = max(300, len(frags))
    p.evalue_thr = 5000.0
    l2 = ctypes.CDLL(None).sysconf(191)
    p.host_l2_bytes = l2 if l2 and l2 > 0 else 262144
    p6 = api.default_params()
    for f in ("sensitivity", "profile_search", "max_seqs", "evalue_thr", "host_l2_bytes"):
        setattr(p6, f, getattr(p, f))
    p.kmer_size = 7
    db = api.TargetDB(frags, p)
    assert db.kmer_size() == 7
    q = api.Profiles(entries, p6)                       # thresholds of k = 6: re-derived when the batch meets the k = 7 database
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    (tmp_path / "p.bin").write_bytes(b"".join(entries))
    off, lines = 0, []
    for k, e in enumerate(entries):
        lines.append("%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
    (tmp_path / "p.index").write_text("".join(lines))
    (tmp_path / "f.txt").write_text("\n".join(frags) + "\n")
    mat = oracle.write_matrix_files(str(tmp_path / "mat"))
    out = subprocess.check_output([oracle.REF, "profilesearch", mat, str(tmp_path / "p.bin"), str(tmp_path / "p.index"), str(tmp_path / "f.txt"), str(tmp_path / "o"),
                                   "-s", "4", "-k", "7", "--eval-abs", "5000", "--threads", "16"], stderr=subprocess.DEVNULL)
    import json
    info = json.loads(out.decode().strip().splitlines()[-1])
    assert info["k"] == 7 and info["kmer_thr"] == 121
    pref = "".join(">%d\n%s" % (i, api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode()) for i in range(q.n))
    aln = "".join(">%d\n%s" % (i, api.format_alignments_bulk(alns, int(aoff[i]), int(aoff[i + 1])).decode()) for i in range(q.n))
    assert pref == open(tmp_path / "o" / "pref.txt").read()
    assert aln == open(tmp_path / "o" / "aln.txt").read()
    assert int(hoff[-1]) > 300
