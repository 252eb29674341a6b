"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.
Integer work => bit-exact."""
import random

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

AA = "ACDEFGHIKLMNPQRSTVWY"


def _rand_seq(rng, n, alphabet=AA):
    return "".join(rng.choice(alphabet) for _ in range(n))


def _mutate(rng, s, rate, indel=0.0):
    out = []
    for ch in s:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out.append(rng.choice(AA))
        out.append(rng.choice(AA) if rng.random() < rate else ch)
    return "".join(out)


def _check_sw(api, targets, queries, pairs, **lanes):
    params = api.default_params()
    if lanes:
        params.simd_lanes_byte = lanes["lb"]
        params.simd_lanes_word = lanes["lw"]
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    qi = np.array([p[0] for p in pairs], dtype=np.uint32)
    ti = np.array([p[1] for p in pairs], dtype=np.uint32)
    got = api.sw_pairs(db, q, qi, ti, with_start=True, params=params)
    bad = []
    for k, (a, b) in enumerate(pairs):
        exp = oracle.sw(queries[a], targets[b], lanes_byte=params.simd_lanes_byte, lanes_word=params.simd_lanes_word)
        assert exp[5] == 0
        if tuple(int(x) for x in got[k]) != exp[:5]:
            bad.append((a, b, tuple(int(x) for x in got[k]), exp[:5]))
    assert not bad, "SW mismatches (q, t, gpu, oracle): %s" % bad[:5]


def test_sw_random_pairs(gpu_api, small_workload):
    targets, queries = small_workload
    rng = random.Random(1)
    pairs = [(rng.randrange(len(queries)), rng.randrange(len(targets))) for _ in range(400)]
    _check_sw(gpu_api, targets, queries, pairs)


def test_sw_related_and_long(gpu_api):
    """related pairs (real alignments incl. gaps), every kernel bucket, word mode and multi-tile queries"""
    rng = random.Random(2)
    targets, queries, pairs = [], [], []
    for L in (15, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 257, 400, 700, 1024, 1025, 1500, 2300):
        base = _rand_seq(rng, L)
        queries.append(base)
        for rate, indel in ((0.0, 0.0), (0.1, 0.0), (0.3, 0.05), (0.5, 0.1)):
            flank_l, flank_r = _rand_seq(rng, rng.randrange(0, 60)), _rand_seq(rng, rng.randrange(0, 60))
            targets.append(flank_l + _mutate(rng, base, rate, indel) + flank_r)
            pairs.append((len(queries) - 1, len(targets) - 1))
        targets.append(_rand_seq(rng, rng.randrange(20, 300)))
        pairs.append((len(queries) - 1, len(targets) - 1))
    _check_sw(gpu_api, targets, queries, pairs)


def test_sw_stripe_semantics(gpu_api):
    """inputs built to cross stripe heads with vertical-then-horizontal gaps; SSE and AVX2 lane counts"""
    rng = random.Random(3)
    targets, queries, pairs = [], [], []
    for rep in range(60):
        a, b, c = _rand_seq(rng, rng.randrange(20, 60)), _rand_seq(rng, rng.randrange(8, 30)), _rand_seq(rng, rng.randrange(20, 60))
        ins = _rand_seq(rng, rng.randrange(8, 30))
        queries.append(a + b + c)
        targets.append(a + ins + c)            # query-gap next to target-gap
        pairs.append((len(queries) - 1, len(targets) - 1))
        queries.append(a + c)
        targets.append(a + ins + c)
        pairs.append((len(queries) - 1, len(targets) - 1))
    _check_sw(gpu_api, targets, queries, pairs)
    _check_sw(gpu_api, targets, queries, pairs, lb=16, lw=8)


def test_full_dp_position_passes_equal_the_early_exit(gpu_api, monkeypatch):
    """MK_SW_EARLY_EXIT=0 -- the position / reverse passes over EVERY column of their jobs, the reference's literal second and third DP
    (StripedSmithWaterman.cpp:400-476) -- against the default (the score pass bounds the end column, both passes leave at the known score:
    DESIGN.md 4.2 item 5): identical mk_alignments, and no forward / position / reverse score mismatch (mk_search would return
    MK_ERR_SW_MISMATCH).  Inputs: the adversarial pairs of test_sw_stripe_semantics as queries x targets, related pairs of every tile
    configuration, and a slice of the headline workload."""
    api = gpu_api
    from metaeuk_amd import synth
    rng = random.Random(3)
    targets, queries = [], []
    for rep in range(60):
        a, b, c = _rand_seq(rng, rng.randrange(20, 60)), _rand_seq(rng, rng.randrange(8, 30)), _rand_seq(rng, rng.randrange(20, 60))
        ins = _rand_seq(rng, rng.randrange(8, 30))
        queries += [a + b + c, a + c]
        targets += [a + ins + c]
    for L in (31, 33, 48, 65, 100, 129, 200, 257, 400, 700, 1025, 1500):
        base = _rand_seq(rng, L)
        queries.append(base)
        for rate, indel in ((0.05, 0.0), (0.2, 0.05), (0.3, 0.1)):
            targets.append(_rand_seq(rng, rng.randrange(0, 60)) + _mutate(rng, base, rate, indel) + _rand_seq(rng, rng.randrange(0, 60)))
        # the maximum early in a long target, and twice in one target: the position pass must report the FIRST column that reaches it
        targets.append(base + _rand_seq(rng, 300))
        targets.append(base + _rand_seq(rng, 40) + base)
    cases = [("adversarial", targets, queries)]
    t2, q2 = synth.make_workload(60, 2000, seed=41)
    cases.append(("headline slice", t2, q2))
    for name, T, Q in cases:
        params = api.default_params()
        db = api.TargetDB(T, params)
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("MK_SW_EARLY_EXIT", flag)
            q = api.Queries(Q, params)
            (hits, hoff), (alns, aoff) = api.search(db, q)          # (raises on MK_ERR_SW_MISMATCH: the mismatch counter is zero)
            got[flag] = (int(hoff[-1]), [api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) for i in range(len(Q))])
            q.close()
        monkeypatch.delenv("MK_SW_EARLY_EXIT")
        assert got["1"] == got["0"], name
        assert sum(len(b) for b in got["1"][1]) > 1000, name
        db.close()


def test_ungapped(gpu_api, small_workload):
    import ctypes as C
    targets, queries = small_workload
    api = gpu_api
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    masked = db.masked()
    rng = random.Random(4)
    qi = np.array([rng.randrange(len(queries)) for _ in range(3000)], dtype=np.uint32)
    ti = np.array([rng.randrange(len(targets)) for _ in range(3000)], dtype=np.uint32)
    diag = np.array([(rng.randrange(-400, 400)) & 0xFFFF for _ in range(3000)], dtype=np.uint16)
    got = api.ungapped(db, q, qi, ti, diag)
    L = oracle.lib()
    kmer_mat, ung_mat = oracle.submat(1, 8.0, -0.2), oracle.submat(0, 2.0, -0.2)
    for k in range(len(qi)):
        qs = oracle.encode(queries[qi[k]])
        bias = np.zeros(len(qs), dtype=np.float32)
        L.mko_comp_bias(C.byref(kmer_mat), qs.ctypes.data_as(C.c_void_p), C.c_int(len(qs)), C.c_float(1.0), bias.ctypes.data_as(C.c_void_p))
        prof = np.zeros(len(qs) * 21, dtype=np.int8)
        L.mko_ungapped_profile(C.byref(ung_mat), qs.ctypes.data_as(C.c_void_p), C.c_int(len(qs)), bias.ctypes.data_as(C.c_void_p), prof.ctypes.data_as(C.c_void_p))
        t = np.ascontiguousarray(masked[int(db.off[ti[k]]):int(db.off[ti[k] + 1])])
        exp = L.mko_ungapped_score(prof.ctypes.data_as(C.c_void_p), C.c_int(len(qs)), t.ctypes.data_as(C.c_void_p), C.c_int(len(t)), C.c_uint16(int(diag[k])))
        assert int(got[k]) == exp, (k, int(qi[k]), int(ti[k]), int(diag[k]), int(got[k]), exp)


@pytest.fixture(params=["fused", "fused-tiny", "wide", "wide-tiny", "global"])
def pf_path(request, monkeypatch):
    """front end of the prefilter: per-query LDS kernels (production tiers / miniature tiers that force the overflow
    hand-over on small inputs), the wide per-query kernel (partitioned hit regions: what k = 7, profile queries and databases beyond
    2^22 targets take; production shape / a miniature whose classes fill up, whose groups are many and whose survivors need sub-classes)
    or the global sort path; the library reads the variables on every call"""
    monkeypatch.setenv("MK_PREFILTER_PATH", request.param.split("-")[0])
    monkeypatch.setenv("MK_PREFILTER_TIERS", "tiny" if request.param.endswith("-tiny") else "default")
    return request.param


def test_pipeline_vs_oracle(gpu_api, small_workload, tmp_path, pf_path):
    targets, queries = small_workload
    api = gpu_api
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    # masked targets + index come from the host-side builder: compare with the oracle's dump
    hits, hoff = api.prefilter(db, q)
    alns, aoff = api.align(db, q)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes)])
    bad = []
    for i in range(len(queries)):
        if api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) != opref[i]:
            bad.append(("pref", i))
        if api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) != oaln[i]:
            bad.append(("aln", i))
    assert not bad, bad[:10]
    assert int(hoff[-1]) > 100 and int(aoff[-1]) > 50


@pytest.mark.parametrize("sens", [4.0, 7.5])
def test_other_sensitivities_vs_oracle(gpu_api, small_workload, tmp_path, pf_path, sens):
    """-s 4 (predictexons' own default) and -s 7.5 (createindex's): other k-mer thresholds -- at 7.5 a k-mer start has several
    times the similar k-mers of 5.7 and the enumerator needs more than one step of first-half candidates -- through mk_search,
    and where the harness exists against the reference's own code as well"""
    targets, queries = small_workload
    queries = queries[:600]
    api = gpu_api
    params = api.default_params()
    params.sensitivity = sens
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q)
    extra = ["--l2", str(params.host_l2_bytes), "-s", str(sens)]
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=extra)
    bad = [i for i in range(len(queries)) if api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) != opref[i]
           or api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) != oaln[i]]
    assert not bad, bad[:10]
    assert int(hoff[-1]) > 50
    if os.path.exists(oracle.REF):
        rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["-s", str(sens), "--threads", "4"])
        assert rpref == opref and raln == oaln


# ---- golden fixtures (produced by the reference's own compiled code, tests/golden/make_golden.py) ----
import gzip
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lines(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read().split("\n")[:-1]


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def _blocks(name):
    return oracle.read_blocks(os.path.join(GOLD, name))


@pytest.mark.parametrize("tag", ["small", "edge"])
def test_pipeline_vs_golden(gpu_api, tag, pf_path):
    api = gpu_api
    targets, queries = _lines(tag + "_targets.txt.gz"), _lines(tag + "_queries.txt.gz")
    params = api.default_params()
    params.host_l2_bytes = 2097152            # tests/golden/PROVENANCE.txt
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    hits, hoff = api.prefilter(db, q)
    alns, aoff = api.align(db, q)
    gpref, galn = _blocks(tag + "_pref.txt.gz"), _blocks(tag + "_aln.txt.gz")
    assert len(gpref) == len(queries)
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == gpref[i], ("pref", i)
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == galn[i], ("aln", i)


def test_index_list_starts_beyond_32_bits(gpu_api, pf_path, monkeypatch):
    """An index of more than 2^32 entries (UniRef50-scale databases): the slots hold 40-bit list starts.  MK_TEST_ENTRY_BASE shifts every
    start by 6e9 entries (and the device pointer back), so the wide arithmetic of every prefilter front end runs on the small fixture."""
    monkeypatch.setenv("MK_TEST_ENTRY_BASE", "6000000000")
    api = gpu_api
    targets, queries = _lines("small_targets.txt.gz"), _lines("small_queries.txt.gz")
    params = api.default_params()
    params.host_l2_bytes = 2097152
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q)
    gpref, galn = _blocks("small_pref.txt.gz"), _blocks("small_aln.txt.gz")
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == gpref[i], ("pref", i)
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == galn[i], ("aln", i)


def test_database_hits_overflow_path(gpu_api, tmp_path, monkeypatch):
    """QueryMatcher::match's overflow path (QueryMatcher.cpp:281-334): 70 000 near-copies of one protein make three of four queries gather
    10 ... 27 million index entries where the reference's buffer holds 2 million -- segments with a double-diagonal rule each, merged as
    the overflow events merge them, ties at the --max-seqs cut in the reference's array order.  Against the oracle's literal restatement
    (itself equal to the reference harness: tests/test_oracle_golden.py)."""
    rng = random.Random(3)
    base = "".join(rng.choice(AA) for _ in range(400))
    mut = lambda s, r: "".join(rng.choice(AA) if rng.random() < r else c for c in s)
    targets = [mut(base, 0.02) for _ in range(70000)]
    queries = [base, mut(base, 0.05), base[:150], "".join(rng.choice(AA) for _ in range(300))]
    api = gpu_api
    params = api.default_params()
    params.host_l2_bytes = 2097152
    db = api.TargetDB(targets, params)
    # (the overflowing queries between ordinary ones: they are isolated piece by piece; base[:40] / base[:52] gather 2-4 million entries -- just beyond the
    #  buffer, and since round 6 well within what the wide per-query kernel holds: it must hand them to the overflow path, not apply the plain rule)
    batch = queries + [mut(base, 0.4), queries[0], base[:40], base[:52], base[:30]]
    opref, oaln = oracle.run_pipeline(targets, batch, str(tmp_path), extra=["--l2", "2097152"])
    for path in ("auto", "wide"):
        monkeypatch.setenv("MK_PREFILTER_PATH", path)
        api.kernel_stats(reset=True)
        q = api.Queries(batch, params)
        (hits, hoff), (alns, aoff) = api.search(db, q)
        for i in range(len(batch)):
            assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], ("pref", path, i)
            assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], ("aln", path, i)
        assert "host_prefilter_overflow" in api.kernel_stats(), path
        q.close()


def test_wide_kernel_query_parts_and_one_class_groups(gpu_api, tmp_path, monkeypatch, capfd):
    """Round 6: a query whose hits do not fit one region of the wide per-query kernel is taken by M workgroups, each keeping the target classes of
    one residue modulo M (every part enumerates everything; arrival ranks count all hits); a part that still fills a class is run again as its two
    halves; and a query with more hits than the rank bits beside a whole target id number takes its classes one by one (the class number leaves
    the sort key: mk_prefilter.hip wide_fwd / wide_inv).  The miniature shape (4 classes of 192 records: parts of two classes) on families of
    near-copies -- 1 000 ... 40 000 index hits per query: one part, two parts, retries, and beyond them the global path -- with and without parts,
    multi-class / one-class groups: always the oracle's bytes (QueryMatcher.cpp:213-346)."""
    rng = random.Random(11)
    mut = lambda s, r: "".join(rng.choice(AA) if rng.random() < r else c for c in s)
    targets, queries = [], []
    for copies in (8, 20, 60, 500):
        base = _rand_seq(rng, 160)
        targets += [mut(base, 0.03) for _ in range(copies)]
        queries += [base[10:40], base[10:60], mut(base[40:140], 0.05), base]
    targets += [_rand_seq(rng, rng.randrange(50, 300)) for _ in range(300)]
    queries += [_rand_seq(rng, 40), ""]
    order = list(range(len(targets)))
    rng.shuffle(order)
    targets = [targets[i] for i in order]
    api = gpu_api
    params = api.default_params()
    params.host_l2_bytes = 2097152
    db = api.TargetDB(targets, params)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", "2097152"])
    monkeypatch.setenv("MK_PREFILTER_PATH", "wide")
    monkeypatch.setenv("MK_PREFILTER_TIERS", "tiny")
    monkeypatch.setenv("MK_PREFILTER_DEBUG", "1")
    for max_logm in ("0", "4"):
        for one_class in ("0", "1"):
            monkeypatch.setenv("MK_PREFILTER_WIDE_MAX_LOGM", max_logm)
            monkeypatch.setenv("MK_TEST_WIDE_ONE_CLASS", one_class)
            for rep in range(2):                                 # (the second call routes the queries with the hits per k-mer the first one has seen)
                q = api.Queries(queries, params)
                (hits, hoff), (alns, aoff) = api.search(db, q)
                for i in range(len(queries)):
                    assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], ("pref", max_logm, one_class, rep, i)
                    assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], ("aln", max_logm, one_class, rep, i)
                q.close()
            err = capfd.readouterr().err
            # (the library's debug lines: with parts there are retry rounds and work lists longer than the queries; without, the global path steps in)
            import re
            halved = sum(int(x) for x in re.findall(r"halved in the kernel (\d+)", err))
            assert (halved > 0) == (max_logm == "4"), (max_logm, err[-1500:])
    assert int(hoff[-1]) > 100
    db.close()


def test_sw_vs_golden(gpu_api):
    """400 adversarial pairs (gap next to gap, poly-residue inserts, long related pairs): coordinates and bit
    scores printed by the reference (AVX2 == SSE4.1) vs the kernel's integers"""
    api = gpu_api
    t, q = _lines("sw_targets.txt.gz"), _lines("sw_queries.txt.gz")
    with gzip.open(os.path.join(GOLD, "sw_expected.tsv.gz"), "rt") as f:
        exp = [l.rstrip("\n").split("\t") for l in f]
    params = api.default_params()
    db = api.TargetDB(t, params)
    qq = api.Queries(q, params)
    idx = np.arange(len(q), dtype=np.uint32)
    got = api.sw_pairs(db, qq, idx, idx, with_start=True)
    for k, e in enumerate(exp):
        # columns: q t key bits seqid evalue qStart qEnd qLen tStart tEnd tLen
        assert (int(got[k][3]), int(got[k][1]), int(got[k][4]), int(got[k][2])) == (int(e[6]), int(e[7]), int(e[9]), int(e[10])), (k, got[k], e)
        assert int(got[k][0]) == oracle.sw(q[k], t[k])[0]


@pytest.mark.parametrize("max_seqs", [100, 300])
def test_saturated_score_threshold(gpu_api, tmp_path, pf_path, max_seqs):
    """more than --max-seqs targets reach 255 on the diagonal: QueryMatcher's threshold saturates and is rescaled by the query's exact
    self score (QueryMatcher.cpp:163-170, :525-544); the hit list, its tie order at the cut and the alignments vs the oracle"""
    api = gpu_api
    targets, queries = oracle.saturated_threshold_workload()
    params = api.default_params()
    params.max_seqs = max_seqs
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    hits, hoff = api.prefilter(db, q, params)
    alns, aoff = api.align(db, q, params)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes), "--max-seqs", str(max_seqs)])
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], i
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], i
    if max_seqs == 100:
        assert max(sum(1 for h in hits[int(hoff[i]):int(hoff[i + 1])] if int(h["pref_score"]) >= 255) for i in range(3)) >= 100


@pytest.mark.parametrize("max_seqs", [300, 20000])
def test_hit_lists_of_every_size_class(gpu_api, tmp_path, max_seqs):
    """the per-query tail of the prefilter (mk_prefilter.hip finish_*: one wave up to 64 survivors (the other tests), one wave with LDS up to 512, a workgroup
    in LDS up to 4096, a workgroup over HBM beyond -- one tile, and several tiles with merge levels in HBM): queries whose motif sits in 40 ...
    13 000 targets, below the clamp value so that nothing is left to the host; with --max-seqs 300 every list but the first is cut in the
    reference's (score, bin, arrival) order, with 20 000 none is"""
    rng = random.Random(17)
    targets, queries = [], []
    for n_copies in (40, 300, 2000, 5000, 13000):
        motif = "".join(rng.choice(AA) for _ in range(26))
        for _ in range(n_copies):
            m = "".join(rng.choice(AA) if rng.random() < 0.12 else c for c in motif)
            targets.append(_rand_seq(rng, rng.randint(5, 40)) + m + _rand_seq(rng, rng.randint(5, 40)))
        queries.append(_rand_seq(rng, 20) + motif + _rand_seq(rng, 25))
    order = list(range(len(targets)))
    rng.shuffle(order)
    targets = [targets[i] for i in order]
    api = gpu_api
    params = api.default_params()
    params.max_seqs = max_seqs
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    api.kernel_stats(reset=True)
    hits, hoff = api.prefilter(db, q, params)
    alns, aoff = api.align(db, q, params)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes), "--max-seqs", str(max_seqs)])
    sizes = [int(hoff[i + 1] - hoff[i]) for i in range(len(queries))]
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], ("pref", i, sizes)
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], ("aln", i)
    assert "host_prefilter_maxseqs" not in api.kernel_stats(), "every list should be finished on the device"
    if max_seqs == 300:
        assert sizes[0] < 300 and sizes[1:] == [300] * 4, sizes
    else:
        assert 64 < sizes[0] < sizes[1] <= 512 < sizes[2] <= 4096 < sizes[3] <= 8192 < sizes[4], sizes


_ORACLE_CACHE = {}


def test_long_sequences(gpu_api, tmp_path, pf_path):
    """targets of 40 k / 66 k residues and a 35 k-residue query: wrapped 16-bit index positions and diagonals, every real diagonal
    scored (UngappedAlignment::computeLongScore), multi-tile alignment of the long query; hit lists and alignments vs the oracle"""
    api = gpu_api
    targets, queries = oracle.long_sequence_workload()
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q)
    key = ("long", int(params.host_l2_bytes))
    if key not in _ORACLE_CACHE:            # (the oracle's plain-C 35 k x 66 k Smith-Waterman takes 20 s: once for the three prefilter paths)
        _ORACLE_CACHE[key] = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes)])
    opref, oaln = _ORACLE_CACHE[key]
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], i
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], i
    long_ids = {i for i, t in enumerate(targets) if len(t) >= 32768}
    assert any(int(h["seq_id"]) in long_ids for h in hits) and int(hoff[4]) - int(hoff[3]) >= 2


def test_sw_golden_top_of_int16_range(gpu_api):
    """identical / near-identical pairs of 2000 .. 6500 residues: scores up to the saturation value 32767 of the reference's word
    pass, multi-tile queries; coordinates printed by the reference vs the kernel's, raw score vs the oracle's"""
    api = gpu_api
    t, q = _lines("sw2_targets.txt.gz"), _lines("sw2_queries.txt.gz")
    with gzip.open(os.path.join(GOLD, "sw2_expected.tsv.gz"), "rt") as f:
        exp = [l.rstrip("\n").split("\t") for l in f]
    params = api.default_params()
    db = api.TargetDB(t, params)
    qq = api.Queries(q, params)
    idx = np.arange(len(q), dtype=np.uint32)
    got = api.sw_pairs(db, qq, idx, idx, with_start=True)
    top = 0
    for k, e in enumerate(exp):
        assert (int(got[k][3]), int(got[k][1]), int(got[k][4]), int(got[k][2])) == (int(e[6]), int(e[7]), int(e[9]), int(e[10])), (k, got[k], e)
        assert int(got[k][0]) == oracle.sw(q[k], t[k])[0]
        top = max(top, int(got[k][0]))
    assert top == 32767, "the fixture should contain a pair that saturates int16"


def test_max_seqs_truncation_order(gpu_api, small_workload, tmp_path, pf_path):
    """--max-seqs smaller than the number of qualifying targets: the cut follows the reference's bin order"""
    targets, queries = small_workload
    api = gpu_api
    params = api.default_params()
    params.max_seqs = 3
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    hits, hoff = api.prefilter(db, q, params)
    alns, aoff = api.align(db, q, params)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes), "--max-seqs", "3"])
    ntrunc = 0
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], ("pref", i)
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], ("aln", i)
        ntrunc += int(hoff[i + 1] - hoff[i]) == 3
    assert ntrunc > 5


def _write_seq_db(base, seqs, keys=None):
    """MMseqs2 sequence DB: data 'SEQ\\n\\0', index 'key\\toffset\\tlength', dbtype 0 (amino acids)"""
    keys = keys if keys is not None else list(range(len(seqs)))
    off = 0
    with open(base, "wb") as d, open(base + ".index", "w") as i:
        for k, s in zip(keys, seqs):
            d.write(s.encode() + b"\n\0")
            i.write("%d\t%d\t%d\n" % (k, off, len(s) + 2))
            off += len(s) + 2
    with open(base + ".dbtype", "wb") as t:
        t.write((0).to_bytes(4, "little"))


def _read_result_db(base):
    data = open(base, "rb").read()
    out = {}
    for line in open(base + ".index"):
        k, o, l = line.split("\t")
        out[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
    return out


def test_cli_prefilter_align_db_roundtrip(gpu_api, tmp_path):
    """the `prefilter` / `align` commands over MMseqs2-format DBs, with the predictexons argv"""
    import subprocess
    from metaeuk_amd import build
    targets, queries = _lines("small_targets.txt.gz"), _lines("small_queries.txt.gz")[:600]
    qkeys = [3 * i + 1 for i in range(len(queries))]          # non-contiguous keys
    _write_seq_db(str(tmp_path / "q"), queries, qkeys)
    _write_seq_db(str(tmp_path / "t"), targets)
    pre = [build.BIN, "prefilter", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "pref_0"),
           "--sub-mat", "aa:blosum62.out,nucl:nucleotide.out", "--seed-sub-mat", "aa:VTML80.out,nucl:nucleotide.out", "-k", "0",
           "--k-score", "seq:2147483647,prof:2147483647", "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535", "--max-seqs", "300",
           "--split", "0", "--split-mode", "2", "-c", "0", "--comp-bias-corr", "1", "--diag-score", "1", "--exact-kmer-matching", "0",
           "--mask", "1", "--mask-prob", "0.9", "--min-ungapped-score", "15", "--spaced-kmer-mode", "1", "-s", "5.7", "--threads", "4",
           "--ref-l2-bytes", "2097152"]
    subprocess.check_call(pre)
    aln = [build.BIN, "align", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "pref_0"), str(tmp_path / "res"),
           "--alignment-mode", "2", "-e", "100", "--min-aln-len", "11", "--min-seq-id", "0", "-c", "0", "--max-rejected", "2147483647",
           "--max-accept", "2147483647", "--alt-ali", "0", "--realign", "0", "--gap-open", "aa:11,nucl:5", "--gap-extend", "aa:1,nucl:2",
           "--comp-bias-corr", "1", "-a", "0", "--threads", "4"]
    subprocess.check_call(aln)
    assert open(tmp_path / "pref_0.dbtype", "rb").read() == (7).to_bytes(4, "little")
    assert open(tmp_path / "res.dbtype", "rb").read() == (5).to_bytes(4, "little")
    pref, res = _read_result_db(str(tmp_path / "pref_0")), _read_result_db(str(tmp_path / "res"))
    gpref, galn = _blocks("small_pref.txt.gz"), _blocks("small_aln.txt.gz")
    assert sorted(pref) == qkeys and sorted(res) == qkeys
    for i, k in enumerate(qkeys):
        assert pref[k] == gpref[i], ("pref", i)
        assert res[k] == galn[i], ("aln", i)
    # unknown flags are a hard error and leave no "done" marker behind
    bad = subprocess.run([build.BIN, "prefilter", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "x"), "--no-such-flag", "1"], capture_output=True)
    assert bad.returncode != 0 and not (tmp_path / "x.dbtype").exists()


def test_device_derive_matches_oracle(gpu_api, small_workload):
    """k-mer thresholds, int8 diagonal correction and int8 SW bias derived by the device kernels (mk_derive.hip)
    against the oracle's comp-bias functions: float/double expression types must match bit for bit."""
    import ctypes as C
    targets, queries = small_workload
    rng = random.Random(5)
    queries = list(queries[:400]) + ["A" * 60, "ACDEFGHIKLX" * 7, "WWWWWWWWWWWWKKKKKKKKKKKK", "MK", "ACDEFGHIKL",
                                      _rand_seq(rng, 700), "X" * 30, "".join(rng.choice("DEKR") for _ in range(90))]
    q = gpu_api.Queries(queries)
    kt, dc, sb = q.derived()
    L = oracle.lib()
    kmer_mat, aln_mat = oracle.submat(1, 8.0, -0.2), oracle.submat(0, 2.0, 0.0)
    p = gpu_api.default_params()
    thr = int(np.float32(163.2) - np.float32(p.sensitivity) * 8.917)
    sp = (0, 1, 3, 5, 8, 9)
    for i, s in enumerate(queries):
        lo, hi = int(q.off[i]), int(q.off[i + 1])
        n = hi - lo
        codes = oracle.encode(s)
        b1 = np.zeros(max(1, n), dtype=np.float32)
        L.mko_comp_bias(C.byref(kmer_mat), codes.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_float(1.0), b1.ctypes.data_as(C.c_void_p))
        cb = np.zeros(max(1, n), dtype=np.int8)
        bias = C.c_int()
        L.mko_sw_query_init(C.byref(aln_mat), codes.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_float(1.0), cb.ctypes.data_as(C.c_void_p), C.byref(bias))
        assert np.array_equal(sb[lo:hi], cb[:n]), "SW bias, query %d" % i
        exp_corr = np.zeros(n, dtype=np.int8)
        exp_kt = np.full(n, -1, dtype=np.int16)
        for k in range(n):
            c = b1[k]
            c = np.float32(np.float64(c / np.float32(4)) - 0.5) if c < 0 else np.float32(np.float64(c / np.float32(4)) + 0.5)
            exp_corr[k] = int(c)                      # C float -> char truncates toward zero
            if k + 10 <= n and not any(codes[k + d] == 20 for d in sp):
                acc = np.float32(0)
                for d in sp:
                    acc = np.float32(acc + b1[k + d])
                r = int(np.float64(acc) - 0.5) if acc < 0 else int(np.float64(acc) + 0.5)
                exp_kt[k] = max(thr - r, 0)
        assert np.array_equal(dc[lo:hi], exp_corr), "diagonal correction, query %d" % i
        assert np.array_equal(kt[lo:hi], exp_kt), "k-mer threshold, query %d" % i


def test_search_equals_prefilter_then_align(gpu_api, pf_path):
    """mk_search (the two stages pipelined on two streams, chunk by chunk) returns exactly what the two module calls return"""
    from metaeuk_amd import synth
    api = gpu_api
    targets, queries = synth.make_workload(40, 1500, seed=23)
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q1 = api.Queries(queries, params)
    hits1, hoff1 = api.prefilter(db, q1)
    alns1, aoff1 = api.align(db, q1)
    q2 = api.Queries(queries, params)
    (hits2, hoff2), (alns2, aoff2) = api.search(db, q2)
    assert np.array_equal(np.asarray(hoff1), np.asarray(hoff2)) and np.array_equal(np.asarray(aoff1), np.asarray(aoff2))
    assert hits1.tobytes() == hits2.tobytes()
    n = int(aoff1[-1])
    assert n > 100
    assert api.format_alignments(alns1, 0, n) == api.format_alignments(alns2, 0, n)


def test_switches_are_read_only_under_mk_debug(gpu_api, small_workload, monkeypatch):
    """experiment / test switches (MK_PREFILTER_PATH ...) change tiers and paths: the library reads them only when MK_DEBUG=1 is set as well, so that
    a stray variable cannot reach a production run (metaeuk_amd/csrc/mk_host.cpp: knob).  A value the library would refuse shows which way it went."""
    targets, queries = small_workload
    api = gpu_api
    params = api.default_params()
    db = api.TargetDB(targets, params)
    monkeypatch.setenv("MK_PREFILTER_PATH", "no-such-path")
    monkeypatch.setenv("MK_DEBUG", "0")
    q = api.Queries(queries[:200], params)
    hits, hoff = api.prefilter(db, q)                     # ignored: runs as usual
    assert int(hoff[-1]) > 0
    monkeypatch.setenv("MK_DEBUG", "1")
    with pytest.raises(api.MkError) as e:
        api.prefilter(db, api.Queries(queries[:200], params))
    assert "MK_PREFILTER_PATH" in str(e.value)
    monkeypatch.setenv("MK_PREFILTER_PATH", "auto")


def test_two_databases_interleaved_between_align_and_search(gpu_api):
    """The alignment stage keeps its e-value / bit-score tables on the device per scratch lane: worker 0 of mk_search and the caller of
    mk_align share lane 0.  mk_align(A), mk_search(B), mk_align(A) again must not align A against B's tables: the
    databases differ in size, hence in every e-value."""
    from metaeuk_amd import synth
    api = gpu_api
    tA, queries = synth.make_workload(20, 400, seed=31)
    tB, _ = synth.make_workload(20, 2500, seed=32)
    tB = list(tA)[:150] + list(tB)                          # shares homologs with the queries, five times the residues
    params = api.default_params()
    dbA, dbB = api.TargetDB(list(tA), params), api.TargetDB(tB, params)
    def run_align(db):
        q = api.Queries(queries, params)
        api.prefilter(db, q)
        alns, aoff = api.align(db, q)
        return api.format_alignments(alns, 0, int(aoff[-1]))
    def run_search(db):
        q = api.Queries(queries, params)
        (_, _), (alns, aoff) = api.search(db, q)
        return api.format_alignments(alns, 0, int(aoff[-1]))
    a1 = run_align(dbA)
    b1 = run_search(dbB)
    a2 = run_align(dbA)
    b2 = run_align(dbB)
    a3 = run_search(dbA)
    assert len(a1) > 2000 and a1 != b1
    assert a1 == a2 == a3 and b1 == b2


def test_long_and_degenerate_inputs(gpu_api, tmp_path, pf_path):
    """queries beyond the largest SW tile (row tiles with an HBM border), long targets, query == target (prefilter scores
    above 255), queries without any k-mer, and an empty batch"""
    api = gpu_api
    rng = random.Random(17)
    base = [_rand_seq(rng, n) for n in (1500, 1100, 700, 3000, 90, 45)]
    targets = base + [_mutate(rng, base[0], 0.15, 0.02), _mutate(rng, base[1][200:900], 0.1), _mutate(rng, base[3], 0.2, 0.03)]
    targets += [_rand_seq(rng, rng.randint(40, 400)) for _ in range(120)]
    queries = [base[0], base[1], _mutate(rng, base[0], 0.1, 0.01), base[3][500:2100], _mutate(rng, base[2], 0.05), base[4], base[5],
               "ACDEFGHIK", "X" * 40, "", _rand_seq(rng, 10), _mutate(rng, base[1], 0.3, 0.05)]
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q)
    opref, oaln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes)])
    for i in range(len(queries)):
        assert api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])) == opref[i], ("pref", i)
        assert api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1])) == oaln[i], ("aln", i)
    assert int(aoff[1]) - int(aoff[0]) >= 2            # the 1500-residue query aligns to itself and to its mutated copy
    # empty batch
    q0 = api.Queries([], params)
    (h0, ho0), (a0, ao0) = api.search(db, q0)
    assert len(ho0) == 1 and int(ho0[0]) == 0 and len(ao0) == 1 and int(ao0[0]) == 0


def _orf_lines(o, contig_count):
    """Orfs handle -> the `ref_harness orfs` text format"""
    out, k = [], 0
    for c in range(contig_count):
        out.append(">%d" % c)
        while k < o.n and int(o.orfs[k]["contig"]) == c:
            out.append(o.header(k) + "\t" + o.protein(k))
            k += 1
    assert k == o.n
    return "\n".join(out) + "\n"


def test_extract_orfs_vs_golden(gpu_api):
    """extractorfs --translate on the device against the reference's own output (IUPAC codes, N runs, lower case, U, junk
    characters, contigs shorter than a codon, stop-only and stop-free contigs)"""
    contigs = _lines("orf_contigs.txt.gz")
    with gzip.open(os.path.join(GOLD, "orf_expected.txt.gz"), "rt") as f:
        expected = f.read()
    o = gpu_api.Orfs(contigs)
    assert _orf_lines(o, len(contigs)) == expected


def test_orfs_feed_the_search(gpu_api, tmp_path):
    """contigs -> ORF fragments -> query batch without leaving HBM: same fragments as the oracle's extractorfs, and the
    search over them equals the search over the same proteins handed in as strings"""
    from metaeuk_amd import synth
    import subprocess
    api = gpu_api
    targets, founders = synth.make_targets(300, 9)
    tstr = [synth.codes_to_str(t) for t in targets]
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(25, founders, 9)]
    o = api.Orfs(contigs)
    (tmp_path / "c.txt").write_text("\n".join(contigs) + "\n")
    oracle.build()
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "o.txt")], stdout=subprocess.DEVNULL)
    assert _orf_lines(o, len(contigs)) == (tmp_path / "o.txt").read_text()
    params = api.default_params()
    db = api.TargetDB(tstr, params)
    q1 = o.queries(params)
    (h1, ho1), (a1, ao1) = api.search(db, q1)
    q2 = api.Queries([o.protein(k) for k in range(o.n)], params)
    (h2, ho2), (a2, ao2) = api.search(db, q2)
    assert np.array_equal(np.asarray(ho1), np.asarray(ho2)) and h1.tobytes() == h2.tobytes()
    assert np.array_equal(np.asarray(ao1), np.asarray(ao2))
    n = int(ao1[-1])
    assert n > 20 and api.format_alignments(a1, 0, n) == api.format_alignments(a2, 0, n)


def _prediction_text(pred, n_contigs):
    return "".join(">%d\n%s" % (c, pred.lines(c)) for c in range(n_contigs))


def test_predict_exons_end_to_end_vs_golden(gpu_api):
    """contigs -> mk_extract_orfs -> mk_search -> mk_predict_exons: the product chain, against the exon sets the reference's
    own extractorfs / prefilter / align / resultspercontig / collectoptimalset code produced (tests/golden/PROVENANCE.txt)"""
    api = gpu_api
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    params = api.default_params()
    params.host_l2_bytes = 2097152
    db = api.TargetDB(targets, params)
    o = api.Orfs(contigs)
    q = o.queries(params)
    api.search(db, q)
    pred = api.Predictions(db, o, q)
    exp = _text("e2e_exons_expected.txt.gz")
    assert pred.n > 100 and _prediction_text(pred, len(contigs)) == exp
    # the arrays say what the text says
    P, E = pred.predictions, pred.exons
    assert int(pred.contig_off[-1]) == pred.n and int(P["n_exons"].sum()) == len(E) == exp.count("\n") - len(contigs)
    assert (P["n_exons"] > 1).any() and (P["strand"] == -1).any()
    assert (P["low_coord"] <= P["high_coord"]).all()
    first = E[P["first_exon"]]
    assert (np.where(P["strand"] == 1, first["contig_start"], 0) >= 0).all()


def test_predict_exons_vs_oracle_on_synthetic_contigs(gpu_api, tmp_path):
    """same chain on a seeded workload with homolog-bearing contigs on both strands, checked against the oracle's exon stage fed
    with the oracle's own ORFs and alignments; plus non-default exon parameters and the error path"""
    from metaeuk_amd import synth
    import subprocess
    api = gpu_api
    targets, founders = synth.make_targets(400, 21)
    tstr = [synth.codes_to_str(t) for t in targets]
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(60, founders, 21)]
    params = api.default_params()
    db = api.TargetDB(tstr, params)
    o = api.Orfs(contigs)
    q = o.queries(params)
    with pytest.raises(api.MkError):
        api.Predictions(db, o, q)                                   # no alignments in the batch yet
    api.search(db, q)
    pred = api.Predictions(db, o, q)
    oracle.build()
    (tmp_path / "t.txt").write_text("\n".join(tstr) + "\n")
    (tmp_path / "c.txt").write_text("\n".join(contigs) + "\n")
    subprocess.check_call([oracle.CLI, "orfs", str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt")], stdout=subprocess.DEVNULL)
    prots = [l.rstrip("\n").rsplit("\t", 1)[1] for l in open(tmp_path / "orfs.txt") if not l.startswith(">")]
    (tmp_path / "q.txt").write_text("\n".join(prots) + "\n")
    subprocess.check_call([oracle.CLI, "pipeline", str(tmp_path / "t.txt"), str(tmp_path / "q.txt"), str(tmp_path / "out"), "--l2", str(params.host_l2_bytes)],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call([oracle.CLI, "exons", str(tmp_path / "t.txt"), str(tmp_path / "c.txt"), str(tmp_path / "orfs.txt"),
                           str(tmp_path / "out" / "aln.txt"), str(tmp_path / "exons.txt")], stdout=subprocess.DEVNULL)
    exp = (tmp_path / "exons.txt").read_text()
    assert pred.n > 20 and _prediction_text(pred, len(contigs)) == exp
    # a second set per target and strand, looser coverage: more predictions, the default ones still among them
    xp = api.default_exon_params()
    xp.max_exon_sets, xp.target_cov_thr = 2, 0.3
    pred2 = api.Predictions(db, o, q, xp)
    assert pred2.n >= pred.n
    # an empty contig set is a valid, empty answer
    o0 = api.Orfs([])
    q0 = o0.queries(params)
    api.search(db, q0)
    assert api.Predictions(db, o0, q0).n == 0


def test_cli_extractorfs_db_roundtrip(gpu_api, tmp_path):
    """`metaeuk-amd extractorfs` over MMseqs2 DBs: renumbered ORF keys, header DB, nucleotide fragments, and the translated
    sibling DB that lets predictexons.sh skip translatenucs -- against the reference's golden output"""
    import subprocess
    from metaeuk_amd import build
    contigs = [c for c in _lines("orf_contigs.txt.gz")]
    keys = [3 * i + 5 for i in range(len(contigs))]                # DB keys need not be 0..n-1: the headers carry them
    _write_seq_db(str(tmp_path / "contigs"), contigs, keys)
    subprocess.check_call([build.BIN, "extractorfs", str(tmp_path / "contigs"), str(tmp_path / "nucl_6f"), "--min-length", "15",
                           "--orf-start-mode", "1", "--contig-start-mode", "2", "--contig-end-mode", "2", "--translation-table", "1",
                           "--threads", "4", "--aa-sibling", "aa_6f"])
    assert open(tmp_path / "nucl_6f.dbtype", "rb").read() == (1).to_bytes(4, "little")
    assert open(tmp_path / "aa_6f.dbtype", "rb").read() == (0).to_bytes(4, "little")
    hdr, nuc, aa = _read_result_db(str(tmp_path / "nucl_6f_h")), _read_result_db(str(tmp_path / "nucl_6f")), _read_result_db(str(tmp_path / "aa_6f"))
    exp = []                                                        # (header with the contig's DB key, protein) in output order
    with gzip.open(os.path.join(GOLD, "orf_expected.txt.gz"), "rt") as f:
        for line in f:
            if line.startswith(">"):
                continue
            h, prot = line.rstrip("\n").rsplit("\t", 1)
            c, rest = h.split("\t", 1)
            exp.append(("%d\t%s" % (keys[int(c)], rest), prot))
    assert sorted(hdr) == list(range(len(exp))) and sorted(aa) == list(range(len(exp))) and sorted(nuc) == list(range(len(exp)))
    codon = {a + b + c: "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"[16 * i + 4 * j + k]
             for i, a in enumerate("TCAG") for j, b in enumerate("TCAG") for k, c in enumerate("TCAG")}
    for k, (h, prot) in enumerate(exp):
        assert hdr[k] == h + "\n", k
        assert aa[k] == prot + "\n", k
        n = nuc[k].rstrip("\n")
        assert len(n) == 3 * len(prot), k
        if set(n) <= set("ACGT"):                                   # unambiguous upper-case fragments translate with the plain table
            assert "".join(codon[n[i:i + 3]] for i in range(0, len(n), 3)) == prot, k
    bad = subprocess.run([build.BIN, "extractorfs", str(tmp_path / "contigs"), str(tmp_path / "x"), "--orf-start-mode", "0"], capture_output=True)
    assert bad.returncode != 0 and not (tmp_path / "x.dbtype").exists()


def test_cli_predictexons_db_in_db_out(gpu_api, tmp_path):
    """`metaeuk-amd predictexons contigsDB targetsDB outDB tmp`: the whole workflow in one process over MMseqs2 DBs whose data
    order differs from their key order and whose keys are not 0..n-1, against the reference-made exon sets"""
    import subprocess
    from metaeuk_amd import build
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    rs = np.random.RandomState(5)
    tp, cp = rs.permutation(len(targets)), rs.permutation(len(contigs))
    tkey, ckey = (lambda i: 7 * i + 3), (lambda i: 2 * i + 10)
    _write_seq_db(str(tmp_path / "targets"), [targets[i] for i in tp], [tkey(int(i)) for i in tp])
    _write_seq_db(str(tmp_path / "contigs"), [contigs[i] for i in cp], [ckey(int(i)) for i in cp])
    for base in ("targets", "contigs"):                            # a real .index is sorted by key
        rows = sorted(open(tmp_path / (base + ".index")).read().splitlines(), key=lambda r: int(r.split("\t")[0]))
        (tmp_path / (base + ".index")).write_text("\n".join(rows) + "\n")
    (tmp_path / "contigs.dbtype").write_bytes((1).to_bytes(4, "little"))
    cmd = [build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls"), str(tmp_path / "tmp"),
           "-s", "5.7", "--ref-l2-bytes", "2097152", "--metaeuk-eval", "0.001", "--metaeuk-tcov", "0.5", "--max-intron", "10000", "--min-intron", "15",
           "--min-exon-aa", "11", "--max-overlap", "10", "--max-exon-sets", "1", "--set-gap-open", "-1", "--set-gap-extend", "-1", "--min-length", "15",
           "--threads", "4", "--remove-tmp-files", "1"]
    subprocess.check_call(cmd)
    assert open(tmp_path / "calls.dbtype", "rb").read() == (12).to_bytes(4, "little")
    got = _read_result_db(str(tmp_path / "calls"))
    exp, cur = {}, None
    for line in _text("e2e_exons_expected.txt.gz").splitlines(True):
        if line.startswith(">"):
            cur = ckey(int(line[1:]))
            exp[cur] = ""
        else:
            t, rest = line.split("\t", 1)
            exp[cur] += "%d\t%s" % (tkey(int(t)), rest)
    assert sorted(got) == sorted(exp) and sum(1 for v in exp.values() if v) > 50
    assert got == exp
    # the finished output is the workflow's "done" marker: a second run refuses to overwrite it (predictexons.sh:32)
    assert subprocess.call(cmd, stderr=subprocess.DEVNULL) != 0
    # a flag value this build does not implement is an error, not a silent default
    assert subprocess.call(cmd[:4] + [str(tmp_path / "calls2"), cmd[5], "--translation-table", "4"], stderr=subprocess.DEVNULL) != 0
    assert not os.path.exists(tmp_path / "calls2.dbtype")


def test_targetdb_from_index_db(gpu_api, tmp_path):
    """a target database opened from a createindex DB -- written by this library and, where the harness exists, by the
    reference's own createIndexFile -- searches exactly like the one built from the sequences"""
    import subprocess
    from metaeuk_amd import synth
    api = gpu_api
    targets, queries = synth.make_workload(25, 300, seed=8)
    targets, queries = list(targets), list(queries)
    n = len(targets)
    rs = random.Random(2)
    order = list(range(n))
    rs.shuffle(order)
    keys = [3 * i + 1 for i in range(n)]
    image = api.seq_db_image(targets, keys, order)
    params = api.default_params()
    api.index_write(str(tmp_path / "own.idx"), image, params)
    idx = [str(tmp_path / "own.idx")]
    if os.path.exists(oracle.REF):
        api.write_seq_db(str(tmp_path / "T"), image)
        subprocess.check_call([oracle.REF, "createindex", oracle.write_matrix_files(str(tmp_path / "mat")), str(tmp_path / "T"), "-s", "5.7"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        idx.append(str(tmp_path / "T.idx"))
    db0 = api.TargetDB(targets, params)                             # ids = key order = what the index numbers
    q0 = api.Queries(queries, params)
    (h0, ho0), (a0, ao0) = api.search(db0, q0)
    na = int(ao0[-1])
    assert len(h0) > 300 and na > 50
    for path in idx:
        db = api.TargetDB.from_index(path, params)
        assert db.n == n and list(db.keys) == keys
        q = api.Queries(queries, params)
        (h, ho), (a, ao) = api.search(db, q)
        assert np.array_equal(np.asarray(ho), np.asarray(ho0)) and h.tobytes() == h0.tobytes(), path
        assert np.array_equal(np.asarray(ao), np.asarray(ao0)) and api.format_alignments(a, 0, na) == api.format_alignments(a0, 0, na), path


def test_cli_createindex_then_predictexons(gpu_api, tmp_path):
    """`metaeuk-amd createindex` writes <targets>.idx; `predictexons` picks it up (and ignores it under MMSEQS_IGNORE_INDEX) with
    the same called exons"""
    import subprocess
    from metaeuk_amd import build
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    _write_seq_db(str(tmp_path / "targets"), targets)
    _write_seq_db(str(tmp_path / "contigs"), contigs)
    (tmp_path / "contigs.dbtype").write_bytes((1).to_bytes(4, "little"))
    subprocess.check_call([build.BIN, "createindex", str(tmp_path / "targets"), str(tmp_path / "tmp"), "-s", "5.7", "--threads", "4"])
    assert open(tmp_path / "targets.idx.dbtype", "rb").read() == (9).to_bytes(4, "little")
    common = ["-s", "5.7", "--ref-l2-bytes", "2097152", "--threads", "4"]
    log = subprocess.run([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls_idx"), str(tmp_path / "tmp")] + common,
                         check=True, stderr=subprocess.PIPE).stderr.decode()
    env = dict(os.environ, MMSEQS_IGNORE_INDEX="1")
    subprocess.check_call([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls_seq"), str(tmp_path / "tmp")] + common, env=env)
    got_idx, got_seq = _read_result_db(str(tmp_path / "calls_idx")), _read_result_db(str(tmp_path / "calls_seq"))
    exp = _text("e2e_exons_expected.txt.gz")
    assert got_idx == got_seq
    assert "".join(">%d\n%s" % (c, got_idx[c]) for c in range(len(contigs))) == exp
    # the index DB itself as the target argument (what the search workflow passes on, blastp.sh)
    subprocess.check_call([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets.idx"), str(tmp_path / "calls_idx2"), str(tmp_path / "tmp")] + common)
    assert _read_result_db(str(tmp_path / "calls_idx2")) == got_idx


def test_cli_two_workers_share_the_queries(gpu_api, tmp_path):
    """`prefilter` and `align` as two workers (RANK / WORLD_SIZE of a launcher, both on this GPU): each takes its residue-balanced
    query range and writes its shard, worker 0 merges -- same DBs as the single-process run, no collective anywhere"""
    import subprocess
    from metaeuk_amd import build, shard, synth
    targets, queries = synth.make_workload(6, 300, seed=12)
    targets, queries = list(targets), list(queries)
    _write_seq_db(str(tmp_path / "q"), queries, [2 * i + 7 for i in range(len(queries))])
    _write_seq_db(str(tmp_path / "t"), targets)
    flags = ["-s", "5.7", "--ref-l2-bytes", "2097152", "--threads", "2", "--gpu", "0"]
    aflags = ["-e", "100", "--min-aln-len", "11", "--alignment-mode", "2"]
    def run(cmd, out, world):
        procs = []
        for r in range(world):
            env = dict(os.environ)
            if world > 1:
                env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r))
            tail = [str(tmp_path / "tmp")] if cmd[0] == "search" else []
            procs.append(subprocess.Popen([build.BIN] + cmd + [out] + tail + flags + (aflags if cmd[0] == "search" else []), env=env, stderr=subprocess.DEVNULL))
        assert all(p.wait() == 0 for p in procs)
    run(["prefilter", str(tmp_path / "q"), str(tmp_path / "t")], str(tmp_path / "pref1"), 1)
    run(["prefilter", str(tmp_path / "q"), str(tmp_path / "t")], str(tmp_path / "pref2"), 2)
    p1, p2 = shard.read_result_db(str(tmp_path / "pref1")), shard.read_result_db(str(tmp_path / "pref2"))
    assert p1 == p2 and len(p1) == len(queries) and sum(len(v) for v in p1.values()) > 1000
    assert not os.path.exists(tmp_path / "pref2_0") and not os.path.exists(tmp_path / "pref2_1.dbtype")
    run(["align", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "pref1")] + aflags, str(tmp_path / "res1"), 1)
    run(["align", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "pref2")] + aflags, str(tmp_path / "res2"), 2)
    a1, a2 = shard.read_result_db(str(tmp_path / "res1")), shard.read_result_db(str(tmp_path / "res2"))
    assert a1 == a2 and len(a1) == len(queries) and sum(len(v) for v in a1.values()) > 300
    assert open(tmp_path / "res2.dbtype", "rb").read() == (5).to_bytes(4, "little")
    # `search` = the two modules as one pass, nothing written in between; alone and as two workers
    run(["search", str(tmp_path / "q"), str(tmp_path / "t")], str(tmp_path / "sres1"), 1)
    run(["search", str(tmp_path / "q"), str(tmp_path / "t")], str(tmp_path / "sres2"), 2)
    assert shard.read_result_db(str(tmp_path / "sres1")) == a1 and shard.read_result_db(str(tmp_path / "sres2")) == a1


def test_cli_predictexons_two_workers(gpu_api, tmp_path):
    """the whole-workflow command split over two workers by contigs: fragment keys stay numbered over all contigs (each worker
    adds what the workers before it found), the merged called-exons DB equals the reference-made exon sets"""
    import subprocess
    from metaeuk_amd import build
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    _write_seq_db(str(tmp_path / "targets"), targets)
    _write_seq_db(str(tmp_path / "contigs"), contigs)
    (tmp_path / "contigs.dbtype").write_bytes((1).to_bytes(4, "little"))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls"), str(tmp_path / "tmp"),
                                       "-s", "5.7", "--ref-l2-bytes", "2097152", "--threads", "2", "--gpu", "0"], env=env, stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    got = _read_result_db(str(tmp_path / "calls"))
    assert "".join(">%d\n%s" % (c, got[c]) for c in range(len(contigs))) == _text("e2e_exons_expected.txt.gz")
    assert not os.path.exists(tmp_path / "calls_0") and not os.path.exists(tmp_path / "calls_1.orfs")


def test_cli_shards_of_a_crashed_launch_are_not_merged(gpu_api, tmp_path):
    """a shard file left behind by an earlier launch (another token, or none) must not reach the merged DB, and a worker that fails
    makes worker 0 give up instead of waiting"""
    import subprocess
    import time
    from metaeuk_amd import build, shard, synth
    targets, queries = synth.make_workload(4, 200, seed=13)
    _write_seq_db(str(tmp_path / "q"), list(queries))
    _write_seq_db(str(tmp_path / "t"), list(targets))
    flags = ["-s", "5.7", "--ref-l2-bytes", "2097152", "--threads", "2", "--gpu", "0"]
    subprocess.check_call([build.BIN, "prefilter", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "one")] + flags, stderr=subprocess.DEVNULL)
    # leftovers of a "crashed" launch: a complete-looking shard 1 with foreign content and no token, and a stale fragment count
    shard.write_result_db(str(tmp_path / "two_1"), [(0, "999\t1\t0\n")], 7)
    (tmp_path / "two_1.orfs").write_text("12345\n")
    procs = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MK_SHARD_TIMEOUT_S="120")
        procs.append(subprocess.Popen([build.BIN, "prefilter", str(tmp_path / "q"), str(tmp_path / "t"), str(tmp_path / "two")] + flags, env=env, stderr=subprocess.DEVNULL))
        if r == 0:
            time.sleep(3.0)                 # worker 0 finishes its half and meets the stale shard before worker 1 has started
    assert all(p.wait() == 0 for p in procs)
    assert shard.read_result_db(str(tmp_path / "two")) == shard.read_result_db(str(tmp_path / "one"))
    # worker 1 dies (its target DB does not exist): worker 0 sees its .failed marker and exits non-zero, nothing named <out>.dbtype appears
    t0 = time.time()
    procs = []
    for r in (0, 1):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MK_SHARD_TIMEOUT_S="120")
        procs.append(subprocess.Popen([build.BIN, "prefilter", str(tmp_path / "q"), str(tmp_path / ("t" if r == 0 else "missing")), str(tmp_path / "three")] + flags,
                                      env=env, stderr=subprocess.DEVNULL))
    assert [p.wait() != 0 for p in procs] == [True, True]
    assert time.time() - t0 < 60 and not os.path.exists(tmp_path / "three.dbtype")


def test_cli_contig_batches(gpu_api, tmp_path):
    """predictexons and extractorfs walk the contigs in batches bounded by nucleotides (MK_CLI_BATCH_NT forces tiny ones here): fragment
    keys run on from batch to batch, the output equals the single-batch run; also as two workers"""
    import subprocess
    from metaeuk_amd import build
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    _write_seq_db(str(tmp_path / "targets"), targets)
    _write_seq_db(str(tmp_path / "contigs"), contigs)
    (tmp_path / "contigs.dbtype").write_bytes((1).to_bytes(4, "little"))
    flags = ["-s", "5.7", "--ref-l2-bytes", "2097152", "--threads", "2", "--gpu", "0"]
    env = dict(os.environ, MK_CLI_BATCH_NT="60000")                 # ~12 contigs of 5 kb per batch
    subprocess.check_call([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls"), str(tmp_path / "tmp")] + flags,
                          env=env, stderr=subprocess.DEVNULL)
    got = _read_result_db(str(tmp_path / "calls"))
    assert "".join(">%d\n%s" % (c, got[c]) for c in range(len(contigs))) == _text("e2e_exons_expected.txt.gz")
    # round 6: the batches are queued in the search engine (batch k + 1 is read, scanned and translated while batch k is searched and the exon
    # sets of batch k - 1 are written); the DB is byte-identical to the run with everything in ONE batch, and the command accounts for its stages
    r = subprocess.run([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls_one"), str(tmp_path / "tmp")] + flags,
                       env=dict(os.environ, MK_CLI_BATCH_NT=str(1 << 29)), stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert b"predictexons stages: 1 batches" in r.stderr and b"search (blocked)" in r.stderr, r.stderr.decode()[-600:]
    for sfx in ("", ".index", ".dbtype"):
        assert open(str(tmp_path / "calls") + sfx, "rb").read() == open(str(tmp_path / "calls_one") + sfx, "rb").read(), sfx
    r = subprocess.run([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls_dflt"), str(tmp_path / "tmp")] + flags,
                       env=dict(os.environ), stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    assert open(str(tmp_path / "calls"), "rb").read() == open(str(tmp_path / "calls_dflt"), "rb").read()
    procs = []
    for r in range(2):
        e2 = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0")
        procs.append(subprocess.Popen([build.BIN, "predictexons", str(tmp_path / "contigs"), str(tmp_path / "targets"), str(tmp_path / "calls2"), str(tmp_path / "tmp")] + flags,
                                      env=e2, stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    assert _read_result_db(str(tmp_path / "calls2")) == got
    for name, e in (("orfs_one", dict(os.environ)), ("orfs_batched", env)):
        subprocess.check_call([build.BIN, "extractorfs", str(tmp_path / "contigs"), str(tmp_path / name), "--translate", "1", "--min-length", "15"], env=e, stderr=subprocess.DEVNULL)
    assert _read_result_db(str(tmp_path / "orfs_one")) == _read_result_db(str(tmp_path / "orfs_batched"))
    assert _read_result_db(str(tmp_path / "orfs_one_h")) == _read_result_db(str(tmp_path / "orfs_batched_h"))


def test_cli_equals_the_real_process(gpu_api, tmp_path):
    """the commands against the DBs the REAL `metaeuk predictexons` binary wrote (tests/golden/e2e_process_*): extractorfs --translate,
    prefilter, align with the argv the workflow passes (-s 5.7), and predictexons with NO -s (the workflow's own default, -s 4)"""
    import subprocess
    from metaeuk_amd import build
    targets, contigs = _lines("e2e_targets.txt.gz"), _lines("e2e_contigs.txt.gz")
    _write_seq_db(str(tmp_path / "targets"), targets)
    _write_seq_db(str(tmp_path / "contigs"), contigs)
    (tmp_path / "contigs.dbtype").write_bytes((1).to_bytes(4, "little"))
    run = lambda *a: subprocess.check_call([build.BIN] + [str(x) for x in a], stderr=subprocess.DEVNULL)
    run("extractorfs", tmp_path / "contigs", tmp_path / "aa_6f", "--translate", "1", "--min-length", "15", "--max-length", "32734", "--max-gaps", "2147483647",
        "--contig-start-mode", "2", "--contig-end-mode", "2", "--orf-start-mode", "1", "--forward-frames", "1,2,3", "--reverse-frames", "1,2,3",
        "--translation-table", "1", "--use-all-table-starts", "0", "--id-offset", "0", "--create-lookup", "0", "--threads", "4", "--compressed", "0", "-v", "3")
    aa, hdr = _read_result_db(str(tmp_path / "aa_6f")), _read_result_db(str(tmp_path / "aa_6f_h"))
    assert "".join("%s\t%s" % (hdr[k].rstrip("\n"), aa[k]) for k in sorted(aa)) == _text("e2e_process_orfs.txt.gz")
    run("prefilter", tmp_path / "aa_6f", tmp_path / "targets", tmp_path / "pref_0", "--sub-mat", "aa:blosum62.out,nucl:nucleotide.out",
        "--seed-sub-mat", "aa:VTML80.out,nucl:nucleotide.out", "-k", "0", "--target-search-mode", "0", "--k-score", "seq:2147483647,prof:2147483647",
        "--alph-size", "aa:21,nucl:5", "--max-seq-len", "65535", "--max-seqs", "300", "--split", "0", "--split-mode", "2", "--split-memory-limit", "0",
        "-c", "0", "--cov-mode", "0", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "1", "--diag-score", "1", "--exact-kmer-matching", "0",
        "--mask", "1", "--mask-prob", "0.9", "--mask-lower-case", "0", "--mask-n-repeat", "0", "--min-ungapped-score", "15", "--add-self-matches", "0",
        "--spaced-kmer-mode", "1", "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400", "--pcb", "substitution:4.100,context:5.800",
        "--threads", "4", "--compressed", "0", "-v", "3", "-s", "5.7", "--ref-l2-bytes", "2097152")
    blocks = lambda d: "".join(">%d\n%s" % (k, d[k]) for k in sorted(d))
    assert blocks(_read_result_db(str(tmp_path / "pref_0"))) == _text("e2e_process_pref.txt.gz")
    # the run statistics the reference's prefilter logs (Prefiltering::printStatistics): the same six lines, from the device counters
    r = subprocess.run([build.BIN, "prefilter", str(tmp_path / "aa_6f"), str(tmp_path / "targets"), str(tmp_path / "pref_stats"), "-s", "5.7", "--ref-l2-bytes", "2097152"],
                       stderr=subprocess.PIPE)
    assert r.returncode == 0
    expected_stats = open(os.path.join(GOLD, "e2e_process_prefilter_stats.txt")).read()
    assert expected_stats.startswith("246.638184 k-mers per position\n12 DB matches per sequence\n") and expected_stats in r.stderr.decode(), r.stderr.decode()[-600:]
    run("align", tmp_path / "aa_6f", tmp_path / "targets", tmp_path / "pref_0", tmp_path / "search_res", "--sub-mat", "aa:blosum62.out,nucl:nucleotide.out",
        "-a", "0", "--alignment-mode", "2", "--alignment-output-mode", "0", "--wrapped-scoring", "0", "-e", "100", "--min-seq-id", "0", "--min-aln-len", "11",
        "--seq-id-mode", "0", "--alt-ali", "0", "-c", "0", "--cov-mode", "0", "--max-seq-len", "65535", "--comp-bias-corr", "1", "--comp-bias-corr-scale", "1",
        "--max-rejected", "2147483647", "--max-accept", "2147483647", "--add-self-matches", "0", "--db-load-mode", "0", "--pca", "substitution:1.100,context:1.400",
        "--pcb", "substitution:4.100,context:5.800", "--score-bias", "0", "--realign", "0", "--realign-score-bias", "-0.2", "--realign-max-seqs", "2147483647",
        "--corr-score-weight", "0", "--gap-open", "aa:11,nucl:5", "--gap-extend", "aa:1,nucl:2", "--zdrop", "40", "--threads", "4", "--compressed", "0", "-v", "3",
        "--ref-l2-bytes", "2097152")
    assert blocks(_read_result_db(str(tmp_path / "search_res"))) == _text("e2e_process_aln.txt.gz")
    run("predictexons", tmp_path / "contigs", tmp_path / "targets", tmp_path / "calls4", tmp_path / "tmp", "--threads", "4", "--ref-l2-bytes", "2097152")
    assert blocks(_read_result_db(str(tmp_path / "calls4"))) == _text("e2e_process_calls_default_s4.txt.gz")
