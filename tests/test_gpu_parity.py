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


def test_pipeline_vs_oracle(gpu_api, small_workload, tmp_path):
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
