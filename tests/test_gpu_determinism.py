"""The same batch searched again returns the same bytes.  Round 6 found a scheduling-dependent race this way (a workgroup barrier at the head of the stage
loop of the LDS sort: the compiler dropped the wait for the previous stage's ds_writes, profiles/r06_barrier_at_loop_head.txt): a handful of wrong prefilter
candidates per 2 * 10^6 queries, other ones every run, which the fixed-seed parity tests only see by luck.  The reference is a pure function of its input
(QueryMatcher::matchQuery, QueryMatcher.cpp:213-346: one thread per query, no shared state), so any run-to-run difference is a defect."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bytes(a):
    return np.ascontiguousarray(a).view(np.uint8).copy()


def test_repeated_searches_return_identical_bytes(gpu_api):
    api = gpu_api
    import bench
    targets, queries, _ = bench.make_inputs(2500, 30000, 5, 0)          # ~ 5 * 10^5 ORF fragments: every prefilter tier and tile configuration
    t_res, t_off = bench.pack(targets)
    q_res, q_off = bench.pack(queries)
    params = api.default_params()
    db = api.TargetDB.from_codes(t_res, t_off, params)
    ref = None
    for rnd in range(6):
        q = api.Queries.from_codes(q_res, q_off, params)
        if rnd % 2 == 0:
            (hits, hoff), (alns, aoff) = api.search(db, q)
        else:                                                            # the two stages as separate calls: the prefilter with the GPU to itself
            hits, hoff = api.prefilter(db, q)
            alns, aoff = api.align(db, q)
        got = (_bytes(hits), np.asarray(hoff).copy(), _bytes(alns), np.asarray(aoff).copy())
        q.close()
        if ref is None:
            ref = got
            assert int(ref[1][-1]) > 10 ** 6
            continue
        for name, a, b in zip(("hits", "hit offsets", "alignments", "alignment offsets"), got, ref):
            if not np.array_equal(a, b):
                off_a, off_b = (got[1], ref[1]) if name.startswith("hit") else (got[3], ref[3])
                bad = np.nonzero(np.diff(off_a.astype(np.int64)) != np.diff(off_b.astype(np.int64)))[0]
                pytest.fail("run %d: %s differ from run 0 (queries with another count: %s)" % (rnd, name, bad[:8].tolist()))
