"""mk_search_begin / mk_search_wait (round 5): batches queued behind each other return exactly what the blocking mk_search returns --
the reference has no notion of a caller's batches (Prefiltering::runSplit and Alignment::run loop over ALL queries of the DB,
Prefiltering.cpp:817-886, Alignment.cpp:312-514), so the result of a query must not depend on what is in flight around it."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _blocks(api, res, n):
    (hits, hoff), (alns, aoff) = res
    return [(api.format_hits(hits, int(hoff[i]), int(hoff[i + 1])), api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1]))) for i in range(n)]


def test_queued_batches_equal_the_blocking_search_and_the_oracle(gpu_api, tmp_path):
    api = gpu_api
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(30, 400, seed=23)
    params = api.default_params()
    db = api.TargetDB(targets, params)
    # three batches of DIFFERENT query lengths (the score tables of the alignment stage are snapshots per set of lengths: a batch in flight
    # keeps its own while the next one extends them), the middle one tiny, the last one with an empty query
    third = len(queries) // 3
    short = [q for q in queries if len(q) <= 40][:200]
    batches = [queries[:third], short[:7], queries[third:2 * third] + [""] + queries[2 * third:]]
    want = []
    for b in batches:
        q = api.Queries(b, params)
        want.append(_blocks(api, api.search(db, q), len(b)))
        q.close()
    # all three in flight at once, collected in order
    qs = [api.Queries(b, params) for b in batches]
    for q in qs:
        api.search_begin(db, q)
    got = [_blocks(api, api.search_wait(q), len(b)) for q, b in zip(qs, batches)]
    assert got == want
    # begin the next while the previous one is collected (the loop of bench.py), twice over
    for rnd in range(2):
        pending, out = None, []
        for b in batches:
            q = api.Queries(b, params)
            api.search_begin(db, q)
            if pending is not None:
                out.append(_blocks(api, api.search_wait(pending[0]), len(pending[1])))
                pending[0].close()
            pending = (q, b)
        out.append(_blocks(api, api.search_wait(pending[0]), len(pending[1])))
        pending[0].close()
        assert out == want, "round %d" % rnd
    # ... and the oracle on the first batch
    opref, oaln = oracle.run_pipeline(targets, batches[0], str(tmp_path), extra=["--l2", str(params.host_l2_bytes)])
    assert [w[0] for w in want[0]] == opref
    assert [w[1] for w in want[0]] == oaln
    for q in qs:
        q.close()


def test_blocking_calls_and_destruction_while_a_search_is_in_flight(gpu_api):
    api = gpu_api
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(20, 300, seed=29)
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q1 = api.Queries(queries, params)
    ref = _blocks(api, api.search(db, q1), len(queries))
    qa, qb = api.Queries(queries, params), api.Queries(queries[:50], params)
    api.search_begin(db, qa)
    # a blocking entry point first lets the search in flight finish; the queued batch stays collectable afterwards
    hb, ob = api.prefilter(db, qb)
    assert [api.format_hits(hb, int(ob[i]), int(ob[i + 1])) for i in range(50)] == [r[0] for r in ref[:50]]
    assert _blocks(api, api.search_wait(qa), len(queries)) == ref
    # waiting twice, or for a batch that was never begun, is an error -- not a hang
    with pytest.raises(Exception):
        api.search_wait(qa)
    with pytest.raises(Exception):
        api.search_wait(qb)
    # beginning a batch twice is refused; destroying a batch in flight waits for it
    api.search_begin(db, qa)
    with pytest.raises(Exception):
        api.search_begin(db, qa)
    qa.close()
    api.search_begin(db, qb)
    qb.close()
    q1.close()
    db.close()


def test_queued_searches_against_two_databases(gpu_api):
    """batches queued against different databases: each is searched against its own (the sizing memo and the score tables follow the database)"""
    api = gpu_api
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(10, 200, seed=31)
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q_ok = api.Queries(queries, params)
    ref = _blocks(api, api.search(db, q_ok), len(queries))
    small = api.TargetDB(targets[:20], params)          # its hits name targets < 20 only; the big database's name up to 199
    qa, qb, qc = api.Queries(queries, params), api.Queries(queries, params), api.Queries(queries, params)
    api.search_begin(db, qa)
    api.search_begin(small, qb)
    api.search_begin(db, qc)
    assert _blocks(api, api.search_wait(qa), len(queries)) == ref
    (hits_b, hoff_b), _ = api.search_wait(qb)
    assert int(hoff_b[-1]) == 0 or int(np.array(hits_b["seq_id"]).max()) < 20
    assert _blocks(api, api.search_wait(qc), len(queries)) == ref
    for x in (qa, qb, qc, q_ok):
        x.close()
    small.close()
    db.close()


def test_one_launch_per_register_class_equals_the_launches_per_tile_configuration(gpu_api, monkeypatch):
    """MK_SW_MULTI=1: the position / reverse passes as one persistent launch per register class with the bounds of the tile configurations read on the
    device (mk_sw.hip: sw_multi_kernel) -- measured, not the default (profiles/r05_search_engine.txt) -- return what the launches per tile
    configuration return; queries of every tile class, and one beyond the largest tile (row tiles: the per-configuration path serves that batch)."""
    api = gpu_api
    import random
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(30, 400, seed=37)
    rng = random.Random(3)
    aa = synth.AA
    # long queries made of target pieces so that they align: 300 .. 1 000 rows
    for L in (300, 420, 600, 800, 1000):
        t = rng.choice([x for x in targets if len(x) >= 150])
        s = (t * (L // len(t) + 1))[:L]
        queries.append("".join(c if rng.random() > 0.1 else rng.choice(aa) for c in s))
    params = api.default_params()
    db = api.TargetDB(targets, params)
    results = {}
    for name, batch in (("single tile", queries), ("with a query of 1 500 rows", queries + [(targets[0] * 12)[:1500]])):
        for multi in ("0", "1"):
            monkeypatch.setenv("MK_SW_MULTI", multi)
            q = api.Queries(batch, params)
            results[(name, multi)] = _blocks(api, api.search(db, q), len(batch))
            q.close()
        assert results[(name, "0")] == results[(name, "1")], name
    assert sum(len(b[1]) for b in results[("single tile", "1")]) > 1000
    monkeypatch.delenv("MK_SW_MULTI")
    db.close()


def test_shutdown_joins_the_search_threads_and_the_next_search_restarts_them(gpu_api):
    """mk_shutdown (round 6): waits for the batch in flight, stops and joins the engine's threads; the batch stays collectable; the next search
    starts the threads again and returns the same bytes."""
    api = gpu_api
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(10, 200, seed=43)
    params = api.default_params()
    db = api.TargetDB(targets, params)
    q0 = api.Queries(queries, params)
    ref = _blocks(api, api.search(db, q0), len(queries))
    qa = api.Queries(queries, params)
    api.search_begin(db, qa)
    api.shutdown()
    api.shutdown()                                   # idempotent
    assert _blocks(api, api.search_wait(qa), len(queries)) == ref
    qb = api.Queries(queries, params)
    assert _blocks(api, api.search(db, qb), len(queries)) == ref
    for x in (q0, qa, qb):
        x.close()
    db.close()


def test_a_process_that_exits_with_a_search_in_flight_ends_cleanly():
    """the library registers mk_shutdown with atexit when the first search begins: a host that exits between mk_search_begin and mk_search_wait
    (an exception, an interpreter shutting down) ends with its own exit code -- no hang, no crash in the teardown of the statistics map, the
    scratch buffers or the HIP runtime under the feet of the engine's threads."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from metaeuk_amd import api, synth\n"
            "api.init(0)\n"
            "t, qs = synth.make_workload(30, 400, seed=47)\n"
            "p = api.default_params(); db = api.TargetDB(t, p); q = api.Queries(qs, p)\n"
            "api.search_begin(db, q)\n"
            "print('begun', flush=True)\n"
            "sys.exit(7)\n") % root
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-2000:])
    assert b"begun" in r.stdout
