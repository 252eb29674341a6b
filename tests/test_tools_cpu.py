"""CPU tests (no GPU) of the host-side helpers of the experiment drivers under tools/: the job / unit lists the inter-sequence score-pass
experiment is driven with (tools/interseq_experiment.py) and the lane-step model's scheduling helpers (tools/sw_schedule_model.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_interseq_jobs_are_ordered_by_query_and_falling_target_length_and_cut_into_units_of_one_query():
    import interseq_experiment as ie
    rng = np.random.default_rng(9)
    nq, nt = 400, 300
    q_len = rng.integers(10, 90, nq)                       # some beyond 64 rows: not part of the experiment
    t_len = rng.integers(30, 600, nt)
    q_off = np.zeros(nq + 1, dtype=np.uint64); q_off[1:] = np.cumsum(q_len)
    t_off = np.zeros(nt + 1, dtype=np.uint64); t_off[1:] = np.cumsum(t_len)
    per_q = rng.integers(0, 40, nq); per_q[7] = 2500         # one query beyond a unit's size
    hit_off = np.zeros(nq + 1, dtype=np.uint64); hit_off[1:] = np.cumsum(per_q)
    hit_target = rng.integers(0, nt, int(hit_off[-1])).astype(np.uint32)
    q_len[7] = 30
    q_off[1:] = np.cumsum(q_len)
    jobs = ie.build_jobs(hit_target, hit_off, q_off, t_off)
    seen = np.zeros(len(hit_target), dtype=np.int32)
    for rows, (idx, j_tstart, j_tlen, jq, unit_start) in jobs.items():
        ql = q_len[jq.astype(np.int64)]
        assert ql.max() <= rows and np.all(ie.tile_rows(ql) == rows)
        assert np.array_equal(j_tstart, t_off[hit_target[idx]]) and np.array_equal(j_tlen, t_len[hit_target[idx]].astype(np.uint32))
        assert np.all(np.diff(jq.astype(np.int64)) >= 0)
        assert unit_start[0] == 0 and unit_start[-1] == len(jq)
        sizes = np.diff(unit_start.astype(np.int64))
        assert sizes.min() >= 1 and sizes.max() <= ie.UNIT_MAX
        for u in range(len(unit_start) - 1):
            a, b = int(unit_start[u]), int(unit_start[u + 1])
            assert len(set(jq[a:b].tolist())) == 1
            assert np.all(np.diff(j_tlen[a:b].astype(np.int64)) <= 0)
        seen[idx] += 1
    # every pair of a query of at most 64 residues exactly once, the others not at all
    pair_q = np.repeat(np.arange(nq), per_q)
    assert np.array_equal(seen, (q_len[pair_q] <= 64).astype(np.int32))
    if 32 in jobs:
        assert (np.diff(jobs[32][4].astype(np.int64)) == ie.UNIT_MAX).any()      # the 2 500-pair query was cut


def test_schedule_model_helpers():
    import sw_schedule_model as sm
    tl = [400, 390, 380, 200, 100, 90, 80, 70, 60]
    # the built kernel: two waves (8 + 1 pairs), steps = whole blocks of 16 beyond the longest target + the 15-step ramp
    assert sm.anti(tl, 32) == ((400 + 30) & ~15) * 28 + ((60 + 30) & ~15) * 28
    mk, tot = sm.group_makespan(tl, 8)
    assert tot == 400 + 380 + 100 + 80 + 60 and mk == 400              # five lane pairs on eight lanes: the longest one decides
    mk2, _ = sm.group_makespan(tl, 2)
    assert mk2 == max(400 + 80, 380 + 100 + 60)
    assert sm.tile_of(32) == 32 and sm.tile_of(33) == 48 and sm.tile_of(65) == 96
