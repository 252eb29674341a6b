"""Go / no-go measurement for an inter-sequence score pass ("next round" item 3): tools/micro/mk_experiments.hip,
mkx_interseq_score, on the REAL pairs of the headline workload -- the prefilter hits of the first N contigs' ORF fragments against the
100 000 proteins, the queries of at most 64 residues -- against the product's own score pass (sw_fwd_rows32 / 48 / 64 of mk_align alone) on
the same pairs; scores checked against the product's mk_sw_pairs on a sample.  The model that chose this variant: tools/sw_schedule_model.py.
   tools/micro/build.sh && python tools/interseq_experiment.py [n_contigs] > gpurun_out/interseq.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

UNIT_MAX = 2048


def tile_rows(q_len):
    """the product's tile of a query (mk_kernels.hpp: sw_cfg_rows), 0 beyond 64 rows"""
    return np.where(q_len <= 32, 32, np.where(q_len <= 48, 48, np.where(q_len <= 64, 64, 0)))


def build_jobs(hit_target, hit_off, q_off, t_off):
    """pairs of the queries of at most 64 residues, per tile: ordered by query and falling target length, cut into units of one query
    (at most UNIT_MAX pairs).  -> {rows: (pair index into the hit list, j_tstart, j_tlen, j_q, unit_start)}"""
    nq = len(hit_off) - 1
    q_len = (q_off[1:] - q_off[:-1]).astype(np.int64)
    per_q = (hit_off[1:] - hit_off[:-1]).astype(np.int64)
    pair_q = np.repeat(np.arange(nq, dtype=np.uint32), per_q)
    t_len_all = (t_off[1:] - t_off[:-1]).astype(np.int64)
    pair_tlen = t_len_all[hit_target]
    rows_q = tile_rows(q_len)
    pair_rows = rows_q[pair_q]
    out = {}
    for rows in (32, 48, 64):
        sel = np.nonzero(pair_rows == rows)[0]
        if len(sel) == 0:
            continue
        order = np.lexsort((-pair_tlen[sel], pair_q[sel]))            # by query, inside it by falling target length (stable)
        idx = sel[order]
        jq = pair_q[idx].astype(np.uint32)
        jt = hit_target[idx]
        j_tstart = t_off[jt].astype(np.uint64)
        j_tlen = t_len_all[jt].astype(np.uint32)
        # units: a new one where the query changes, and every UNIT_MAX pairs inside a query
        change = np.ones(len(jq), dtype=bool)
        change[1:] = jq[1:] != jq[:-1]
        starts = np.nonzero(change)[0]
        run_id = np.cumsum(change) - 1
        within = np.arange(len(jq)) - starts[run_id]
        unit_flag = change | (within % UNIT_MAX == 0)
        unit_start = np.concatenate([np.nonzero(unit_flag)[0], [len(jq)]]).astype(np.uint32)
        out[rows] = (idx, np.ascontiguousarray(j_tstart), np.ascontiguousarray(j_tlen), np.ascontiguousarray(jq), np.ascontiguousarray(unit_start))
    return out


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    from metaeuk_amd import api, synth
    import oracle
    api.init(0)
    t, founders = synth.make_targets(100000, seed=11)
    q = synth.make_queries(n_contigs, founders, seed=11)
    p = api.default_params()
    t_res = np.concatenate(t).astype(np.uint8); t_off = np.zeros(len(t) + 1, dtype=np.uint64); t_off[1:] = np.cumsum([len(x) for x in t])
    q_res = np.concatenate(q).astype(np.uint8); q_off = np.zeros(len(q) + 1, dtype=np.uint64); q_off[1:] = np.cumsum([len(x) for x in q])
    db = api.TargetDB.from_codes(t_res, t_off, p)
    Q = api.Queries.from_codes(q_res, q_off, p)
    hits, hoff = api.prefilter(db, Q)
    hit_target = np.array(hits["seq_id"], dtype=np.uint32)
    hoff = np.array(hoff, dtype=np.uint64)
    # the product's alignment stage alone on these pairs: kernel time of its score pass per tile
    api.kernel_stats(reset=True)
    api.align(db, Q)
    ks = api.kernel_stats()
    product = {rows: ks.get("sw_fwd_rows%d" % rows, {}).get("ms", 0.0) for rows in (32, 48, 64)}
    bias = Q.derived()[2]
    m = oracle.submat(0, 2.0, 0.0)
    mat = np.zeros(441, dtype=np.int8)
    for a in range(21):
        for b in range(21):
            mat[a * 21 + b] = m.sub[a][b]
    jobs = build_jobs(hit_target, hoff, q_off, t_off)
    x = C.CDLL(os.path.join(ROOT, "tools", "micro", "_build", "libmk_experiments.so"))
    x.mkx_last_error.restype = C.c_char_p
    P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rep = {"contigs": n_contigs, "queries": len(q), "pairs": int(hoff[-1]), "tiles": {}}
    rng = np.random.default_rng(5)
    for rows, (idx, j_tstart, j_tlen, jq, unit_start) in jobs.items():
        n = len(jq)
        cells = float(np.sum((q_off[jq.astype(np.int64) + 1] - q_off[jq.astype(np.int64)]).astype(np.float64) * j_tlen.astype(np.float64)))
        entry = {"pairs": n, "units": len(unit_start) - 1, "cells": cells, "product_score_pass_ms": product[rows], "inter_sequence": []}
        score = None
        for waves in (2, 4, 8):
            out_score = np.zeros(n, dtype=np.int32)
            out = (C.c_double * 8)()
            rc = x.mkx_interseq_score(P(q_res), P(bias), P(q_off), C.c_uint32(len(q)), P(t_res), C.c_uint64(len(t_res)), P(mat),
                                      P(j_tstart), P(j_tlen), P(jq), C.c_uint64(n), P(unit_start), C.c_uint32(len(unit_start) - 1),
                                      C.c_int(rows), C.c_int(p.gap_open), C.c_int(p.gap_extend), C.c_int(waves), C.c_int(3), P(out_score), out)
            if rc != 0:
                raise SystemExit("mkx_interseq_score: " + x.mkx_last_error().decode())
            entry["inter_sequence"].append({"waves_per_cu": waves, "ms": out[0], "lanes_busy": out[1], "profile_builds": out[2], "lds_bytes_per_wave": out[3],
                                            "vs_product": product[rows] / out[0] if out[0] > 0 else None, "gcups": cells / out[0] / 1e6})
            if score is None:
                score = out_score
            else:
                entry["same_scores_for_every_launch_shape"] = bool(np.array_equal(score, out_score)) and entry.get("same_scores_for_every_launch_shape", True)
        # the product's scores of a sample of the pairs (mk_sw_pairs: the int32 kernel, own profile per pair)
        take = rng.choice(n, size=min(n, 50000), replace=False)
        ref = api.sw_pairs(db, Q, jq[take], hit_target[idx[take]], with_start=False)[:, 0]
        bad = int(np.count_nonzero(ref != score[take]))
        entry["checked_pairs"] = int(len(take)); entry["mismatching_scores"] = bad
        rep["tiles"][str(rows)] = entry
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
