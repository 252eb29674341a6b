#!/usr/bin/env python3
"""The predictexons chain inside the library: contigs -> mk_extract_orfs -> mk_search -> mk_predict_exons, stage by stage,
on BASELINE.json's config-2 shape by default (10 000 contigs x 100 000 targets).  Optionally times the reference's exon stage
(oracle/_ref/ref_harness exons, one thread) on a sample, fed by the library's own alignments written as text.

  python tools/bench_chain.py --contigs 10000 --targets 100000
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=10000)
    ap.add_argument("--targets", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    from metaeuk_amd import api, synth
    targets, founders = synth.make_targets(args.targets, args.seed)
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(args.contigs, founders, args.seed)]
    api.init(0)
    params = api.default_params()
    import numpy as np
    off = np.zeros(len(targets) + 1, dtype=np.uint64)
    np.cumsum([len(t) for t in targets], out=off[1:])
    db = api.TargetDB.from_codes(np.concatenate(targets).astype(np.uint8), off, params)
    best = None
    for _ in range(args.repeat + 1):                          # first pass warms pools and sizing memos
        t0 = time.time()
        o = api.Orfs(contigs)
        t1 = time.time()
        q = o.queries(params)
        t2 = time.time()
        api.search(db, q)
        t3 = time.time()
        pred = api.Predictions(db, o, q)
        t4 = time.time()
        cur = {"extract_orfs_s": t1 - t0, "queries_from_orfs_s": t2 - t1, "search_s": t3 - t2, "predict_exons_s": t4 - t3, "total_s": t4 - t0}
        if best is None or cur["total_s"] < best["total_s"]:
            best = cur
        n_frag, n_aln, n_pred, n_exon = o.n, int(api.align_result(q)[1][-1]), pred.n, len(pred.exons)
        multi = int((pred.predictions["n_exons"] > 1).sum()) if pred.n else 0
        pred.close(); q.close(); o.close()
    out = {"contigs": len(contigs), "nucleotides": sum(len(c) for c in contigs), "targets": len(targets), "fragments": n_frag,
           "alignments": n_aln, "predictions": n_pred, "exons": n_exon, "multi_exon_predictions": multi,
           "host_threads": int(api.lib().mk_host_threads())}
    out.update({k: round(v, 4) for k, v in best.items()})
    out["contigs_per_s"] = round(len(contigs) / best["total_s"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
