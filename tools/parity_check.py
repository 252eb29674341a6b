#!/usr/bin/env python3
"""Large-scale parity check on a GPU box: the HIP path vs the reference's own compiled code
(oracle/_ref/ref_harness) -- or the C oracle when the harness is absent -- on a synthetic workload.

  python tools/parity_check.py --contigs 300 --targets 5000 [--seed 11] [--ref|--oracle]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=300)
    ap.add_argument("--targets", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--oracle", action="store_true", help="compare with the C oracle instead of the reference harness")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    import oracle
    from metaeuk_amd import api, synth
    targets, queries = synth.make_workload(args.contigs, args.targets, args.seed)
    api.init(0)
    params = api.default_params()
    import ctypes
    l2 = ctypes.CDLL(None).sysconf(191)   # _SC_LEVEL2_CACHE_SIZE (glibc x86-64); Util::getL2CacheSize (Util.cpp:317-332)
    params.host_l2_bytes = l2 if l2 and l2 > 0 else 262144      # reproduce THIS host's reference run
    t0 = time.time()
    db = api.TargetDB(targets, params)
    t_db = time.time() - t0
    t0 = time.time()
    q = api.Queries(queries, params)
    (hits, hoff), (alns, aoff) = api.search(db, q)          # the pipelined pass bench.py times
    t_pref = time.time() - t0
    t0 = time.time()
    q2 = api.Queries(queries, params)                       # ... and the two module calls must give the same bytes
    hits2, hoff2 = api.prefilter(db, q2)
    alns2, aoff2 = api.align(db, q2)
    t_aln = time.time() - t0
    import numpy as np
    same_calls = (np.array_equal(np.asarray(hoff), np.asarray(hoff2)) and np.array_equal(np.asarray(aoff), np.asarray(aoff2))
                  and hits.tobytes() == hits2.tobytes()
                  and api.format_alignments(alns, 0, int(aoff[-1])) == api.format_alignments(alns2, 0, int(aoff2[-1])))
    with tempfile.TemporaryDirectory() as tmp:
        use_ref = os.path.exists(oracle.REF) and not args.oracle
        t0 = time.time()
        if use_ref:
            matdir = oracle.write_matrix_files(os.path.join(tmp, "mat"))
            oracle.REF_MATDIR = matdir
            rpref, raln = oracle.run_ref_pipeline(targets, queries, tmp, extra=["--threads", str(args.threads)])
        else:
            rpref, raln = oracle.run_pipeline(targets, queries, tmp, extra=["--l2", str(params.host_l2_bytes)])
        t_cpu = time.time() - t0
    bad_p = bad_a = 0
    first = None
    for i in range(len(queries)):
        gp = api.format_hits(hits, int(hoff[i]), int(hoff[i + 1]))
        ga = api.format_alignments(alns, int(aoff[i]), int(aoff[i + 1]))
        if gp != rpref[i]:
            bad_p += 1
            first = first or ("pref", i, gp[:200], rpref[i][:200])
        if ga != raln[i]:
            bad_a += 1
            first = first or ("aln", i, ga[:300], raln[i][:300])
    print(json.dumps({"queries": len(queries), "targets": len(targets), "against": "reference" if use_ref else "oracle",
                      "pref_hits": int(hoff[-1]), "alignments": int(aoff[-1]),
                      "pref_blocks_differ": bad_p, "aln_blocks_differ": bad_a,
                      "mk_search_equals_prefilter_then_align": bool(same_calls),
                      "gpu_s": {"db": round(t_db, 2), "search": round(t_pref, 2), "prefilter_then_align": round(t_aln, 2)},
                      "cpu_total_s": round(t_cpu, 2), "cpu_threads": args.threads}))
    if not same_calls:
        print("mk_search and mk_prefilter+mk_align DIFFER")
        sys.exit(1)
    if first:
        print("FIRST DIFFERENCE:", first)
        sys.exit(1)


if __name__ == "__main__":
    main()
