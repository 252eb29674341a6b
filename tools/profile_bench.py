#!/usr/bin/env python3
"""Timing of the profile-target path (BASELINE config 4) on one GPU: synthetic profiles (metaeuk_amd.synth.make_profiles of
the protein families of the main workload) searched against the six-frame fragments of synthetic contigs -- profiles = queries,
fragments = indexed targets, then swapresults.  Prints one JSON object with the stage times and the per-kernel statistics; with
--cpu-sample N the oracle's restatement of the reference path (oracle/_build/mko_cli profilesearch, OpenMP) is timed on the first N
profiles beside it.     gpurun -- 'python tools/profile_bench.py --profiles 50000 --contigs 10000 > gpurun_out/profile_bench.json'"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profiles", type=int, default=50000)
    ap.add_argument("--contigs", type=int, default=10000)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--cpu-sample", type=int, default=0)
    args = ap.parse_args()
    from metaeuk_amd import api, synth
    api.init(0)
    t0 = time.time()
    proteins, founders = synth.make_targets(args.profiles, args.seed)
    entries = synth.make_profiles(proteins, args.seed)
    contigs = ["".join("ACGT"[b] for b in c) for c in synth.make_contigs(args.contigs, founders, args.seed)]
    orfs = api.Orfs(contigs)
    frags = [orfs.protein(k) for k in range(orfs.n)]
    t_gen = time.time() - t0
    n_frag, n_prof = len(frags), len(entries)
    p = api.default_params()
    p.sensitivity = 4.0
    p.profile_search = 1
    p.max_seqs = max(300, n_frag)
    p.evalue_thr = float("%g" % (100.0 * (np.float32(n_frag) / np.float32(n_prof))))
    t0 = time.time()
    db = api.TargetDB(frags, p)
    t_index = time.time() - t0
    cols = sum(len(e) // 25 for e in entries)
    res = {"workload": "%d synthetic profiles (%d columns) x %d fragments (%d aa) of %d contigs, -s 4" % (n_prof, cols, n_frag, int(db.off[-1]), args.contigs),
           "generate_s": round(t_gen, 2), "fragment_index_s": round(t_index, 2), "steps": []}
    residues = sum(len(e) for e in entries) // 25 - n_prof
    p_cols, p_off = api.Profiles.pack(entries)
    for s in range(args.steps):
        api.kernel_stats(reset=True)
        t0 = time.time()
        q = api.Profiles.from_columns(p_cols, p_off, p)
        t1 = time.time()
        hits, hoff = api.prefilter(db, q, p)
        t2 = time.time()
        alns, aoff = api.align(db, q, p)
        t3 = time.time()
        sp = api.default_params()
        sp.evalue_thr = 1.7976931348623157e308
        sw, soff = api.swap_alignments(alns, aoff, n_frag, residues, params=sp)
        t4 = time.time()
        st = api.kernel_stats()
        cells = sum(v["cells"] for k, v in st.items() if k.startswith("sw_fwd"))
        res["steps"].append({"upload_derive_s": round(t1 - t0, 3), "prefilter_s": round(t2 - t1, 3), "align_s": round(t3 - t2, 3), "swap_s": round(t4 - t3, 3),
                             "total_s": round(t4 - t0, 3), "profiles_per_s": round(n_prof / (t4 - t0), 1), "prefilter_hits": int(hoff[-1]),
                             "alignments": int(aoff[-1]), "sw_fwd_cells": cells,
                             "kernels_ms": {k: round(v["ms"], 2) for k, v in sorted(st.items()) if v["ms"] >= 0.5}})
        del q
    if args.cpu_sample:
        n = min(args.cpu_sample, n_prof)
        with tempfile.TemporaryDirectory() as tmp:
            open(os.path.join(tmp, "p.bin"), "wb").write(b"".join(entries[:n]))
            off, lines = 0, []
            for k, e in enumerate(entries[:n]):
                lines.append("%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
            open(os.path.join(tmp, "p.index"), "w").write("".join(lines))
            open(os.path.join(tmp, "f.txt"), "w").write("\n".join(frags) + "\n")
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/mko_cli"], stdout=subprocess.DEVNULL)
            t0 = time.time()
            out = subprocess.check_output([os.path.join(ROOT, "oracle", "_build", "mko_cli"), "profilesearch", os.path.join(tmp, "p.bin"), os.path.join(tmp, "p.index"),
                                           os.path.join(tmp, "f.txt"), os.path.join(tmp, "out"), "-e", repr(100.0 * n / n_prof)])
            res["cpu_port"] = {"profiles": n, "wall_s_incl_index_build": round(time.time() - t0, 2), "threads": api.lib().mk_host_threads(), "info": json.loads(out)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
