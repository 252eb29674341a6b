"""diagnostic: heavy queries one by one -- index hits of the query vs whether the wide kernel agrees with the global path"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import config5_digest as c5
from metaeuk_amd import api
n_targets = int(sys.argv[1]); n_q = int(sys.argv[2]); lens = (int(sys.argv[3]), int(sys.argv[4]))
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
fr, foff, src = api.synth_fragments(n_q, res, off, seed=5, mutation_rate=0.1, min_len=lens[0], max_len=lens[1], random_every=10)
p = api.default_params(); p.kmer_size = 7
db = api.TargetDB.from_codes(res, off, p)
del res
os.environ["MK_PREFILTER_WIDE_POOL_GB"] = "96"
rows = []
for i in range(n_q):
    a, b = int(foff[i]), int(foff[i + 1])
    one_res, one_off = fr[a:b].copy(), np.array([0, b - a], dtype=np.uint64)
    out = {}
    for path in ("global", "wide"):
        os.environ["MK_PREFILTER_PATH"] = path
        q = api.Queries.from_codes(one_res, one_off, p)
        api.kernel_stats(reset=True)
        hits, hoff = api.prefilter(db, q, p)
        st = api.kernel_stats()
        out[path] = api.format_hits_bulk(hits, 0, int(hoff[1]))
        if path == "wide":
            w = st.get("prefilter_query_wide", {"alg_bytes": 0, "cells": 0})
            nh = (w["alg_bytes"] - 16 * w["cells"]) / 6
    ok = out["global"] == out["wide"]
    g, w_ = out["global"].decode().split("\n"), out["wide"].decode().split("\n")
    lost = [x for x in g if x and x not in set(w_)][:2]
    rows.append((b - a, int(nh), ok, lost))
    print(i, "len", b - a, "hits", int(nh), "OK" if ok else "DIFF", lost, flush=True)
