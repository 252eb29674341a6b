"""Experiment: do the prefilter's index probes get cheaper when the k-mer starts of many queries are walked in the order of their HOME
TILE (the 4096-cell tile of the index table that holds the in-quad variants of the query k-mer) instead of query by query?
Runs the enumerate + probe kernel (counting form: bitmap, slots, no entries, no stores) over the first N fragments of the headline
workload both ways.   python tools/probe_order_experiment.py [n_contigs] [n_targets] [n_queries]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metaeuk_amd import api, synth  # noqa: E402


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n_targets = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    nq_arg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    api.init(0)
    t, founders = synth.make_targets(n_targets, seed=11)
    q = [synth.codes_to_str(c) for c in synth.make_queries(n_contigs, founders, seed=11)]
    p = api.default_params()
    import numpy as np
    res = np.concatenate(t); off = np.zeros(len(t) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(x) for x in t])
    db = api.TargetDB.from_codes(res, off, p)
    Q = api.Queries(q, p)
    nq = min(Q.n, nq_arg or Q.n)
    out = (C.c_double * 4)()
    res = {}
    for mode, name in ((0, "query_order"), (1, "home_tile_order"), (0, "query_order_again")):
        api._chk(api.lib().mk_debug_probe_order(db.h, Q.h, C.c_uint32(nq), C.c_int(mode), out))
        res[name] = dict(ms=out[0], kmers=out[1], hits=out[2], starts=out[3], kmers_per_s=out[1] / out[0] * 1e3)
    print(json.dumps(dict(queries=nq, **res), indent=1))


if __name__ == "__main__":
    main()
