"""Go / no-go measurement for a radix-partitioned index probe: tools/micro/mk_experiments.hip on the real probe
stream of the headline workload -- the similar k-mers of the first N fragments against the real 100 000-protein index.
   tools/micro/build.sh && python tools/partition_probe_experiment.py [n_queries] > gpurun_out/partition_probe.json"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metaeuk_amd import api, synth  # noqa: E402
import numpy as np  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    api.init(0)
    t, founders = synth.make_targets(100000, seed=11)
    q = synth.make_queries(n_contigs, founders, seed=11)[:nq]
    p = api.default_params()
    res = np.concatenate(t).astype(np.uint8); off = np.zeros(len(t) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(x) for x in t])
    db = api.TargetDB.from_codes(res, off, p)
    qres = np.concatenate(q).astype(np.uint8); qoff = np.zeros(len(q) + 1, dtype=np.uint64); qoff[1:] = np.cumsum([len(x) for x in q])
    Q = api.Queries.from_codes(qres, qoff, p)
    x = C.CDLL(os.path.join(ROOT, "tools", "micro", "_build", "libmk_experiments.so"))
    x.mkx_last_error.restype = C.c_char_p
    view = C.create_string_buffer(1024)
    vbytes = 0
    qoff_host = C.c_void_p()
    for cand in range(64, 1024, 8):      # the view's size is the library's business: ask until it agrees
        if api.lib().mk_debug_prefilter_view(db.h, Q.h, view, C.c_size_t(cand), C.byref(qoff_host)) == 0:
            vbytes = cand
            break
    if not vbytes:
        raise SystemExit("mk_debug_prefilter_view: " + api.lib().mk_last_error().decode())
    cases = [(P, rw) for P in (64, 128, 256, 512, 1024) for rw in (1, 2)]
    parts = (C.c_int * len(cases))(*[c[0] for c in cases])
    words = (C.c_int * len(cases))(*[c[1] for c in cases])
    out = (C.c_double * (8 + 4 * len(cases)))()
    if x.mkx_partition_probe(view, C.c_size_t(vbytes), qoff_host, C.c_uint32(len(q)), parts, words, C.c_int(len(cases)), out) != 0:
        raise SystemExit("mkx_partition_probe: " + x.mkx_last_error().decode())
    nk = out[0]
    rep = {"queries": len(q), "similar_kmers": nk, "present_kmers": out[1], "enumerate_count_ms": out[2], "enumerate_fill_ms": out[3],
           "direct_ms": {"ilp2": out[4], "ilp4": out[5], "ilp8": out[6]}, "direct_best_Gkmers_per_s": nk / min(out[4], out[5], out[6]) / 1e6,
           "all_variants_agree": bool(out[7]), "partitioned": []}
    for k, (P, rw) in enumerate(cases):
        mp, mq, mc, skew = out[8 + 4 * k: 12 + 4 * k]
        rep["partitioned"].append({"partitions": P, "record_bytes": 4 * rw, "count_ms": mc, "partition_ms": mp, "probe_ms": mq, "total_ms": mc + mp + mq,
                                   "Gkmers_per_s": nk / (mc + mp + mq) / 1e6, "vs_direct": min(out[4], out[5], out[6]) / (mc + mp + mq),
                                   "probe_alone_vs_direct": min(out[4], out[5], out[6]) / mq,
                                   "slot_slice_MB": 512.0 / P, "bitmap_slice_KB": 8192.0 / P, "records_written_GB": nk * 4 * rw / 1e9,
                                   "largest_partition_over_mean": skew})
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
