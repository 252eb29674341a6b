"""Run the prefilter (and the pipelined search) of the headline workload repeatedly and compare every run's hit list with the first run's.
   python tools/determinism.py [--runs N] [--search M]     (METAEUK_AMD_LIB selects a library variant)"""
import argparse, os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from metaeuk_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=12)
ap.add_argument("--search", type=int, default=4)
ap.add_argument("--contigs", type=int, default=10000)
ap.add_argument("--targets", type=int, default=100000)
ap.add_argument("--k7", action="store_true", help="k = 7 forced: every query through the wide per-query kernel (what a 60 M-protein database takes)")
a = ap.parse_args()
api.init(0)
params = api.default_params()
if a.k7:
    params.kmer_size = 7
targets, queries, _ = bench.make_inputs(a.contigs, a.targets, 11, 0)
t_res, t_off = bench.pack(targets)
q_res, q_off = bench.pack(queries)
db = api.TargetDB.from_codes(t_res, t_off, params)
ref = None
bad = 0
def check(tag, hits, hoff):
    global ref, bad
    h = np.frombuffer(memoryview(hits), dtype=np.uint8).copy() if not isinstance(hits, np.ndarray) else hits.view(np.uint8).copy()
    o = np.asarray(hoff).copy()
    if ref is None:
        ref = (h, o)
        print(tag, "reference run:", int(o[-1]), "hits", flush=True)
        return
    if o[-1] == ref[1][-1] and np.array_equal(o, ref[1]) and np.array_equal(h, ref[0]):
        print(tag, "equal", flush=True)
        return
    bad += 1
    dq = np.nonzero(np.diff(o) != np.diff(ref[1]))[0]
    hv = h.view(api.HIT_DTYPE); rv = ref[0].view(api.HIT_DTYPE)
    for i in dq[:3]:
        a = hv[int(o[i]):int(o[i + 1])]; b = rv[int(ref[1][i]):int(ref[1][i + 1])]
        sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
        print("   query", int(i), "len", len(queries[i]), "only in this run:", sorted(sa - sb)[:12], "only in the reference run:", sorted(sb - sa)[:12], flush=True)
    print(tag, "DIFFERS: total", int(o[-1]), "vs", int(ref[1][-1]), "; queries with another hit count:", dq[:10].tolist(),
          "lengths", [len(queries[i]) for i in dq[:10]], "counts", [(int(o[i + 1] - o[i]), int(ref[1][i + 1] - ref[1][i])) for i in dq[:10]], flush=True)
for r in range(a.runs):
    q = api.Queries.from_codes(q_res, q_off, params)
    hits, hoff = api.prefilter(db, q)
    check("prefilter run %d" % r, hits, hoff)
    q.close()
pend = []
for r in range(a.search):
    q = api.Queries.from_codes(q_res, q_off, params)
    api.search_begin(db, q)
    pend.append(q)
    if len(pend) >= 2:
        q0 = pend.pop(0)
        (hits, hoff), _ = api.search_wait(q0)
        check("search run", hits, hoff)
        q0.close()
while pend:
    q0 = pend.pop(0)
    (hits, hoff), _ = api.search_wait(q0)
    check("search run", hits, hoff)
    q0.close()
print("RESULT: %d differing runs" % bad)
