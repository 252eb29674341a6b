"""Where the time of a search goes at BASELINE config 5's scale on one GPU: the database of tests/test_gpu_scale.py's config-5 case
(11.8 M proteins, 4.4e9 residues, k = 7), searched with N planted / background fragments; per-kernel time of mk_search.
   gpurun -- 'python tools/config5_search_profile.py 11800000 20000 100000 > gpurun_out/config5_search.json'"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config5_digest as c5
from metaeuk_amd import api

n_targets = int(sys.argv[1]) if len(sys.argv) > 1 else 11800000
sizes = [int(x) for x in sys.argv[2:]] or [20000]
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
p = api.default_params()
t0 = time.time()
db = api.TargetDB.from_codes(res, off, p)
out = {"n_targets": n_targets, "target_residues": int(off[-1]), "kmer_size": db.kmer_size(), "index_entries": db.index_entries(), "t_targetdb_s": round(time.time() - t0, 2), "runs": []}
for n in sizes:
    fr, foff, src = api.synth_fragments(n, res, off, **c5.FRAGMENTS)
    q = api.Queries.from_codes(fr, foff, p)
    for rep in range(2):
        api.kernel_stats(reset=True)
        t0 = time.time()
        (hits, hoff), (alns, aoff) = api.search(db, q, p)
        t = time.time() - t0
    st = api.kernel_stats()
    out["runs"].append({"fragments": n, "fragment_residues": int(foff[-1]), "t_search_s": round(t, 3), "fragments_per_s": round(n / t, 1), "pref_hits": int(hoff[-1]), "alignments": int(aoff[-1]),
                        "kernels_ms": {k: round(v["ms"], 1) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 1.0}})
print(json.dumps(out, indent=1))
