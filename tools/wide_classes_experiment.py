"""16 or 64 target classes in the wide per-query kernel at config 5's real size (60 M proteins) and at 11.8 M: python tools/wide_classes_experiment.py <n_targets> <n_fragments>"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import config5_digest as c5
from metaeuk_amd import api
n_targets, n_q = int(sys.argv[1]), int(sys.argv[2])
os.environ["MK_DEBUG"] = "1"
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
fr, foff, src = api.synth_fragments(n_q, res, off, **c5.FRAGMENTS)
p = api.default_params()
db = api.TargetDB.from_codes(res, off, p)
del res
out = {"n_targets": n_targets, "fragments": n_q, "runs": []}
ref = None
for classes in ("64", "16", "64", "16"):       # (128 and 256 classes were instantiated for GPU call r04n and removed afterwards)
    os.environ["MK_PREFILTER_WIDE_CLASSES"] = classes
    q = api.Queries.from_codes(fr, foff, p)
    api.kernel_stats(reset=True)
    t0 = time.time()
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    t = time.time() - t0
    st = api.kernel_stats()
    sig = (hits.tobytes(), bytes(alns))
    if ref is None:
        ref = sig
    out["runs"].append({"classes": int(classes), "t_search_s": round(t, 3), "fragments_per_s": round(n_q / t, 1), "same_result": sig == ref,
                        "kernels_ms": {k: round(v["ms"], 1) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:8]}})
    print(out["runs"][-1], flush=True)
print(json.dumps(out))
