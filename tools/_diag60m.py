"""diagnostic: which prefilter front ends agree at large hit counts (GPU call F)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np
import config5_digest as c5
from metaeuk_amd import api

n_targets = int(sys.argv[1]); n_q = int(sys.argv[2]); lens = (int(sys.argv[3]), int(sys.argv[4]))
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
fr, foff, src = api.synth_fragments(n_q, res, off, seed=5, mutation_rate=0.1, min_len=lens[0], max_len=lens[1], random_every=10)
p = api.default_params()
db = api.TargetDB.from_codes(res, off, p)
del res
out = {}
for name, env in (("global", dict(MK_PREFILTER_PATH="global")), ("wide_enum7", dict(MK_PREFILTER_PATH="wide")), ("wide_lists", dict(MK_PREFILTER_PATH="wide", MK_PREFILTER_K7_LISTS="1")),
                  ("wide_enum7_again", dict(MK_PREFILTER_PATH="wide"))):
    for k in ("MK_PREFILTER_PATH", "MK_PREFILTER_K7_LISTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    q = api.Queries.from_codes(fr, foff, p)
    api.kernel_stats(reset=True)
    t0 = time.time()
    hits, hoff = api.prefilter(db, q, p)
    t = time.time() - t0
    st = api.kernel_stats()
    out[name] = (np.array(hits, copy=True), np.array(hoff, copy=True))
    print(name, "%.2f s" % t, "hits", int(hoff[-1]), {k: round(v["ms"]) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:6]}, flush=True)
ref = out["global"]
for name in out:
    if name == "global":
        continue
    h, ho = out[name]
    same_off = np.array_equal(ho, ref[1])
    bad = []
    if same_off:
        for i in range(len(ho) - 1):
            a, b = h[int(ho[i]):int(ho[i + 1])], ref[0][int(ho[i]):int(ho[i + 1])]
            if a.tobytes() != b.tobytes():
                d = [k for k in range(len(a)) if a[k].tobytes() != b[k].tobytes()]
                bad.append((i, len(a), d[:4], [tuple(int(x) for x in (a[k]["seq_id"], a[k]["pref_score"], a[k]["diagonal"])) for k in d[:2]],
                            [tuple(int(x) for x in (b[k]["seq_id"], b[k]["pref_score"], b[k]["diagonal"])) for k in d[:2]]))
    print(name, "vs global: offsets equal", same_off, "differing queries", len(bad), bad[:6], flush=True)
