"""What bounds pass 1 of the wide per-query kernel at config-5 scale: the kernel with parts of its work switched off (MK_PREFILTER_WIDE_EXP; the results are
wrong, only the times count).  gpurun -- 'python tools/wide_exp.py 60000000 4000 > gpurun_out/wide_exp.txt 2>&1'"""
import os
import sys
import time

os.environ["MK_DEBUG"] = "1"
os.environ["MK_PREFILTER_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config5_digest as c5  # noqa: E402
from metaeuk_amd import api  # noqa: E402

n_targets = int(sys.argv[1]) if len(sys.argv) > 1 else 60000000
n_frag = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
exps = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 3, 4, 5, 7, 8, 15, 0]
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
p = api.default_params()
db = api.TargetDB.from_codes(res, off, p)
fr, foff, src = api.synth_fragments(n_frag, res, off, **c5.FRAGMENTS)
del res
q = api.Queries.from_codes(fr, foff, p)
for e in exps:
    os.environ["MK_PREFILTER_WIDE_EXP"] = str(e)
    for rep in range(2):
        api.kernel_stats(reset=True)
        t0 = time.time()
        try:
            api.prefilter(db, q, p)
        except Exception as ex:
            print("exp", e, "failed:", ex)
        t = time.time() - t0
    st = api.kernel_stats()
    print("EXP %2d  prefilter %.3f s  %s" % (e, t, {k: round(v["ms"], 1) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 5.0}), flush=True)
