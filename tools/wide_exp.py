"""The prefilter at config-5 scale with the library's debug lines (MK_PREFILTER_DEBUG: work items, halved parts, workgroup ticks per phase of the wide
per-query kernel).  Early in round 6 the kernel had a switch that left parts of its work out (MK_PREFILTER_WIDE_EXP: no region stores / no class
atomic / no pass 2; profiles/r06_config5.txt item 2); the switch is gone, the third argument is kept as a repetition count.
   gpurun -- 'timeout 500 python tools/wide_exp.py 60000000 4000 > gpurun_out/wide_exp.txt 2>&1'"""
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
exps = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
p = api.default_params()
db = api.TargetDB.from_codes(res, off, p)
fr, foff, src = api.synth_fragments(n_frag, res, off, **c5.FRAGMENTS)
del res
q = api.Queries.from_codes(fr, foff, p)
for e in exps:
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
