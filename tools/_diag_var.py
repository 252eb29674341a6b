"""diagnostic: the wide kernel under debugging variants against the global path (== the reference) on heavy queries"""
import os, sys, time
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
KN = ("MK_PREFILTER_PATH", "MK_PREFILTER_K7_LISTS", "MK_PREFILTER_WIDE_POOL_GB", "MK_PREFILTER_WIDE_FLAGS", "MK_PREFILTER_WIDE_REGS", "MK_PREFILTER_DEBUG")
def run(env):
    for k in KN: os.environ.pop(k, None)
    os.environ.update(env)
    q = api.Queries.from_codes(fr, foff, p)
    hits, hoff = api.prefilter(db, q, p)
    return [api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])) for i in range(n_q)]
ref = run(dict(MK_PREFILTER_PATH="global"))
for name, env in (("wide", {}), ("wide pool 96 GB", dict(MK_PREFILTER_WIDE_POOL_GB="96")), ("wide pool 96 GB, no exact filter", dict(MK_PREFILTER_WIDE_POOL_GB="96", MK_PREFILTER_WIDE_FLAGS="1")),
                  ("wide pool 96 GB, no optimistic collect", dict(MK_PREFILTER_WIDE_POOL_GB="96", MK_PREFILTER_WIDE_FLAGS="2")), ("wide pool 96 GB, neither", dict(MK_PREFILTER_WIDE_POOL_GB="96", MK_PREFILTER_WIDE_FLAGS="3")),
                  ("wide pool 96 GB, 128 regs", dict(MK_PREFILTER_WIDE_POOL_GB="96", MK_PREFILTER_WIDE_REGS="128")), ("wide pool 96 GB, lists", dict(MK_PREFILTER_WIDE_POOL_GB="96", MK_PREFILTER_K7_LISTS="1"))):
    env = dict(env, MK_PREFILTER_PATH="wide", MK_PREFILTER_DEBUG="1")
    got = run(env)
    bad = [i for i in range(n_q) if got[i] != ref[i]]
    print(name, ": differing queries", len(bad), bad[:12], flush=True)
