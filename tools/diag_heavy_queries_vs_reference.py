"""diagnostic: wide kernel / global path / the reference's own code on queries with millions of index hits (GPU call G)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import config5_digest as c5
import oracle
from metaeuk_amd import api

n_targets = int(sys.argv[1]); n_q = int(sys.argv[2]); lens = (int(sys.argv[3]), int(sys.argv[4]))
api.init(0)
res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
fr, foff, src = api.synth_fragments(n_q, res, off, seed=5, mutation_rate=0.1, min_len=lens[0], max_len=lens[1], random_every=10)
work = "/tmp/diag_ref"; os.makedirs(work, exist_ok=True)
c5.write_lines(os.path.join(work, "targets.txt"), res, off)
c5.write_lines(os.path.join(work, "queries.txt"), fr, foff)
p = api.default_params()
p.kmer_size = 7
import ctypes
l2 = ctypes.CDLL(None).sysconf(191)
p.host_l2_bytes = l2 if l2 and l2 > 0 else 262144
db = api.TargetDB.from_codes(res, off, p)
del res
mat = oracle.write_matrix_files(os.path.join(work, "mat"))
t0 = time.time()
line = subprocess.check_output([oracle.REF, "pipeline", mat, os.path.join(work, "targets.txt"), os.path.join(work, "queries.txt"), os.path.join(work, "ref"),
                                "--threads", "16", "-k", "7", "--no-align"], stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
print("reference:", line, "%.1f s" % (time.time() - t0), flush=True)
ref = oracle.read_blocks(os.path.join(work, "ref", "pref.txt"))
for name, env in (("global", dict(MK_PREFILTER_PATH="global")), ("wide", dict(MK_PREFILTER_PATH="wide"))):
    for k in ("MK_PREFILTER_PATH", "MK_PREFILTER_K7_LISTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    q = api.Queries.from_codes(fr, foff, p)
    api.kernel_stats(reset=True)
    hits, hoff = api.prefilter(db, q, p)
    st = api.kernel_stats()
    bad = [i for i in range(n_q) if api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode() != ref[i]]
    print(name, "hits", int(hoff[-1]), "queries differing from the reference:", len(bad), bad[:10], {k: round(v["ms"]) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"])[:5]},
          {k: v["cells"] for k, v in st.items() if k == "prefilter_query_wide"}, flush=True)
    if bad:
        i = bad[0]
        a = api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode().split("\n"); b = ref[i].split("\n")
        d = [k for k in range(min(len(a), len(b))) if a[k] != b[k]][:3]
        print("   query", i, "length", int(foff[i + 1] - foff[i]), "first differing lines (ours | reference):", [(a[k], b[k]) for k in d], "counts", len(a), len(b), flush=True)
