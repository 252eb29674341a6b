#!/bin/bash
# Kernel timeline of one mk_search pass: how much of the time both stages have a kernel on the GPU, per-kernel stretch.
#   gpurun -- 'bash tools/overlap_trace.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --config4-profiles 0 --config5-targets 0 --alone-steps 0 --e2e-sample -1 --blocking-steps 0 ${BENCH_EXTRA:-} > $OUT/kt.log 2>&1
python - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    al = ("sw_kernel", "swp_kernel", "swq_kernel", "gate_count", "gate_emit", "expand_pairs", "rev_jobs", "collect_kernel", "plan_sums", "plan_offsets",
          "order_wave", "order_block", "bin_scan", "bin_scatter", "assemble", "scan_sums", "scan_apply")
    pf = ("stream_kernel", "wide_kernel", "kmer_count", "diag_score", "finish_", "keep_", "double_hit", "select_", "derive_kernel", "kthr_kernel",
          "probe_", "global_", "sort_", "hits_")
    stage = "align" if any(k in n for k in al) else ("pref" if any(k in n for k in pf) else "other")
    ev.append((s, e, stage, n, r.get("Queue_Id", "")))
ev.sort()
t0 = ev[len(ev) // 2][0]      # second half = the timed step (roughly)
half = [x for x in ev if x[0] >= t0]
lo, hi = min(x[0] for x in half), max(x[1] for x in half)
# sweep line: time with >=1 pref kernel, >=1 align kernel, both
pts = []
for s, e, st, n, q in half:
    if st in ("pref", "align"): pts.append((s, 1, st)); pts.append((e, -1, st))
pts.sort()
cnt = {"pref": 0, "align": 0}; last = lo; acc = collections.Counter()
for t, d, st in pts:
    key = ("P" if cnt["pref"] else "-") + ("A" if cnt["align"] else "-")
    acc[key] += t - last; last = t
    cnt[st] += d
tot = hi - lo
print("window %.1f ms: only-prefilter %.1f  only-align %.1f  both %.1f  neither %.1f" % (tot / 1e6, acc["P-"] / 1e6, acc["-A"] / 1e6, acc["PA"] / 1e6, (tot - acc["P-"] - acc["-A"] - acc["PA"]) / 1e6))
dur = collections.defaultdict(lambda: [0, 0.0])
for s, e, st, n, q in half:
    k = n.replace("(anonymous namespace)::", "").replace("void mk::", "")[:48]
    dur[k][0] += 1; dur[k][1] += (e - s) / 1e6
for k, v in sorted(dur.items(), key=lambda x: -x[1][1])[:16]: print("%-50s %5d %9.2f ms" % (k, v[0], v[1]))
queues = collections.Counter((q, st) for s, e, st, n, q in half)
print(dict(queues))
PY
