#!/bin/bash
# SQ counters per kernel (VALU share, wait share, occupancy) of one bench pass with the two stages run back to back
# (no overlap, so the counters of a kernel are its own):  gpurun -- 'bash tools/sq_profile.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --config5-targets 0 --config4-profiles 0 --e2e-sample -1 --blocking-steps 0 ${BENCH_MODE---two-calls} ${BENCH_EXTRA:-}"
rm -rf $OUT/pmc
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc -- $BENCH > $OUT/pmc.log 2>&1
python - <<PY
import csv, glob, collections
out = "$OUT"
f = sorted(glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void mk::", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
with open(out + "/sq_summary.txt", "w") as w:
    w.write("# rocprofv3 --pmc " + " ".join(names) + " -- bench.py --steps 1 --warmup 1 --two-calls (summed over dispatches incl. warm-up)\n")
    w.write("kernel".ljust(62) + "disp".rjust(6) + "".join(x[3:].rjust(18) for x in names) + "  valu_act/wave_cyc  wait/wave_cyc  insts_valu/wave\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1]["SQ_WAVE_CYCLES"])[:30]:
        wc = max(v["SQ_WAVE_CYCLES"], 1)
        w.write(k.ljust(62) + ("%d" % n[k]).rjust(6) + "".join(("%.4g" % v[x]).rjust(18) for x in names) +
                "  %17.3f  %13.3f  %15.1f\n" % (v["SQ_ACTIVE_INST_VALU"] / wc, v["SQ_WAIT_ANY"] / wc, v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))
PY
cut -c1-330 $OUT/sq_summary.txt | head -32
