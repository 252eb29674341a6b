#!/bin/bash
# L1->L2 requests, L2 hits/misses and fabric read requests per kernel (two stages back to back), in separate counter passes.
#   gpurun -- 'bash tools/mem_profile.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --config5-targets 0 --config4-profiles 0 --e2e-sample -1 --two-calls ${BENCH_EXTRA:-}"
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_READ_SECTORS_sum --kernel-trace --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1
rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-trace --output-format csv -d $OUT/p3 -- $BENCH > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); names = []
for p in ("p1", "p2", "p3"):
    fs = sorted(glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void mk::", "")[:52]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
with open(out + "/mem_summary.txt", "w") as w:
    w.write("# rocprofv3 --pmc (three passes) -- bench.py --steps 1 --warmup 1 --two-calls; sums over all dispatches of both passes\n")
    w.write("kernel".ljust(54) + "".join(x.replace("_sum", "")[-18:].rjust(19) for x in names) + "\n")
    for k, v in sorted(agg.items(), key=lambda x: -x[1].get("TCC_REQ_sum", 0))[:16]:
        w.write(k.ljust(54) + "".join(("%.4g" % v.get(x, 0)).rjust(19) for x in names) + "\n")
PY
cat $OUT/mem_summary.txt
