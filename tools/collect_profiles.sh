#!/bin/bash
# Round evidence on a GPU box: the default bench line, rocprofv3 kernel statistics of the same command, the HBM traffic artefact
# (tools/pmc_traffic.sh), SQ counters, the two-stage timeline and the host-thread sensitivity.  Writes under gpurun_out/evidence/ ; copy
# the files to be judged into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'QUICK=1 bash tools/collect_profiles.sh r04'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/evidence
mkdir -p $OUT
cd $R
# the HBM traffic of THIS build first: bench.py reads profiles/<tag>_pmc_hbm_traffic.json for roofline.traffic
bash tools/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; cp gpurun_out/pmc/${TAG}_pmc_hbm_traffic.json $OUT/ 2>/dev/null; cp gpurun_out/pmc/${TAG}_pmc_hbm_traffic.json profiles/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1.json 2> $OUT/bench.err
X="--config4-profiles 0 --config5-targets 0 --alone-steps 0 --e2e-sample -1 --blocking-steps 0"
[ -n "${QUICK:-}" ] || for th in 16 2; do MK_HOST_THREADS=$th python bench.py --steps 2 --warmup 1 --cpu-sample 0 $X 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('MK_HOST_THREADS=$th  ms_per_step %.1f  fragments/s %.0f  host phases (ms/step): %s' % (d['ms_per_step'], d['value'], {k: round(v / d['steps'], 1) for k, v in d['kernels_ms'].items() if k.startswith('host_')}))"; done > $OUT/${TAG}_bench_host_threads.txt
python bench.py --steps 2 --warmup 1 --cpu-sample 0 --two-calls $X > $OUT/${TAG}_bench_two_calls.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 $X > $OUT/prof_stats.log 2>&1
python - <<PY
import csv, glob
out = "$OUT"
f = sorted(glob.glob(out + "/prof_stats/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
with open(out + "/${TAG}_bench_kernel_stats.txt", "w") as w:
    w.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --blocking-steps 0   (3 passes: warm-up + 2 steps, 10 chunks each)\n")
    w.write("%-96s %8s %12s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows[:56]:
        w.write("%-96s %8d %12.2f %10.3f %7.2f\n" % (r["Name"].replace("(anonymous namespace)::", "")[:96], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
cd $R
if [ -z "${QUICK:-}" ]; then
bash tools/sq_profile.sh > /dev/null 2>&1; cp gpurun_out/sq/sq_summary.txt $OUT/${TAG}_sq_counters.txt 2>/dev/null
bash tools/overlap_trace.sh 2>&1 | tail -22 > $OUT/${TAG}_stage_timeline.txt
bash tools/mem_profile.sh > /dev/null 2>&1; cp gpurun_out/mem/mem_summary.txt $OUT/${TAG}_mem_counters.txt 2>/dev/null
fi
rm -rf $OUT/prof_stats gpurun_out/pmc/rd gpurun_out/pmc/wr gpurun_out/pmc/dram gpurun_out/sq/pmc gpurun_out/trace/kt gpurun_out/mem/p1 gpurun_out/mem/p2 gpurun_out/mem/p3
head -12 $OUT/${TAG}_bench_kernel_stats.txt; cat $OUT/${TAG}_bench_host_threads.txt 2>/dev/null; tail -14 $OUT/pmc_traffic.log; python -c "import json;d=json.load(open('$OUT/${TAG}_bench_n1.json'));print(d['ms_per_step'], d['value'], d['roofline'], d['valu_roofline'], d['cpu_baseline'], d['result_digest'])"
