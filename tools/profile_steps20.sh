#!/bin/bash
# rocprofv3 kernel statistics of the headline region with the driver's step counts (legs off): gpurun -- 'bash tools/profile_steps20.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof20; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof20
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof20 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --config5-targets 0 --blocking-steps 0 --alone-steps 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import csv, glob, json
f = sorted(glob.glob("/tmp/prof20/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
with open("$OUT/r06_bench_kernel_stats_steps20.txt", "w") as w:
    w.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --config5-targets 0 --blocking-steps 0 --alone-steps 0\n")
    w.write("# (the driver's step counts, the legs beside the headline off; 25 passes of the workload.)  The same run's bench line: ms_per_step %.1f, roofline kernel %s avg_launch_ms %.3f over %d launches (HIP events, timed steps only)\n" % (d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"]))
    w.write("%-96s %8s %12s %10s %7s\n" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in rows[:40]:
        w.write("%-96s %8d %12.2f %10.3f %7.2f\n" % (r["Name"].replace("(anonymous namespace)::", "")[:96], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
print(open("$OUT/r06_bench_kernel_stats_steps20.txt").read()[:1500])
PY
rm -rf /tmp/prof20
