#!/bin/bash
# Who spends the vector-ALU issue slots of a config-2 step: SQ_INSTS_VALU (wave instructions) of EVERY kernel of one pass of the headline workload, one
# rocprofv3 --pmc pass (dispatches serialised: a kernel's counters are its own).   gpurun -- 'bash tools/valu_budget.sh'  -> gpurun_out/valu_budget/r06_valu_budget.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/valu_budget
RAW=/tmp/mk_valu_raw
mkdir -p $OUT; rm -rf $RAW; mkdir -p $RAW
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --config5-targets 0 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --blocking-steps 0 --alone-steps 0"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $RAW -- python $R/bench.py $COMMON > $OUT/pass.log 2>&1 || echo "pass failed (see $OUT/pass.log)"
python - <<PY
import csv, glob, collections, re
raw, out = "$RAW", "$OUT"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter(); dur = collections.defaultdict(float)
fs = sorted(glob.glob(raw + "/**/*counter_collection.csv", recursive=True))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|mk::|void ", "", n)
    return re.sub(r"\(.*\)$", "", n)[:70]
for r in csv.DictReader(open(fs[-1])):
    k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": disp[k] += 1
for r in csv.DictReader(open(sorted(glob.glob(raw + "/**/*kernel_trace.csv", recursive=True))[-1])):
    dur[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
tot = sum(a["SQ_INSTS_VALU"] for a in agg.values())
with open(out + "/r06_valu_budget.txt", "w") as w:
    w.write("# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace -- python bench.py --steps 1 --warmup 0 ...: ONE pass of the config-2 workload\n")
    w.write("# wave instructions per kernel (all dispatches); ms = serialised kernel time under the profiler; issue_ms = VALU instructions / 6.1e11 per s (1 024 SIMDs, one per 4 cycles at 2.4 GHz: profiles/r02_valu_issue_rates.txt)\n")
    w.write("%-72s %6s %9s %10s %10s %10s %10s %9s %6s\n" % ("kernel", "disp", "ms", "VALU", "SALU", "LDS", "VMEM", "issue_ms", "share"))
    for k, a in sorted(agg.items(), key=lambda x: -x[1]["SQ_INSTS_VALU"]):
        w.write("%-72s %6d %9.1f %10.3e %10.3e %10.3e %10.3e %9.1f %6.3f\n" % (k, disp[k], dur[k], a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_LDS"], a["SQ_INSTS_VMEM_RD"] + a["SQ_INSTS_VMEM_WR"], a["SQ_INSTS_VALU"] / 6.1e8, a["SQ_INSTS_VALU"] / max(tot, 1)))
    w.write("%-72s %6d %9.1f %10.3e %50s %9.1f\n" % ("total", sum(disp.values()), sum(dur.values()), tot, "", tot / 6.1e8))
print(open(out + "/r06_valu_budget.txt").read())
PY
rm -rf $RAW
