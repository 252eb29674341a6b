#!/bin/bash
# What the prefilter's third tier (stream_kernel<32768,...>) and the small-tile score pass (swp_kernel<2|3|4,...>) contend for: the same counters with the
# two stages back to back (--two-calls: a kernel's counters are its own, "alone") and in the pipelined search (co-resident), one rocprofv3 --pmc pass per
# counter group (no trace domains beside --kernel-trace).   gpurun -- 'bash tools/contention_pmc.sh'   -> gpurun_out/contention/r06_contention.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/contention
RAW=/tmp/mk_contention_raw          # the counter CSVs are hundreds of MB: only the summary goes to gpurun_out/
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 1 --warmup 0 --config5-targets 0 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --blocking-steps 0 --alone-steps 0"
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"
G3="TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum"
G4="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
for mode in alone coresident; do
    EXTRA=""; [ $mode = alone ] && EXTRA="--two-calls"
    k=0
    for G in "$G1" "$G2" "$G3" "$G4"; do
        k=$((k + 1))
        rm -rf $RAW/${mode}_g$k
        timeout 400 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $RAW/${mode}_g$k -- python $R/bench.py $COMMON $EXTRA > $OUT/${mode}_g$k.log 2>&1 || echo "pass $mode g$k failed (see $OUT/${mode}_g$k.log)"
    done
done
python - <<PY
import csv, glob, collections, re
out, raw = "$OUT", "$RAW"
want = [("stream_kernel<32768", "prefilter third tier  stream_kernel<32768,2048,65536,1024,8,2>"), ("stream_kernel<131072", "prefilter largest tier stream_kernel<131072,...,16,2>"),
        ("stream_kernel<2048", "prefilter first tier  stream_kernel<2048,...,1,2>"), ("swp_kernel<2, 2, 16>", "score pass rows32  swp_kernel<2,2,16>"),
        ("swp_kernel<3, 4, 16>", "score pass rows48  swp_kernel<3,4,16>"), ("swp_kernel<4, 4, 16>", "score pass rows64  swp_kernel<4,4,16>"), ("swp_kernel<6, 6, 16>", "score pass rows96  swp_kernel<6,6,16>")]
data = {}
for mode in ("alone", "coresident"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.Counter()
    dur = collections.defaultdict(float)
    for g in range(1, 5):
        fs = sorted(glob.glob("%s/%s_g%d/**/*counter_collection.csv" % (raw, mode, g), recursive=True))
        if not fs:
            continue
        for r in csv.DictReader(open(fs[-1])):
            for key, _ in want:
                if key in r["Kernel_Name"]:
                    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                    if r["Counter_Name"] in ("SQ_WAVES",):
                        disp[key] += 1
        ks = sorted(glob.glob("%s/%s_g%d/**/*kernel_trace.csv" % (raw, mode, g), recursive=True))
        if ks and g == 1:
            for r in csv.DictReader(open(ks[-1])):
                for key, _ in want:
                    if key in r["Kernel_Name"]:
                        dur[key] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    data[mode] = (agg, disp, dur)
names = []
for mode in data:
    for key in data[mode][0]:
        for c in data[mode][0][key]:
            if c not in names:
                names.append(c)
with open(out + "/r06_contention.txt", "w") as w:
    w.write("# rocprofv3 --pmc (four passes per mode, --kernel-trace only) -- python bench.py --steps 1 --warmup 0 [--two-calls]; sums over all dispatches of the one pass of the workload\\n")
    w.write("# alone = mk_prefilter then mk_align (a kernel's counters are its own); coresident = the queued mk_search (the counters of a dispatch include what shares its CUs)\\n")
    for key, label in want:
        w.write("\\n== %s\\n" % label)
        w.write("%-34s %16s %16s %8s\\n" % ("counter", "alone", "co-resident", "ratio"))
        a, c = data["alone"], data["coresident"]
        w.write("%-34s %16.1f %16.1f %8.2f\\n" % ("kernel time, ms (pass g1)", a[2][key], c[2][key], c[2][key] / max(a[2][key], 1e-9)))
        w.write("%-34s %16d %16d\\n" % ("dispatches", a[1][key], c[1][key]))
        for n in names:
            va, vc = a[0][key].get(n), c[0][key].get(n)
            if va is None and vc is None:
                continue
            w.write("%-34s %16.4g %16.4g %8.2f\\n" % (n, va or 0, vc or 0, (vc or 0) / max(va or 0, 1e-9)))
        for mode, (agg, _, _) in data.items():
            v = agg[key]
            wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
            w.write("  [%s] per wave-cycle: VALU active %.3f, waiting on any instruction %.3f, on LDS %.3f; L2 hit rate %.3f; LDS bank-conflict cycles / LDS active %.3f; TCP pending-stall / TA busy %.3f\\n" % (
                mode, v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_WAIT_INST_LDS", 0) / wc,
                v.get("TCC_HIT_sum", 0) / max(v.get("TCC_REQ_sum", 0), 1), v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_ACTIVE_INST_LDS", 0), 1),
                v.get("TCP_PENDING_STALL_CYCLES_sum", 0) / max(v.get("TA_BUSY_sum", 0), 1)))
PY
rm -rf $RAW
cat $OUT/r06_contention.txt | head -150
