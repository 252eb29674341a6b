#!/bin/bash
# HBM traffic per kernel from the L2's fabric-side request counters, in their own rocprofv3 passes (no API trace domains), written as
# the JSON artefact bench.py reads for roofline.traffic:  profiles/rNN_pmc_hbm_traffic.json
#   gpurun -- 'bash tools/pmc_traffic.sh r02'        (then copy gpurun_out/pmc/rNN_pmc_hbm_traffic.json to profiles/)
# Bytes = 128 B x 128-byte requests + 32 B x 32-byte requests + 64 B x the rest (reads); 64 B x 64-byte requests + 32 B x the rest
# (writes).  FETCH_SIZE is NOT used: it tallies a 128-byte request as 64 B (profiles/r02_random_probe_rates.txt).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --config5-targets 0 --config4-profiles 0 --blocking-steps 0 --alone-steps 0 --e2e-sample -1"
rm -rf $OUT/rd $OUT/wr $OUT/dram
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/rd -- $BENCH > $OUT/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/wr -- $BENCH > $OUT/wr.log 2>&1
# round 5: how many of those fabric requests are bound for the memory controllers (TCC_EA0_RDREQ_DRAM / WRREQ_DRAM: "destined for DRAM (MC)" -- the
# Infinity Cache sits on the memory side of the fabric, so its hits are INSIDE this count; rocprofv3 -L on gfx950 lists no MALL hit counter) and how
# often the read interface ran out of DRAM credits
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --kernel-trace --output-format csv -d $OUT/dram -- $BENCH > $OUT/dram.log 2>&1
python - <<PY
import csv, glob, collections, hashlib, json, re
out, root = "$OUT", "$R"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
def name_of(k):
    m = re.search(r"stream_kernel<(\d+),", k)
    if m: return "prefilter_query_cap" + m.group(1)
    if "wide_kernel<" in k: return "prefilter_query_wide"
    if "probe_kernel<true" in k: return "kmer_probe_gather"
    if "probe_kernel<false" in k: return "kmer_probe_count"
    if "kmer_count_kernel" in k: return "kmer_count"
    if "diag_score_kernel" in k: return "diag_score"
    m = re.search(r"swp_kernel<(\d+), (\d+), (\d+)>", k)
    if m: return "sw_fwd_rows%d" % {(2, 2, 16): 32, (3, 4, 16): 48, (4, 4, 16): 64, (6, 6, 16): 96, (8, 8, 16): 128, (12, 12, 16): 192, (16, 16, 16): 256, (12, 12, 32): 384, (16, 16, 32): 512}.get(tuple(int(x) for x in m.groups()), 0)
    return re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void mk::", "").replace("mk::", ""))[:48]
for d in ("rd", "wr", "dram"):
    fs = sorted(glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        k = name_of(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_EA0_RDREQ_sum": n[k] += 1
h = hashlib.sha256()
for f in sorted(glob.glob(root + "/metaeuk_amd/csrc/*.hip") + glob.glob(root + "/metaeuk_amd/csrc/mk_enum.hpp")): h.update(open(f, "rb").read())
kern = {}
for k, v in agg.items():
    rd, r32, r128 = v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCC_EA0_RDREQ_32B_sum", 0), v.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr, w64 = v.get("TCC_EA0_WRREQ_sum", 0), v.get("TCC_EA0_WRREQ_64B_sum", 0)
    kern[k] = {"launches": n[k], "fetch_bytes": 128 * r128 + 32 * r32 + 64 * max(rd - r128 - r32, 0), "write_bytes": 64 * w64 + 32 * max(wr - w64, 0),
               "read_requests": rd, "read_requests_128B": r128, "write_requests": wr,
               "read_requests_dram": v.get("TCC_EA0_RDREQ_DRAM_sum"), "write_requests_dram": v.get("TCC_EA0_WRREQ_DRAM_sum"),
               "read_dram_credit_stall_cycles": v.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum")}
art = {"build": h.hexdigest()[:16], "passes": 2, "command": "rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum (pass 1) / TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (pass 2) --kernel-trace -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --config4-profiles 0",
       "note": "sums over all dispatches of a pass of the run (warm-up + 1 step; every pass is its own run of the same command); bench.py divides by launches. "
               "read_requests_dram / write_requests_dram: requests bound for the memory controllers (Infinity Cache hits included: no counter separates them)", "kernels": kern}
json.dump(art, open(out + "/${TAG}_pmc_hbm_traffic.json", "w"), indent=1)
for k, v in sorted(kern.items(), key=lambda x: -x[1]["fetch_bytes"])[:12]:
    print("%-32s launches %5d  read %8.2f GB/launch  write %7.2f GB/launch  reads bound for DRAM/MALL %s" % (k, v["launches"], v["fetch_bytes"] / max(v["launches"], 1) / 1e9, v["write_bytes"] / max(v["launches"], 1) / 1e9,
          "%.3f of the read requests" % (v["read_requests_dram"] / v["read_requests"]) if v.get("read_requests_dram") is not None and v["read_requests"] else "-"))
PY
