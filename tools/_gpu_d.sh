#!/bin/bash
# round 4, GPU call D: full GPU suite (incl. the 60 M-protein case), wide kernel v3 at config-5 scale (64 / 128 registers), partition experiment, quick bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; mkdir -p $O
export MK_DEBUG=1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider --deselect tests/test_gpu_scale.py::test_config5_full_scale_60M_proteins_on_one_gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -12 $O/pytest.txt
rm -rf /tmp/pytest-of-root
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "60M" -p no:cacheprovider -s > $O/pytest_60m.txt 2>&1; echo "pytest 60M rc $?" >> $O/pytest_60m.txt
tail -6 $O/pytest_60m.txt | cut -c1-1500
rm -rf /tmp/pytest-of-root
for regs in 64 128; do
  MK_PREFILTER_WIDE_REGS=$regs MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 100000 > $O/config5_search_r$regs.json 2> $O/config5_search_r$regs.err; echo "c5 profile regs $regs rc $?"
  grep "wide piece" $O/config5_search_r$regs.err | tail -2
  python - $regs <<'P'
import json, sys
try:
    d=json.load(open("gpurun_out/r04d/config5_search_r%s.json" % sys.argv[1]))
    for r in d["runs"]:
        print(r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:14]})
except Exception as e:
    print("no config5 profile:", e)
P
done
timeout 600 python tools/partition_probe_experiment.py 131072 1000 > $O/partition_probe.json 2> $O/partition_probe.err; echo "partition probe rc $?"; tail -3 $O/partition_probe.err
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/r04d/partition_probe.json"))
    print({k: v for k, v in d.items() if k != "partitioned"})
    for c in d["partitioned"]:
        print(c["partitions"], c["record_bytes"], "B: count %.2f partition %.2f probe %.2f total %.2f ms = %.2fx direct (probe alone %.2fx), skew %.2f" % (c["count_ms"], c["partition_ms"], c["probe_ms"], c["total_ms"], c["vs_direct"], c["probe_alone_vs_direct"], c["largest_partition_over_mean"]))
except Exception as e:
    print("no partition probe result:", e)
P
unset MK_DEBUG
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc $?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r04d/bench_quick.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    c4 = d.get("config4_profile_targets", {})
    print("config4", c4.get("s_per_pass"), c4.get("result_digest", {}).get("match"), {k: v for k, v in c4.get("kernels_ms", {}).items() if not k.startswith("sw_")})
    print({k: round(v / d["steps"], 1) for k, v in d["kernels_ms"].items() if not k.startswith("sw_")})
except Exception as e:
    print("no bench line:", e)
P
