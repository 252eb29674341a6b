#!/bin/bash
# round 4, GPU call C: wide kernel v2 (16 classes, subsets, in-kernel 7-mers) through the tests; config-5 scale profile with tick counters;
# the partitioned-probe experiment
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04c; mkdir -p $O
export MK_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k7.py tests/test_gpu_profile.py -m gpu -q --maxfail=12 -k "wide or k7 or profile" -p no:cacheprovider > $O/pytest_wide.txt 2>&1; echo "pytest wide rc $?" >> $O/pytest_wide.txt
tail -25 $O/pytest_wide.txt
rm -rf /tmp/pytest-of-root
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "config5" -p no:cacheprovider > $O/pytest_c5.txt 2>&1; echo "pytest c5 rc $?" >> $O/pytest_c5.txt
tail -8 $O/pytest_c5.txt
rm -rf /tmp/pytest-of-root
for mode in enum7 lists; do
  if [ $mode = lists ]; then export MK_PREFILTER_K7_LISTS=1; else unset MK_PREFILTER_K7_LISTS; fi
  MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 20000 100000 > $O/config5_search_$mode.json 2> $O/config5_search_$mode.err; echo "c5 profile $mode rc $?"
  grep "wide piece" $O/config5_search_$mode.err | tail -4
  python - $mode <<'P'
import json, sys
try:
    d=json.load(open("gpurun_out/r04c/config5_search_%s.json" % sys.argv[1]))
    for r in d["runs"]:
        print(r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:12]})
except Exception as e:
    print("no config5 profile:", e)
P
done
unset MK_PREFILTER_K7_LISTS
timeout 600 python tools/partition_probe_experiment.py 131072 1000 > $O/partition_probe.json 2> $O/partition_probe.err; echo "partition probe rc $?"; tail -3 $O/partition_probe.err
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/r04c/partition_probe.json"))
    print({k: v for k, v in d.items() if k != "partitioned"})
    for c in d["partitioned"]:
        print(c["partitions"], c["record_bytes"], "B: partition %.2f ms probe %.2f ms total %.2f ms = %.2fx direct, skew %.2f ovf %d" % (c["partition_ms"], c["probe_ms"], c["total_ms"], c["vs_direct"], c["largest_partition_over_mean"], c["overflowing_reservations"]))
except Exception as e:
    print("no partition probe result:", e)
P
