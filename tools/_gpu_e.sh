#!/bin/bash
# round 4, GPU call E: wide kernel v4 (runtime key bits, exact-target filter): parity, 60 M proteins, config-5 scale variants; shard-size variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
export MK_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k7.py tests/test_gpu_profile.py -m gpu -q --maxfail=12 -k "wide or k7 or profile" -p no:cacheprovider > $O/pytest_wide.txt 2>&1; echo "pytest wide rc $?" >> $O/pytest_wide.txt
tail -6 $O/pytest_wide.txt
rm -rf /tmp/pytest-of-root
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "60M or 4e9" -p no:cacheprovider -s > $O/pytest_60m.txt 2>&1; echo "pytest 60M rc $?" >> $O/pytest_60m.txt
grep "config-5 full scale" $O/pytest_60m.txt | cut -c1-1700; tail -3 $O/pytest_60m.txt
rm -rf /tmp/pytest-of-root
run_c5() {
  tag=$1; shift
  env "$@" MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 100000 > $O/config5_search_$tag.json 2> $O/config5_search_$tag.err; echo "c5 profile $tag rc $?"
  grep "wide piece" $O/config5_search_$tag.err | tail -1
  python - $tag <<'P'
import json, sys
try:
    d=json.load(open("gpurun_out/r04e/config5_search_%s.json" % sys.argv[1]))
    for r in d["runs"]:
        print(sys.argv[1], r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:12]})
except Exception as e:
    print("no config5 profile:", e)
P
}
run_c5 base MK_X=0
run_c5 wg2 MK_PREFILTER_WG_PER_CU_W=2
run_c5 regs128 MK_PREFILTER_WIDE_REGS=128
unset MK_DEBUG
for chunk in 65536 131072; do for tail in 1 2; do
  MK_DEBUG=1 MK_SEARCH_CHUNK_QUERIES=$chunk MK_ALIGN_TAIL_PIECES=$tail timeout 300 python bench.py --contigs 1250 --steps 6 --warmup 2 --cpu-sample 0 --config4-profiles 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard 1250 contigs chunk $chunk tail $tail: ms_per_step %.1f' % d['ms_per_step'])"
done; done
