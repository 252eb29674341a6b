#!/bin/bash
# round 4, GPU call K: full suite incl. the 60 M-protein case after the partition-function fix; config-5 scale profile (1 and 2 workgroups per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04k; mkdir -p $O
export MK_DEBUG=1
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -s > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
grep "config-5 full scale" $O/pytest.txt | cut -c1-1800; tail -5 $O/pytest.txt
rm -rf /tmp/pytest-of-root
run_c5() {
  tag=$1; shift
  env "$@" MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 100000 > $O/config5_search_$tag.json 2> $O/config5_search_$tag.err; echo "c5 profile $tag rc $?"
  grep "wide piece" $O/config5_search_$tag.err | tail -1
  python - $tag <<'P'
import json, sys
try:
    d=json.load(open("gpurun_out/r04k/config5_search_%s.json" % sys.argv[1]))
    for r in d["runs"]:
        print(sys.argv[1], r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:12]})
except Exception as e:
    print("no config5 profile:", e)
P
}
run_c5 base MK_X=0
run_c5 wg2 MK_PREFILTER_WG_PER_CU_W=2
