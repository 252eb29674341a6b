#!/bin/bash
# round 4, GPU call L: mixed partition hashes: parity (wide / k7 / profile / scale), config-5 scale profile, quick bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04l; mkdir -p $O
export MK_DEBUG=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k7.py tests/test_gpu_profile.py tests/test_gpu_scale.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "wide or k7 or profile or config5 or headline" -s > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
grep "config-5 full scale" $O/pytest.txt | cut -c1-1300 | tail -1; tail -4 $O/pytest.txt
rm -rf /tmp/pytest-of-root
run_c5() {
  tag=$1; shift
  env "$@" MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 100000 > $O/config5_search_$tag.json 2> $O/config5_search_$tag.err; echo "c5 profile $tag rc $?"
  grep "wide piece" $O/config5_search_$tag.err | tail -1
  python - $tag <<'P'
import json, sys
try:
    d=json.load(open("gpurun_out/r04l/config5_search_%s.json" % sys.argv[1]))
    for r in d["runs"]:
        print(sys.argv[1], r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:12]})
except Exception as e:
    print("no config5 profile:", e)
P
}
run_c5 base MK_X=0
run_c5 wg1 MK_PREFILTER_WG_PER_CU_W=1
unset MK_DEBUG
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-sample 0 > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc $?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r04l/bench_quick.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    c4 = d.get("config4_profile_targets", {})
    print("config4", c4.get("s_per_pass"), c4.get("result_digest", {}).get("match"), {k: v for k, v in c4.get("kernels_ms", {}).items() if not k.startswith("sw_")})
except Exception as e:
    print("no bench line:", e)
P
