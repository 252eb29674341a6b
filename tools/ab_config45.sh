#!/bin/bash
# config-4 and config-5 legs of bench.py for library variants on one box: tools/ab_config45.sh <out dir> <variants...>
O=gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  if [ $v = product ]; then unset METAEUK_AMD_LIB; else export METAEUK_AMD_LIB=$PWD/tools/_variants/libmetaeuk_amd_$v.so; fi
  python bench.py --steps 1 --warmup 0 --cpu-sample 0 --config4-profiles 50000 --config4-sample 0 --e2e-sample -1 --blocking-steps 0 --alone-steps 0 --config5-digest 0 > $O/b_$v.json 2> $O/b_$v.err
  python - $O/b_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c4 = d.get("config4_profile_targets", {}); c5 = d.get("config5_60M", {})
print("%-8s c4 s/pass %s wide %s | c5 frag/s %s pass %s wide %s hits %s" % (sys.argv[2], c4.get("s_per_pass"), c4.get("kernels_ms", {}).get("prefilter_query_wide"), c5.get("fragments_per_s"), c5.get("s_per_pass_warm"),
      c5.get("kernels_ms", {}).get("prefilter_query_wide"), c5.get("prefilter_hits")))
PY
done
