#!/bin/bash
# A/B of library variants on ONE box: tools/ab_variants.sh <out dir under gpurun_out> <variant names ...>   ("product" = the in-tree library)
O=gpurun_out/$1; shift
mkdir -p $O
ARGS=${AB_ARGS:---steps 8 --warmup 3 --cpu-sample 0 --config4-profiles 0 --e2e-sample -1 --config5-targets 0 --blocking-steps 0}
for v in "$@"; do
  # name[.tag][@VAR=value[,VAR=value]]: environment of the run after the @ (separate with + instead when a value holds commas)
  envs=""; case $v in *@*) envs=${v#*@}; v=${v%%@*};; esac
  lib=${v%%.*}
  if [ $lib = product ]; then unset METAEUK_AMD_LIB; else export METAEUK_AMD_LIB=$PWD/tools/_variants/libmetaeuk_amd_$lib.so; fi
  v=$v${envs:+@$envs}
  env $(case "$envs" in *+*) echo $envs | tr "+" " ";; *) echo $envs | tr "," " ";; esac) python bench.py $ARGS > $O/bench_$v.json 2> $O/bench_$v.err
  python - $O/bench_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels_ms"]; s = d["steps"]
print("%-12s step %7.1f ms  hits %d alns %d  tiers: %s  alone third %s" % (sys.argv[2], d["ms_per_step"], d["prefilter_hits"], d["alignments_passed"],
      " ".join("%s %.1f" % (n[len("prefilter_query_"):], k[n] / s) for n in sorted(k) if n.startswith("prefilter_query_")), d["roofline"].get("alone", {}).get("kernel_ms_per_step")))
PY
done
