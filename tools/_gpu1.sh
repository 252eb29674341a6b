cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in default mw6 mw4; do
  if [ $v != default ]; then export METAEUK_AMD_LIB=$PWD/metaeuk_amd/lib/variants/lib$v.so; else unset METAEUK_AMD_LIB; fi
  for mode in "" "--two-calls"; do
    timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-profiles 0 $mode > gpurun_out/v_${v}_${mode#--}.json 2> gpurun_out/v_${v}_${mode#--}.err
    python - <<PY
import json
l=json.load(open("gpurun_out/v_${v}_${mode#--}.json"))
k=l["kernels_ms"]; st=l["steps"]
print("$v", "$mode", round(l["ms_per_step"],1), l["prefilter_hits"], l["alignments_passed"], {n: round(k[n]/st,1) for n in k if n.startswith("prefilter_query") or n in ("host_prefilter_total","host_align_total","kmer_count")})
PY
  done
done
