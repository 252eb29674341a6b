cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --config4-profiles 0 > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err
  python - <<PY
import json
l=json.load(open("gpurun_out/x_$name.json"))
k=l["kernels_ms"]; st=l["steps"]
g=lambda p: round(sum(v for n,v in k.items() if n.startswith(p))/st,1)
print("$name", round(l["ms_per_step"],1), l["prefilter_hits"], l["alignments_passed"], "fwd",g("sw_fwd"),"pos",g("sw_pos"),"rev",g("sw_rev"), "pf", g("prefilter_query"), "helpers", round((k.get("diag_score",0)+k.get("select_hits",0)+k.get("sort_hits",0)+k.get("kmer_count",0)+k.get("double_hit",0))/st,1))
PY
}
run w12_c256 MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=262144
run w12_c256_ramp MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=262144 MK_SEARCH_CHUNK_RAMP=1
run w12_c384_ramp MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=393216 MK_SEARCH_CHUNK_RAMP=1
run w12_c512_ramp MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=524288 MK_SEARCH_CHUNK_RAMP=1
run w10_c256_ramp MK_SW_WAVES_PER_CU=10 MK_SEARCH_CHUNK_QUERIES=262144 MK_SEARCH_CHUNK_RAMP=1
run w14_c256_ramp MK_SW_WAVES_PER_CU=14 MK_SEARCH_CHUNK_QUERIES=262144 MK_SEARCH_CHUNK_RAMP=1
run w12_c256_ramp_pfS MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=262144 MK_SEARCH_CHUNK_RAMP=1 MK_PREFILTER_WG_PER_CU_S=12
run w12_c256_ramp_pfA3 MK_SW_WAVES_PER_CU=12 MK_SEARCH_CHUNK_QUERIES=262144 MK_SEARCH_CHUNK_RAMP=1 MK_PREFILTER_WG_PER_CU_A=3
