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
print("$name", round(l["ms_per_step"],1), l["prefilter_hits"], l["alignments_passed"], "fwd",g("sw_fwd"),"pos",g("sw_pos"),"rev",g("sw_rev"), "pf", g("prefilter_query"), "helpers", round((k.get("diag_score",0)+k.get("select_hits",0)+k.get("sort_hits",0)+k.get("kmer_count",0)+k.get("double_hit",0))/st,1), "align_", g("align_"))
PY
}
run base A=1
run known MK_SW_KNOWN=1
run known_w8 MK_SW_KNOWN=1 MK_SW_KNOWN_WAVES=8
run known_w16 MK_SW_KNOWN=1 MK_SW_KNOWN_WAVES=16
