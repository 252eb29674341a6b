#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-sample 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; st=d['steps']
print('$1', round(d['ms_per_step'],1), {n: round(k.get(n,0)/st,1) for n in ('host_prefilter_total','sort_hits','double_hit','kmer_probe_gather','kmer_probe_count')})
c=d['config4_profile_targets']; kk=c['kernels_ms']; print('   config4', c['s_per_pass'], c['result_digest']['match'], {n: round(kk.get(n,0),1) for n in ('host_prefilter_total','sort_hits','double_hit','kmer_probe_gather','kmer_probe_count','host_search_total')})"; }
run seg
MK_PREFILTER_SEGSORT=0 run whole
