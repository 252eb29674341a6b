#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
run() { timeout 90 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-profiles 0 $2 2>gpurun_out/run_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; st=d['steps']
print('$1', round(d['ms_per_step'],1), {n: round(k.get(n,0)/st,1) for n in ('host_prefilter_total','sort_hits','kmer_probe_count','kmer_probe_gather','double_hit')}, 'stream', {n[16:]: round(v/st,1) for n,v in k.items() if n.startswith('prefilter_query')})" || tail -5 gpurun_out/run_$1.err; }
run base
METAEUK_AMD_LIB=$R/metaeuk_amd/lib/variants/libcapB.so run capB
run base2
METAEUK_AMD_LIB=$R/metaeuk_amd/lib/variants/libcapB.so run capB2
