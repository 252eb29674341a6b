#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py::test_headline_scale_parity -x -q -m gpu 2>&1 | tail -4
run() { timeout 90 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-profiles 0 2>gpurun_out/run_$1.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; st=d['steps']
print('$1', round(d['ms_per_step'],1), {n: round(k.get(n,0)/st,1) for n in ('host_prefilter_total','wait_align','host_align_total')}, 'pos/rev', round(sum(v for n,v in k.items() if n.startswith('sw_pos') or n.startswith('sw_rev'))/st,1), 'fwd', round(sum(v for n,v in k.items() if n.startswith('sw_fwd'))/st,1), 'stream', round(sum(v for n,v in k.items() if n.startswith('prefilter_query'))/st,1))" || tail -5 gpurun_out/run_$1.err; }
for i in 1 2 3 4 5 6; do run w2_$i; done
MK_ALIGN_WORKERS=1 run w1
MK_ALIGN_WORKERS=3 run w3
