#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04m; mkdir -p $O
timeout 600 python tools/wide_classes_experiment.py 60000000 4000 > $O/classes_60m.txt 2>&1; tail -5 $O/classes_60m.txt | cut -c1-700
timeout 600 python tools/wide_classes_experiment.py 11800000 40000 > $O/classes_11m.txt 2>&1; tail -5 $O/classes_11m.txt | cut -c1-700
b() { tag=$1; shift; env MK_DEBUG=1 "$@" timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --config4-profiles 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; s=d['steps']
print('$tag: ms_per_step %.1f  roofline frac %.4f (%.1f ms/launch)  host_pf %.0f wait_align %.0f  sw_fwd %.0f pos %.0f rev %.0f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], k['host_prefilter_total']/s, k['wait_align']/s, sum(v for n,v in k.items() if n.startswith('sw_fwd'))/s, sum(v for n,v in k.items() if n.startswith('sw_pos'))/s, sum(v for n,v in k.items() if n.startswith('sw_rev'))/s))"; }
b base MK_X=0
b sw10 MK_SW_WAVES_PER_CU=10
b sw8 MK_SW_WAVES_PER_CU=8
b sw6 MK_SW_WAVES_PER_CU=6
b w3 MK_ALIGN_WORKERS=3
b sw8w3 MK_SW_WAVES_PER_CU=8 MK_ALIGN_WORKERS=3
b pfB3 MK_PREFILTER_WG_PER_CU_B=3
