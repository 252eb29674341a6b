#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
b() { tag=$1; shift; env MK_DEBUG=1 "$@" timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --config4-profiles 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms']; s=d['steps']
print('$tag: ms_per_step %.1f  roofline frac %.4f (%.1f ms/launch)  host_pf %.0f wait_align %.0f  sw_fwd %.0f pos %.0f rev %.0f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], k['host_prefilter_total']/s, k['wait_align']/s, sum(v for n,v in k.items() if n.startswith('sw_fwd'))/s, sum(v for n,v in k.items() if n.startswith('sw_pos'))/s, sum(v for n,v in k.items() if n.startswith('sw_rev'))/s))"; }
b base MK_X=0
b known12 MK_SW_KNOWN=1
b known8 MK_SW_KNOWN=1 MK_SW_KNOWN_WAVES=8
b known6 MK_SW_KNOWN=1 MK_SW_KNOWN_WAVES=6
b known16 MK_SW_KNOWN=1 MK_SW_KNOWN_WAVES=16
