#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
b() { tag=$1; shift; env MK_DEBUG=1 "$@" timeout 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config4_profile_targets']; k=c['kernels_ms']
print('$tag: config2 ms_per_step %.1f | config4 s_per_pass %.3f (first %.3f) host_pf %.0f wait_align %.0f wide %.0f lists %.0f' % (d['ms_per_step'], c['s_per_pass'], c['first_pass_s'], k.get('host_prefilter_total',0), k.get('wait_align',0), k.get('prefilter_query_wide',0), k.get('profile_kmer_count',0)+k.get('profile_kmer_fill',0)))"; }
b base MK_X=0
b w3 MK_ALIGN_WORKERS=3
b w4 MK_ALIGN_WORKERS=4
b w3c16k MK_ALIGN_WORKERS=3 MK_SEARCH_PROFILE_CHUNK=16384
b w3c4k MK_ALIGN_WORKERS=3 MK_SEARCH_PROFILE_CHUNK=4096
