#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_profile.py tests/test_gpu_k7.py tests/test_gpu_scale.py::test_headline_scale_parity tests/test_gpu_scale.py::test_config5_shape_1e9_residues_pinned_to_the_reference tests/test_gpu_scale.py::test_profile_path_scale_parity -x -q -m gpu 2>&1 | tail -4
python tools/config5_search_profile.py 11800000 100000 > gpurun_out/config5_search3.json 2> gpurun_out/config5_search3.err; tail -2 gpurun_out/config5_search3.err
timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --config4-sample 64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms']; st=d['steps']
print(round(d['ms_per_step'],1), {n: round(k.get(n,0)/st,1) for n in ('host_prefilter_total','sort_hits','pair_filter','double_hit','kmer_probe_gather','kmer_probe_count')})
c=d['config4_profile_targets']; kk=c['kernels_ms']; print(c['s_per_pass'], c['result_digest']['match'], {n: round(kk.get(n,0),1) for n in ('host_prefilter_total','sort_hits','pair_filter','double_hit','kmer_probe_gather','kmer_probe_count','host_search_total')})"
