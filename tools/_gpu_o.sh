#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04o; mkdir -p $O
export MK_DEBUG=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k7.py tests/test_gpu_profile.py tests/test_gpu_scale.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "wide or k7 or profile or config5 or headline or heavy" -s > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
grep "config-5 full scale" $O/pytest.txt | cut -c1-1300 | tail -1; tail -4 $O/pytest.txt
rm -rf /tmp/pytest-of-root
MK_PREFILTER_DEBUG=1 timeout 600 python tools/config5_search_profile.py 11800000 20000 100000 > $O/config5_search.json 2> $O/config5_search.err; echo "c5 profile rc $?"
grep "wide piece" $O/config5_search.err | tail -1
python - <<'P'
import json
d=json.load(open("gpurun_out/r04o/config5_search.json"))
for r in d["runs"]:
    print(r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:14]})
P
