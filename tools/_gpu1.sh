#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_profile.py tests/test_gpu_scale.py::test_profile_path_scale_parity tests/test_gpu_k7.py -x -q -m gpu 2>&1 | tail -5
