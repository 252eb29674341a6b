cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_index.py -q -m gpu -s > gpurun_out/t_index.log 2>&1
echo "index rc=$?" >> gpurun_out/t_index.log
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -s -k "config5_scale_4e9" > gpurun_out/t_c5_4e9.log 2>&1
echo "c5 4e9 rc=$?" >> gpurun_out/t_c5_4e9.log
