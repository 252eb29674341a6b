cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_profile.py -q -m gpu -x > gpurun_out/t_prof.log 2>&1
echo "rc=$?" >> gpurun_out/t_prof.log
tail -n 30 gpurun_out/t_prof.log
