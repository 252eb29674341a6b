cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/t_all.log 2>&1
echo "all rc=$?" >> gpurun_out/t_all.log
tail -n 40 gpurun_out/t_all.log
