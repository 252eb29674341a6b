cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/t_all.log 2>&1
echo "all rc=$?" >> gpurun_out/t_all.log
tail -n 60 gpurun_out/t_all.log
