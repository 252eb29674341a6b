cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
MK_PREFILTER_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 1 --two-calls --cpu-sample 0 --config4-profiles 0 > gpurun_out/bench_two_calls_debug.log 2>&1
echo "rc=$?" >> gpurun_out/bench_two_calls_debug.log
grep -c prefilter gpurun_out/bench_two_calls_debug.log
