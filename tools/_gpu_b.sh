#!/bin/bash
# round 4, GPU call B: the wide per-query kernel through the parity tests, config-5 scale search profile, bench + shard sweep + RCCL path
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04b; mkdir -p $O
export MK_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_k7.py tests/test_gpu_profile.py -m gpu -q --maxfail=12 -k "wide or k7 or profile" -p no:cacheprovider > $O/pytest_wide.txt 2>&1; echo "pytest wide rc $?" >> $O/pytest_wide.txt
tail -25 $O/pytest_wide.txt
rm -rf /tmp/pytest-of-root
timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "config5" -p no:cacheprovider > $O/pytest_c5.txt 2>&1; echo "pytest c5 rc $?" >> $O/pytest_c5.txt
tail -8 $O/pytest_c5.txt
rm -rf /tmp/pytest-of-root
timeout 600 python tools/config5_search_profile.py 11800000 20000 100000 > $O/config5_search.json 2> $O/config5_search.err; echo "c5 profile rc $?"
python - <<'P'
import json
try:
    d=json.load(open("gpurun_out/r04b/config5_search.json"))
    for r in d["runs"]:
        print(r["fragments"], "fragments", r["t_search_s"], "s", r["fragments_per_s"], "frag/s", {k: v for k, v in list(r["kernels_ms"].items())[:12]})
except Exception as e:
    print("no config5 profile:", e)
P
unset MK_DEBUG
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], d["metric"])
    print("digest", {k: v for k, v in d.get("result_digest", {}).items() if k not in ("note",)})
    c4 = d.get("config4_profile_targets", {})
    print("config4", c4.get("s_per_pass"), c4.get("result_digest", {}).get("match"), c4.get("result_digest", {}).get("profiles"), {k: v for k, v in c4.get("kernels_ms", {}).items() if not k.startswith("sw_")})
    print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    print({k: v for k, v in d["kernels_ms"].items() if k.startswith("align_") or k.startswith("host_")})
except Exception as e:
    print("no bench line:", e)
P
timeout 900 python tools/shard_sweep.py --steps 4 --chunks 0,65536 > $O/shard_sweep.txt 2>$O/shard_sweep.err; echo "sweep rc $?"; cat $O/shard_sweep.txt
MK_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --config4-profiles 0 > $O/bench_dist.json 2> $O/bench_dist.err; echo "dist rc $?"; tail -c 400 $O/bench_dist.json
