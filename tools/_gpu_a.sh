#!/bin/bash
# round 4, GPU call A: box facts, the GPU suite, the bench line with whole-result digests, the shard sweep, the RCCL path on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04a; mkdir -p $O
{ nproc; free -g; df -h /tmp . ; cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/memory.max 2>/dev/null; rocm-smi --showmeminfo vram 2>/dev/null | tail -4; } > $O/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -x -p no:cacheprovider > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04a/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], d["metric"])
print("digest", {k: v for k, v in d.get("result_digest", {}).items() if k not in ("note",)})
print("config4", d.get("config4_profile_targets", {}).get("s_per_pass"), d.get("config4_profile_targets", {}).get("result_digest", {}).get("match"), d.get("config4_profile_targets", {}).get("result_digest", {}).get("profiles"))
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
P
timeout 900 python tools/shard_sweep.py --steps 4 --chunks 0,65536 > $O/shard_sweep.txt 2>$O/shard_sweep.err; echo "sweep rc $?"; cat $O/shard_sweep.txt
MK_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --config4-profiles 0 > $O/bench_dist.json 2> $O/bench_dist.err; echo "dist rc $?"; tail -c 600 $O/bench_dist.json
