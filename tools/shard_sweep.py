"""Predicts the strong-scaling curve of BASELINE config 3 on ONE GPU: the 10 000-contig workload query-sharded over N ranks gives every rank
10 000 / N contigs, so a rank's step time at N = 1, 2, 4, 8 is bench.py's step time at 10 000, 5 000, 2 500, 1 250 contigs (same target DB,
same index replica, no collective on the data path).  Implied efficiency at N = t(10 000) / (N * t(10 000 / N)).
   python tools/shard_sweep.py [--steps 4] [--chunks 0,32768,65536]  > gpurun_out/shard_sweep.txt
--chunks: values of MK_SEARCH_CHUNK_QUERIES to compare (0 = the library's own schedule)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(contigs, steps, chunk):
    env = dict(os.environ)
    if chunk:
        env.update(MK_DEBUG="1", MK_SEARCH_CHUNK_QUERIES=str(chunk))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--contigs", str(contigs), "--steps", str(steps), "--warmup", "2",
                                   "--cpu-sample", "0", "--config4-profiles", "0", "--e2e-sample", "-1"], env=env, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
    d = json.loads(out)
    host = {k: round(v / d["steps"], 1) for k, v in d["kernels_ms"].items() if k.startswith("host_") or k.startswith("wait_")}
    blocking = (d.get("blocking") or {}).get("ms_per_step")
    return d["ms_per_step"], d["value"], int(d["config"]["workload"].split("(")[1].split(" ")[0]), host, blocking


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--chunks", default="0")
    a = ap.parse_args()
    for chunk in [int(x) for x in a.chunks.split(",")]:
        print("# chunk schedule: %s" % ("library default (a function of the batch size, mk_abi.cpp: mk_search)" if not chunk else "MK_SEARCH_CHUNK_QUERIES=%d" % chunk))
        print("# ranks  contigs/rank  fragments/rank  ms_per_step  fragments/s(rank)  implied node fragments/s  implied efficiency  | blocking mk_search: ms_per_step  implied efficiency |  host phases (ms/step)")
        base = base_b = None
        for n in (1, 2, 4, 8):
            ms, fps, nq, host, blk = run(10000 // n, a.steps, chunk)
            if base is None:
                base, base_b = ms, blk
            print("  %d      %6d        %8d       %8.1f      %10.0f          %10.0f            %.3f          |  %8s  %6s  |  %s" % (
                n, 10000 // n, nq, ms, fps, fps * n, base / (n * ms), "%.1f" % blk if blk else "-", "%.3f" % (base_b / (n * blk)) if blk and base_b else "-", host), flush=True)


if __name__ == "__main__":
    main()
