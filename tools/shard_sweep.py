"""Predicts the strong-scaling curve of BASELINE config 3 on ONE GPU: the 10 000-contig workload query-sharded over N ranks gives every rank
1 / N of the fragments (the reference's residue-balanced rule, DBReader::decomposeDomainByAminoAcid), so a rank's step time at N = 1, 2, 4, 8 is the
step time of the first shard of that split against the same index replica (no collective on the data path).  Implied efficiency at
N = t(1) / (N * t(N)).  Round 6: ONE process, the workload generated once, the four shard sizes timed in turn `--reps` times (interleaved, so that a
drift of the box hits every size alike), mean and spread per point; queued batches (mk_search_begin / mk_search_wait two deep, what bench.py and
the commands do) and the blocking call.
   python tools/shard_sweep.py [--steps 8] [--reps 5] [--host-threads N] [--dummy-load M]  > gpurun_out/shard_sweep.txt
--host-threads: OpenMP threads of this rank (a rank of an 8-rank launch on a 16-core box has 2); --dummy-load M: M busy host processes beside it
(the other ranks' host sides)."""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--contigs", type=int, default=10000)
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--dummy-load", type=int, default=0)
    a = ap.parse_args()
    if a.host_threads:
        os.environ["OMP_NUM_THREADS"] = str(a.host_threads)
    import bench
    from metaeuk_amd import api, shard
    api.init(0)
    params = api.default_params()
    targets, queries, founders = bench.make_inputs(a.contigs, 100000, 11, 0)
    t_res, t_off = bench.pack(targets)
    db = api.TargetDB.from_codes(t_res, t_off, params)
    shards = {}
    for n in (1, 2, 4, 8):
        first, count = shard.decompose_by_residues([len(x) + 2 for x in queries], 0, n)
        shards[n] = bench.pack(queries[first:first + count]) + (count,)
    loads = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(a.dummy_load)]

    def queued(q_res, q_off, steps):
        pending = []
        for k in range(steps):
            q = api.Queries.from_codes(q_res, q_off, params)
            api.search_begin(db, q)
            pending.append(q)
            if len(pending) >= 2:
                p = pending.pop(0)
                api.search_wait(p)
                p.close()
        for p in pending:
            api.search_wait(p)
            p.close()

    def blocking(q_res, q_off, steps):
        for k in range(steps):
            q = api.Queries.from_codes(q_res, q_off, params)
            api.search(db, q)
            q.close()

    try:
        res = {(n, m): [] for n in shards for m in ("queued", "blocking")}
        for n in shards:                                   # warm-up of every size
            queued(shards[n][0], shards[n][1], 3)
        for rep in range(a.reps):
            for n in (1, 2, 4, 8):
                for mode, fn in (("queued", queued), ("blocking", blocking)):
                    steps = a.steps if mode == "queued" else max(3, a.steps // 2)
                    t0 = time.time()
                    fn(shards[n][0], shards[n][1], steps)
                    res[(n, mode)].append((time.time() - t0) / steps * 1e3)
    finally:
        for p in loads:
            p.kill()
    print("# tools/shard_sweep.py --steps %d --reps %d --host-threads %s --dummy-load %d; %d contigs, host threads of the library: %d" % (
        a.steps, a.reps, a.host_threads or "default", a.dummy_load, a.contigs, int(api.lib().mk_host_threads()) if not a.host_threads else a.host_threads))
    print("# ranks  fragments/rank   queued ms/step: mean  min  max (reps)        implied efficiency mean [min of t1/max of tN .. max/min]     blocking ms/step mean   implied efficiency")
    base = {m: res[(1, m)] for m in ("queued", "blocking")}
    for n in (1, 2, 4, 8):
        q, b = np.array(res[(n, "queued")]), np.array(res[(n, "blocking")])
        eff = np.mean(base["queued"]) / (n * q.mean())
        lo, hi = np.min(base["queued"]) / (n * q.max()), np.max(base["queued"]) / (n * q.min())
        effb = np.mean(base["blocking"]) / (n * b.mean())
        print("  %d      %8d       %8.1f %8.1f %8.1f   %s      %.3f [%.3f .. %.3f]        %8.1f      %.3f" % (
            n, shards[n][2], q.mean(), q.min(), q.max(), " ".join("%.1f" % x for x in q), eff, lo, hi, b.mean(), effb), flush=True)


if __name__ == "__main__":
    main()
