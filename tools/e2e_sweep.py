"""`metaeuk-amd predictexons` over the headline workload's DBs on disk under different contig-batch sizes (MK_CLI_BATCH_NT) and host-thread
counts: the command's own stage account per run.  gpurun -- 'python tools/e2e_sweep.py > gpurun_out/e2e_sweep.txt'"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MK_DEBUG"] = "1"
import bench  # noqa: E402
from metaeuk_amd import api, build, synth  # noqa: E402


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    budgets = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1 << 22, 1 << 23, 3 << 22, 1 << 24, 1 << 25, 1 << 26]
    api.init(0)
    targets, queries, founders = bench.make_inputs(n_contigs, 100000, 11, 0)
    t_res, t_off = bench.pack(targets)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        contigs = synth.make_contigs(n_contigs, founders, 11)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        blobs = [lut[c].tobytes() + b"\n\0" for c in contigs]
        offs = np.zeros(len(blobs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
        api.write_seq_db(os.path.join(tmp, "contigs"), (b"".join(blobs), np.arange(len(blobs), dtype=np.uint32), offs[:-1], np.array([len(b) for b in blobs], dtype=np.uint32)), dbtype=1)
        api.write_seq_db(os.path.join(tmp, "targets"), api.synth_seqdb(t_res, t_off))
        cmd = [build.BIN, "predictexons", os.path.join(tmp, "contigs"), os.path.join(tmp, "targets"), os.path.join(tmp, "calls"), os.path.join(tmp, "tmp"),
               "-s", "5.7", "--threads", str(int(api.lib().mk_host_threads()))]
        ref = None
        for extra in ([{}] + [{"MK_CLI_BATCH_NT": str(b)} for b in budgets if b] + [{"MK_CLI_WARM": "0"}]):
            for rep in range(2):
                for suffix in ("", ".index", ".dbtype"):
                    if os.path.exists(os.path.join(tmp, "calls" + suffix)):
                        os.remove(os.path.join(tmp, "calls" + suffix))
                t0 = time.time()
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **extra))
                dt = time.time() - t0
                if r.returncode != 0:
                    print("FAILED", extra, r.stderr.decode()[-500:])
                    continue
                data = open(os.path.join(tmp, "calls"), "rb").read()
                if ref is None:
                    ref = data
                lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("predictexons") or ln.startswith("[predictexons]")]
                print("%-32s rep %d wall %.3f s same-bytes %s\n    %s" % (extra, rep, dt, data == ref, "\n    ".join(lines)), flush=True)


if __name__ == "__main__":
    main()
