#!/usr/bin/env python3
"""bench.py -- prefilter+align throughput of the MI355X hot path on the BASELINE workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one full pass of the hot path (query-side derivation, k-mer prefilter, ungapped diagonal
scoring, hit selection, gapped SW forward/reverse, e-values) over one batch of synthetic ORF
fragments against the HBM-resident target DB + index.  The workload is BASELINE.json configs[1]
(10k synthetic 5-kb contigs x 100k proteins, -s 5.7) unless --contigs/--targets shrink it.
Multi-GPU = query sharding, no collective on the data path (only the timing barrier/max):
  --scaling strong (default for --gpus > 1, BASELINE config 3): the SAME 10k-contig workload is split over the
                   ranks with the reference's residue-balanced rule (DBReader::decomposeDomainByAminoAcid);
  --scaling weak   every rank searches its own 10k contigs.
Every rank holds a full replica of the target index.

The JSON line carries `result_digest`: SHA-256 digests over the formatted prefilter hits and alignments of the last
timed step -- ALL of its queries by default, in pieces of --cpu-sample queries -- next to the same digests of what the
reference's own code (oracle/_ref/ref_harness) wrote for the same queries on this box's host cores, outside the timed
region.  The CPU baseline is TIMED on the first piece only (a bounded sample).  "bit-exact" is checked, not asserted: the
metric string says "(bit-exact hits)" only when that comparison ran and matched, "(hits UNVERIFIED ...)" when it did not run.
The config-4 leg (profile targets) carries its own digest against the reference harness's `profilesearch` mode.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Integer / packed-int16 / DPP vector instructions issue once per 4 cycles per SIMD for a 64-lane wave (measured: 0.60 T wave-instructions/s
# on the whole chip, profiles/r02_valu_issue_rates.txt) -- half the fp32 FMA rate the microarchitecture guide quotes for the SIMDs:
VALU_PEAK_TOPS = 256 * 4 * 64 * 2.4e9 / 4 / 1e12   # = 39.3 T lane-ops/s
# lane-ops per DP cell: the int32 kernels (position / reverse pass, large tiles) spend 10 (add, min, max3, lshl_or, max, sub, sub, max3, sub,
# max3), the packed int16 score pass 10 per PAIR of cells (perm, add, max, max, max, sub, sub, max, sub, max) = 5 per cell
SW_OPS_PER_CELL_INT32, SW_OPS_PER_CELL_PACKED = 10, 5
PACKED_ROWS_MAX = 768     # tiles of at most 768 rows run the packed score pass (mk_kernels.hpp: sw_cfg_packed)


PMC_LAST = {}             # the artefact's record of the kernel pmc_traffic() was last asked about (per-step figures of the roofline object)
RANDOM_REQUEST_PEAK = 55.0e9   # independent random 128-byte fabric requests per second, measured on this chip (profiles/r02_random_probe_rates.txt: 56.8 G/s from
                               # 128 MB, 54.4 G/s from 2 GB)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of a kernel from the newest PMC artefact under profiles/ (written by tools/pmc_aggregate.py
    from separate rocprofv3 --pmc passes): {"build": <sha of the kernel sources>, "kernels": {name: {"fetch_bytes":..,
    "write_bytes":.., "launches":..}}}.  None when there is no artefact or it was taken on other kernel sources."""
    import glob
    import hashlib
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not cands:
        return None, None, None
    art = json.load(open(cands[-1]))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "metaeuk_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "metaeuk_amd", "csrc", "mk_enum.hpp"))):
        h.update(open(f, "rb").read())
    k = art.get("kernels", {}).get(kernel_name)
    if not k:
        return None, os.path.basename(cands[-1]), None
    PMC_LAST.clear()
    PMC_LAST.update(k, passes=art.get("passes", 2))              # (the artefact's command runs warm-up + 1 step: two passes of the workload)
    note = os.path.basename(cands[-1]) + ("" if art.get("build") == h.hexdigest()[:16] else " (taken on an older build of the kernels)")
    total = (k["fetch_bytes"] + k["write_bytes"]) / max(k["launches"], 1)
    dram = None
    if k.get("read_requests_dram") is not None and k.get("read_requests"):
        # the share of the requests that is bound for the memory controllers, applied to the bytes (Infinity Cache hits are inside it)
        rd_share = k["read_requests_dram"] / k["read_requests"]
        wr_share = (k["write_requests_dram"] / k["write_requests"]) if k.get("write_requests") and k.get("write_requests_dram") is not None else 1.0
        dram = (k["fetch_bytes"] * rd_share + k["write_bytes"] * wr_share) / max(k["launches"], 1)
    return total, note, dram


def make_inputs(n_contigs, n_targets, seed, rank):
    from metaeuk_amd import synth
    targets, founders = synth.make_targets(n_targets, seed)
    queries = synth.make_queries(n_contigs, founders, seed + 7919 * rank)
    return targets, queries, founders


def pack(codes_list):
    off = np.zeros(len(codes_list) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(c) for c in codes_list], dtype=np.uint64)
    res = np.concatenate(codes_list).astype(np.uint8) if codes_list else np.zeros(1, np.uint8)
    return np.ascontiguousarray(res), off


def gpu_digest(api, hits, hoff, alns, aoff, n, first=0):
    """SHA-256 over the formatted hits / alignments of the queries [first, first + n) (tests/oracle.py: digest_arrays)"""
    import oracle
    hb = api.format_hits_bulk(hits, int(hoff[first]), int(hoff[first + n]))
    ab = api.format_alignments_bulk(alns, int(aoff[first]), int(aoff[first + n]))
    return {"prefilter": oracle.digest_arrays(hoff[first:first + n + 1], hb, n), "alignments": oracle.digest_arrays(aoff[first:first + n + 1], ab, n)}


def cpu_baseline(targets, queries, budget_queries, threads, first=0):
    """Reference AVX2 code (oracle/_ref/ref_harness, built from the reference's own sources) timed on the
    host cores of this box on a bounded sample of the same workload (the queries [first, first + budget_queries));
    falls back to the C oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from metaeuk_amd import synth
    sample = queries[first:first + budget_queries]
    with tempfile.TemporaryDirectory() as tmp:
        tf, qf = os.path.join(tmp, "t.txt"), os.path.join(tmp, "q.txt")
        with open(tf, "w") as f:
            f.write("\n".join(synth.codes_to_str(t) for t in targets) + "\n")
        with open(qf, "w") as f:
            f.write("\n".join(synth.codes_to_str(q) for q in sample) + "\n")
        if os.path.exists(oracle.REF):
            matdir = os.path.join(tmp, "mat")
            oracle.write_matrix_files(matdir)
            out = subprocess.check_output([oracle.REF, "pipeline", matdir, tf, qf, os.path.join(tmp, "o"), "--threads", str(threads)],
                                          stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1]
            st = json.loads(out)
            t = st["t_prefilter"] + st["t_align"]
            kind = "reference"
            odir = os.path.join(tmp, "o")
        else:
            oracle.build()
            env = dict(os.environ, OMP_NUM_THREADS=str(threads))
            out = subprocess.check_output([oracle.CLI, "pipeline", tf, qf, os.path.join(tmp, "o")], env=env).decode().strip().splitlines()[-1]
            st = json.loads(out)
            t = st["t_prefilter_align"]
            kind = "port"
            odir = os.path.join(tmp, "o")
        digest = {"prefilter": oracle.digest_blocks_file(os.path.join(odir, "pref.txt"))[0],
                  "alignments": oracle.digest_blocks_file(os.path.join(odir, "aln.txt"))[0]}
    return {"value": len(sample) / t, "unit": "fragments/s", "cores": threads, "kind": kind,
            "sample": "%s %d ORF fragments of the same workload vs the full target DB; prefilter %.2fs + align %.2fs (index build excluded)" % (
                "first" if first == 0 else "next", len(sample), st.get("t_prefilter", t), st.get("t_align", 0.0)),
            "gcups_align": st["cells_fwd"] / max(st.get("t_align", t), 1e-9) / 1e9, "digest": digest}


def config4_leg(api, args, params, q_res, q_off, nq):
    """BASELINE config 4 on the fragments of the headline workload: synthetic profiles as queries (prefilter + align), swapresults; the hits
    and alignments of a SAMPLE of the profiles are compared with the reference harness's `profilesearch` mode (the reference's own
    Sequence::mapProfile / QueryMatcher / Matcher, oracle/Makefile.ref) run on the host cores over the same fragments."""
    from metaeuk_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    t1 = time.time()
    proteins, _ = synth.make_targets(args.config4_profiles, args.seed)
    entries = synth.make_profiles(proteins, args.seed)
    pp = api.default_params()
    pp.sensitivity = 4.0                                   # predictexons' own default
    pp.profile_search = 1
    pp.max_seqs = max(300, nq)
    pp.evalue_thr = float("%g" % (100.0 * (np.float32(nq) / np.float32(len(entries)))))
    pp.host_l2_bytes = params.host_l2_bytes
    t_gen4 = time.time() - t1
    t1 = time.time()
    fdb = api.TargetDB.from_codes(q_res, q_off, pp)         # the fragments of the headline workload as the indexed side
    t_idx4 = time.time() - t1
    residues = sum(len(e) for e in entries) // 25 - len(entries)
    p_cols, p_off = api.Profiles.pack(entries)               # host buffers as the boundary takes them (like q_res / q_off above)
    times = []
    for it in range(2):
        api.kernel_stats(reset=True)
        t1 = time.time()
        pq = api.Profiles.from_columns(p_cols, p_off, pp)
        (ph, pho), (pa, pao) = api.search(fdb, pq, pp)
        sp = api.default_params()
        sp.evalue_thr = 1.7976931348623157e308
        sw, soff = api.swap_alignments(pa, pao, nq, residues, params=sp)
        times.append(time.time() - t1)
        st4 = api.kernel_stats()
        counts = (int(pho[-1]), int(pao[-1]), int(soff[-1]))
        if it == 0:
            del pq, sw
    cells4 = sum(v["cells"] for k, v in st4.items() if k.startswith("sw_fwd"))
    out = {
        "workload": "%d synthetic profiles (%d columns) as queries x the %d fragments above as the indexed side, -s 4; prefilter + align + swapresults" % (
            len(entries), sum(len(e) // 25 for e in entries), nq),
        "s_per_pass": round(times[-1], 3), "profiles_per_s": round(len(entries) / times[-1], 1), "first_pass_s": round(times[0], 3),
        "prefilter_hits": counts[0], "alignments": counts[1], "swapped_records": counts[2], "sw_fwd_cells": cells4,
        "setup_s": {"generate_profiles": round(t_gen4, 2), "fragment_index_build_upload": round(t_idx4, 2)},
        "kernels_ms": {k: round(v["ms"], 2) for k, v in sorted(st4.items()) if v["ms"] >= 1.0}}
    # digest of the last pass against the reference on a sample of the profiles
    n_s = min(args.config4_sample, len(entries))
    if n_s > 0 and os.path.exists(oracle.REF):
        g = gpu_digest(api, ph, pho, pa, pao, n_s)
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "p.bin"), "wb") as f:
                f.write(b"".join(entries[:n_s]))
            off = 0
            with open(os.path.join(tmp, "p.index"), "w") as f:
                for k, e in enumerate(entries[:n_s]):
                    f.write("%d\t%d\t%d\n" % (k, off, len(e)))
                    off += len(e)
            letters = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYX", dtype=np.uint8)
            with open(os.path.join(tmp, "f.txt"), "wb") as f:                  # one fragment per line
                lens = np.diff(q_off.astype(np.int64))
                txt = np.full(int(q_off[-1]) + len(lens), 10, dtype=np.uint8)
                pos = np.arange(int(q_off[-1]), dtype=np.int64) + np.repeat(np.arange(len(lens), dtype=np.int64), lens)
                txt[pos] = letters[q_res[:int(q_off[-1])]]
                f.write(txt.tobytes())
            mat = oracle.write_matrix_files(os.path.join(tmp, "mat"))
            t1 = time.time()
            subprocess.check_call([oracle.REF, "profilesearch", mat, os.path.join(tmp, "p.bin"), os.path.join(tmp, "p.index"), os.path.join(tmp, "f.txt"),
                                   os.path.join(tmp, "o"), "-s", "4", "--eval-abs", repr(float(pp.evalue_thr)), "--threads", str(int(api.lib().mk_host_threads()))],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t_ref = time.time() - t1
            c = {"prefilter": oracle.digest_blocks_file(os.path.join(tmp, "o", "pref.txt"))[0],
                 "alignments": oracle.digest_blocks_file(os.path.join(tmp, "o", "aln.txt"))[0]}
        out["cpu_baseline"] = {"value": round(n_s / max(t_ref, 1e-9), 2), "unit": "profiles/s", "cores": int(api.lib().mk_host_threads()), "kind": "reference",
                               "sample": "%d of the profiles against all %d fragments (ref_harness profilesearch, fragment index build included)" % (n_s, nq)}
        out["result_digest"] = {"profiles": n_s, "gpu": g, "cpu": c, "match": c == g, "reference_s": round(t_ref, 1),
                                "reference": "oracle/_ref/ref_harness profilesearch (index build of the fragments included in reference_s)"}
    else:
        out["result_digest"] = {"profiles": 0, "match": None}
    fdb.close()
    return out


def e2e_leg(api, args, params, targets, founders, t_res, t_off):
    """The drop-in as a user meets it: `metaeuk-amd predictexons <contigsDB> <targetsDB> <out> <tmp>` (data/predictexons.sh:42-87: extractorfs ->
    search = prefilter + align -> resultspercontig + collectoptimalset) as ONE process over MMseqs2 DBs on disk, on the headline workload -- the
    contigs as nucleotides this time.  Wall clock of the command (process start, DB reads, target masking + index, every stage, DB writes), contigs/s,
    and the exon sets of a SAMPLE of the contigs against the reference's own chain (oracle/_ref/ref_harness orfs | pipeline | exons: Orf.cpp,
    the prefilter / align code, collectoptimalset.cpp compiled in place) run on the host cores.  Outside the timed region of the headline figure."""
    from metaeuk_amd import build, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    out = {"command": "metaeuk-amd predictexons contigsDB targetsDB outDB tmp -s 5.7 (predictexons defaults)", "contigs": args.contigs}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        contigs = synth.make_contigs(args.contigs, founders, args.seed)
        cstr = ["".join("ACGT"[x] for x in c) for c in contigs] if args.contigs <= 2000 else None
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        # contigs DB (nucleotides, dbtype 1) and targets DB, data order = key order
        blobs = [lut[c].tobytes() + b"\n\0" for c in contigs]
        offs = np.zeros(len(blobs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
        api.write_seq_db(os.path.join(tmp, "contigs"), (b"".join(blobs), np.arange(len(blobs), dtype=np.uint32), offs[:-1], np.array([len(b) for b in blobs], dtype=np.uint32)), dbtype=1)
        api.write_seq_db(os.path.join(tmp, "targets"), api.synth_seqdb(t_res, t_off))
        out["nucleotides"] = int(sum(len(c) for c in contigs))
        cmd = [build.BIN, "predictexons", os.path.join(tmp, "contigs"), os.path.join(tmp, "targets"), os.path.join(tmp, "calls"), os.path.join(tmp, "tmp"),
               "-s", "5.7", "--ref-l2-bytes", str(params.host_l2_bytes), "--threads", str(int(api.lib().mk_host_threads())), "--remove-tmp-files", "1"]
        best = None
        for rep in range(2):                                     # the second run finds the DB files in the page cache
            for suffix in ("", ".index", ".dbtype"):
                if os.path.exists(os.path.join(tmp, "calls" + suffix)):
                    os.remove(os.path.join(tmp, "calls" + suffix))
            t0 = time.time()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.time() - t0
            if r.returncode != 0:
                raise RuntimeError("metaeuk-amd predictexons failed: " + r.stderr.decode()[-800:])
            if best is None or dt < best:
                best = dt
                rep_lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("predictexons:")]
                out["command_report"] = rep_lines[-1] if rep_lines else None      # the command's own account of its stages
                st_lines = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("predictexons stages:")]
                if st_lines:                                                       # round 6: per-stage seconds of the command's main thread
                    stages = {k.strip().replace(" ", "_").replace("(", "").replace(")", "").replace("+_", ""): float(v)
                              for k, v in re.findall(r"([a-z+ ()]+?) (\d+\.\d+) s", st_lines[-1].split(";", 1)[1])}
                    m = re.search(r"(\d+) batches of <= (\d+) nt", st_lines[-1])
                    out["stages_s"] = stages
                    if m:
                        out["batches"] = int(m.group(1)); out["batch_nt"] = int(m.group(2))
                m = re.search(r"-> (\d+) fragments", out["command_report"] or "") if rep_lines else None
                if m:
                    out["fragments"] = int(m.group(1))
                m = re.search(r"; (\d+\.\d+) s \(", out["command_report"] or "")
                if m:
                    out["command_s"] = float(m.group(1))
        out.update({"wall_s": round(best, 3), "contigs_per_s": round(args.contigs / best, 1), "runs": 2})
        if out.get("fragments"):
            out["fragments_per_s"] = round(out["fragments"] / best, 1)
        data = open(os.path.join(tmp, "calls"), "rb").read()
        got = {}
        for line in open(os.path.join(tmp, "calls.index")):
            k, o, l = line.split("\t")
            got[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
        out["contigs_with_predictions"] = sum(1 for v in got.values() if v)
        out["prediction_lines"] = sum(v.count("\n") for v in got.values())
        n_s = min(args.e2e_sample, args.contigs)
        if n_s > 0 and os.path.exists(oracle.REF):
            t1 = time.time()
            d = os.path.join(tmp, "ref")
            os.makedirs(os.path.join(d, "out"))
            matdir = oracle.write_matrix_files(os.path.join(d, "mat"))
            with open(os.path.join(d, "t.txt"), "w") as f:
                f.write("\n".join(synth.codes_to_str(t) for t in targets) + "\n")
            with open(os.path.join(d, "c.txt"), "w") as f:
                f.write("\n".join(lut[c].tobytes().decode() for c in contigs[:n_s]) + "\n")
            subprocess.check_call([oracle.REF, "orfs", os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt")], stdout=subprocess.DEVNULL)
            with open(os.path.join(d, "q.txt"), "w") as f:
                for line in open(os.path.join(d, "orfs.txt")):
                    if not line.startswith(">"):
                        f.write(line.rstrip("\n").rsplit("\t", 1)[1] + "\n")
            subprocess.check_call([oracle.REF, "pipeline", matdir, os.path.join(d, "t.txt"), os.path.join(d, "q.txt"), os.path.join(d, "out"),
                                   "--threads", str(int(api.lib().mk_host_threads()))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            subprocess.check_call([oracle.REF, "exons", os.path.join(d, "t.txt"), os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt"),
                                   os.path.join(d, "out", "aln.txt"), os.path.join(d, "exons.txt")], stdout=subprocess.DEVNULL)
            exp, cur = {}, None
            for line in open(os.path.join(d, "exons.txt")):
                if line.startswith(">"):
                    cur = int(line[1:])
                    exp[cur] = ""
                else:
                    exp[cur] += line
            bad = [c for c in range(n_s) if got.get(c, "") != exp.get(c, "")]
            import hashlib
            dig = lambda dd: hashlib.sha256("".join(">%d\n%s" % (c, dd.get(c, "")) for c in range(n_s)).encode()).hexdigest()
            n_frag_ref = sum(1 for line in open(os.path.join(d, "q.txt")))
            # the reference's chain on the host cores over the sample, as a rate (its target index build is inside: it is inside wall_s too)
            out["cpu_baseline"] = {"value": round(n_frag_ref / max(time.time() - t1, 1e-9), 1), "unit": "fragments/s", "cores": int(api.lib().mk_host_threads()), "kind": "reference",
                                   "sample": "the first %d contigs (%d fragments) through ref_harness orfs | pipeline | exons, target index build included" % (n_s, n_frag_ref)}
            out["result_digest"] = {"contigs": n_s, "gpu": dig(got), "cpu": dig(exp), "match": not bad, "differing_contigs": bad[:10],
                                    "contigs_with_predictions_in_the_sample": sum(1 for c in range(n_s) if exp.get(c)),
                                    "reference_s": round(time.time() - t1, 1), "reference": "oracle/_ref/ref_harness orfs | pipeline | exons on the first contigs"}
        else:
            out["result_digest"] = {"contigs": 0, "match": None}
    return out



def config5_leg_child(args):
    """Runs in a process of its own (`bench.py --config5-only`; the 60 M-protein index takes 230 of the 288 GB): BASELINE config 5's target side at its
    stated size -- 60 000 000 proteins, 2.25e10 residues, k = 7 -- masked and indexed in HBM once, then (a) `--config5-fragments` planted fragments searched
    UNSPLIT with mk_search, twice: fragments/s of the warm pass and its kernel table; (b) the committed golden's fragments through `metaeuk-amd prefilter
    --split N --split-mode 0` + `align`, digests against tests/golden/config5_digest_60000000_split<N>.json -- the reference's own run in TARGET_DB_SPLIT
    mode (Prefiltering.cpp:273-377,352-362) on a GPU box's host cores, recorded there with its times."""
    import glob
    import hashlib
    import shutil
    from metaeuk_amd import api, build
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config5_digest as c5
    api.init(0)
    n_targets = args.config5_targets
    out = {"workload": "%d planted fragments (30-120 aa, 10 %% redrawn, every tenth background) x %d proteins of the native generator, -s 5.7, k = 7; mk_search unsplit" % (
        args.config5_fragments, n_targets)}
    free, total = api.device_memory()
    if n_targets >= 60000000 and total < 280e9:
        return {"skipped": "needs a 288 GB device (this one: %.0f GB)" % (total / 1e9)}
    t0 = time.time()
    res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
    out["target_residues"] = int(off[-1])
    fr, foff, src = api.synth_fragments(args.config5_fragments, res, off, **c5.FRAGMENTS)
    golds = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "config5_digest_%d_split*.json" % n_targets)))
    gold = json.load(open(golds[-1])) if golds else None
    tmp = None
    if gold and args.config5_digest:
        need = 1.3 * gold["target_residues"] + 4e9
        for cand in (os.environ.get("MK_TEST_SCRATCH"), os.environ.get("TMPDIR"), "/tmp", "/var/tmp", os.path.join(ROOT, "gpurun_out")):
            if cand and os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free >= need:
                tmp = tempfile.mkdtemp(prefix="mk_bench_config5_", dir=cand)
                break
        if tmp:
            gfr, gfoff, _ = c5.make_fragments(api, res, off, gold["n_queries"], gold.get("n_long_queries", 0))
            api.synth_write_seqdb(os.path.join(tmp, "T"), res, off)
            api.synth_write_seqdb(os.path.join(tmp, "Q"), gfr, gfoff)
    out["setup_s"] = {"generate_and_write_dbs": round(time.time() - t0, 1)}
    p = api.default_params()
    t0 = time.time()
    db = api.TargetDB.from_codes(res, off, p)
    out["setup_s"]["mask_and_index_in_hbm"] = round(time.time() - t0, 1)
    out["kmer_size"], out["index_entries"] = db.kmer_size(), db.index_entries()
    del res
    times = []
    for it in range(3):
        api.kernel_stats(reset=True)
        q = api.Queries.from_codes(fr, foff, p)
        t0 = time.time()
        (hits, hoff), (alns, aoff) = api.search(db, q, p)
        times.append(time.time() - t0)
        st = api.kernel_stats()
        counts = (int(hoff[-1]), int(aoff[-1]))
        q.close()
    warm = min(times[1:])
    out.update({"fragments": args.config5_fragments, "fragment_residues": int(foff[-1]), "s_per_pass_warm": round(warm, 3), "first_pass_s": round(times[0], 3),
                "fragments_per_s": round(args.config5_fragments / warm, 1), "prefilter_hits": counts[0], "alignments": counts[1],
                "device_memory_free_gb": round(api.device_memory()[0] / 1e9, 1),
                "kernels_ms": {k: round(v["ms"], 1) for k, v in sorted(st.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 2.0}})
    wide = {k: v for k, v in st.items() if k.startswith("prefilter_query_wide")}
    if wide:
        ms = sum(v["ms"] for v in wide.values())
        kmers = sum(v["cells"] for v in wide.values())
        ab = sum(v["alg_bytes"] for v in wide.values())
        out["roofline"] = {"bound": "hbm", "kernel": "prefilter_query_wide", "achieved": ab / max(ms * 1e-3, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ab / max(ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms, "similar_kmers": kmers, "traffic": None,
                           "note": "algorithmic bytes = 16 B per probed k-mer + 6 B per index entry (SURVEY 8(d)) of the wide per-query kernel over the warm pass / its event time"}
    db.close()
    if gold:
        ref = gold["reference"]
        out["cpu_baseline"] = {"value": round((gold["n_queries"] + gold.get("n_long_queries", 0)) / max(ref["t_prefilter_s"] + ref["t_align_s"], 1e-9), 1), "unit": "fragments/s",
                               "cores": ref["threads"], "kind": "reference", "recorded": True,
                               "sample": "RECORDED in %s, not re-run here (%.0f s): ref_harness pipeline --split %d over %d fragments on a GPU box's host cores, prefilter %.1f s + align %.1f s, "
                                         "the four index builds (%.0f s) excluded" % (os.path.basename(golds[-1]), ref["wall_s"], gold["target_splits"],
                                                                                    gold["n_queries"] + gold.get("n_long_queries", 0), ref["t_prefilter_s"], ref["t_align_s"], ref["t_index_s"])}
    if gold and tmp:
        def db_digest(base, n):
            data = np.fromfile(base, dtype=np.uint8)
            rows = np.loadtxt(base + ".index", dtype=np.int64, ndmin=2)
            rows = rows[np.argsort(rows[:, 0], kind="stable")]
            counts, bodies = np.zeros(n, dtype="<u8"), []
            for k in range(n):
                b = data[rows[k, 1]:rows[k, 1] + rows[k, 2] - 1].tobytes()
                counts[k] = b.count(b"\n")
                bodies.append(b)
            h = hashlib.sha256()
            h.update(counts.tobytes())
            h.update(b"".join(bodies))
            return h.hexdigest(), int(counts.sum())
        N, nq = gold["target_splits"], gold["n_queries"] + gold.get("n_long_queries", 0)
        thr = str(int(api.lib().mk_host_threads()))
        t0 = time.time()
        r1 = subprocess.run([build.BIN, "prefilter", os.path.join(tmp, "Q"), os.path.join(tmp, "T"), os.path.join(tmp, "pref"), "--split", str(N), "--split-mode", "0", "-s", "5.7",
                             "--ref-l2-bytes", str(gold["host_l2_bytes"]), "--threads", thr], stderr=subprocess.PIPE)
        t_pref = time.time() - t0
        t0 = time.time()
        r2 = subprocess.run([build.BIN, "align", os.path.join(tmp, "Q"), os.path.join(tmp, "T"), os.path.join(tmp, "pref"), os.path.join(tmp, "aln"), "--alignment-mode", "2", "-e", "100",
                             "--min-aln-len", "11", "--threads", thr], stderr=subprocess.PIPE) if r1.returncode == 0 else None
        t_aln = time.time() - t0
        if r1.returncode == 0 and r2 is not None and r2.returncode == 0:
            dp, nh = db_digest(os.path.join(tmp, "pref"), nq)
            da, na = db_digest(os.path.join(tmp, "aln"), nq)
            out["result_digest"] = {"fragments": nq, "target_splits": N, "gpu": {"prefilter": dp, "alignments": da}, "cpu": {"prefilter": gold["sha256_pref"], "alignments": gold["sha256_aln"]},
                                    "match": dp == gold["sha256_pref"] and da == gold["sha256_aln"], "prefilter_hits": nh, "alignments": na,
                                    "prefilter_command_s": round(t_pref, 1), "align_command_s": round(t_aln, 1),
                                    "reference": "tests/golden/%s: oracle/_ref/ref_harness pipeline --split %d (the reference's TARGET_DB_SPLIT run, mergeTargetSplits)" % (os.path.basename(golds[-1]), N)}
        else:
            out["result_digest"] = {"match": False, "error": ((r1.stderr if r1.returncode else r2.stderr) or b"").decode()[-400:]}
        shutil.rmtree(tmp, ignore_errors=True)
    elif gold:
        out["result_digest"] = {"match": None, "skipped": "no scratch directory with room for the 23 GB sequence DB" if args.config5_digest else "--config5-digest 0"}
    return out


def config5_leg(args):
    """the config-5 leg in a process of its own; its one JSON line comes back on stdout"""
    cmd = [sys.executable, os.path.abspath(__file__), "--config5-only", "--config5-targets", str(args.config5_targets), "--config5-fragments", str(args.config5_fragments),
           "--config5-digest", str(args.config5_digest)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=args.config5_timeout)
    except subprocess.TimeoutExpired:
        return {"error": "the config-5 leg did not finish in %d s" % args.config5_timeout}
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "config-5 leg failed (rc %d): %s" % (r.returncode, r.stderr.decode()[-600:])}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=10000)
    ap.add_argument("--targets", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--two-calls", action="store_true", help="mk_prefilter then mk_align instead of the pipelined mk_search")
    ap.add_argument("--blocking", action="store_true", help="time the blocking mk_search (one batch at a time) as the headline figure instead of the "
                    "queued mk_search_begin / mk_search_wait loop")
    ap.add_argument("--queue-depth", type=int, default=2, help="batches begun before the oldest one is collected (queued mode)")
    ap.add_argument("--blocking-steps", type=int, default=5, help="steps of the blocking mk_search timed after the headline region and reported beside it "
                    "(`blocking`; 0 = skip)")
    ap.add_argument("--alone-steps", type=int, default=2, help="steps of mk_prefilter alone (no alignment stage beside it) after the headline region: the dominant "
                    "kernel's time per step with the GPU to itself, for request_roofline (0 = skip)")
    ap.add_argument("--cpu-sample", type=int, default=400000, help="queries in the CPU baseline sample (0 = skip)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None, help="multi-GPU mode (default: strong when --gpus > 1)")
    ap.add_argument("--config4-profiles", type=int, default=50000, help="BASELINE config 4 beside the headline number (N = 1 only): this many synthetic "
                    "profiles searched against the same fragments, profiles as queries (0 = skip)")
    ap.add_argument("--target-index", default=None, help="path of a createindex DB: rank 0 writes it when it is missing, EVERY rank then loads the target side "
                    "from that one file (mk_targetdb_open_index) instead of masking and indexing its own replica")
    ap.add_argument("--digest-queries", type=int, default=-1, help="queries of the last timed step whose hits and alignments are compared with the reference's own "
                    "run (-1 = all of them, in pieces of --cpu-sample queries outside the timed region; 0 = only the CPU-baseline sample)")
    ap.add_argument("--e2e-sample", type=int, default=500, help="contigs of the end-to-end leg (`metaeuk-amd predictexons` over DBs on disk, N = 1 only) whose exon sets are "
                    "compared with the reference's chain; -1 = skip the leg")
    ap.add_argument("--config5-targets", type=int, default=60000000, help="BASELINE config 5's target side beside the headline number (N = 1 only, a process of its own): "
                    "this many proteins of the native generator, k = 7, masked and indexed in HBM; 0 = skip the leg")
    ap.add_argument("--config5-fragments", type=int, default=20000, help="planted fragments of the config-5 leg's unsplit search")
    ap.add_argument("--config5-digest", type=int, default=1, help="1: the leg also runs the committed golden's fragments through `metaeuk-amd prefilter --split N` + `align` and "
                    "compares the digests with the reference's recorded TARGET_DB_SPLIT run (writes a 23 GB sequence DB to scratch)")
    ap.add_argument("--config5-timeout", type=int, default=900)
    ap.add_argument("--config5-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--config4-sample", type=int, default=2048, help="profiles of the config-4 leg whose hits and alignments are compared with the reference")
    args = ap.parse_args()

    if args.config5_only:
        print(json.dumps(config5_leg_child(args)))
        return
    # the one JSON line goes to the real stdout; whatever libraries print there (RCCL's version banner at communicator creation) is
    # sent to stderr instead
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its workers unless the caller set the variable.  The host side of the path
    # (result assembly, --max-seqs tie logic, e-value tables) is OpenMP code that the library sizes itself -- the cores this process
    # may use divided by LOCAL_WORLD_SIZE (mk_init) -- so the launcher's placeholder is dropped; MK_HOST_THREADS pins it explicitly.
    if world > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
        del os.environ["OMP_NUM_THREADS"]
    if os.environ.get("MK_HOST_THREADS"):
        os.environ["OMP_NUM_THREADS"] = os.environ["MK_HOST_THREADS"]
    dist = None
    if world > 1 or os.environ.get("MK_BENCH_FORCE_DIST"):      # the variable exercises the RCCL path on a single GPU
        import torch
        import torch.distributed as dist_
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist_.init_process_group(backend="nccl")
        dist = dist_

    from metaeuk_amd import api
    api.init(local_rank)
    params = api.default_params()
    # the reference sizes its hit bins from the L2 of the host it runs on (Util::getL2CacheSize, Util.cpp:317-332), which decides
    # the tie order at the --max-seqs cut: reproduce THIS host's reference run, like the metaeuk-amd commands do (mk_cli.cpp)
    import ctypes
    l2 = ctypes.CDLL(None).sysconf(191)          # _SC_LEVEL2_CACHE_SIZE
    params.host_l2_bytes = l2 if l2 and l2 > 0 else 262144

    scaling = args.scaling or ("strong" if world > 1 else "weak")
    t0 = time.time()
    targets, queries, founders = make_inputs(args.contigs, args.targets, args.seed, rank if scaling == "weak" else 0)
    if scaling == "strong" and world > 1:
        # BASELINE config 3: the same workload, query-sharded with the reference's residue-balanced rule
        from metaeuk_amd import shard
        first, count = shard.decompose_by_residues([len(x) + 2 for x in queries], rank, world)
        queries = queries[first:first + count]
    t_res, t_off = pack(targets)
    q_res, q_off = pack(queries)
    t_gen = time.time() - t0
    t0 = time.time()
    if args.target_index:
        if rank == 0 and not os.path.exists(args.target_index + ".dbtype"):
            api.index_write(args.target_index, api.synth_seqdb(t_res, t_off), params)
        if dist is not None:
            dist.barrier()
        db = api.TargetDB.from_index(args.target_index, params)
    else:
        db = api.TargetDB.from_codes(t_res, t_off, params)
    t_index = time.time() - t0

    def barrier():
        # every library call returns with its results on the host (mk_search ends in stream synchronisations), so with one rank
        # there is nothing in flight to wait for; with several, the device sync + RCCL barrier bracket the timed region
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    last = {}

    def step(keep=False):
        q = api.Queries.from_codes(q_res, q_off, params)
        if args.two_calls:
            hits, hoff = api.prefilter(db, q)
            alns, aoff = api.align(db, q)
        else:
            (hits, hoff), (alns, aoff) = api.search(db, q)
        res = (int(hoff[-1]), int(aoff[-1]))
        if keep:                                   # the last step's results stay alive for the digest (outside the timed region)
            last.update(q=q, hits=hits, hoff=hoff, alns=alns, aoff=aoff)
        else:
            q.close()
        return res

    def collect(q, keep=False):
        (hits, hoff), (alns, aoff) = api.search_wait(q)
        res = (int(hoff[-1]), int(aoff[-1]))
        if keep:
            last.update(q=q, hits=hits, hoff=hoff, alns=alns, aoff=aoff)
        else:
            q.close()
        return res

    def run_steps(n, keep_last):
        """n steps; the result of every step is on the host when this returns.  Default: the batches are QUEUED (mk_search_begin /
        mk_search_wait) -- batch k + 1 is uploaded, derived and begun before the results of batch k are collected, so the prefilter of
        k + 1 runs beside the alignment tail of k (what a caller that walks a DB in batches gets, e.g. `metaeuk-amd predictexons`).
        --blocking / --two-calls: one batch at a time."""
        res = (0, 0)
        if args.two_calls or args.blocking:
            for k in range(n):
                res = step(keep=(keep_last and k == n - 1))
                step_counts.append(res)
            return res
        pending = []
        for k in range(n):
            q = api.Queries.from_codes(q_res, q_off, params)
            api.search_begin(db, q)
            pending.append(q)
            if len(pending) >= max(1, args.queue_depth):
                step_counts.append(collect(pending.pop(0)))
        while pending:
            q = pending.pop(0)
            res = collect(q, keep=keep_last and not pending)
            step_counts.append(res)
        return res

    step_counts = []           # (hits, alignments) of every step, warm-up included: the same input every step, so the counts must not move
    run_steps(args.warmup, False)
    api.kernel_stats(reset=True)
    barrier()
    t0 = time.time()
    nhits, npass = run_steps(args.steps, True)
    barrier()
    elapsed = time.time() - t0
    stats = api.kernel_stats()
    blocking = None
    if not (args.two_calls or args.blocking) and args.blocking_steps > 0:
        # the same step with the blocking call, one batch at a time (round 4's figure), outside the headline region
        barrier()
        tb0 = time.time()
        for _ in range(args.blocking_steps):
            step()
        barrier()
        blocking = {"ms_per_step": (time.time() - tb0) / args.blocking_steps * 1e3, "steps": args.blocking_steps,
                    "note": "mk_search, one batch at a time: every step pays the fill and drain of the two-stage pipeline"}
    alone = None
    if not (args.two_calls or args.blocking) and args.alone_steps > 0 and world == 1:
        # the prefilter stage with the GPU to itself (mk_prefilter, no alignment workers beside it): the dominant kernel's time per step alone, for
        # the request roofline's co-resident / alone pair.  Outside the headline region
        api.kernel_stats(reset=True)
        for _ in range(args.alone_steps):
            qa = api.Queries.from_codes(q_res, q_off, params)
            api.prefilter(db, qa)
            qa.close()
        alone = {k: v["ms"] / args.alone_steps for k, v in api.kernel_stats().items()}
    total_queries = len(queries)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tq = torch.tensor([float(len(queries))], dtype=torch.float64, device="cuda")   # ranks draw different contigs
        dist.all_reduce(tq, op=dist.ReduceOp.SUM)
        total_queries = int(tq.item())

    nq = len(queries)
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    frag_per_s = total_queries * args.steps / elapsed
    cells_sw = sum(v["cells"] for k, v in stats.items() if k.startswith("sw_"))
    gcups_total = world * cells_sw / elapsed / 1e9
    sw_ms = sum(v["ms"] for k, v in stats.items() if k.startswith("sw_"))

    def _packed(name):
        return name.startswith("sw_fwd_rows") and int(name[len("sw_fwd_rows"):]) <= PACKED_ROWS_MAX
    sw_lane_ops = sum(v["cells"] * (SW_OPS_PER_CELL_PACKED if _packed(k) else SW_OPS_PER_CELL_INT32) for k, v in stats.items() if k.startswith("sw_"))
    kstats = {k: v for k, v in stats.items() if not k.startswith(("host_", "wait_"))}
    if rank == 0:
        for k, v in sorted(stats.items()):
            print("# %-24s ms/step %10.2f launches/step %8.1f bytes/step %.4g cells/step %.4g" % (
                k, v["ms"] / max(args.steps, 1), v["launches"] / max(args.steps, 1), v["alg_bytes"] / max(args.steps, 1),
                v["cells"] / max(args.steps, 1)), file=sys.stderr)
    dom_name, dom = max(kstats.items(), key=lambda kv: kv[1]["ms"]) if kstats else ("none", dict(ms=0, launches=1, alg_bytes=0, cells=0))
    per_launch_ms = dom["ms"] / max(dom["launches"], 1)
    achieved = (dom["alg_bytes"] / max(dom["launches"], 1)) / max(per_launch_ms * 1e-3, 1e-12) / 1e9
    traffic, traffic_note, traffic_dram = pmc_traffic(dom_name)
    line = {
        "metric": "prefilter+align ORF-fragments/sec (hits UNVERIFIED in this run: no CPU reference digest)",
        "value": frag_per_s, "unit": "fragments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "packed i16 / i32 DP on u8 residues, i8 scores", "data": "synthetic",
        "config": {"workload": "predictexons hot path: %d synthetic 5-kb contigs (%d ORF fragments/rank, %d aa) x %d-protein DB (%d aa), -s 5.7" % (
            args.contigs, nq, int(q_off[-1]), args.targets, int(t_off[-1])), "parallelism": "query-shard x%d" % world,
            "seed": args.seed},
        "mode": "two calls" if args.two_calls else ("blocking mk_search" if args.blocking else "queued batches (mk_search_begin / mk_search_wait): every "
                "result of every step on the host inside the timed region"),
        "blocking": blocking,
        "gcups_sw": gcups_total,
        "gcups_sw_kernel_only": (cells_sw / max(sw_ms * 1e-3, 1e-12) / 1e9) if sw_ms else None,      # (kernel durations summed: overlapping launches count twice)
        "prefilter_hits": nhits, "alignments_passed": npass,
        "steps_with_other_counts": sum(1 for c in step_counts if c != (nhits, npass)),      # every step searches the same batch: 0, or a race
        "setup_s": {"generate": round(t_gen, 2), "target_index_build_upload": round(t_index, 2), "target_from_index_db": bool(args.target_index)},
        "host": {"threads": int(api.lib().mk_host_threads()) if not os.environ.get("OMP_NUM_THREADS") else int(os.environ["OMP_NUM_THREADS"]),
                 "max_rss_mb": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1024.0, 1)},
        "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(stats.items())},
        # dominant kernel against HBM; `traffic` (PMC FETCH_SIZE + WRITE_SIZE per launch) comes from the rocprofv3 passes kept
        # under profiles/ -- bench.py cannot read PMC counters itself
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_fabric": traffic, "traffic_dram": traffic_dram, "traffic_source": traffic_note,
                     "avg_launch_ms": per_launch_ms, "launches": dom["launches"],
                     "note": "achieved = SURVEY 8(d) algorithmic bytes (16 B per probed k-mer + 6 B per index entry) / launch time; traffic = "
                             "fabric bytes per launch from the L2 request counters (PMC): every 8-byte index probe that misses L2 fetches a "
                             "128-byte line, so the prefilter kernels move ~5x their algorithmic bytes and are bound by HBM line bandwidth "
                             "(~5 TB/s when they run alone); the Smith-Waterman kernels are vector-ALU bound (valu_roofline).  traffic_fabric = traffic; "
                             "traffic_dram = the part of it bound for the memory controllers (TCC_EA0_*REQ_DRAM; the Infinity Cache is on that side of the fabric "
                             "and no counter separates its hits).  The bound is the FABRIC's request rate, not DRAM: random 8-byte reads run at 56.8 G/s from a "
                             "128 MB table that the 256 MB Infinity Cache holds and at 54.4 G/s from 2 GB (profiles/r02_random_probe_rates.txt)"},
        # the Smith-Waterman kernels against the integer vector-ALU issue rate: the cell arithmetic alone (5 lane-ops per cell in the packed
        # score pass, 10 in the int32 passes), not counting the wavefront's hand-over instructions, padding rows or fill / drain steps
        # (denominator: the wall clock of the timed region -- the kernels of the two alignment workers overlap each other and the prefilter's,
        # so the sum of their own durations counts the same seconds more than once)
        "valu_roofline": {"kernels": "sw_fwd_* + sw_pos_* + sw_rev_*", "achieved": (sw_lane_ops / max(elapsed, 1e-12) / 1e12) if sw_ms else None,
                          "peak": VALU_PEAK_TOPS, "unit": "Tlane-op/s",
                          "frac": (sw_lane_ops / max(elapsed, 1e-12) / 1e12 / VALU_PEAK_TOPS) if sw_ms else None,
                          "peak_guide": 2.0 * VALU_PEAK_TOPS,
                          "frac_guide": (sw_lane_ops / max(elapsed, 1e-12) / 1e12 / (2.0 * VALU_PEAK_TOPS)) if sw_ms else None,
                          "note": "lane-ops of the DP cells per second of the whole step (the prefilter shares the GPU); peak = one integer / packed-int16 "
                                  "wave-instruction per 4 cycles per SIMD (MEASURED on this chip with 8 independent chains per wave, flat from 4 to 16 waves "
                                  "per CU: profiles/r02_valu_issue_rates.txt, tools/micro/hammer.hip) -- the figure this line trusts; peak_guide / frac_guide price "
                                  "the same work against MI355X_MICROARCH.md's SIMD-32 with a 2-cycle wave64 issue (twice the measured rate)"},
    }
    # one launch definition for both byte counts: PER STEP.  Algorithmic bytes of the dominant kernel per step (this run), fabric bytes per pass of the
    # same workload from the PMC artefact (its run is warm-up + 1 step = `passes` passes), their ratio; and the same kernel against the chip's measured
    # random-request rate -- the bound DESIGN.md 4.1 names -- beside the other stage (this run's event time) and alone (mk_prefilter on its own)
    steps_n = max(args.steps, 1)
    rl = line["roofline"]
    rl["per_step"] = {"algorithmic_bytes": dom["alg_bytes"] / steps_n, "kernel_ms": dom["ms"] / steps_n, "launches": dom["launches"] / steps_n}
    if PMC_LAST:
        fabric = (PMC_LAST["fetch_bytes"] + PMC_LAST["write_bytes"]) / max(PMC_LAST["passes"], 1)
        rl["per_step"].update({"fabric_bytes": fabric, "fabric_over_algorithmic": fabric / max(dom["alg_bytes"] / steps_n, 1.0),
                               "pmc_launches_per_pass": PMC_LAST["launches"] / max(PMC_LAST["passes"], 1)})
        req = (PMC_LAST.get("read_requests", 0.0) + PMC_LAST.get("write_requests", 0.0)) / max(PMC_LAST["passes"], 1)
        rr = {"kernel": dom_name, "requests_per_step": req, "peak": RANDOM_REQUEST_PEAK, "unit": "requests/s",
              "co_resident": {"kernel_ms_per_step": dom["ms"] / steps_n, "achieved": req / max(dom["ms"] / steps_n * 1e-3, 1e-12)},
              "note": "fabric requests (reads + writes, PMC) of the dominant kernel per step / its event time per step, against the measured rate of independent "
                      "random 128-byte requests (profiles/r02_random_probe_rates.txt); co_resident = beside the alignment stage in the timed region, alone = "
                      "mk_prefilter with the GPU to itself"}
        rr["co_resident"]["frac"] = rr["co_resident"]["achieved"] / RANDOM_REQUEST_PEAK
        if alone and alone.get(dom_name):
            rr["alone"] = {"kernel_ms_per_step": alone[dom_name], "achieved": req / max(alone[dom_name] * 1e-3, 1e-12)}
            rr["alone"]["frac"] = rr["alone"]["achieved"] / RANDOM_REQUEST_PEAK
            rl["alone"] = {"kernel_ms_per_step": alone[dom_name], "achieved": dom["alg_bytes"] / steps_n / max(alone[dom_name] * 1e-3, 1e-12) / 1e9,
                           "frac": dom["alg_bytes"] / steps_n / max(alone[dom_name] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS}
        line["request_roofline"] = rr
    if rank == 0 and world == 1 and args.config4_profiles > 0:
        # BASELINE config 4 (profile targets: the reference's inverted search -- profiles as queries, the fragments as the indexed side,
        # swapresults), reported beside the headline metric and outside its timed region.  DESIGN.md 4.7; parity: tests/test_gpu_profile.py.
        # A failure here is a failure of the run (no blanket except): the leg is part of the reported result.
        line["config4_profile_targets"] = config4_leg(api, args, params, q_res, q_off, nq)
    if rank == 0 and world == 1 and args.e2e_sample >= 0:
        line["end_to_end_predictexons"] = e2e_leg(api, args, params, targets, founders, t_res, t_off)
    run_config5 = rank == 0 and world == 1 and args.config5_targets > 0
    if rank == 0:
        if world == 1 and args.cpu_sample > 0:
            n_s = min(args.cpu_sample, nq)
            try:
                line["cpu_baseline"] = cpu_baseline(targets, queries, n_s, int(api.lib().mk_host_threads()))
            except Exception as e:  # the baseline is reported, never required
                line["cpu_baseline"] = {"value": None, "unit": "fragments/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
            # "bit-exact": SHA-256 over the formatted hits + alignments of the LAST TIMED STEP, restricted to the CPU sample's queries,
            # against the same digest of what the CPU baseline wrote (tests/oracle.py: digest_arrays / digest_blocks_file)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import hashlib
                g = gpu_digest(api, last["hits"], last["hoff"], last["alns"], last["aoff"], n_s)
                c = line["cpu_baseline"].pop("digest", None)
                pieces = [{"first": 0, "queries": n_s, "match": (c == g) if c else None}]
                gs, cs = [g], [c]
                # the rest of the step's result, piece by piece against further runs of the reference (outside the timed region, ~19 s per
                # 400 000 fragments on 16 cores): every query of the last timed step is compared
                want = nq if args.digest_queries < 0 else min(nq, max(n_s, args.digest_queries))
                t_more = time.time()
                at = n_s
                while c and at < want:
                    m = min(n_s, want - at)
                    cb = cpu_baseline(targets, queries, m, int(api.lib().mk_host_threads()), first=at)
                    gp = gpu_digest(api, last["hits"], last["hoff"], last["alns"], last["aoff"], m, first=at)
                    pieces.append({"first": at, "queries": m, "match": cb["digest"] == gp})
                    gs.append(gp); cs.append(cb["digest"])
                    at += m
                fold = lambda ds, key: hashlib.sha256("".join(d[key] for d in ds).encode()).hexdigest() if all(ds) else None
                line["result_digest"] = {"queries": at if c else n_s, "of": nq, "covered": round((at if c else n_s) / max(nq, 1), 4), "pieces": len(pieces),
                                         "gpu": {"prefilter": fold(gs, "prefilter"), "alignments": fold(gs, "alignments")},
                                         "cpu": {"prefilter": fold(cs, "prefilter"), "alignments": fold(cs, "alignments")} if c else None,
                                         "match": all(p["match"] for p in pieces) if c else None,
                                         "mismatching_pieces": [p["first"] for p in pieces if p["match"] is False],
                                         "reference_s_beyond_the_sample": round(time.time() - t_more, 1),
                                         "note": "SHA-256 over the formatted hits and alignments of the LAST TIMED STEP, piece by piece (each piece's digest against the "
                                                 "reference harness's own output for the same queries); gpu / cpu = SHA-256 over the pieces' digests"}
            except Exception as e:
                line["result_digest"] = {"queries": n_s, "error": repr(e)}
        if run_config5:
            # (this process gives its HBM back first: the 60 M-protein index needs 230 of the 288 GB)
            last.clear()
            db.close()
            api.shutdown()
            line["config5_60M"] = config5_leg(args)
        # the metric names what was CHECKED in this run: matched / mismatched / not compared
        verdicts = [line.get("result_digest", {}).get("match")]
        if line.get("config5_60M", {}).get("result_digest", {}).get("match") is not None:
            verdicts.append(line["config5_60M"]["result_digest"]["match"])
        if "config4_profile_targets" in line:
            verdicts.append(line["config4_profile_targets"].get("result_digest", {}).get("match"))
        if line.get("end_to_end_predictexons", {}).get("result_digest", {}).get("match") is not None:
            verdicts.append(line["end_to_end_predictexons"]["result_digest"]["match"])
        if any(v is False for v in verdicts):
            line["metric"] = "prefilter+align ORF-fragments/sec (RESULT MISMATCH vs the CPU reference)"
        elif verdicts[0] is True and all(v is True for v in verdicts):
            line["metric"] = "prefilter+align ORF-fragments/sec (bit-exact hits)"
        else:
            line["metric"] = "prefilter+align ORF-fragments/sec (hits UNVERIFIED in this run: no CPU reference digest)"
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
