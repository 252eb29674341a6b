"""GPU parity at the headline scale: the 100 000-protein target DB of BASELINE config 2 (the queries are a 300-contig
slice of its 10 000), mk_search against the reference's own compiled code (oracle/_ref/ref_harness, when it travelled
to the box) or the C oracle.  At this size the prefilter behaves differently in kind from the small fixtures: most
similar k-mers belong to queries whose index hits do not fit one workgroup's LDS, --max-seqs truncation with the real
BINSIZE fires, and the alignments cross every Smith-Waterman tile shape.
Reference: QueryMatcher::matchQuery (M/src/prefiltering/QueryMatcher.cpp:149-209,422-450), Alignment::run."""
import ctypes
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _blocks(body, off):
    lines = body.split(b"\n")
    return [b"".join(l + b"\n" for l in lines[int(off[i]):int(off[i + 1])]).decode() for i in range(len(off) - 1)]


def test_headline_scale_parity(gpu_api, tmp_path):
    from metaeuk_amd import synth
    api = gpu_api
    targets, queries = synth.make_workload(300, 100000, seed=11)
    assert len(targets) == 100000 and len(queries) > 50000
    params = api.default_params()
    use_ref = os.path.exists(oracle.REF)
    if use_ref:       # the reference sizes BINSIZE from the L2 of the host it runs on (Util::getL2CacheSize, Util.cpp:317-332)
        l2 = ctypes.CDLL(None).sysconf(191)
        params.host_l2_bytes = l2 if l2 and l2 > 0 else 262144
    db = api.TargetDB(targets, params)
    q = api.Queries(queries, params)
    api.kernel_stats(reset=True)
    (hits, hoff), (alns, aoff) = api.search(db, q)
    stats = api.kernel_stats()
    hbody = api.format_hits_bulk(hits, 0, int(hoff[-1]))
    abody = api.format_alignments_bulk(alns, 0, int(aoff[-1]))
    if use_ref:
        rpref, raln = oracle.run_ref_pipeline(targets, queries, str(tmp_path), extra=["--threads", str(api.lib().mk_host_threads())])
        sub = "ref"
    else:
        rpref, raln = oracle.run_pipeline(targets, queries, str(tmp_path), extra=["--l2", str(params.host_l2_bytes)])
        sub = "oracle"
    gp, ga = _blocks(hbody, hoff), _blocks(abody, aoff)
    bad_p = [i for i in range(len(queries)) if gp[i] != rpref[i]]
    bad_a = [i for i in range(len(queries)) if ga[i] != raln[i]]
    assert not bad_p, "%d prefilter blocks differ, first: query %d\n%s\nvs\n%s" % (len(bad_p), bad_p[0], gp[bad_p[0]][:300], rpref[bad_p[0]][:300])
    assert not bad_a, "%d alignment blocks differ, first: query %d\n%s\nvs\n%s" % (len(bad_a), bad_a[0], ga[bad_a[0]][:400], raln[bad_a[0]][:400])
    # the same through the digests bench.py prints
    for name, off, body in (("pref.txt", hoff, hbody), ("aln.txt", aoff, abody)):
        d_cpu, n = oracle.digest_blocks_file(os.path.join(str(tmp_path), sub, name))
        assert n == len(queries)
        assert oracle.digest_arrays(off, body, len(queries)) == d_cpu
    # this workload must exercise what the small fixtures cannot
    counts = np.diff(np.asarray(hoff))
    max_hits = min(params.max_seqs, len(targets))
    assert int((counts == max_hits).sum()) >= 1, "no query reached the --max-seqs cut"
    assert int((counts > max_hits).sum()) == 0
    # similar k-mers by front end: per-query kernels by region size (2 K / 4 K one wave per query, 8 K / 128 K several waves), global path
    k_small = sum(v["cells"] for k, v in stats.items() if k in ("prefilter_query_cap2048", "prefilter_query_cap4096", "prefilter_query_cap8192"))
    k_big = sum(v["cells"] for k, v in stats.items() if k.startswith("prefilter_query_cap") and int(k[len("prefilter_query_cap"):]) > 8192)
    k_glob = stats.get("kmer_probe_gather", {"cells": 0})["cells"]
    print("similar k-mers: queries <= 8 K hits %.3g, <= 128 K hits %.3g, global path %.3g" % (k_small, k_big, k_glob))
    assert k_big + k_glob > k_small, "most similar k-mers should belong to queries with more than 8 K index hits (%g + %g vs %g)" % (k_big, k_glob, k_small)
    assert k_glob > 0, "the global path should see the largest queries"
    assert int(aoff[-1]) > 100000 and int(hoff[-1]) > 1000000


def test_profile_path_scale_parity(gpu_api, tmp_path):
    """BASELINE config 4 at a tenth of its size: 5 000 synthetic profiles as queries against the ~400 000 fragments of 2 000 contigs
    (what the small fixtures cannot reach: pieces of hundreds of millions of listed k-mers, ~18 000 index hits per profile, every
    Smith-Waterman tile up to 768 rows incl. the 16-lane variant, thousands of accepted alignments per profile in the rank-count order),
    against the C oracle's restatement of the reference path on a sample of the profiles."""
    import subprocess
    from metaeuk_amd import synth
    api = gpu_api
    n_prof, sample = 5000, 160
    proteins, founders = synth.make_targets(n_prof, seed=11)
    entries = synth.make_profiles(proteins, seed=11)
    frags = [synth.codes_to_str(c) for c in synth.make_queries(2000, founders, seed=11)]
    assert len(frags) > 300000
    p = api.default_params()
    p.sensitivity = 4.0
    p.profile_search = 1
    p.max_seqs = max(300, len(frags))
    p.evalue_thr = float("%g" % (100.0 * (np.float32(len(frags)) / np.float32(n_prof))))
    p.host_l2_bytes = 2097152
    db = api.TargetDB(frags, p)
    q = api.Profiles(entries, p)
    api.kernel_stats(reset=True)
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    st = api.kernel_stats()
    assert int(hoff[-1]) > 500000 and int(aoff[-1]) > 200000
    assert st["profile_kmer_fill"]["cells"] > 2e8 and "sw_fwd_rows768" in st and "sw_fwd_rows384" in st
    # the oracle on the first `sample` profiles: same fragments, same thresholds (-s 4, the e-value threshold of the whole search)
    (tmp_path / "p.bin").write_bytes(b"".join(entries[:sample]))
    off, lines = 0, []
    for k, e in enumerate(entries[:sample]):
        lines.append("%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
    (tmp_path / "p.index").write_text("".join(lines))
    (tmp_path / "f.txt").write_text("\n".join(frags) + "\n")
    subprocess.check_call([oracle.CLI, "profilesearch", str(tmp_path / "p.bin"), str(tmp_path / "p.index"), str(tmp_path / "f.txt"), str(tmp_path / "out"),
                           "-s", "4", "--eval-abs", repr(float(p.evalue_thr)), "--l2", "2097152"], stdout=subprocess.DEVNULL)
    pref = "".join(">%d\n%s" % (i, api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])).decode()) for i in range(sample))
    aln = "".join(">%d\n%s" % (i, api.format_alignments_bulk(alns, int(aoff[i]), int(aoff[i + 1])).decode()) for i in range(sample))
    assert pref == open(tmp_path / "out" / "pref.txt").read()
    assert aln == open(tmp_path / "out" / "aln.txt").read()
    assert max(int(aoff[i + 1] - aoff[i]) for i in range(sample)) > 200


def _config5_case(api, tmp_path, n_targets, forced_k, n_queries=20000, index_db=True):
    """A database shaped like BASELINE config 5's (native generator: families of ten, 375 residues on average), masked and indexed ON THE
    DEVICE, searched with planted fragments; the same database through a k = 7 index DB (written while the lists stream out of HBM, read
    back from the mapped file in pieces) must give the same tables and the same bytes; the digest of the result against the reference's
    own run over the same seeded input (tests/golden/config5_digest_<n>.json, tools/config5_digest.py)."""
    import json
    import sys
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config5_digest as c5
    t0 = time.time()
    res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
    fr, foff, src = api.synth_fragments(n_queries, res, off, **c5.FRAGMENTS)
    t_gen = time.time() - t0
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config5_digest_%d.json" % n_targets)
    gold = json.load(open(gold_path)) if os.path.exists(gold_path) else None
    p = api.default_params()
    p.kmer_size = forced_k
    if gold:
        assert gold["target_residues"] == int(off[-1]) and gold["n_queries"] == n_queries and gold["fragments"] == c5.FRAGMENTS
        p.host_l2_bytes = gold["host_l2_bytes"]
    api.kernel_stats(reset=True)
    t0 = time.time()
    db = api.TargetDB.from_codes(res, off, p)
    t_build = time.time() - t0
    st = api.kernel_stats()
    assert db.kmer_size() == 7
    q = api.Queries.from_codes(fr, foff, p)
    t0 = time.time()
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    t_search = time.time() - t0
    hbody = api.format_hits_bulk(hits, 0, int(hoff[-1]))
    abody = api.format_alignments_bulk(alns, 0, int(aoff[-1]))
    report = dict(n_targets=n_targets, target_residues=int(off[-1]), index_entries=db.index_entries(), masked_residues=db.masked_residues(),
                  longest_list=db.longest_list(), t_generate_s=round(t_gen, 2), t_targetdb_s=round(t_build, 2), t_search_s=round(t_search, 2),
                  index_kernels_ms={k: round(v["ms"], 1) for k, v in st.items() if k.startswith("index_") or k.startswith("host_index")},
                  pref_hits=int(hoff[-1]), alignments=int(aoff[-1]))
    # planted homologs: the fragment's source (or a member of its family) is among its alignments
    alns_np = np.frombuffer(alns, dtype=np.dtype([("db_key", "<u4"), ("rest", "V60")])) if int(aoff[-1]) else None
    planted = np.flatnonzero(src != 0xFFFFFFFF)
    found = 0
    for k in planted[:4000]:
        keys = alns_np["db_key"][int(aoff[k]):int(aoff[k + 1])] if alns_np is not None else []
        found += int(np.any(keys // 10 == src[k] // 10))
    recall = found / float(min(len(planted), 4000))
    report["planted_recall"] = round(recall, 4)
    background = np.flatnonzero(src == 0xFFFFFFFF)
    report["alignments_per_background_fragment"] = float(np.mean([int(aoff[k + 1] - aoff[k]) for k in background[:2000]]))
    assert recall > 0.97, report
    d_pref = oracle.digest_arrays(hoff, hbody, n_queries)
    d_aln = oracle.digest_arrays(aoff, abody, n_queries)
    if gold:
        report["reference"] = gold["reference"]
        assert (int(hoff[-1]), d_pref) == (gold["reference"]["pref_hits"], gold["sha256_pref"]), report
        assert d_aln == gold["sha256_aln"], report
    if index_db:
        # the index DB of a k = 7 database: 10.2 GB of list offsets + 6 B per entry + the sequences.  A GPU box's scratch disk may be smaller
        # than it looks (ephemeral-storage limit of the pod): the round trip is skipped -- and said so in the report -- when the file cannot
        # be written; the writer and the reader themselves are pinned at small scale (tests/test_gpu_index.py)
        import shutil
        need = 8 * 1280000000 + 6 * db.index_entries() + 2 * int(off[-1])
        free = shutil.disk_usage(str(tmp_path)).free
        image = api.synth_seqdb(res, off)
        t0 = time.time()
        written = False
        if free > 1.15 * need:
            try:
                api.index_write(str(tmp_path / "T.idx"), image, p)
                written = True
            except api.MkError as e:
                report["index_db_round_trip"] = "skipped: %s (%.1f GB needed, %.1f GB reported free)" % (e, need / 1e9, free / 1e9)
        else:
            report["index_db_round_trip"] = "skipped: %.1f GB needed, %.1f GB free on the scratch disk" % (need / 1e9, free / 1e9)
        del image
        if not written:
            for ext in ("", ".index", ".dbtype"):
                if os.path.exists(str(tmp_path / "T.idx") + ext):
                    os.remove(str(tmp_path / "T.idx") + ext)
            index_db = False
    if index_db:
        report["t_index_db_write_s"] = round(time.time() - t0, 2)
        report["index_db_bytes"] = os.path.getsize(str(tmp_path / "T.idx"))
        t0 = time.time()
        db2 = api.TargetDB.from_index(str(tmp_path / "T.idx"), p)
        report["t_index_db_open_s"] = round(time.time() - t0, 2)
        assert db2.kmer_size() == 7 and db2.index_entries() == db.index_entries()
        assert db.index_compare(db2) == (0, 0, 0, 0)
        db.close()
        q2 = api.Queries.from_codes(fr, foff, p)
        (hits2, hoff2), (alns2, aoff2) = api.search(db2, q2, p)
        assert oracle.digest_arrays(hoff2, api.format_hits_bulk(hits2, 0, int(hoff2[-1])), n_queries) == d_pref
        assert oracle.digest_arrays(aoff2, api.format_alignments_bulk(alns2, 0, int(aoff2[-1])), n_queries) == d_aln
        db2.close()
        for ext in ("", ".index", ".dbtype"):
            os.remove(str(tmp_path / "T.idx") + ext)
    print("config-5-shaped case:", json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config5_case_%d.json" % n_targets), "w") as f:
            json.dump(report, f, indent=1)
    return report


def test_config5_shape_1e9_residues_pinned_to_the_reference(gpu_api, tmp_path):
    """2.7 M proteins = 1.0e9 residues, -k 7: device-built index, planted homologs, and the result digest of 20 000 fragments equal to the
    reference's own run over the same input (the largest size its index build fits the build container: 62 GB of host memory)"""
    r = _config5_case(gpu_api, tmp_path, 2700000, 7, index_db=False)
    assert r["index_entries"] > 9e8


def test_config5_scale_4e9_residues_index_beyond_2_32_entries(gpu_api, tmp_path):
    """11.8 M proteins = 4.4e9 residues: k = 7 chosen automatically (IndexTable::computeKmerSize, from 3.35e9 residues on), a real index of
    more than 2^32 entries (40-bit list starts without MK_TEST_ENTRY_BASE), built in HBM, written as an index DB and loaded back"""
    r = _config5_case(gpu_api, tmp_path, int(os.environ.get("MK_TEST_CONFIG5_TARGETS", "11800000")), 0)
    assert r["target_residues"] >= 4.4e9 and r["index_entries"] > 2 ** 32


def test_config5_full_scale_60M_proteins_on_one_gpu(gpu_api, tmp_path, monkeypatch):
    """BASELINE config 5's target side at its stated size on ONE MI355X: 60 000 000 proteins (2.2e10 residues with the native generator's
    375-residue families -- UniRef50's ~1.8e10 and more), k = 7 chosen by IndexTable::computeKmerSize, masked and indexed in HBM (2.2e10
    entries: 175 GB of the 288), searched with planted fragments.  What is checked at this size: the index totals, planted-homolog recall,
    and that the two independent prefilter front ends -- the wide per-query kernel (7-mers enumerated in the kernel) and the sort-based
    global path (7-mers as lists) -- return byte-identical hits and alignments for a sample of the fragments.  NOT checked here: the
    reference's own run (its index build alone needs ~270 GB of host memory and minutes of a 16-core quota at this size; the same code
    paths are digest-equal to it at 1.0e9 and 4.4e9 residues above), and the index-DB round trip (133 GB of scratch disk)."""
    import json
    import sys
    import time
    api = gpu_api
    n_targets = int(os.environ.get("MK_TEST_CONFIG5_FULL", "60000000"))
    free, total = api.device_memory()
    if total < 280e9 and n_targets >= 60000000:
        pytest.skip("needs a 288 GB device")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config5_digest as c5
    n_queries = 4000
    t0 = time.time()
    res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
    fr, foff, src = api.synth_fragments(n_queries, res, off, **c5.FRAGMENTS)
    t_gen = time.time() - t0
    p = api.default_params()
    api.kernel_stats(reset=True)
    t0 = time.time()
    db = api.TargetDB.from_codes(res, off, p)
    t_build = time.time() - t0
    st = api.kernel_stats()
    assert db.kmer_size() == 7
    n_res = int(off[-1])
    report = dict(n_targets=n_targets, target_residues=n_res, index_entries=db.index_entries(), masked_residues=db.masked_residues(), longest_list=db.longest_list(),
                  t_generate_s=round(t_gen, 2), t_targetdb_s=round(t_build, 2),
                  index_kernels_ms={k: round(v["ms"], 1) for k, v in st.items() if k.startswith("index_") or k.startswith("host_index")},
                  device_memory_free_after_build_gb=round(api.device_memory()[0] / 1e9, 1))
    del res
    assert report["index_entries"] > 0.9 * n_res and report["index_entries"] < n_res
    q = api.Queries.from_codes(fr, foff, p)
    api.kernel_stats(reset=True)
    t0 = time.time()
    (hits, hoff), (alns, aoff) = api.search(db, q, p)
    report["t_search_s"] = round(time.time() - t0, 2)
    report["fragments"] = n_queries
    report["fragments_per_s"] = round(n_queries / (time.time() - t0), 1)
    report["search_kernels_ms"] = {k: round(v["ms"], 1) for k, v in sorted(api.kernel_stats().items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] >= 5.0}
    report["pref_hits"], report["alignments"] = int(hoff[-1]), int(aoff[-1])
    alns_np = np.frombuffer(alns, dtype=np.dtype([("db_key", "<u4"), ("rest", "V60")])) if int(aoff[-1]) else None
    planted = np.flatnonzero(src != 0xFFFFFFFF)
    found = sum(int(np.any(alns_np["db_key"][int(aoff[k]):int(aoff[k + 1])] // 10 == src[k] // 10)) for k in planted) if alns_np is not None else 0
    report["planted_recall"] = round(found / float(len(planted)), 4)
    assert report["planted_recall"] > 0.97, report
    # the other front end on a sample: same bytes
    n_s = 400
    monkeypatch.setenv("MK_PREFILTER_PATH", "global")
    qs = api.Queries.from_codes(fr[:int(foff[n_s])], foff[:n_s + 1], p)
    t0 = time.time()
    (h2, ho2), (a2, ao2) = api.search(db, qs, p)
    report["t_search_global_path_sample_s"] = round(time.time() - t0, 2)
    monkeypatch.delenv("MK_PREFILTER_PATH")
    assert np.array_equal(np.asarray(ho2), np.asarray(hoff[:n_s + 1])) and np.array_equal(np.asarray(ao2), np.asarray(aoff[:n_s + 1]))
    assert api.format_hits_bulk(h2, 0, int(ho2[-1])) == api.format_hits_bulk(hits, 0, int(hoff[n_s]))
    assert api.format_alignments_bulk(a2, 0, int(ao2[-1])) == api.format_alignments_bulk(alns, 0, int(aoff[n_s]))
    report["front_ends_agree_on_sample"] = n_s
    report["sha256_pref"] = oracle.digest_arrays(hoff, api.format_hits_bulk(hits, 0, int(hoff[-1])), n_queries)
    report["sha256_aln"] = oracle.digest_arrays(aoff, api.format_alignments_bulk(alns, 0, int(aoff[-1])), n_queries)
    print("config-5 full scale:", json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config5_case_%d.json" % n_targets), "w") as f:
            json.dump(report, f, indent=1)
    db.close()


def test_heavy_queries_wide_kernel_equals_the_global_path(gpu_api, monkeypatch):
    """Fragments that gather millions of index hits (100 ... 2 000 residues against 8 M proteins, k = 7: 0.7 ... 15 M hits each): the wide per-query
    kernel cuts a target class of such a query into subsets and sub-classes -- the regime in which round 4 found (against the reference's own run,
    profiles/r04_wide_kernel.txt) that one subset of eleven was lost.  The sort-based global path, equal to the reference there, is the check."""
    import sys
    api = gpu_api
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config5_digest as c5
    n_q = 60
    res, off = api.synth_targets(8000000, seed=c5.TARGET_SEED)
    fr, foff, src = api.synth_fragments(n_q, res, off, seed=5, mutation_rate=0.1, min_len=100, max_len=2000, random_every=10)
    p = api.default_params()
    p.kmer_size = 7
    db = api.TargetDB.from_codes(res, off, p)
    del res
    out = {}
    for name, env in (("global", {"MK_PREFILTER_PATH": "global"}), ("wide", {"MK_PREFILTER_PATH": "wide"}), ("wide, room for every query", {"MK_PREFILTER_PATH": "wide", "MK_PREFILTER_WIDE_POOL_GB": "64"})):
        for k in ("MK_PREFILTER_PATH", "MK_PREFILTER_WIDE_POOL_GB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        q = api.Queries.from_codes(fr, foff, p)
        hits, hoff = api.prefilter(db, q, p)
        out[name] = [api.format_hits_bulk(hits, int(hoff[i]), int(hoff[i + 1])) for i in range(n_q)]
        if name != "global":
            st = api.kernel_stats()
            assert st["prefilter_query_wide"]["cells"] > 5e7
    for name in out:
        bad = [i for i in range(n_q) if out[name][i] != out["global"][i]]
        assert not bad, (name, bad)
    assert sum(len(x) for x in out["global"]) > 100000
    db.close()


def test_contigs_to_exon_sets_against_a_k7_database(gpu_api, tmp_path):
    """data/predictexons.sh:42-87 end to end at the k-mer size of a UniRef50-scale database: contigs -> six-frame fragments -> prefilter + align
    with k = 7 against 2.7 M proteins (1.0e9 residues, the native generator's database; -k 7 as IndexTable::computeKmerSize picks it from 3.35e9
    residues on) -> resultspercontig + collectoptimalset; every contig's exon sets against the reference's own chain run on this box's host cores
    (oracle/_ref/ref_harness orfs | pipeline -k 7 | exons: Orf.cpp, the prefilter / align code and collectoptimalset.cpp compiled in place)."""
    import subprocess
    import sys
    import time
    api = gpu_api
    if not os.path.exists(oracle.REF):
        pytest.skip("needs oracle/_ref/ref_harness")
    from metaeuk_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config5_digest as c5
    n_targets = int(os.environ.get("MK_TEST_K7_EXON_TARGETS", "2700000"))
    res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
    rs = np.random.RandomState(17)
    picked = rs.randint(0, n_targets, size=48)
    founders = [np.array(res[int(off[t]):int(off[t + 1])], dtype=np.uint8) for t in picked]
    contigs = ["".join("ACGT"[x] for x in c) for c in synth.make_contigs(48, founders, seed=3)]
    contigs.append("ACGT" * 600)                                    # a contig without a gene: an empty record
    p = api.default_params()
    p.kmer_size = 7
    l2 = ctypes.CDLL(None).sysconf(191)                             # the reference sizes its hit bins from the L2 of the host it runs on: this host's
    p.host_l2_bytes = l2 if l2 and l2 > 0 else 262144
    db = api.TargetDB.from_codes(res, off, p)
    assert db.kmer_size() == 7
    t0 = time.time()
    o = api.Orfs(contigs)
    q = o.queries(p)
    api.search(db, q, p)
    pred = api.Predictions(db, o, q)
    t_gpu = time.time() - t0
    got = [pred.lines(c) for c in range(len(contigs))]
    # the reference's chain over the same files
    d = str(tmp_path)
    c5.write_lines(os.path.join(d, "t.txt"), res, off)
    del res
    with open(os.path.join(d, "c.txt"), "w") as f:
        f.write("\n".join(contigs) + "\n")
    matdir = oracle.write_matrix_files(os.path.join(d, "mat"))
    subprocess.check_call([oracle.REF, "orfs", os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt")], stdout=subprocess.DEVNULL)
    n_orf = 0
    with open(os.path.join(d, "q.txt"), "w") as f:
        for line in open(os.path.join(d, "orfs.txt")):
            if not line.startswith(">"):
                f.write(line.rstrip("\n").rsplit("\t", 1)[1] + "\n")
                n_orf += 1
    assert n_orf == o.n
    os.makedirs(os.path.join(d, "out"))
    t0 = time.time()
    subprocess.check_call([oracle.REF, "pipeline", matdir, os.path.join(d, "t.txt"), os.path.join(d, "q.txt"), os.path.join(d, "out"), "-k", "7",
                           "--threads", str(int(api.lib().mk_host_threads()))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([oracle.REF, "exons", os.path.join(d, "t.txt"), os.path.join(d, "c.txt"), os.path.join(d, "orfs.txt"),
                           os.path.join(d, "out", "aln.txt"), os.path.join(d, "exons.txt")], stdout=subprocess.DEVNULL)
    t_ref = time.time() - t0
    exp, cur = {}, None
    for line in open(os.path.join(d, "exons.txt")):
        if line.startswith(">"):
            cur = int(line[1:])
            exp[cur] = ""
        else:
            exp[cur] += line
    bad = [c for c in range(len(contigs)) if got[c] != exp.get(c, "")]
    assert not bad, (bad[:5], got[bad[0]][:300], exp.get(bad[0], "")[:300])
    with_pred = sum(1 for c in range(len(contigs)) if exp.get(c))
    multi = sum(1 for c in range(len(contigs)) if exp.get(c, "").count("\n") > 1)
    assert with_pred >= 40 and multi >= 20 and got[-1] == ""
    print("k = 7 contigs -> exon sets: %d contigs, %d fragments, %d with predictions; GPU chain %.2f s, reference chain %.1f s" % (len(contigs), o.n, with_pred, t_gpu, t_ref))
    os.remove(os.path.join(d, "t.txt"))
    pred.close(); q.close(); o.close(); db.close()


def _db_digest(base, n):
    """digest of a result DB (entries in key order 0 .. n-1) as tests/oracle.py digests the harness's block files: per-entry line counts, then the bodies"""
    import hashlib
    data = np.fromfile(base, dtype=np.uint8)
    rows = np.loadtxt(base + ".index", dtype=np.int64, ndmin=2)
    rows = rows[np.argsort(rows[:, 0], kind="stable")]
    assert rows.shape[0] == n and np.array_equal(rows[:, 0], np.arange(n))
    counts, bodies = np.zeros(n, dtype="<u8"), []
    for k in range(n):
        b = data[rows[k, 1]:rows[k, 1] + rows[k, 2] - 1].tobytes()
        counts[k] = b.count(b"\n")
        bodies.append(b)
    h = hashlib.sha256()
    h.update(counts.tobytes())
    h.update(b"".join(bodies))
    return h.hexdigest(), int(counts.sum())


@pytest.mark.parametrize("n_targets", [200000, 60000000])
def test_config5_60M_proteins_pinned_to_the_reference_by_target_splits(gpu_api, tmp_path, n_targets):
    """BASELINE config 5's database at its stated size against the REFERENCE's own run: 60 000 000 proteins (2.25e10 residues) searched in
    TARGET_DB_SPLIT mode -- how the reference itself runs a database whose index does not fit its host (Prefiltering.cpp:273-377,352-362; the
    unsplit index needs ~270 GB).  tools/config5_digest.py --split N ran oracle/_ref/ref_harness pipeline --split N on a GPU box's host cores
    (tests/golden/config5_digest_60000000_split<N>.json: N residue-balanced ranges of 15 M proteins, k = 7 from the residues per range, the
    lists joined by the reference's own mergeTargetSplits); here `metaeuk-amd prefilter --split N --split-mode 0` + `align` over the same
    database written as an MMseqs2 DB (23 GB) and the same planted + long fragments: digests of the prefilter DB and of the alignment DB equal.
    The 200 000-protein case (three ranges, k = 6; its digest comes from the build container) keeps the path under test where the big one cannot run."""
    import glob
    import json
    import shutil
    import subprocess
    import sys
    import time
    from metaeuk_amd import build
    api = gpu_api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    golds = sorted(glob.glob(os.path.join(root, "tests", "golden", "config5_digest_%d_split*.json" % n_targets)))
    if os.environ.get("MK_TEST_CONFIG5_SPLIT_GOLDEN"):
        golds = [os.environ["MK_TEST_CONFIG5_SPLIT_GOLDEN"]]
    if not golds:
        pytest.skip("no reference digest for %d proteins in target-split mode (tools/config5_digest.py --split N)" % n_targets)
    gold = json.load(open(golds[-1]))
    free, total = api.device_memory()
    if total < 280e9 and n_targets >= 60000000:
        pytest.skip("needs a 288 GB device")
    # the sequence DB of the 60 M proteins is 23 GB: with the golden present and a device that can hold the database the case must RUN -- it looks for
    # room beyond pytest's temp directory, and FAILS (not skips) when no directory has it (the case had been skipped silently once)
    need = 1.3 * gold["target_residues"] + 4e9
    if shutil.disk_usage(str(tmp_path)).free < need:
        import tempfile
        for cand in (os.environ.get("MK_TEST_SCRATCH"), os.environ.get("TMPDIR"), "/tmp", "/var/tmp", os.path.join(root, "gpurun_out"), "/dev/shm"):
            if cand and os.path.isdir(cand) and os.access(cand, os.W_OK) and shutil.disk_usage(cand).free >= need:
                import pathlib
                tmp_path = pathlib.Path(tempfile.mkdtemp(prefix="mk_config5_", dir=cand))
                request_cleanup = tmp_path
                break
        else:
            pytest.fail("the reference digest for %d proteins exists and the device holds the database, but no scratch directory has %.0f GB free "
                        "(set MK_TEST_SCRATCH)" % (n_targets, need / 1e9))
    else:
        request_cleanup = None
    sys.path.insert(0, os.path.join(root, "tools"))
    import config5_digest as c5
    t0 = time.time()
    res, off = api.synth_targets(n_targets, seed=c5.TARGET_SEED)
    assert int(off[-1]) == gold["target_residues"]
    fr, foff, src = c5.make_fragments(api, res, off, gold["n_queries"], gold.get("n_long_queries", 0))
    nq = len(foff) - 1
    # round 6: the END-TO-END pin over the same database (tools/config5_digest.py --contigs: ref_harness orfs | pipeline --split N | exons on
    # seeded contigs whose genes derive from proteins of this database -> tests/golden/config5_e2e_<n>_split<N>.json)
    e2e_path = os.path.join(root, "tests", "golden", "config5_e2e_%d_split%d.json" % (n_targets, gold["target_splits"]))
    e2e = json.load(open(e2e_path)) if os.path.exists(e2e_path) else None
    contigs = c5.make_contigs(api, res, off, e2e["n_contigs"] - 1) if e2e else None
    api.synth_write_seqdb(str(tmp_path / "T"), res, off)
    api.synth_write_seqdb(str(tmp_path / "Q"), fr, foff)
    del res
    t_gen = time.time() - t0
    N = gold["target_splits"]
    run = lambda *a: subprocess.run([build.BIN] + [str(x) for x in a], stderr=subprocess.PIPE)
    t0 = time.time()
    r = run("prefilter", tmp_path / "Q", tmp_path / "T", tmp_path / "pref", "--split", N, "--split-mode", "0", "-s", "5.7", "--ref-l2-bytes", gold["host_l2_bytes"],
            "--threads", int(api.lib().mk_host_threads()))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert ("%d target splits" % N).encode() in r.stderr, r.stderr.decode()[-600:]
    t_pref = time.time() - t0
    t0 = time.time()
    r = run("align", tmp_path / "Q", tmp_path / "T", tmp_path / "pref", tmp_path / "aln", "--alignment-mode", "2", "-e", "100", "--min-aln-len", "11",
            "--threads", int(api.lib().mk_host_threads()))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    t_aln = time.time() - t0
    d_pref, n_hits = _db_digest(str(tmp_path / "pref"), nq)
    d_aln, n_aln = _db_digest(str(tmp_path / "aln"), nq)
    report = dict(n_targets=n_targets, target_residues=gold["target_residues"], target_splits=N, fragments=nq, pref_hits=n_hits, alignments=n_aln,
                  reference=dict(pref_hits=gold["reference"]["pref_hits"], passed=gold["reference"]["passed"], wall_s=gold["reference"]["wall_s"]),
                  t_generate_and_write_db_s=round(t_gen, 1), t_prefilter_command_s=round(t_pref, 1), t_align_command_s=round(t_aln, 1),
                  sha256_pref=d_pref, sha256_aln=d_aln, match=(d_pref == gold["sha256_pref"] and d_aln == gold["sha256_aln"]))
    print("config 5, target splits:", json.dumps(report))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "config5_split_case_%d.json" % n_targets), "w") as f:
            json.dump(report, f, indent=1)
    e2e_report = None
    if e2e:
        # `metaeuk-amd predictexons --split N --split-mode 0`: extractorfs on the device, every target range masked, indexed and searched in turn, the
        # joined lists aligned against the whole database, resultspercontig + collectoptimalset (data/predictexons.sh:42-87)
        assert len(contigs) == e2e["n_contigs"]
        api.write_seq_db(str(tmp_path / "C"), api.seq_db_image(contigs), dbtype=1)
        t0 = time.time()
        r = run("predictexons", tmp_path / "C", tmp_path / "T", tmp_path / "calls", tmp_path / "tmp", "--split", N, "--split-mode", "0", "-s", "5.7",
                "--ref-l2-bytes", e2e["host_l2_bytes"], "--threads", int(api.lib().mk_host_threads()))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        t_e2e = time.time() - t0
        frag = [ln for ln in r.stderr.decode().splitlines() if ln.startswith("predictexons:") and "fragments" in ln]
        n_frag = int(frag[-1].split("->")[1].split("fragments")[0]) if frag else -1
        data = open(str(tmp_path / "calls"), "rb").read()
        got = {}
        for line in open(str(tmp_path / "calls.index")):
            k, o, l = line.split("\t")
            got[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
        per = [got.get(c, "") for c in range(len(contigs))]
        e2e_report = dict(contigs=len(contigs), fragments=n_frag, t_predictexons_command_s=round(t_e2e, 1), contigs_with_predictions=sum(1 for x in per if x),
                          prediction_lines=sum(x.count("\n") for x in per), sha256_exon_sets=c5.exon_sets_digest(per),
                          reference=dict(fragments=e2e["fragments"], contigs_with_predictions=e2e["reference"]["contigs_with_predictions"],
                                         prediction_lines=e2e["reference"]["prediction_lines"], sha256_exon_sets=e2e["sha256_exon_sets"]),
                          match=(c5.exon_sets_digest(per) == e2e["sha256_exon_sets"]))
        print("config 5, contigs -> exon sets over target splits:", json.dumps(e2e_report))
        if os.path.isdir(out):
            with open(os.path.join(out, "config5_e2e_case_%d.json" % n_targets), "w") as f:
                json.dump(e2e_report, f, indent=1)
    for name in ("T", "Q"):
        for sfx in ("", ".index", ".dbtype"):
            os.remove(str(tmp_path / (name + sfx)))
    if request_cleanup is not None:
        shutil.rmtree(str(request_cleanup), ignore_errors=True)
    assert n_hits == gold["reference"]["pref_hits"], report
    assert d_pref == gold["sha256_pref"], report
    assert n_aln == gold["reference"]["passed"] and d_aln == gold["sha256_aln"], report
    if e2e_report is not None:
        assert e2e_report["fragments"] == e2e["fragments"], e2e_report
        assert e2e_report["contigs_with_predictions"] == e2e["reference"]["contigs_with_predictions"] and e2e_report["match"], e2e_report
        assert e2e_report["contigs_with_predictions"] >= 0.8 * (len(contigs) - 1), e2e_report
