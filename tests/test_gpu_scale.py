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
    # similar k-mers by front end: per-query kernels by region size (2 K / 4 K one wave per query, 8 K / 64 K several waves), global path
    k_small = sum(v["cells"] for k, v in stats.items() if k in ("prefilter_query_cap2048", "prefilter_query_cap4096", "prefilter_query_cap8192"))
    k_big = stats.get("prefilter_query_cap65536", {"cells": 0})["cells"]
    k_glob = stats.get("kmer_probe_gather", {"cells": 0})["cells"]
    print("similar k-mers: queries <= 8 K hits %.3g, <= 64 K hits %.3g, global path %.3g" % (k_small, k_big, k_glob))
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
