"""GPU tests of the target side built ON THE DEVICE (metaeuk_amd/csrc/mk_index.hip): tantan masking and the k-mer index of
IndexBuilder::fillDatabase (M/src/prefiltering/IndexBuilder.cpp:55-239) as kernels, held word for word to the host builder
(mk::build_index, itself pinned to the reference's index dump in tests/test_oracle_golden.py), and the k = 7 index DB in both
directions against the reference's own PrefilteringIndexReader (oracle/_ref/ref_harness)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read()


def _both(api, seqs, params):
    """the same database through the host builder and through the device builder"""
    os.environ["MK_INDEX_BUILD"] = "host"
    try:
        a = api.TargetDB(seqs, params)
    finally:
        del os.environ["MK_INDEX_BUILD"]
    b = api.TargetDB(seqs, params)
    return a, b


def _assert_same(a, b, what):
    assert a.index_entries() == b.index_entries(), what
    assert a.masked_residues() == b.masked_residues(), what
    assert a.longest_list() == b.longest_list(), what
    assert a.index_compare(b) == (0, 0, 0, 0), what
    assert np.array_equal(a.masked(), b.masked()), what


def _edge_targets():
    from metaeuk_amd import synth
    rs = np.random.RandomState(5)
    t, _ = synth.make_targets(900, seed=3)
    seqs = [synth.codes_to_str(x) for x in t]
    seqs += ["", "M", "MKVLAAGIVG", "MKVLAAGIVGL", "ACDEFGHIKLMNPQRSTVWY" * 3, "A" * 300, "AQ" * 200, "PEPTIDE" * 40, "MSTNPKPQRKTKRNTNRRPQDVKFPGG" * 12,
             "acdefghiklmnpqrstvwyACDXXXBZJUOACDEFGHIKLM", "XXXXXXXXXXXXXXXXXXXXXXXXXXXX", "MKV" + "X" * 40 + "LAAGIVGLLL" * 3]
    # a k-mer in more targets than a lane sorts (64), than a workgroup sorts in LDS (4096), and a sequence with its own k-mers repeated
    motif = synth.codes_to_str(synth._rand_protein(rs, 60))
    for n_copies, tag in ((70, "W"), (300, "Y")):
        seqs += [synth.codes_to_str(synth._rand_protein(rs, 20 + (k % 7))) + motif[:30 if tag == "W" else 60] + tag * (k % 5) for k in range(n_copies)]
    unit = synth.codes_to_str(synth._rand_protein(rs, 37))
    seqs += [unit * 30, unit[:20] * 11 + unit]
    # long sequences: the quadratic first-position test in tiles, 16-bit positions that wrap
    seqs += [synth.codes_to_str(synth._rand_protein(rs, n)) for n in (1024, 1033, 1290, 5000, 40000, 70001)]
    return seqs


def test_device_index_equals_the_host_builder(gpu_api):
    api = gpu_api
    seqs = _edge_targets()
    for name, mod in (("default", {}), ("sse41 partial sums", {"simd_lanes_double": 2}), ("no masking", {"mask": 0}),
                      ("profile search: blosum background, unfiltered index", {"profile_search": 1}), ("-s 7.5", {"sensitivity": 7.5}),
                      ("mask-prob 0.5", {"mask_prob": 0.5})):
        p = api.default_params()
        for k, v in mod.items():
            setattr(p, k, v)
        a, b = _both(api, seqs, p)
        _assert_same(a, b, name)
        if name == "default":
            assert a.masked_residues() > 500 and a.index_entries() > 300000 and a.longest_list() >= 300
        a.close(); b.close()


def test_device_index_in_several_launches(gpu_api, monkeypatch):
    """count / fill launches are capped (the HIP runtime refuses 2^32 threads per launch: 16.7 M sequences of 256 lanes): with a cap of 97
    workgroups the 1 200 edge sequences take a dozen launches each and give the same tables"""
    api = gpu_api
    seqs = _edge_targets()
    p = api.default_params()
    a = api.TargetDB(seqs, p)
    monkeypatch.setenv("MK_TEST_INDEX_GRID_MAX", "97")
    b = api.TargetDB(seqs, p)
    _assert_same(a, b, "grid cap 97")
    p.kmer_size = 7
    c = api.TargetDB(seqs, p)
    monkeypatch.delenv("MK_TEST_INDEX_GRID_MAX")
    d = api.TargetDB(seqs, p)
    _assert_same(c, d, "grid cap 97, k = 7")
    for x in (a, b, c, d):
        x.close()


def test_corrupt_k7_index_db_is_an_error_not_a_crash(gpu_api, tmp_path):
    """an index DB is not trusted: an entry that names a sequence beyond the database, or k-mer list offsets that fall, are reported
    (mk_targetdb_open_index) instead of becoming out-of-bounds reads of the prefilter kernels"""
    import shutil
    import struct
    api = gpu_api
    targets = _text("e2e_targets.txt.gz").splitlines()[:60]
    keys = list(range(len(targets)))
    p = api.default_params()
    p.kmer_size = 7
    api.index_write(str(tmp_path / "good.idx"), api.seq_db_image(targets, keys), p)
    db = api.TargetDB.from_index(str(tmp_path / "good.idx"), p)
    assert db.kmer_size() == 7 and db.index_entries() > 1000
    db.close()
    where = {int(l.split("\t")[0]): (int(l.split("\t")[1]), int(l.split("\t")[2])) for l in open(tmp_path / "good.idx.index")}
    for name, key, patch in (("entry", 9, lambda blob: struct.pack("<I", 4000000000) + blob[4:]),            # ENTRIES: first record's sequence number
                             ("offsets", 10, lambda blob: blob[:8 * 1000] + struct.pack("<Q", 1 << 50) + blob[8 * 1001:])):   # ENTRIESOFFSETS: one huge offset
        for ext in ("", ".index", ".dbtype"):
            shutil.copy(str(tmp_path / "good.idx") + ext, str(tmp_path / (name + ".idx")) + ext)
        off, length = where[key]
        with open(tmp_path / (name + ".idx"), "r+b") as f:
            f.seek(off)
            blob = f.read(length)
            f.seek(off)
            f.write(patch(blob))
        with pytest.raises(Exception) as e:
            api.TargetDB.from_index(str(tmp_path / (name + ".idx")), p)
        assert "corrupt index DB" in str(e.value) or "do not add up" in str(e.value), str(e.value)


def test_device_index_long_lists_and_the_e2e_fixture(gpu_api):
    """10 000 near-copies of one protein: k-mer lists of 10 000 targets (the bitonic network in HBM, strides beyond one LDS chunk), and
    the e2e fixture's 200 proteins"""
    from metaeuk_amd import synth
    api = gpu_api
    rs = np.random.RandomState(9)
    base = synth._rand_protein(rs, 120)
    seqs = []
    for k in range(10000):
        m = base.copy()
        hit = rs.random_sample(len(m)) < 0.02
        m[hit] = synth._rand_protein(rs, int(hit.sum()))
        seqs.append(synth.codes_to_str(m))
    p = api.default_params()
    a, b = _both(api, seqs, p)
    assert a.longest_list() > 4096
    _assert_same(a, b, "near-copies")
    a.close(); b.close()
    a, b = _both(api, _text("e2e_targets.txt.gz").splitlines(), p)
    _assert_same(a, b, "e2e fixture")
    a.close(); b.close()


def test_device_index_k7_and_shifted_entry_base(gpu_api):
    api = gpu_api
    seqs = _edge_targets()[:400] + _edge_targets()[-10:]
    p = api.default_params()
    p.kmer_size = 7
    a, b = _both(api, seqs, p)
    assert b.kmer_size() == 7
    _assert_same(a, b, "k = 7")
    a.close(); b.close()
    os.environ["MK_TEST_ENTRY_BASE"] = "6000000000"
    try:
        a, b = _both(api, seqs, api.default_params())
        _assert_same(a, b, "list starts shifted beyond 2^32")
    finally:
        del os.environ["MK_TEST_ENTRY_BASE"]
    a.close(); b.close()


def test_headline_database_masked_and_indexed_on_the_device(gpu_api):
    """BASELINE config 2's 100 000 proteins (37.7 M residues): device == host, and the device builder's time"""
    import time
    from metaeuk_amd import synth
    api = gpu_api
    t, _ = synth.make_targets(100000, seed=11)
    res = np.concatenate(t)
    off = np.zeros(len(t) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in t])
    p = api.default_params()
    os.environ["MK_INDEX_BUILD"] = "host"
    try:
        t0 = time.time()
        a = api.TargetDB.from_codes(res, off, p)
        t_host = time.time() - t0
    finally:
        del os.environ["MK_INDEX_BUILD"]
    api.kernel_stats(reset=True)
    t0 = time.time()
    b = api.TargetDB.from_codes(res, off, p)
    t_dev = time.time() - t0
    st = api.kernel_stats()
    print("target side of config 2: host builder %.2f s, device builder %.2f s; kernels: %s" %
          (t_host, t_dev, {k: round(v["ms"], 1) for k, v in st.items() if k.startswith("index_") or k.startswith("host_index")}))
    _assert_same(a, b, "config 2")
    assert a.index_entries() > 30000000


@pytest.mark.skipif(not os.path.exists(oracle.REF), reason="reference harness not on this box")
def test_k7_index_db_both_directions_against_the_reference(gpu_api, tmp_path):
    """a k = 7 index DB (what `createindex` writes for a database of 3.35e9 residues or more, IndexTable.h:439-449; forced with -k 7):
    ours read by the reference's PrefilteringIndexReader, the reference's read by ours, and both equal to the database built directly"""
    api = gpu_api
    mat = oracle.write_matrix_files(str(tmp_path / "mat"))
    targets = _text("e2e_targets.txt.gz").splitlines()
    frags = [l.rsplit("\t", 1)[1] for l in _text("e2e_process_orfs.txt.gz").splitlines()][:1500]
    keys = [3 * i + 1 for i in range(len(targets))]
    image = api.seq_db_image(targets, keys)
    api.write_seq_db(str(tmp_path / "T"), image)
    p = api.default_params()
    p.kmer_size = 7
    p.host_l2_bytes = 2097152
    api.index_write(str(tmp_path / "own.idx"), image, p)
    subprocess.check_call([oracle.REF, "createindex", mat, str(tmp_path / "T"), "-s", "5.7", "-k", "7"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = lambda path: {int(l.split("\t")[0]): int(l.split("\t")[2]) for l in open(path)}
    r_ref, r_own = rows(tmp_path / "T.idx.index"), rows(tmp_path / "own.idx.index")
    assert {k: v for k, v in r_ref.items() if k not in (2, 22)} == {k: v for k, v in r_own.items() if k not in (2, 22)}
    # byte-identical k-mer lists, offsets and masked sequences in the two files
    def entry(path, key):
        for l in open(path + ".index"):
            k, o, n = l.split("\t")
            if int(k) == key:
                with open(path, "rb") as f:
                    f.seek(int(o))
                    return f.read(int(n))
        raise KeyError(key)
    for key in (9, 10, 12, 13, 14, 15, 16, 1):
        assert entry(str(tmp_path / "T.idx"), key) == entry(str(tmp_path / "own.idx"), key), key
    # our reader: both files give the tables of the database built directly (device tables compared in HBM)
    direct = api.TargetDB(targets, p)
    for idx in ("own.idx", "T.idx"):
        db = api.TargetDB.from_index(str(tmp_path / idx), p)
        assert db.kmer_size() == 7 and list(db.keys) == keys
        assert direct.index_compare(db) == (0, 0, 0, 0), idx
        q = api.Queries(frags, p)
        (hits, hoff), (alns, aoff) = api.search(db, q, p)
        if idx == "own.idx":
            first = (api.format_hits_bulk(hits, 0, int(hoff[-1]), db.keys), api.format_alignments_bulk(alns, 0, int(aoff[-1])))
        else:
            assert first == (api.format_hits_bulk(hits, 0, int(hoff[-1]), db.keys), api.format_alignments_bulk(alns, 0, int(aoff[-1])))
        db.close()
    # the reference's reader on OUR file: same hits as our search (target keys as the reference prints them)
    (tmp_path / "q.txt").write_text("\n".join(frags) + "\n")
    (tmp_path / "t_unused.txt").write_text("A\n")
    subprocess.check_call([oracle.REF, "pipeline", mat, str(tmp_path / "t_unused.txt"), str(tmp_path / "q.txt"), str(tmp_path / "pipe"), "-s", "5.7", "--threads", "8",
                           "--no-align", "--index", str(tmp_path / "own.idx")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref_pref = "".join(oracle.read_blocks(str(tmp_path / "pipe" / "pref.txt")))
    assert ref_pref == first[0].decode() and ref_pref.count("\n") > 200
