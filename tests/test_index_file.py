"""createindex's precomputed index DB: the product's writer and reader against the reference's own
PrefilteringIndexReader / DBWriter (oracle/_ref/ref_harness createindex, pipeline --index).  Host code: no GPU."""
import os
import random
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.skipif(not os.path.exists(oracle.REF), reason="reference harness not built here")


def _workload():
    from metaeuk_amd import synth
    targets, queries = synth.make_workload(3, 180, seed=4)
    tstr = list(targets)
    tstr += ["ACDEFGHIKLMNPQRSTVWY" * 3, "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", "MKV", "acdefghiklmnpqrstvwyACDXXXBZJUOACDEFGHIKLM", "MSTNPKPQRKTKRNTNRRPQDVKFPGG" * 4]
    qstr = list(queries)[:400]
    return tstr, qstr


def test_index_db_writer_and_reader_against_the_reference(tmp_path):
    from metaeuk_amd import api
    oracle.build()
    mat = oracle.write_matrix_files(str(tmp_path / "mat"))
    tstr, qstr = _workload()
    n = len(tstr)
    rs = random.Random(3)
    order = list(range(n))
    rs.shuffle(order)                                               # data order != key order, keys not 0..n-1
    keys = [5 * i + 2 for i in range(n)]
    image = api.seq_db_image(tstr, keys, order)
    api.write_seq_db(str(tmp_path / "T"), image)
    params = api.default_params()
    # the reference's createindex, and ours
    subprocess.check_call([oracle.REF, "createindex", mat, str(tmp_path / "T"), "-s", "5.7"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    api.index_write(str(tmp_path / "P.idx"), image, params)
    assert open(tmp_path / "P.idx.dbtype", "rb").read() == open(tmp_path / "T.idx.dbtype", "rb").read() == (9).to_bytes(4, "little")
    # same entry keys and lengths in the two index DBs, entry by entry (free text: the generator string, and the seed matrix as
    # its .out text -- ours is printed from the matrix table; the pipeline runs below parse it with the reference's own parser)
    rows = lambda p: {int(l.split("\t")[0]): int(l.split("\t")[2]) for l in open(p)}
    r_ref, r_own = rows(tmp_path / "T.idx.index"), rows(tmp_path / "P.idx.index")
    assert sorted(r_ref) == sorted(r_own)
    assert {k: v for k, v in r_ref.items() if k not in (2, 22)} == {k: v for k, v in r_own.items() if k not in (2, 22)}
    # OUR reader on both files: identical content
    for tag, idx in (("ref", "T.idx"), ("own", "P.idx")):
        os.makedirs(tmp_path / ("dump_" + tag))
        api.index_dump(str(tmp_path / idx), str(tmp_path / ("dump_" + tag)))
    for f in ("meta.txt", "seqs.txt", "masked_targets.txt", "index.txt"):
        a, b = open(tmp_path / "dump_ref" / f).read(), open(tmp_path / "dump_own" / f).read()
        assert a == b and len(a) > 50, f
    # THE REFERENCE'S reader on both files: same tables, same prefilter hits and alignments
    (tmp_path / "q.txt").write_text("\n".join(qstr) + "\n")
    (tmp_path / "t_unused.txt").write_text("A\n")
    out = {}
    for tag, idx in (("ref", "T.idx"), ("own", "P.idx")):
        d = str(tmp_path / ("pipe_" + tag))
        subprocess.check_call([oracle.REF, "pipeline", mat, str(tmp_path / "t_unused.txt"), str(tmp_path / "q.txt"), d, "-s", "5.7", "--threads", "2",
                               "--dump", "--index", str(tmp_path / idx)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[tag] = {f: open(os.path.join(d, f)).read() for f in ("pref.txt", "aln.txt", "index.txt", "masked_targets.txt", "stats.txt")}
    assert out["ref"] == out["own"]
    assert out["ref"]["pref.txt"].count("\n") > len(qstr) + 50 and out["ref"]["aln.txt"].count("\n") > len(qstr) + 20
    # and the two readers see the same lists and masked sequences
    assert out["ref"]["index.txt"] == open(tmp_path / "dump_own" / "index.txt").read()
    assert out["ref"]["masked_targets.txt"] == open(tmp_path / "dump_own" / "masked_targets.txt").read()
    # reader errors are errors
    with open(tmp_path / "P.idx.index") as f:
        rows_txt = f.read().replace("0\t0\t3\n", "0\t0\t3\n", 1)
    bad = tmp_path / "bad.idx"
    for ext in ("", ".dbtype"):
        os.link(tmp_path / ("P.idx" + ext), str(bad) + ext)
    (tmp_path / "bad.idx.index").write_text("\n".join(l for l in rows_txt.splitlines() if not l.startswith("9\t")) + "\n")   # no ENTRIES
    with pytest.raises(api.MkError):
        api.index_dump(str(bad), str(tmp_path / "dump_own"))


def test_cli_createindex_without_a_gpu(tmp_path):
    """`metaeuk-amd createindex` is host code: it runs here, and the reference's reader searches from its index DB exactly as from
    the index DB the reference builds itself"""
    from metaeuk_amd import api, build
    oracle.build()
    mat = oracle.write_matrix_files(str(tmp_path / "mat"))
    tstr, qstr = _workload()
    image = api.seq_db_image(tstr)
    for name in ("own", "ref"):
        os.makedirs(tmp_path / name)
        api.write_seq_db(str(tmp_path / name / "T"), image)
    subprocess.check_call([build.BIN, "createindex", str(tmp_path / "own" / "T"), str(tmp_path / "tmp"), "-s", "5.7", "--threads", "4"], stderr=subprocess.DEVNULL)
    subprocess.check_call([oracle.REF, "createindex", mat, str(tmp_path / "ref" / "T"), "-s", "5.7"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    (tmp_path / "q.txt").write_text("\n".join(qstr) + "\n")
    (tmp_path / "t_unused.txt").write_text("A\n")
    out = {}
    for name in ("own", "ref"):
        d = str(tmp_path / ("pipe_" + name))
        subprocess.check_call([oracle.REF, "pipeline", mat, str(tmp_path / "t_unused.txt"), str(tmp_path / "q.txt"), d, "-s", "5.7", "--threads", "2",
                               "--index", str(tmp_path / name / "T.idx")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out[name] = (open(os.path.join(d, "pref.txt")).read(), open(os.path.join(d, "aln.txt")).read())
    assert out["own"] == out["ref"] and out["own"][0].count("\n") > len(qstr) + 50
    # (k = 7 index DBs -- 10 GB of list offsets each -- are written and read back in tests/test_gpu_index.py, on the GPU box's disk)
