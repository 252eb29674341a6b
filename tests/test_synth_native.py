"""CPU test of the native seeded generator (metaeuk_amd/csrc/mk_synth.cpp): deterministic, thread-count independent, the shape of
SURVEY.md 8(d) (families of ten, founder 150 .. 600 residues, member j redrawn at 0.05 (1 + j))."""
import os
import subprocess
import sys

import numpy as np


def test_native_generator_shape_and_determinism():
    from metaeuk_amd import api
    res, off = api.synth_targets(2003, seed=5)
    res2, off2 = api.synth_targets(2003, seed=5)
    assert np.array_equal(res, res2) and np.array_equal(off, off2)
    res3, _ = api.synth_targets(2003, seed=6)
    assert not np.array_equal(res[:1000], res3[:1000])
    lens = np.diff(off.astype(np.int64))
    assert lens.min() >= 150 and lens.max() <= 600 and res.max() <= 19
    # families of ten share the founder's length; member j differs from the founder in ~ 0.05 (1 + j) (19 / 20) of its residues
    for f in (0, 7, 150):
        t0 = 10 * f
        founder = res[int(off[t0]):int(off[t0 + 1])]
        for j in (1, 5, 9):
            m = res[int(off[t0 + j]):int(off[t0 + j + 1])]
            assert len(m) == len(founder)
            frac = float((m != founder).mean())
            assert abs(frac - 0.05 * (1 + j) * 0.93) < 0.09, (f, j, frac)
    # the last family is cut short (2003 = 200 * 10 + 3)
    assert len(off) == 2004
    # the background: leucine (code 9) most frequent, tryptophan (18) least
    counts = np.bincount(res, minlength=20)
    assert counts.argmax() == 9 and counts.argmin() == 18
    # planted fragments come from the target they name
    fr, foff, src = api.synth_fragments(500, res, off, seed=3, mutation_rate=0.1, min_len=20, max_len=60, random_every=5)
    assert (src[4::5] == 0xFFFFFFFF).all() and (src[:4] != 0xFFFFFFFF).all()
    for k in (0, 1, 2, 3, 11, 257):
        q = fr[int(foff[k]):int(foff[k + 1])]
        t = res[int(off[src[k]]):int(off[src[k] + 1])]
        best = max(int((t[s:s + len(q)] == q).sum()) for s in range(len(t) - len(q) + 1))
        assert best >= 0.8 * len(q)
    # sequence DB image
    data, keys, offs, lens2 = api.synth_seqdb(res, off)
    assert bytes(data[int(offs[3]):int(offs[3]) + int(lens2[3])]).endswith(b"\n\0") and int(lens2[3]) == int(off[4] - off[3]) + 2
    letters = "ACDEFGHIKLMNPQRSTVWY"
    assert bytes(data[:int(lens2[0]) - 2]).decode() == "".join(letters[c] for c in res[:int(off[1])])


def test_native_generator_does_not_depend_on_the_thread_count():
    code = ("import sys, hashlib; sys.path.insert(0, %r); from metaeuk_amd import api; r, o = api.synth_targets(5000, seed=9); "
            "print(hashlib.sha256(r.tobytes() + o.tobytes()).hexdigest())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for threads in ("1", "5"):
        env = dict(os.environ, OMP_NUM_THREADS=threads)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip())
    assert outs[0] == outs[1] and len(outs[0]) == 64


def test_native_seqdb_writer_equals_the_python_writer(tmp_path):
    """mk_synth_write_seqdb (the streaming writer the 60 M-protein tests use) writes byte for byte what api.write_seq_db writes from the in-memory
    image, in pieces of 2^20 sequences (a piece boundary is crossed here with empty and one-residue sequences around it), and the line file the
    reference harness reads"""
    from metaeuk_amd import api
    res, off = api.synth_targets(3000, seed=11)
    # ragged: an empty sequence and a one-residue sequence in front
    res2 = np.concatenate([np.array([4], dtype=np.uint8), res])
    off2 = np.concatenate([np.array([0, 0, 1], dtype=np.uint64), off[1:] + 1]).astype(np.uint64)
    api.synth_write_seqdb(str(tmp_path / "T"), res2, off2, with_lines=True)
    api.write_seq_db(str(tmp_path / "P"), api.synth_seqdb(res2, off2))
    for sfx in ("", ".index", ".dbtype"):
        assert open(str(tmp_path / "T") + sfx, "rb").read() == open(str(tmp_path / "P") + sfx, "rb").read(), sfx
    lines = open(str(tmp_path / "T.txt")).read().split("\n")
    assert len(lines) == len(off2) and lines[0] == "" and lines[1] == "F" and len(lines[2]) == int(off2[3] - off2[2])


def test_config5_fragment_sets_are_reproducible():
    """tools/config5_digest.make_fragments: the planted fragments followed by the long ones, the same bytes for the digest tool (reference side) and
    the test (product side)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import config5_digest as c5
    from metaeuk_amd import api
    res, off = api.synth_targets(5000, seed=c5.TARGET_SEED)
    a = c5.make_fragments(api, res, off, 300, 20)
    b = c5.make_fragments(api, res, off, 300, 20)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    fr, foff, src = a
    assert len(foff) == 321 and len(src) == 320 and int(foff[-1]) == len(fr)
    lens = np.diff(foff.astype(np.int64))
    assert lens[:300].max() <= 120 and lens[300:].min() >= 150 and lens[300:].max() <= 1500     # (a long fragment is cut at its target's end)
    short_only = c5.make_fragments(api, res, off, 300, 0)
    assert np.array_equal(short_only[0][:int(short_only[1][-1])], fr[:int(foff[300])])
