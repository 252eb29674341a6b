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
