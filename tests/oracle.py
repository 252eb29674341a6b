"""Test-side access to the parity oracle (oracle/): builds it on demand and wraps a few C entry
points with ctypes.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "_build", "liboracle.so")
CLI = os.path.join(ODIR, "_build", "mko_cli")
REF = os.path.join(ODIR, "_ref", "ref_harness")
REF_MATDIR = "/root/reference/lib/mmseqs/data"


def build():
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not (os.path.exists(LIB) and os.path.exists(CLI) and os.path.getmtime(LIB) >= newest and os.path.getmtime(CLI) >= newest):
        subprocess.check_call(["make", "-C", ODIR, "_build/liboracle.so", "_build/mko_cli"], stdout=subprocess.DEVNULL)
    return LIB


class SubMat(C.Structure):
    _fields_ = [("sub", (C.c_short * 21) * 21), ("prob", (C.c_double * 21) * 21), ("pback", C.c_double * 21),
                ("lambda_", C.c_double), ("name", C.c_char_p)]


class SwResult(C.Structure):
    _fields_ = [("score", C.c_int), ("q_end", C.c_int), ("t_end", C.c_int), ("q_start", C.c_int), ("t_start", C.c_int),
                ("word", C.c_int), ("rev_mismatch", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
    return _lib


def submat(which, bit_factor, bias):
    m = SubMat()
    lib().mko_submat_init(C.byref(m), C.c_int(which), C.c_float(bit_factor), C.c_float(bias))
    return m


def encode(s):
    out = np.zeros(max(1, len(s)), dtype=np.uint8)
    lib().mko_map_sequence(s.encode(), C.c_int(len(s)), out.ctypes.data_as(C.c_void_p))
    return out[:len(s)]


_ALN = None


def sw(q, t, lanes_byte=32, lanes_word=16, gap_open=11, gap_extend=1, with_start=True):
    """oracle SW for two amino-acid strings -> (score, q_end, t_end, q_start, t_start)"""
    global _ALN
    if _ALN is None:
        _ALN = submat(0, 2.0, 0.0)
    qc, tc = encode(q), encode(t)
    cb = np.zeros(max(1, len(q)), dtype=np.int8)
    bias = C.c_int()
    L = lib()
    L.mko_sw_query_init(C.byref(_ALN), qc.ctypes.data_as(C.c_void_p), C.c_int(len(q)), C.c_float(1.0), cb.ctypes.data_as(C.c_void_p), C.byref(bias))
    r = SwResult()
    L.mko_sw_forward(C.byref(_ALN), qc.ctypes.data_as(C.c_void_p), cb.ctypes.data_as(C.c_void_p), bias, C.c_int(len(q)),
                     tc.ctypes.data_as(C.c_void_p), C.c_int(len(t)), C.c_int(gap_open), C.c_int(gap_extend), C.c_int(lanes_byte), C.c_int(lanes_word), C.byref(r))
    if with_start and r.score > 0:
        L.mko_sw_reverse(C.byref(_ALN), qc.ctypes.data_as(C.c_void_p), cb.ctypes.data_as(C.c_void_p), bias, C.c_int(len(q)),
                         tc.ctypes.data_as(C.c_void_p), C.c_int(len(t)), C.c_int(gap_open), C.c_int(gap_extend), C.c_int(lanes_byte), C.c_int(lanes_word), C.byref(r))
    return (r.score, r.q_end, r.t_end, r.q_start, r.t_start, r.rev_mismatch)


class Profile(C.Structure):
    _fields_ = [("L", C.c_int), ("query", C.POINTER(C.c_uint8)), ("consensus", C.POINTER(C.c_uint8)), ("aln", C.POINTER(C.c_int8)),
                ("sorted_score", C.POINTER(C.c_short)), ("sorted_idx", C.POINTER(C.c_uint8))]


def profile_arrays(entry, n_cols):
    """Sequence::mapProfile by the oracle -> (sorted scores [n, 20] i8, residue numbers [n, 20] i8, alignment profile [n, 21] i8, query letters [n] u8)"""
    L = lib()
    L.mko_profile_map.restype = C.POINTER(Profile)
    p = L.mko_profile_map(C.c_char_p(entry), C.c_int(n_cols))
    pr = p.contents
    scores = np.ctypeslib.as_array(pr.sorted_score, shape=(n_cols, 20)).astype(np.int8)
    idx = np.ctypeslib.as_array(pr.sorted_idx, shape=(n_cols, 20)).astype(np.int8)
    aln = np.ctypeslib.as_array(pr.aln, shape=(21, n_cols)).T.copy()
    query = np.ctypeslib.as_array(pr.query, shape=(n_cols,)).copy()
    L.mko_profile_free(p)
    return scores, idx, aln, query


def run_pipeline(targets, queries, outdir, extra=()):
    """oracle CLI over sequence lists -> (pref dict, aln dict) keyed by query index"""
    build()
    os.makedirs(outdir, exist_ok=True)
    tf, qf = os.path.join(outdir, "targets.txt"), os.path.join(outdir, "queries.txt")
    open(tf, "w").write("\n".join(targets) + "\n")
    open(qf, "w").write("\n".join(queries) + "\n")
    subprocess.check_call([CLI, "pipeline", tf, qf, os.path.join(outdir, "oracle")] + list(extra), stdout=subprocess.DEVNULL)
    return read_blocks(os.path.join(outdir, "oracle", "pref.txt")), read_blocks(os.path.join(outdir, "oracle", "aln.txt"))


def run_ref_pipeline(targets, queries, outdir, extra=()):
    os.makedirs(outdir, exist_ok=True)
    tf, qf = os.path.join(outdir, "targets.txt"), os.path.join(outdir, "queries.txt")
    open(tf, "w").write("\n".join(targets) + "\n")
    open(qf, "w").write("\n".join(queries) + "\n")
    matdir = REF_MATDIR if os.path.isdir(REF_MATDIR) else write_matrix_files(os.path.join(outdir, "mat"))   # no reference tree on a GPU box
    subprocess.check_call([REF, "pipeline", matdir, tf, qf, os.path.join(outdir, "ref")] + list(extra),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return read_blocks(os.path.join(outdir, "ref", "pref.txt")), read_blocks(os.path.join(outdir, "ref", "aln.txt"))


def read_blocks(path):
    """'>key' separated blocks -> list of strings indexed by key order"""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    blocks, cur = [], None
    with op(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if cur is not None:
                    blocks.append("".join(cur))
                cur = []
            else:
                cur.append(line)
    if cur is not None:
        blocks.append("".join(cur))
    return blocks


def write_matrix_files(outdir):
    """Materialise blosum62.out / VTML80.out (MMseqs2 text format) from the repository's matrix DATA
    table so that the reference harness can run where /root/reference does not exist."""
    import re
    os.makedirs(outdir, exist_ok=True)
    inc = open(os.path.join(ROOT, "metaeuk_amd", "data", "matrices.inc")).read()
    for name, fname in (("BLOSUM62", "blosum62.out"), ("VTML80", "VTML80.out")):
        alpha = re.search(r'MK_%s_ALPHABET\[\] = "(\w+)"' % name, inc).group(1)
        lam = re.search(r"MK_%s_LAMBDA = ([^;]+);" % name, inc).group(1)
        bg = re.search(r"MK_%s_BACKGROUND\[\d+\] = \{([^}]*)\}" % name, inc).group(1).replace(",", " ").split()
        body = re.search(r"MK_%s_SCORES\[\d+\]\[\d+\] = \{(.*?)\n\};" % name, inc, re.S).group(1)
        rows = [r.replace(",", " ").split() for r in re.findall(r"\{([^}]*)\}", body)]
        with open(os.path.join(outdir, fname), "w") as f:
            f.write("# %s\n" % name)
            f.write("# Background (precomputed optional): %s\n" % " ".join(bg))
            f.write("# Lambda     (precomputed optional): %s\n" % lam)
            f.write("   " + " ".join(alpha) + "\n")
            for a, r in zip(alpha, rows):
                f.write(a + " " + " ".join(r) + "\n")
    return outdir


def digest_blocks_file(path, n_blocks=None):
    """Canonical SHA-256 of a '>key'-separated block file (pref.txt / aln.txt of the oracle or the reference harness):
    the per-block line counts as little-endian uint64 followed by the block bodies back to back.  The HIP side hashes
    its offsets and mk_format_hits / mk_format_alignments output the same way (digest_arrays)."""
    import hashlib
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        return hashlib.sha256(b"").hexdigest(), 0
    nl = np.flatnonzero(raw == 10)
    starts = np.concatenate(([0], nl[:-1] + 1)) if nl.size else np.zeros(1, dtype=np.int64)
    is_head = raw[starts] == ord(">")
    heads = np.flatnonzero(is_head)
    total = heads.size
    counts = np.diff(np.concatenate((heads, [starts.size]))) - 1
    if n_blocks is not None and n_blocks < total:
        end_line = heads[n_blocks]                     # first line of block n_blocks
        cut = starts[end_line]
        raw, counts = raw[:cut], counts[:n_blocks]
        starts, is_head, nl = starts[:end_line], is_head[:end_line], nl[:end_line]
    delta = np.zeros(raw.size + 1, dtype=np.int32)
    hs = starts[is_head]
    he = nl[is_head] + 1
    np.add.at(delta, hs, 1)
    np.add.at(delta, he, -1)
    keep = np.cumsum(delta[:-1]) == 0
    h = hashlib.sha256()
    h.update(counts.astype("<u8").tobytes())
    h.update(raw[keep].tobytes())
    return h.hexdigest(), int(counts.size)


def digest_arrays(offsets, body, n_blocks):
    """the same digest from per-query offsets (uint64[n+1]) and the formatted lines of the first n_blocks queries"""
    import hashlib
    off = np.asarray(offsets[:n_blocks + 1], dtype=np.uint64)
    h = hashlib.sha256()
    h.update(np.diff(off).astype("<u8").tobytes())
    h.update(body)
    return h.hexdigest()


def saturated_threshold_workload(seed=5):
    """Targets with more than --max-seqs near-copies of every query: the prefilter's score threshold saturates at 255 and the
    reference rescales by the query's self score (QueryMatcher.cpp:163-170, rescoreHits :525-544)."""
    import random
    rng = random.Random(seed)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rs = lambda n: "".join(rng.choice(aa) for _ in range(n))
    mut = lambda s, r: "".join(ch if rng.random() > r else rng.choice(aa) for ch in s)
    queries = [rs(130), rs(95), rs(210)]
    targets = []
    for q in queries:
        for k in range(130):
            targets.append(rs(rng.randrange(0, 30)) + mut(q, 0.02 + 0.002 * k) + rs(rng.randrange(0, 30)))
    targets += [rs(rng.randrange(60, 300)) for _ in range(60)]
    rng.shuffle(targets)
    queries += [rs(60), mut(queries[0], 0.3)]
    return targets, queries


def long_sequence_workload(seed=9):
    """Sequences of 32768 residues and more: 16-bit index positions and diagonals wrap, the prefilter scores every real diagonal
    a wrapped one can stand for (UngappedAlignment::computeLongScore, UngappedAlignment.cpp:312-329)."""
    import random
    rng = random.Random(seed)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    rs = lambda n: "".join(rng.choice(aa) for _ in range(n))
    mut = lambda s, r: "".join(ch if rng.random() > r else rng.choice(aa) for ch in s)
    qa, qb, qc = rs(120), rs(200), rs(90)
    fam = [rs(rng.randrange(150, 400)) for _ in range(6)]
    targets = [rs(rng.randrange(100, 500)) for _ in range(120)]
    targets += [mut(f, 0.1) for f in fam for _ in range(3)]
    targets.append(rs(34000) + mut(qa, 0.08) + rs(5500))              # a homolog of qa behind position 32768 (titin-sized target)
    targets.append(rs(300) + mut(qb, 0.1) + rs(33000) + mut(qc, 0.05) + rs(40))
    targets.append(mut(qa, 0.15) + rs(66000) + mut(qa, 0.05))          # ... and beyond 65536: the index position itself wraps
    rng.shuffle(targets)
    long_q = rs(20000) + mut(fam[0], 0.1) + rs(13000) + mut(fam[1], 0.12) + rs(1500)      # 35 k-residue query, homologs on both sides of 32768
    queries = [qa, qb, qc, long_q, rs(80), mut(fam[2], 0.2)[:140]]
    return targets, queries
