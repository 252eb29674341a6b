"""ctypes binding of the C ABI in include/metaeuk_amd.h (tests / bench plumbing only).

The product is the shared library; nothing in this file computes anything.  If the library is
missing it is built; if it cannot be loaded the import fails loudly (no fallback).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None


class Params(C.Structure):
    _fields_ = [("sensitivity", C.c_float), ("kmer_score", C.c_int), ("max_seqs", C.c_int),
                ("min_ungapped_score", C.c_int), ("comp_bias_corr", C.c_int), ("comp_bias_scale", C.c_float),
                ("mask", C.c_int), ("mask_prob", C.c_float), ("gap_open", C.c_int), ("gap_extend", C.c_int),
                ("evalue_thr", C.c_double), ("min_aln_len", C.c_int), ("simd_lanes_byte", C.c_int),
                ("simd_lanes_word", C.c_int), ("simd_lanes_double", C.c_int), ("host_l2_bytes", C.c_uint64),
                ("profile_search", C.c_int), ("kmer_size", C.c_int)]


class Hit(C.Structure):
    _fields_ = [("seq_id", C.c_uint32), ("pref_score", C.c_int32), ("diagonal", C.c_uint16), ("pad_", C.c_uint16)]


class Alignment(C.Structure):
    _fields_ = [("db_key", C.c_uint32), ("bit_score", C.c_int32), ("seq_id", C.c_float), ("qcov", C.c_float),
                ("dbcov", C.c_float), ("evalue", C.c_double), ("q_start", C.c_int32), ("q_end", C.c_int32),
                ("q_len", C.c_int32), ("db_start", C.c_int32), ("db_end", C.c_int32), ("db_len", C.c_int32),
                ("aln_len", C.c_int32), ("raw_score", C.c_int32)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_uint64), ("alg_bytes", C.c_double),
                ("cells", C.c_double)]


HIT_DTYPE = np.dtype([("seq_id", "<u4"), ("pref_score", "<i4"), ("diagonal", "<u2"), ("pad_", "<u2")])

EXPORTS = ["mk_init", "mk_last_error", "mk_default_params", "mk_device_name", "mk_encode", "mk_targetdb_create",
           "mk_targetdb_destroy", "mk_targetdb_residues", "mk_targetdb_index_entries", "mk_targetdb_masked",
           "mk_queries_create", "mk_queries_destroy", "mk_queries_derived", "mk_prefilter", "mk_prefilter_result", "mk_prefilter_result_set",
           "mk_align", "mk_align_result", "mk_search", "mk_search_begin", "mk_search_wait", "mk_shutdown", "mk_extract_orfs", "mk_orfs_result", "mk_queries_from_orfs",
           "mk_orfs_destroy", "mk_format_orf_header", "mk_sw_pairs", "mk_ungapped",
           "mk_kernel_stats", "mk_kernel_stats_reset", "mk_format_hit", "mk_format_alignment", "mk_format_hits", "mk_format_alignments", "mk_targetdb_set_keys",
           "mk_profiles_create", "mk_profiles_derived", "mk_swap_alignments", "mk_swapped_result", "mk_swapped_destroy",
           "mk_targetdb_masked_residues", "mk_targetdb_kmer_size", "mk_targetdb_longest_list", "mk_targetdb_index_compare",
           "mk_synth_targets", "mk_synth_fragments", "mk_synth_seqdb", "mk_synth_write_seqdb", "mk_targetdb_create_sequences", "mk_prefilter_statistics",
           "mk_format_prefilter_statistics", "mk_device_memory"]


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("METAEUK_AMD_LIB") or _build.LIB   # the override serves kernel-variant experiments
        if not os.path.exists(path):
            _build.build()
        L = C.CDLL(path)
        L.mk_last_error.restype = C.c_char_p
        L.mk_targetdb_residues.restype = C.c_uint64
        L.mk_targetdb_index_entries.restype = C.c_uint64
        L.mk_targetdb_masked_residues.restype = C.c_uint64
        L.mk_targetdb_longest_list.restype = C.c_uint32
        L.mk_format_hit.restype = C.c_size_t
        L.mk_format_alignment.restype = C.c_size_t
        L.mk_format_hits.restype = C.c_size_t
        L.mk_format_alignments.restype = C.c_size_t
        L.mk_format_orf_header.restype = C.c_size_t
        L.mk_format_prediction_exon.restype = C.c_size_t
        _LIB = L
    return _LIB


class MkError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise MkError("metaeuk_amd error %d: %s" % (rc, lib().mk_last_error().decode()))


def init(device=0):
    _chk(lib().mk_init(int(device)))


def default_params():
    p = Params()
    lib().mk_default_params(C.byref(p))
    return p


def device_name():
    buf = C.create_string_buffer(256)
    _chk(lib().mk_device_name(buf, 256))
    return buf.value.decode()


def device_memory():
    """(free, total) bytes of the device's HBM"""
    f, t = C.c_uint64(), C.c_uint64()
    _chk(lib().mk_device_memory(C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def encode(seqs):
    """list of str -> (uint8 residues, uint64 offsets[n+1])"""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    joined = "".join(seqs).encode()
    out = np.zeros(max(1, len(joined)), dtype=np.uint8)
    lib().mk_encode(joined, C.c_size_t(len(joined)), out.ctypes.data_as(C.c_void_p))
    return out, off


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class TargetDB:
    def __init__(self, seqs, params=None, _codes=None):
        self.params = params or default_params()
        self.res, self.off = _codes if _codes is not None else encode(seqs)
        self.n = len(self.off) - 1
        self.h = C.c_void_p()
        _chk(lib().mk_targetdb_create(_p(self.res), _p(self.off), C.c_uint32(self.n), C.byref(self.params), C.byref(self.h)))

    @classmethod
    def from_index(cls, index_db, params=None):
        """target side from a createindex DB (type 9) written by this library or by the reference"""
        self = cls.__new__(cls)
        self.params = params or default_params()
        self.h = C.c_void_p()
        _chk(lib().mk_targetdb_open_index(C.c_char_p(index_db.encode()), C.byref(self.params), C.byref(self.h)))
        kp, n = C.c_void_p(), C.c_uint32()
        _chk(lib().mk_targetdb_keys(self.h, C.byref(kp), C.byref(n)))
        self.n = int(n.value)
        self.keys = np.array(np.ctypeslib.as_array(C.cast(kp, C.POINTER(C.c_uint32)), shape=(self.n,))) if kp.value else None
        return self

    @classmethod
    def from_codes(cls, res, off, params=None):
        """res: uint8 codes 0..20 in matrix alphabet order (ACDEFGHIKLMNPQRSTVWYX), off: uint64[n+1]"""
        return cls(None, params, _codes=(np.ascontiguousarray(res, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint64)))

    def masked(self):
        out = np.zeros(max(1, int(self.off[-1])), dtype=np.uint8)
        _chk(lib().mk_targetdb_masked(self.h, _p(out)))
        return out

    def index_entries(self):
        return int(lib().mk_targetdb_index_entries(self.h))

    def masked_residues(self):
        return int(lib().mk_targetdb_masked_residues(self.h))

    def kmer_size(self):
        return int(lib().mk_targetdb_kmer_size(self.h))

    def longest_list(self):
        return int(lib().mk_targetdb_longest_list(self.h))

    def index_compare(self, other):
        """test hook: (differing slot words, presence words, entries, masked residues) between the device tables of two databases"""
        d = (C.c_uint64 * 4)()
        _chk(lib().mk_targetdb_index_compare(self.h, other.h, d))
        return tuple(int(x) for x in d)

    def close(self):
        if self.h:
            lib().mk_targetdb_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synth_targets(n_targets, seed=11):
    """native seeded generator (mk_synth.cpp): protein families of ten -> (uint8 codes, uint64 offsets[n + 1])"""
    off = np.zeros(n_targets + 1, dtype=np.uint64)
    total = C.c_uint64()
    _chk(lib().mk_synth_targets(C.c_uint64(n_targets), C.c_uint64(seed), None, C.c_uint64(0), _p(off), C.byref(total)))
    res = np.empty(max(1, int(total.value)), dtype=np.uint8)
    _chk(lib().mk_synth_targets(C.c_uint64(n_targets), C.c_uint64(seed), _p(res), C.c_uint64(res.size), _p(off), C.byref(total)))
    return res, off


def synth_fragments(n_fragments, t_res, t_off, seed=11, mutation_rate=0.15, min_len=20, max_len=80, random_every=0):
    """planted homologs: fragments cut out of the targets and mutated -> (uint8 codes, uint64 offsets, uint32 source target)"""
    n_t = len(t_off) - 1
    off = np.zeros(n_fragments + 1, dtype=np.uint64)
    src = np.zeros(n_fragments, dtype=np.uint32)
    total = C.c_uint64()
    args = (C.c_uint64(n_fragments), C.c_uint64(seed), _p(t_res), _p(t_off), C.c_uint64(n_t), C.c_double(mutation_rate), C.c_uint32(min_len),
            C.c_uint32(max_len), C.c_uint64(random_every))
    _chk(lib().mk_synth_fragments(*args, None, C.c_uint64(0), _p(off), _p(src), C.byref(total)))
    res = np.empty(max(1, int(total.value)), dtype=np.uint8)
    _chk(lib().mk_synth_fragments(*args, _p(res), C.c_uint64(res.size), _p(off), _p(src), C.byref(total)))
    return res, off, src


def synth_seqdb(res, off):
    """residue codes -> the image of an MMseqs2 sequence DB (data bytes as a numpy array, keys, offsets, lengths) for index_write / write_seq_db"""
    n = len(off) - 1
    data = np.empty(int(off[-1]) + 2 * n, dtype=np.uint8)
    keys = np.zeros(n, dtype=np.uint32); offs = np.zeros(n, dtype=np.uint64); lens = np.zeros(n, dtype=np.uint32)
    _chk(lib().mk_synth_seqdb(_p(res), _p(off), C.c_uint64(n), _p(data), _p(keys), _p(offs), _p(lens)))
    return data, keys, offs, lens


def synth_write_seqdb(base, res, off, with_lines=False):
    """residue codes -> an MMseqs2 sequence DB on disk (<base>, .index, .dbtype; key = position), written natively piece by piece; with_lines: <base>.txt too"""
    _chk(lib().mk_synth_write_seqdb(C.c_char_p(base.encode()), _p(res), _p(off), C.c_uint64(len(off) - 1), C.c_int(1 if with_lines else 0)))


def seq_db_image(seqs, keys=None, order=None):
    """an MMseqs2 sequence DB in memory: data blob of 'SEQ\\n\\0' entries laid out in `order`, and the .index rows (key, offset, length)
    sorted by key, the way createdb leaves them"""
    n = len(seqs)
    keys = list(range(n)) if keys is None else list(keys)
    order = list(range(n)) if order is None else list(order)
    off, pos, chunks = [0] * n, 0, []
    for i in order:
        b = seqs[i].encode("latin-1") + b"\n\0"
        off[i] = pos
        pos += len(b)
        chunks.append(b)
    rows = sorted(range(n), key=lambda i: keys[i])
    return (b"".join(chunks), np.array([keys[i] for i in rows], dtype=np.uint32), np.array([off[i] for i in rows], dtype=np.uint64),
            np.array([len(seqs[i]) + 2 for i in rows], dtype=np.uint32))


def write_seq_db(base, image, dbtype=0):
    data, keys, offs, lens = image
    with open(base, "wb") as f:
        f.write(data if not isinstance(data, np.ndarray) else data.tobytes())
    with open(base + ".index", "w") as f:
        for k, o, l in zip(keys, offs, lens):
            f.write("%d\t%d\t%d\n" % (k, o, l))
    with open(base + ".dbtype", "wb") as f:
        f.write(int(dbtype).to_bytes(4, "little"))


def index_write(index_db, image, params=None, dbtype=0):
    """createindex: <index_db>, .index, .dbtype in the reference's format (no GPU needed)"""
    data, keys, offs, lens = image
    p = params or default_params()
    dptr = _p(data) if isinstance(data, np.ndarray) else C.c_char_p(data)
    _chk(lib().mk_index_write(C.c_char_p(index_db.encode()), dptr, C.c_uint64(len(data)), _p(keys), _p(offs), _p(lens), C.c_uint32(len(keys)),
                              C.c_int(dbtype), C.byref(p)))


def index_dump(index_db, out_dir):
    _chk(lib().mk_index_dump(C.c_char_p(index_db.encode()), C.c_char_p(out_dir.encode())))


class Queries:
    def __init__(self, seqs, params=None, _codes=None):
        self.params = params or default_params()
        self.res, self.off = _codes if _codes is not None else encode(seqs)
        self.n = len(self.off) - 1
        self.h = C.c_void_p()
        _chk(lib().mk_queries_create(_p(self.res), _p(self.off), C.c_uint32(self.n), C.byref(self.params), C.byref(self.h)))

    @classmethod
    def from_codes(cls, res, off, params=None):
        return cls(None, params, _codes=(np.ascontiguousarray(res, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint64)))

    def derived(self):
        """(kmer_thr i16, diag_corr i8, sw_bias8 i8) as derived on the device -- test hook."""
        total = int(self.off[-1])
        kt = np.zeros(total, dtype=np.int16); dc = np.zeros(total, dtype=np.int8); sb = np.zeros(total, dtype=np.int8)
        _chk(lib().mk_queries_derived(self.h, _p(kt), _p(dc), _p(sb)))
        return kt, dc, sb

    def close(self):
        if self.h:
            lib().mk_queries_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Profiles(Queries):
    """a batch of profile queries (the reference's profile-target search makes the profiles the queries): `entries` = the profile DB's
    entries as bytes, 25 bytes per column (a trailing NUL per entry is dropped).  Search it against a TargetDB whose params have
    profile_search = 1."""

    def __init__(self, entries, params=None, _columns=None):
        self.params = params or default_params()
        self.columns, self.off = _columns if _columns is not None else self.pack(entries)
        self.n = len(self.off) - 1
        self.h = C.c_void_p()
        _chk(lib().mk_profiles_create(_p(self.columns), _p(self.off), C.c_uint32(self.n), C.byref(self.params), C.byref(self.h)))

    @staticmethod
    def pack(entries):
        """(columns uint8 [25 * total], col_offsets uint64 [n + 1]) of a list of profile DB entries: what mk_profiles_create takes"""
        cols = [e[:len(e) - (len(e) % 25)] for e in entries]
        off = np.zeros(len(cols) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(c) // 25 for c in cols], dtype=np.uint64)
        return (np.frombuffer(b"".join(cols), dtype=np.uint8).copy() if cols else np.zeros(0, dtype=np.uint8)), off

    @classmethod
    def from_columns(cls, columns, off, params=None):
        return cls(None, params, _columns=(np.ascontiguousarray(columns, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint64)))

    def derived(self):
        """(query letters u8 [cols], sorted scores + residue numbers i8 [cols, 40], alignment profile i8 [cols, 32], k-mer threshold i16 [cols])"""
        total = int(self.off[-1])
        le = np.zeros(total, dtype=np.uint8); so = np.zeros((total, 40), dtype=np.int8); al = np.zeros((total, 32), dtype=np.int8)
        kt = np.zeros(total, dtype=np.int16)
        _chk(lib().mk_profiles_derived(self.h, _p(le), _p(so), _p(al), _p(kt)))
        return le, so, al, kt


class _SwappedHandle:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                lib().mk_swapped_destroy(self.h)
                self.h = None
        except Exception:
            pass


def swap_alignments(alns, aln_off, n_targets, swapped_db_residues, query_keys=None, params=None):
    """swapresults on the arrays of align_result(): -> (ctypes array of Alignment, offsets uint64[n_targets + 1]) per target index"""
    p = params or default_params()
    aln_off = np.ascontiguousarray(aln_off, dtype=np.uint64)
    keys = None if query_keys is None else np.ascontiguousarray(query_keys, dtype=np.uint32)
    h = C.c_void_p()
    _chk(lib().mk_swap_alignments(alns, _p(aln_off), C.c_uint32(len(aln_off) - 1), None if keys is None else _p(keys), C.c_uint32(n_targets),
                                  C.c_uint64(swapped_db_residues), C.byref(p), C.byref(h)))
    owner = _SwappedHandle(h)
    ap, op = C.c_void_p(), C.c_void_p()
    _chk(lib().mk_swapped_result(h, C.byref(ap), C.byref(op)))
    off = np.array(np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(n_targets + 1,)))
    total = int(off[-1])
    if total == 0 or not ap.value:
        return (Alignment * 0)(), off
    out = (Alignment * total).from_address(ap.value)          # a view into the handle's memory: it lives as long as the array does
    out._owner = owner
    return out, off


def prefilter(db, q, params=None):
    """-> (hits structured array [total], offsets uint64[n+1]); views into memory owned by the batch handle"""
    p = params or db.params
    _chk(lib().mk_prefilter(db.h, q.h, C.byref(p)))
    return prefilter_result(q)


def prefilter_result(q):
    hp, op = C.c_void_p(), C.c_void_p()
    _chk(lib().mk_prefilter_result(q.h, C.byref(hp), C.byref(op)))
    off = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(q.n + 1,))
    total = int(off[-1])
    if total == 0:
        return np.zeros(0, dtype=HIT_DTYPE), off
    raw = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(total * HIT_DTYPE.itemsize,))
    return raw.view(HIT_DTYPE), off


ORF_DTYPE = np.dtype([("contig", "<u4"), ("from", "<u4"), ("to", "<u4"), ("incomplete_start", "u1"), ("incomplete_end", "u1"),
                      ("minus_strand", "u1"), ("pad_", "u1")])


class Orfs:
    """six-frame ORF fragments of a list of contigs (nucleotide strings), translated on the GPU"""

    def __init__(self, contigs, min_codons=15):
        raw = "".join(contigs).encode("latin-1")
        off = np.zeros(len(contigs) + 1, dtype=np.uint64)
        np.cumsum([len(c) for c in contigs], out=off[1:])
        self.h = C.c_void_p()
        self.n_contigs = len(contigs)
        _chk(lib().mk_extract_orfs(C.c_char_p(raw), _p(off), C.c_uint32(len(contigs)), C.c_int(min_codons), C.byref(self.h)))
        op, fp, ap, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        _chk(lib().mk_orfs_result(self.h, C.byref(op), C.byref(fp), C.byref(ap), C.byref(n)))
        self.n = int(n.value)
        if self.n:
            self.orfs = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint8)), shape=(self.n * ORF_DTYPE.itemsize,)).view(ORF_DTYPE)
            self.aa_off = np.ctypeslib.as_array(C.cast(fp, C.POINTER(C.c_uint64)), shape=(self.n + 1,))
            self.aa = C.string_at(ap, int(self.aa_off[-1]))
        else:
            self.orfs, self.aa_off, self.aa = np.zeros(0, dtype=ORF_DTYPE), np.zeros(1, dtype=np.uint64), b""

    def protein(self, k):
        return self.aa[int(self.aa_off[k]):int(self.aa_off[k + 1])].decode("latin-1")

    def header(self, k):
        buf = C.create_string_buffer(96)
        n = lib().mk_format_orf_header(buf, self.orfs[k:k + 1].ctypes.data_as(C.c_void_p))
        return buf.raw[:n].decode()

    def queries(self, params=None):
        """the fragments as a query batch (device-resident hand-over)"""
        q = Queries.__new__(Queries)
        q.params = params or default_params()
        q.h = C.c_void_p()
        _chk(lib().mk_queries_from_orfs(self.h, C.byref(q.params), C.byref(q.h)))
        q.n = self.n
        q.res, q.off = None, np.array(self.aa_off, dtype=np.uint64)
        return q

    def close(self):
        if self.h:
            lib().mk_orfs_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ExonParams(C.Structure):
    _fields_ = [("evalue_thr", C.c_double), ("target_cov_thr", C.c_double), ("max_intron", C.c_uint64), ("min_intron", C.c_uint64),
                ("min_exon_aa", C.c_uint64), ("max_aa_overlap", C.c_uint64), ("max_exon_sets", C.c_uint64), ("gap_open", C.c_int32),
                ("gap_extend", C.c_int32)]


PREDICTION_DTYPE = np.dtype([("target", "<u4"), ("strand", "<i4"), ("total_bit_score", "<u4"), ("n_exons", "<u4"), ("evalue", "<f8"),
                             ("low_coord", "<u4"), ("high_coord", "<u4"), ("first_exon", "<u8")])
EXON_DTYPE = np.dtype([("orf", "<u4"), ("bit_score", "<i4"), ("seq_id", "<f8"), ("evalue", "<f8"), ("target_start", "<i4"), ("target_end", "<i4"),
                       ("target_len", "<i4"), ("contig_start", "<i4"), ("contig_end", "<i4"), ("nucleotide_len", "<i4"), ("orf_from", "<i4"),
                       ("orf_to", "<i4")])


def default_exon_params():
    p = ExonParams()
    lib().mk_default_exon_params(C.byref(p))
    return p


class Predictions:
    """exon sets per contig, target and strand (resultspercontig + collectoptimalset) from an aligned ORF batch"""

    def __init__(self, db, orfs, q, params=None, target_keys=None):
        self.params = params or default_exon_params()
        self.h = C.c_void_p()
        keys = None if target_keys is None else np.ascontiguousarray(target_keys, dtype=np.uint32)
        _chk(lib().mk_predict_exons(db.h, orfs.h, q.h, C.byref(self.params), None if keys is None else _p(keys), C.byref(self.h)))
        self._views(orfs.n_contigs)

    @classmethod
    def from_arrays(cls, orfs, n_contigs, alns, aln_off, db_residues, params=None):
        """host-only form: orfs = ORF_DTYPE array, alns = ctypes array of Alignment, aln_off = uint64[n_orfs + 1]"""
        self = cls.__new__(cls)
        self.params = params or default_exon_params()
        self.h = C.c_void_p()
        orfs = np.ascontiguousarray(orfs, dtype=ORF_DTYPE)
        aln_off = np.ascontiguousarray(aln_off, dtype=np.uint64)
        _chk(lib().mk_predict_exons_arrays(_p(orfs), C.c_uint64(len(orfs)), C.c_uint32(n_contigs), alns, _p(aln_off), C.c_uint64(db_residues),
                                           C.byref(self.params), None, C.byref(self.h)))
        self._views(n_contigs)
        return self

    def _views(self, n_contigs):
        pp, op, ep, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        _chk(lib().mk_predictions_result(self.h, C.byref(pp), C.byref(op), C.byref(ep), C.byref(n)))
        self.n = int(n.value)
        self.contig_off = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(n_contigs + 1,))
        if self.n:
            self.predictions = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint8)), shape=(self.n * PREDICTION_DTYPE.itemsize,)).view(PREDICTION_DTYPE)
            ne = int(self.predictions["first_exon"][-1] + self.predictions["n_exons"][-1])
            self.exons = np.ctypeslib.as_array(C.cast(ep, C.POINTER(C.c_uint8)), shape=(ne * EXON_DTYPE.itemsize,)).view(EXON_DTYPE)
        else:
            self.predictions, self.exons = np.zeros(0, dtype=PREDICTION_DTYPE), np.zeros(0, dtype=EXON_DTYPE)

    def lines(self, contig):
        """the contig's record of the reference's prediction DB (one line per exon)"""
        buf = C.create_string_buffer(512)
        out = []
        for k in range(int(self.contig_off[contig]), int(self.contig_off[contig + 1])):
            p = self.predictions[k:k + 1]
            for j in range(int(p["first_exon"][0]), int(p["first_exon"][0] + p["n_exons"][0])):
                n = lib().mk_format_prediction_exon(buf, p.ctypes.data_as(C.c_void_p), self.exons[j:j + 1].ctypes.data_as(C.c_void_p))
                out.append(buf.raw[:n].decode())
        return "".join(out)

    def close(self):
        if self.h:
            lib().mk_predictions_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def search(db, q, params=None):
    """prefilter + align in one pipelined pass -> ((hits, hit_offsets), (alignments, aln_offsets))"""
    p = params or db.params
    _chk(lib().mk_search(db.h, q.h, C.byref(p)))
    return prefilter_result(q), align_result(q)


def search_begin(db, q, params=None):
    """queues the batch for the pipelined pass and returns at once (mk_search_begin); search_wait(q) collects it.  `db`, `q` and the
    parameters must stay alive until then (the library copies the parameters)."""
    p = params or db.params
    _chk(lib().mk_search_begin(db.h, q.h, C.byref(p)))


def search_wait(q):
    """-> ((hits, hit_offsets), (alignments, aln_offsets)) of a batch begun with search_begin"""
    _chk(lib().mk_search_wait(q.h))
    return prefilter_result(q), align_result(q)


def shutdown():
    """waits for the searches in flight, stops and joins the library's search threads (mk_shutdown); the next search starts them again"""
    lib().mk_shutdown.restype = None
    lib().mk_shutdown()


def align(db, q, params=None):
    """-> (ctypes array of Alignment [total], offsets uint64[n+1])"""
    p = params or db.params
    _chk(lib().mk_align(db.h, q.h, C.byref(p)))
    return align_result(q)


def align_result(q):
    ap, op = C.c_void_p(), C.c_void_p()
    _chk(lib().mk_align_result(q.h, C.byref(ap), C.byref(op)))
    off = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint64)), shape=(q.n + 1,))
    total = int(off[-1])
    if total == 0 or not ap.value:
        return (Alignment * 0)(), off
    alns = C.cast(ap, C.POINTER(Alignment * total)).contents
    return alns, off


def sw_pairs(db, q, q_idx, t_idx, with_start=True, params=None):
    p = params or db.params
    q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32)
    t_idx = np.ascontiguousarray(t_idx, dtype=np.uint32)
    out = np.zeros((len(q_idx), 5), dtype=np.int32)
    _chk(lib().mk_sw_pairs(db.h, q.h, C.byref(p), _p(q_idx), _p(t_idx), C.c_uint64(len(q_idx)), C.c_int(1 if with_start else 0), _p(out)))
    return out


def ungapped(db, q, q_idx, t_idx, diag):
    q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32)
    t_idx = np.ascontiguousarray(t_idx, dtype=np.uint32)
    diag = np.ascontiguousarray(diag, dtype=np.uint16)
    out = np.zeros(len(q_idx), dtype=np.int32)
    _chk(lib().mk_ungapped(db.h, q.h, _p(q_idx), _p(t_idx), _p(diag), C.c_uint64(len(q_idx)), _p(out)))
    return out


def kernel_stats(reset=False):
    arr = (KernelStat * 160)()
    n = lib().mk_kernel_stats(arr, 160)
    res = {arr[i].name.decode(): dict(ms=arr[i].ms, launches=int(arr[i].launches), alg_bytes=arr[i].alg_bytes, cells=arr[i].cells)
           for i in range(n)}
    if reset:
        lib().mk_kernel_stats_reset()
    return res


def format_hits_bulk(hits, lo, hi, target_keys=None):
    """bytes of the prefilter lines of hits[lo:hi] (key = target index, or target_keys[index]), formatted by the library in one call"""
    n = hi - lo
    if n <= 0:
        return b""
    buf = np.empty(32 * n, dtype=np.uint8)
    tk = None if target_keys is None else np.ascontiguousarray(target_keys, dtype=np.uint32)
    w = lib().mk_format_hits(_p(buf), C.c_size_t(buf.size), C.c_void_p(hits.ctypes.data + lo * HIT_DTYPE.itemsize), C.c_uint64(n), None if tk is None else _p(tk))
    return buf[:w].tobytes()


def format_alignments_bulk(alns, lo, hi):
    n = hi - lo
    if n <= 0:
        return b""
    buf = np.empty(160 * n, dtype=np.uint8)
    w = lib().mk_format_alignments(_p(buf), C.c_size_t(buf.size), C.c_void_p(C.addressof(alns) + lo * C.sizeof(Alignment)), C.c_uint64(n))
    return buf[:w].tobytes()


def format_hits(hits, lo, hi, key_of=None):
    buf = C.create_string_buffer(64)
    out = []
    for h in hits[lo:hi]:
        key = int(h["seq_id"]) if key_of is None else key_of(int(h["seq_id"]))
        n = lib().mk_format_hit(buf, C.c_uint32(key), C.c_int32(int(h["pref_score"])), C.c_uint16(int(h["diagonal"])))
        out.append(buf.raw[:n].decode())
    return "".join(out)


def format_alignments(alns, lo, hi):
    buf = C.create_string_buffer(256)
    out = []
    for i in range(lo, hi):
        n = lib().mk_format_alignment(buf, C.byref(alns[i]))
        out.append(buf.raw[:n].decode())
    return "".join(out)
