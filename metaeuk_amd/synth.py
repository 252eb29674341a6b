"""Deterministic synthetic workload generator.

Targets: proteins in families of 10 (founder + 9 diverged members).
Contigs: 5 kb nucleotide contigs carrying one multi-exon gene derived from a
founder, odd contigs reverse-complemented.  Queries for the prefilter+align hot
path are the 6-frame stop-to-stop ORF fragments (>= 15 codons) of the contigs,
i.e. what `extractorfs --orf-start-mode 1 --min-length 15` + `translatenucs`
hand to `search` in predictexons.sh:42-68 of the reference.

Everything is numpy/MT19937 based so the same seed gives the same bytes on the
build container and on the GPU box.
"""
import numpy as np

AA = "ACDEFGHIKLMNPQRSTVWY"
# Robinson & Robinson background frequencies (order of AA above)
_BG = np.array([0.07805, 0.01925, 0.05364, 0.06295, 0.03856, 0.07377, 0.02199,
                0.05142, 0.05744, 0.09019, 0.02243, 0.04487, 0.05203, 0.04264,
                0.05129, 0.07120, 0.05841, 0.06441, 0.01330, 0.03216])
_BG = _BG / _BG.sum()

# standard genetic code, index = 16*b0 + 4*b1 + b2 with A,C,G,T = 0,1,2,3
_BASES = "ACGT"
_CODON_AA = {}
_std = ("KNKNTTTTRSRSIIMI" "QHQHPPPPRRRRLLLL" "EDEDAAAAGGGGVVVV" "*Y*YSSSS*CWCLFLF")
_CODON_TABLE = np.frombuffer(_std.encode(), dtype=np.uint8).copy()
# a fixed back-translation codon for each amino acid (first codon found)
_BACK = {}
for _i, _a in enumerate(_std):
    if _a != "*" and _a not in _BACK:
        _BACK[_a] = _i
_BACK_ARR = np.zeros((20, 3), dtype=np.uint8)
for _k, _a in enumerate(AA):
    _c = _BACK[_a]
    _BACK_ARR[_k] = (_c // 16, (_c // 4) % 4, _c % 4)


def _rand_protein(rs, n):
    return np.searchsorted(np.cumsum(_BG), rs.random_sample(n)).clip(0, 19).astype(np.uint8)


def _mutate(rs, prot, rate):
    out = prot.copy()
    mask = rs.random_sample(len(prot)) < rate
    k = int(mask.sum())
    if k:
        out[mask] = _rand_protein(rs, k)
    return out


def make_targets(n_targets, seed=11):
    """Return (list of uint8 code arrays in AA order 0..19, list of founders)."""
    rs = np.random.RandomState(seed)
    targets, founders = [], []
    fam = 0
    while len(targets) < n_targets:
        L = int(rs.randint(150, 601))
        founder = _rand_protein(rs, L)
        founders.append(founder)
        for j in range(10):
            if len(targets) >= n_targets:
                break
            targets.append(founder if j == 0 else _mutate(rs, founder, 0.05 * (1 + j)))
        fam += 1
    return targets, founders


def make_contigs(n_contigs, founders, seed=11, contig_len=5000):
    """Return list of uint8 arrays of base codes (A,C,G,T = 0..3)."""
    rs = np.random.RandomState(seed + 1000003)
    contigs = []
    for c in range(n_contigs):
        founder = founders[int(rs.randint(0, len(founders)))]
        gene = _mutate(rs, founder, 0.25)[:500]
        n_ex = int(rs.randint(2, 5))
        cuts = np.sort(rs.randint(1, max(2, len(gene) - 1), size=n_ex - 1))
        pieces = np.split(gene, cuts)
        parts = [rs.randint(0, 4, size=200).astype(np.uint8)]
        for e, ex in enumerate(pieces):
            if e > 0:
                intron_len = int(rs.randint(60, 401))
                parts.append(np.array([2, 3], dtype=np.uint8))            # GT
                parts.append(rs.randint(0, 4, size=intron_len).astype(np.uint8))
                parts.append(np.array([0, 2], dtype=np.uint8))            # AG
            if len(ex):
                parts.append(_BACK_ARR[ex].reshape(-1))
        seq = np.concatenate(parts)
        if len(seq) < contig_len:
            seq = np.concatenate([seq, rs.randint(0, 4, size=contig_len - len(seq)).astype(np.uint8)])
        seq = seq[:contig_len]
        if c % 2 == 1:
            seq = (3 - seq)[::-1].copy()
        contigs.append(seq)
    return contigs


_AA_INDEX = np.full(256, 255, dtype=np.uint8)
for _k, _a in enumerate(AA):
    _AA_INDEX[ord(_a)] = _k


def six_frame_orfs(contig, min_len=15):
    """Stop-to-stop ORF fragments (as uint8 AA codes 0..19) in all 6 frames."""
    out = []
    for strand in (0, 1):
        s = contig if strand == 0 else (3 - contig)[::-1]
        for frame in range(3):
            n = (len(s) - frame) // 3
            if n <= 0:
                continue
            cod = s[frame:frame + 3 * n].reshape(n, 3).astype(np.int32)
            aa = _CODON_TABLE[cod[:, 0] * 16 + cod[:, 1] * 4 + cod[:, 2]]
            stops = np.flatnonzero(aa == ord("*"))
            bounds = np.concatenate([[-1], stops, [n]])
            for b in range(len(bounds) - 1):
                lo, hi = bounds[b] + 1, bounds[b + 1]
                if hi - lo >= min_len:
                    out.append(_AA_INDEX[aa[lo:hi]])
    return out


def make_queries(n_contigs, founders, seed=11):
    q = []
    for contig in make_contigs(n_contigs, founders, seed):
        q.extend(six_frame_orfs(contig))
    return q


def codes_to_str(codes):
    return "".join(AA[c] if c < 20 else "X" for c in codes)


def make_workload(n_contigs, n_targets, seed=11):
    """Convenience: (targets, queries) as lists of strings."""
    t, founders = make_targets(n_targets, seed)
    q = make_queries(n_contigs, founders, seed)
    return [codes_to_str(x) for x in t], [codes_to_str(x) for x in q]


if __name__ == "__main__":
    import sys
    nc, nt = int(sys.argv[1]), int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
    t, q = make_workload(nc, nt, seed)
    with open(sys.argv[4] if len(sys.argv) > 4 else "targets.txt", "w") as f:
        f.write("\n".join(t) + "\n")
    with open(sys.argv[5] if len(sys.argv) > 5 else "queries.txt", "w") as f:
        f.write("\n".join(q) + "\n")
    print(len(t), len(q), sum(map(len, q)) / max(1, len(q)))


def make_profiles(proteins, seed=11):
    """Synthetic profile DB entries (25 bytes per column + a trailing NUL, Sequence.h:458-471) for a list of uint8 code arrays: every
    column scores its own residue with the top value of a template, the other residues get the remaining values in a fixed per-residue
    similarity order, plus Gaussian noise (sigma 5).  The template is the mean sorted column of result2profile's output on the e2e fixture
    (tests/golden/prof_db.bin.gz) with ranks 2-5 raised by 2: ~160 similar k-mers per start at -s 4, like the real profiles' 171.  For timing the profile path."""
    rs = np.random.RandomState(seed + 77)
    template = np.array([21.6, 10.5, 5.9, 2.8, 0.4, -3.5, -5.0, -6.0, -6.9, -7.7, -8.4, -9.1, -9.6, -10.2, -10.8, -11.3, -11.9, -12.4, -13.1, -14.1])
    rank_of = np.zeros((20, 20), dtype=np.int64)            # rank_of[c][a] = position of residue a in c's similarity order (c itself first)
    for c in range(20):
        others = [a for a in rs.permutation(20) if a != c]
        rank_of[c, c] = 0
        for r, a in enumerate(others):
            rank_of[c, a] = r + 1
    out = []
    for p in proteins:
        c = np.asarray(p, dtype=np.int64)
        col = np.rint(template[rank_of[c]] + rs.normal(0.0, 5.0, size=(len(c), 20)))
        cols = np.zeros((len(c), 25), dtype=np.uint8)
        cols[:, :20] = np.clip(col, -128, 127).astype(np.int8).view(np.uint8)
        cols[:, 20] = c
        cols[:, 21] = c
        cols[:, 22] = 10
        out.append(cols.tobytes() + b"\0")
    return out
