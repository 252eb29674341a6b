"""Query sharding across the GPUs of a node and the host-side merge of the per-GPU result DBs.

The hot path shards embarrassingly over queries (reference: QUERY_DB_SPLIT, Prefiltering.cpp:751-756, and the
MPI mode of `align`, Alignment.cpp:219-242); there is no collective on the data path.  Each rank takes the
contiguous range given by the reference's residue-balanced rule and writes its own result shard; rank 0 merges
the shards the way DBWriter::mergeResults does (DBWriter.cpp:531-623: concatenate the data files, rebase the
index offsets, sort the index by key).
"""
import math
import os


def decompose_by_residues(entry_lengths, world_rank, world_size):
    """DBReader::decomposeDomainByAminoAcid (DBReader.cpp:1216-1257).  entry_lengths = index lengths
    (sequence length + 2) in DB order.  Returns (start_entry, num_entries)."""
    n = len(entry_lengths)
    data_size = sum(entry_lengths)
    if world_size > data_size:
        raise ValueError("World Size: %d dbSize: %d" % (world_size, data_size))
    if world_size == 1:
        return 0, n
    if n <= world_size:
        return (world_rank, 1) if world_rank < n else (0, 0)
    chunk = math.ceil(data_size / world_size)
    per_worker = [0] * world_size
    rank, acc = 0, 0
    for length in entry_lengths:
        if acc >= chunk:
            acc = 0
            rank += 1
        acc += length
        per_worker[rank] += 1
    return sum(per_worker[:world_rank]), per_worker[world_rank]


def write_result_db(base, items, dbtype):
    """items: iterable of (key, text).  One data file + index + dbtype (MMseqs2 format)."""
    off = 0
    with open(base, "wb") as d, open(base + ".index", "w") as idx:
        for key, text in items:
            b = text.encode() + b"\0"
            d.write(b)
            idx.write("%d\t%d\t%d\n" % (key, off, len(b)))
            off += len(b)
    with open(base + ".dbtype", "wb") as t:
        t.write(int(dbtype).to_bytes(4, "little"))


def merge_result_dbs(out_base, shard_bases, dbtype):
    """DBWriter::mergeResults: concatenate shard data files, rebase offsets, sort the index by key."""
    index = []
    offset = 0
    with open(out_base, "wb") as out:
        for base in shard_bases:
            data = open(base, "rb").read()
            out.write(data)
            for line in open(base + ".index"):
                k, o, l = line.split("\t")
                index.append((int(k), int(o) + offset, int(l)))
            offset += len(data)
    index.sort(key=lambda e: e[0])
    with open(out_base + ".index", "w") as f:
        for k, o, l in index:
            f.write("%d\t%d\t%d\n" % (k, o, l))
    with open(out_base + ".dbtype", "wb") as t:
        t.write(int(dbtype).to_bytes(4, "little"))
    return len(index)


def read_result_db(base):
    data = open(base, "rb").read()
    out = {}
    for line in open(base + ".index"):
        k, o, l = line.split("\t")
        out[int(k)] = data[int(o):int(o) + int(l) - 1].decode()
    return out


def db_hash(base):
    """layout-independent DB hash: sha256 over key-sorted (key, entry bytes)"""
    import hashlib
    h = hashlib.sha256()
    for k, v in sorted(read_result_db(base).items()):
        h.update(("%d\t" % k).encode() + v.encode())
    return h.hexdigest()
