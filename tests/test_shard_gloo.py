"""CPU test of the N > 1 path: two gloo ranks shard the queries with the reference's residue-balanced rule,
each writes its result shard, rank 0 merges; the merged DB must equal the unsharded one.  The per-query
results come from the golden fixture (the GPU is not needed to test the sharding/merge logic)."""
import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _lines(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return f.read().split("\n")[:-1]


def _worker(rank, world, tmp, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle
    from metaeuk_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    queries = _lines("small_queries.txt.gz")
    aln = oracle.read_blocks(os.path.join(GOLD, "small_aln.txt.gz"))
    start, num = shard.decompose_by_residues([len(q) + 2 for q in queries], rank, world)
    shard.write_result_db(os.path.join(tmp, "res_%d" % rank), [(i, aln[i]) for i in range(start, start + num)], 5)
    dist.barrier()
    if rank == 0:
        n = shard.merge_result_dbs(os.path.join(tmp, "res"), [os.path.join(tmp, "res_%d" % r) for r in range(world)], 5)
        assert n == len(queries)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_query_sharding_and_merge(tmp_path):
    import torch.multiprocessing as mp
    from metaeuk_amd import shard
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, str(tmp_path), port), nprocs=2, join=True)
    queries = _lines("small_queries.txt.gz")
    aln = oracle.read_blocks(os.path.join(GOLD, "small_aln.txt.gz"))
    shard.write_result_db(str(tmp_path / "single"), [(i, aln[i]) for i in range(len(queries))], 5)
    assert shard.db_hash(str(tmp_path / "res")) == shard.db_hash(str(tmp_path / "single"))
    merged = shard.read_result_db(str(tmp_path / "res"))
    assert [merged[i] for i in range(len(queries))] == aln


def test_decompose_matches_reference_rule():
    from metaeuk_amd import shard
    lens = [10, 10, 10, 10, 10, 10, 10, 10]
    # chunk = ceil(80/3) = 27 -> ranks take entries until their sum reaches the chunk: 3,3,2
    assert [shard.decompose_by_residues(lens, r, 3) for r in range(3)] == [(0, 3), (3, 3), (6, 2)]
    assert shard.decompose_by_residues(lens, 0, 1) == (0, 8)
    assert [shard.decompose_by_residues([5, 5], r, 4) for r in range(4)] == [(0, 1), (1, 1), (0, 0), (0, 0)]
    cover = []
    big = [7 + (i * 37) % 400 for i in range(1000)]
    for r in range(8):
        s, n = shard.decompose_by_residues(big, r, 8)
        cover += list(range(s, s + n))
    assert cover == list(range(1000))


def test_cli_shard_rule_and_merge(tmp_path):
    """the CLI's worker ranges follow the same reference rule as shard.py, and `mergeshards` is DBWriter::mergeResults"""
    import random
    import subprocess
    from metaeuk_amd import build, shard
    rs = random.Random(9)
    lens = [rs.randint(1, 400) for _ in range(137)]
    base = str(tmp_path / "q")
    shard.write_result_db(base, [(3 * i + 1, "A" * (l - 1)) for i, l in enumerate(lens)], 0)     # entry length = text + NUL
    for world in (1, 2, 3, 8):
        for rank in range(world):
            out = subprocess.check_output([build.BIN, "shardinfo", base, "%d/%d" % (rank, world)]).decode().split()
            assert (int(out[0]), int(out[1])) == shard.decompose_by_residues(lens, rank, world)
    # three shards with interleaved keys -> one DB, index sorted by key, shards removed
    items = [(k, "entry %d\n" % k) for k in range(40)]
    rs.shuffle(items)
    for r in range(3):
        shard.write_result_db(str(tmp_path / ("res_%d" % r)), items[r::3], 5)
    subprocess.check_call([build.BIN, "mergeshards", str(tmp_path / "res"), "3", "5"])
    assert shard.read_result_db(str(tmp_path / "res")) == dict(items)
    assert [int(l.split("\t")[0]) for l in open(tmp_path / "res.index")] == list(range(40))
    assert open(tmp_path / "res.dbtype", "rb").read() == (5).to_bytes(4, "little") and not os.path.exists(tmp_path / "res_1")
