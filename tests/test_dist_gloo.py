"""world_size-2 `gloo` test of the multi-GPU driver logic on CPU: the reads are sharded over two
ranks, each rank maps its slice (here through the CPU oracle -- the device kernels need a GPU; what
is under test is the sharding and the triple/counter merge), and the merged IndexMap equals the
single-process result.  Second half: the sharded-index configuration -- each rank votes with the
part of the contig k-mer map that its shard holds, all-reduce(MAX) of the votes, j_index test.  It also
checks the claim the device code rests on: the winner over the whole map is the maximum of the
per-shard winners."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from arcs_amd import dist as adist, synth
    from oracle import pyoracle as O
    from util import oracle_pairs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs = synth.make_draft(120000, seed=31, lengths=(9000, 14000, 30000))
    cs = synth.contigs_to_strings(contigs)
    ox = O.OracleIndex(40).build(O.contig_ends(cs, 500, 4000))      # every rank: index replica
    batch = synth.make_read_pairs(contigs, n_pairs, seed=32, mol_len=8000, pairs_per_mol=10)
    reads = synth.reads_to_strings(batch)
    lo, hi = adist.shard_pairs(n_pairs, rank, world)
    pair_ok = batch["pair_ok"].numpy()[lo:hi]
    barcode = batch["barcode_id"].numpy()[lo:hi]
    _, pair, st, triples = oracle_pairs(O, ox, reads[2 * lo:2 * hi], pair_ok, barcode, 0.4)
    merged = adist.merge_triples(np.array(triples, dtype=np.uint32).reshape(-1, 3))
    stats = adist.sum_stats([st[f] for f in ("total_valid", "bad", "found", "recorded", "dups",
                                             "reads_pass", "reads_fail", "windows")])
    np.save(os.path.join(out_dir, f"merged{rank}.npy"), merged)
    np.save(os.path.join(out_dir, f"stats{rank}.npy"), stats)
    dist.destroy_process_group()


def test_two_rank_sharding_and_merge(tmp_path, oracle):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from arcs_amd import dist as adist, synth
    from util import oracle_pairs
    n_pairs, world = 901, 2
    assert [adist.shard_pairs(n_pairs, r, world) for r in range(world)] == [(0, 451), (451, 901)]
    assert adist.shard_pairs(5, 2, 3) == (4, 5) and adist.shard_pairs(0, 0, 2) == (0, 0)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pairs, str(tmp_path)), nprocs=world, join=True)
    contigs = synth.make_draft(120000, seed=31, lengths=(9000, 14000, 30000))
    cs = synth.contigs_to_strings(contigs)
    ox = oracle.OracleIndex(40).build(oracle.contig_ends(cs, 500, 4000))
    batch = synth.make_read_pairs(contigs, n_pairs, seed=32, mol_len=8000, pairs_per_mol=10)
    reads = synth.reads_to_strings(batch)
    _, pair, st, triples = oracle_pairs(oracle, ox, reads, batch["pair_ok"].numpy(),
                                        batch["barcode_id"].numpy(), 0.4)
    want = np.array(triples, dtype=np.uint32).reshape(-1, 3)
    assert len(want) > 20
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"merged{r}.npy"))
        assert got.tolist() == want.tolist()
        stats = np.load(os.path.join(str(tmp_path), f"stats{r}.npy"))
        assert stats.tolist() == [st[f] for f in ("total_valid", "bad", "found", "recorded", "dups",
                                                  "reads_pass", "reads_fail", "windows")]


def test_sum_triples():
    from arcs_amd import dist as adist
    rows = np.array([[5, 2, 1], [1, 9, 4], [5, 2, 3], [1, 3, 1]], dtype=np.uint32)
    assert adist.sum_triples(rows).tolist() == [[1, 3, 1], [1, 9, 4], [5, 2, 4]]
    assert adist.sum_triples(np.zeros((0, 3))).shape == (0, 3)


def _shard_votes(O, ox, reads, k, shard, n_shards):
    """per-read vote (dist.pack_vote) of the ends of one shard: bestContig's walk (Arcs.cpp:959-1004)
    restricted to the conreci the shard owns -- the values are those of the whole map, which is what a
    shard built by arks_index_build_shard holds"""
    from arcs_amd import dist as adist
    out = np.zeros(len(reads), dtype=np.int64)
    for r, read in enumerate(reads):
        hist = {}
        for i in range(len(read) - k + 1):
            key = O.key(read, i, k)
            if key is None:
                continue
            c = ox.get(key)
            if c > 0 and ((c - 1) // 2) % n_shards == shard:       # any partition of the contigs serves here
                hist[c] = hist.get(c, 0) + 1
        best, cnt = 0, 0
        for c in sorted(hist):
            if hist[c] > cnt:
                best, cnt = c, hist[c]
        out[r] = adist.pack_vote(cnt, best)
    return out


def _sharded_case(O):
    from arcs_amd import synth
    k = 40
    contigs = synth.make_draft(60000, seed=51, lengths=(5000, 8000, 3000), inject=False)
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 150, seed=52, mol_len=4000, pairs_per_mol=10)
    reads = synth.reads_to_strings(batch)
    ends = O.contig_ends(cs, 500, 2000)
    ends += [ends[0][100:160] + "A", "C" + ends[3][5:75]]        # keys shared between shards read 0
    # 21 windows each of ends 4 and 1 (two shards): the smaller conreci wins; 22 v 21: the larger count
    reads = reads + [ends[3][200:260] + "N" + ends[0][300:360], ends[2][200:261] + "N" + ends[5][300:360]]
    return k, ends, reads


def _sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from arcs_amd import dist as adist
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, ends, reads = _sharded_case(O)
    ox = O.OracleIndex(k).build(ends)
    votes = torch.from_numpy(_shard_votes(O, ox, reads, k, rank, world))
    adist.reduce_votes(votes)
    np.save(os.path.join(out_dir, f"votes{rank}.npy"), votes.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_index_votes(tmp_path, oracle):
    import torch.multiprocessing as mp
    from arcs_amd import dist as adist
    assert adist.unpack_vote(adist.pack_vote(7, 12)) == (7, 12) and adist.pack_vote(0, 5) == 0
    assert adist.pack_vote(7, 3) > adist.pack_vote(7, 4) > adist.pack_vote(6, 1)      # tie -> smaller conreci
    world = 2
    port = _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    k, ends, reads = _sharded_case(oracle)
    ox = oracle.OracleIndex(k).build(ends)
    v0 = np.load(os.path.join(str(tmp_path), "votes0.npy"))
    v1 = np.load(os.path.join(str(tmp_path), "votes1.npy"))
    assert v0.tolist() == v1.tolist()
    for j in (0.55, 0.2, 0.0):
        got = []
        for v, read in zip(v0, reads):
            cnt, c = adist.unpack_vote(v)
            total = max(len(read) - k + 1, 0)
            got.append(c if cnt > 0 and cnt / total > j else 0)
        want = [ox.best_contig(r, j) for r in reads]
        assert got == want, j
    assert len({c for c in want if c}) > 5
    assert want[-2] == 1 and want[-1] == 3


def _seed_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from arcs_amd import api, dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # an owner's answer is a function of (owner, m-mer): the test can tell who answered what
    api.seeds_probe = lambda index, asked: torch.stack([asked * 3 + index, asked ^ 0x5A5A], 1).reshape(-1)
    g = torch.Generator().manual_seed(100 + rank)
    for n in ((0, 7, 1000)[rank % 3], 513, 0 if rank == 1 else 64):         # uneven, also empty, batches
        mmer = torch.randint(0, 1 << 40, (n,), generator=g, dtype=torch.int64)
        owner = (mmer % world).to(torch.int32)                               # any deterministic ownership
        ans = adist.exchange_seeds(rank, mmer, owner)                        # `index` only reaches the fake probe
        want = torch.stack([mmer * 3 + owner.to(torch.int64), mmer ^ 0x5A5A], 1).reshape(-1)
        assert torch.equal(ans, want), (rank, n)
    open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_seed_exchange_routes_to_owners_and_back(tmp_path, world):
    """the all-to-all of the sharded seed table (arcs_amd.dist.exchange_seeds) on CPU tensors: every seed is
    answered by its owner and the answers come back in the asking rank's own seed order, for batches of
    different sizes per rank, an empty batch on one rank included"""
    import torch.multiprocessing as mp
    mp.spawn(_seed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), f"ok{r}")) for r in range(world))
