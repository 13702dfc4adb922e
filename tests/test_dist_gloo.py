"""world_size-2 `gloo` test of the multi-GPU driver logic on CPU: the reads are sharded over two
ranks, each rank maps its slice (here through the CPU oracle -- the device kernels need a GPU; what
is under test is the sharding and the triple/counter merge), and the merged IndexMap equals the
single-process result."""
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
