"""The north star's sharded configuration (BASELINE configs[3]): the seed table of the index split over
the ranks by a hash prefix of the m-mer (arks_index_build_seed_shard), every read mapped on ONE rank, its
seeds routed to their owners and the answers routed back (arcs_amd.dist.exchange_seeds: the all-to-all).
Against the CPU oracle and against the whole index:
  * one process holding all shards one after the other (the exchange by hand, answers stitched by owner);
  * three processes sharing the GPU (gloo; each builds and holds its shard only, each maps its own reads)."""
import os
import socket
import sys

import numpy as np
import pytest

from test_gpu_sharded import _draft, _reads

pytestmark = pytest.mark.gpu
STAT_NAMES = ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows")


@pytest.mark.parametrize("k,n_ranks", [(20, 2), (30, 3), (31, 2), (60, 2), (60, 3), (60, 8), (80, 3), (96, 2)])
def test_answers_of_the_owners_give_the_whole_index(arks, gpu, oracle, k, n_ranks):
    import torch
    cs = _draft(k, seed=700 + k)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    reads = _reads(cs, ends, k, seed=800 + k)
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, n_ranks, device=gpu) for r in range(n_ranks)]
    whole = arks.ArksIndex.build(ends, k, device=gpu)
    assert all(sh.kind == 2 and sh.seed_ranks == n_ranks and len(sh) == len(ox) for sh in shards)
    # a shard is smaller than the whole seed index (the table is the bulk of it; the text and, for the general
    # kernels, a minimizer table are replicated -- at small k, i.e. short minimizer windows, that one is big)
    assert not (k >= 60 and n_ranks >= 3) or max(sh.device_bytes for sh in shards) < whole.device_bytes
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    home = shards[0]
    counts = arks.api.seed_counts(home, packed)
    seed_off = torch.zeros(packed.n_reads + 1, dtype=torch.int64, device="cuda")
    seed_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
    mmer, owner = arks.api.seeds_fill(home, packed, seed_off)
    assert int(owner.min()) >= 0 and int(owner.max()) < n_ranks
    answers = torch.zeros(2 * mmer.numel(), dtype=torch.int64, device="cuda")
    for r, sh in enumerate(shards):                       # the exchange, by hand: owner r answers its seeds
        sel = (owner == r).nonzero().flatten()
        answers.view(-1, 2)[sel] = arks.api.seeds_probe(sh, mmer[sel].contiguous()).view(-1, 2)
    # a seed asked of a rank that does not own it finds nothing there: the shards are disjoint
    other = arks.api.seeds_probe(shards[1], mmer[(owner == 0).nonzero().flatten()].contiguous())
    assert int((other.view(-1, 2)[:, 0] != 0).sum()) == 0
    for j in (0.55, 0.0):
        stats = torch.zeros(8, dtype=torch.int64, device="cuda")
        got = arks.api.map_reads_seeded(home, packed, j, seed_off, answers, stats=stats)
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        assert got.cpu().tolist() == want, j
        assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == st.as_dict()
        # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
        assert arks.api.map_reads_seeded(home, packed, j, seed_off, answers).cpu().tolist() == want, j
        assert arks.map_reads_packed(whole, packed, j).cpu().tolist() == want
    with pytest.raises(arks.ArksError):                   # a shard cannot answer the plain call
        arks.map_reads_packed(home, packed, 0.5)
    for sh in shards:
        sh.close()
    whole.close()


def _case():
    from arcs_amd import synth
    contigs = synth.make_draft(400000, seed=51, lengths=(9000, 14000, 30000, 61000))
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 6000, seed=52, mol_len=8000, pairs_per_mol=10)
    return cs, batch, synth.reads_to_strings(batch)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import arcs_amd as arks
    from arcs_amd import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cs, batch, reads = _case()
    ends = arks.contig_ends(cs, 500, 30000)
    sh = arks.ArksIndex.build_seed_shard(ends, 60, rank, world, device=0)       # this rank's seeds only
    lo, hi = adist.shard_pairs(len(reads) // 2, rank, world)                     # this rank's read pairs
    if rank == world - 1:
        hi = lo                                                                  # ... and a rank without any
    packed = arks.PackedReads.from_ascii(reads[2 * lo:2 * hi], device=0)
    imap = arks.ImapAccumulator(1 << 12, device=0)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    for _ in range(2):                                                           # two batches: buffers are reused
        stats.zero_()
        conreci, pair = adist.map_pairs_seed_sharded(sh, packed, 0.55, pair_ok=batch["pair_ok"][lo:hi].cuda(),
                                                     barcode_id=batch["barcode_id"][lo:hi].cuda().contiguous(),
                                                     imap=imap if _ == 0 else None, stats=stats)
    torch.cuda.synchronize()
    merged = adist.merge_triples(imap.triples())
    total = adist.sum_stats(stats.cpu().numpy())
    np.save(os.path.join(out_dir, f"conreci{rank}.npy"), conreci.cpu().numpy()[:2 * (hi - lo)])
    np.save(os.path.join(out_dir, f"bytes{rank}.npy"), np.array([sh.device_bytes]))
    if rank == 0:
        np.save(os.path.join(out_dir, "triples.npy"), merged)
        np.save(os.path.join(out_dir, "stats.npy"), total)
    dist.destroy_process_group()


def test_three_ranks_route_their_seeds(arks, gpu, oracle, tmp_path):
    """three processes (gloo; they share the box's one GPU): rank r holds shard r of the seed table and maps
    its own third of the read pairs -- one rank has none --, seeds and answers travel by all_to_all_single;
    conreci of every read, the merged IndexMap and the summed counters equal the oracle's"""
    import torch.multiprocessing as mp
    from util import oracle_pairs
    from arcs_amd import dist as adist
    world = 3
    cs, batch, reads = _case()
    ends = arks.contig_ends(cs, 500, 30000)
    ox = oracle.OracleIndex(60).build(ends)
    n_used = adist.shard_pairs(len(reads) // 2, world - 1, world)[0]            # the last rank maps nothing
    want_c, want_pair, want_st, want_triples = oracle_pairs(
        oracle, ox, reads[:2 * n_used], batch["pair_ok"].numpy()[:n_used], batch["barcode_id"].numpy()[:n_used], 0.55)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(os.path.join(str(tmp_path), f"conreci{r}.npy")) for r in range(world)])
    assert got.tolist() == [int(x) for x in want_c]
    assert np.load(os.path.join(str(tmp_path), "triples.npy")).tolist() == want_triples
    assert dict(zip(STAT_NAMES, np.load(os.path.join(str(tmp_path), "stats.npy")).tolist())) == \
        {f: want_st[f] for f in STAT_NAMES}
    whole = arks.ArksIndex.build(ends, 60, device=gpu)
    sizes = [int(np.load(os.path.join(str(tmp_path), f"bytes{r}.npy"))[0]) for r in range(world)]
    assert whole.kind == 2 and max(sizes) < 0.7 * whole.device_bytes             # nobody holds the whole table


def _rccl_worker(port):
    """one rank, backend nccl (= RCCL): the exchange runs through all_to_all_single although the rank owns
    every seed (ARKS_FORCE_EXCHANGE); in its own process, a hung RCCL start-up must not take the session along"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import arcs_amd as arks
    from arcs_amd import dist as adist
    from oracle import pyoracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ARKS_FORCE_EXCHANGE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cs, batch, reads = _case()
    ends = arks.contig_ends(cs, 500, 30000)
    sh = arks.ArksIndex.build_seed_shard(ends, 60, 0, 1, device=0)
    packed = arks.PackedReads.from_ascii(reads, device=0)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    got = adist.map_reads_seed_sharded(sh, packed, 0.55, stats=stats).cpu().tolist()
    ox = O.OracleIndex(60).build(ends)
    st = O.MapStats()
    assert got == [ox.best_contig(r, 0.55, st) for r in reads]
    assert adist.map_reads_seed_sharded(sh, packed, 0.55).cpu().tolist() == got      # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
    assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == st.as_dict()
    dist.destroy_process_group()
    print("rccl exchange ok")


def test_exchange_over_rccl(arks, gpu, tmp_path):
    """exchange_seeds over a real RCCL process group (one rank is what a single-GPU box offers)"""
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "rccl-worker", str(port)], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "rccl exchange ok" in res.stdout, res.stderr[-3000:]


if __name__ == "__main__" and len(sys.argv) == 3 and sys.argv[1] == "rccl-worker":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _rccl_worker(int(sys.argv[2]))
