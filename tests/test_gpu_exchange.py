"""arks_exchange (include/arks_hip.h): the sharded seed table as one collective call per batch -- seeds bucketed by
owner on the device, routed, answered, routed back, mapped -- against the CPU oracle.
  * the ranks of one process (arks_exchange_create_local: a host thread per rank, all shards on the box's one GPU,
    device copies behind a barrier): 1, 2, 3 and 8 ranks, uneven batches, a rank without reads, two batches in a row
    (buffers are reused), reads with invalid bases, long reads (more seeds than the kernels keep in registers);
  * one rank over a real RCCL communicator (what a single-GPU box offers), in a process of its own;
  * the per-pair flow with the IndexMap."""
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from test_gpu_sharded import _draft, _reads

pytestmark = pytest.mark.gpu
STAT_NAMES = ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows")


def _run_ranks(arks, xs, batches, j, with_stats=True):
    """every rank in its own thread and torch stream; batches[r] = PackedReads (or None) -> (conreci lists, stats)"""
    import torch
    world = len(xs)
    out, stats, err = [None] * world, [None] * world, [None] * world

    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                st = torch.zeros(8, dtype=torch.int64, device="cuda") if with_stats else None
                reads = batches[r]
                got = xs[r].map_reads(reads, j, stats=st)
                torch.cuda.current_stream().synchronize()
                out[r] = got.cpu().tolist()[:reads.n_reads]
                stats[r] = st.cpu().numpy() if with_stats else None
        except Exception as e:           # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    assert err == [None] * world, err
    return out, stats


@pytest.mark.parametrize("k,world", [(60, 1), (60, 2), (31, 3), (60, 8), (20, 2), (96, 3)])
def test_local_ranks_against_the_oracle(arks, gpu, oracle, k, world):
    import torch
    cs = _draft(k, seed=900 + k)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    reads = _reads(cs, ends, k, seed=901 + k, n=1500)
    genome = "".join(ends)
    reads += [genome[100:100 + 5000], genome[7000:7000 + 1300] + "N" + genome[9000:9700]]      # long reads: > 4 seeds
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    empty = arks.PackedReads.from_ascii([], device=gpu)
    for round_, j in enumerate((0.55, 0.0)):
        # uneven shares; the last rank of a group of three or more gets nothing in the first round
        cuts = sorted(set([0, len(reads)] + [len(reads) * (i + 1) // (world + 1) for i in range(world - 1)]))
        while len(cuts) < world + 1:
            cuts.append(len(reads))
        parts = [reads[cuts[r]:cuts[r + 1]] for r in range(world)]
        if world >= 3 and round_ == 0:
            parts[-2] = parts[-2] + parts[-1]
            parts[-1] = []
        batches = [arks.PackedReads.from_ascii(p, device=gpu) if p else empty for p in parts]
        got, stats = _run_ranks(arks, xs, batches, j)
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        assert sum(got, []) == want, (k, world, j)
        assert dict(zip(STAT_NAMES, np.sum(stats, axis=0).tolist())) == st.as_dict()
        ex = [x.last_stats() for x in xs]
        assert sum(e["sent"] for e in ex) == sum(e["received"] for e in ex)
        assert world == 1 or sum(e["sent"] for e in ex) > 0
    for x in xs:
        x.close()
    for sh in shards:
        sh.close()


def test_pairs_flow_with_the_indexmap(arks, gpu, oracle):
    """gate -> exchanged map -> pair rule + IndexMap on three local ranks; merged triples and summed counters == oracle"""
    import torch
    from util import oracle_pairs
    from arcs_amd import synth, dist as adist
    contigs = synth.make_draft(400000, seed=51, lengths=(9000, 14000, 30000, 61000))
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 6000, seed=52, mol_len=8000, pairs_per_mol=10)
    reads = synth.reads_to_strings(batch)
    ends = arks.contig_ends(cs, 500, 30000)
    ox = oracle.OracleIndex(60).build(ends)
    want_c, want_pair, want_st, want_triples = oracle_pairs(oracle, ox, reads, batch["pair_ok"].numpy(),
                                                            batch["barcode_id"].numpy(), 0.55)
    world = 3
    shards = [arks.ArksIndex.build_seed_shard(ends, 60, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    res = [None] * world
    err = [None] * world

    def work(r):
        try:
            lo, hi = adist.shard_pairs(len(reads) // 2, r, world)
            with torch.cuda.stream(torch.cuda.Stream()):
                packed = arks.PackedReads.from_ascii(reads[2 * lo:2 * hi], device=gpu)
                imap = arks.ImapAccumulator(1 << 12, device=gpu)
                st = torch.zeros(8, dtype=torch.int64, device="cuda")
                conreci, pair = xs[r].map_pairs(packed, 0.55, pair_ok=batch["pair_ok"][lo:hi].cuda(),
                                                barcode_id=batch["barcode_id"][lo:hi].cuda().contiguous(), imap=imap,
                                                stats=st)
                torch.cuda.current_stream().synchronize()
                res[r] = (conreci.cpu().numpy(), pair.cpu().numpy(), st.cpu().numpy(), imap.triples())
        except Exception as e:           # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert err == [None] * world, err
    assert np.concatenate([r[0] for r in res]).tolist() == [int(x) for x in want_c]
    assert np.concatenate([r[1] for r in res]).tolist() == [int(x) for x in want_pair]
    assert dict(zip(STAT_NAMES, np.sum([r[2] for r in res], axis=0).tolist())) == {f: want_st[f] for f in STAT_NAMES}
    assert adist.sum_triples(np.concatenate([r[3] for r in res])).tolist() == want_triples
    for x in xs:
        x.close()


def test_a_rank_that_gives_up_does_not_hang_the_others(arks, gpu):
    """a local rank whose driver fails outside the library (here: it simply aborts) -- the other ranks' call returns an
    error instead of waiting at the barrier for ever, and so does every later call on the group"""
    import torch
    k, world = 60, 3
    cs = _draft(k, seed=33)
    ends = arks.contig_ends(cs, 500, 3000)
    reads = _reads(cs, ends, k, seed=34, n=300)
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    res = [None] * world

    def work(r):
        try:
            if r == 2:
                xs[r].abort()
                res[r] = "gave up"
                return
            with torch.cuda.stream(torch.cuda.Stream()):
                xs[r].map_reads(packed, 0.55)
            res[r] = "mapped"
        except arks.ArksError as e:
            res[r] = "error: " + str(e)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not any(t.is_alive() for t in ts)
    assert res[2] == "gave up" and all(str(res[r]).startswith("error") for r in (0, 1)), res
    with pytest.raises(arks.ArksError):
        xs[0].map_reads(packed, 0.55)
    for x in xs:
        x.close()


def _rccl_worker():
    """one rank with a real RCCL communicator (ncclCommInitRank through the library's dlopen of librccl)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import arcs_amd as arks
    from oracle import pyoracle as O
    k = 60
    cs = _draft(k, seed=77)
    ends = arks.contig_ends(cs, 500, 3000)
    reads = _reads(cs, ends, k, seed=78, n=800)
    sh = arks.ArksIndex.build_seed_shard(ends, k, 0, 1, device=0)
    x = arks.SeedExchange.create(sh, 0, 1, unique_id=arks.SeedExchange.unique_id())
    packed = arks.PackedReads.from_ascii(reads, device=0)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    got = x.map_reads(packed, 0.55, stats=stats).cpu().tolist()
    ox = O.OracleIndex(k).build(ends)
    st = O.MapStats()
    assert got == [ox.best_contig(r, 0.55, st) for r in reads]
    assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == st.as_dict()
    x.close()
    print("rccl rank ok")


def test_one_rank_over_rccl(arks, gpu):
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "rccl-worker"], capture_output=True, text=True,
                         timeout=600, cwd=os.path.dirname(os.path.abspath(__file__)))
    assert res.returncode == 0 and "rccl rank ok" in res.stdout, res.stderr[-3000:]


if __name__ == "__main__" and len(sys.argv) == 2 and sys.argv[1] == "rccl-worker":
    _rccl_worker()
