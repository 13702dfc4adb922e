"""The sharded index (arks_index_build_shard / arks_map_votes_device / arks_votes_resolve_device,
BASELINE configs[3]) against the whole index and the CPU oracle.  One GPU holds all shards here, one
after the other; what is under test is that (1) a shard is exactly the part of the whole map that its
contig ends touch, with the values of the WHOLE map (keys shared with another shard read 0), and
(2) the maximum of the per-shard votes followed by the j_index test is bestContig over the whole map."""
import os
import sys

import numpy as np
import pytest

from util import index_digest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("index_layout")]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))


def _draft(k, seed):
    from arcs_amd import synth
    contigs = synth.make_draft(70000, seed=seed, lengths=(4000, 9000, 2500, 12000), small_frac=0.2,
                               inject=False)
    big = [c for c in contigs if len(c) >= 2500]
    big[0][300:400] = ord("N")
    big[1][50] = ord("N"); big[1][55] = ord("N")
    big[2][200:900] = big[3][100:800]            # the same 700 bases in two contigs (two shards)
    big[4][1000:1400] = big[1][600:1000]         # ... and in contigs two shards apart
    big[5][20:20 + 300] = big[0][900:1200]
    if k % 2 == 0:
        big[4][100:100 + 3 * k] = np.frombuffer((b"AT" * (2 * k))[:3 * k], dtype=np.uint8)
        big[6][700:700 + 2 * k] = np.frombuffer((b"AT" * (2 * k))[:2 * k], dtype=np.uint8)
    return synth.contigs_to_strings(contigs)


def _reads(cs, ends, k, seed, n=1200):
    rng = np.random.Generator(np.random.PCG64(seed))
    genome = "".join(cs)
    reads = []
    for i in range(n):
        L = int(rng.choice([128, 151, 250, k - 1, k, k + 1, 64 + k, 700]))
        p = int(rng.integers(0, len(genome) - L))
        r = list(genome[p:p + L])
        for q in rng.integers(0, L, size=int(rng.integers(0, 3))):
            r[q] = "ACGTNacgtn"[int(rng.integers(10))]
        r = "".join(r)
        reads.append(_rc(r) if i % 2 else r)
    # chimeras of ends that live in different shards: equal counts (the smaller conreci wins), one more
    # window on either side, three-way
    w = k + 9
    for a, b in ((3, 1), (1, 3), (2, 5), (7, 0), (4, 6)):
        reads.append(ends[a][100:100 + w] + ends[b][100:100 + w])
        reads.append(ends[a][100:100 + w + 1] + ends[b][100:100 + w])
        reads.append(ends[a][100:100 + w] + ends[b][100:100 + w + 1])
        reads.append(_rc(ends[a][200:200 + w]) + ends[b][300:300 + w] + ends[(a + b) % 8][50:50 + w])
    reads += ["", "A", "N" * 200, ("AT" * 400)[:k + 70], genome[:k].lower(),
              ends[2][:1000] + ends[1][:1000], ends[2][:1000] + "N" * 500 + ends[1][:1001]]
    return reads


@pytest.mark.parametrize("k,n_shards", [(16, 2), (20, 3), (30, 2), (31, 3), (60, 2), (60, 3), (60, 8), (80, 2)])
def test_shards_equal_whole_index(arks, gpu, oracle, k, n_shards):
    import torch
    cs = _draft(k, 500 + k)
    ends = arks.contig_ends(cs, 500, 1500)
    assert len(ends) >= 16
    ox = oracle.OracleIndex(k).build(ends)
    ok, ov = ox.dump()
    whole = {bytes(kk): int(v) for kk, v in zip(ok, ov)}
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    shards = [arks.ArksIndex.build_shard(ends, k, s, n_shards, device=gpu, want_stats=True) for s in range(n_shards)]

    # (0) the counters of getContigKmers (Arcs.cpp:1093-1107): every shard reports its share, the sums are the
    #     whole map's -- the oracle's serial loop over all the ends -- counter by counter
    want_stats = ox.stats.as_dict()
    assert {f: ix.build_stats[f] for f in want_stats} == want_stats
    assert {f: sum(sh.build_stats[f] for sh in shards) for f in want_stats} == want_stats, (k, n_shards)
    assert sum(sh.build_stats["collisions"] > 0 for sh in shards) >= 2 and want_stats["removed_dup"] > 100
    plain = arks.ArksIndex.build_shard(ends, k, 0, n_shards, device=gpu)      # without counters: the same table
    assert plain.build_stats is None and index_digest(*plain.export()) == index_digest(*shards[0].export())
    plain.close()

    # (1) content: of the keys its own ends visit, with the values of the whole map, those the shard is the FIRST
    #     HOLDER of -- the smallest end of the list that visits the key is one of its own; every key of the whole map
    #     is in exactly one shard
    n_zeroed = n_left = 0
    owner = arks.shard_of_ends([len(e) for e in ends], n_shards)
    assert all(owner[i] == owner[i + 1] for i in range(0, len(ends), 2))      # head and tail together
    load = np.bincount(owner, weights=[len(e) for e in ends], minlength=n_shards)
    assert load.max() - load.min() <= 2 * max(len(e) for e in ends)           # balanced to within one contig
    first_end = {}
    for e, end in enumerate(ends):
        for kk in oracle.OracleIndex(k).build([end]).dump()[0]:
            first_end.setdefault(bytes(kk), e)
    assert set(first_end) == set(whole)
    held = 0
    for s, sh in enumerate(shards):
        own = [e if owner[i] == s else "" for i, e in enumerate(ends)]
        oxs = oracle.OracleIndex(k).build(own)
        keys, local_vals = oxs.dump()
        want_vals = np.array([whole[bytes(kk)] for kk in keys], dtype=np.int32)
        n_zeroed += int(np.count_nonzero((want_vals == 0) & (local_vals != 0)))
        first = np.array([owner[first_end[bytes(kk)]] == s for kk in keys], dtype=bool)
        n_left += int(np.count_nonzero(~first))
        assert not np.any(want_vals[~first])                                  # only keys that read 0 are given up
        gk, gv = sh.export()
        assert len(sh) == int(np.count_nonzero(first))
        assert index_digest(gk, gv) == index_digest(keys[first], want_vals[first]), (k, n_shards, s)
        assert sh.kind == ix.kind
        held += len(sh)
    assert n_zeroed > 100 and n_left > 50        # the draft does have keys shared between shards
    assert held == len(whole)

    # (2) mapping: max of the shard votes == the whole index == the oracle
    reads = _reads(cs, ends, k, 900 + k)
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    ev = torch.ones(len(reads), dtype=torch.uint8, device="cuda")
    ev[5::17] = 0                                         # reads bestContig is not called for
    votes = None
    for sh in shards:
        v = arks.map_votes_packed(sh, packed, eval_mask=ev).clone()
        votes = v if votes is None else torch.maximum(votes, v)
    whole_votes = arks.map_votes_packed(ix, packed, eval_mask=ev)
    assert torch.equal(votes, whole_votes)
    evh = ev.cpu().numpy()
    for j in (0.55, 0.05, 0.0, -1.0):
        got = arks.resolve_votes(votes, packed, k, j).cpu().tolist()
        plain = arks.map_reads_packed(ix, packed, j, eval_mask=ev).cpu().tolist()
        want = [ox.best_contig(r, j) if evh[i] else 0 for i, r in enumerate(reads)]
        assert plain == want, (k, j)
        assert got == want, (k, n_shards, j)
    assert len({c for c in want if c}) > 8

    # (3) the counters of the read stage (Arcs.cpp:1329-1340) over the shards: found, recorded and dups add up (every
    #     key is in one shard), total_valid / bad / windows are the same in every shard, reads_pass / reads_fail come
    #     from the folded votes -- equal to the whole index's and to the oracle's serial loop
    for j in (0.55, 0.0):
        st = oracle.MapStats()
        for i, r in enumerate(reads):
            if evh[i]:
                ox.best_contig(r, j, st)
        want_st = st.as_dict()
        whole_st = torch.zeros(8, dtype=torch.int64, device="cuda")
        arks.map_reads_packed(ix, packed, j, eval_mask=ev, stats=whole_st)
        names = list(want_st)
        assert dict(zip(names, whole_st.cpu().tolist())) == want_st, (k, j)
        parts = []
        for sh in shards:
            t = torch.zeros(8, dtype=torch.int64, device="cuda")
            arks.map_reads_packed(sh, packed, j, eval_mask=ev, stats=t)
            parts.append(dict(zip(names, t.cpu().tolist())))
        folded = torch.zeros(8, dtype=torch.int64, device="cuda")
        arks.count_votes(votes, packed, k, j, folded, eval_mask=ev)
        got_st = dict(zip(names, folded.cpu().tolist()))
        assert got_st["reads_pass"] + got_st["reads_fail"] == int(evh.sum())
        for f in ("found", "recorded", "dups"):
            got_st[f] = sum(p[f] for p in parts)
        for f in ("total_valid", "bad", "windows"):
            assert len({p[f] for p in parts}) == 1
            got_st[f] = parts[0][f]
        assert got_st == want_st, (k, n_shards, j)
        assert want_st["dups"] > 0
    for sh in shards:
        sh.close()
    ix.close()


def test_hash_layout_shards(arks, gpu, oracle, monkeypatch):
    """the exact hash table layout (index_kind "hash") shards the same way"""
    import torch
    monkeypatch.setitem(arks.api.BUILD_DEFAULTS, "index_kind", "hash")
    k, n_shards = 40, 3
    cs = _draft(k, 77)
    ends = arks.contig_ends(cs, 500, 1500)
    ox = oracle.OracleIndex(k).build(ends)
    shards = [arks.ArksIndex.build_shard(ends, k, s, n_shards, device=gpu) for s in range(n_shards)]
    assert all(sh.kind == 0 for sh in shards)
    reads = _reads(cs, ends, k, 78, n=600)
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    votes = None
    for sh in shards:
        v = arks.map_votes_packed(sh, packed).clone()
        votes = v if votes is None else arks.max_votes(votes, v, device=gpu)      # arks_votes_max_device
    for j in (0.55, 0.0):
        got = arks.resolve_votes(votes, packed, k, j).cpu().tolist()
        assert got == [ox.best_contig(r, j) for r in reads], j
    # first holders only: the tables of the shards hold every key of the whole map once, and the found / recorded /
    # duplicate counters of the read stage add up
    assert sum(len(sh) for sh in shards) == len(ox)
    st = oracle.MapStats()
    for r in reads:
        ox.best_contig(r, 0.55, st)
    got = {"found": 0, "recorded": 0, "dups": 0}
    for sh in shards:
        t = torch.zeros(8, dtype=torch.int64, device="cuda")
        arks.map_reads_packed(sh, packed, 0.55, stats=t)
        part = dict(zip(st.as_dict(), t.cpu().tolist()))
        for f in got:
            got[f] += part[f]
        assert part["total_valid"] == st.total_valid and part["bad"] == st.bad
    assert got == {f: st.as_dict()[f] for f in got} and st.dups > 0


def _one_rank_case():
    from arcs_amd import synth
    contigs = synth.make_draft(200000, seed=41, lengths=(9000, 14000, 30000))
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 3000, seed=42, mol_len=8000, pairs_per_mol=10)
    return cs, batch, synth.reads_to_strings(batch)


def _one_rank_worker(want_path):
    """runs in its own process (a hung RCCL start-up must not take the test session with it)"""
    import json
    import torch
    import torch.distributed as dist
    import arcs_amd as arks
    from arcs_amd import dist as adist
    want = json.load(open(want_path))
    k, j = want["k"], want["j"]
    cs, batch, reads = _one_rank_case()
    ends = arks.contig_ends(cs, 500, 4000)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(want["port"]))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sh = arks.ArksIndex.build_shard(ends, k, dist.get_rank(), dist.get_world_size(), device=0)
    packed = arks.PackedReads.from_ascii(reads, device=0)
    imap = arks.ImapAccumulator(1 << 16, device=0)
    step = adist.ShardedPairStep(sh, packed, j, pair_ok=batch["pair_ok"].cuda(),
                                 barcode_id=batch["barcode_id"].cuda(), imap=imap)
    step.run()
    votes = step.votes.clone()
    adist.reduce_votes(votes)                       # the collective itself, on one rank: identity
    torch.cuda.synchronize()
    assert torch.equal(votes, step.votes)
    assert step.pair.cpu().tolist() == want["pair"]
    assert imap.triples().tolist() == want["triples"]
    conreci, pair = adist.map_pairs_sharded(sh, packed, j, pair_ok=batch["pair_ok"].cuda())
    assert pair.cpu().tolist() == want["pair"]
    dist.destroy_process_group()
    print("one-rank ok")


def test_one_rank_group_end_to_end(arks, gpu, oracle, tmp_path):
    """dist.map_pairs_sharded / ShardedPairStep under a real RCCL process group of one rank (what a
    single-GPU box offers): gate -> votes -> all-reduce(MAX) -> j_index test -> pair rule -> imap,
    equal to the oracle's pair flow"""
    import json
    import socket
    import subprocess
    import sys
    from util import oracle_pairs
    k, j = 60, 0.55
    cs, batch, reads = _one_rank_case()
    ends = arks.contig_ends(cs, 500, 4000)
    ox = oracle.OracleIndex(k).build(ends)
    _, want_pair, _, want_triples = oracle_pairs(oracle, ox, reads, batch["pair_ok"].numpy(),
                                                 batch["barcode_id"].numpy(), j)
    assert len(want_triples) > 20
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    wp = tmp_path / "want.json"
    wp.write_text(json.dumps({"k": k, "j": j, "port": port, "pair": [int(x) for x in want_pair],
                              "triples": [[int(a), int(b), int(c)] for a, b, c in want_triples]}))
    res = subprocess.run([sys.executable, os.path.abspath(__file__), "one-rank-worker", str(wp)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "one-rank ok" in res.stdout, res.stderr[-3000:]


def _shared_gpu_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    import arcs_amd as arks
    from arcs_amd import dist as adist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    cs, batch, reads = _one_rank_case()
    ends = arks.contig_ends(cs, 500, 4000)
    sh = arks.ArksIndex.build_shard(ends, 60, rank, world, device=0)          # this rank's part only
    packed = arks.PackedReads.from_ascii(reads, device=0)
    imap = arks.ImapAccumulator(1 << 16, device=0) if rank == 0 else None
    step = adist.ShardedPairStep(sh, packed, 0.55, pair_ok=batch["pair_ok"].cuda(),
                                 barcode_id=batch["barcode_id"].cuda(), imap=imap)
    step.run()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"pair{rank}.npy"), step.pair.cpu().numpy())
    np.save(os.path.join(out_dir, f"keys{rank}.npy"), np.array([len(sh)]))
    if rank == 0:
        np.save(os.path.join(out_dir, "triples.npy"), imap.triples())
    dist.destroy_process_group()


def test_three_ranks_share_the_gpu(arks, gpu, oracle, tmp_path):
    """the multi-process flow of the sharded configuration with the device kernels: three ranks (gloo;
    they share the one GPU of the box), rank r builds and holds only shard r, all map the same batch,
    all-reduce(MAX) of the votes -- every rank ends with the oracle's pairs, rank 0 with its IndexMap"""
    import socket
    import torch.multiprocessing as mp
    from util import oracle_pairs
    world = 3
    cs, batch, reads = _one_rank_case()
    ends = arks.contig_ends(cs, 500, 4000)
    ox = oracle.OracleIndex(60).build(ends)
    _, want_pair, _, want_triples = oracle_pairs(oracle, ox, reads, batch["pair_ok"].numpy(),
                                                 batch["barcode_id"].numpy(), 0.55)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_shared_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sizes = []
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"pair{r}.npy"))
        assert got.tolist() == [int(x) for x in want_pair], r
        sizes.append(int(np.load(os.path.join(str(tmp_path), f"keys{r}.npy"))[0]))
    assert np.load(os.path.join(str(tmp_path), "triples.npy")).tolist() == want_triples
    assert sum(sizes) >= len(ox) and max(sizes) < len(ox)       # nobody holds the whole map


if __name__ == "__main__" and len(sys.argv) == 3 and sys.argv[1] == "one-rank-worker":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _one_rank_worker(sys.argv[2])
