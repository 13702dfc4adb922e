"""arks_exchange (include/arks_hip.h): the sharded seed table -- seeds bucketed by owner on the device, routed,
answered, routed back, mapped -- against the CPU oracle.
  * the ranks of one process (arks_exchange_create_local: a host thread per rank, all shards on the box's one GPU,
    the owners read and write the askers' buffers directly): 1, 2, 3 and 8 ranks, uneven batches, a rank without
    reads, two batches in a row (buffers are reused), reads with invalid bases, long reads (more seeds than the
    kernels keep in registers: the first batch overflows its regions and is bucketed again);
  * two batches in flight (submit n + 1 before complete n, two streams) over several rounds;
  * the RCCL code path with world = 2, 3 and 8: the library's ncclAllGather / ncclSend / ncclRecv calls go through a
    table of function pointers (include/arks_hip_debug.h), and tests/mock_rccl.cpp fills it with threads-and-copies
    stand-ins -- the offset tables, the groups and the error paths are the ones RCCL would be driven with; bytes
    moved are accounted for; an injected failure leaves no rank waiting and no group open;
  * one rank over a real RCCL communicator (what a single-GPU box offers), in a process of its own;
  * the per-pair flow with the IndexMap."""
import ctypes
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


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mock_rccl(arks, tmp_path_factory):
    """tests/mock_rccl.cpp built against the HIP runtime and installed as the library's RCCL table for the test"""
    out = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "mock_rccl.cpp"),
                           "-L/opt/rocm/lib", "-lamdhip64", "-o", out])
    m = ctypes.CDLL(out)
    m.mock_rccl_table.restype = ctypes.c_void_p
    m.mock_rccl_fail.argtypes = [ctypes.c_int, ctypes.c_long]
    m.mock_rccl_counters.argtypes = [ctypes.POINTER(ctypes.c_long)]

    def counters():
        c = (ctypes.c_long * 7)()
        m.mock_rccl_counters(c)
        return dict(zip(("sends", "recvs", "allgathers", "groups_opened", "groups_closed", "aborts", "bytes"), list(c)))
    m.counters = counters
    assert arks.lib().arks_exchange_debug_set_rccl(ctypes.c_void_p(m.mock_rccl_table())) == 0
    yield m
    arks.lib().arks_exchange_debug_set_rccl(None)


def _create_over_mock(arks, shards):
    """every rank its communicator, concurrently (each one's self test meets the others in the mock's barrier)"""
    world = len(shards)
    uid = arks.SeedExchange.unique_id()
    xs, err = [None] * world, [None] * world

    def mk(r):
        try:
            xs[r] = arks.SeedExchange.create(shards[r], r, world, unique_id=uid)
        except Exception as e:           # noqa: BLE001
            err[r] = e
    ts = [threading.Thread(target=mk, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert err == [None] * world, err
    return xs


def _run_ranks(arks, xs, batches, j, with_stats=True):
    """every rank in its own thread and torch stream; batches[r] = PackedReads (or None) -> (conreci lists, stats).
    With counters the batch is mapped twice: the kernels WITHOUT counters (the timed path of bench.py and of a
    front end that does not pass -v) settle the reads that have no seed entry before the tiles are made -- a code
    path of its own -- and must give the same conreci."""
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
                if with_stats:
                    again = xs[r].map_reads(reads, j)
                    torch.cuda.current_stream().synchronize()
                    assert again.cpu().tolist()[:reads.n_reads] == out[r], "the kernels without counters disagree"
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


def _uneven_parts(reads, world, round_):
    """uneven shares; the last rank of a group of three or more gets nothing in the first round"""
    cuts = sorted(set([0, len(reads)] + [len(reads) * (i + 1) // (world + 1) for i in range(world - 1)]))
    while len(cuts) < world + 1:
        cuts.append(len(reads))
    parts = [reads[cuts[r]:cuts[r + 1]] for r in range(world)]
    if world >= 3 and round_ == 0:
        parts[-2] = parts[-2] + parts[-1]
        parts[-1] = []
    return parts


@pytest.mark.parametrize("k,world", [(60, 2), (31, 3), (60, 8)])
def test_rccl_code_path_over_the_mock_transport(arks, gpu, oracle, mock_rccl, k, world):
    """arks_exchange_create with world > 1: ncclAllGather + ncclSend / ncclRecv groups, against the oracle and with
    every byte accounted for"""
    cs = _draft(k, seed=700 + k)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    reads = _reads(cs, ends, k, seed=701 + k, n=1500)
    genome = "".join(ends)
    reads += [genome[100:100 + 5000], genome[7000:7000 + 1300] + "N" + genome[9000:9700]]
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = _create_over_mock(arks, shards)
    empty = arks.PackedReads.from_ascii([], device=gpu)
    for round_, j in enumerate((0.55, 0.0)):
        parts = _uneven_parts(reads, world, round_)
        batches = [arks.PackedReads.from_ascii(p, device=gpu) if p else empty for p in parts]
        c0 = mock_rccl.counters()
        got, stats = _run_ranks(arks, xs, batches, j)
        c1 = mock_rccl.counters()
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        assert sum(got, []) == want, (k, world, j)
        assert dict(zip(STAT_NAMES, np.sum(stats, axis=0).tolist())) == st.as_dict()
        ex = [x.last_stats() for x in xs]
        assert sum(e["sent"] for e in ex) == sum(e["received"] for e in ex) > 0
        # 8 B per seed that travels and 16 B per answer, nothing else; one all-gather per rank; every group closed
        # (the batch is mapped twice, with and without counters)
        assert c1["bytes"] - c0["bytes"] == 2 * 24 * sum(e["sent"] for e in ex)
        assert c1["allgathers"] - c0["allgathers"] == 2 * world
        assert c1["groups_opened"] - c0["groups_opened"] == 4 * world == c1["groups_closed"] - c0["groups_closed"]
        assert c1["sends"] - c0["sends"] == c1["recvs"] - c0["recvs"]
    for x in xs:
        x.close()
    for sh in shards:
        sh.close()


@pytest.mark.parametrize("kind,nth", [(0, 1), (1, 0), (0, 5), (2, 1)])
def test_a_failing_send_leaves_nobody_waiting(arks, gpu, mock_rccl, kind, nth):
    """the nth ncclSend (0) / ncclRecv (1) / ncclAllGather (2) of the batch fails on whichever rank makes it: that
    rank closes its group, aborts its communicator and returns an error; the others get errors too (their receives
    find no sender, or the barrier breaks) -- nobody hangs, no group stays open"""
    import torch
    k, world = 60, 3
    cs = _draft(k, seed=33)
    ends = arks.contig_ends(cs, 500, 3000)
    reads = _reads(cs, ends, k, seed=34, n=300)
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = _create_over_mock(arks, shards)
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    res = [None] * world
    c0 = mock_rccl.counters()
    mock_rccl.mock_rccl_fail(kind, nth)

    def work(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                xs[r].map_reads(packed, 0.55)
                torch.cuda.current_stream().synchronize()
            res[r] = "mapped"
        except arks.ArksError as e:
            res[r] = "error: " + str(e)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    mock_rccl.mock_rccl_fail(kind, -1)
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    assert any(str(x).startswith("error") for x in res), res
    c1 = mock_rccl.counters()
    assert c1["groups_opened"] - c0["groups_opened"] == c1["groups_closed"] - c0["groups_closed"]
    assert c1["aborts"] > c0["aborts"]
    for x in xs:
        x.close()


@pytest.mark.parametrize("transport,world", [("direct", 3), ("mock", 2)])
def test_two_batches_in_flight(arks, gpu, oracle, mock_rccl, transport, world):
    """submit(n + 1) before complete(n) on two streams, five uneven batches per rank (one rank has fewer and
    completes empty ones): conreci, pair results, counters and the IndexMap against the oracle"""
    import torch
    from util import oracle_pairs
    from arcs_amd import synth, dist as adist
    contigs = synth.make_draft(400000, seed=61, lengths=(9000, 14000, 30000, 61000))
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 6000, seed=62, mol_len=8000, pairs_per_mol=10)
    reads = synth.reads_to_strings(batch)
    ends = arks.contig_ends(cs, 500, 30000)
    ox = oracle.OracleIndex(60).build(ends)
    want_c, want_pair, want_st, want_triples = oracle_pairs(oracle, ox, reads, batch["pair_ok"].numpy(),
                                                            batch["barcode_id"].numpy(), 0.55)
    shards = [arks.ArksIndex.build_seed_shard(ends, 60, r, world, device=gpu) for r in range(world)]
    xs = _create_over_mock(arks, shards) if transport == "mock" else arks.SeedExchange.create_local(shards)
    n_calls = 5
    res, err = [None] * world, [None] * world

    def work(r):
        try:
            torch.cuda.set_device(0)
            lo, hi = adist.shard_pairs(len(reads) // 2, r, world)
            nb = n_calls if r else 3                      # rank 0 has three batches, the others five
            cuts = [lo + (hi - lo) * i * i // (nb * nb) for i in range(nb + 1)]     # growing sizes
            streams = [torch.cuda.Stream() for _ in range(3)]
            batches = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                batches.append((arks.PackedReads.from_ascii(reads[2 * a:2 * b], device=gpu),
                                batch["pair_ok"][a:b].cuda(), batch["barcode_id"][a:b].cuda().contiguous()))
            imap = arks.ImapAccumulator(1 << 12, device=gpu)
            st = torch.zeros(8, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            out = xs[r].map_pairs_pipelined(batches, 0.55, streams, imap=imap, stats=st, n_calls=n_calls)
            for q in streams:
                q.synchronize()
            out = [o for o in out if o[1] is not None]
            # the same batches through the other two paths: the kernels WITHOUT counters (what `arcs` runs without -v
            # and bench.py times: absent reads settled per chunk), once with the pair gate folded into the bucketing
            # kernel (arks_exchange_submit_pairs) and once with the gate as a launch of its own (arks_exchange_submit)
            for fold in (True, False):
                again = xs[r].map_pairs_pipelined(batches, 0.55, streams, n_calls=n_calls, fold_gate=fold)
                for q in streams:
                    q.synchronize()
                again = [o for o in again if o[1] is not None]
                assert len(again) == len(out)
                for (c1, p1), (c2, p2) in zip(out, again):
                    assert torch.equal(c1, c2) and torch.equal(p1, p2), ("no counters", fold)
            res[r] = (np.concatenate([o[0].cpu().numpy() for o in out]), np.concatenate([o[1].cpu().numpy() for o in out]),
                      st.cpu().numpy(), imap.triples())
        except Exception as e:           # noqa: BLE001
            err[r] = e
            xs[r].abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=300) for t in ts]
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    assert err == [None] * world, err
    assert np.concatenate([r[0] for r in res]).tolist() == [int(x) for x in want_c]
    assert np.concatenate([r[1] for r in res]).tolist() == [int(x) for x in want_pair]
    assert dict(zip(STAT_NAMES, np.sum([r[2] for r in res], axis=0).tolist())) == {f: want_st[f] for f in STAT_NAMES}
    assert adist.sum_triples(np.concatenate([r[3] for r in res])).tolist() == want_triples
    for x in xs:
        x.close()


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
        parts = _uneven_parts(reads, world, round_)
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


def test_one_thread_drives_the_whole_group(arks, gpu, oracle):
    """arks_exchange_complete_group: the ranks' stages interleaved by one caller (what arcs --index-sharded does), two
    rounds in flight"""
    import torch
    k, world = 60, 3
    cs = _draft(k, seed=515)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    reads = _reads(cs, ends, k, seed=516, n=1800)
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    streams = [[torch.cuda.Stream() for _ in range(2)] for _ in range(world)]
    empty = arks.PackedReads.from_ascii([], device=gpu)
    rounds = [reads[0:500], reads[500:1300], reads[1300:1800]]
    kept, outs = [], []
    st = [torch.zeros(8, dtype=torch.int64, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()

    def submit(i):
        parts = _uneven_parts(rounds[i], world, i)
        row = []
        for r in range(world):
            with torch.cuda.stream(streams[r][i % 2]):
                b = arks.PackedReads.from_ascii(parts[r], device=gpu) if parts[r] else empty
                row.append((b, xs[r].submit(b, 0.55, stats=st[r])))
        kept.append(row)
    for i in range(len(rounds)):
        submit(i)
        if i:
            arks.SeedExchange.complete_group(xs)
    arks.SeedExchange.complete_group(xs)
    torch.cuda.synchronize()
    ost = oracle.MapStats()
    for i, row in enumerate(kept):
        got = sum([c.cpu().tolist()[:b.n_reads] for b, c in row], [])
        assert got == [ox.best_contig(r, 0.55, ost) for r in rounds[i]], i
    assert dict(zip(STAT_NAMES, np.sum([x.cpu().numpy() for x in st], axis=0).tolist())) == ost.as_dict()
    for x in xs:
        x.close()


@pytest.mark.parametrize("with_stats", [True, False])
def test_two_batches_in_flight_on_one_stream(arks, gpu, oracle, with_stats):
    """Both batches in flight on the SAME stream (the header allows it), the second one larger: the order on the stream
    is bucket A, bucket B, map A, map B -- A's map dirties the scratch block B's bucket launch zeroed, and B's queues
    have replaced A's (the map step trusted what the bucket step had cached).  Then the same shapes again:
    no stream is drained any more."""
    import torch
    k, world = 60, 2
    cs = _draft(k, seed=915)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    reads = _reads(cs, ends, k, seed=916, n=2600)
    genome = "".join(ends)
    reads += [genome[a:a + 900] for a in range(0, 40000, 1700)]          # long reads: work for the queued kernels
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    rounds = [reads[0:300], reads[300:2100], reads[2100:], reads[300:2100], reads[2100:]]   # B larger than A: the queues grow
    kept = []
    st = [torch.zeros(8, dtype=torch.int64, device="cuda") for _ in range(world)]
    drains = []

    def submit(i):
        parts = [rounds[i][r::world] for r in range(world)]
        row = []
        for r in range(world):
            b = arks.PackedReads.from_ascii(parts[r], device=gpu)
            row.append((parts[r], b, xs[r].submit(b, 0.55, stats=st[r] if with_stats else None)))
        kept.append(row)
    for i in range(len(rounds)):
        submit(i)
        if i:
            arks.SeedExchange.complete_group(xs)
            drains.append(sum(x.last_stats()["stream_syncs"] for x in xs))
    arks.SeedExchange.complete_group(xs)
    torch.cuda.synchronize()
    ost = oracle.MapStats()
    for i, row in enumerate(kept):
        for part, b, c in row:
            assert c.cpu().tolist()[:b.n_reads] == [ox.best_contig(r, 0.55, ost) for r in part], i
    if with_stats:
        assert dict(zip(STAT_NAMES, np.sum([x.cpu().numpy() for x in st], axis=0).tolist())) == ost.as_dict()
    assert drains[-1] == drains[-2], drains   # shapes seen before: the host waits for the counts' event and nothing else
    for x in xs:
        x.close()
    for sh in shards:
        sh.close()


def test_regions_grow_when_a_batch_does_not_fit(arks, gpu, oracle):
    """the regions of the send buffer are sized for 3.3 seeds per read; a batch of long reads (80 seeds each) overflows
    them, is bucketed again with what it needs, and the sizes stick for the next batch"""
    k, world = 60, 2
    cs = _draft(k, seed=411)
    ends = arks.contig_ends(cs, 500, 3000)
    ox = oracle.OracleIndex(k).build(ends)
    genome = "".join(ends)
    rng = np.random.default_rng(5)
    starts = rng.integers(0, len(genome) - 3300, size=700)
    reads = [genome[a:a + 3200] for a in starts]
    shards = [arks.ArksIndex.build_seed_shard(ends, k, r, world, device=gpu) for r in range(world)]
    xs = arks.SeedExchange.create_local(shards)
    for round_ in range(2):
        batches = [arks.PackedReads.from_ascii(reads[r::world], device=gpu) for r in range(world)]
        got, stats = _run_ranks(arks, xs, batches, 0.3)
        st = oracle.MapStats()
        for r in range(world):
            assert got[r] == [ox.best_contig(x, 0.3, st) for x in reads[r::world]]
        assert dict(zip(STAT_NAMES, np.sum(stats, axis=0).tolist())) == st.as_dict()
        assert [x.last_stats()["reruns"] for x in xs] == [1] * world, round_      # once, in the first map of the first round
    for x in xs:
        x.close()


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
