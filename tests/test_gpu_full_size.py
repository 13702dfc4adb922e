"""Full-size checks through size-independent properties of the path and oracle comparisons: BASELINE configs[1] on a
10 Mbp / 2 M-pair cut (properties) AND at its size -- 50 Mbp + 20 M pairs, the first 1.5 M pairs read for read against
the oracle over the WHOLE draft --, configs[2-4] at 3 Gbp against sub-draft oracles."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_properties_at_scale(arks, gpu, oracle):
    import torch
    from arcs_amd import synth
    k, j = 60, 0.55
    contigs = synth.make_draft(10_000_000, seed=77)
    cs = synth.contigs_to_strings(contigs)
    ends = arks.contig_ends(cs)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    st = ix.build_stats
    # counters are consistent with each other (Arcs.cpp:903-920)
    assert st["recorded"] + st["collisions"] == st["total_kmers"]
    assert st["unique"] <= st["recorded"] and st["removed_dup"] <= st["collisions"]
    assert len(ix) == st["recorded"]
    n_pairs = 2_000_000
    batch = synth.make_read_pairs(contigs, n_pairs, seed=78, device="cuda")
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    stored = torch.zeros(1, dtype=torch.int64, device="cuda")
    imap = arks.ImapAccumulator(1 << 18, device=gpu)
    conreci, pair = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"],
                                          barcode_id=batch["barcode_id"], imap=imap, stats=stats,
                                          stored=stored)
    torch.cuda.synchronize()
    s = dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail",
                  "windows"), stats.cpu().tolist()))
    # every evaluated window is valid or NULL; found = recorded + duplicates (Arcs.cpp:961-991)
    assert s["total_valid"] + s["bad"] == s["windows"]
    assert s["found"] == s["recorded"] + s["dups"]
    c = conreci.cpu().numpy()
    p = pair.cpu().numpy()
    ok = batch["pair_ok"].cpu().numpy().astype(bool)
    # pair rule (Arcs.cpp:1280): agreed end iff both mates name the same non-zero end
    assert ((p != 0) == ((c[0::2] != 0) & (c[0::2] == c[1::2]))).all()
    assert (p[p != 0] == c[0::2][p != 0]).all()
    assert (c[0::2][~ok] == 0).all() and (c[1::2][~ok] == 0).all()   # unpaired names are gated out
    assert s["reads_pass"] == int((c != 0).sum())
    assert int(stored.item()) == int((p != 0).sum())
    t = imap.triples()
    # the IndexMap is a histogram of the stored pairs: checksum of checksums
    assert int(t[:, 2].sum()) == int((p != 0).sum())
    key = batch["barcode_id"].cpu().numpy().astype(np.int64)[p != 0] * (1 << 32) + p[p != 0]
    uk, cnt = np.unique(key, return_counts=True)
    assert len(uk) == len(t)
    assert (t[:, 0].astype(np.int64) * (1 << 32) + t[:, 1] == uk).all() and (t[:, 2] == cnt).all()
    # idempotence: mapping the same resident batch again gives the same per-read result
    c2 = arks.map_reads_packed(ix, reads, j).cpu().numpy()
    rc = reads.read_class.cpu().numpy().astype(bool)
    gate = np.repeat(ok & rc[0::2] & rc[1::2], 2)
    assert (c2[gate] == c[gate]).all()
    # strand symmetry: a read and its reverse complement name the same end
    strs = synth.reads_to_strings({"ascii": batch["ascii"][: 279 * 3000], "offsets": batch["offsets"][:6001]})
    comp = str.maketrans("ACGTN", "TGCAN")
    rcs = [x[::-1].translate(comp) for x in strs]
    assert ix.map_reads(strs, j).tolist() == ix.map_reads(rcs, j).tolist()
    # oracle on a slice (bit-exact)
    ox = oracle.OracleIndex(k).build(oracle.contig_ends(cs))
    want = [ox.best_contig(x, j) for x in strs[:2000]]
    assert ix.map_reads(strs[:2000], j).tolist() == want
    imap.close()
    ix.close()


STAT_NAMES = ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows")


def test_configs1_at_its_size_against_the_whole_oracle(arks, gpu, oracle):
    """BASELINE configs[1] as named: synthetic 50 Mbp draft + 20 M linked-read pairs, k = 60, j = 0.55, one MI355X.
    The whole draft's oracle map (3e7 keys) fits the host, so no sub-draft argument is needed: all six build counters,
    and conreci / pair result / all eight counters of the first 1.5 M pairs, are the oracle's; the 20 M pairs go
    through in one launch with the properties of the path (pair rule, histogram) holding over all of them and the
    first 1.5 M of that launch equal to the oracle as well (results do not depend on what else is in the batch)."""
    import torch
    from arcs_amd import synth
    k, j = 60, 0.55
    contigs = synth.make_draft(50_000_000, seed=synth.SEED)
    cs = synth.contigs_to_strings(contigs)
    ends = arks.contig_ends(cs)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    ox = oracle.OracleIndex(k).build(oracle.contig_ends(cs))
    del cs, ends
    assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
    assert len(ix) == len(ox) > 20_000_000
    threads = min(64, len(os.sched_getaffinity(0)))
    n_all, n_chk = 20_000_000, 1_500_000
    batch = synth.make_read_pairs(contigs, n_all, seed=synth.SEED + 1, device="cuda")
    a = np.concatenate([batch["ascii"][: int(batch["offsets"][2 * n_chk].item())].cpu().numpy(), np.zeros(1, np.uint8)])
    lens = batch["lens"][: 2 * n_chk].cpu().numpy().astype(np.uint32)
    offs = batch["offsets"][: 2 * n_chk].cpu().numpy().astype(np.uint64)
    ok = batch["pair_ok"][:n_chk].cpu().numpy()
    want_c, want_p, want_st = ox.map_pairs(a, offs, lens, j, pair_ok=ok, threads=threads)
    # the 1.5 M pairs alone, with the counters
    head = arks.PackedReads.from_arrays_device(batch["ascii"][: int(batch["offsets"][2 * n_chk].item())],
                                               batch["offsets"][: 2 * n_chk + 1], batch["lens"][: 2 * n_chk], device=gpu)
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    c, p = arks.map_pairs_packed(ix, head, j, pair_ok=batch["pair_ok"][:n_chk], stats=st)
    torch.cuda.synchronize()
    assert (c.cpu().numpy() == want_c).all() and (p.cpu().numpy() == want_p).all()
    # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
    c, p = arks.map_pairs_packed(ix, head, j, pair_ok=batch["pair_ok"][:n_chk])
    torch.cuda.synchronize()
    assert (c.cpu().numpy() == want_c).all() and (p.cpu().numpy() == want_p).all()
    assert dict(zip(STAT_NAMES, st.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}
    del head
    # all 20 M pairs in one launch
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    stored = torch.zeros(1, dtype=torch.int64, device="cuda")
    imap = arks.ImapAccumulator(1 << 20, device=gpu)
    c, p = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"], imap=imap,
                                 stats=stats, stored=stored)
    torch.cuda.synchronize()
    c, p = c.cpu().numpy(), p.cpu().numpy()
    assert (c[: 2 * n_chk] == want_c).all() and (p[:n_chk] == want_p).all()
    s = dict(zip(STAT_NAMES, stats.cpu().tolist()))
    assert s["windows"] == 3_220_000_000 - 161 * int((~(np.repeat(batch["pair_ok"].cpu().numpy().astype(bool), 2)
                                                       & reads.read_class.cpu().numpy().astype(bool)[0::2].repeat(2)
                                                       & reads.read_class.cpu().numpy().astype(bool)[1::2].repeat(2))).sum()) // 2
    assert s["total_valid"] + s["bad"] == s["windows"] and s["found"] == s["recorded"] + s["dups"]
    assert ((p != 0) == ((c[0::2] != 0) & (c[0::2] == c[1::2]))).all()
    assert s["reads_pass"] == int((c != 0).sum()) and int(stored.item()) == int((p != 0).sum())
    t = imap.triples()
    key = batch["barcode_id"].cpu().numpy().astype(np.int64)[p != 0] * (1 << 32) + p[p != 0]
    uk, cnt = np.unique(key, return_counts=True)
    assert len(uk) == len(t) and (t[:, 0].astype(np.int64) * (1 << 32) + t[:, 1] == uk).all() and (t[:, 2] == cnt).all()
    imap.close()
    ix.close()


def _ends_of(arks, contigs, end_length=30000):
    ends = []
    for c in contigs:
        cut = arks.end_cutoff(len(c), 500, end_length)
        if cut is not None:
            ends.append(c[:cut].tobytes())
            ends.append(c[len(c) - cut:].tobytes())
    return ends


def _sub_draft(synth, contigs, dup_events, sub_mbp):
    """(bases of the first contigs that make up sub_mbp, the contigs an oracle index for reads drawn
    from them needs)"""
    acc, n_first = 0, 0
    while n_first < len(contigs) and acc < sub_mbp * 1e6:
        acc += len(contigs[n_first])
        n_first += 1
    return acc, synth.closed_contig_set(n_first, dup_events)


def _expected_ends(arks, contigs, touched, origin, r1_len=128, r2_len=151, frag=350, end_length=30000):
    """for error-free pairs with known origin: the contig end (conreci) that holds R1 / R2 completely, or -1
    where the mate is not wholly inside one end of a contig that no injection touched"""
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    starts = np.zeros(len(contigs) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    cut = np.array([arks.end_cutoff(int(n), 500, end_length) or 0 for n in lens], dtype=np.int64)
    valid = cut > 0
    rank = np.cumsum(valid)                      # 1-based rank among the contigs that have ends
    plain = valid & ~np.isin(np.arange(len(contigs)), np.fromiter(touched, dtype=np.int64, count=len(touched)))
    out = []
    for lo, n in ((origin, r1_len), (origin + frag - r2_len, r2_len)):
        ci = np.searchsorted(starts, lo, side="right") - 1
        off = lo - starts[ci]
        inside = off + n <= lens[ci]
        head = inside & (off + n <= cut[ci])
        tail = inside & (off >= lens[ci] - cut[ci])
        e = np.where(head, 2 * rank[ci] - 1, np.where(tail, 2 * rank[ci], -1))   # head wins where both hold (L <= 2e)
        out.append(np.where(plain[ci] & (head != tail), e, -1))
    return out


def test_human_scale_draft(arks, gpu, oracle):
    """BASELINE configs[2]'s index -- 3 Gbp draft, 1.4 G keys -- built on the device, then (1) an oracle
    comparison on 4 M pairs drawn from a sub-draft whose ends the CPU can index (conreci, pair rule, all
    eight counters, IndexMap), (2) known answers for error-free reads of known origin all over the draft,
    (3) the size-independent properties of the path."""
    import torch
    from arcs_amd import synth
    k, j = 60, 0.55
    dup_events, touched = [], set()
    contigs = synth.make_draft(3_000_000_000, seed=synth.SEED, dup_events=dup_events, touched=touched)
    ix = arks.ArksIndex.build(_ends_of(arks, contigs), k, device=gpu)
    st = ix.build_stats
    assert ix.kind == 2          # the seed index: 46 GB of table at this size, chosen because it fits
    assert st["recorded"] + st["collisions"] == st["total_kmers"] and len(ix) == st["recorded"]
    assert st["unique"] <= st["recorded"] and st["removed_dup"] <= st["collisions"]
    assert len(ix) > 1_400_000_000
    genome = torch.from_numpy(np.concatenate(contigs)).cuda()

    # (1) oracle on a sub-draft slice
    acc, members = _sub_draft(synth, contigs, dup_events, 40.0)
    # reads that reach into an (AT)n microsatellite stay in the comparison: the oracle is given the windows around
    # every such stretch of the WHOLE draft (their k-mers recur between sites: fallback table, heavy seeds, the
    # medium and slow kernels at a 1.4 G-key index)
    at_runs = synth.alternating_at_runs(genome, run=12)
    assert len(at_runs) >= 2500
    ox = oracle.sub_draft_index(k, contigs, members, site_runs=at_runs)
    n_pairs = 4_000_000
    batch = synth.make_read_pairs(genome[:acc], n_pairs, seed=4242, device="cuda")
    n_at = int(synth.pairs_touching_microsatellite(batch).sum().item())
    assert n_at > 100, n_at
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    imap = arks.ImapAccumulator(1 << 20, device=gpu)
    conreci, pair = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"],
                                          imap=imap, stats=stats)
    torch.cuda.synchronize()
    ok = batch["pair_ok"].cpu().numpy()
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    want_c, want_p, want_st = ox.map_pairs(a, batch["offsets"].cpu().numpy().astype(np.uint64)[:-1],
                                           batch["lens"].cpu().numpy().astype(np.uint32), j, pair_ok=ok,
                                           threads=min(64, os.cpu_count() or 1))
    assert (conreci.cpu().numpy() == want_c).all()
    assert (pair.cpu().numpy() == want_p).all()
    assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}
    # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
    c2, p2 = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"])
    torch.cuda.synchronize()
    assert (c2.cpu().numpy() == want_c).all() and (p2.cpu().numpy() == want_p).all()
    del c2, p2
    sel = (want_p != 0) & (ok != 0)
    key = batch["barcode_id"].cpu().numpy().astype(np.int64)[sel] * (1 << 32) + want_p[sel]
    uk, cnt = np.unique(key, return_counts=True)
    t = imap.triples()
    assert len(t) == len(uk) and (t[:, 0].astype(np.int64) * (1 << 32) + t[:, 1] == uk).all() and (t[:, 2] == cnt).all()
    imap.close()
    del batch, reads, a

    # (2) error-free pairs from all over the draft: a mate wholly inside one end of an untouched contig is
    # made of that end's k-mers only, every one of them unique in the draft -> count / total = 1 > j
    batch = synth.make_read_pairs(genome, n_pairs, seed=4243, device="cuda", sub_rate=0.0, one_n_rate=0.0,
                                  many_n_rate=0.0, unpaired_rate=0.0, want_origin=True)
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    c = arks.map_reads_packed(ix, reads, j).cpu().numpy()
    e1, e2 = _expected_ends(arks, contigs, touched, batch["origin"].cpu().numpy())
    assert (e1 >= 0).sum() > n_pairs // 10 and (e2 >= 0).sum() > n_pairs // 10
    assert (c[0::2][e1 >= 0] == e1[e1 >= 0]).all()
    assert (c[1::2][e2 >= 0] == e2[e2 >= 0]).all()
    # the conreci of any read is one of the (at most four) ends of the contigs its bases come from, or 0
    assert c.max() <= 2 * sum(1 for x in contigs if len(x) >= 500)

    # (3) properties on ordinary reads (errors, Ns, unpaired names)
    batch = synth.make_read_pairs(genome, n_pairs, seed=4244, device="cuda")
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    stats.zero_()
    stored = torch.zeros(1, dtype=torch.int64, device="cuda")
    imap = arks.ImapAccumulator(1 << 20, device=gpu)
    conreci, pair = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"],
                                          imap=imap, stats=stats, stored=stored)
    torch.cuda.synchronize()
    s = dict(zip(STAT_NAMES, stats.cpu().tolist()))
    assert s["total_valid"] + s["bad"] == s["windows"] and s["found"] == s["recorded"] + s["dups"]
    c, p = conreci.cpu().numpy(), pair.cpu().numpy()
    ok = batch["pair_ok"].cpu().numpy().astype(bool)
    assert ((p != 0) == ((c[0::2] != 0) & (c[0::2] == c[1::2]))).all()
    assert (c[0::2][~ok] == 0).all() and (c[1::2][~ok] == 0).all()
    assert s["reads_pass"] == int((c != 0).sum()) and int(stored.item()) == int((p != 0).sum())
    t = imap.triples()
    assert int(t[:, 2].sum()) == int((p != 0).sum())
    # strand symmetry: the reverse complement of a read names the same end (same multiset of canonical keys)
    strs = synth.reads_to_strings({"ascii": batch["ascii"][: 279 * 5000], "offsets": batch["offsets"][:10001]})
    comp = str.maketrans("ACGTN", "TGCAN")
    assert ix.map_reads(strs, j).tolist() == ix.map_reads([x[::-1].translate(comp) for x in strs], j).tolist()
    # order independence: the same reads in another order give the permuted result
    perm = np.random.default_rng(1).permutation(len(strs))
    assert ix.map_reads([strs[i] for i in perm], j).tolist() == ix.map_reads(strs, j)[perm].tolist()
    imap.close()
    ix.close()


def test_beyond_one_index_in_shards(arks, gpu, oracle):
    """A draft whose contig-end text exceeds what one locality index addresses (2^32 positions: 4.6 Gbp with
    -e 120000, the ends cover every contig) mapped through two index shards (arks_index_build_shard,
    arks_map_votes_device, arks_votes_max_device, arks_votes_resolve_device), compared with the oracle on
    reads drawn from a sub-draft -- not with another sharding."""
    import ctypes as C
    import torch
    from arcs_amd import synth
    from arcs_amd._lib import check, lib
    k, j, END = 60, 0.55, 120000
    dup_events = []
    contigs = synth.make_draft(4_600_000_000, seed=synth.SEED, dup_events=dup_events)
    parts, lens = [], []
    for c in contigs:
        cut = arks.end_cutoff(len(c), 500, END)
        if cut is None:
            continue
        parts += [c[:cut], c[len(c) - cut:]]
        lens += [cut, cut]
    lens = np.array(lens, dtype=np.uint32)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    assert int(offs[-1]) > 2**32
    data = np.concatenate(parts + [np.zeros(1, np.uint8)])
    del parts
    acc, members = _sub_draft(synth, contigs, dup_events, 30.0)
    at_runs = synth.alternating_at_runs(torch.from_numpy(np.concatenate(contigs)).cuda(), run=12)
    ox = oracle.sub_draft_index(k, contigs, members, end_length=END, site_runs=at_runs)
    n_pairs = 2_000_000
    n_sub = 0
    while sum(len(c) for c in contigs[:n_sub]) < acc:
        n_sub += 1
    genome_sub = torch.from_numpy(np.concatenate(contigs[:n_sub])).cuda()
    assert genome_sub.numel() == acc
    batch = synth.make_read_pairs(genome_sub, n_pairs, seed=4343, device="cuda")
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    ev = arks.pair_gate(reads, batch["pair_ok"])
    votes = None
    for s in range(2):
        h = C.c_void_p()
        check(lib().arks_index_build_shard(C.byref(h), k, data.ctypes.data, offs.ctypes.data, lens.ctypes.data,
                                           len(lens), s, 2, gpu), "arks_index_build_shard")
        sh = arks.ArksIndex(h, k, gpu, None)
        assert sh.kind in (1, 2)
        v = arks.map_votes_packed(sh, reads, eval_mask=ev)
        votes = v.clone() if votes is None else arks.max_votes(votes, v)
        torch.cuda.synchronize()
        sh.close()
    conreci = arks.resolve_votes(votes, reads, k, j)
    pair = arks.pairs_rule(conreci, reads, batch["pair_ok"])
    torch.cuda.synchronize()
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    want_c, want_p, _ = ox.map_pairs(a, batch["offsets"].cpu().numpy().astype(np.uint64)[:-1],
                                     batch["lens"].cpu().numpy().astype(np.uint32), j,
                                     pair_ok=batch["pair_ok"].cpu().numpy(), threads=min(64, os.cpu_count() or 1))
    assert (conreci.cpu().numpy() == want_c).all()
    assert (pair.cpu().numpy() == want_p).all()
    assert int((want_p != 0).sum()) > n_pairs // 10


def test_arks_long_human_scale_multi_k(arks, gpu, oracle):
    """BASELINE configs[4] on ONE MI355X at the draft's full size: the 3 Gbp draft indexed for k = 40, 60 and 80
    (three seed indexes resident together, as `arcs --arks -k 40,60,80` keeps them), 250-bp pseudo-linked pairs as
    long-to-linked-pe cuts them from long reads (a read's two mates are adjacent stretches, the second one
    reverse-complemented; 2 % substitutions: ONT-like), j = 0.05 (bin/arcs-make:299-313).  Every k against the
    CPU oracle on 2 M pairs drawn from a sub-draft: conreci, pair rule, all eight counters."""
    import torch
    from arcs_amd import synth
    j, L = 0.05, 250
    dup_events = []
    contigs = synth.make_draft(3_000_000_000, seed=synth.SEED, dup_events=dup_events)
    ends = _ends_of(arks, contigs)
    ixs = {k: arks.ArksIndex.build(ends, k, device=gpu) for k in (40, 60, 80)}
    del ends
    # the layout is chosen by what is free when an index is built: seed tables (46 GB each) while they fit beside
    # the build's scratch, the minimizer layout after that -- both are under test here
    assert ixs[40].kind == 2 and all(ix.kind in (1, 2) for ix in ixs.values())
    acc, members = _sub_draft(synth, contigs, dup_events, 30.0)
    n_sub = 0
    while sum(len(c) for c in contigs[:n_sub]) < acc:
        n_sub += 1
    genome_sub = torch.from_numpy(np.concatenate(contigs[:n_sub])).cuda()
    at_runs = synth.alternating_at_runs(torch.from_numpy(np.concatenate(contigs)).cuda(), run=12)
    n_pairs = 2_000_000
    batch = synth.make_read_pairs(genome_sub, n_pairs, seed=4545, device="cuda", r1_len=L, r2_len=L, frag=2 * L,
                                  sub_rate=0.02)
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    ok = batch["pair_ok"].cpu().numpy()
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    offs = batch["offsets"].cpu().numpy().astype(np.uint64)[:-1]
    lens = batch["lens"].cpu().numpy().astype(np.uint32)
    passed = 0
    for k, ix in ixs.items():
        stats = torch.zeros(8, dtype=torch.int64, device="cuda")
        conreci, pair = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], barcode_id=batch["barcode_id"],
                                              stats=stats)
        torch.cuda.synchronize()
        ox = oracle.sub_draft_index(k, contigs, members, site_runs=at_runs)
        want_c, want_p, want_st = ox.map_pairs(a, offs, lens, j, pair_ok=ok, threads=min(64, os.cpu_count() or 1))
        assert (conreci.cpu().numpy() == want_c).all(), k
        assert (pair.cpu().numpy() == want_p).all(), k
        assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}, k
        # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
        conreci, pair = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"])
        torch.cuda.synchronize()
        assert (conreci.cpu().numpy() == want_c).all() and (pair.cpu().numpy() == want_p).all(), (k, "no counters")
        passed += int((want_p != 0).sum())
        del ox
    assert passed > n_pairs          # most pairs name an end at every k with j = 0.05
    for ix in ixs.values():
        ix.close()


def test_seed_table_in_eight_shards_human_scale(arks, gpu, oracle):
    """BASELINE configs[3] at its size on ONE MI355X: the 3 Gbp draft's seed table split into 8 shards by the hash
    prefix of the m-mer (arks_index_build_seed_shard, what 8 ranks would hold one each) -- all eight resident here --,
    2 M read pairs mapped on a home shard with every seed answered by the shard that owns it (the all-to-all of
    arcs_amd.dist.exchange_seeds done by hand, as tests/test_gpu_seed_shards.py does at 0.4 Mbp).  Against the
    sub-draft oracle: conreci, pair rule, all eight counters, IndexMap; microsatellite reads included."""
    import torch
    from arcs_amd import synth
    k, j, n_ranks = 60, 0.55, 8
    dup_events = []
    contigs = synth.make_draft(3_000_000_000, seed=synth.SEED, dup_events=dup_events)
    ends = _ends_of(arks, contigs)
    shards = []
    for r in range(n_ranks):
        shards.append(arks.ArksIndex.build_seed_shard(ends, k, r, n_ranks, device=gpu))
        assert shards[-1].kind == 2 and shards[-1].seed_ranks == n_ranks
    del ends
    sizes = [sh.device_bytes for sh in shards]
    assert len(shards[0]) > 1_400_000_000
    # nobody holds the whole 43 GB table: a shard is the replicated text / fallback / minimizer table + 1/8 of the seeds
    assert max(sizes) < 14 * 2**30, sizes
    genome = torch.from_numpy(np.concatenate(contigs)).cuda()
    acc, members = _sub_draft(synth, contigs, dup_events, 30.0)
    ox = oracle.sub_draft_index(k, contigs, members, site_runs=synth.alternating_at_runs(genome, run=12))
    n_pairs = 2_000_000
    batch = synth.make_read_pairs(genome[:acc], n_pairs, seed=4646, device="cuda")
    del genome
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    ev = arks.pair_gate(reads, batch["pair_ok"])
    home = shards[3]
    counts = arks.api.seed_counts(home, reads, ev)
    seed_off = torch.zeros(reads.n_reads + 1, dtype=torch.int64, device="cuda")
    seed_off[1:] = torch.cumsum(counts.to(torch.int64), 0)
    mmer, owner = arks.api.seeds_fill(home, reads, seed_off, ev)
    per_owner = torch.bincount(owner.to(torch.int64), minlength=n_ranks).cpu().numpy()
    assert per_owner.min() > 0.8 * per_owner.mean() and per_owner.max() < 1.2 * per_owner.mean()   # the hash spreads
    answers = torch.zeros(2 * mmer.numel(), dtype=torch.int64, device="cuda")
    for r, sh in enumerate(shards):                       # the exchange, by hand: owner r answers its seeds
        sel = (owner == r).nonzero().flatten()
        answers.view(-1, 2)[sel] = arks.api.seeds_probe(sh, mmer[sel].contiguous()).view(-1, 2)
    # a foreign shard knows nothing of a seed it does not own
    foreign = arks.api.seeds_probe(shards[1], mmer[(owner == 0).nonzero().flatten()[:100000]].contiguous())
    assert int((foreign.view(-1, 2)[:, 0] != 0).sum()) == 0
    stats = torch.zeros(8, dtype=torch.int64, device="cuda")
    imap = arks.ImapAccumulator(1 << 20, device=gpu)
    conreci = arks.api.map_reads_seeded(home, reads, j, seed_off, answers, eval_mask=ev, stats=stats)
    pair = arks.pairs_rule(conreci, reads, batch["pair_ok"], batch["barcode_id"], imap)
    torch.cuda.synchronize()
    ok = batch["pair_ok"].cpu().numpy()
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    want_c, want_p, want_st = ox.map_pairs(a, batch["offsets"].cpu().numpy().astype(np.uint64)[:-1],
                                           batch["lens"].cpu().numpy().astype(np.uint32), j, pair_ok=ok,
                                           threads=min(64, os.cpu_count() or 1))
    assert (conreci.cpu().numpy() == want_c).all()
    assert (pair.cpu().numpy() == want_p).all()
    assert dict(zip(STAT_NAMES, stats.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}
    # (the instantiation without counters: what `arcs` runs without -v and what bench.py times)
    c2 = arks.api.map_reads_seeded(home, reads, j, seed_off, answers, eval_mask=ev)
    torch.cuda.synchronize()
    assert (c2.cpu().numpy() == want_c).all()
    del c2
    sel = (want_p != 0) & (ok != 0)
    key = batch["barcode_id"].cpu().numpy().astype(np.int64)[sel] * (1 << 32) + want_p[sel]
    uk, cnt = np.unique(key, return_counts=True)
    t = imap.triples()
    assert len(t) == len(uk) and (t[:, 0].astype(np.int64) * (1 << 32) + t[:, 1] == uk).all() and (t[:, 2] == cnt).all()
    assert int((want_p != 0).sum()) > n_pairs // 10
    for sh in shards:
        sh.close()
