"""The medium kernel on a human-like repeat spectrum (synth.plant_human_like: families of 35 k copies of a 300-bp
SINE-like element at 5-20 % divergence, 2.5 k copies of a 6-kbp LINE-like one, satellite arrays): one read in ten
carries heavy seeds and is decided by the general kernels -- extra seeds for a diagonal, the text along it, proofs of
absence from the seed table for the windows over sequencing errors (T6e), exact keys for the rest -- against the CPU
oracle over the WHOLE draft.  The families have a fixed size, so a 100 Mbp draft shows a read the multiplicities of
the 3 Gbp one (bench.py's configs2_human_like uses the same pair for its sample parity)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu]

STAT_NAMES = ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail", "windows")


def _ends_of(arks, contigs):
    ends = []
    for c in contigs:
        cut = arks.end_cutoff(len(c))
        if cut is not None:
            ends.append(c[:cut].tobytes())
            ends.append(c[len(c) - cut:].tobytes())
    return ends


@pytest.mark.parametrize("sub_rate,heavy_over", [(0.005, None), (0.02, None), (0.005, "8")])
def test_human_like_spectrum_against_the_whole_oracle(arks, gpu, oracle, sub_rate, heavy_over, monkeypatch):
    import torch
    from arcs_amd import synth
    k, j = 60, 0.55
    if heavy_over:   # the index of rounds 1-5: seeds with 3-8 entries, their windows walk the entries
        monkeypatch.setitem(arks.api.BUILD_DEFAULTS, "heavy_over", int(heavy_over))
    sites = []
    contigs = synth.make_draft(100_000_000, seed=synth.SEED, repeats="human", repeat_sites=sites)
    assert len(sites) > 35_000
    ends = _ends_of(arks, contigs)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    assert ix.kind == 2
    ox = oracle.OracleIndex(k).build(ends)
    assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
    del ends
    genome = torch.from_numpy(np.concatenate(contigs)).cuda()
    n_pairs = 600_000
    batch = synth.make_read_pairs(genome, n_pairs, seed=synth.SEED + 780, device="cuda", sub_rate=sub_rate)
    a = np.concatenate([batch["ascii"].cpu().numpy(), np.zeros(1, np.uint8)])
    lens = batch["lens"].cpu().numpy().astype(np.uint32)
    offs = batch["offsets"].cpu().numpy().astype(np.uint64)
    ok = batch["pair_ok"].cpu().numpy()
    want_c, want_p, want_st = ox.map_pairs(a, offs[: 2 * n_pairs], lens, j, pair_ok=ok, threads=16)
    reads = arks.PackedReads.from_arrays_device(batch["ascii"], batch["offsets"], batch["lens"], device=gpu)
    got_c, got_p = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"])
    q = arks.queue_counts(ix)
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    got_c2, got_p2 = arks.map_pairs_packed(ix, reads, j, pair_ok=batch["pair_ok"], stats=st)
    torch.cuda.synchronize()
    assert q[1] > n_pairs // 20, q          # the medium kernel decided a good share of the reads
    bad = np.nonzero(got_c.cpu().numpy() != want_c)[0]
    assert len(bad) == 0, (len(bad), bad[:10].tolist(), got_c.cpu().numpy()[bad[:10]].tolist(), want_c[bad[:10]].tolist())
    assert (got_p.cpu().numpy() == want_p).all()
    assert (got_c2.cpu().numpy() == want_c).all() and (got_p2.cpu().numpy() == want_p).all()
    assert dict(zip(STAT_NAMES, st.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}
    assert int((want_c != 0).sum()) > n_pairs // 8


def test_pair_gate_inside_the_map_kernel(arks, gpu, oracle, index_layout):
    """arks_map_pairs_device (the gate of chromiumRead worked out by the seed tile kernel from pair_ok and the reads'
    classes) against arks_pair_gate_device + arks_map_reads_device and against the oracle's pair flow: gated pairs
    (pair_ok clear), pairs with a mate that checkReadSequence rejects (> 2 % N, a foreign letter), reads with a few N,
    with and without counters, the IndexMap -- on both layouts of the locality index (the minimizer layout takes the two
    launches inside the call)."""
    import torch
    from arcs_amd import synth
    from util import oracle_pairs
    k, j = 60, 0.55
    contigs = synth.make_draft(3_000_000, seed=synth.SEED + 5, repeats=True)
    ends = _ends_of(arks, contigs)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    ox = oracle.OracleIndex(k).build(ends)
    genome = torch.from_numpy(np.concatenate(contigs)).cuda()
    n_pairs = 30_000
    batch = synth.make_read_pairs(genome, n_pairs, seed=77, device="cuda", many_n_rate=0.02, one_n_rate=0.05)
    a = batch["ascii"].cpu().numpy().copy()
    offs = batch["offsets"].cpu().numpy().astype(np.int64)
    for r in range(0, 2 * n_pairs, 997):      # a letter checkReadSequence does not accept
        a[offs[r] + 7] = ord("X")
    d_ascii = torch.from_numpy(a).cuda()
    reads = arks.PackedReads.from_arrays_device(d_ascii, batch["offsets"], batch["lens"], device=gpu)
    ok = batch["pair_ok"]
    assert 0 < int(ok.sum().item()) < n_pairs
    cls = reads.read_class.cpu().numpy()
    assert (cls == 0).any() and (cls == 1).any() and (cls == 3).any()
    st_f = torch.zeros(8, dtype=torch.int64, device="cuda")
    st_g = torch.zeros(8, dtype=torch.int64, device="cuda")
    im_f = arks.ImapAccumulator(1 << 12, device=gpu)
    im_g = arks.ImapAccumulator(1 << 12, device=gpu)
    c_f, p_f = arks.map_pairs_packed(ix, reads, j, pair_ok=ok, barcode_id=batch["barcode_id"], imap=im_f, stats=st_f)
    c_g, p_g = arks.map_pairs_packed(ix, reads, j, pair_ok=ok, barcode_id=batch["barcode_id"], imap=im_g, stats=st_g,
                                     fused=False)
    c_f2, p_f2 = arks.map_pairs_packed(ix, reads, j, pair_ok=ok)              # without counters
    c_n, _ = arks.map_pairs_packed(ix, reads, j)                             # no pair_ok: every pair
    c_ng, _ = arks.map_pairs_packed(ix, reads, j, fused=False)
    torch.cuda.synchronize()
    assert (c_f == c_g).all() and (p_f == p_g).all() and (c_f2 == c_g).all() and (p_f2 == p_g).all()
    assert (c_n == c_ng).all()
    assert st_f.cpu().tolist() == st_g.cpu().tolist()
    assert (im_f.triples() == im_g.triples()).all()
    lens = batch["lens"].cpu().numpy()
    rs = [a[offs[r]:offs[r] + lens[r]].tobytes().decode() for r in range(2 * n_pairs)]
    want_c, want_p, want_st, want_t = oracle_pairs(oracle, ox, rs, ok.cpu().numpy(), batch["barcode_id"].cpu().numpy(), j)
    assert (c_f.cpu().numpy() == want_c).all() and (p_f.cpu().numpy() == want_p).all()
    assert dict(zip(STAT_NAMES, st_f.cpu().tolist())) == {f: want_st[f] for f in STAT_NAMES}
    assert im_f.triples().tolist() == want_t
    assert int((want_p != 0).sum()) > 1000
