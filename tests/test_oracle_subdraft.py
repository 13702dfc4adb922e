"""The sub-draft oracle used by the human-scale GPU tests and bench.py's sample parity (oracle.pyoracle.sub_draft_index
with the windows around every alternating A/T stretch of the whole draft) against the oracle over the WHOLE draft, at a
size where the whole map is cheap: reads that reach into (AT)n microsatellites get the whole draft's answers."""
import numpy as np
import torch

from arcs_amd import synth
from oracle import pyoracle as O


def _draft(seed, n_contigs=60, length=50000, sites_per_contig=6):
    rng = np.random.Generator(np.random.PCG64(seed))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = [acgt[rng.integers(0, 4, size=length)] for _ in range(n_contigs)]
    for ci, c in enumerate(contigs):
        for s in range(sites_per_contig):
            p = int(rng.integers(100, 24000)) if s % 2 == 0 else int(rng.integers(length - 24000, length - 200))
            n = int(rng.integers(20, 45))
            c[p:p + 2 * n] = np.frombuffer(b"AT" * n, dtype=np.uint8)
        if ci % 7 == 3:                       # an N pair in front of a site: the visit rule of the whole end matters
            c[50] = ord("N")
            c[55] = ord("N")
    return contigs


def test_at_runs_detection():
    g = torch.from_numpy(np.frombuffer(b"CCATATATATATATATGGATATCAATATATATATATATATATAT", dtype=np.uint8).copy())
    for chunk in (1 << 28, 5, 7, 1):
        assert synth.alternating_at_runs(g, run=14, chunk=chunk).tolist() == [[2, 16], [24, 44]]
    assert synth.alternating_at_runs(g, run=4, chunk=3).tolist() == [[2, 16], [18, 22], [24, 44]]
    assert synth.alternating_at_runs(g[:1], run=2).shape == (0, 2)


def test_sub_draft_with_at_runs_equals_whole_draft():
    k, j = 60, 0.55
    contigs = _draft(11)
    n_first = 10
    members = list(range(n_first))
    whole = O.sub_draft_index(k, contigs, range(len(contigs)))
    genome = torch.from_numpy(np.concatenate(contigs))
    runs = synth.alternating_at_runs(genome, run=12)
    assert len(runs) >= 300
    sub = O.sub_draft_index(k, contigs, members, site_runs=runs)
    bare = O.sub_draft_index(k, contigs, members)
    assert len(bare) < len(sub) < len(whole)
    batch = synth.make_read_pairs(genome[:n_first * 50000], 30000, seed=5, device="cpu")
    touching = synth.pairs_touching_microsatellite(batch).numpy()
    assert touching.sum() > 300
    a = np.concatenate([batch["ascii"].numpy(), np.zeros(1, np.uint8)])
    offs = batch["offsets"].numpy().astype(np.uint64)[:-1]
    lens = batch["lens"].numpy().astype(np.uint32)
    ok = batch["pair_ok"].numpy()
    cw, pw, sw = whole.map_pairs(a, offs, lens, j, pair_ok=ok, threads=4)
    cs, ps, ss = sub.map_pairs(a, offs, lens, j, pair_ok=ok, threads=4)
    cb, pb, sb = bare.map_pairs(a, offs, lens, j, pair_ok=ok, threads=4)
    assert (cw == cs).all() and (pw == ps).all() and sw == ss
    # and the windows are needed: without them the counters (and usually some reads) differ
    assert sb != sw


def test_map_kmers_range_pieces_give_the_whole_scan():
    """the range variant walks like the whole scan (i += k after a NULL window) whatever the range"""
    rng = np.random.Generator(np.random.PCG64(3))
    for k in (20, 31, 60):
        s = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=5000)].tobytes())
        for p in rng.integers(0, 5000, size=25):
            s[p] = ord("N")
        s[700:700 + 130] = b"N" * 130
        s[1200] = ord("N"); s[1205] = ord("N")
        s = bytes(s)
        whole = O.OracleIndex(k)
        whole.map_kmers(s, 7)
        cuts = sorted(set([0, len(s)] + [int(x) for x in rng.integers(0, len(s), size=6)]))
        pieces = O.OracleIndex(k)
        for a, b in zip(cuts[:-1], cuts[1:]):
            pieces.map_kmers_range(s, 7, a, b)
        kw, vw = whole.dump()
        kp, vp = pieces.dump()
        assert sorted(map(bytes, kw)) == sorted(map(bytes, kp)) and len(kw) == len(whole)
        assert set(vw.tolist()) == set(vp.tolist()) == {7}


def test_sub_draft_with_repeat_sites_equals_whole_draft():
    """a draft with planted repeat families (synth.plant_repeats): the sub-draft oracle that is given the windows
    around every copy answers like the oracle over the whole draft"""
    k, j = 60, 0.55
    sites, dup = [], []
    contigs = synth.make_draft(4_000_000, seed=9, dup_events=dup)
    synth.plant_repeats(contigs, 9, scale=0.03, sites=sites, sat_arrays=200)     # 3000 SINE copies, 30 LINE, 6 arrays
    n_first = 12
    members = synth.closed_contig_set(n_first, dup)
    genome = torch.from_numpy(np.concatenate(contigs))
    runs = np.concatenate([synth.alternating_at_runs(genome, run=12), synth.sites_to_runs(contigs, sites)])
    whole = O.sub_draft_index(k, contigs, range(len(contigs)))
    sub = O.sub_draft_index(k, contigs, members, site_runs=runs)
    acc = int(sum(len(c) for c in contigs[:n_first]))
    batch = synth.make_read_pairs(genome[:acc], 30000, seed=6, device="cpu")
    a = np.concatenate([batch["ascii"].numpy(), np.zeros(1, np.uint8)])
    offs = batch["offsets"].numpy().astype(np.uint64)[:-1]
    lens = batch["lens"].numpy().astype(np.uint32)
    ok = batch["pair_ok"].numpy()
    cw, pw, sw = whole.map_pairs(a, offs, lens, j, pair_ok=ok, threads=4)
    cs, ps, ss = sub.map_pairs(a, offs, lens, j, pair_ok=ok, threads=4)
    assert (cw == cs).all() and (pw == ps).all() and sw == ss
    assert sw["dups"] > 0
