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
    sub = O.sub_draft_index(k, contigs, members, at_runs=runs)
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
