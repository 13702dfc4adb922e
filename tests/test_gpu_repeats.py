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
        monkeypatch.setenv("ARKS_HEAVY_OVER", heavy_over)
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
