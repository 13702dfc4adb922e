"""The pair rule + IndexMap accumulator (arks_pairs_device, arks_imap_*; Arcs.cpp:1280-1288) on key patterns
chosen for the kernel's wave-wide grouping: every lane of a wave another key, all lanes one key, a few keys in
any order with unstored pairs between them, runs; several batches with their pair bases, growth of the table."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Reads:
    """what pairs_rule needs of a PackedReads"""
    def __init__(self, n_reads, device, torch):
        self.n_reads = n_reads
        self.device = device
        self.codes = torch.empty(1, device=f"cuda:{device}")


@pytest.mark.parametrize("pattern", ["distinct", "one_key", "few_mixed", "runs", "sorted_barcodes"])
def test_imap_accumulates_like_a_group_by(arks, gpu, pattern):
    import torch
    rng = np.random.Generator(np.random.PCG64(["distinct", "one_key", "few_mixed", "runs", "sorted_barcodes"].index(pattern) + 31))
    imap = arks.ImapAccumulator(64, device=gpu)          # small: the table has to grow on the way
    want_count, want_first, stored_total = {}, {}, 0
    for batch in range(4):
        n = int(rng.integers(1, 200_000))
        if pattern == "distinct":                        # 64 different keys in every wave
            bid = rng.permutation(n).astype(np.uint32)
            end = rng.integers(1, 1000, size=n)
        elif pattern == "one_key":
            bid = np.full(n, 7, np.uint32)
            end = np.full(n, 12)
        elif pattern == "few_mixed":                     # a handful of keys, any order
            bid = rng.integers(0, 3, size=n).astype(np.uint32)
            end = rng.integers(1, 4, size=n)
        elif pattern == "runs":
            bid = np.repeat(rng.integers(0, 50, size=n // 7 + 1), 7)[:n].astype(np.uint32)
            end = np.repeat(rng.integers(1, 9, size=n // 3 + 1), 3)[:n]
        else:                                            # linked reads: sorted by barcode, ends vary inside
            bid = np.sort(rng.integers(0, n // 80 + 1, size=n)).astype(np.uint32)
            end = rng.integers(1, 6, size=n) + 10 * (bid % 97)
        c1 = end.astype(np.int32).copy()
        c2 = c1.copy()
        miss = rng.random(n) < 0.4                       # mates that do not agree, or name no end
        c2[miss] = np.where(rng.random(miss.sum()) < 0.5, 0, c1[miss] + 1)
        ok = (rng.random(n) < 0.9).astype(np.uint8)
        conreci = np.empty(2 * n, np.int32)
        conreci[0::2], conreci[1::2] = c1, c2
        base = (batch << 24)
        imap.set_pair_base(base)
        stored = torch.zeros(1, dtype=torch.int64, device="cuda")
        pair = arks.pairs_rule(torch.from_numpy(conreci).cuda(), _Reads(2 * n, gpu, torch), torch.from_numpy(ok).cuda(),
                               torch.from_numpy(bid).cuda(), imap=imap, stored=stored)
        torch.cuda.synchronize()
        agreed = np.where((c1 != 0) & (c1 == c2), c1, 0)
        assert (pair.cpu().numpy() == agreed).all()
        keep = (agreed != 0) & (ok != 0)
        assert int(stored.item()) == int(keep.sum())
        stored_total += int(keep.sum())
        for p in np.flatnonzero(keep):
            key = (int(bid[p]), int(agreed[p]))
            want_count[key] = want_count.get(key, 0) + 1
            want_first.setdefault(key, base + int(p))
    t, first = imap.triples_ordered()
    got = {(int(a), int(b)): (int(c), int(f)) for (a, b, c), f in zip(t, first)}
    assert len(got) == len(t) == len(want_count)
    assert got == {k: (want_count[k], want_first[k]) for k in want_count}
    assert int(t[:, 2].sum()) == stored_total
    ts = imap.triples()
    assert [tuple(x) for x in ts[:, :2]] == sorted(want_count)     # arks_imap_export: by (barcode, end)
    imap.close()
