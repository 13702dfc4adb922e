"""Differential test of the oracle's key function and control flow against the reference's own
Common/ReadsProcessor.cpp (oracle/_ref, built only where the upstream checkout exists)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libarks_ref.so not present (no upstream checkout on this machine)")
    return oracle


KS = [12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 26, 28, 30, 31, 32, 33, 34, 40, 45, 50, 59, 60,
      61, 62, 63, 64, 65, 72, 79, 80, 81, 95, 96]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def test_key_fuzz(ref):
    rng = np.random.Generator(np.random.PCG64(2024))
    n_win = n_pal = 0
    for k in KS:
        for trial in range(40):
            L = int(rng.integers(k, k + 220))
            alpha = "ACGT" if trial % 3 == 0 else ("ACGTacgt" if trial % 3 == 1 else "ACGTacgtNnRYx-")
            p = None
            if alpha.endswith("-"):
                p = np.array([20] * 8 + [1] * 6, dtype=float)
                p /= p.sum()
            s = "".join(rng.choice(list(alpha), size=L, p=p))
            if k % 2 == 0 and trial % 4 == 0:  # plant reverse-complement palindromes
                h = "".join(rng.choice(list("ACGT"), size=k // 2))
                pos = int(rng.integers(0, L - k + 1))
                s = s[:pos] + h + _rc(h) + s[pos + k:]
                n_pal += 1
            rk, rv = ref.ref_keys_all(s, k)
            ok, ov = ref.oracle_keys_all(s, k)
            assert (rv == ov).all(), (k, s)
            assert (rk == ok).all(), (k, s)
            n_win += len(rv)
    assert n_win > 100000 and n_pal > 100


def test_flow_on_demo_draft(ref, demo_contigs):
    ends = ref.contig_ends([s for _, s in demo_contigs])
    rng = np.random.Generator(np.random.PCG64(9))
    for k in (30, 60):
        rx = ref.RefIndex(k).build(ends)
        ox = ref.OracleIndex(k).build(ends)
        assert rx.stats() == ox.stats.as_dict()
        # reads sampled from the draft, with errors, Ns and both strands
        for _ in range(300):
            c = demo_contigs[int(rng.integers(3))][1]
            L = int(rng.choice([128, 151, 59, 250]))
            p = int(rng.integers(0, len(c) - L))
            r = list(c[p:p + L])
            for q in rng.integers(0, L, size=int(rng.integers(0, 4))):
                r[q] = "ACGTN"[int(rng.integers(5))]
            r = "".join(r)
            if rng.random() < 0.5:
                r = _rc(r)
            for j in (0.55, 0.05):
                assert rx.best_contig(r, j) == ox.best_contig(r, j)
