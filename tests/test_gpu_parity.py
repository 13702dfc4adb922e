"""Parity of the HIP path (through the C ABI) with the CPU oracle and the committed golden
fixtures.  Bit-exact: keys, values, per-read contig ends, pair results, counters, triples."""
import numpy as np
import pytest

from util import index_digest, map_reads_both_ways, oracle_pairs

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("index_layout")]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))


def test_demo_index_build(arks, gpu, demo_contigs, golden_demo_index):
    """index build on the reference demo draft: counters of ..._arks.log:53-58 and the exact
    key -> value content"""
    ends = arks.contig_ends([s for _, s in demo_contigs])
    for k, want in golden_demo_index["k"].items():
        ix = arks.ArksIndex.build(ends, int(k), device=gpu)
        got = {f: ix.build_stats[f] for f in want["stats"]}
        assert got == want["stats"], k
        assert len(ix) == want["size"]
        keys, vals = ix.export()
        assert index_digest(keys, vals) == want["digest"], k
        ix.close()


def test_mini_cases_all_outputs(arks, gpu, oracle, golden_mini):
    import torch
    cs, reads = golden_mini["contigs"], golden_mini["reads"]
    ends = arks.contig_ends(cs, golden_mini["params"]["min_size"], golden_mini["params"]["end_length"])
    pair_ok = np.array(golden_mini["pair_ok"], dtype=np.uint8)
    barcode = np.array(golden_mini["barcode_id"], dtype=np.int32)
    for name, case in golden_mini["cases"].items():
        ix = arks.ArksIndex.build(ends, case["k"], device=gpu)
        assert {f: ix.build_stats[f] for f in case["build_stats"]} == case["build_stats"], name
        keys, vals = ix.export()
        assert index_digest(keys, vals) == case["index_digest"], name
        packed = arks.PackedReads.from_ascii(reads, device=gpu)
        stats = torch.zeros(8, dtype=torch.int64, device="cuda")
        stored = torch.zeros(1, dtype=torch.int64, device="cuda")
        imap = arks.ImapAccumulator(4096, device=gpu)
        conreci, pair = arks.map_pairs_packed(
            ix, packed, case["j"], pair_ok=torch.from_numpy(pair_ok).cuda(),
            barcode_id=torch.from_numpy(barcode).cuda(), imap=imap, stats=stats, stored=stored)
        torch.cuda.synchronize()
        assert conreci.cpu().tolist() == case["conreci"], name
        assert pair.cpu().tolist() == case["pair"], name
        got = dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass",
                        "reads_fail", "windows"), stats.cpu().tolist()))
        assert got == case["map_stats"], name
        assert imap.triples().tolist() == case["triples"], name
        assert int(stored.item()) == sum(n for _, _, n in case["triples"])
        imap.close()
        ix.close()


@pytest.mark.parametrize("k", [12, 20, 21, 30, 31, 32, 33, 45, 59, 60, 61, 63, 64, 65, 80, 96])
def test_random_draft_and_reads_vs_oracle(arks, gpu, oracle, k):
    """seeded random draft with every quirk, reads from both strands with errors and Ns, through
    the host convenience entry point (arks_map_reads == bestContig on every read)"""
    from arcs_amd import synth
    contigs = synth.make_draft(60000, seed=100 + k, lengths=(4000, 9000, 2500, 12000),
                               small_frac=0.2, inject=False)
    big = [c for c in contigs if len(c) >= 2500]
    big[0][300:400] = ord("N")
    big[1][50] = ord("N"); big[1][55] = ord("N")
    big[2][200:900] = big[3][100:800]
    if k % 2 == 0:
        big[4][100:100 + 3 * k] = np.frombuffer((b"AT" * (2 * k))[:3 * k], dtype=np.uint8)
        big[5][700:700 + 2 * k] = np.frombuffer((b"AT" * (2 * k))[:2 * k], dtype=np.uint8)
    cs = synth.contigs_to_strings(contigs)
    ends = arks.contig_ends(cs, 500, 1500)
    ox = oracle.OracleIndex(k).build(ends)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
    ok, ov = ox.dump()
    gk, gv = ix.export()
    assert index_digest(gk, gv) == index_digest(ok, ov)
    rng = np.random.Generator(np.random.PCG64(k))
    reads = []
    genome = "".join(cs)
    for i in range(1500):
        L = int(rng.choice([128, 151, 250, k - 1, k, k + 1, 64 + k, 700]))
        p = int(rng.integers(0, len(genome) - L))
        r = list(genome[p:p + L])
        for q in rng.integers(0, L, size=int(rng.integers(0, 3))):
            r[q] = "ACGTNacgtn"[int(rng.integers(10))]
        r = "".join(r)
        reads.append(_rc(r) if i % 2 else r)
    reads += ["", "A", "N" * 200, ("AT" * 400)[:k + 70], genome[:k].lower()]
    for j in (0.55, 0.05, 0.0, -1.0):
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        got, gst = map_reads_both_ways(ix, reads, j)
        assert got.tolist() == want, (k, j)
        assert gst == st.as_dict(), (k, j)
    ix.close()


def test_vote_rules(arks, gpu, oracle):
    """ties -> smallest contig end, strict '>', NULL windows in the denominator, > 256 windows"""
    rng = np.random.Generator(np.random.PCG64(77))
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    ends = [rnd(2000) for _ in range(6)]
    k = 30
    ox = oracle.OracleIndex(k).build(ends)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    reads = []
    reads.append(ends[3][100:139] + ends[1][100:139])              # 10 v 10: tie -> end 2
    reads.append(ends[5][100:140] + ends[1][100:139])              # 11 v 10
    reads.append(ends[0][:60] + "N")                               # NULL window counted
    reads.append("".join(ends[i][50:50 + 45] for i in (5, 4, 3, 2, 1, 0)))   # six candidates
    reads.append("".join(ends[i % 6][10 * i:10 * i + 60] for i in range(40)))  # 2400 bp, long path
    reads.append(ends[2][:1000] + ends[1][:1000])                  # long, tie -> 2
    reads.append(ends[2][:1000] + "N" * 500 + ends[1][:1001])      # long, 2 wins by one
    for j in (0.0, 0.1, 10 / 49, 10 / 49 - 1e-12, 0.5, 0.55):
        want = [ox.best_contig(r, j) for r in reads]
        got = ix.map_reads(reads, j)
        assert got.tolist() == want, j
    assert ix.map_reads(reads, 0.0).tolist()[0] == 2
    ix.close()


def test_device_packer_matches_host_packer(arks, gpu):
    import torch
    rng = np.random.Generator(np.random.PCG64(4))
    seqs = ["", "A", "ACGT" * 8, "ACGT" * 8 + "T", "acgtnNRyx-" * 7]
    for _ in range(200):
        L = int(rng.integers(1, 500))
        seqs.append("".join(rng.choice(list("ACGTacgtNn.R"), size=L,
                                       p=np.array([20] * 8 + [1, 1, 1, 1]) / 164.0)))
    h = arks.pack_reads_host(seqs)
    d = arks.PackedReads.from_ascii(seqs, device=gpu)
    n = int(h["word_off"][-1])
    assert d.codes.cpu().numpy().view(np.uint64)[:n].tolist() == h["codes"][:n].tolist()
    assert d.nmask.cpu().numpy().view(np.uint32)[:n].tolist() == h["nmask"][:n].tolist()
    nonempty = np.array([len(s) > 0 for s in seqs])
    assert (d.read_class.cpu().numpy()[nonempty] == h["read_class"][nonempty]).all()


def test_errors(arks, gpu):
    for k, status in ((3, 1), (6, 1), (10, 1), (97, 2)):
        with pytest.raises(arks.ArksError) as e:
            arks.ArksIndex.build(["ACGT" * 50], k, device=gpu)
        assert e.value.status == status
    # an end shorter than k adds nothing (the reference prints a warning, Arcs.cpp:877-882)
    ix = arks.ArksIndex.build(["ACGT", "ACGTTGCAAGGCTTAACGGATCCATG" * 3], 30, device=gpu)
    assert ix.build_stats["short_ends"] == 1 and len(ix) > 0
    assert ix.map_reads([], 0.5).tolist() == []
    ix.close()
    empty = arks.ArksIndex.build([], 30, device=gpu)
    assert len(empty) == 0 and empty.map_reads(["ACGT" * 20], 0.5).tolist() == [0]
    empty.close()


def _unpack_key(key_bytes, k):
    s = []
    for i in range(k):
        s.append("ACGT"[(key_bytes[i // 4] >> (6 - 2 * (i % 4))) & 3])
    return "".join(s)


def test_locality_index_exceptions(arks, gpu, oracle):
    """the rare paths of the locality index: heavy minimizers (low-complexity repeats), palindromes
    (quirk keys), quirk IMAGES (a regular k-mer that spells a palindrome's damaged key) and
    duplicated segments (one minimizer, several diagonals)"""
    rng = np.random.Generator(np.random.PCG64(321))
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    for k in (30, 60, 64):
        # a palindrome P whose quirk image X' is a canonical, non-palindromic k-mer
        while True:
            h = rnd(k // 2)
            P = h + _rc(h)
            X = _unpack_key(oracle.key(P, 0, k), k)
            if X < _rc(X) and oracle.key(X, 0, k) == oracle.key(P, 0, k):
                break
        unit = rnd(37)
        rep = rnd(400)
        ends = [
            rnd(300) + P + rnd(300),                      # 1: holds the palindrome
            rnd(200) + X + rnd(250),                      # 2: holds its quirk image -> value 0
            rnd(150) + unit * 30 + rnd(150),              # 3: tandem repeat: heavy minimizers
            rnd(100) + "A" * 200 + rnd(100) + "AT" * 90 + rnd(50),   # 4: poly-A, (AT)n
            rnd(100) + rep + rnd(100),                    # 5
            rnd(250) + rep + rnd(30),                     # 6: same segment, other diagonal/end
            rnd(100) + _rc(rep) + rnd(60) + unit * 3,     # 7: and reverse-complemented
        ]
        # a second palindrome whose image is NOT in the text: regular queries for X2 must find it
        while True:
            h2 = rnd(k // 2)
            P2 = h2 + _rc(h2)
            X2 = _unpack_key(oracle.key(P2, 0, k), k)
            if X2 < _rc(X2) and oracle.key(X2, 0, k) == oracle.key(P2, 0, k):
                break
        ends.append(rnd(120) + P2 + rnd(120))             # 8
        ox = oracle.OracleIndex(k).build(ends)
        assert ox.get(oracle.key(P, 0, k)) == 0            # P (end 1) and X (end 2) share a key
        assert ox.get(oracle.key(X2, 0, k)) == 8
        ix = arks.ArksIndex.build(ends, k, device=gpu)
        assert ix.kind in (1, 2)
        assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
        assert index_digest(*ix.export()) == index_digest(*ox.dump())
        genome = "".join(ends)
        reads = [P, X, _rc(X), X2, _rc(X2), P2, rnd(20) + X2 + rnd(20), rnd(9) + _rc(X2) + rnd(31),
                 rnd(10) + P + rnd(10), unit * 6, (unit * 6)[5:], _rc(unit * 5), "A" * 151,
                 "AT" * 75, "TA" * 64, rep[:151], _rc(rep[100:251]), rep[300:] + rnd(40),
                 ends[2][100:251], ends[3][60:211], ends[3][250:401]]
        for i in range(400):
            L = int(rng.choice([128, 151, 250]))
            p = int(rng.integers(0, len(genome) - L))
            r = genome[p:p + L]
            reads.append(_rc(r) if i % 2 else r)
        for j in (0.0, 0.3, 0.55):
            st = oracle.MapStats()
            want = [ox.best_contig(r, j, st) for r in reads]
            got, gst = map_reads_both_ways(ix, reads, j)
            assert got.tolist() == want, (k, j)
            assert gst == st.as_dict(), (k, j)
        ix.close()


def test_reads_across_adjacent_text_sequences(arks, gpu, oracle):
    """contig ends whose lengths are multiples of 32 sit back to back in the packed text: a chimeric
    read that continues from one end into the next stays on ONE diagonal but collects two different
    contig ends there (per-diagonal owner bookkeeping of the tile kernel); also reads whose windows
    match on both of their diagonals (a segment duplicated inside one end)"""
    rng = np.random.Generator(np.random.PCG64(99))
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    for k in (32, 60, 64):
        seg = rnd(200)
        ends = [rnd(320), rnd(640), rnd(96) + seg + rnd(24), rnd(32) + seg + rnd(8) + seg + rnd(200),
                rnd(1024)]
        assert all(len(e) % 32 == 0 for e in ends)
        ox = oracle.OracleIndex(k).build(ends)
        ix = arks.ArksIndex.build(ends, k, device=gpu)
        assert index_digest(*ix.export()) == index_digest(*ox.dump())
        genome = "".join(ends)
        reads = []
        for b in np.cumsum([len(e) for e in ends])[:-1]:
            for left in (k - 1, k, k + 10, 75, 100, 128, 151 - k, 151 - k + 1):
                for L in (128, 151, 250):
                    if 0 < left < L and b - left >= 0 and b - left + L <= len(genome):
                        r = genome[b - left:b - left + L]
                        reads += [r, _rc(r)]
        reads += [seg[:151], _rc(seg[20:171]), seg[60:] + rnd(11), ends[3][10:161], _rc(ends[3][200:351]),
                  ends[3][232 - 75:232 + 76]]
        for j in (0.0, 0.2, 0.55):
            st = oracle.MapStats()
            want = [ox.best_contig(r, j, st) for r in reads]
            got, gst = map_reads_both_ways(ix, reads, j)
            assert got.tolist() == want, (k, j)
            assert gst == st.as_dict(), (k, j)
        ix.close()


def test_hash_kind_still_exact(arks, gpu, oracle, golden_mini, monkeypatch):
    """index_kind "hash" (arks_build_options) forces the plain hash-table index (design A): same results"""
    monkeypatch.setitem(arks.api.BUILD_DEFAULTS, "index_kind", "hash")
    cs, reads = golden_mini["contigs"], golden_mini["reads"]
    ends = arks.contig_ends(cs, golden_mini["params"]["min_size"], golden_mini["params"]["end_length"])
    case = golden_mini["cases"]["k60_j0.55"]
    ix = arks.ArksIndex.build(ends, 60, device=gpu)
    assert ix.kind == 0
    assert index_digest(*ix.export()) == case["index_digest"]
    ox = oracle.OracleIndex(60).build(ends)
    st = oracle.MapStats()
    want = [ox.best_contig(r, 0.55, st) for r in reads]
    got, gst = map_reads_both_ways(ix, reads, 0.55)
    assert got.tolist() == want and gst == st.as_dict()
    ix.close()


@pytest.mark.parametrize("k,mlen", [(24, 21), (30, 15), (60, 15), (64, 15), (96, 15), (40, 21)])
def test_minimizer_length_variants(arks, gpu, oracle, golden_mini, monkeypatch, k, mlen):
    """arks_build_options.minimizer_len forces the other minimizer length (default: 21-mers from k = 24 on, the short
    17-mers below; any value < 19 selects the short one): identical results, incl. the exception paths"""
    monkeypatch.setitem(arks.api.BUILD_DEFAULTS, "minimizer_len", mlen)
    cs, reads = golden_mini["contigs"], golden_mini["reads"]
    ends = arks.contig_ends(cs, golden_mini["params"]["min_size"], golden_mini["params"]["end_length"])
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    ox = oracle.OracleIndex(k).build(ends)
    assert {f: ix.build_stats[f] for f in ox.stats.as_dict()} == ox.stats.as_dict()
    assert index_digest(*ix.export()) == index_digest(*ox.dump())
    rng = np.random.Generator(np.random.PCG64(k))
    genome = "".join(cs)
    more = []
    for i in range(600):
        L = int(rng.choice([128, 151, 250, 700]))
        p = int(rng.integers(0, len(genome) - L))
        r = genome[p:p + L]
        more.append(_rc(r) if i % 2 else r)
    more += ["AT" * 75, "A" * 151, ("ACGT" * 40)[:151]]
    allreads = reads + more
    for j in (0.05, 0.55):
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in allreads]
        got, gst = map_reads_both_ways(ix, allreads, j)
        assert got.tolist() == want, (k, j)
        assert gst == st.as_dict(), (k, j)
    ix.close()


def test_reads_without_seed_entries_in_every_pattern(arks, gpu, oracle, index_layout):
    """The seed tile kernel over a sharded seed table settles the reads none of whose seeds has an entry before it makes
    its tiles and moves the others up into their places (a -DARKS_SKIP_DEAD_FUSED build of the library does the same
    against a whole index).  Chunks of 56 reads with live and dead reads in every arrangement -- runs, alternations,
    one dead read, dead pairs at either end, random -- with and without the -v counters, against the oracle."""
    import torch
    from arcs_amd import synth
    k, j = 60, 0.55
    contigs = synth.make_draft(600_000, seed=91, lengths=(20000, 50000, 100000, 30000))
    cs = synth.contigs_to_strings(contigs)
    ends = arks.contig_ends(cs)
    ox = oracle.OracleIndex(k).build(oracle.contig_ends(cs))
    rng = np.random.default_rng(3)

    def live(n):
        e = ends[int(rng.integers(0, len(ends)))]
        a = int(rng.integers(0, len(e) - n))
        return e[a:a + n]

    def dead(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    pats = [lambda l: (l // 2) % 2 == 0, lambda l: (l // 2) % 2 == 1, lambda l: l >= 28, lambda l: l < 28, lambda l: l != 10,
            lambda l: l >= 2, lambda l: l < 54, lambda l: l % 2 == 0, lambda l: True, lambda l: False,
            lambda l: rng.random() < 0.5, lambda l: rng.random() < 0.9, lambda l: rng.random() < 0.1]
    reads = []
    for f in pats:
        for l in range(56):
            n = (128, 151, 75, 300)[l % 4] if f is pats[-3] else (128 if l % 2 == 0 else 151)
            reads.append(live(n) if f(l) else dead(n))
    reads = reads + reads[:31]                      # a last chunk that is not full
    want = [ox.best_contig(r, j) for r in reads]
    assert sum(1 for w in want if w) > len(reads) // 3
    packed = arks.PackedReads.from_ascii(reads, device=gpu)
    ix = arks.ArksIndex.build(ends, k, device=gpu)
    st = torch.zeros(8, dtype=torch.int64, device="cuda")
    assert arks.map_reads_packed(ix, packed, j).cpu().tolist() == want
    assert arks.map_reads_packed(ix, packed, j, stats=st).cpu().tolist() == want
    if index_layout != "seeds":
        return                                      # (only the seed table is sharded)
    sh = arks.ArksIndex.build_seed_shard(ends, k, 0, 1, device=gpu)
    x = arks.SeedExchange.create_local([sh])[0]
    assert x.map_reads(packed, j).cpu().tolist() == want
    assert x.map_reads(packed, j, stats=st).cpu().tolist() == want
    x.close()


@pytest.mark.parametrize("k", [20, 24, 31])
def test_long_reads_with_more_than_32_seeds(arks, gpu, oracle, k):
    """Reads of 300-512 bases at k = 20 and 24 have a seed every 4 windows -- up to ~120 per read -- and the hot kernel's
    mask of "seeds without an entry" has 32 bits (shifts by the seed number wrapped onto other seeds' bits).
    Reads that are flagged for the medium queue (a heavy seed: a planted repeat) and hold stretches that are in no
    contig end (seeds without entries, anywhere in the read), kernels without counters: the early-settle count of the
    flagged reads is what the mask feeds.  Against the oracle, for several j."""
    rng = np.random.default_rng(1000 + k)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    rep = rnd(90)                                           # its m-mers occur more than twice: heavy seeds
    ends = []
    for e in range(6):
        s = rnd(1500) + rep + rnd(1500)
        if e % 2:
            s = s[:700] + rep + s[700:]
        ends.append(s)
    ox = oracle.OracleIndex(k).build(ends)
    ix = arks.ArksIndex.build(ends, k, device=gpu, index_kind="seeds")
    assert ix.kind == 2
    reads = []
    for i in range(600):
        L = int(rng.choice([300, 350, 400, 450, 500, 511, 512]))
        e = ends[int(rng.integers(len(ends)))]
        parts, n = [], 0
        while n < L:                                         # pieces: from an end, novel sequence, the repeat
            kind = int(rng.integers(0, 5))
            m = int(rng.integers(20, 160))
            if kind <= 2:
                a = int(rng.integers(0, len(e) - m))
                p = e[a:a + m]
            elif kind == 3:
                p = rnd(m)
            else:
                a = int(rng.integers(0, 40))
                p = rep[a:a + min(m, 50)]
            parts.append(p)
            n += len(p)
        r = "".join(parts)[:L]
        reads.append(_rc(r) if i % 2 else r)
    # ... and reads that lie in an end whole, over the planted repeat (flagged, every window found or absent)
    for i in range(200):
        e = ends[int(rng.integers(len(ends)))]
        L = int(rng.choice([300, 400, 512]))
        a = int(rng.integers(1200, 1600))
        r = list(e[a:a + L])
        if i % 3 == 0:
            r[int(rng.integers(len(r)))] = "N"
        reads.append("".join(r))
    for j in (0.0, 0.2, 0.55, 0.8):
        st = oracle.MapStats()
        want = [ox.best_contig(r, j, st) for r in reads]
        got, gst = map_reads_both_ways(ix, reads, j)
        bad = [i for i, (a, b) in enumerate(zip(got.tolist(), want)) if a != b]
        assert not bad, (k, j, bad[:5])
        assert gst == st.as_dict(), (k, j)
    assert any(w for w in want)
    ix.close()
