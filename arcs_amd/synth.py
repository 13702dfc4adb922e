"""Deterministic synthetic workloads of the shape SURVEY.md section 8(d) names (there is no network
and the demo read files are absent): an i.i.d. ACGT draft with the reference's awkward cases
injected (N runs, N pairs, cross-contig duplicates, (AT)n palindromic microsatellites, contigs
below -z) and 10x-like linked read pairs (R1 128 bp forward, R2 151 bp reverse complement, 0.5 %
substitutions, a few Ns, a few unpaired names).  Used by tests and bench.py only -- data, not product.
"""
import numpy as np

SEED = 20241108
_ACGT = np.array(list(b"ACGT"), dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def revcomp_ascii(a):
    return _COMP[a[::-1]]


def plant_repeats(contigs, seed, scale=1.0, sites=None, alu_copies=100_000, l1_copies=1_000, sat_arrays=100):
    """Repeat families of a human-like draft, planted in place into `contigs` (uint8 arrays); `scale` =
    draft size / 3 Gbp scales the copy numbers.  `sites` (a list) receives (contig, start, end) of every copy.
      * a 300-bp SINE-like element, alu_copies copies, each 10-15 % diverged from the consensus: its 21-mers
        recur thousands of times (heavy seeds), its 60-mers almost never;
      * a 6-kbp LINE-like element, l1_copies copies at 3-12 % divergence, most of them 5'-truncated;
      * tandem arrays of a 171-bp satellite monomer (20-60 kbp each), monomers 1-4 % diverged: windows that
        recur inside an array and between arrays (value 0), second and third diagonals a few bases apart."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED5EED))
    big = np.array([j for j, c in enumerate(contigs) if len(c) >= 12000], dtype=np.int64)
    if len(big) == 0:
        return

    def mutated(cons, n, dlo, dhi):
        d = rng.uniform(dlo, dhi, size=(n, 1))
        out = np.broadcast_to(cons, (n, len(cons))).copy()
        hit = rng.random(out.shape) < d
        out[hit] = _ACGT[rng.integers(0, 4, size=int(hit.sum()), dtype=np.uint8)]
        return out

    def place(block, lens=None):
        n = len(block)
        cj = big[rng.integers(0, len(big), size=n)]
        for i in range(n):
            L = int(lens[i]) if lens is not None else block.shape[1]
            c = contigs[int(cj[i])]
            p = int(rng.integers(0, len(c) - L))
            c[p:p + L] = block[i][block.shape[1] - L:] if lens is not None else block[i]
            if sites is not None:
                sites.append((int(cj[i]), p, p + L))

    n_alu = max(1, int(alu_copies * scale))
    alu = _ACGT[rng.integers(0, 4, size=300, dtype=np.uint8)]
    for lo in range(0, n_alu, 20000):
        place(mutated(alu, min(20000, n_alu - lo), 0.10, 0.15))
    n_l1 = max(1, int(l1_copies * scale))
    l1 = _ACGT[rng.integers(0, 4, size=6000, dtype=np.uint8)]
    place(mutated(l1, n_l1, 0.03, 0.12), lens=np.where(rng.random(n_l1) < 0.7, rng.integers(500, 6000, size=n_l1), 6000))
    mono = _ACGT[rng.integers(0, 4, size=171, dtype=np.uint8)]
    for _ in range(max(1, int(sat_arrays * scale))):
        j = int(big[rng.integers(0, len(big))])
        c = contigs[j]
        L = int(min(rng.integers(20000, 60000), len(c) - 200))
        p = int(rng.integers(0, len(c) - L))
        arr = mutated(mono, L // 171 + 1, 0.01, 0.04).reshape(-1)[:L]
        c[p:p + L] = arr
        if sites is not None:
            sites.append((j, p, p + L))


def plant_human_like(contigs, seed, scale=1.0, sites=None, sine_frac=0.105, line_frac=0.10,
                     sine_copies=35_000, line_copies=2_500, sat_arrays=100):
    """A human-like repeat SPECTRUM planted in place into `contigs` (hg38 is ~10 % Alu in 1.2 M
    copies and ~17 % L1; `plant_repeats` above is 1.3 % of the draft).  Families, not one element: what a read sees
    is the copy number and divergence WITHIN a family, so a family has a fixed size and the NUMBER of families
    scales with the draft (`scale` = draft size / 3 Gbp) -- a 100 Mbp draft (one family of each class) shows a read
    the same multiplicities as the 3 Gbp one (30 of each), which is what lets a small whole-draft oracle stand
    for the big draft's repeat handling.
      * SINE-like: families of `sine_copies` copies of a 300-bp subfamily consensus (a master consensus changed at
        2-8 % of its bases, as AluJ / S / Y are), every copy 5-20 % diverged from its subfamily consensus:
        sine_frac of the draft (3 Gbp: 30 families, 1.05 M copies, 315 Mbp);
      * LINE-like: families of `line_copies` copies of a 6-kbp subfamily consensus, 3-15 % diverged, 70 % of them
        5'-truncated to 0.5-6 kbp: line_frac of the draft (3 Gbp: 30 families, 75 k copies, ~300 Mbp);
      * the satellite arrays of plant_repeats.
    Young copies (5-8 %) share most 21-mers AND many 60-mers with their siblings (heavy seeds, value-0 keys), old
    ones (15-20 %) neither.  `sites` (a list) receives (contig, start, end) of every copy."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x48554D41))
    big = np.array([j for j, c in enumerate(contigs) if len(c) >= 12000], dtype=np.int64)
    if len(big) == 0:
        return
    total = sum(len(c) for c in contigs)

    def mutate_rows(cons, n, dlo, dhi):
        d = rng.uniform(dlo, dhi, size=(n, 1))
        out = np.broadcast_to(cons, (n, len(cons))).copy()
        hit = rng.random(out.shape) < d
        out[hit] = _ACGT[rng.integers(0, 4, size=int(hit.sum()), dtype=np.uint8)]
        return out

    def place(block, lens=None):
        n = len(block)
        cj = big[rng.integers(0, len(big), size=n)]
        width = block.shape[1]
        u = rng.random(n)
        for i in range(n):
            L = int(lens[i]) if lens is not None else width
            c = contigs[int(cj[i])]
            p = int(u[i] * (len(c) - L))
            c[p:p + L] = block[i][width - L:]
            if sites is not None:
                sites.append((int(cj[i]), p, p + L))

    def family_class(master_len, copies, frac, dlo, dhi, trunc):
        n_fam = max(1, int(round(frac * total / (copies * master_len * (0.3 + 0.7 * 0.54 if trunc else 1.0)))))
        master = _ACGT[rng.integers(0, 4, size=master_len, dtype=np.uint8)]
        for _ in range(n_fam):
            sub = mutate_rows(master, 1, 0.02, 0.08)[0]
            for lo in range(0, copies, 5000):
                n = min(5000, copies - lo)
                lens = None
                if trunc:
                    lens = np.where(rng.random(n) < 0.7, rng.integers(500, master_len, size=n), master_len)
                place(mutate_rows(sub, n, dlo, dhi), lens)
        return n_fam

    n_sine = family_class(300, sine_copies, sine_frac, 0.05, 0.20, False)
    n_line = family_class(6000, line_copies, line_frac, 0.03, 0.15, True)
    mono = _ACGT[rng.integers(0, 4, size=171, dtype=np.uint8)]
    for _ in range(max(1, int(sat_arrays * scale))):
        j = int(big[rng.integers(0, len(big))])
        c = contigs[j]
        L = int(min(rng.integers(20000, 60000), len(c) - 200))
        p = int(rng.integers(0, len(c) - L))
        c[p:p + L] = mutate_rows(mono, L // 171 + 1, 0.01, 0.04).reshape(-1)[:L]
        if sites is not None:
            sites.append((j, p, p + L))
    return n_sine, n_line


def sites_to_runs(contigs, sites):
    """(contig, start, end) -> int64[n, 2] intervals of the concatenated draft (what sub_draft_index takes)"""
    lens = np.fromiter((len(c) for c in contigs), dtype=np.int64, count=len(contigs))
    cstart = np.zeros(len(contigs) + 1, dtype=np.int64)
    np.cumsum(lens, out=cstart[1:])
    if not sites:
        return np.zeros((0, 2), dtype=np.int64)
    a = np.asarray(sites, dtype=np.int64)
    return np.stack([cstart[a[:, 0]] + a[:, 1], cstart[a[:, 0]] + a[:, 2]], axis=1)


def make_draft(total_bp, seed=SEED, lengths=(20000, 50000, 100000, 230000), small_frac=0.005,
               inject=True, dup_events=None, touched=None, repeats=False, repeat_sites=None):
    """list of uint8 ASCII arrays (contigs, FASTA order).  dup_events (a list) receives the
    (source contig, destination contig) of every copied segment, so that a caller can find the
    contigs that share k-mers with a given set (closed_contig_set).  repeats=True plants human-like
    repeat families (plant_repeats; copy numbers scaled by total_bp / 3 Gbp; repeats="human": the human-like
    spectrum of plant_human_like, ~20 % of the draft) before the other quirks are injected; repeat_sites (a list) receives their (contig, start, end)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    contigs = []
    acc = 0
    i = 0
    while acc < total_bp:
        L = int(lengths[i % len(lengths)])
        L = min(L, max(total_bp - acc, 600))
        contigs.append(_ACGT[rng.integers(0, 4, size=L, dtype=np.uint8)])
        acc += L
        i += 1
        if rng.random() < small_frac:  # a contig shorter than -z 500: skipped by the index
            contigs.append(_ACGT[rng.integers(0, 4, size=int(rng.integers(50, 499)), dtype=np.uint8)])
    if repeats == "human":
        plant_human_like(contigs, seed, scale=total_bp / 3e9, sites=repeat_sites)
    elif repeats:
        plant_repeats(contigs, seed, scale=total_bp / 3e9, sites=repeat_sites)
    if inject:
        big = [j for j, c in enumerate(contigs) if len(c) >= 12000]
        n_events = max(1, total_bp // 1_000_000)
        for _ in range(n_events):
            # one 100-N run
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, len(contigs[j]) - 200))
            contigs[j][p:p + 100] = ord("N")
            if touched is not None:
                touched.add(j)
            # one N pair 5 bp apart (the i += k rule then skips valid windows)
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, len(contigs[j]) - 200))
            contigs[j][p] = ord("N")
            contigs[j][p + 5] = ord("N")
            if touched is not None:
                touched.add(j)
            # one 5-kbp segment copied into another contig (forces value 0), placed near an end so
            # that it lands inside the indexed 30-kbp region
            a, b = big[int(rng.integers(len(big)))], big[int(rng.integers(len(big)))]
            if a != b:
                pa = int(rng.integers(0, min(len(contigs[a]) - 5000, 20000)))
                pb = int(rng.integers(0, min(len(contigs[b]) - 5000, 20000)))
                contigs[b][pb:pb + 5000] = contigs[a][pa:pa + 5000]
                if dup_events is not None:
                    dup_events.append((a, b))
                if touched is not None:
                    touched.update((a, b))
            # one (AT)x40 microsatellite: reverse-complement palindromes at every even k
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, min(len(contigs[j]) - 100, 25000)))
            contigs[j][p:p + 80] = np.frombuffer(b"AT" * 40, dtype=np.uint8)
            if touched is not None:
                touched.add(j)
    return contigs


def closed_contig_set(n_first, dup_events):
    """sorted indices of the first n_first contigs plus every contig connected to one of them through
    copied segments (transitively): a k-mer of these contigs occurs in no contig outside the set,
    except by chance or inside the (AT)n microsatellites, so an index of the set's ends gives the
    whole index's answers for reads drawn from the first n_first contigs (which is what a CPU check
    of a human-scale index needs: the oracle cannot hold 1.4 G keys in a test's time)."""
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in dup_events:
        parent[find(a)] = find(b)
    roots = {find(i) for i in range(n_first)}
    extra = {x for x in list(parent) if find(x) in roots}
    return sorted(set(range(n_first)) | extra)


def contigs_to_strings(contigs):
    return [c.tobytes().decode() for c in contigs]


READ_LENS = (128, 151)          # R1, R2 of the synthetic 10x-like pair (make_read_pairs' defaults)


def make_read_pairs(contigs, n_pairs, seed=SEED, device="cpu", r1_len=128, r2_len=151,
                    frag=350, mol_len=50000, pairs_per_mol=40, sub_rate=0.005, one_n_rate=0.01,
                    many_n_rate=0.001, unpaired_rate=0.001, chunk=2_000_000, want_origin=False):
    """Linked read pairs sampled from the concatenated draft (a list of contigs, or the uint8
    tensor of their concatenation), generated with torch on `device`.

    Returns a dict of torch tensors on `device`:
      ascii   uint8[n_pairs * (r1_len + r2_len)]  reads back to back (R1 of pair 0, R2 of pair 0, ...)
      offsets int64[2 n_pairs + 1], lens int32[2 n_pairs]
      barcode_id int32[n_pairs]   (one barcode per two molecules)
      pair_ok uint8[n_pairs]      0 for the pairs whose mate names would not match
      origin  int64[n_pairs]      (want_origin) position of R1's first base in the concatenated draft;
                                  R2 is the reverse complement of [origin + frag - r2_len, origin + frag)
    """
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if isinstance(contigs, torch.Tensor):      # the concatenated draft, already where it is needed
        genome = contigs.to(dev)
    else:
        genome = torch.from_numpy(np.concatenate(contigs)).to(dev)
    G = genome.numel()
    comp = torch.from_numpy(_COMP).to(dev)
    n_mol = (n_pairs + pairs_per_mol - 1) // pairs_per_mol
    span = max(G - mol_len, 1)
    mol_start = torch.randint(0, span, (n_mol,), generator=g, device=dev)
    out = torch.empty(n_pairs * (r1_len + r2_len), dtype=torch.uint8, device=dev)
    acgt = torch.from_numpy(_ACGT).to(dev)
    ar1 = torch.arange(r1_len, device=dev)
    ar2 = torch.arange(r2_len, device=dev)
    mol_eff = min(mol_len, G)
    origin = torch.empty(n_pairs, dtype=torch.int64, device=dev) if want_origin else None
    for lo in range(0, n_pairs, chunk):
        hi = min(n_pairs, lo + chunk)
        n = hi - lo
        mol = torch.arange(lo, hi, device=dev) // pairs_per_mol
        u = mol_start[mol] + torch.randint(0, max(mol_eff - frag, 1), (n,), generator=g, device=dev)
        u = torch.clamp(u, max=G - frag)
        if origin is not None:
            origin[lo:hi] = u
        r1 = genome[u[:, None] + ar1[None, :]]
        # R2: reverse complement of the r2_len bases that end at u + frag
        r2 = comp[genome[(u + frag - 1)[:, None] - ar2[None, :]].long()]
        both = torch.cat([r1, r2], dim=1)  # n x (r1_len + r2_len)
        L = r1_len + r2_len
        # substitutions
        sub = torch.rand((n, L), generator=g, device=dev) < sub_rate
        rnd = acgt[torch.randint(0, 4, (n, L), generator=g, device=dev)]
        both = torch.where(sub, rnd, both)
        # 1 % of reads get one N ; 0.1 % of reads get 4..6 Ns (those fail the 2 % rule)
        for (off, rl) in ((0, r1_len), (r1_len, r2_len)):
            sel = torch.rand((n,), generator=g, device=dev)
            one = sel < one_n_rate
            many = (sel >= one_n_rate) & (sel < one_n_rate + many_n_rate)
            pos = torch.randint(0, rl, (n, 6), generator=g, device=dev) + off
            rows = torch.arange(n, device=dev)
            r_one = rows[one]
            both[r_one, pos[one, 0]] = ord("N")
            r_many = rows[many]
            for t in range(6):
                both[r_many, pos[many, t]] = ord("N")
        out[lo * L:hi * L] = both.reshape(-1)
    lens = torch.empty(2 * n_pairs, dtype=torch.int32, device=dev)
    lens[0::2] = r1_len
    lens[1::2] = r2_len
    offsets = torch.zeros(2 * n_pairs + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(lens.to(torch.int64), 0)
    barcode_id = (torch.arange(n_pairs, device=dev) // (2 * pairs_per_mol)).to(torch.int32)
    pair_ok = (torch.rand((n_pairs,), generator=g, device=dev) >= unpaired_rate).to(torch.uint8)
    res = {"ascii": out, "offsets": offsets, "lens": lens, "barcode_id": barcode_id, "pair_ok": pair_ok}
    if origin is not None:
        res["origin"] = origin
    return res


def pairs_touching_microsatellite(batch, r1_len=128, r2_len=151, run=14, chunk=500_000):
    """bool[n_pairs]: a mate holds `run` alternating A/T bases, i.e. reaches >= run bases into one of
    the injected (AT)n microsatellites.  K-mers made of a short flank plus (AT)n collide by chance
    between microsatellite sites, so their index value depends on the WHOLE draft; every other k-mer of
    a read depends on its own contig and the contigs it shares copied segments with
    (closed_contig_set).  A CPU check against a sub-draft leaves these pairs out."""
    import torch
    L = r1_len + r2_len
    a = batch["ascii"]
    n = a.numel() // L
    out = torch.zeros(n, dtype=torch.bool, device=a.device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        x = a[lo * L:hi * L].reshape(hi - lo, L)
        isa, ist = x == ord("A"), x == ord("T")
        alt = (isa[:, :-1] & ist[:, 1:]) | (ist[:, :-1] & isa[:, 1:])     # positions i with x[i]x[i+1] in {AT, TA}
        alt[:, r1_len - 1] = False                                         # not across the mates
        c = torch.cumsum(alt.to(torch.int16), 1)
        z = torch.zeros((hi - lo, 1), dtype=torch.int16, device=a.device)
        c = torch.cat([z, c], 1)
        w = run - 1
        out[lo:hi] = ((c[:, w:] - c[:, :-w]) == w).any(1)
    return out


def alternating_at_runs(genome, run=14, chunk=1 << 28):
    """int64[n, 2] numpy array of the maximal stretches [start, end) of `genome` (a uint8 torch tensor of
    ASCII bases, e.g. the concatenated draft; any device) in which A and T alternate over at least `run`
    bases: the injected (AT)n microsatellites, their copies inside duplicated segments, and the few
    stretches that arise by chance.  A k-mer that holds such a stretch recurs between sites all over a
    draft, so its index value is a function of the WHOLE draft: a sub-draft oracle is given the windows
    around every one of these stretches (oracle.pyoracle.sub_draft_index)."""
    import torch
    G = int(genome.numel())
    starts, ends = [], []
    for lo in range(0, max(G - 1, 0), chunk):
        hi = min(G - 1, lo + chunk)                       # alt[i] for i in [lo, hi): bases i, i + 1
        a = max(lo - 1, 0)
        b = min(hi + 1, G - 1)
        x = genome[a:b + 1]
        isa, ist = x == ord("A"), x == ord("T")
        alt = (isa[:-1] & ist[1:]) | (ist[:-1] & isa[1:])  # alt[i - a] for i in [a, b)
        cur = alt[lo - a:hi - a]
        prev = alt[lo - a - 1:hi - a - 1] if lo > a else torch.cat([alt.new_zeros(1), alt[:hi - a - 1]])
        nxt = alt[lo - a + 1:hi - a + 1] if b > hi else torch.cat([alt[lo - a + 1:hi - a], alt.new_zeros(1)])
        starts.append(torch.nonzero(cur & ~prev).flatten() + lo)
        ends.append(torch.nonzero(cur & ~nxt).flatten() + lo)
    if not starts:
        return np.zeros((0, 2), dtype=np.int64)
    s = torch.cat(starts).cpu().numpy()
    e = torch.cat(ends).cpu().numpy()
    assert len(s) == len(e)
    keep = (e - s + 2) >= run                              # alt[s..e] true: bases s .. e + 1
    return np.stack([s[keep], e[keep] + 2], axis=1).astype(np.int64)


def fastq_bytes(batch, first_pair=0, r1_len=128, r2_len=151):
    """The batch of make_read_pairs as interleaved 4-line FASTQ text (uint8 array), vectorised: record p is
    `@p<9 digits>/1 BX:Z:<16-base barcode>-1`, the bases, `+`, qualities -- then its mate with `/2`.  A pair whose
    pair_ok is 0 gets a mate with another name (what the flag stands for); the barcode spells barcode_id in base 4."""
    a = batch["ascii"].cpu().numpy()
    n = a.size // (r1_len + r2_len)
    ok = batch["pair_ok"].cpu().numpy().astype(bool)
    bid = batch["barcode_id"].cpu().numpy().astype(np.int64)
    idx = np.arange(first_pair, first_pair + n, dtype=np.int64)
    both = a.reshape(n, r1_len + r2_len)

    def digits(v, width, base, alphabet):
        out = np.empty((len(v), width), dtype=np.uint8)
        for i in range(width - 1, -1, -1):
            out[:, i] = alphabet[v % base]
            v = v // base
        return out

    dec = np.frombuffer(b"0123456789", dtype=np.uint8)
    name = digits(idx.copy(), 9, 10, dec)
    bc = digits(bid.copy(), 16, 4, _ACGT)
    recs = []
    for mate, (lo, L) in enumerate(((0, r1_len), (r1_len, r2_len))):
        nm = name if mate == 0 else np.where(ok[:, None], name, digits(idx + 500_000_000, 9, 10, dec))
        head = np.concatenate([np.full((n, 2), ord("@"), np.uint8), nm,
                               np.broadcast_to(np.frombuffer(b"/%d BX:Z:" % (mate + 1), dtype=np.uint8), (n, 8)), bc,
                               np.broadcast_to(np.frombuffer(b"-1\n", dtype=np.uint8), (n, 3))], axis=1)
        head[:, 1] = ord("p")
        rec = np.concatenate([head, both[:, lo:lo + L], np.broadcast_to(np.frombuffer(b"\n+\n", dtype=np.uint8), (n, 3)),
                              np.full((n, L), ord("F"), np.uint8), np.full((n, 1), ord("\n"), np.uint8)], axis=1)
        recs.append(rec)
    return np.concatenate(recs, axis=1).reshape(-1)


def write_gz_members(path, data, threads=16, level=1, member_bytes=32 << 20):
    """`data` (bytes-like) as a .gz of back-to-back members compressed on `threads` threads (a valid gzip file:
    zlib's gzread, kseq and every other reader take the members as one stream)"""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(data)
    cuts = list(range(0, len(mv), member_bytes)) + [len(mv)]

    def member(i):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(mv[cuts[i]:cuts[i + 1]]) + c.flush()

    with ThreadPoolExecutor(threads) as ex, open(path, "wb") as f:
        for blob in ex.map(member, range(len(cuts) - 1)):
            f.write(blob)


def reads_to_strings(batch):
    """list of python str, one per read (small batches only)"""
    a = batch["ascii"].cpu().numpy()
    off = batch["offsets"].cpu().numpy()
    return [a[off[i]:off[i + 1]].tobytes().decode() for i in range(len(off) - 1)]
