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


def make_draft(total_bp, seed=SEED, lengths=(20000, 50000, 100000, 230000), small_frac=0.005,
               inject=True):
    """list of uint8 ASCII arrays (contigs, FASTA order)"""
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
    if inject:
        big = [j for j, c in enumerate(contigs) if len(c) >= 12000]
        n_events = max(1, total_bp // 1_000_000)
        for _ in range(n_events):
            # one 100-N run
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, len(contigs[j]) - 200))
            contigs[j][p:p + 100] = ord("N")
            # one N pair 5 bp apart (the i += k rule then skips valid windows)
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, len(contigs[j]) - 200))
            contigs[j][p] = ord("N")
            contigs[j][p + 5] = ord("N")
            # one 5-kbp segment copied into another contig (forces value 0), placed near an end so
            # that it lands inside the indexed 30-kbp region
            a, b = big[int(rng.integers(len(big)))], big[int(rng.integers(len(big)))]
            if a != b:
                pa = int(rng.integers(0, min(len(contigs[a]) - 5000, 20000)))
                pb = int(rng.integers(0, min(len(contigs[b]) - 5000, 20000)))
                contigs[b][pb:pb + 5000] = contigs[a][pa:pa + 5000]
            # one (AT)x40 microsatellite: reverse-complement palindromes at every even k
            j = big[int(rng.integers(len(big)))]
            p = int(rng.integers(0, min(len(contigs[j]) - 100, 25000)))
            contigs[j][p:p + 80] = np.frombuffer(b"AT" * 40, dtype=np.uint8)
    return contigs


def contigs_to_strings(contigs):
    return [c.tobytes().decode() for c in contigs]


def make_read_pairs(contigs, n_pairs, seed=SEED, device="cpu", r1_len=128, r2_len=151,
                    frag=350, mol_len=50000, pairs_per_mol=40, sub_rate=0.005, one_n_rate=0.01,
                    many_n_rate=0.001, unpaired_rate=0.001, chunk=2_000_000):
    """Linked read pairs sampled from the concatenated draft, generated with torch on `device`.

    Returns a dict of torch tensors on `device`:
      ascii   uint8[n_pairs * (r1_len + r2_len)]  reads back to back (R1 of pair 0, R2 of pair 0, ...)
      offsets int64[2 n_pairs + 1], lens int32[2 n_pairs]
      barcode_id int32[n_pairs]   (one barcode per two molecules)
      pair_ok uint8[n_pairs]      0 for the pairs whose mate names would not match
    """
    import torch
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    genome_np = np.concatenate(contigs)
    genome = torch.from_numpy(genome_np).to(dev)
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
    for lo in range(0, n_pairs, chunk):
        hi = min(n_pairs, lo + chunk)
        n = hi - lo
        mol = torch.arange(lo, hi, device=dev) // pairs_per_mol
        u = mol_start[mol] + torch.randint(0, max(mol_eff - frag, 1), (n,), generator=g, device=dev)
        u = torch.clamp(u, max=G - frag)
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
    return {"ascii": out, "offsets": offsets, "lens": lens, "barcode_id": barcode_id,
            "pair_ok": pair_ok}


def reads_to_strings(batch):
    """list of python str, one per read (small batches only)"""
    a = batch["ascii"].cpu().numpy()
    off = batch["offsets"].cpu().numpy()
    return [a[off[i]:off[i + 1]].tobytes().decode() for i in range(len(off) - 1)]
