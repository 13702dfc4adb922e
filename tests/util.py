"""helpers shared by the tests"""
import hashlib

import numpy as np


def index_digest(keys, vals):
    """order-independent digest of an exported index (same recipe as tests/golden/make_golden.py)"""
    keys = np.ascontiguousarray(keys)
    order = np.lexsort(keys.T[::-1]) if len(keys) else np.zeros(0, dtype=np.int64)
    h = hashlib.sha256()
    h.update(keys[order].tobytes())
    h.update(np.asarray(vals)[order].astype("<i4").tobytes())
    return h.hexdigest()


def oracle_pairs(oracle_mod, ox, reads, pair_ok, barcode, j):
    """chromiumRead's pair flow through the oracle: (conreci, pair, stats, triples)"""
    data = "".join(reads).encode()
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offsets = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        offsets[1:] = np.cumsum(lens[:-1])
    conreci, pair, st = ox.map_pairs(data + b"\0", offsets, lens, j,
                                     pair_ok=np.asarray(pair_ok, dtype=np.uint8))
    imap = {}
    for p, c in enumerate(pair):
        if c and pair_ok[p]:
            imap[(int(barcode[p]), int(c))] = imap.get((int(barcode[p]), int(c)), 0) + 1
    triples = sorted([b, c, n] for (b, c), n in imap.items())
    return conreci, pair, st, triples


def strip_read_num(name):
    """stripReadNum, Arcs/Arcs.cpp:243-254"""
    pos = name.rfind("/")
    if pos in (-1, 0, len(name) - 1) or not name[pos + 1].isdigit():
        return name
    return name[:pos]


def bx_of(comment):
    """the BX:Z: barcode of a FASTQ comment as chromiumRead extracts it (Arcs.cpp:1225-1251); "" = none"""
    t = comment.find("BX:Z:")
    if t < 0:
        return ""
    e = comment.find(" ", t)
    return comment[t + 5:e] if e >= 0 else comment[t + 5:]


def expected_cli_outputs(oracle_mod, G, names, cs, recs, mult, k, j, P, threads=8):
    """What `arcs --arks` must write for a draft (names, cs) and read pairs recs = [(name1, comment1, seq1,
    name2, comment2, seq2)] with the multiplicity map `mult`: the CPU oracle for the k-mer mapping
    (chromiumRead's gates and pair rule, Arcs.cpp:1185-1292), tests/graph_ref.py (G) for the graph stage.
    Returns a dict: texts of _original.gv / _pair.tsv / _main.tsv, plus the pieces the callers check."""
    ends = oracle_mod.contig_ends(cs)
    kept = [n for n, s in zip(names, cs) if len(s) >= 500]
    record = [None] + [(n, h) for n in kept for h in (True, False)]
    lengths = {n: len(s) for n, s in zip(names, cs) if len(s) >= 500}
    ox = oracle_mod.OracleIndex(k).build(ends)
    pair_ok, barcode, reads = [], [], []
    for (n1, c1, s1, n2, c2, s2) in recs:
        b1, b2 = bx_of(c1), bx_of(c2)
        ok = strip_read_num(n1) == strip_read_num(n2) and b1 != "" and b2 != "" and b1 in mult and b1 == b2
        pair_ok.append(1 if ok else 0)
        barcode.append(b1)
        reads += [s1, s2]
    data = "".join(reads).encode() + b"\0"
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    offs[1:] = np.cumsum(lens[:-1])
    conreci, pair, st = ox.map_pairs(data, offs, lens, j, pair_ok=np.array(pair_ok, dtype=np.uint8), threads=threads)
    imap = {}
    for p, c in enumerate(pair):
        if c:
            sm = imap.setdefault(barcode[p], {})
            sm[record[int(c)]] = sm.get(record[int(c)], 0) + 1
    G.add_opposite_ends(imap)
    pmap = G.pair_contigs(imap, mult, P)
    ids, edges = G.create_graph(pmap, P)
    dead = set()
    if P["max_degree"]:
        dead, edges = G.remove_degree_nodes(ids, edges, P["max_degree"])
    return {"original.gv": G.graph_text(ids, edges, dead), "pair.tsv": G.pair_text(pmap),
            "main.tsv": G.tsv_text(imap, pmap, mult, P), "imap": imap, "pmap": pmap, "ids": ids, "edges": edges,
            "dead": dead, "lengths": lengths, "stats": st, "pair": pair, "pair_ok": pair_ok,
            "build_stats": ox.stats.as_dict()}


def map_reads_both_ways(ix, reads, j):
    """ix.map_reads with the -v counters AND without: two instantiations of every map kernel -- the one without
    counters is what `arcs` runs by default and what bench.py times, and it has shortcuts of its own (absent reads
    settled per chunk, flagged reads finished in place when the found windows settle the vote).  Every oracle
    comparison of per-read results goes through both.  Returns (conreci, counters)."""
    got, gst = ix.map_reads(reads, j, want_stats=True)
    plain = ix.map_reads(reads, j)
    bad = [i for i, (a, b) in enumerate(zip(plain.tolist(), got.tolist())) if a != b]
    assert not bad, ("the kernels without counters differ from the ones with", j, bad[:5], [len(reads[i]) for i in bad[:5]])
    return got, gst
