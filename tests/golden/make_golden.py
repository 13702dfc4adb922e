#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE's own encoder (Common/ReadsProcessor.cpp compiled
into oracle/_ref/libarks_ref.so by oracle/Makefile).  Runs only where /root/reference exists; the
JSON files it writes are committed, this script is the record of how they were made.

  python tests/golden/make_golden.py

Files:
  keys.json      packed keys (hex) of chosen windows incl. palindromes, several k  [SURVEY 8c]
  demo_index.json  index-build counters on Examples/arks_test-demo/test_scaffolds.fa for k=30
                 (== ..._arks.log:53-58 of the reference's demo output) and other k
  mini.json      a small synthetic draft + read pairs with per-read bestContig, pair results and
                 (barcode, conreci, count) triples computed through the reference encoder
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from arcs_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def fasta(path):
    seqs, name, cur = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                seqs.append((name, "".join(cur)))
            name, cur = line[1:].split()[0], []
        else:
            cur.append(line)
    if name is not None:
        seqs.append((name, "".join(cur)))
    return seqs


def ref_key_hex(seq, pos, k):
    keys, valid = O.ref_keys_all(seq, k)
    return keys[pos].tobytes().hex() if valid[pos] else None


def index_digest(ix_get_dump):
    keys, vals = ix_get_dump
    order = np.lexsort(keys.T[::-1])
    h = hashlib.sha256()
    h.update(keys[order].tobytes())
    h.update(vals[order].astype("<i4").tobytes())
    return h.hexdigest()


def ref_dump(rx, ox):
    """(keys, vals) of the reference-flow index, enumerated through the oracle's key set after
    checking both hold the same keys and values"""
    keys, vals = ox.dump()
    assert len(rx) == len(keys)
    for i in range(len(keys)):
        assert rx.get(keys[i].tobytes()) == vals[i]
    return keys, vals


def main():
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    demo = fasta(os.path.join(HERE, "arks_test-demo.test_scaffolds.fa"))
    c7771 = demo[0][1]

    # ---- keys.json --------------------------------------------------------------------------
    cases = []
    def add(k, seq, pos=0, note=""):
        cases.append({"k": k, "seq": seq, "pos": pos, "key": ref_key_hex(seq, pos, k), "note": note})
    add(60, c7771[:60], 0, "demo contig 7771[0:60], forward canonical")
    add(60, c7771[:60].lower(), 0, "lower case")
    add(60, c7771[:70], 5, "demo contig 7771[5:65]")
    add(60, "A" * 60, 0, "poly-A")
    add(60, "T" * 60, 0, "poly-T == poly-A key")
    for k in (12, 14, 16, 18, 20, 22, 26, 28, 30, 34, 40, 60, 62, 64, 80, 96):
        add(k, ("AT" * 60)[:k], 0, "(AT)n palindrome")
        add(k, ("CG" * 60)[:k], 0, "(CG)n palindrome")
    add(60, "ACGT" * 15, 0, "(ACGT)n palindrome")
    add(30, c7771[:30], 0, "k=30 forward")
    add(30, c7771[:40], 3, "k=30 reverse complement chosen")
    add(30, "N" + c7771[:29], 0, "N first -> NULL")
    add(30, c7771[:29] + "N", 0, "N last -> NULL")
    add(30, c7771[:15] + "R" + c7771[16:30], 0, "IUPAC -> NULL")
    rng = np.random.Generator(np.random.PCG64(7))
    for k in (13, 20, 21, 30, 31, 32, 33, 45, 59, 60, 61, 63, 64, 65, 79, 80, 81, 96):
        s = "".join("ACGT"[i] for i in rng.integers(0, 4, size=k + 9))
        for pos in (0, 4, 9):
            add(k, s, pos, "random")
        half = "".join("ACGT"[i] for i in rng.integers(0, 4, size=k // 2))
        if k % 2 == 0:
            rc = half[::-1].translate(str.maketrans("ACGT", "TGCA"))
            add(k, "G" + half + rc + "T", 1, "random palindrome")
    json.dump(cases, open(os.path.join(HERE, "keys.json"), "w"), indent=0)

    # ---- demo_index.json --------------------------------------------------------------------
    demo_out = {"source": "Examples/arks_test-demo/test_scaffolds.fa (-z 500 -e 30000)", "k": {}}
    ends = O.contig_ends([s for _, s in demo])
    for k in (20, 30, 31, 60, 80):
        rx = O.RefIndex(k).build(ends)
        ox = O.OracleIndex(k).build(ends)
        assert rx.stats() == ox.stats.as_dict(), (k, rx.stats(), ox.stats.as_dict())
        demo_out["k"][str(k)] = {"stats": rx.stats(), "size": len(rx),
                                 "digest": index_digest(ref_dump(rx, ox))}
    assert demo_out["k"]["30"]["stats"] == {  # ..._arks.log:53-58
        "total_kmers": 123190, "null_kmers": 303, "recorded": 118710, "collisions": 4480,
        "removed_dup": 547, "unique": 118334}
    json.dump(demo_out, open(os.path.join(HERE, "demo_index.json"), "w"), indent=1)

    # ---- mini.json ---------------------------------------------------------------------------
    contigs = synth.make_draft(40000, seed=11, lengths=(3000, 5200, 9000, 7001), small_frac=0.3,
                               inject=False)
    # hand-placed quirks (the generic injector assumes 12-kbp contigs)
    big = [c for c in contigs if len(c) >= 3000]
    big[0][100:200] = ord("N")                      # 100-N run inside a head
    big[1][50] = ord("N"); big[1][55] = ord("N")    # N pair 5 bp apart
    big[2][300:700] = big[3][200:600]               # duplicate across contigs -> value 0
    big[4][120:200] = np.frombuffer(b"AT" * 40, dtype=np.uint8)  # palindromes
    big[5][400:480] = np.frombuffer(b"AT" * 40, dtype=np.uint8)  # the same palindromes elsewhere
    big[1][-10] = ord("r")                          # IUPAC, lower case, in a tail
    big[0][1500:1510] = np.frombuffer(b"acgtacgtac", dtype=np.uint8)
    cs = synth.contigs_to_strings(contigs)
    batch = synth.make_read_pairs(contigs, 600, seed=12, mol_len=4000, pairs_per_mol=20,
                                  one_n_rate=0.05, many_n_rate=0.02, unpaired_rate=0.02)
    reads = synth.reads_to_strings(batch)
    # extra hand-made reads appended as pairs: short read, read with palindromes, read == N*
    extra = [cs[0][:29], cs[0][:29],
             big[4][100:251].tobytes().decode(), big[4][100:228].tobytes().decode(),
             "N" * 128, "N" * 151,
             cs[0][:128].lower(), synth.revcomp_ascii(np.frombuffer(cs[0][200:351].encode(), dtype=np.uint8)).tobytes().decode()]
    reads += extra
    n_pairs = len(reads) // 2
    barcode = batch["barcode_id"].numpy().tolist() + [900, 901, 902, 903]
    pair_ok = batch["pair_ok"].numpy().tolist() + [1, 1, 1, 1]
    mini = {"contigs": cs, "reads": reads, "barcode_id": barcode, "pair_ok": pair_ok,
            "params": {"min_size": 500, "end_length": 1000}, "cases": {}}
    ends = O.contig_ends(cs, 500, 1000)
    for k, j in ((30, 0.55), (60, 0.55), (60, 0.05), (20, 0.05), (80, 0.3)):
        rx = O.RefIndex(k).build(ends)
        ox = O.OracleIndex(k).build(ends)
        assert rx.stats() == ox.stats.as_dict()
        counters = np.zeros(8, dtype=np.uint64)
        conreci = np.zeros(len(reads), dtype=np.int64)
        pair = np.zeros(n_pairs, dtype=np.int64)
        imap = {}
        for p in range(n_pairs):
            r1, r2 = reads[2 * p], reads[2 * p + 1]
            c1 = c2 = 0
            if pair_ok[p] and O.check_read_sequence(r1) and O.check_read_sequence(r2):
                c1 = rx.best_contig(r1, j, counters)
                c2 = rx.best_contig(r2, j, counters)
            conreci[2 * p], conreci[2 * p + 1] = c1, c2
            if c1 != 0 and c1 == c2:
                pair[p] = c1
                if pair_ok[p]:
                    imap[(barcode[p], c1)] = imap.get((barcode[p], c1), 0) + 1
        mini["cases"][f"k{k}_j{j}"] = {
            "k": k, "j": j, "build_stats": rx.stats(), "index_size": len(rx),
            "index_digest": index_digest(ref_dump(rx, ox)),
            "conreci": conreci.tolist(), "pair": pair.tolist(),
            "map_stats": dict(zip(("total_valid", "bad", "found", "recorded", "dups", "reads_pass",
                                   "reads_fail", "windows"), (int(x) for x in counters))),
            "triples": sorted([b, c, n] for (b, c), n in imap.items())}
    json.dump(mini, open(os.path.join(HERE, "mini.json"), "w"))
    for f in ("keys.json", "demo_index.json", "mini.json"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
    for name, case in mini["cases"].items():
        print(name, case["build_stats"], case["map_stats"], "stored", sum(1 for x in case["pair"] if x),
              "triples", len(case["triples"]))


if __name__ == "__main__":
    main()
