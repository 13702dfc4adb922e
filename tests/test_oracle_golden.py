"""The CPU oracle against the golden vectors made from the reference's own encoder
(tests/golden/make_golden.py) and against the reference demo's log counters."""
import numpy as np

from util import index_digest, oracle_pairs


def test_golden_keys(oracle, golden_keys):
    assert len(golden_keys) > 100
    n_null = n_pal = 0
    for c in golden_keys:
        got = oracle.key(c["seq"], c["pos"], c["k"])
        want = bytes.fromhex(c["key"]) if c["key"] is not None else None
        assert got == want, c
        n_null += want is None
        n_pal += "palindrome" in c["note"]
    assert n_null >= 3 and n_pal >= 30


def test_survey_golden_keys(oracle):
    """the table of SURVEY.md section 8 (keys produced by the compiled reference encoder)"""
    t = [(60, "TGTATCTACAATTTTATACTCACATTTCAAATTGAAGGATAAAGAGCAAAAAGTTGAAAA",
          "ecdc43fcc744fd03e0a3022400be00"),
         (60, "A" * 60, "00" * 15), (60, "T" * 60, "00" * 15),
         (60, "AT" * 30, "33333333333333330033cc33cc33cc"),
         (60, "ACGT" * 15, "1b1b1b1b1b1b1b1b001bc6b16c1bc6"),
         (60, "CG" * 30, "666666666666666600669966996699"),
         (30, "AT" * 15, "333333330033ccc0"), (20, "AT" * 10, "3333330033"),
         (40, "AT" * 20, "33333333330033cc33cc"),
         (80, "AT" * 40, "333333333333333333330033cc33cc33cc33cc33")]
    for k, s, h in t:
        assert oracle.key(s, 0, k).hex() == h


def test_demo_index_counters(oracle, demo_contigs, golden_demo_index):
    """Examples/arks_test-demo/output/..._arks.log:53-58: Total 123190, Null 303, Recorded 118710,
    Collisions 4480, Removed 547, Unique 118334 (k=30, -z 500 -e 30000)"""
    ends = oracle.contig_ends([s for _, s in demo_contigs])
    assert len(ends) == 6
    for k, want in golden_demo_index["k"].items():
        ix = oracle.OracleIndex(int(k)).build(ends)
        assert ix.stats.as_dict() == want["stats"]
        assert len(ix) == want["size"]
        assert index_digest(*ix.dump()) == want["digest"]
    assert golden_demo_index["k"]["30"]["stats"]["total_kmers"] == 123190
    assert golden_demo_index["k"]["30"]["stats"]["unique"] == 118334


def test_mini_cases(oracle, golden_mini):
    cs, reads = golden_mini["contigs"], golden_mini["reads"]
    ends = oracle.contig_ends(cs, golden_mini["params"]["min_size"], golden_mini["params"]["end_length"])
    for name, case in golden_mini["cases"].items():
        ox = oracle.OracleIndex(case["k"]).build(ends)
        assert ox.stats.as_dict() == case["build_stats"], name
        assert index_digest(*ox.dump()) == case["index_digest"], name
        conreci, pair, st, triples = oracle_pairs(oracle, ox, reads, golden_mini["pair_ok"],
                                                  golden_mini["barcode_id"], case["j"])
        assert conreci.tolist() == case["conreci"], name
        assert pair.tolist() == case["pair"], name
        for f, v in case["map_stats"].items():
            assert st[f] == v, (name, f)
        assert triples == case["triples"], name


def test_visit_rule_closed_forms(oracle):
    """an isolated run of r invalid characters loses (k + 1 - r) mod k valid
    windows after it; two Ns 5 apart at k=60 lose 55."""
    rng = np.random.Generator(np.random.PCG64(3))
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    for k in (30, 60):
        for r in (1, 2, 3, 100, 3000):
            left, right = rnd(500), rnd(700)
            s = left + "N" * r + right
            ix = oracle.OracleIndex(k)
            n = ix.map_kmers(s, 1)
            assert n == (500 - k + 1) + (700 - k + 1) - ((k + 1 - r) % k), (k, r)
    s = rnd(500) + "N" + rnd(4) + "N" + rnd(700)
    ix = oracle.OracleIndex(60)
    assert ix.map_kmers(s, 1) == (500 - 59) + (700 - 59) - 55


def test_best_contig_rules(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    a, b = rnd(400), rnd(400)
    k = 30
    ix = oracle.OracleIndex(k).build([a, b])
    # read shorter than k: no window, returns 0 and counts as failing
    st = oracle.MapStats()
    assert ix.best_contig(a[:k - 1], 0.5, st) == 0 and st.reads_fail == 1 and st.windows == 0
    # exact tie between ends 1 and 2 -> smallest index (Q4)
    read = a[100:100 + k + 9] + b[100:100 + k + 9]
    assert ix.best_contig(read, 0.0) == 1
    # strict '>' (Q5): 10 of 48+... windows
    n_win = len(read) - k + 1
    assert ix.best_contig(read, 10 / n_win) == 0
    assert ix.best_contig(read, 10 / n_win - 1e-9) == 1
    # NULL windows count in the denominator (Q3)
    read2 = a[:60] + "N"
    st = oracle.MapStats()
    assert ix.best_contig(read2, 0.5, st) == 1  # 31 of 32 windows
    assert st.windows == 32 and st.bad == 1 and st.total_valid == 31
    # checkReadSequence (Arcs.cpp:366-389): 2 % rule and foreign characters
    assert oracle.check_read_sequence("ACGT" * 25)
    assert oracle.check_read_sequence("ACGT" * 24 + "ACNN")           # 2 / 100 == 0.02 passes
    assert not oracle.check_read_sequence("ACGT" * 24 + "ANNN")       # 3 / 100
    assert not oracle.check_read_sequence("ACGT" * 24 + "ACGR")
    assert oracle.check_read_sequence("acgtn" + "ACGT" * 30)
