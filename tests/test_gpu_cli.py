"""End-to-end check of the host front end `arcs --arks` (arcs_amd/host/arcs.cpp over libarks_hip.so):
FASTA draft + interleaved FASTQ.gz with BX:Z: barcodes in, <base>_original.gv / _main.tsv /
_pair.tsv / barcode counts / .dist.gv out, compared with the Python restatement of the same flow
(CPU oracle for the k-mer mapping, tests/graph_ref.py for the graph stage)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import graph_ref as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_read_num(name):
    pos = name.rfind("/")
    if pos in (-1, 0, len(name) - 1) or not name[pos + 1].isdigit():
        return name
    return name[:pos]


@pytest.mark.parametrize("use_mult_file,k,extra", [(False, 60, []), (True, 40, ["-d", "2", "-l", "1"])])
def test_arcs_cli_end_to_end(arks, gpu, oracle, tmp_path, use_mult_file, k, extra):
    from arcs_amd import build as b, synth
    exe = b.build_host()
    rng = np.random.Generator(np.random.PCG64(5 + k))
    contigs = synth.make_draft(400_000, seed=40 + k, lengths=(60000, 20000, 90000, 45000), small_frac=0.5)
    cs = synth.contigs_to_strings(contigs)
    names = [str(i + 1) for i in range(len(cs))]
    names[2] = "scaf_three"
    fa = tmp_path / "draft.fa"
    with open(fa, "w") as f:
        for n, s in zip(names, cs):
            f.write(f">{n} some description\n")
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
    n_pairs = 6000
    batch = synth.make_read_pairs(contigs, n_pairs, seed=41 + k, mol_len=30000, pairs_per_mol=30,
                                  one_n_rate=0.03, many_n_rate=0.01)
    reads = synth.reads_to_strings(batch)
    bid = batch["barcode_id"].numpy()
    recs = []   # (name1, comment1, seq1, name2, comment2, seq2)
    for p in range(n_pairs):
        bc = "".join("ACGT"[(int(bid[p]) >> (2 * t)) & 3] for t in range(12)) + "-1"
        n1, n2 = f"read{p}/1", f"read{p}/2"
        c1 = c2 = f"BX:Z:{bc}"
        u = rng.random()
        if u < 0.01:
            n2 = f"other{p}/2"                       # unpaired names
        elif u < 0.02:
            c1 = "RG:Z:x"                            # no barcode on mate 1
        elif u < 0.03:
            c2 = f"BX:Z:{bc[:-1]}2"                  # barcodes differ
        elif u < 0.04:
            c1 = c2 = f"XY:i:1 BX:Z:{bc} QT:Z:FFF"   # tag in the middle of the comment
        elif u < 0.05:
            n1, n2 = f"read{p}", f"read{p}"           # no /1 /2
        recs.append((n1, c1, reads[2 * p], n2, c2, reads[2 * p + 1]))
    fq = tmp_path / "reads.fq.gz"
    with gzip.open(fq, "wt") as f:
        for (n1, c1, s1, n2, c2, s2) in recs:
            f.write(f"@{n1} {c1}\n{s1}\n+\n{'F' * len(s1)}\n@{n2} {c2}\n{s2}\n+\n{'F' * len(s2)}\n")
    # ---- barcode multiplicities as the reference derives them (reads per barcode, Arcs.cpp:481-547)
    def bx(c):
        t = c.find("BX:Z:")
        if t < 0:
            return ""
        e = c.find(" ", t)
        return c[t + 5:e] if e >= 0 else c[t + 5:]
    mult = {}
    for (n1, c1, s1, n2, c2, s2) in recs:
        for c in (c1, c2):
            if "BX:Z:" in c:
                mult[bx(c)] = mult.get(bx(c), 0) + 1
    args = [exe, "--arks", "-v", "-f", str(fa), "-c", "3", "-m", "8-10000", "-r", "0.05", "-e", "30000",
            "-z", "500", "-j", "0.55", "-k", str(k), "-t", "8", "--gap", "100", "-b", str(tmp_path / "out"),
            "-P", "--barcode-counts", str(tmp_path / "counts"), "--batch-pairs", "1700"] + extra
    if use_mult_file:
        # a multiplicity file that lacks some barcodes: those pairs are "invalid barcode"
        keep = {b: m for i, (b, m) in enumerate(sorted(mult.items())) if i % 7}
        mf = tmp_path / "mult.csv"
        mf.write_text("".join(f"{b},{m}\n" for b, m in keep.items()))
        mult = keep
        args += ["-u", str(mf)]
    args.append(str(fq))
    res = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    # ---- the same flow in Python ------------------------------------------------------------
    ends = oracle.contig_ends(cs)
    kept = [n for n, s in zip(names, cs) if len(s) >= 500]
    record = [None] + [(n, h) for n in kept for h in (True, False)]
    lengths = {n: len(s) for n, s in zip(names, cs) if len(s) >= 500}
    ox = oracle.OracleIndex(k).build(ends)
    pair_ok, barcode = [], []
    for (n1, c1, s1, n2, c2, s2) in recs:
        b1, b2 = bx(c1), bx(c2)
        ok = _strip_read_num(n1) == _strip_read_num(n2) and b1 != "" and b2 != "" and b1 in mult and b1 == b2
        pair_ok.append(1 if ok else 0)
        barcode.append(b1)
    data = "".join(reads).encode() + b"\0"
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    offs[1:] = np.cumsum(lens[:-1])
    conreci, pair, st = ox.map_pairs(data, offs, lens, 0.55, pair_ok=np.array(pair_ok, dtype=np.uint8))
    imap = {}
    for p, c in enumerate(pair):
        if c:
            sm = imap.setdefault(barcode[p], {})
            sm[record[int(c)]] = sm.get(record[int(c)], 0) + 1
    G.add_opposite_ends(imap)
    P = {"min_reads": 3, "min_links": 0, "min_mult": 8, "max_mult": 10000, "max_degree": 0,
         "error_percent": 0.05, "gap": 100}
    for i in range(0, len(extra), 2):
        P[{"-d": "max_degree", "-l": "min_links"}[extra[i]]] = int(extra[i + 1])
    pmap = G.pair_contigs(imap, mult, P)
    ids, edges = G.create_graph(pmap, P)
    dead = set()
    if P["max_degree"]:
        dead, edges = G.remove_degree_nodes(ids, edges, P["max_degree"])
    assert len(edges) >= 2, "the synthetic data should link some contigs"
    base = str(tmp_path / "out")
    assert open(base + "_original.gv").read() == G.graph_text(ids, edges, dead)
    assert open(base + "_pair.tsv").read() == G.pair_text(pmap)
    assert open(base + "_main.tsv").read() == G.tsv_text(imap, pmap, mult, P)
    assert open(str(tmp_path / "counts.tsv")).read() == G.counts_text(mult)
    # .dist.gv as TEXT: the vertex order is the iteration order of the ContigToLength unordered_map, which
    # tests/graph_ref.py takes from the local libstdc++'s own container (tests/umap_order.cpp)
    live = [(u, v, o, w) for (u, v, o, w) in edges if u not in dead and v not in dead]
    assert open(base + ".dist.gv").read() == G.dist_graph_text(lengths, ids, live, 100)
    # ---- the -v counters (Arcs.cpp:1107-1128, 1321-1340) -----------------------------------------
    out = res.stdout
    bs = ox.stats.as_dict()
    for label, key in (("Total number of Kmers: ", "total_kmers"), ("Number Null Kmers: ", "null_kmers"),
                       ("Number Kmers Recorded: ", "recorded"), ("Number Kmer Collisions: ", "collisions"),
                       ("Number Times Kmers Removed (since duplicate in different contig): ", "removed_dup"),
                       ("Number of unique kmers (only one contig): ", "unique")):
        assert f"{label} {bs[key]}\n" in out, label
    stored = int((pair != 0).sum())
    gated = int(sum(pair_ok))
    assert f"Stored read pairs: {stored}\n" in out
    assert f"Skipped reads pairs without a good contig: {gated - stored}\n" in out
    assert f"Total valid kmers: {st['total_valid']}\n" in out
    assert f"Number of kmers found in ContigKmap: {st['found']}\n" in out
    assert f"Number of reads passing jaccard threshold: {st['reads_pass']}\n" in out
    n_unpaired = sum(1 for r in recs if _strip_read_num(r[0]) != _strip_read_num(r[3]))
    assert out.count("File contains unpaired reads:") == n_unpaired
    assert f"Skipped unpaired reads: {n_unpaired}\n" in out
    # ---- the same reads split over three files (gz, plain, empty), more ingest threads: files are
    #      parsed in parallel, every output and the cumulative counters stay the same -----------------
    cut = (n_pairs // 3) | 1
    parts = [recs[:cut], recs[cut:], []]
    paths = [tmp_path / "part0.fq.gz", tmp_path / "part1.fq", tmp_path / "part2.fastq"]
    for path, part in zip(paths, parts):
        opener = gzip.open if str(path).endswith(".gz") else open
        with opener(path, "wt") as f:
            for (n1, c1, s1, n2, c2, s2) in part:
                f.write(f"@{n1} {c1}\n{s1}\n+\n{'F' * len(s1)}\n@{n2} {c2}\n{s2}\n+\n{'F' * len(s2)}\n")
    args2 = [a for a in args[:-1]]
    args2[args2.index("-b") + 1] = str(tmp_path / "multi")
    args2[args2.index("-t") + 1] = "5"
    args2[args2.index("--batch-pairs") + 1] = "333"
    args2[args2.index("--barcode-counts") + 1] = str(tmp_path / "counts2")
    res2 = subprocess.run(args2 + [str(x) for x in paths], capture_output=True, text=True, timeout=300)
    assert res2.returncode == 0, res2.stderr[-2000:]
    # ... and with the reads packed and classified on the device (ARKS_DEVICE_PACK=1: the workers ship the bases
    # as text, "Skipped invalid read pairs" is then counted by arks_gate_count_device): the same log and files
    args3 = list(args2)
    args3[args3.index("-b") + 1] = str(tmp_path / "multi_dp")
    args3[args3.index("--barcode-counts") + 1] = str(tmp_path / "counts3")
    res3 = subprocess.run(args3 + [str(x) for x in paths], capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, ARKS_DEVICE_PACK="1"))
    assert res3.returncode == 0, res3.stderr[-2000:]
    counters = lambda t: [ln for ln in t.split("\n") if ln.startswith(("Stored ", "Skipped ", "Total valid", "Number ", "File contains", "WARNING"))]
    assert counters(res3.stdout) == counters(res2.stdout) and len(counters(res3.stdout)) > 20
    assert any(ln.startswith("Skipped invalid read pairs: ") and not ln.endswith(": 0") for ln in counters(res3.stdout))
    for suffix in ("_original.gv", "_pair.tsv", "_main.tsv"):
        assert open(str(tmp_path / "multi_dp") + suffix).read() == open(base + suffix).read(), suffix
    for suffix in ("_original.gv", "_pair.tsv", "_main.tsv"):
        assert open(str(tmp_path / "multi") + suffix).read() == open(base + suffix).read(), suffix
    assert open(str(tmp_path / "counts2.tsv")).read() == open(str(tmp_path / "counts.tsv")).read()
    out2 = res2.stdout
    assert out2.count("Stored read pairs:") == 3 and out2.count("File contains unpaired reads:") == n_unpaired
    last = out2[out2.rindex("Stored read pairs:"):]
    assert f"Total valid kmers: {st['total_valid']}\n" in last          # s_* counters are cumulative
    assert f"Number of reads passing jaccard threshold: {st['reads_pass']}\n" in last
    stored_parts = [int(x.split("\n")[0]) for x in out2.split("Stored read pairs: ")[1:]]
    assert sum(stored_parts) == stored and stored_parts[2] == 0
    # messages keep file order
    assert out2.index(f"Reading chrom {paths[0]}") < out2.index(f"Reading chrom {paths[1]}") < \
        out2.index(f"Reading chrom {paths[2]}")
    # ---- one process per GPU (--ranks N; here the ranks share the one GPU): the files are dealt to the
    #      ranks, rank 0 merges their results -- same log (but for pid and time stamps), same files ---------
    import re as _re

    def _norm(text, basename, countsname):
        keep = []
        for ln in text.split("\n"):
            if ln.startswith(" pid ") or "Cumulative memory usage" in ln or _re.search(r"\d\d:\d\d:\d\d \d{4}$", ln):
                continue
            keep.append(ln.replace("/" + countsname, "/COUNTS").replace("/" + basename, "/BASE"))
        return "\n".join(keep)

    for n_ranks in (2, 3):
        argsr = list(args2)
        tag = f"ranks{n_ranks}"
        argsr[argsr.index("-b") + 1] = str(tmp_path / tag)
        argsr[argsr.index("--barcode-counts") + 1] = str(tmp_path / (tag + "_counts"))
        resr = subprocess.run(argsr + ["--ranks", str(n_ranks), "--share-devices"] + [str(x) for x in paths], capture_output=True,
                              text=True, timeout=300)
        assert resr.returncode == 0, resr.stderr[-2000:]
        # every GPU rank says which device it got (stderr; here they share the box's GPU, which has to be asked for)
        said = [ln for ln in resr.stderr.split("\n") if ln.startswith("arcs: GPU rank ")]
        assert len(said) == n_ranks and all("-> device " in ln for ln in said), resr.stderr[-2000:]
        for suffix in ("_original.gv", "_pair.tsv", "_main.tsv", ".dist.gv"):
            assert open(str(tmp_path / tag) + suffix).read() == open(str(tmp_path / "multi") + suffix).read(), suffix
        assert open(str(tmp_path / (tag + "_counts.tsv"))).read() == open(str(tmp_path / "counts2.tsv")).read()
        assert _norm(resr.stdout, tag, tag + "_counts") == _norm(res2.stdout, "multi", "counts2")
    # a file that cannot be opened: with --ranks the same log, stderr and exit status as one process (the workers
    # are not even started: the reference's sequence -- the files in front of it, then the message -- is rank 0's)
    bad_paths = [str(paths[0]), str(tmp_path / "no_such_reads.fq"), str(paths[2])]
    one = subprocess.run(args2 + bad_paths, capture_output=True, text=True, timeout=300)
    two = subprocess.run(args2 + ["--ranks", "2", "--share-devices"] + bad_paths, capture_output=True, text=True, timeout=300)
    assert one.returncode == two.returncode and one.returncode != 0
    two_err = "\n".join(ln for ln in two.stderr.split("\n") if not ln.startswith("arcs: GPU rank "))
    assert "no_such_reads.fq" in one.stderr and one.stderr == two_err
    # more GPU ranks than devices is an error unless it is asked for: a run meant for eight GPUs that finds one says so
    import torch as _torch
    if _torch.cuda.device_count() == 1:
        for extra in (["--ranks", "2"], ["--index-sharded=2"]):
            refused = subprocess.run(args2 + extra + [str(x) for x in paths], capture_output=True, text=True, timeout=300)
            assert refused.returncode != 0 and "--share-devices" in refused.stderr, (extra, refused.stderr[-500:])
    assert _norm(one.stdout, "multi", "counts2") == _norm(two.stdout, "multi", "counts2")
    # ---- the contig k-mer index in three parts (--index-shards): per batch the votes of every part,
    #      their maximum, then the j_index test -- same files, same stored pairs ----------------------
    args4 = list(args)
    args4[args4.index("-b") + 1] = str(tmp_path / "sharded")
    args4[args4.index("--barcode-counts") + 1] = str(tmp_path / "counts4")
    args4[-1:-1] = ["--index-shards", "3"]
    res4 = subprocess.run(args4, capture_output=True, text=True, timeout=300)
    assert res4.returncode == 0, res4.stderr[-2000:]
    for suffix in ("_original.gv", "_pair.tsv", "_main.tsv"):
        assert open(str(tmp_path / "sharded") + suffix).read() == open(base + suffix).read(), suffix
    assert f"Stored read pairs: {stored}\n" in res4.stdout
    # -v: the counters of the index build and of the read stage are collected over the parts (every key is in one
    # part: its first holder, arks_index_build_shard) and read as in the run with one index -- the whole log does
    counter_lines = [ln for ln in res.stdout.split("\n") if ln.startswith(("Total number of Kmers", "Number Null Kmers",
                     "Number Kmers Recorded", "Number Kmer Collisions", "Number Times Kmers Removed", "Number of unique kmers",
                     "Total valid kmers", "Number invalid kmers", "Number of kmers found", "Number of kmers recorded",
                     "Number of reads passing", "Number of reads failing"))]
    assert len(counter_lines) == 13
    for ln in counter_lines:
        assert ln + "\n" in res4.stdout, ln
    assert "not collected" not in res4.stdout
    assert _norm(res4.stdout, "sharded", "counts4") == _norm(res.stdout, "out", "counts")
    # ---- -D: distance estimates (dist_est.hpp) on the same run: d= / maxd= on the edges, --dist_tsv,
    #      --samples_tsv, d= of the ABySS graph ------------------------------------------------------------
    args5 = list(args)
    args5[args5.index("-b") + 1] = str(tmp_path / "dist")
    args5[args5.index("--barcode-counts") + 1] = str(tmp_path / "counts5")
    args5[-1:-1] = ["-D", "-B", "3", "--dist_upper", "--dist_tsv", str(tmp_path / "dist.tsv"), "--samples_tsv",
                    str(tmp_path / "samples.tsv")]
    res5 = subprocess.run(args5, capture_output=True, text=True, timeout=300)
    assert res5.returncode == 0, res5.stderr[-2000:]
    PD = dict(P, end_length=30000, dist_bin_size=3)
    ids_d, all_edges = G.create_graph(pmap, P)
    samples = G.dist_samples(imap, lengths, mult, PD)
    pstats = G.pair_barcode_stats(imap, mult, lengths, PD)
    est = G.edge_distances(ids_d, all_edges, pstats, G.jaccard_to_dist(samples), PD)
    assert len(samples) >= 2
    assert open(str(tmp_path / "samples.tsv")).read() == G.samples_text(samples)
    assert open(str(tmp_path / "dist.tsv")).read() == G.dist_tsv_text(ids_d, all_edges, est, pstats)
    assert open(str(tmp_path / "dist") + "_original.gv").read() == G.graph_text_with_distances(ids_d, all_edges, est, dead)
    dl = open(str(tmp_path / "dist") + ".dist.gv").read()
    for (u, v, o, w), e in zip(all_edges, est):
        if u not in dead and v not in dead:
            dd = e[2] if e is not None else 2**31 - 1                # --dist_upper
            assert f'"{ids_d[u]}{"-" if o < 2 else "+"}" -> "{ids_d[v]}{"-" if o % 2 else "+"}" [d={dd} e=100.0 n={w}]' in dl
    assert "=> Calculating distance estimates..." in res5.stdout and "=> Adding edge distances..." in res5.stdout
    if not use_mult_file:
        # ---- no -u: the barcode pre-pass is fused into the mapping pass; the literal two-pass flow
        #      (ARKS_TWO_PASS=1) must print the same log and write the same files ------------------
        import re
        def normalized(text, basename, countsname="counts"):
            keep = []
            for ln in text.split("\n"):
                if ln.startswith(" pid ") or "Cumulative memory usage" in ln or re.search(r"\d\d:\d\d:\d\d \d{4}$", ln):
                    continue
                keep.append(ln.replace("/" + countsname, "/COUNTS").replace("/" + basename, "/BASE"))
            return "\n".join(keep)
        args3 = list(args)
        args3[args3.index("-b") + 1] = str(tmp_path / "twopass")
        args3[args3.index("--barcode-counts") + 1] = str(tmp_path / "counts3")
        res3 = subprocess.run(args3, capture_output=True, text=True, timeout=300,
                              env=dict(os.environ, ARKS_TWO_PASS="1"))
        assert res3.returncode == 0, res3.stderr[-2000:]
        assert normalized(res3.stdout, "twopass", "counts3") == normalized(res.stdout, "out", "counts")
        assert res3.stdout.count("Reading chrom") == 2 and "distinct barcode." in res3.stdout
        for suffix in ("_original.gv", "_pair.tsv", "_main.tsv"):
            assert open(str(tmp_path / "twopass") + suffix).read() == open(base + suffix).read(), suffix
        assert open(str(tmp_path / "counts3.tsv")).read() == open(str(tmp_path / "counts.tsv")).read()
        # a zero-length read: readBarcodes stops counting there, the pair loop goes on -> the fused
        # pass notices and falls back; both flows agree again
        zl = tmp_path / "zero.fq"
        with open(zl, "w") as f:
            for i, (n1, c1, s1, n2, c2, s2) in enumerate(recs[:900]):
                if i == 450:
                    f.write(f"@z/1 {c1}\n\n+\n\n@z/2 {c1}\n\n+\n\n")
                f.write(f"@{n1} {c1}\n{s1}\n+\n{'F' * len(s1)}\n@{n2} {c2}\n{s2}\n+\n{'F' * len(s2)}\n")
        outs = []
        for tag, env in (("zf", {}), ("zt", {"ARKS_TWO_PASS": "1"})):
            a = list(args[:-1])
            a[a.index("-b") + 1] = str(tmp_path / tag)
            a[a.index("--barcode-counts") + 1] = str(tmp_path / (tag + "_counts"))
            r = subprocess.run(a + [str(zl)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append((normalized(r.stdout, tag, tag + "_counts").replace("/" + tag, "/BASE"), open(str(tmp_path / tag) + "_main.tsv").read(),
                         open(str(tmp_path / (tag + "_counts.tsv"))).read()))
        assert outs[0] == outs[1]
        assert "invalid barcode" not in outs[0][0] or "barcodes not in the barcode multiplicity file" in outs[0][0]


def test_arcs_cli_k_list_single_pass(arks, gpu, oracle, tmp_path):
    """-k 40,60,80: one pass over the reads against three resident indexes; every output set equals the
    one of a single-k run (arks-long style input: pseudo-linked pairs with a multiplicity file)"""
    from arcs_amd import build as b, synth
    exe = b.build_host()
    contigs = synth.make_draft(300_000, seed=91, lengths=(50000, 30000, 80000), small_frac=0.3)
    cs = synth.contigs_to_strings(contigs)
    fa = tmp_path / "draft.fa"
    with open(fa, "w") as f:
        for i, s_ in enumerate(cs):
            f.write(f">{i + 1}\n{s_}\n")
    n_pairs = 5000
    batch = synth.make_read_pairs(contigs, n_pairs, seed=92, mol_len=20000, pairs_per_mol=25)
    reads = synth.reads_to_strings(batch)
    bid = batch["barcode_id"].numpy()
    mult = {}
    fq = tmp_path / "reads.fq"
    with open(fq, "w") as f:
        for p in range(n_pairs):
            bc = f"{int(bid[p]) + 1}"
            mult[bc] = mult.get(bc, 0) + 2
            for m in (0, 1):
                r = reads[2 * p + m]
                f.write(f"@r{p}/{m + 1} BX:Z:{bc}\n{r}\n+\n{'F' * len(r)}\n")
    mf = tmp_path / "mult.tsv"
    mf.write_text("".join(f"{k}\t{v}\n" for k, v in mult.items()))
    common = [exe, "--arks", "-v", "-f", str(fa), "-c", "3", "-m", "8-10000", "-e", "30000", "-z", "500", "-j", "0.5",
              "-t", "4", "-u", str(mf), "-P", "--batch-pairs", "1500"]
    r = subprocess.run(common + ["-k", "40,60,80", "-b", str(tmp_path / "multi"), str(fq)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    differ = set()
    for k in (40, 60, 80):
        r1 = subprocess.run(common + ["-k", str(k), "-b", str(tmp_path / f"single{k}"), str(fq)], capture_output=True,
                            text=True, timeout=300)
        assert r1.returncode == 0, r1.stderr[-2000:]
        for suffix in ("_original.gv", "_pair.tsv", "_main.tsv", ".dist.gv"):
            a = open(str(tmp_path / f"multi_k{k}") + suffix).read()
            assert a == open(str(tmp_path / f"single{k}") + suffix).read(), (k, suffix)
        differ.add(open(str(tmp_path / f"single{k}") + "_main.tsv").read())
        # the per-k counter blocks of the single pass carry the single-k numbers
        block = r.stdout[r.stdout.index(f"k = {k}:\nStored read pairs"):]
        line = [ln for ln in r1.stdout.split("\n") if ln.startswith("Total valid kmers:")][0]
        assert line in block.split("k = ")[1]
    assert len(differ) > 1, "the three k should not all give the same evidence"
    # every k of the single pass against the CPU oracle + tests/graph_ref.py (not against another CLI run)
    from util import expected_cli_outputs
    recs = [(f"r{p}/1", f"BX:Z:{int(bid[p]) + 1}", reads[2 * p], f"r{p}/2", f"BX:Z:{int(bid[p]) + 1}", reads[2 * p + 1])
            for p in range(n_pairs)]
    P = {"min_reads": 3, "min_links": 0, "min_mult": 8, "max_mult": 10000, "max_degree": 0, "error_percent": 0.05,
         "gap": 100}
    names = [str(i + 1) for i in range(len(cs))]
    n_edges = 0
    for k in (40, 60, 80):
        want = expected_cli_outputs(oracle, G, names, cs, recs, mult, k, 0.5, P)
        base = str(tmp_path / f"multi_k{k}")
        assert open(base + "_original.gv").read() == want["original.gv"], k
        assert open(base + "_pair.tsv").read() == want["pair.tsv"], k
        assert open(base + "_main.tsv").read() == want["main.tsv"], k
        block = r.stdout[r.stdout.index(f"k = {k}:\nStored read pairs"):].split("k = ")[1]
        assert f"Stored read pairs: {int((want['pair'] != 0).sum())}\n" in block
        assert f"Number of kmers found in ContigKmap: {want['stats']['found']}\n" in block
        n_edges += len(want["edges"])
    assert n_edges > 0


def test_arks_long_pipe(arks, gpu, oracle, tmp_path):
    """the arks-long flow of bin/arcs-make:299-313: long reads -> long-to-linked-pe (pseudo-linked pairs,
    BX = read number) piped into `arcs --arks ... -u multiplicities /dev/stdin`; same outputs as reading
    the materialised pairs from a file, for a k list in one pass"""
    from arcs_amd import build as b, synth
    exe = b.build_host()
    feeder = os.path.join(os.path.dirname(exe), "long-to-linked-pe")
    rng = np.random.Generator(np.random.PCG64(33))
    contigs = synth.make_draft(600_000, seed=93, lengths=(70000, 40000, 110000), small_frac=0.2)
    cs = synth.contigs_to_strings(contigs)
    genome = "".join(cs)
    fa = tmp_path / "draft.fa"
    fa.write_text("".join(f">{i + 1}\n{s_}\n" for i, s_ in enumerate(cs)))
    comp = str.maketrans("ACGT", "TGCA")
    long_reads = []
    with gzip.open(tmp_path / "long.fa.gz", "wt") as f:
        for i in range(400):
            n = int(rng.integers(3000, 30000))
            p0 = int(rng.integers(0, len(genome) - n))
            r = list(genome[p0:p0 + n])
            for q in rng.integers(0, n, size=n // 200):          # 0.5 % substitutions
                r[q] = "ACGT"[int(rng.integers(4))]
            r = "".join(r)
            if i % 2:
                r = r[::-1].translate(comp)
            long_reads.append(r)
            f.write(f">long{i}\n{r}\n")
    mult = tmp_path / "bx.tsv"
    subprocess.run([feeder, "-l", "250", "-m", "2000", "--bx-only", "-b", str(mult), str(tmp_path / "long.fa.gz")], check=True)
    pairs = subprocess.run([feeder, "-l", "250", "-m", "2000", "-t", "2", str(tmp_path / "long.fa.gz")],
                           capture_output=True, check=True).stdout
    (tmp_path / "pairs.fq").write_bytes(pairs)
    common = [exe, "--arks", "-v", "-f", str(fa), "-c", "4", "-m", "8-10000", "-e", "30000", "-z", "500", "-j", "0.05",
              "-k", "20,40", "-t", "3", "-u", str(mult), "--batch-pairs", "2000"]
    piped = subprocess.run(common + ["-b", str(tmp_path / "piped"), "/dev/stdin"], input=pairs, capture_output=True,
                           timeout=300)
    assert piped.returncode == 0, piped.stderr[-2000:]
    filed = subprocess.run(common + ["-b", str(tmp_path / "filed"), str(tmp_path / "pairs.fq")], capture_output=True,
                           timeout=300)
    assert filed.returncode == 0, filed.stderr[-2000:]
    n_edges = 0
    for k in (20, 40):
        for suffix in ("_original.gv", "_main.tsv", ".dist.gv"):
            a = open(str(tmp_path / f"piped_k{k}") + suffix).read()
            assert a == open(str(tmp_path / f"filed_k{k}") + suffix).read(), (k, suffix)
        n_edges += open(str(tmp_path / f"piped_k{k}") + "_original.gv").read().count("--")
    assert n_edges > 0, "long reads spanning contigs should link some ends"
    # the pseudo-linked pairs by the rules of src/long-to-linked-pe.cpp:224-287 (restated in
    # tests/test_host_longreads.py::expected), then chromiumRead's flow through the CPU oracle and the
    # graph stage through tests/graph_ref.py: what the pipe must have produced, independent of any CLI run
    from test_host_longreads import expected as expected_pairs
    from util import expected_cli_outputs
    want_text, want_bx = expected_pairs([(f"long{i}", s_, "") for i, s_ in enumerate(long_reads)], 250, 2000, False)
    assert pairs.decode() == want_text and mult.read_text() == want_bx
    lines = want_text.split("\n")
    recs = []
    for i in range(0, len(lines) - 1, 8):
        n1, c1 = lines[i][1:].split(" ", 1)
        n2, c2 = lines[i + 4][1:].split(" ", 1)
        recs.append((n1, c1, lines[i + 1], n2, c2, lines[i + 5]))
    multd = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in want_bx.split("\n") if ln}
    P = {"min_reads": 4, "min_links": 0, "min_mult": 8, "max_mult": 10000, "max_degree": 0, "error_percent": 0.05,
         "gap": 100}
    names = [str(i + 1) for i in range(len(cs))]
    for k in (20, 40):
        want = expected_cli_outputs(oracle, G, names, cs, recs, multd, k, 0.05, P)
        assert open(str(tmp_path / f"piped_k{k}") + "_original.gv").read() == want["original.gv"], k
        assert open(str(tmp_path / f"piped_k{k}") + "_main.tsv").read() == want["main.tsv"], k


def test_arks_long_multi_k_at_scale(arks, gpu, oracle, tmp_path):
    """BASELINE configs[4]'s shape on one GPU, cut to a test's size: a 100 Mbp draft, ONT-like long reads cut
    into 250-bp pseudo-linked pairs by long-to-linked-pe (bin/arcs-make:299-313), `arcs --arks -k 40,60,80
    -j 0.05` in one pass over the pipe; every k's outputs against the CPU oracle + tests/graph_ref.py."""
    from arcs_amd import build as b, synth
    from test_host_longreads import expected as expected_pairs
    from util import expected_cli_outputs
    exe = b.build_host()
    feeder = os.path.join(os.path.dirname(exe), "long-to-linked-pe")
    rng = np.random.Generator(np.random.PCG64(55))
    contigs = synth.make_draft(100_000_000, seed=95)
    cs = synth.contigs_to_strings(contigs)
    names = [f"s{i + 1}" for i in range(len(cs))]
    fa = tmp_path / "draft.fa"
    with open(fa, "w") as f:
        for n, s_ in zip(names, cs):
            f.write(f">{n}\n{s_}\n")
    genome = np.concatenate(contigs)
    comp = np.zeros(256, dtype=np.uint8)
    for a_, b_ in zip(b"ACGTN", b"TGCAN"):
        comp[a_] = b_
    long_reads = []
    with open(tmp_path / "long.fa", "w") as f:
        bounds = np.cumsum([len(c) for c in contigs])[:-1]
        hot = bounds[rng.choice(len(bounds), size=60, replace=False)]   # joins that several reads span
        for i in range(1500):
            n = int(rng.integers(5000, 40000))
            if i % 2:
                p0 = int(hot[int(rng.integers(len(hot)))]) - int(rng.integers(2000, n - 2000))
                p0 = min(max(p0, 0), len(genome) - n)
            else:
                p0 = int(rng.integers(0, len(genome) - n))
            r = genome[p0:p0 + n].copy()
            q = rng.integers(0, n, size=n // 50)                 # 2 % substitutions: ONT-like
            r[q] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(q))]
            if i % 4 >= 2:
                r = comp[r[::-1]]
            long_reads.append(r.tobytes().decode())
            f.write(f">ont{i}\n{long_reads[-1]}\n")
    mult = tmp_path / "bx.tsv"
    subprocess.run([feeder, "-l", "250", "-m", "2000", "--bx-only", "-b", str(mult), str(tmp_path / "long.fa")], check=True)
    pairs = subprocess.run([feeder, "-l", "250", "-m", "2000", "-t", "4", str(tmp_path / "long.fa")],
                           capture_output=True, check=True).stdout
    res = subprocess.run([exe, "--arks", "-v", "-f", str(fa), "-c", "4", "-m", "8-10000", "-e", "30000", "-z", "500",
                          "-j", "0.05", "-k", "40,60,80", "-t", "8", "-u", str(mult), "-b", str(tmp_path / "out"),
                          "/dev/stdin"], input=pairs, capture_output=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    want_text, want_bx = expected_pairs([(f"ont{i}", s_, "") for i, s_ in enumerate(long_reads)], 250, 2000, False)
    assert pairs.decode() == want_text and mult.read_text() == want_bx
    lines = want_text.split("\n")
    recs = []
    for i in range(0, len(lines) - 1, 8):
        n1, c1 = lines[i][1:].split(" ", 1)
        n2, c2 = lines[i + 4][1:].split(" ", 1)
        recs.append((n1, c1, lines[i + 1], n2, c2, lines[i + 5]))
    assert len(recs) > 50_000
    multd = {ln.split("\t")[0]: int(ln.split("\t")[1]) for ln in want_bx.split("\n") if ln}
    P = {"min_reads": 4, "min_links": 0, "min_mult": 8, "max_mult": 10000, "max_degree": 0, "error_percent": 0.05,
         "gap": 100}
    out = res.stdout.decode()
    n_edges = 0
    for k in (40, 60, 80):
        want = expected_cli_outputs(oracle, G, names, cs, recs, multd, k, 0.05, P, threads=min(64, os.cpu_count() or 1))
        base = str(tmp_path / f"out_k{k}")
        assert open(base + "_original.gv").read() == want["original.gv"], k
        assert open(base + "_main.tsv").read() == want["main.tsv"], k
        block = out[out.index(f"k = {k}:\nStored read pairs"):].split("k = ")[1]
        assert f"Stored read pairs: {int((want['pair'] != 0).sum())}\n" in block
        assert f"Number of kmers found in ContigKmap: {want['stats']['found']}\n" in block
        assert f"Number of reads passing jaccard threshold: {want['stats']['reads_pass']}\n" in block
        n_edges += len(want["edges"])
    assert n_edges > 20, "long reads spanning contigs should link many ends"


def _norm_log(text, basename):
    import re
    keep = []
    for ln in text.split("\n"):
        if ln.startswith(" pid ") or "Cumulative memory usage" in ln or re.search(r"\d\d:\d\d:\d\d \d{4}$", ln):
            continue
        keep.append(ln.replace("/" + basename, "/BASE"))
    return "\n".join(keep)


@pytest.mark.parametrize("kind", ["fq", "bgzf", "fq.gz"])
def test_one_reads_file_many_gpus(arks, gpu, tmp_path, kind):
    """What the pipeline passes is ONE reads file (bin/arcs-make:290).  --ranks N then deals the batches of that file to
    N GPU lanes of the one process (an index replica each; here the lanes share the box's GPU), and --index-sharded=N
    maps them against a seed table hash-sharded over the N lanes (arks_exchange, rounds of N batches, two rounds in
    flight).  Log (-v counters included) and every output file are those of the one-GPU run, byte for byte; that
    run is checked against the oracle flow in test_arcs_cli_end_to_end."""
    from arcs_amd import build as b, synth
    from test_host_ingest import write_bgzf
    exe = b.build_host()
    contigs = synth.make_draft(400_000, seed=77, lengths=(60000, 20000, 90000, 45000), small_frac=0.5)
    cs = synth.contigs_to_strings(contigs)
    fa = tmp_path / "draft.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(cs):
            f.write(f">{i + 1}\n{s}\n")
    n_pairs = 9000
    batch = synth.make_read_pairs(contigs, n_pairs, seed=78, mol_len=30000, pairs_per_mol=30, one_n_rate=0.03,
                                  many_n_rate=0.01)
    reads = synth.reads_to_strings(batch)
    bid = batch["barcode_id"].numpy()
    text = []
    for p in range(n_pairs):
        bc = "".join("ACGT"[(int(bid[p]) >> (2 * t)) & 3] for t in range(12)) + "-1"
        n2 = f"other{p}" if p % 97 == 5 else f"read{p}"
        for name, s in ((f"read{p}/1", reads[2 * p]), (n2 + "/2", reads[2 * p + 1])):
            text.append(f"@{name} BX:Z:{bc}\n{s}\n+\n{'F' * len(s)}\n")
    text = "".join(text).encode()
    if kind == "fq":
        fq = tmp_path / "reads.fq"
        fq.write_bytes(text)
    elif kind == "bgzf":
        fq = tmp_path / "reads.fastq.gz"
        write_bgzf(str(fq), text)
    else:
        fq = tmp_path / "reads.fq.gz"
        with gzip.open(fq, "wb") as f:
            f.write(text)

    def run(tag, extra, k="60"):
        args = [exe, "--arks", "-v", "-f", str(fa), "-c", "3", "-m", "8-10000", "-e", "30000", "-z", "500", "-j", "0.55",
                "-k", k, "-t", "6", "-b", str(tmp_path / tag), "-P", "--batch-pairs", "700"] + extra + [str(fq)]
        res = subprocess.run(args, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, (tag, res.stderr[-2000:])
        return res
    one = run("one", [])
    assert "Stored read pairs: " in one.stdout and "Stored read pairs: 0\n" not in one.stdout
    for tag, extra in (("r2", ["--ranks", "2"]), ("r3", ["--ranks", "3"]), ("s2", ["--index-sharded=2"]),
                       ("s3", ["--index-sharded", "--ranks", "3"])):
        res = run(tag, extra + ["--share-devices"])
        said = [ln for ln in res.stderr.split("\n") if ln.startswith("arcs: GPU rank ")]
        assert len(said) == int(extra[-1][-1]), (tag, res.stderr[-1000:])
        assert _norm_log(res.stdout, tag) == _norm_log(one.stdout, "one"), tag
        for suffix in ("_original.gv", "_pair.tsv", "_main.tsv", ".dist.gv"):
            assert open(str(tmp_path / tag) + suffix).read() == open(str(tmp_path / "one") + suffix).read(), (tag, suffix)
    if kind == "fq":
        # two k in one pass over a sharded seed table (an exchange group per k), and the pipe of arcs-make:305
        base = run("k2", [], k="40,60")
        res = run("k2s", ["--index-sharded=3", "--share-devices"], k="40,60")
        assert _norm_log(res.stdout, "k2s") == _norm_log(base.stdout, "k2")
        for kk in (40, 60):
            for suffix in ("_original.gv", "_main.tsv"):
                a = [x for x in os.listdir(tmp_path) if x.startswith("k2s") and f"k{kk}" in x and x.endswith(suffix)]
                assert len(a) == 1, (kk, suffix, a)
                assert open(tmp_path / a[0]).read() == open(tmp_path / a[0].replace("k2s", "k2", 1)).read()
        args = [exe, "--arks", "-f", str(fa), "-c", "3", "-m", "8-10000", "-k", "60", "-t", "4", "-b", str(tmp_path / "pipe"),
                "--batch-pairs", "700", "--ranks", "3", "--share-devices", "/dev/stdin"]
        with open(fq, "rb") as src:
            res = subprocess.run(args, stdin=src, capture_output=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        assert open(str(tmp_path / "pipe") + "_original.gv").read() == open(str(tmp_path / "one") + "_original.gv").read()
