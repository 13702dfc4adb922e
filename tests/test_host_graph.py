"""The host graph stage (arcs_amd/host/graph.hpp: IndexMap -> PairMap -> graph -> .gv / .dist.gv /
TSV writers) on CPU: (1) pinned byte-for-byte to the reference's own arks-long demo outputs
(_original.gv + contig lengths -> .dist.gv), (2) compared with the Python restatement on random
IndexMaps."""
import os
import subprocess

import numpy as np
import pytest

import graph_ref as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def graph_check(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bin") / "graph_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(), "-I" + os.path.join(ROOT, "arcs_amd", "host"),
                           os.path.join(ROOT, "arcs_amd", "host", "graph_check.cpp"), "-o", out])
    return out


def test_demo_dist_gv(graph_check, tmp_path):
    """Examples/arks-long_test-demo/output: the committed _original.gv and .dist.gv come from one
    reference run; contig lengths 1:61348 2:38068 3:23450 (its dist.gv), --gap 100.  The ORDER of the
    vertex (and hence edge) lines is the iteration order of a std::unordered_map<std::string,int>
    (Arcs.cpp:1622), which depends on the libstdc++ version the reference was built with (the bucket
    growth sequence changed between GCC releases), so lines are compared as a set, the per-contig
    "+"/"-" adjacency and the file frame exactly."""
    lengths = tmp_path / "len.tsv"
    lengths.write_text("1\t61348\n2\t38068\n3\t23450\n")
    out = tmp_path / "out.dist.gv"
    subprocess.check_call([graph_check, "gv", os.path.join(GOLDEN, "arks-long_demo_original.gv"),
                           str(lengths), str(out), "100"])
    got = out.read_text().split("\n")
    want = open(os.path.join(GOLDEN, "arks-long_demo.dist.gv")).read().split("\n")
    assert got[0] == want[0] == "digraph arcs {" and got[-2:] == want[-2:] == ["}", ""]
    assert sorted(got) == sorted(want) and len(got) == len(want)
    assert sorted(got[1:7]) == sorted(want[1:7])          # vertex block, then edge block
    for i in range(1, 7, 2):
        assert got[i].split('"')[1][:-1] == got[i + 1].split('"')[1][:-1]
        assert got[i].split('"')[1][-1] == "+" and got[i + 1].split('"')[1][-1] == "-"


def test_gv_format_matches_reference_demo():
    """write_graph reproduces the boost::write_graphviz layout of the arks demo output"""
    want = open(os.path.join(GOLDEN, "arks_demo_original.gv")).read()
    ids, edges = ["1", "2", "3"], [(0, 1, 2, 15), (1, 2, 2, 28)]
    assert G.graph_text(ids, edges) == want


@pytest.mark.parametrize("mode", ["imap", "imapfast"])    # graph.hpp (the reference's containers) / graph_fast.hpp
@pytest.mark.parametrize("seed,c,l,d,r", [(1, 5, 0, 0, 0.05), (2, 3, 2, 0, 0.05), (3, 5, 0, 3, 0.05),
                                          (4, 1, 0, 0, 0.5), (5, 4, 1, 2, 0.01), (6, 0, 0, 0, 0.05)])
def test_graph_stage_vs_python(graph_check, tmp_path, seed, c, l, d, r, mode):
    rng = np.random.Generator(np.random.PCG64(seed))
    contigs = [str(x) for x in rng.permutation(40)[:25] + 1] + ["ctgA", "scaf_10", "9"]
    lengths = {x: int(rng.integers(500, 200000)) for x in contigs}
    imap, mult = {}, {}
    for b in range(400):
        bc = "".join(rng.choice(list("ACGT"), size=16)) + "-1"
        mult[bc] = int(rng.integers(1, 300))
        if rng.random() < 0.2:
            continue
        sm = {}
        base = int(rng.integers(len(contigs)))
        for t in range(int(rng.integers(1, 5))):
            ctg = contigs[(base + t) % len(contigs)]
            if rng.random() < 0.8:
                sm[(ctg, bool(rng.random() < 0.5))] = int(rng.integers(1, 30))
            if rng.random() < 0.3:
                sm[(ctg, bool(rng.random() < 0.5))] = int(rng.integers(1, 12))
        if sm:
            imap[bc] = sm
    P = {"min_reads": c, "min_links": l, "min_mult": 20, "max_mult": 250, "max_degree": d,
         "error_percent": r, "gap": 77}
    (tmp_path / "imap.tsv").write_text("".join(
        f"{bc}\t{ctg}\t{'H' if h else 'T'}\t{n}\n" for bc, sm in imap.items() for (ctg, h), n in sm.items()))
    (tmp_path / "mult.tsv").write_text("".join(f"{b}\t{m}\n" for b, m in mult.items()))
    (tmp_path / "len.tsv").write_text("".join(f"{k}\t{v}\n" for k, v in lengths.items()))
    base = str(tmp_path / "out")
    # (graph_fast.hpp puts the TSV together on several threads, a block of pairs each: blocks of 3 pairs here)
    summary = subprocess.check_output([graph_check, mode, str(tmp_path / "imap.tsv"), str(tmp_path / "mult.tsv"),
                                       str(tmp_path / "len.tsv"), base, str(c), str(l), "20", "250", str(d), str(r),
                                       "77", "x"], text=True, env=dict(os.environ, GRAPH_THREADS=str(1 + seed % 4), GRAPH_TSV_BLOCK="3"))
    assert f'"Scaffold_end_barcodes":{len(imap)},' in summary and f'"All_barcodes_unfiltered":{len(mult)},' in summary
    G.add_opposite_ends(imap)
    pmap = G.pair_contigs(imap, mult, P)
    assert len(pmap) > 5
    ids, edges = G.create_graph(pmap, P)
    dead = set()
    if d:
        dead, edges = G.remove_degree_nodes(ids, edges, d)
    assert open(base + "_pair.tsv").read() == G.pair_text(pmap)
    assert open(base + "_original.gv").read() == G.graph_text(ids, edges, dead)
    assert open(base + "_main.tsv").read() == G.tsv_text(imap, pmap, mult, P)
    assert open(base + "_counts.tsv").read() == G.counts_text(mult)
    lines = open(base + ".dist.gv").read().split("\n")
    assert lines[0] == "digraph arcs {" and lines[-2] == "}" and lines[-1] == ""
    # as text: the vertex order is the ContigToLength unordered_map's, taken from the local libstdc++ itself
    assert open(base + ".dist.gv").read() == G.dist_graph_text(lengths, ids, edges, 77)
    vl, el = G.dist_graph_lines(lengths, ids, edges, 77)
    body = lines[1:-2]
    assert set(body[: 2 * len(lengths)]) == vl and len(body[: 2 * len(lengths)]) == len(vl)
    assert set(body[2 * len(lengths):]) == el and len(body[2 * len(lengths):]) == len(el)
    # '+' and '-' of a contig are adjacent, '+' first (ContigGraph::add_vertex)
    for i in range(0, 2 * len(lengths), 2):
        assert body[i].split('"')[1][:-1] == body[i + 1].split('"')[1][:-1]
        assert body[i].split('"')[1][-1] == "+" and body[i + 1].split('"')[1][-1] == "-"


@pytest.mark.parametrize("seed,c,d,end_length,bin_size,upper", [(11, 3, 0, 20000, 20, 0), (12, 2, 0, 5000, 3, 1),
                                                                 (13, 4, 3, 30000, 7, 0), (14, 1, 0, 100000, 20, 0),
                                                                 (15, 2, 2, 1000, 1, 1)])
def test_distance_estimates_vs_python(graph_check, tmp_path, seed, c, d, end_length, bin_size, upper):
    """-D (dist_est.hpp) against the Python restatement of Arcs/DistanceEst.h: samples, Jaccard map with
    this build's tie rule, pair statistics, closest keys, the reference's quantile, edge attributes in
    _original.gv, the --dist_tsv and --samples_tsv files and the d= of the ABySS graph"""
    rng = np.random.Generator(np.random.PCG64(seed))
    contigs = [str(x) for x in rng.permutation(60)[:30] + 1] + ["ctgA", "scaf_10", "9"]
    lengths = {x: int(rng.choice([900, 30000, 45000, 61000, 90000, 150000, 400000])) + int(rng.integers(0, 50))
               for x in contigs}
    imap, mult = {}, {}
    # the contigs form a chain with random orientations; a barcode is a molecule that covers the joint of
    # two neighbours (the facing ends get the read pairs) and now and then both ends of a contig
    facing = [bool(rng.random() < 0.5) for _ in contigs]       # which end of contig i faces contig i + 1
    for b in range(900):
        bc = "".join(rng.choice(list("ACGT"), size=16)) + "-1"
        mult[bc] = int(rng.integers(1, 300))
        sm = {}
        for _ in range(int(rng.integers(1, 3))):
            i = int(rng.integers(len(contigs) - 1))
            sm[(contigs[i], facing[i])] = int(rng.integers(4, 30))
            sm[(contigs[i + 1], not facing[i + 1])] = int(rng.integers(4, 30))
            if rng.random() < 0.35 and i % 4:                    # a long molecule: the far end of contig i too
                # (never for every fourth contig: their Jaccard index is 0 and the samples collide)
                sm[(contigs[i], not facing[i])] = int(rng.integers(1, 8))
            if rng.random() < 0.1:
                sm[(contigs[int(rng.integers(len(contigs)))], bool(rng.random() < 0.5))] = int(rng.integers(1, 6))
        imap[bc] = sm
    P = {"min_reads": c, "min_links": 0, "min_mult": 20, "max_mult": 250, "max_degree": d, "error_percent": 0.05,
         "gap": 77, "end_length": end_length, "dist_bin_size": bin_size}
    (tmp_path / "imap.tsv").write_text("".join(
        f"{bc}\t{ctg}\t{'H' if h else 'T'}\t{n}\n" for bc, sm in imap.items() for (ctg, h), n in sm.items()))
    (tmp_path / "mult.tsv").write_text("".join(f"{b}\t{m}\n" for b, m in mult.items()))
    (tmp_path / "len.tsv").write_text("".join(f"{k}\t{v}\n" for k, v in lengths.items()))
    base = str(tmp_path / "out")
    subprocess.check_call([graph_check, "imap", str(tmp_path / "imap.tsv"), str(tmp_path / "mult.tsv"),
                           str(tmp_path / "len.tsv"), base, str(c), "0", "20", "250", str(d), "0.05", "77", "x",
                           str(end_length), str(bin_size), str(upper)], stdout=subprocess.DEVNULL)
    G.add_opposite_ends(imap)
    pmap = G.pair_contigs(imap, mult, P)
    ids, all_edges = G.create_graph(pmap, P)
    assert len(all_edges) > 5
    samples = G.dist_samples(imap, lengths, mult, P)
    j2d = G.jaccard_to_dist(samples)
    stats = G.pair_barcode_stats(imap, mult, lengths, P)
    est = G.edge_distances(ids, all_edges, stats, j2d, P)
    if end_length <= 30000:
        assert len(samples) > 5 and sum(e is not None for e in est) > 3
        assert len(j2d[0]) < len(samples)           # equal Jaccard indices do collide
    else:
        assert len(samples) < 8                      # only the 400 kbp contigs are two end lengths long
    dead = set()
    if d:
        dead, _ = G.remove_degree_nodes(ids, all_edges, d)
    assert open(base + "_samples.tsv").read() == G.samples_text(samples)
    assert open(base + "_dist.tsv").read() == G.dist_tsv_text(ids, all_edges, est, stats)
    assert open(base + "_original.gv").read() == G.graph_text_with_distances(ids, all_edges, est, dead)
    # the ABySS graph: d = median (or upper bound) of the estimate; an edge without one keeps INT_MAX
    lines = open(base + ".dist.gv").read().split("\n")
    want = set()
    for (u, v, o, w), e in zip(all_edges, est):
        if u in dead or v in dead:
            continue
        dd = (e[2] if upper else e[1]) if e is not None else 2**31 - 1
        un, vn = ids[u] + ("-" if o < 2 else "+"), ids[v] + ("-" if o % 2 else "+")
        flip = lambda s: s[:-1] + ("+" if s[-1] == "-" else "-")
        want.add(f'"{un}" -> "{vn}" [d={dd} e=77.0 n={w}]')
        if un != flip(vn):
            want.add(f'"{flip(vn)}" -> "{flip(un)}" [d={dd} e=77.0 n={w}]')
    assert set(lines[1 + 2 * len(lengths):-2]) == want


def test_closest_keys_and_quantile_vs_reference_headers(oracle):
    """the restatements of Common/MapUtil.h and Common/StatUtil.h against those headers themselves,
    compiled where they lie into oracle/_ref (this container only)"""
    if not oracle.have_ref() or not hasattr(oracle.ref(), "ref_closest_keys"):
        pytest.skip("oracle/_ref is not built here")
    import ctypes as C
    R = oracle.ref()
    rng = np.random.Generator(np.random.PCG64(5))
    for case in range(3000):
        n = int(rng.integers(1, 40))
        keys = sorted(set(np.round(rng.random(n) * (1 if case % 3 else 0.2), int(rng.integers(1, 4))).tolist()))
        arr = (C.c_double * len(keys))(*keys)
        key = float(rng.choice(keys)) if case % 4 == 0 else float(rng.random())
        nn = int(rng.integers(1, 25))
        first, last = C.c_int(), C.c_int()
        R.ref_closest_keys(arr, len(keys), C.c_double(key), nn, C.byref(first), C.byref(last))
        assert (first.value, last.value) == G.closest_keys(keys, key, nn), (keys, key, nn)
        vals = sorted(int(x) for x in rng.integers(0, 400000, size=int(rng.integers(1, 30))))
        varr = (C.c_uint * len(vals))(*vals)
        for q in (0.01, 0.5, 0.99, float(rng.random())):
            R.ref_quantile.restype = C.c_double
            assert R.ref_quantile(varr, len(vals), C.c_double(q)) == G.quantile(vals, q), (vals, q)


def test_normal_estimation_cases():
    """headOrTail / checkSignificance edge cases: sum below -c, all on one end, even split"""
    P = {"min_reads": 5, "error_percent": 0.05}
    assert G.head_or_tail(2, 2, P) == (False, False)
    assert G.head_or_tail(10, 0, P) == (True, True)
    assert G.head_or_tail(0, 10, P) == (True, False)
    assert G.head_or_tail(5, 5, P) == (False, False)
    assert G.head_or_tail(4, 1, P) == (False, False)   # 1 - Phi(1.34) = 0.09 > 0.05
    assert G.head_or_tail(5, 0, P) == (True, True)


def _make_tsv(gv_path, fasta_path):
    """what bin/makeTSVfile.py does with an _original.gv (label -> LINKS orientation pairs,
    bin/makeTSVfile.py:38-113), restated for the test; the committed checkpoint files were produced
    by the reference script itself (tests/golden/make_tigpair_golden.py)"""
    import re
    num, n = {}, 0
    for line in open(fasta_path):
        if line[0] == ">":
            n += 1
            num[line.rstrip().split()[0][1:]] = str(n)
    name = {}
    for line in open(gv_path):
        m = re.match(r"(\d+)\s+\[id=\"?([^\]\"]+)\"?\]", line.rstrip())
        if m:
            name[m.group(1)] = m.group(2)
    out = []
    table = {0: ("r", "f", "r", "f"), 1: ("r", "r", "f", "f"), 2: ("f", "f", "r", "r"), 3: ("f", "r", "f", "r")}
    for line in open(gv_path):
        m = re.search(r"(\d+)--(\d+)\s+\[label=(\d+), weight=(\d+)", line.rstrip())
        if not m:
            continue
        a, b = name[m.group(1)], name[m.group(2)]
        if a > b:
            a, b = b, a
        oa, ob, rb, ra = table[int(m.group(3))]
        links = int(m.group(4))
        out.append(f"10\t{oa}{num[a]}\t{ob}{num[b]}\t{links}\t{links * 100}\n")
        out.append(f"10\t{rb}{num[b]}\t{ra}{num[a]}\t{links}\t{links * 100}\n")
    return "".join(out)


@pytest.mark.parametrize("mode", ["imap", "imapfast"])
def test_gv_to_tigpair_chain(graph_check, tmp_path, mode):
    """fixed IndexMap -> host graph stage -> _original.gv identical to the committed one, whose
    tigpair_checkpoint.tsv was written by the reference's bin/makeTSVfile.py; and the reference demo's
    own (_original.gv, tigpair_checkpoint.tsv) pair"""
    base = str(tmp_path / "out")
    subprocess.check_call([graph_check, mode, os.path.join(GOLDEN, "tigpair_imap.tsv"),
                           os.path.join(GOLDEN, "tigpair_mult.tsv"), os.path.join(GOLDEN, "tigpair_lengths.tsv"),
                           base, "5", "0", "50", "10000", "0", "0.05", "100", "x"], stdout=subprocess.DEVNULL)
    gv = open(base + "_original.gv").read()
    assert gv == open(os.path.join(GOLDEN, "tigpair_original.gv")).read()
    assert gv.count("--") >= 4
    assert _make_tsv(base + "_original.gv", os.path.join(GOLDEN, "tigpair_draft_headers.fa")) == \
        open(os.path.join(GOLDEN, "tigpair_checkpoint.tsv")).read()
    demo_fa = tmp_path / "demo.fa"
    demo_fa.write_text(">1\nA\n>2\nA\n>3\nA\n")
    assert _make_tsv(os.path.join(GOLDEN, "arks_demo_original.gv"), str(demo_fa)) == \
        open(os.path.join(GOLDEN, "arks_demo.tigpair_checkpoint.tsv")).read()


def test_multiplicity_files_of_the_reference_demos(graph_check, tmp_path):
    """the -u parser (createIndexMultMap, Arcs.cpp:392-448) on the reference's own files: the arks demo's CSV
    (1085 barcodes, 55288 reads = the reads of the demo log) and the arks-long demo's TSV (the name decides the
    format); the --barcode-counts writer gives a CSV-sorted file back (count descending, barcode ascending)"""
    import shutil
    for name, fmt in (("arks_demo.test_reads_multiplicities.csv", "csv"), ("arks-long_demo.barcodeMultiplicityArcs.tsv", "tsv")):
        src = os.path.join(GOLDEN, name)
        rows = [ln.replace(",", "\t").split() for ln in open(src).read().split("\n") if ln]
        want = {b: int(m) for b, m in rows}
        out = tmp_path / f"counts_{fmt}.tsv"
        got = subprocess.check_output([graph_check, "mult", src, str(out)], text=True).split()
        assert [int(x) for x in got] == [len(rows), len(want), sum(want.values())]
        back = [ln.split("\t") for ln in out.read_text().split("\n") if ln]
        assert {b: int(m) for b, m in back} == want
        assert [(b, int(m)) for b, m in back] == sorted(want.items(), key=lambda kv: (-kv[1], kv[0]))
        # the same content under the other extension is read the other way: a CSV named .tsv has no white space
        # to split at, so the whole line is the "barcode" and the count is missing -- the reference then dies in
        # std::stoi; only the matching pairs of name and content are usable
    rows = [ln.split(",") for ln in open(os.path.join(GOLDEN, "arks_demo.test_reads_multiplicities.csv")).read().split("\n") if ln]
    assert len(rows) == 1085 and sum(int(m) for _, m in rows) == 55288     # the demo log's read count


def test_multiplicity_lines_of_unusual_shape(graph_check, tmp_path):
    """the -u parser takes the usual line apart itself and hands any other line to the calls the reference makes
    (stringstream >>, getline(',') and std::stoi, Arcs.cpp:404-427): leading blanks, extra fields, digits followed
    by letters, signs, ten digits, carriage returns, a barcode with a blank in a CSV, an empty CSV barcode; a
    line without a number ends the program as the reference's uncaught std::invalid_argument does"""
    tsv = tmp_path / "m.tsv"
    tsv.write_bytes(b"BX1\t12\n  BX2   34  extra\nBX3\t56abc\nBX4\t+7\nBX5\t-8\nBX6\t0012\nBX7\t1234567890\nBX8\t9\r\nBX1 13")
    want = {"BX1": 13, "BX2": 34, "BX3": 56, "BX4": 7, "BX5": -8, "BX6": 12, "BX7": 1234567890, "BX8": 9}
    got = subprocess.check_output([graph_check, "mult", str(tsv), str(tmp_path / "o1.tsv")], text=True).split()
    assert [int(x) for x in got] == [9, len(want), sum(want.values())]
    csv = tmp_path / "m.csv"
    csv.write_bytes(b"BY1,12\nBY2, 34\nBY3,56abc\n,78\nBY4 with space,5\nBY5,6\r\nBY1,1\n")
    want = {"BY1": 1, "BY2": 34, "BY3": 56, "BY4 with space": 5, "BY5": 6}
    out = subprocess.check_output([graph_check, "mult", str(csv), str(tmp_path / "o2.tsv")], text=True)
    assert "Please check your multiplicity file." in out          # the line with the empty barcode
    assert [int(x) for x in out.split()[-3:]] == [7, len(want), sum(want.values())]
    back = dict(ln.rsplit("\t", 1) for ln in (tmp_path / "o2.tsv").read_text().split("\n") if ln)
    assert {k: int(v) for k, v in back.items()} == want
    for bad in (b"BX1\t12\nBX2\n", b"BX1\t12\n\nBX3\t4\n", b"BX1\tabc\n"):
        tsv.write_bytes(bad)
        r = subprocess.run([graph_check, "mult", str(tsv), str(tmp_path / "o3.tsv")], capture_output=True)
        assert r.returncode != 0                                   # terminate: std::invalid_argument from stoi


def test_parallel_sort_of_the_number_based_stages(tmp_path):
    """graph_fast.hpp sorts its (barcode, contig) records and its contig-pair hits over the -t threads by sampled
    splitters (parallel_sort_by): the same order of keys as std::sort, every record kept, for skewed keys too"""
    src = tmp_path / "psort.cpp"
    src.write_text(r'''
#include "graph_fast.hpp"
#include <random>
struct R { uint64_t key; uint32_t payload; };
int main() {
	std::mt19937_64 rng(5);
	for (int mode = 0; mode < 4; ++mode)
		for (unsigned threads : { 1u, 2u, 5u, 16u, 100u }) {
			const size_t n = mode == 3 ? 70000 : 600000;
			std::vector<R> v(n);
			for (size_t i = 0; i < n; ++i) {
				uint64_t k = rng();
				if (mode == 1) k %= 7;                   // seven keys
				if (mode == 2) k = (k % 100 < 90) ? 42 : k; // one key holds most records
				v[i] = R{ k, (uint32_t)i };
			}
			std::vector<R> w = v;
			std::sort(w.begin(), w.end(), [](const R& a, const R& b) { return a.key < b.key; });
			arks_host::parallel_sort_by(v, [](const R& r) { return r.key; }, threads);
			uint64_t s1 = 0, s2 = 0;
			for (size_t i = 0; i < n; ++i) {
				if (v[i].key != w[i].key) { std::printf("order differs mode %d threads %u at %zu\n", mode, threads, i); return 1; }
				s1 += (uint64_t)v[i].payload * 2654435761ull; s2 += (uint64_t)w[i].payload * 2654435761ull;
			}
			if (s1 != s2) { std::printf("records lost mode %d threads %u\n", mode, threads); return 1; }
		}
	std::printf("ok\n");
	return 0;
}
''')
    exe = str(tmp_path / "psort")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(),
                           "-I" + os.path.join(ROOT, "arcs_amd", "host"), str(src), "-o", exe])
    assert subprocess.check_output([exe], text=True).strip() == "ok"
