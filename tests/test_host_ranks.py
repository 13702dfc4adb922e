"""One process per GPU on the host side (arcs --ranks N, arcs_amd/host/rank_merge.hpp), without a GPU: the
results of 1, 2 and 3 ranks -- real processes, worker results through pipes, every rank with its own
barcode numbering -- merge into the same log, the same multiplicities and the same IndexMap, entry for
entry and in the same container order (Arcs/Arcs.cpp:1282-1285 is a commutative sum; the creation order
of the barcodes is that of a single-threaded run: by first stored pair)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ranks_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ranks") / "ranks_check")
    flags = os.environ.get("ARKS_TEST_CXXFLAGS", "").split()
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", *flags,
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "arcs_amd", "host"),
                           os.path.join(ROOT, "arcs_amd", "host", "ranks_check.cpp"), "-lz", "-ldl", "-o", exe])
    return exe


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("seed", [1, 7, 23])
def test_ranks_merge_like_one_process(ranks_check, seed, fused):
    outs = [subprocess.run([ranks_check, str(w), str(seed), str(fused)], capture_output=True, text=True, timeout=60)
            for w in (1, 2, 3, 5)]
    for o in outs:
        assert o.returncode == 0, o.stderr
    assert len({o.stdout for o in outs}) == 1           # world size does not show anywhere
    lines = outs[0].stdout.split("\n")
    # the IndexMap of both k against the scenario: counts per (barcode, contig end), plus the zero entry of
    # the opposite end that the reference adds after every file (Arcs.cpp:1304-1319)
    want = [{}, {}]
    first = [{}, {}]
    for ln in lines:
        if ln.startswith("PAIR "):
            _, f, bc, conreci, seq = ln.split()
            conreci, seq = int(conreci), int(seq)
            for ki in (0, 1):
                if ki == 1 and seq & 1:
                    continue
                end = (f"ctg{(conreci - 1) // 2 + 1}", "H" if (conreci - 1) % 2 == 0 else "T")
                want[ki].setdefault(bc, {})
                want[ki][bc][end] = want[ki][bc].get(end, 0) + 1
                first[ki][bc] = min(first[ki].get(bc, seq), seq)
    for ki in (0, 1):
        for bc, ends in want[ki].items():
            for (ctg, side) in list(ends):
                ends.setdefault((ctg, "T" if side == "H" else "H"), 0)
    got = [{}, {}]
    order = [[], []]
    for ln in lines:
        if ln.startswith("IMAP "):
            _, ki, bc, ctg, side, n = ln.split()
            got[int(ki)].setdefault(bc, {})[(ctg, side)] = int(n)
            if not order[int(ki)] or order[int(ki)][-1] != bc:
                order[int(ki)].append(bc)
    assert got == want
    # the number-based merge (graph_fast.hpp) holds the same content: its pairs and its TSV equal those of the
    # IndexMap above, whatever the number of ranks
    compact = [ln for ln in lines if ln.startswith("COMPACT ")]
    assert len(compact) == 4 and all(" same " in ln for ln in compact), compact
    # multiplicities of the fused mode are the per-file read counts summed; the log names every file once
    assert outs[0].stdout.count("Reading chrom reads") == (10 if fused else 5)
    assert "Stored read pairs: 0\n" in outs[0].stdout   # the empty file


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lanes_of_one_process_merge_like_a_map(ranks_check, seed):
    """merge_lane_entries (arcs --ranks N on one file, --index-sharded: the GPU lanes of one process share the barcode
    ids; each lane's IndexMap entries come sorted by key): counts add, the first stored pair is the earliest"""
    res = subprocess.run([ranks_check, "lanes", str(seed)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and res.stdout.strip() == "lanes ok", res.stdout + res.stderr
