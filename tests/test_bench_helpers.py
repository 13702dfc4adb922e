"""CPU-side pieces of bench.py and of the synthetic workloads: the read set's generation blocks and how they are
dealt to ranks, the FASTQ writer, the CPU port end to end from a gzipped FASTQ against the in-memory oracle."""
import gzip
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from arcs_amd import synth          # noqa: E402
from oracle import pyoracle as O    # noqa: E402


def test_read_blocks_and_their_dealing():
    import bench
    for pairs in (1, 7, 40, 41, 400000, 500_000_000):
        blocks = bench.read_blocks(pairs)
        assert len(blocks) == min(40, pairs)
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == pairs
        assert all(a + n == b for (a, n), (b, _) in zip(blocks, blocks[1:]))
        for world in (1, 2, 3, 8):
            shares = [bench.blocks_of_rank(len(blocks), r, world) for r in range(world)]
            assert shares[0][0] == 0 and shares[-1][1] == len(blocks)
            assert all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
            sizes = [hi - lo for lo, hi in shares]
            assert max(sizes) - min(sizes) <= 1
    # 8 ranks of the headline workload: five blocks of 12.5 M pairs each
    assert [bench.blocks_of_rank(40, r, 8) for r in (0, 7)] == [(0, 5), (35, 40)]
    assert bench.alg_bytes_per_window(60, 279, 161) == 2 * 279 / (8 * 161) + 15 + 4


def test_fastq_writer_and_the_port_from_a_file(tmp_path):
    contigs = synth.make_draft(1_500_000, seed=3)
    batch = synth.make_read_pairs(contigs, 2000, seed=4)
    text = synth.fastq_bytes(batch, first_pair=11)
    path = tmp_path / "reads.fq.gz"
    synth.write_gz_members(str(path), text.tobytes(), threads=4, member_bytes=70000)
    assert gzip.open(path).read() == text.tobytes()
    lines = text.tobytes().decode().split("\n")
    assert lines[0].startswith("@p000000011/1 BX:Z:") and lines[0].endswith("-1") and lines[2] == "+"
    assert len(lines[1]) == len(lines[3]) == 128 and len(lines[5]) == 151
    strs = synth.reads_to_strings(batch)
    assert lines[1] == strs[0] and lines[5] == strs[1]
    # mates of a pair whose names must not match carry different names
    ok = batch["pair_ok"].numpy()
    p = int(np.argmin(ok)) if (ok == 0).any() else None
    if p is not None:
        assert lines[8 * p].split()[0][:-2] != lines[8 * p + 4].split()[0][:-2]
    cs = synth.contigs_to_strings(contigs)
    ox = O.OracleIndex(60).build(O.contig_ends(cs))
    n, stored, st = O.map_fastq_gz(ox, str(path), 0.55, threads=3)
    a = np.concatenate([batch["ascii"].numpy(), np.zeros(1, np.uint8)])
    _, want_p, want = ox.map_pairs(a, batch["offsets"].numpy().astype(np.uint64)[:-1],
                                   batch["lens"].numpy().astype(np.uint32), 0.55, pair_ok=ok, threads=2)
    assert n == 2000 and stored == want["stored_pairs"] == int((want_p != 0).sum())
    assert st == {k: v for k, v in want.items() if k != "stored_pairs"}
