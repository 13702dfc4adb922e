#!/usr/bin/env python3
"""Pins the .gv -> tigpair_checkpoint hand-off: a fixed synthetic IndexMap is pushed through the host
graph stage (arcs_amd/host/graph_check.cpp) to an _original.gv, and the REFERENCE's own pipeline
script bin/makeTSVfile.py (imported from /root/reference here; it cannot travel) turns that .gv into
the tigpair_checkpoint.tsv LINKS consumes.  Committed: the inputs (imap/mult/lengths/FASTA header
list), the .gv and the checkpoint file.  Also checks the reference demo pair
(Examples/arks_test-demo/output: _original.gv -> .tigpair_checkpoint.tsv) through the same script."""
import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/bin/makeTSVfile.py"


def ref_make_tsv(gv, out, fasta):
    spec = importlib.util.spec_from_file_location("ref_makeTSVfile", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.index2scaff_name.clear()
    m.links_numbering.clear()
    m.readGraphFile(gv)
    m.makeLinksNumbering(fasta)
    m.writeTSVFile(gv, out)


def main():
    tmp = tempfile.mkdtemp()
    exe = os.path.join(tmp, "graph_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "arcs_amd", "host"),
                           os.path.join(ROOT, "arcs_amd", "host", "graph_check.cpp"), "-o", exe])
    rng = np.random.Generator(np.random.PCG64(99))
    contigs = [str(i + 1) for i in range(30)]
    lengths = {c: int(rng.integers(600, 150000)) for c in contigs}
    rows, mult = [], {}
    for b in range(300):
        bc = "".join(rng.choice(list("ACGT"), size=16)) + "-1"
        mult[bc] = int(rng.integers(40, 400))
        base = int(rng.integers(len(contigs)))
        for t in range(int(rng.integers(1, 4))):
            ctg = contigs[(base + t) % len(contigs)]
            rows.append((bc, ctg, "H" if rng.random() < 0.5 else "T", int(rng.integers(3, 40))))
    open(os.path.join(HERE, "tigpair_imap.tsv"), "w").write("".join("%s\t%s\t%s\t%d\n" % r for r in rows))
    open(os.path.join(HERE, "tigpair_mult.tsv"), "w").write("".join(f"{b}\t{m}\n" for b, m in mult.items()))
    open(os.path.join(HERE, "tigpair_lengths.tsv"), "w").write("".join(f"{c}\t{l}\n" for c, l in lengths.items()))
    open(os.path.join(HERE, "tigpair_draft_headers.fa"), "w").write("".join(f">{c}\nA\n" for c in contigs))
    base = os.path.join(tmp, "out")
    subprocess.check_call([exe, "imap", os.path.join(HERE, "tigpair_imap.tsv"), os.path.join(HERE, "tigpair_mult.tsv"),
                           os.path.join(HERE, "tigpair_lengths.tsv"), base, "5", "0", "50", "10000", "0", "0.05",
                           "100", "x"], stdout=subprocess.DEVNULL)
    gv = os.path.join(HERE, "tigpair_original.gv")
    open(gv, "w").write(open(base + "_original.gv").read())
    ref_make_tsv(gv, os.path.join(HERE, "tigpair_checkpoint.tsv"), os.path.join(HERE, "tigpair_draft_headers.fa"))
    print(open(gv).read().count("--"), "edges;", len(open(os.path.join(HERE, "tigpair_checkpoint.tsv")).readlines()), "checkpoint lines")
    # the reference demo pair through the same script
    demo_gv = os.path.join(HERE, "arks_demo_original.gv")
    fa = os.path.join(tmp, "demo.fa")
    open(fa, "w").write(">1\nA\n>2\nA\n>3\nA\n")
    out = os.path.join(tmp, "demo.tsv")
    ref_make_tsv(demo_gv, out, fa)
    want = open("/root/reference/Examples/arks_test-demo/output/test_scaffolds_c5_m50-6000_k30_r0.05_e30000_z500.tigpair_checkpoint.tsv").read()
    assert open(out).read() == want
    open(os.path.join(HERE, "arks_demo.tigpair_checkpoint.tsv"), "w").write(want)
    print("demo .gv -> tigpair reproduces the committed demo checkpoint")


if __name__ == "__main__":
    main()
