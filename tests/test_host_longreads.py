"""arcs_amd/host/long_to_linked_pe.cpp (the arks-long feeder) against a Python restatement of
src/long-to-linked-pe.cpp:185-292, incl. the remainder pair and the barcode multiplicity file; and the
kseq-compatible reader on awkward FASTA/FASTQ input."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMP = str.maketrans("ACGTacgt", "TGCAtgca")     # N / n map to themselves


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bin") / "long-to-linked-pe")
    subprocess.check_call(["g++", "-O1", "-std=c++17", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(), "-pthread", "-I" + os.path.join(ROOT, "arcs_amd", "host"),
                           os.path.join(ROOT, "arcs_amd", "host", "long_to_linked_pe.cpp"), "-lz", "-ldl", "-o", out])
    return out


def rc(s):
    return s[::-1].translate(COMP)


def expected(records, l, m, fasta):
    """records: list of (id, seq, qual or '')"""
    out, bx = [], []
    sym = ">" if fasta else "@"
    for num, (rid, seq, qual) in enumerate(records):
        n, step = len(seq), 2 * l
        if step > n or m > n:
            continue
        bx.append(f"{num + 1}\t{(n // step + 1) * 2 if n % step else n // l}\n")
        def emit(k, s, q):
            out.append(f"{sym}{rid}_f{k} BX:Z:{num + 1}\n{s}\n" + ("" if fasta else f"+\n{q}\n"))
        k = 1
        for i in range(0, n - step + 1, step):
            emit(k, seq[i:i + l], qual[i:i + l] if qual else "#" * l)
            emit(k, rc(seq[i + l:i + 2 * l]), (qual[i + l:i + 2 * l] if qual else "#" * l)[::-1])
            k += 1
        rem = n % step
        if rem:
            cur = n - rem
            fwd = seq[cur:cur + l]
            emit(k, fwd, qual[cur:cur + l] if qual else "#" * len(fwd))
            emit(k, rc(seq[n - len(fwd):]), (qual[n - len(fwd):] if qual else "#" * len(fwd))[::-1])
    return "".join(out), "".join(bx)


@pytest.mark.parametrize("fastq,fasta_out", [(False, False), (True, False), (True, True)])
def test_long_to_linked_pe(exe, tmp_path, fastq, fasta_out):
    rng = np.random.Generator(np.random.PCG64(17))
    recs = []
    for i, n in enumerate([100, 499, 500, 501, 749, 750, 751, 1000, 1999, 2000, 2001, 2250, 2499, 2500, 5003, 1200]):
        seq = "".join(rng.choice(list("ACGTN"), size=n, p=[0.245] * 4 + [0.02]))
        qual = "".join(chr(33 + int(x)) for x in rng.integers(0, 40, size=n)) if fastq else ""
        recs.append((f"read{i}", seq, qual))
    # soft-masked (lower-case) stretches pass through as they are, complemented in their case: the reference opens its
    # reader with flags = LONG_MODE only (src/long-to-linked-pe.cpp:186-188), i.e. without btllib's FOLD_CASE, and
    # btllib's reverse_complement keeps the case (btllib itself is absent from this image: pinned to that reading)
    lower = "".join(rng.choice(list("ACGTacgtNn"), size=1777, p=[0.15] * 4 + [0.09] * 4 + [0.02, 0.02]))
    recs.append(("softmasked", lower, "".join(chr(33 + int(x)) for x in rng.integers(0, 40, size=1777)) if fastq else ""))
    path = tmp_path / ("reads.fq.gz" if fastq else "reads.fa.gz")
    with gzip.open(path, "wt") as f:
        for rid, seq, qual in recs:
            if fastq:
                f.write(f"@{rid} some comment\n{seq}\n+\n{qual}\n")
            else:
                f.write(f">{rid} some comment\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + "\n")
    for l, m in ((250, 2000), (250, 400), (100, 150)):
        bxf = tmp_path / f"bx_{l}_{m}.tsv"
        args = [exe, "-l", str(l), "-m", str(m), "-t", "4", "--bx", "-b", str(bxf)] + (["--fasta"] if fasta_out else [])
        got = subprocess.run(args + [str(path)], capture_output=True, text=True, check=True).stdout
        want, want_bx = expected(recs, l, m, fasta_out)
        assert got == want, (l, m)
        assert bxf.read_text() == want_bx
        bx2 = tmp_path / f"bxonly_{l}_{m}.tsv"
        only = subprocess.run([exe, "-l", str(l), "-m", str(m), "--bx-only", "-b", str(bx2), str(path)],
                              capture_output=True, text=True, check=True)
        assert only.stdout == "" and bx2.read_text() == want_bx


def test_span_and_dist_parameters(exe, tmp_path):
    recs = [(f"r{i}", "ACGT" * (n // 4), "") for i, n in enumerate([800, 1200, 4000, 8000, 12000])]
    path = tmp_path / "r.fa"
    path.write_text("".join(f">{r}\n{s}\n" for r, s, _ in recs))
    cfg = tmp_path / "p.tsv"
    subprocess.run([exe, "-l", "250", "-s", "-d", "-g", "1e4", "-f", str(cfg), str(path)], capture_output=True, check=True)
    total = sum(len(s) for _, s, _ in recs)
    lens = sorted(len(s) for _, s, _ in recs if len(s) > 1000)
    assert cfg.read_text() == f"span\t{int(total / 10000 * 0.25)}\nread_p50\t{(lens[1] + lens[2]) // 2}\n"
