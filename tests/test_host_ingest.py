"""arcs_amd/host/ingest.hpp (the pipelined ingest of `arcs --arks`: producers per file, packer pool)
against a Python restatement of the record-pair loop of chromiumRead (Arcs/Arcs.cpp:1185-1268) and of
the packed read layout; the result must not depend on thread count, batch size or gzip."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "arcs_amd", "host")
MASK = (1 << 64) - 1


@pytest.fixture(scope="module")
def exe(tmp_path_factory, arks):
    out = str(tmp_path_factory.mktemp("bin") / "ingest_check")
    libdir = os.path.join(ROOT, "arcs_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(), "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), "-I" + HOST, os.path.join(HOST, "ingest_check.cpp"),
                           "-L" + libdir, "-larks_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lz", "-ldl",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def fnv(h, data):
    for b in data:
        h = ((h ^ b) * 1099511628211) & MASK
    return h


FNV0 = 1469598103934665603
CODE = {c: i for i, c in enumerate("ACGT")}
CODE.update({c.lower(): i for c, i in list(CODE.items())})


def read_hash(seq, cls):
    """length, class, then per 32-base word: 2-bit codes MSB-first (u64) and invalid-base mask (u32,
    bit 31 = first base of the word) -- the layout of include/arks_hip.h"""
    h = fnv(FNV0, len(seq).to_bytes(4, "little") + bytes([cls]))
    for w0 in range(0, len(seq), 32):
        codes = nm = 0
        for p, ch in enumerate(seq[w0:w0 + 32]):
            if ch in CODE:
                codes |= CODE[ch] << (62 - 2 * p)
            else:
                nm |= 1 << (31 - p)
        h = fnv(h, codes.to_bytes(8, "little") + nm.to_bytes(4, "little"))
    return h


def strip_read_num(name):
    pos = name.rfind("/")
    if pos in (-1, 0, len(name) - 1) or not name[pos + 1].isdigit():
        return name
    return name[:pos]


def bx(comment):
    tag = comment.find("BX:Z:")
    if tag < 0:
        return ""
    end = comment.find(" ", tag)
    return comment[tag + 5:end] if end >= 0 else comment[tag + 5:]


def expected(records, mult, oracle, fused=False):
    """records: (name, comment, seq) of the well-formed records in file order; `None` = a record
    that ends the stream (kseq returns a negative length)"""
    c = dict(pairs=0, reads=0, unpaired=0, empty=0, invalid=0, gated=0, skipped_invalid=0)
    rh, ph, msgs = [], [], []
    i = 0
    while True:
        r1 = records[i] if i < len(records) else None
        r2 = records[i + 1] if r1 is not None and i + 1 < len(records) else None
        i += 2
        n1 = strip_read_num(r1[0]) if r1 else ""
        n2 = strip_read_num(r2[0]) if r2 else ""
        if n1 != n2:
            msgs.append(f"File contains unpaired reads: {n1} {n2}\n")
            c["unpaired"] += 1
        if r1 is None or r2 is None:
            break
        b1, b2 = bx(r1[1]), bx(r2[1])
        valid = False
        if not b1 or not b2:
            c["empty"] += 1
        else:
            valid = True if fused else b1 in mult
            if not valid:
                c["invalid"] += 1
        ok = n1 == n2 and valid and b1 == b2
        # the packers' class byte: bit 0 = checkReadSequence accepts, bit 1 = accepted and ACGT only (include/arks_hip.h)
        cls = [int(oracle.check_read_sequence(r[2])) for r in (r1, r2)]
        cls = [c | (2 if c and not set(r[2]) - set("ACGTacgt") else 0) for c, r in zip(cls, (r1, r2))]
        rh += [read_hash(r1[2], cls[0]), read_hash(r2[2], cls[1])]
        ph.append(fnv(fnv(FNV0, bytes([1])), b1.encode()) if ok else fnv(FNV0, bytes([0])))
        c["pairs"] += 1
        c["reads"] += 2
        if ok:
            c["gated"] += 1
            c["skipped_invalid"] += not (cls[0] and cls[1])
    d = FNV0
    for h in rh + ph:
        d = fnv(d, h.to_bytes(8, "little"))
    return c, f"{d:016x}", "".join(msgs)


def make_records(rng, n_pairs, barcodes):
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))
    recs = []
    for p in range(n_pairs):
        kind = int(rng.integers(0, 20))
        bc = barcodes[int(rng.integers(len(barcodes)))]
        name = f"read{p}"
        c1 = c2 = f"BX:Z:{bc}"
        if kind == 0:
            c1 = c2 = ""                                  # no comment at all
        elif kind == 1:
            c2 = "BX:Z:" + barcodes[(barcodes.index(bc) + 1) % len(barcodes)]   # mates disagree
        elif kind == 2:
            c1 = c2 = "BX:Z:NOTINTHEMAP-1"
        elif kind == 3:
            c1 = f"RX:Z:x BX:Z:{bc} QX:Z:y"              # tag in the middle
            c2 = f"BX:Z:{bc} more"
        s1, s2 = rnd(int(rng.choice([128, 151, 59, 33, 250]))), rnd(int(rng.choice([128, 151, 1, 64])))
        if kind == 4:
            s1 = s1[:10] + "N" + s1[11:]                  # one N: still under the 2 % rule for >= 51 b
        if kind == 5:
            s2 = "N" * 8 + s2[8:]
        if kind == 6:
            s1 = s1.lower()
        n1, n2 = name + "/1", name + "/2"
        if kind == 7:
            n2 = name + "x/2"                             # names differ
        if kind == 8:
            n1, n2 = name, name                           # no read number suffix
        recs.append((n1, c1, s1))
        recs.append((n2, c2, s2))
    return recs


def write_fastq(path, recs, tail=""):
    text = "".join(f"@{n}{' ' + c if c else ''}\n{s}\n+\n{'I' * len(s)}\n" for n, c, s in recs) + tail
    if path.endswith(".gz"):
        with gzip.open(path, "wt") as f:
            f.write(text)
    else:
        with open(path, "w") as f:
            f.write(text)


def run(exe, threads, batch, mult_path, files, sequential=False, extra_env=None):
    """sequential: ARKS_SEQUENTIAL_INGEST=1, every file through the kseq-compatible loop alone (no fast path)"""
    env = dict(os.environ, ARKS_SEQUENTIAL_INGEST="1") if sequential else dict(os.environ)
    env.update(extra_env or {})
    out = subprocess.check_output([exe, str(threads), str(batch), mult_path] + files, text=True, env=env)
    lines = [ln for ln in out.splitlines(keepends=True) if not ln.startswith(("prepass ", "mult\t"))]
    res, cur = [], None
    for ln in lines[1:]:
        if ln.startswith("file "):
            kv = dict(t.split("=") for t in ln.split()[2:])
            cur = dict(kv=kv, msgs="")
            res.append(cur)
        else:
            cur["msgs"] += ln
    return lines[0], res


def test_pipeline_matches_restatement(exe, oracle, tmp_path):
    rng = np.random.Generator(np.random.PCG64(11))
    barcodes = [f"{''.join('ACGT'[i] for i in rng.integers(0, 4, size=16))}-1" for _ in range(40)]
    mult = {b: int(rng.integers(1, 500)) for b in barcodes[:35]}      # five barcodes are not in the map
    mult_path = str(tmp_path / "mult.tsv")
    with open(mult_path, "w") as f:
        f.writelines(f"{b}\t{m}\n" for b, m in mult.items())
    files, want = [], []
    for fi, (n_pairs, ext, tail_kind) in enumerate([(700, ".fq", 0), (300, ".fq.gz", 1), (0, ".fastq", 0),
                                                    (257, ".fq.gz", 2), (120, ".fq", 3)]):
        recs = make_records(rng, n_pairs, barcodes)
        tail, exp_recs = "", list(recs)
        if tail_kind == 1:     # an odd record at the end: mate missing -> "unpaired" message, stream ends
            recs.append(("lonely/1", f"BX:Z:{barcodes[0]}", "ACGTACGTAC"))
            exp_recs = list(recs)
        if tail_kind == 2:     # quality shorter than the sequence: kseq returns -2, the loop stops
            tail = "@trunc/1 BX:Z:x\nACGTACGT\n+\nIII\n"
        path = str(tmp_path / f"reads{fi}{ext}")
        write_fastq(path, recs, tail)
        if tail_kind == 3:     # DOS line ends: kseq drops the carriage return of a line longer than one character
            data = open(path, "rb").read().replace(b"\n", b"\r\n")
            open(path, "wb").write(data)
        files.append(path)
        want.append(expected(exp_recs, mult, oracle))
    base = None
    for threads, batch, seql in [(1, 1 << 20, False), (2, 64, False), (8, 50, False), (3, 1, False), (6, 1000, False),
                                 (1, 1 << 20, True), (4, 50, True)]:
        head, res = run(exe, threads, batch, mult_path, files, sequential=seql)
        assert len(res) == len(files)
        for fi, r in enumerate(res):
            c, digest, msgs = want[fi]
            kv = r["kv"]
            got = dict(pairs=int(kv["pairs"]), reads=int(kv["reads"]), unpaired=int(kv["unpaired"]),
                       empty=int(kv["empty"]), invalid=int(kv["invalid"]), gated=int(kv["gated"]),
                       skipped_invalid=int(kv["skipped_invalid"]))
            assert got == c, (threads, batch, fi)
            assert kv["digest"] == digest, (threads, batch, fi)
            assert r["msgs"] == msgs, (threads, batch, fi)
        if batch <= 64:
            assert res[0]["kv"]["multibatch"] == "1"
        base = base or res
    # the thread split the front end reports
    head, _ = run(exe, 8, 100, mult_path, files)
    assert head.strip() == "threads producers=2 packers=6"   # five files, eight threads: producers only find lines
    head, _ = run(exe, 8, 100, mult_path, files, sequential=True)
    assert head.strip() == "threads producers=4 packers=4"   # ... without the fast path they parse


def test_fast_path_hands_over_to_the_kseq_loop(exe, tmp_path):
    """The fast path takes whole batches of 4-line FASTQ records and stops at the first group of four lines
    that is anything else; from there the kseq-compatible loop reads on.  Whatever the text and wherever the
    irregular stretch starts, the stage's output equals the sequential loop's (which the tests above pin to
    the restatement): digest of everything the GPU stage receives, counters, messages, fused pre-pass."""
    rng = np.random.Generator(np.random.PCG64(77))
    barcodes = [f"{''.join('ACGT'[i] for i in rng.integers(0, 4, size=16))}-1" for _ in range(12)]
    mult_path = str(tmp_path / "mult.tsv")
    with open(mult_path, "w") as f:
        f.writelines(f"{b}\t{7}\n" for b in barcodes[:10])

    def text_of(recs):
        return "".join(f"@{n}{' ' + c if c else ''}\n{s}\n+\n{'I' * len(s)}\n" for n, c, s in recs)

    def rec(i, m, seq="ACGTACGTAGGCTTAACG" * 4):
        return (f"r{i}/{m}", f"BX:Z:{barcodes[i % 12]}", seq)

    regular = [rec(i, m) for i in range(300) for m in (1, 2)]
    odd = {
        "multiline_seq": "@m/1 BX:Z:%s\nACGTACGT\nACGTAC\n+\nIIIIIIII\nIIIIII\n@m/2 BX:Z:%s\nACGT\n+\nIIII\n" % (barcodes[0], barcodes[0]),
        "fasta_records": ">f/1 BX:Z:%s\nACGTACGTACGT\n>f/2 BX:Z:%s\nACGTACGTAAAA\n" % (barcodes[1], barcodes[1]),
        "blank_lines": "\n\n",
        "qual_starts_with_at": "@q/1 BX:Z:%s\nACGTACGT\n+\n@IIIIIII\n@q/2 BX:Z:%s\nACGTACGG\n+q/2\n+IIIIIII\n" % (barcodes[2], barcodes[2]),
        "empty_read": "@e/1 BX:Z:%s\n\n+\n\n@e/2 BX:Z:%s\n\n+\n\n" % (barcodes[3], barcodes[3]),
        "long_quality": "@l/1 BX:Z:%s\nACGT\n+\nIIIIII\n@l/2 BX:Z:%s\nACGT\n+\nIIII\n" % (barcodes[4], barcodes[4]),
        "cr_lf": "@c/1 BX:Z:%s\r\nACGTACGT\r\n+\r\nIIIIIIII\r\n@c/2 BX:Z:%s\r\nACGTACGA\r\n+\r\nIIIIIIII\r\n" % (barcodes[5], barcodes[5]),
        "header_only_at": "@\nACGT\n+\nIIII\n@\nACGT\n+\nIIII\n",
        "tabs_in_header": "@t/1\tBX:Z:%s\tx\nACGTACGT\n+\nIIIIIIII\n@t/2\tBX:Z:%s\nACGTACGA\n+\nIIIIIIII\n" % (barcodes[6], barcodes[6]),
    }
    files = []
    for name, chunk in odd.items():
        for where in (0, 7, 150, 300):          # irregular stretch at the start, early, in the middle, at the end
            path = str(tmp_path / f"{name}_{where}.fq")
            with open(path, "w") as f:
                f.write(text_of(regular[:2 * where]) + chunk + text_of(regular[2 * where:]))
            files.append(path)
    nonl = str(tmp_path / "no_final_newline.fq")
    with open(nonl, "w") as f:
        f.write(text_of(regular)[:-1])
    files.append(nonl)
    oddcount = str(tmp_path / "odd_count.fq")
    with open(oddcount, "w") as f:
        f.write(text_of(regular[:-1]))
    files.append(oddcount)
    # the same texts as BGZF files (members of 3000 bytes: they end anywhere in a line)
    zfiles = []
    for path in files:
        zfiles.append(path + ".bgzf.gz")
        write_bgzf(zfiles[-1], open(path, "rb").read(), block=3000)
    # ... and as ordinary gzip files (decoded in chunks of a few kilobytes by several threads, pgzip.hpp)
    gfiles = []
    for path in files:
        gfiles.append(path + ".gz")
        with open(gfiles[-1], "wb") as f:
            f.write(gzip.compress(open(path, "rb").read(), 6))
    import re
    for mp in (mult_path, "-"):                  # with a multiplicity file, and fused
        for threads, batch in ((1, 1 << 20), (6, 16), (3, 100)):
            b = subprocess.check_output([exe, str(threads), str(batch), mp] + files, text=True,
                                        env=dict(os.environ, ARKS_SEQUENTIAL_INGEST="1"))
            # (the first line is the thread split; how many batches a file made is not part of the result)
            b = re.sub(r" multibatch=\d", "", "\n".join(b.split("\n")[1:]))
            # whole files in one stretch, and stretches of a few thousand bytes (the hand-over then falls into any
            # stretch, and the text between stretches is carried over or re-read)
            for src, env in ((files, {}), (files, {"ARKS_STRETCH_BYTES": "9000"}), (zfiles, {}),
                             (zfiles, {"ARKS_STRETCH_BYTES": "9000"}), (gfiles, {"ARKS_PGZIP": "1"}),
                             (gfiles, {"ARKS_PGZIP": "1", "ARKS_PGZIP_CHUNK": "4096"}), (gfiles, {})):
                a = subprocess.check_output([exe, str(threads), str(batch), mp] + src, text=True, env=dict(os.environ, **env))
                a = re.sub(r" multibatch=\d", "", "\n".join(a.split("\n")[1:]))
                assert a == b, (mp, threads, batch, env, src[0][-8:])
    # and the fast path did take most batches of a mostly regular file (multibatch = more than one batch seen)
    _, res = run(exe, 4, 16, mult_path, [files[2]])
    assert res[0]["kv"]["multibatch"] == "1" and int(res[0]["kv"]["pairs"]) >= 300


def test_fused_barcode_prepass(exe, oracle, tmp_path):
    """no multiplicity file: the pre-pass of readBarcodes (Arcs.cpp:481-547) rides along with the pair
    loop -- reads per barcode (every tagged record, also of pairs that fail the gate; an empty barcode
    is a key too), the summaries its progress lines are rebuilt from, and the gate (every tagged mate
    is in the map by construction)"""
    rng = np.random.Generator(np.random.PCG64(12))
    barcodes = [f"{''.join('ACGT'[i] for i in rng.integers(0, 4, size=16))}-1" for _ in range(25)]
    files, want, all_recs = [], [], []
    for fi, n_pairs in enumerate((400, 150)):
        recs = make_records(rng, n_pairs, barcodes)
        if fi == 0:   # comments without a tag before the first tagged record, an empty barcode, an odd record
            recs = [("lead1/1", "RG:Z:a", "ACGTACGTAGCT"), ("lead1/2", "RG:Z:a", "ACGTACGTAGCTAA")] + recs
            recs += [("eb/1", "BX:Z: x", "ACGTACGTAC"), ("eb/2", "BX:Z: x", "ACGTACGTAC"), ("odd/1", "BX:Z:" + barcodes[0], "ACGT")]
        path = str(tmp_path / f"f{fi}.fq")
        write_fastq(path, recs)
        files.append(path)
        all_recs.append(recs)
        want.append(expected(recs, {}, oracle, fused=True))
    counts = {}
    for recs in all_recs:
        for (n, c, s) in recs:
            if "BX:Z:" in c:
                counts[bx(c)] = counts.get(bx(c), 0) + 1
    for threads, batch in [(1, 1 << 20), (4, 37)]:
        out = subprocess.check_output([exe, str(threads), str(batch), "-"] + files, text=True)
        got_mult = {ln.split("\t")[1]: int(ln.split("\t")[2]) for ln in out.splitlines() if ln.startswith("mult\t")}
        assert got_mult == counts
        assert "" in got_mult                      # "BX:Z: x": the empty barcode is counted like any other
        pre = [ln for ln in out.splitlines() if ln.startswith("prepass ")]
        assert pre[0].startswith(f"prepass 0 total={sum(1 for r in all_recs[0] if 'BX:Z:' in r[1])} lead=2 zero_len=0")
        _, res = run(exe, threads, batch, "-", files)
        for fi, r in enumerate(res):
            c, digest, msgs = want[fi]
            kv = r["kv"]
            assert int(kv["invalid"]) == 0 and int(kv["gated"]) == c["gated"] and int(kv["empty"]) == c["empty"]
            assert kv["digest"] == digest and r["msgs"] == msgs


def write_bgzf(path, data, block=0xFF00, level=6):
    """BGZF as bgzip / htslib write it (SAM specification 4.1): independent gzip members of <= 64 KiB with a
    'BC' extra field holding the member size, and the 28-byte empty member at the end"""
    import struct
    import zlib
    def member(chunk):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            f.write(member(data[i:i + block]))
        f.write(member(b""))


def test_bgzf_parallel_inflate(exe, oracle, tmp_path):
    """a bgzip'ed FASTQ is cut into its members and inflated by several threads (bgzf.hpp): same records
    as the plain file, for every thread count; a damaged member ends the stream like a truncated gzip"""
    rng = np.random.Generator(np.random.PCG64(13))
    barcodes = [f"{''.join('ACGT'[i] for i in rng.integers(0, 4, size=16))}-1" for _ in range(30)]
    mult = {b: 10 for b in barcodes}
    mult_path = str(tmp_path / "mult.tsv")
    with open(mult_path, "w") as f:
        f.writelines(f"{b}\t{m}\n" for b, m in mult.items())
    recs = make_records(rng, 3000, barcodes)
    plain = str(tmp_path / "reads.fq")
    write_fastq(plain, recs)
    data = open(plain, "rb").read()
    assert len(data) > 20 * 0xFF00                      # dozens of members
    bg = str(tmp_path / "reads_bgzf.fq.gz")
    write_bgzf(bg, data)
    assert gzip.open(bg, "rb").read() == data           # a valid multi-member gzip file for anyone else
    small = str(tmp_path / "reads_small_blocks.fq.gz")
    write_bgzf(small, data, block=777)                  # members that end in the middle of lines
    want = expected(recs, mult, oracle)
    ref = None
    for threads in (1, 2, 5, 16):
        _, res = run(exe, threads, 500, mult_path, [plain, bg, small])
        for r in res:
            kv = r["kv"]
            assert kv["digest"] == want[1] and int(kv["pairs"]) == want[0]["pairs"] and r["msgs"] == want[2], threads
    # damage one member in the middle: the records before it arrive, the stream ends there
    blob = bytearray(open(bg, "rb").read())
    blob[len(blob) // 2] ^= 0x5A
    bad = str(tmp_path / "damaged.fq.gz")
    open(bad, "wb").write(bytes(blob))
    _, res = run(exe, 4, 500, mult_path, [bad])
    assert 0 < int(res[0]["kv"]["pairs"]) < want[0]["pairs"]
    # the members are inflated a stretch of text at a time by the scanning threads (BgzfStretches): stretches much
    # shorter than the file (the text a stretch leaves over is carried into the next), with libdeflate and with
    # zlib; and the damaged file gives what the sequential reader gives, whichever stretch the damage falls into
    _, seq_bad = run(exe, 4, 500, mult_path, [bad], sequential=True)
    for env in ({}, {"ARKS_ZLIB_INFLATE": "1"}):
        for stretch in (4096, 70001, 300000):
            e = dict(env, ARKS_STRETCH_BYTES=str(stretch))
            for threads in (1, 3, 8):
                _, res = run(exe, threads, 100, mult_path, [bg, small, bad], extra_env=e)
                for r in res[:2]:
                    kv = r["kv"]
                    assert kv["digest"] == want[1] and int(kv["pairs"]) == want[0]["pairs"] and r["msgs"] == want[2], (e, threads)
                assert res[2]["kv"]["digest"] == seq_bad[0]["kv"]["digest"] and res[2]["kv"]["pairs"] == seq_bad[0]["kv"]["pairs"], (e, threads)
                assert res[2]["msgs"] == seq_bad[0]["msgs"]


# ---- fast_inflate.hpp / crc32_fold.hpp ---------------------------------------------------------------

@pytest.fixture(scope="module")
def inflate_check(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bin") / "inflate_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(), "-I" + HOST, os.path.join(HOST, "inflate_check.cpp"), "-lz", "-ldl",
                           "-o", out])
    return out


def _gz(data, level=6, strategy=0, mem=8):
    import zlib
    c = zlib.compressobj(level, zlib.DEFLATED, 31, mem, strategy)
    return c.compress(data) + c.flush()


def test_fast_inflate_equals_zlib(inflate_check, tmp_path):
    """GzInflater against zlib's gzread, byte for byte: FASTQ text at every level, the three block types
    (stored / fixed / dynamic), codes longer than the primary tables, runs and short periods, members back
    to back (an empty one among them), every optional header field, trailing garbage; read sizes 1 .. 256 Ki"""
    import zlib
    rng = np.random.Generator(np.random.PCG64(9))
    fastq = "".join(
        f"@r{i}/1 BX:Z:{''.join(rng.choice(list('ACGT'), size=16))}-1\n{''.join(rng.choice(list('ACGTN'), size=int(rng.integers(50, 160)), p=[.24, .24, .24, .24, .04]))}\n+\n"
        f"{''.join(rng.choice(list('FFFFF:,#'), size=int(rng.integers(50, 160))))}\n" for i in range(6000)).encode()
    skew = bytes(rng.geometric(0.08, size=400000).clip(1, 255).astype(np.uint8))       # long Huffman codes
    noise = rng.integers(0, 256, size=300000, dtype=np.uint8).tobytes()                  # stored blocks
    runs = b"".join(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 700)) for _ in range(3000))
    periods = b"".join((bytes(rng.integers(65, 91, size=int(p), dtype=np.uint8)) * 400)[:int(rng.integers(1, 1500))]
                       for p in rng.integers(1, 12, size=2000))
    cases = {}
    for lvl in range(0, 10):
        cases[f"fastq{lvl}"] = _gz(fastq, lvl)
    cases["fixed"] = _gz(fastq[:200000], 6, zlib.Z_FIXED)
    cases["huffman_only"] = _gz(fastq[:200000] + skew, 6, zlib.Z_HUFFMAN_ONLY)
    cases["rle"] = _gz(runs + fastq[:100000], 6, zlib.Z_RLE)
    cases["skew"] = _gz(skew, 9)
    cases["noise"] = _gz(noise, 6)
    cases["runs"] = _gz(runs, 9)
    cases["periods"] = _gz(periods, 9, mem=9)
    cases["tiny"] = _gz(b"A", 6)
    cases["empty_member"] = _gz(b"", 6)
    cases["members"] = _gz(fastq[:70000], 1) + _gz(b"", 6) + _gz(noise[:70000], 6) + _gz(runs[:99999], 9) + _gz(b"x", 6)
    cases["garbage_after"] = _gz(fastq[:50000], 6) + b"\0\0\0not gzip"
    def flushed(data, mode, step=60000):             # as pigz writes: a flush (an empty stored block) every so often
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        return b"".join(c.compress(data[i:i + step]) + c.flush(mode) for i in range(0, len(data), step)) + c.flush()
    cases["sync_flushed"] = flushed(fastq, zlib.Z_SYNC_FLUSH)
    cases["full_flushed"] = flushed(fastq, zlib.Z_FULL_FLUSH)
    third = len(fastq) // 3
    cases["lanes_cat"] = _gz(fastq[:third], 6) + _gz(fastq[third:2 * third], 4) + _gz(fastq[2 * third:], 9)   # cat a.gz b.gz c.gz
    cases["blocked"] = b"".join(_gz(fastq[i:i + 3000], 6) for i in range(0, 300000, 3000))                    # a member per 3 KB
    # every optional header field (RFC 1952): FEXTRA, FNAME, FCOMMENT, FHCRC
    body = _gz(fastq[:30000], 6)[10:]
    hdr = bytes([0x1f, 0x8b, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 3]) + bytes([5, 0]) + b"EXTRA" + b"name.fq\0" + b"a comment\0"
    cases["all_header_fields"] = hdr + (zlib.crc32(hdr) & 0xffff).to_bytes(2, "little") + body
    for name, blob in cases.items():
        path = tmp_path / (name + ".gz")
        path.write_bytes(blob)
        for chunk in ((1 << 18, 1, 7, 4096) if len(blob) < 5000 else (1 << 18, 4099)):
            out = subprocess.run([inflate_check, str(path), str(chunk)], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0 and out.stdout.startswith("same "), (name, chunk, out.stdout)
        # the same stream with its blocks decoded by several threads (pgzip.hpp: speculative block starts, 16-bit
        # symbols with markers for the unknown window, hand-over to the sequential inflater): chunks of a few
        # kilobytes put block searches, chain breaks and the hand-over everywhere in these files
        for threads, pchunk, per in ((3, 4096, 5), (2, 20011, 4), (4, 1 << 20, 8)):
            out = subprocess.run([inflate_check, str(path), str(1 << 18), "pgz", str(threads), str(pchunk), str(per)],
                                 capture_output=True, text=True, timeout=120)
            assert out.returncode == 0 and "pgz same " in out.stdout, (name, threads, pchunk, out.stdout)
            if name == "lanes_cat" and pchunk < (1 << 20):
                # every member of a file of several is decoded on the parallel path, its last block and trailer included
                assert int(out.stdout.split("parallel ")[1].split()[0]) == len(fastq), (name, pchunk, out.stdout)
            if name.startswith("fastq") and name != "fastq0" and pchunk < (1 << 20):
                # ... and on FASTQ text most of the file does come from the parallel path
                par = int(out.stdout.split("parallel ")[1].split()[0])
                assert par > len(fastq) // 2, (name, pchunk, out.stdout)
    assert zlib.decompress(cases["all_header_fields"], 31) == fastq[:30000]


def test_parallel_gzip_output_is_bounded(inflate_check, tmp_path):
    """A highly compressible .gz (ratio ~ 1000 : 1 -- low-complexity or synthetic reads): the text one chunk / one
    stretch of the several-thread decoder produces is capped (pgzip.hpp: kChunkOutCap, kStretchOutCap; here scaled
    down by ARKS_PGZIP_OUT_CAP), chunks beyond the cap are taken by the next stretch, and the bytes are zlib's
    (64 chunks of 1 MB could expand to > 4 GiB, past the record splitter's 32-bit offsets)."""
    import zlib
    rng = np.random.Generator(np.random.PCG64(10))
    rec = b"@r BX:Z:AAAAAAAAAAAAAAAA-1\n" + b"A" * 150 + b"\n+\n" + b"F" * 150 + b"\n"
    text = rec * 40000 + bytes(rng.integers(65, 70, size=200000, dtype=np.uint8)) + rec * 40000     # 26 MB -> ~ 100 KB
    blob = _gz(text, 6)
    assert len(text) > 100 * len(blob)
    path = tmp_path / "high_ratio.gz"
    path.write_bytes(blob)
    for cap in ("65536", "1000000", None):
        env = dict(os.environ)
        if cap:
            env["ARKS_PGZIP_OUT_CAP"] = cap
        for threads, pchunk, per in ((3, 4096, 8), (4, 16384, 16)):
            out = subprocess.run([inflate_check, str(path), str(1 << 18), "pgz", str(threads), str(pchunk), str(per)],
                                 capture_output=True, text=True, timeout=300, env=env)
            assert out.returncode == 0 and "pgz same " in out.stdout, (cap, threads, pchunk, out.stdout, out.stderr[-500:])


def test_fast_inflate_damaged_input(inflate_check, tmp_path):
    """truncated and corrupted files: no crash, no hang, a truncated file delivers exactly what zlib delivers,
    and a failure is reported whenever zlib reports one"""
    rng = np.random.Generator(np.random.PCG64(10))
    data = ("".join(f"@r{i}\n{''.join(rng.choice(list('ACGT'), size=100))}\n+\n{'F' * 100}\n" for i in range(3000))).encode()
    blob = _gz(data, 6) + _gz(data[:5000], 1)
    # (the reader hands a file to GzInflater only when it starts with the three bytes 1f 8b 08)
    cuts = [3, 4, 9, 10, 11, 17, 100, len(blob) // 3, len(blob) // 2, len(blob) - 9, len(blob) - 8, len(blob) - 1]
    n_err = 0
    for i, cut in enumerate(cuts):
        path = tmp_path / f"cut{i}.gz"
        path.write_bytes(blob[:cut])
        out = subprocess.run([inflate_check, str(path), str(1 << 18), "pgz", "3", "9001", "4"], capture_output=True, text=True, timeout=60)
        assert out.stdout.split()[0] in ("same", "DIFFERENT"), out.stdout          # ran to the end
        assert "pgz same " in out.stdout, (cut, out.stdout)     # several threads deliver what the one inflater delivers
        rc = out.stdout.split("rc ")[1].split()[0].split("/")
        # gzread's return value calls a truncated stream a plain end of file (only gzerror tells); what counts for
        # the reader is that the same bytes were delivered before it
        sizes = out.stdout.split(" bytes (zlib ")
        assert sizes[0].split()[-1] == sizes[1].split(")")[0], (cut, out.stdout)
        assert int(rc[1]) >= 0 or int(rc[0]) < 0
        n_err += int(rc[0]) < 0
    assert n_err >= 8
    for i in range(60):
        bad = bytearray(blob)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(8))
        path = tmp_path / f"flip{i}.gz"
        path.write_bytes(bytes(bad))
        out = subprocess.run([inflate_check, str(path), str(1 << 18), "pgz", "3", "9001", "4"], capture_output=True, text=True, timeout=60)
        assert out.stdout.split()[0] in ("same", "DIFFERENT"), (pos, out.stdout, out.stderr)
        assert "pgz same " in out.stdout, (pos, out.stdout)
        rc = out.stdout.split("rc ")[1].split()[0].split("/")
        if int(rc[1]) < 0:                     # zlib saw the damage: so must we (CRC-32 and length are checked)
            assert int(rc[0]) < 0, (pos, out.stdout)


def test_crc32_fold_equals_zlib(tmp_path):
    src = tmp_path / "crc_check.cpp"
    src.write_text(r'''
#include "crc32_fold.hpp"
#include <cstdio>
#include <vector>
int main() {
	std::vector<unsigned char> b(1 << 22);
	unsigned s = 12345;
	for (auto& x : b) { s = s * 1664525u + 1013904223u; x = (unsigned char)(s >> 24); }
	int bad = 0;
	for (int t = 0; t < 30000; ++t) {
		s = s * 1664525u + 1013904223u; const size_t off = s % 1000;
		s = s * 1664525u + 1013904223u; const size_t len = t < 300 ? (size_t)t : s % 9000;
		s = s * 1664525u + 1013904223u; const uint32_t init = (t % 3) ? s : 0;
		bad += (uint32_t)crc32(init, b.data() + off, (uInt)len) != arks_host::crc32_fast(init, b.data() + off, len);
	}
	bad += (uint32_t)crc32(0, b.data(), (uInt)b.size()) != arks_host::crc32_fast(0, b.data(), b.size());
	std::printf("bad=%d\n", bad);
	return bad != 0;
}
''')
    exe = str(tmp_path / "crc_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", *os.environ.get("ARKS_TEST_CXXFLAGS", "").split(), "-I" + HOST, str(src), "-lz", "-ldl", "-o", exe])
    assert subprocess.run([exe], capture_output=True, text=True).stdout.strip() == "bad=0"


def test_inflate_thread_same_stream(exe, tmp_path):
    """ARKS_INFLATE_THREAD=1 (InflateAhead: the .gz inflate on its own thread) delivers the same records"""
    rng = np.random.Generator(np.random.PCG64(21))
    path = str(tmp_path / "reads.fq.gz")
    with gzip.open(path, "wt", compresslevel=4) as f:
        for i in range(40000):
            s1 = "".join(rng.choice(list("ACGTN"), size=int(rng.integers(30, 200)), p=[.245, .245, .245, .245, .02]))
            f.write(f"@r{i}/1 BX:Z:BC{i // 50:06d}-1\n{s1}\n+\n{'F' * len(s1)}\n@r{i}/2 BX:Z:BC{i // 50:06d}-1\n{s1[::-1]}\n+\n{'F' * len(s1)}\n")
    outs = []
    for env in ({}, {"ARKS_INFLATE_THREAD": "1"}, {"ARKS_ZLIB_INFLATE": "1"}):
        r = subprocess.run([exe, "4", "3000", "-", path], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and "pairs=40000" in outs[0]


def test_reads_from_a_pipe(exe, tmp_path):
    """/dev/stdin fed through a real pipe (how arcs-make runs arks-long, bin/arcs-make:305-313), plain and
    gzip'ed: the reader must not look at the first bytes of a stream it cannot rewind"""
    rng = np.random.Generator(np.random.PCG64(22))
    text = "".join(f"@r{i}/1 BX:Z:BC{i // 20:05d}-1\n{s}\n+\n{'F' * len(s)}\n@r{i}/2 BX:Z:BC{i // 20:05d}-1\n{s[::-1]}\n+\n{'F' * len(s)}\n"
                   for i, s in ((i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(40, 160))))) for i in range(3000))).encode()
    plain = tmp_path / "reads.fq"
    plain.write_bytes(text)
    want = subprocess.run([exe, "4", "700", "-", str(plain)], capture_output=True, text=True, timeout=120).stdout
    assert "pairs=3000" in want
    for blob in (text, gzip.compress(text, 4)):
        p = subprocess.Popen([exe, "4", "700", "-", "/dev/stdin"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=False)
        out, _ = p.communicate(blob, timeout=120)
        assert p.returncode == 0 and out.decode() == want


def test_helpdesk_survives_a_throwing_body(exe):
    """a parallel_for body that throws (std::bad_alloc on a helper or on the caller): every body still counts as
    done, the caller gets the exception, the desk keeps working"""
    for helpers in (0, 1, 6):
        r = subprocess.run([exe, "helpdesk-throw", str(helpers)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (helpers, r.stdout, r.stderr[-500:])
