"""CPU-side checks of the drop-in boundary: libarks_hip.so loads, exports every symbol that
include/arks_hip.h declares, refuses to compute without a gfx950 device (no CPU fallback), and its
host-side helpers (packing, word offsets, end cut-off) agree with the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    # the boundary (arks_hip.h) and the diagnostics header beside it (arks_hip_debug.h: tests / profiling only)
    text = open(os.path.join(ROOT, "include", "arks_hip.h")).read()
    assert "arks_debug_" not in text
    text += open(os.path.join(ROOT, "include", "arks_hip_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arks_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree(arks):
    from arcs_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    assert sorted(_lib.SYMBOLS) == names  # the ctypes table binds exactly the declared ABI
    out = subprocess.check_output(["nm", "-D", "--defined-only", arks.lib_path()]).decode()
    exported = set(re.findall(r" T (arks_[a-z0-9_]+)", out))
    assert set(names) <= exported
    assert arks.lib().arks_abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "arks_hip.h")).read()
    assert int(re.search(r"#define\s+ARKS_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_driver_entry_build():
    """__graft_entry__.build() is what the driver runs each round: it must survive an ABI bump (round 5's did not:
    a literal version number in it went stale)"""
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-2000:]


def test_library_reads_no_environment(arks):
    """the release build has no getenv in it: index layout and launch shape come through arks_build_options
    (arks_index_build_ex) and arks_debug_set_medium_blocks, never from the host process's environment"""
    out = subprocess.check_output(["nm", "-D", "--undefined-only", arks.lib_path()]).decode()
    assert not re.search(r"\b(secure_)?getenv\b", out), [l for l in out.splitlines() if "getenv" in l]
    for f in ("arks_capi.hip", "arks_map.hip", "arks_build.hip", "arks_shard.hip", "arks_imap.hip", "arks_exchange.hpp"):
        text = open(os.path.join(ROOT, "arcs_amd", "csrc", f)).read()
        text = re.sub(r"#ifdef ARKS_DEBUG_KNOBS.*?#e(lse|ndif)", "", text, flags=re.S)
        assert "getenv" not in text, f


def test_build_options_are_validated(arks):
    """arks_index_build_ex refuses nonsense before it touches a device"""
    from arcs_amd._lib import BuildOptions
    L = arks.lib()
    h = C.c_void_p()
    data = np.frombuffer(b"ACGT" * 40 + b"\0", dtype=np.uint8)
    offs, lens = np.zeros(1, np.uint64), np.array([160], np.uint32)
    def call(**kw):
        o = BuildOptions(); o.struct_size = C.sizeof(BuildOptions)
        for f, v in kw.items():
            setattr(o, f, v)
        return L.arks_index_build_ex(C.byref(h), 30, data.ctypes.data, offs.ctypes.data, lens.ctypes.data, 1, 0, C.byref(o), None)
    for bad in (dict(index_kind=7), dict(index_kind=-1), dict(heavy_over=1), dict(heavy_over=9), dict(minimizer_len=40),
                dict(fallback_load_inv=3), dict(struct_size=4), dict(n_shards=2, seed_ranks=2), dict(shard=2, n_shards=2)):
        assert call(**bad) == 6, bad                      # ARKS_ERR_BAD_ARG
    assert call() in (0, 5)                               # fine: builds (GPU box) or ARKS_ERR_NO_DEVICE (here)
    if h:
        L.arks_index_free(h)


def test_library_has_gfx950_code_object(arks):
    """the .so carries a gfx950 code object (what the GPU box will load)"""
    data = open(arks.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_key_bytes_and_cutoff(arks, oracle):
    for k in range(4, 97):
        assert arks.key_bytes(k) == oracle.key_bytes(k)
    for L in (0, 499, 500, 501, 59999, 60000, 60001, 61001, 1000000):
        for e in (0, 1000, 30000):
            assert arks.end_cutoff(L, 500, e) == oracle.end_cutoff(L, 500, e)
    s = ["A" * 499, "C" * 501, "G" * 70000]
    assert arks.contig_ends(s) == oracle.contig_ends(s)


def _py_pack(seq):
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    nw = (len(seq) + 31) // 32
    codes = [0] * nw
    mask = [0] * nw
    for i, ch in enumerate(seq):
        c = code.get(ch.upper())
        if c is None:
            mask[i // 32] |= 1 << (31 - i % 32)
        else:
            codes[i // 32] |= c << (62 - 2 * (i % 32))
    return codes, mask


def test_host_packer(arks, oracle):
    rng = np.random.Generator(np.random.PCG64(1))
    seqs = ["", "A", "ACGT" * 8, "ACGT" * 8 + "T", "acgtnNRyx-" * 7]
    for _ in range(30):
        L = int(rng.integers(1, 400))
        seqs.append("".join(rng.choice(list("ACGTacgtNn.R"), size=L,
                                       p=np.array([20] * 8 + [1, 1, 1, 1]) / 164.0)))
    p = arks.pack_reads_host(seqs)
    assert p["word_off"][0] == 0
    for i, s in enumerate(seqs):
        w0, w1 = int(p["word_off"][i]), int(p["word_off"][i + 1])
        assert w1 - w0 == (len(s) + 31) // 32
        codes, mask = _py_pack(s)
        assert [int(x) for x in p["codes"][w0:w1]] == codes
        assert [int(x) for x in p["nmask"][w0:w1]] == mask
        if len(s):
            assert bool(p["read_class"][i]) == oracle.check_read_sequence(s), s
    # the packed code words ARE the reference key bytes: window 0 of a 32-mer == word 0
    s = "".join(rng.choice(list("ACGT"), size=64))
    p = arks.pack_reads_host([s])
    k32 = int(p["codes"][0]).to_bytes(8, "big")
    fw = oracle.key(s[:32] + "T" * 0, 0, 32)
    rc = s[:32][::-1].translate(str.maketrans("ACGT", "TGCA"))
    assert fw == min(k32, int(arks.pack_reads_host([rc])["codes"][0]).to_bytes(8, "big"))


def test_no_device_is_loud(arks):
    """without a gfx950 device the compute entry points return ARKS_ERR_NO_DEVICE -- there is no
    CPU path inside the product"""
    if arks.device_count() > 0:
        pytest.skip("a gfx950 device is present")
    with pytest.raises(arks.ArksError) as e:
        arks.ArksIndex.build(["ACGT" * 100], 20)
    assert e.value.status == 5
    L = arks.lib()
    h = C.c_void_p()
    assert L.arks_imap_create(C.byref(h), 16, 0) == 5
    assert L.arks_index_build(C.byref(h), 3, None, None, None, 0, 0, None) == 1   # bad k first
    assert L.arks_index_build(C.byref(h), 10, None, None, None, 0, 0, None) == 1
    assert L.arks_index_build(C.byref(h), 97, None, None, None, 0, 0, None) == 2
    assert b"gfx950" in L.arks_strerror(5)


def test_product_does_not_reach_into_oracle():
    """the package and the C ABI never import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "arcs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "synth.py", os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "arcs_amd", "lib", "libarks_hip.so")]).decode()
    assert "arks_oracle" not in out and "arks_ref" not in out


def test_shard_of_ends_is_host_only_and_balanced(arks):
    """arks_shard_of_ends needs no device: head and tail of a contig stay together, the shards differ by
    less than one contig whatever the order of the draft, n_shards = 1 puts everything in shard 0"""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(3))
    for pattern in ("cycle", "sorted", "random"):
        per_contig = {"cycle": np.tile([20000, 50000, 100000, 230000], 50),
                      "sorted": np.sort(rng.integers(500, 300000, size=200))[::-1],
                      "random": rng.integers(500, 300000, size=201)}[pattern]
        lens = np.repeat(np.minimum(per_contig // 2, 30000), 2).astype(np.uint32)
        for n in (1, 2, 3, 8):
            owner = arks.shard_of_ends(lens, n)
            assert owner.min() >= 0 and owner.max() < n
            assert (owner[0::2] == owner[1::2]).all()
            load = np.bincount(owner, weights=lens, minlength=n)
            assert load.max() - load.min() <= 2 * int(lens.max()), (pattern, n, load)
            assert (arks.shard_of_ends(lens, n) == owner).all()            # a function of its input
    assert arks.shard_of_ends(np.array([7, 7, 9], dtype=np.uint32), 2).tolist() == [0, 0, 1]   # odd count: last end alone
    assert len(arks.shard_of_ends(np.zeros(0, np.uint32), 4)) == 0


def _kernel_private_sizes(so):
    """{kernel symbol: private_segment_fixed_size} of every gfx950 kernel of a hipcc-built shared object (the
    code objects of the clang offload bundle in .hip_fatbin, their amdhsa metadata read with llvm-readelf)"""
    import struct
    import tempfile
    data = open(so, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}
    pos = 0
    while True:
        b = data.find(magic, pos)
        if b < 0:
            break
        n, = struct.unpack_from("<Q", data, b + 24)
        p = b + 32
        for _ in range(n):
            off, size, idl = struct.unpack_from("<QQQ", data, p)
            p += 24
            tid = data[p:p + idl].decode()
            p += idl
            if "gfx950" in tid and size:
                with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                    f.write(data[b + off:b + off + size])
                    name = f.name
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", name],
                                     stdout=subprocess.PIPE, text=True).stdout
                os.unlink(name)
                for m in re.finditer(r"\.private_segment_fixed_size:\s+(\d+)\n(?:(?!\.private_segment_fixed_size).*\n)*?"
                                     r"\s+\.symbol:\s+(\S+)", txt):
                    out[m.group(2)] = int(m.group(1))
        pos = b + 24
    return out


def test_no_kernel_uses_scratch_memory(arks):
    """No kernel of the library has a private segment: scratch memory is set up per queue by the runtime on
    demand, and the one unexplained fault of round 2 (`Memory access fault by GPU ... address (nil)` with 24
    processes on the device) is what a missing scratch base looks like; the two kernels that kept hoisted loop
    invariants in scratch no longer do."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not installed")
    sizes = _kernel_private_sizes(os.path.join(ROOT, "arcs_amd", "lib", "libarks_hip.so"))
    assert len(sizes) > 100 and any("map_reads_s_kernel" in k for k in sizes)
    assert {k: v for k, v in sizes.items() if v} == {}
