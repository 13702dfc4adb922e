"""ctypes bindings of the CPU oracle (oracle/libarks_oracle.so) and, when present, of the
reference-encoder shim (oracle/_ref/libarks_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (arcs_amd/).  See oracle/arks_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libarks_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libarks_ref.so")


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ("total_kmers", "null_kmers", "recorded", "collisions", "removed_dup", "unique")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class MapStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail",
                 "windows")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _sources_digest():
    import hashlib
    h = hashlib.sha256()
    for f in ("arks_oracle.c", "arks_port_fastq.c", "arks_oracle.h", "Makefile"):
        h.update(open(os.path.join(_HERE, f), "rb").read())
    return h.hexdigest()


def build_oracle(force=False):
    """(re)build oracle/libarks_oracle.so with gcc; also the _ref shim when the upstream checkout
    exists (this container only).  Stale = by content (a digest of the sources beside the library), not by time
    stamp: a copied tree must not look stale to twelve processes at once (tests/test_zz_gpu_stress.py starts that
    many, each calls this).  The library is built beside its final place and renamed into it: a process that has
    the old one mapped keeps its (unlinked) file, nobody ever maps a half-written one."""
    stamp = _LIB + ".digest"
    want = _sources_digest()
    try:
        fresh = os.path.exists(_LIB) and open(stamp).read().strip() == want
    except OSError:
        fresh = os.path.exists(_LIB) and os.path.getmtime(_LIB) >= max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("arks_oracle.c", "arks_port_fastq.c", "arks_oracle.h"))
        if fresh:
            try:
                open(stamp, "w").write(want + "\n")
            except OSError:
                pass
    if force or not fresh:
        tmp = "libarks_oracle.%d.tmp.so" % os.getpid()
        subprocess.check_call(["make", "-C", _HERE, "-B", tmp], stdout=subprocess.DEVNULL)
        os.replace(os.path.join(_HERE, tmp), _LIB)
        with open(stamp + ".%d" % os.getpid(), "w") as f:
            f.write(want + "\n")
        os.replace(stamp + ".%d" % os.getpid(), stamp)
    if os.path.isdir("/root/reference/Common"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build_oracle()
        L = C.CDLL(_LIB)
        L.arks_oracle_key_bytes.argtypes = [C.c_int]
        L.arks_oracle_key.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
        L.arks_oracle_index_new.restype = C.c_void_p
        L.arks_oracle_index_new.argtypes = [C.c_int]
        L.arks_oracle_index_free.argtypes = [C.c_void_p]
        L.arks_oracle_index_size.restype = C.c_size_t
        L.arks_oracle_index_size.argtypes = [C.c_void_p]
        L.arks_oracle_index_get.argtypes = [C.c_void_p, C.c_void_p]
        L.arks_oracle_index_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.arks_oracle_map_kmers.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int,
                                            C.POINTER(BuildStats)]
        L.arks_oracle_map_kmers_range.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.arks_oracle_end_cutoff.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.arks_oracle_best_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_double,
                                              C.POINTER(MapStats)]
        L.arks_oracle_check_read_sequence.argtypes = [C.c_char_p, C.c_int]
        L.arks_oracle_map_pairs.restype = C.c_int64
        L.arks_oracle_map_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_double, C.c_void_p,
                                            C.c_void_p, C.POINTER(MapStats), C.c_int]
        L.arks_oracle_map_fastq_gz.restype = C.c_int64
        L.arks_oracle_map_fastq_gz.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_int, C.POINTER(MapStats),
                                               C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def key_bytes(k):
    return lib().arks_oracle_key_bytes(k)


def key(seq, pos, k):
    """packed canonical key (bytes) of seq[pos:pos+k], or None for the NULL k-mer"""
    if isinstance(seq, str):
        seq = seq.encode()
    out = C.create_string_buffer(key_bytes(k))
    ok = lib().arks_oracle_key(seq, pos, k, out)
    return out.raw if ok else None


def check_read_sequence(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    return bool(lib().arks_oracle_check_read_sequence(seq, len(seq)))


def end_cutoff(length, min_size=500, end_length=30000):
    c = C.c_int(0)
    ok = lib().arks_oracle_end_cutoff(length, min_size, end_length, C.byref(c))
    return c.value if ok else None


def contig_ends(contigs, min_size=500, end_length=30000):
    """getContigKmers' enumeration (Arcs/Arcs.cpp:1047-1093): list of end strings, end i <->
    conreci i+1 (head = 2n-1, tail = 2n of the n-th valid contig)."""
    ends = []
    for s in contigs:
        c = end_cutoff(len(s), min_size, end_length)
        if c is None:
            continue
        ends.append(s[:c])
        ends.append(s[len(s) - c:])
    return ends


class OracleIndex:
    def __init__(self, k):
        self.k = k
        self.kb = key_bytes(k)
        self.h = lib().arks_oracle_index_new(k)
        if not self.h:
            raise ValueError("bad k")
        self.stats = BuildStats()

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # module globals are gone at interpreter exit
            try:
                lib().arks_oracle_index_free(self.h)
            except Exception:
                pass
            self.h = None

    def map_kmers(self, seq, conreci):
        if isinstance(seq, str):
            seq = seq.encode()
        return lib().arks_oracle_map_kmers(self.h, seq, len(seq), conreci, C.byref(self.stats))

    def map_kmers_range(self, seq, conreci, lo, hi):
        """the scan of map_kmers over the whole end, inserting only the windows that start in [lo, hi)"""
        if isinstance(seq, str):
            seq = seq.encode()
        return lib().arks_oracle_map_kmers_range(self.h, seq, len(seq), conreci, int(lo), int(hi))

    def build(self, ends):
        for i, e in enumerate(ends):
            self.map_kmers(e, i + 1)
        return self

    def __len__(self):
        return lib().arks_oracle_index_size(self.h)

    def get(self, keybytes):
        return lib().arks_oracle_index_get(self.h, keybytes)

    def dump(self):
        n = len(self)
        keys = np.zeros((n, self.kb), dtype=np.uint8)
        vals = np.zeros(n, dtype=np.int32)
        lib().arks_oracle_index_dump(self.h, keys.ctypes.data, vals.ctypes.data)
        return keys, vals

    def best_contig(self, read, j_index, stats=None):
        if isinstance(read, str):
            read = read.encode()
        return lib().arks_oracle_best_contig(self.h, read, len(read), j_index,
                                             C.byref(stats) if stats is not None else None)

    def map_pairs(self, bases, offsets, lens, j_index, pair_ok=None, threads=1):
        """bases: bytes / uint8 array of concatenated ASCII reads; offsets uint64[n_reads],
        lens uint32[n_reads]; reads 2p, 2p+1 are mates.  Returns (conreci[n_reads],
        pair[n_pairs], stats dict)."""
        bases = np.frombuffer(bases, dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) \
            else np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n_pairs = len(lens) // 2
        out_c = np.zeros(2 * n_pairs, dtype=np.int32)
        out_p = np.zeros(n_pairs, dtype=np.int32)
        st = MapStats()
        ok_ptr = None
        if pair_ok is not None:
            pair_ok = np.ascontiguousarray(pair_ok, dtype=np.uint8)
            ok_ptr = pair_ok.ctypes.data
        stored = lib().arks_oracle_map_pairs(self.h, bases.ctypes.data, offsets.ctypes.data,
                                             lens.ctypes.data, n_pairs, ok_ptr, j_index,
                                             out_c.ctypes.data, out_p.ctypes.data, C.byref(st),
                                             threads)
        d = st.as_dict()
        d["stored_pairs"] = int(stored)
        return out_c, out_p, d


def map_fastq_gz(index, path, j_index, threads=1):
    """the CPU port end to end from a gzipped interleaved FASTQ (oracle/arks_port_fastq.c): (pairs read, stored
    pairs, counters dict)"""
    st = MapStats()
    stored = C.c_int64(0)
    n = lib().arks_oracle_map_fastq_gz(index.h, os.fsencode(path), float(j_index), int(threads), C.byref(st), C.byref(stored))
    if n < 0:
        raise OSError(f"cannot open {path}")
    return int(n), int(stored.value), st.as_dict()


def sub_draft_index(k, contigs, members, min_size=500, end_length=30000, site_runs=None):
    """OracleIndex over the ends of the contigs whose index is in `members` (contigs: uint8 arrays or
    bytes, FASTA order), numbered as getContigKmers numbers them in the WHOLE draft (head of the n-th
    valid contig = 2n-1, tail = 2n): what a human-scale index is checked against when the whole map
    (1.4 G keys) is out of a test's reach -- see arcs_amd.synth.closed_contig_set for which contigs a
    set of reads needs.

    site_runs (int64[n, 2] intervals of the concatenated draft: arcs_amd.synth.alternating_at_runs, and
    synth.sites_to_runs of a draft's planted repeat copies): the stretches whose k-mers (short flank +
    (AT)n and the reverse-complement palindromes inside; windows of repeat copies) recur between sites
    all over the draft.  For every contig that is NOT a member, the windows of its ends that overlap such a
    stretch are inserted too -- under the visit rule of the whole end (map_kmers_range) -- so that these
    keys carry the whole draft's value (owner or 0) and reads that reach into a microsatellite need not
    be left out of the comparison."""
    members = set(members)
    ox = OracleIndex(k)
    lens = np.fromiter((len(c) for c in contigs), dtype=np.int64, count=len(contigs))
    cstart = np.zeros(len(contigs) + 1, dtype=np.int64)
    np.cumsum(lens, out=cstart[1:])
    per_contig = {}
    if site_runs is not None:
        for s, e in np.asarray(site_runs, dtype=np.int64):
            ci = int(np.searchsorted(cstart, s, side="right") - 1)
            while s < e and ci < len(contigs):               # a stretch that runs over a contig border is cut there
                stop = min(e, cstart[ci + 1])
                if stop > s:
                    per_contig.setdefault(ci, []).append((int(s - cstart[ci]), int(stop - cstart[ci])))
                s = stop
                ci += 1
    n = 0
    for ci, c in enumerate(contigs):
        cut = end_cutoff(len(c), min_size, end_length)
        if cut is None:
            continue
        n += 1
        if ci in members:
            b = c.tobytes() if hasattr(c, "tobytes") else bytes(c)
            ox.map_kmers(b[:cut], 2 * n - 1)
            ox.map_kmers(b[len(b) - cut:], 2 * n)
        elif ci in per_contig:
            b = c.tobytes() if hasattr(c, "tobytes") else bytes(c)
            L = len(b)
            for (a, z) in per_contig[ci]:                    # contig coordinates [a, z)
                # head = b[:cut]: windows that start in [a - k + 1, z), as far as they lie in the end
                if a - k + 1 < cut:
                    ox.map_kmers_range(b[:cut], 2 * n - 1, max(a - k + 1, 0), min(z, cut))
                t0 = L - cut                                 # tail = b[t0:]
                if z > t0:
                    ox.map_kmers_range(b[t0:], 2 * n, max(a - k + 1 - t0, 0), max(z - t0, 0))
    return ox


# ------------------------------------------------------------------------------------------------
# reference-encoder shim (this container only)
# ------------------------------------------------------------------------------------------------
_ref = None


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ref_proc_new.restype = C.c_void_p
        R.ref_proc_new.argtypes = [C.c_int]
        R.ref_proc_free.argtypes = [C.c_void_p]
        R.ref_proc_key.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
        R.ref_proc_keys_all.restype = C.c_int64
        R.ref_proc_keys_all.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]
        R.ref_index_new.restype = C.c_void_p
        R.ref_index_new.argtypes = [C.c_int]
        R.ref_index_free.argtypes = [C.c_void_p]
        R.ref_index_map_kmers.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        R.ref_index_stats.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_index_size.restype = C.c_int64
        R.ref_index_size.argtypes = [C.c_void_p]
        R.ref_index_get.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        R.ref_index_best_contig.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_double,
                                            C.c_void_p]
        _ref = R
    return _ref


def ref_keys_all(seq, k):
    """(keys uint8[n,kb], valid uint8[n]) of every window of seq from the REFERENCE encoder"""
    if isinstance(seq, str):
        seq = seq.encode()
    kb = key_bytes(k)
    n = max(0, len(seq) - k + 1)
    keys = np.zeros((max(n, 1), kb), dtype=np.uint8)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    h = ref().ref_proc_new(k)
    got = ref().ref_proc_keys_all(h, seq, len(seq), k, kb, keys.ctypes.data, valid.ctypes.data)
    ref().ref_proc_free(h)
    assert got == n
    return keys[:n], valid[:n]


def oracle_keys_all(seq, k):
    if isinstance(seq, str):
        seq = seq.encode()
    kb = key_bytes(k)
    n = max(0, len(seq) - k + 1)
    keys = np.zeros((max(n, 1), kb), dtype=np.uint8)
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    L = lib()
    base = keys.ctypes.data
    for i in range(n):
        valid[i] = L.arks_oracle_key(seq, i, k, base + i * kb)
    return keys[:n], valid[:n]


class RefIndex:
    """mapKmers / bestContig control flow over the reference's own ReadsProcessor"""

    def __init__(self, k):
        self.k = k
        self.kb = key_bytes(k)
        self.h = ref().ref_index_new(k)

    def __del__(self):
        if getattr(self, "h", None):
            ref().ref_index_free(self.h)
            self.h = None

    def map_kmers(self, seq, conreci):
        if isinstance(seq, str):
            seq = seq.encode()
        return ref().ref_index_map_kmers(self.h, seq, len(seq), conreci)

    def build(self, ends):
        for i, e in enumerate(ends):
            self.map_kmers(e, i + 1)
        return self

    def stats(self):
        a = np.zeros(6, dtype=np.uint32)
        ref().ref_index_stats(self.h, a.ctypes.data)
        return dict(zip(("total_kmers", "null_kmers", "recorded", "collisions", "removed_dup",
                         "unique"), (int(x) for x in a)))

    def __len__(self):
        return ref().ref_index_size(self.h)

    def get(self, keybytes):
        return ref().ref_index_get(self.h, keybytes, self.kb)

    def best_contig(self, read, j_index, counters=None):
        if isinstance(read, str):
            read = read.encode()
        return ref().ref_index_best_contig(self.h, read, len(read), j_index,
                                           counters.ctypes.data if counters is not None else None)
