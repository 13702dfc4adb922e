"""Host-side helpers over the C ABI (include/arks_hip.h) for tests, bench.py and the multi-GPU
driver.  torch is used only as the owner of device memory and streams; every computation happens
in libarks_hip.so.  Names follow the reference (contig ends, conreci, reads, pairs, barcodes)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import INDEX_KINDS, ArksError, BuildOptions, BuildStats, MapStats, check, lib

# Layout choices (fields of arks_build_options) that every ArksIndex.build* of this process applies unless the call
# says otherwise.  Tests and A/B runs set them here (tests/conftest.py index_layout): rounds 1-5 went through the
# process environment, which the library read at every build; it reads no environment any more.
BUILD_DEFAULTS = {}


def set_medium_blocks(n):
    """arks_debug_set_medium_blocks (include/arks_hip_debug.h): the medium kernel on at most n waves (0: no cap)"""
    check(lib().arks_debug_set_medium_blocks(int(n)), "arks_debug_set_medium_blocks")


def device_count():
    """number of visible gfx950 devices (0 on a machine without an MI355X)"""
    return lib().arks_device_count()


def key_bytes(k):
    return lib().arks_key_bytes(k)


def end_cutoff(length, min_size=500, end_length=30000):
    """getContigKmers' head/tail split (Arcs/Arcs.cpp:1056,1072-1074); None = contig skipped"""
    c = C.c_int(0)
    return c.value if lib().arks_end_cutoff(length, min_size, end_length, C.byref(c)) else None


def contig_ends(contigs, min_size=500, end_length=30000):
    """end strings in conreci order: end i <-> conreci i + 1 (Arcs/Arcs.cpp:1057-1091)"""
    ends = []
    for s in contigs:
        c = end_cutoff(len(s), min_size, end_length)
        if c is None:
            continue
        ends.append(s[:c])
        ends.append(s[len(s) - c:])
    return ends


def _concat(seqs):
    """list of str/bytes -> (uint8 array, uint64 offsets, uint32 lens)"""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    lens = np.fromiter((len(b) for b in bs), dtype=np.uint32, count=len(bs))
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, np.uint8)
    return np.ascontiguousarray(data), offsets, lens


def word_offsets(lens):
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    out = np.zeros(len(lens) + 1, dtype=np.uint64)
    check(lib().arks_word_offsets(lens.ctypes.data, len(lens), out.ctypes.data), "arks_word_offsets")
    return out


def pack_reads_host(seqs):
    """arks_pack_reads_host over a list of sequences -> dict of numpy arrays"""
    data, offsets, lens = _concat(seqs)
    woff = word_offsets(lens)
    total = int(woff[-1])
    codes = np.zeros(total + 4, dtype=np.uint64)
    nmask = np.zeros(total + 4, dtype=np.uint32)
    cls = np.zeros(max(len(lens), 1), dtype=np.uint8)
    if len(lens):
        pad = np.concatenate([data, np.zeros(1, np.uint8)])
        check(lib().arks_pack_reads_host(pad.ctypes.data, offsets.ctypes.data, lens.ctypes.data,
                                         woff.ctypes.data, len(lens), codes.ctypes.data,
                                         nmask.ctypes.data, cls.ctypes.data),
              "arks_pack_reads_host")
    return {"codes": codes, "nmask": nmask, "word_off": woff, "lens": lens,
            "read_class": cls[:len(lens)]}


def shard_of_ends(lens, n_shards):
    """arks_shard_of_ends: int32[n_ends], the shard that holds each contig end"""
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    out = np.zeros(max(len(lens), 1), dtype=np.int32)
    check(lib().arks_shard_of_ends(lens.ctypes.data, len(lens), int(n_shards), out.ctypes.data),
          "arks_shard_of_ends")
    return out[:len(lens)]


class ArksIndex:
    """Device-resident contig-end k-mer index: ContigKMap (Arcs/Arcs.h:158) built the way
    getContigKmers/mapKmers do (Arcs/Arcs.cpp:869-929, 1021-1129)."""

    def __init__(self, handle, k, device, stats):
        self._h = handle
        self.k = k
        self.device = device
        self.build_stats = stats

    @classmethod
    def _build_ex(cls, ends, k, device, want_stats, **choice):
        """arks_index_build_ex: every build goes through it; `choice` = fields of arks_build_options laid over
        BUILD_DEFAULTS (layout choices only: results never depend on them)"""
        data, offsets, lens = _concat(ends)
        data = np.concatenate([data, np.zeros(1, np.uint8)])
        opt = BuildOptions()
        opt.struct_size = C.sizeof(BuildOptions)
        merged = dict(BUILD_DEFAULTS)
        merged.update({f: v for f, v in choice.items() if v is not None})
        for f, v in merged.items():
            if f == "index_kind" and isinstance(v, str):
                v = INDEX_KINDS[v]
            setattr(opt, f, int(v))
        h = C.c_void_p()
        st = BuildStats()
        rc = lib().arks_index_build_ex(C.byref(h), k, data.ctypes.data, offsets.ctypes.data, lens.ctypes.data,
                                       len(lens), device, C.byref(opt), C.byref(st) if want_stats else None)
        check(rc, "arks_index_build_ex")
        return cls(h, k, device, st.as_dict() if want_stats else None)

    @classmethod
    def build(cls, ends, k, device=0, want_stats=True, index_kind=None, heavy_over=None, minimizer_len=None,
              fallback_load_inv=None):
        """arks_index_build; index_kind "auto" | "hash" | "minimizer" | "seeds" and the other layout choices of
        arks_build_options default to BUILD_DEFAULTS (empty: the library's own defaults)"""
        return cls._build_ex(ends, k, device, want_stats, index_kind=index_kind, heavy_over=heavy_over,
                             minimizer_len=minimizer_len, fallback_load_inv=fallback_load_inv)

    @classmethod
    def build_shard(cls, ends, k, shard, n_shards, device=0, want_stats=False, **choice):
        """arks_index_build_shard(_stats): the k-mers of the ends that shard_of_ends gives to `shard`, keys
        shared with any other end of the list read 0; every shard is given the same list.  want_stats: the
        shard's share of the build counters (their sums over the shards are arks_index_build's)"""
        return cls._build_ex(ends, k, device, want_stats, shard=shard, n_shards=n_shards, **choice)

    @classmethod
    def build_seed_shard(cls, ends, k, rank, n_ranks, device=0, want_stats=False, **choice):
        """arks_index_build_seed_shard: text, bitmaps and fallback table whole, the seed table's entries that
        rank `rank` of `n_ranks` owns (by a hash prefix of the m-mer), a replicated minimizer table for the
        general kernels; every rank is given the same list"""
        if n_ranks < 1 or not 0 <= rank < n_ranks:
            raise ArksError(6, "arks_index_build_seed_shard")
        return cls._build_ex(ends, k, device, want_stats, seed_rank=rank, seed_ranks=n_ranks, **choice)

    @property
    def seed_ranks(self):
        return lib().arks_index_seed_ranks(self._h)

    def close(self):
        if self._h:
            lib().arks_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return lib().arks_index_size(self._h)

    @property
    def handle(self):
        return self._h

    @property
    def kind(self):
        """0 = exact hash table, 1 = locality index with a minimizer table, 2 = locality index with a seed table
        (every m-mer position of a visited window)"""
        return lib().arks_index_kind(self._h)

    @property
    def device_bytes(self):
        return lib().arks_index_device_bytes(self._h)

    @property
    def fallback_size(self):
        """(keys, bytes) of the exact table of the keys the text cannot answer (windows under heavy seeds, palindromes)"""
        out = (C.c_int64 * 2)()
        check(lib().arks_index_fallback_size(self._h, out), "arks_index_fallback_size")
        return int(out[0]), int(out[1])

    def export(self):
        """(keys uint8[n, key_bytes] in the reference's byte order, vals int32[n])"""
        n = len(self)
        kb = key_bytes(self.k)
        keys = np.zeros((max(n, 1), kb), dtype=np.uint8)
        vals = np.zeros(max(n, 1), dtype=np.int32)
        check(lib().arks_index_export(self._h, keys.ctypes.data, vals.ctypes.data),
              "arks_index_export")
        return keys[:n], vals[:n]

    def map_reads(self, reads, j_index, want_stats=False):
        """bestContig (Arcs/Arcs.cpp:939-1014) of every read; host in, host out"""
        data, offsets, lens = _concat(reads)
        data = np.concatenate([data, np.zeros(1, np.uint8)])
        out = np.zeros(max(len(lens), 1), dtype=np.int32)
        st = MapStats()
        check(lib().arks_map_reads(self._h, data.ctypes.data, offsets.ctypes.data, lens.ctypes.data,
                                   len(lens), float(j_index), out.ctypes.data,
                                   C.byref(st) if want_stats else None), "arks_map_reads")
        out = out[:len(lens)]
        return (out, st.as_dict()) if want_stats else out


def queue_counts(index):
    """(slow, medium): reads the hot kernel left to the general kernels in the last map call on `index`
    (arks_debug_queue_counts, include/arks_hip_debug.h: a diagnostic, not part of the boundary)"""
    out = (C.c_uint * 4)()
    check(lib().arks_debug_queue_counts(index.handle, out), "arks_debug_queue_counts")
    return int(out[0]), int(out[2])


def _torch():
    import torch
    return torch


def _stream_ptr(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class PackedReads:
    """A batch of reads resident in HBM in the packed layout of include/arks_hip.h."""

    def __init__(self, codes, nmask, word_off, lens, read_class, device):
        self.codes, self.nmask, self.word_off, self.lens = codes, nmask, word_off, lens
        self.read_class = read_class
        self.device = device

    @property
    def n_reads(self):
        return int(self.lens.numel())

    def windows(self, k):
        """sum over reads of max(0, len - k + 1): the unit of the throughput metric"""
        torch = _torch()
        return int(torch.clamp(self.lens.to(torch.int64) - (k - 1), min=0).sum().item())

    @classmethod
    def from_ascii(cls, seqs, device=0):
        """upload ASCII reads and pack them on the device (arks_pack_reads_device)"""
        data, offsets, lens = _concat(seqs)
        return cls.from_arrays(data, offsets, lens, device)

    @classmethod
    def from_arrays(cls, data, offsets, lens, device=0):
        torch = _torch()
        dev = torch.device("cuda", device)
        woff = word_offsets(lens)
        total = int(woff[-1])
        n = len(lens)
        d_ascii = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).to(dev)
        d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
        d_lens = torch.from_numpy(lens.view(np.int32)).to(dev)
        d_woff = torch.from_numpy(woff.view(np.int64)).to(dev)
        codes = torch.zeros(total + 4, dtype=torch.int64, device=dev)
        nmask = torch.zeros(total + 4, dtype=torch.int32, device=dev)
        rclass = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        check(lib().arks_pack_reads_device(d_ascii.data_ptr(), d_off.data_ptr(), d_lens.data_ptr(),
                                           d_woff.data_ptr(), n, codes.data_ptr(),
                                           nmask.data_ptr(), rclass.data_ptr(), device,
                                           _stream_ptr(device)), "arks_pack_reads_device")
        torch.cuda.synchronize(device)
        return cls(codes, nmask, d_woff, d_lens, rclass[:n], device)

    @classmethod
    def from_arrays_device(cls, d_ascii, d_offsets, d_lens, device=0):
        """pack ASCII reads that already live on the device (torch uint8 / int64 / int32)"""
        torch = _torch()
        dev = torch.device("cuda", device)
        n = int(d_lens.numel())
        woff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        woff[1:] = torch.cumsum((d_lens.to(torch.int64) + 31) // 32, 0)
        total = int(woff[-1].item())
        pad = torch.zeros(64, dtype=torch.uint8, device=dev)
        d_ascii = torch.cat([d_ascii, pad])
        codes = torch.zeros(total + 4, dtype=torch.int64, device=dev)
        nmask = torch.zeros(total + 4, dtype=torch.int32, device=dev)
        rclass = torch.zeros(max(n, 1), dtype=torch.uint8, device=dev)
        check(lib().arks_pack_reads_device(d_ascii.data_ptr(), d_offsets.data_ptr(), d_lens.data_ptr(),
                                           woff.data_ptr(), n, codes.data_ptr(), nmask.data_ptr(),
                                           rclass.data_ptr(), device, _stream_ptr(device)),
              "arks_pack_reads_device")
        torch.cuda.synchronize(device)
        return cls(codes, nmask, woff, d_lens.contiguous(), rclass[:n], device)

    @classmethod
    def concat(cls, parts):
        """one batch out of several resident ones (reads start on word boundaries, so the packed words are
        simply put one behind the other and the word offsets shifted)"""
        torch = _torch()
        if len(parts) == 1:
            return parts[0]
        dev = parts[0].codes.device
        totals = [int(p.word_off[-1].item()) for p in parts]
        codes = torch.cat([p.codes[:t] for p, t in zip(parts, totals)] +
                          [torch.zeros(4, dtype=torch.int64, device=dev)])
        nmask = torch.cat([p.nmask[:t] for p, t in zip(parts, totals)] +
                          [torch.zeros(4, dtype=torch.int32, device=dev)])
        woff, base = [], 0
        for i, (p, t) in enumerate(zip(parts, totals)):
            w = p.word_off if i == len(parts) - 1 else p.word_off[:-1]
            woff.append(w + base)
            base += t
        return cls(codes, nmask, torch.cat(woff), torch.cat([p.lens for p in parts]),
                   torch.cat([p.read_class for p in parts]), parts[0].device)

    @classmethod
    def from_host_packed(cls, packed, device=0):
        """upload the output of pack_reads_host"""
        torch = _torch()
        dev = torch.device("cuda", device)
        return cls(torch.from_numpy(packed["codes"].view(np.int64)).to(dev),
                   torch.from_numpy(packed["nmask"].view(np.int32)).to(dev),
                   torch.from_numpy(packed["word_off"].view(np.int64)).to(dev),
                   torch.from_numpy(packed["lens"].view(np.int32)).to(dev),
                   torch.from_numpy(packed["read_class"]).to(dev), device)


def map_reads_packed(index, reads, j_index, eval_mask=None, stats=None, out=None):
    """arks_map_reads_device on the current torch stream.  stats: optional int64[8] device tensor
    that the counters of arks_map_stats are added to.  Returns the int32 conreci tensor."""
    torch = _torch()
    n = reads.n_reads
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
    check(lib().arks_map_reads_device(
        index.handle, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(),
        reads.lens.data_ptr(), eval_mask.data_ptr() if eval_mask is not None else None, n,
        float(j_index), out.data_ptr(), stats.data_ptr() if stats is not None else None,
        _stream_ptr(reads.device)), "arks_map_reads_device")
    return out[:n]


def map_pairs_fused(index, reads, j_index, pair_ok=None, eval_scratch=None, stats=None, out=None):
    """arks_map_pairs_device on the current torch stream: the pair gate worked out inside the map kernel (reads 2p,
    2p + 1 are mates; n_reads even).  eval_scratch: uint8[n_reads] the call may write (allocated when None).
    Returns the int32 conreci tensor."""
    torch = _torch()
    n = reads.n_reads
    assert n % 2 == 0, "arks_map_pairs_device wants whole pairs"
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
    if eval_scratch is None:
        eval_scratch = torch.empty(max(n, 1), dtype=torch.uint8, device=reads.codes.device)
    check(lib().arks_map_pairs_device(
        index.handle, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(),
        reads.lens.data_ptr(), pair_ok.data_ptr() if pair_ok is not None else None, reads.read_class.data_ptr(),
        eval_scratch.data_ptr(), n, float(j_index), out.data_ptr(), stats.data_ptr() if stats is not None else None,
        _stream_ptr(reads.device)), "arks_map_pairs_device")
    return out[:n]


def seed_counts(index, reads, eval_mask=None):
    """arks_seed_counts_device: int32[n_reads], the seeds each read asks of the seed table"""
    torch = _torch()
    n = reads.n_reads
    out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
    check(lib().arks_seed_counts_device(index.handle, reads.lens.data_ptr(),
                                        eval_mask.data_ptr() if eval_mask is not None else None, n,
                                        out.data_ptr(), _stream_ptr(reads.device)), "arks_seed_counts_device")
    return out[:n]


def seeds_fill(index, reads, seed_off, eval_mask=None):
    """arks_seeds_fill_device: (int64[n_seeds] canonical m-mers, int32[n_seeds] owner ranks), read-major;
    seed_off = int64[n_reads + 1], the exclusive prefix sum of seed_counts"""
    torch = _torch()
    n_seeds = int(seed_off[-1].item())
    dev = reads.codes.device
    mmer = torch.empty(max(n_seeds, 1), dtype=torch.int64, device=dev)
    owner = torch.empty(max(n_seeds, 1), dtype=torch.int32, device=dev)
    check(lib().arks_seeds_fill_device(
        index.handle, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(),
        reads.lens.data_ptr(), eval_mask.data_ptr() if eval_mask is not None else None, reads.n_reads,
        seed_off.data_ptr(), mmer.data_ptr(), owner.data_ptr(), _stream_ptr(reads.device)), "arks_seeds_fill_device")
    return mmer[:n_seeds], owner[:n_seeds]


def seeds_probe(index, mmer):
    """arks_seeds_probe_device (owner side): int64[2 n], the answers to the asked m-mers"""
    torch = _torch()
    n = int(mmer.numel())
    ans = torch.empty(max(2 * n, 2), dtype=torch.int64, device=mmer.device)
    check(lib().arks_seeds_probe_device(index.handle, mmer.data_ptr(), n, ans.data_ptr(),
                                        _stream_ptr(index.device)), "arks_seeds_probe_device")
    return ans[:2 * n]


def map_reads_seeded(index, reads, j_index, seed_off, answers, eval_mask=None, stats=None, out=None):
    """arks_map_reads_seeded_device: bestContig of every read with the seed probes answered beforehand"""
    torch = _torch()
    n = reads.n_reads
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
    check(lib().arks_map_reads_seeded_device(
        index.handle, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(),
        reads.lens.data_ptr(), eval_mask.data_ptr() if eval_mask is not None else None, n,
        seed_off.data_ptr(), answers.data_ptr(), float(j_index), out.data_ptr(),
        stats.data_ptr() if stats is not None else None, _stream_ptr(reads.device)), "arks_map_reads_seeded_device")
    return out[:n]


class ExchangeStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("seeds", "sent", "received", "reruns", "stream_syncs")]


class SeedExchange:
    """arks_exchange: one rank of a seed table sharded over `world` ranks (ArksIndex.build_seed_shard), with the
    transport to the others -- RCCL between processes (create), device copies between the ranks of one process
    that share a device (create_local: one host thread per rank).  map_reads is the whole step for a batch of
    this rank's reads and is collective: every rank calls it in step."""

    def __init__(self, handle, index):
        self._h = handle
        self.index = index

    @staticmethod
    def unique_id():
        """rank 0: the 128-byte id every rank passes to create (ncclGetUniqueId)"""
        buf = (C.c_ubyte * 128)()
        check(lib().arks_exchange_unique_id(buf), "arks_exchange_unique_id")
        return bytes(buf)

    @classmethod
    def create(cls, index, rank, world, unique_id=None):
        h = C.c_void_p()
        idb = (C.c_ubyte * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        check(lib().arks_exchange_create(C.byref(h), index.handle, idb, int(rank), int(world)), "arks_exchange_create")
        return cls(h, index)

    @classmethod
    def create_local(cls, shards):
        world = len(shards)
        hs = (C.c_void_p * world)()
        ix = (C.c_void_p * world)(*[sh.handle for sh in shards])
        check(lib().arks_exchange_create_local(hs, ix, world), "arks_exchange_create_local")
        return [cls(C.c_void_p(hs[r]), shards[r]) for r in range(world)]

    def close(self):
        if self._h:
            lib().arks_exchange_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def abort(self):
        """this rank's driver gives up: the other local ranks get an error instead of waiting for it"""
        if self._h:
            lib().arks_exchange_abort(self._h)

    def last_stats(self):
        st = ExchangeStats()
        check(lib().arks_exchange_last_stats(self._h, C.byref(st)), "arks_exchange_last_stats")
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def map_reads(self, reads, j_index, eval_mask=None, stats=None, out=None):
        """arks_map_reads_exchanged_device on the current torch stream -> int32 conreci tensor"""
        torch = _torch()
        n = reads.n_reads
        if out is None:
            out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
        check(lib().arks_map_reads_exchanged_device(
            self._h, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(), reads.lens.data_ptr(),
            eval_mask.data_ptr() if eval_mask is not None else None, n, float(j_index), out.data_ptr(),
            stats.data_ptr() if stats is not None else None, _stream_ptr(reads.device)),
            "arks_map_reads_exchanged_device")
        return out[:n]

    def submit(self, reads, j_index, eval_mask=None, stats=None, out=None):
        """arks_exchange_submit on the current torch stream (not collective): the batch's seeds bucketed by owner.
        Returns the conreci tensor arks_exchange_complete fills (on the same stream); the caller keeps reads,
        eval_mask and the result alive until then."""
        torch = _torch()
        n = reads.n_reads
        if out is None:
            out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
        check(lib().arks_exchange_submit(
            self._h, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(), reads.lens.data_ptr(),
            eval_mask.data_ptr() if eval_mask is not None else None, n, float(j_index), out.data_ptr(),
            stats.data_ptr() if stats is not None else None, _stream_ptr(reads.device)), "arks_exchange_submit")
        return out[:n]

    def submit_pairs(self, reads, j_index, pair_ok=None, stats=None, out=None):
        """arks_exchange_submit_pairs: submit with the pair gate folded into the bucketing kernel (reads 2p, 2p + 1 are
        mates).  Returns (conreci, eval): both filled on the stream -- eval by this call, conreci by complete()."""
        torch = _torch()
        n = reads.n_reads
        if out is None:
            out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
        ev = torch.empty(max(n, 1), dtype=torch.uint8, device=reads.codes.device)
        check(lib().arks_exchange_submit_pairs(
            self._h, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(), reads.lens.data_ptr(),
            reads.read_class.data_ptr(), pair_ok.data_ptr() if pair_ok is not None else None, n, float(j_index),
            ev.data_ptr(), out.data_ptr(), stats.data_ptr() if stats is not None else None, _stream_ptr(reads.device)),
            "arks_exchange_submit_pairs")
        return out[:n], ev[:n]

    def complete(self):
        """arks_exchange_complete (COLLECTIVE): the oldest submitted batch is exchanged and mapped on its stream"""
        check(lib().arks_exchange_complete(self._h), "arks_exchange_complete")

    @staticmethod
    def complete_group(xs):
        """arks_exchange_complete_group: every rank of a local group completed by this one thread"""
        hs = (C.c_void_p * len(xs))(*[x._h for x in xs])
        check(lib().arks_exchange_complete_group(hs, len(xs)), "arks_exchange_complete_group")

    def map_pairs(self, reads, j_index, pair_ok=None, barcode_id=None, imap=None, stored=None, stats=None):
        """chromiumRead's per-pair flow (Arcs.cpp:1264-1292) for this rank's read pairs: gate -> exchanged map ->
        pair rule + IndexMap update.  Returns (conreci, pair)."""
        ev = pair_gate(reads, pair_ok)
        conreci = self.map_reads(reads, j_index, eval_mask=ev, stats=stats)
        pair = pairs_rule(conreci, reads, pair_ok, barcode_id, imap, stored)
        return conreci, pair

    def map_pairs_pipelined(self, batches, j_index, streams, imap=None, stored=None, stats=None, n_calls=None,
                            keep=True, fold_gate=True):
        """The same flow over a list of batches (reads, pair_ok, barcode_id) with two of them in flight: batch n + 1 is
        gated and submitted on streams[(n + 1) % 2] before batch n is completed, so its bucketing runs under batch n's
        probes, transfers and map kernel, and the host never waits for counts.  COLLECTIVE: every rank makes
        n_calls (default len(batches)) complete() calls -- a rank with fewer batches completes empty ones.  The pair
        rule of all batches runs on streams[2] (the IndexMap is updated in one order).  Returns [(conreci, pair)]
        (keep=False: nothing is kept, for timing runs).  fold_gate: the pair gate computed by the bucketing kernel
        (arks_exchange_submit_pairs) instead of a launch of its own."""
        torch = _torch()
        dev = torch.device("cuda", self.index.device)
        n_calls = len(batches) if n_calls is None else n_calls
        empty = getattr(self, "_empty", None)
        if empty is None:
            empty = self._empty = PackedReads.from_arrays_device(
                torch.zeros(0, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                torch.zeros(0, dtype=torch.int32, device=dev), device=self.index.device)
        flight, res = [], []

        def submit(i):
            reads, ok, bid = batches[i] if i < len(batches) else (empty, None, None)
            st = streams[i % 2]
            with torch.cuda.stream(st):
                if fold_gate and reads.n_reads:
                    conreci, ev = self.submit_pairs(reads, j_index, pair_ok=ok, stats=stats)
                else:
                    ev = pair_gate(reads, ok)
                    conreci = self.submit(reads, j_index, eval_mask=ev, stats=stats)
            flight.append((reads, ok, bid, ev, conreci, st))

        def complete():
            reads, ok, bid, ev, conreci, st = flight.pop(0)
            self.complete()
            done = torch.cuda.Event()
            done.record(st)
            streams[2].wait_event(done)
            with torch.cuda.stream(streams[2]):
                conreci.record_stream(streams[2])
                ev.record_stream(streams[2])
                pair = pairs_rule(conreci, reads, ok, bid, imap if bid is not None else None, stored) if reads.n_reads else None
            # the batch's stream may not reuse ev / conreci before the pair rule has read them
            tail = torch.cuda.Event()
            tail.record(streams[2])
            st.wait_event(tail)
            if keep:
                res.append((conreci, pair))

        for i in range(n_calls):
            submit(i)
            if i >= 1:
                complete()
        if n_calls:
            complete()
        return res


def map_votes_packed(index, reads, eval_mask=None, out=None):
    """arks_map_votes_device against one shard of the index, on the current torch stream: int64[n]
    (bit pattern: count << 32 | ~conreci, 0 = nothing recorded); the maximum over shards is the vote
    of the whole index (counts stay below 2^31, so the signed maximum is the unsigned one)"""
    torch = _torch()
    n = reads.n_reads
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int64, device=reads.codes.device)
    check(lib().arks_map_votes_device(
        index.handle, reads.codes.data_ptr(), reads.nmask.data_ptr(), reads.word_off.data_ptr(),
        reads.lens.data_ptr(), eval_mask.data_ptr() if eval_mask is not None else None, n,
        out.data_ptr(), _stream_ptr(reads.device)), "arks_map_votes_device")
    return out[:n]


def max_votes(acc, votes, device=0):
    """arks_votes_max_device: acc = max(acc, votes) in place"""
    check(lib().arks_votes_max_device(acc.data_ptr(), votes.data_ptr(), int(acc.numel()), device,
                                      _stream_ptr(device)), "arks_votes_max_device")
    return acc


def resolve_votes(votes, reads, k, j_index, out=None):
    """arks_votes_resolve_device: the j_index test of bestContig over reduced votes -> int32 conreci"""
    torch = _torch()
    n = reads.n_reads
    if out is None:
        out = torch.empty(max(n, 1), dtype=torch.int32, device=reads.codes.device)
    check(lib().arks_votes_resolve_device(votes.data_ptr(), reads.lens.data_ptr(), n, int(k),
                                          float(j_index), out.data_ptr(), reads.device,
                                          _stream_ptr(reads.device)), "arks_votes_resolve_device")
    return out[:n]


def count_votes(votes, reads, k, j_index, stats, eval_mask=None):
    """arks_votes_count_device: reads_pass / reads_fail of folded votes added to the int64[8] device tensor `stats`"""
    check(lib().arks_votes_count_device(votes.data_ptr(), reads.lens.data_ptr(),
                                        eval_mask.data_ptr() if eval_mask is not None else None, reads.n_reads, int(k),
                                        float(j_index), stats.data_ptr(), reads.device, _stream_ptr(reads.device)),
          "arks_votes_count_device")


def pair_gate(reads, pair_ok=None):
    """arks_pair_gate_device -> uint8[2 * n_pairs]"""
    torch = _torch()
    n_pairs = reads.n_reads // 2
    ev = torch.empty(max(2 * n_pairs, 1), dtype=torch.uint8, device=reads.codes.device)
    check(lib().arks_pair_gate_device(pair_ok.data_ptr() if pair_ok is not None else None,
                                      reads.read_class.data_ptr(), n_pairs, ev.data_ptr(),
                                      reads.device, _stream_ptr(reads.device)), "arks_pair_gate_device")
    return ev


def pairs_rule(conreci, reads, pair_ok=None, barcode_id=None, imap=None, stored=None):
    """arks_pairs_device -> int32[n_pairs]"""
    torch = _torch()
    n_pairs = reads.n_reads // 2
    pair = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=reads.codes.device)
    check(lib().arks_pairs_device(conreci.data_ptr(),
                                  pair_ok.data_ptr() if pair_ok is not None else None,
                                  barcode_id.data_ptr() if barcode_id is not None else None,
                                  n_pairs, pair.data_ptr(), imap.handle if imap is not None else None,
                                  stored.data_ptr() if stored is not None else None,
                                  reads.device, _stream_ptr(reads.device)), "arks_pairs_device")
    return pair[:n_pairs]


class ImapAccumulator:
    """Device accumulator of IndexMap (Arcs/Arcs.h:108-113) as (barcode id, conreci) -> count."""

    def __init__(self, capacity_entries, device=0):
        h = C.c_void_p()
        check(lib().arks_imap_create(C.byref(h), capacity_entries, device), "arks_imap_create")
        self._h = h
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().arks_imap_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = lib().arks_imap_size(self._h)
        if n < 0:
            raise _lib.ArksError(int(-n), "arks_imap_size")
        return int(n)

    def triples(self):
        """uint32[n, 3] (barcode id, conreci, count) sorted by (barcode id, conreci)"""
        n = len(self)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        check(lib().arks_imap_export(self._h, out.ctypes.data), "arks_imap_export")
        return out[:n]

    def set_pair_base(self, first_pair):
        """sequence number of pair 0 of the next batch (arks_imap_set_pair_base)"""
        check(lib().arks_imap_set_pair_base(self._h, int(first_pair)), "arks_imap_set_pair_base")

    def triples_ordered(self):
        """(triples, uint64[n]: sequence number of the first stored pair of each entry)"""
        n = len(self)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        first = np.zeros(max(n, 1), dtype=np.uint64)
        check(lib().arks_imap_export_ordered(self._h, out.ctypes.data, first.ctypes.data),
              "arks_imap_export_ordered")
        return out[:n], first[:n]


def map_pairs_packed(index, reads, j_index, pair_ok=None, barcode_id=None, imap=None,
                     stats=None, stored=None, fused=True):
    """The per-pair flow of chromiumRead (Arcs/Arcs.cpp:1264-1292) for a resident batch whose
    reads 2p, 2p+1 are mates: gate -> bestContig of both mates -> pair rule -> imap update.
    fused (round 5): the gate inside the map kernel (arks_map_pairs_device) instead of a launch of its own
    (arks_pair_gate_device + arks_map_reads_device; what an odd number of reads takes anyway).
    Returns (conreci int32[n_reads], pair int32[n_pairs])."""
    torch = _torch()
    dev = reads.codes.device
    n_pairs = reads.n_reads // 2
    sp = _stream_ptr(reads.device)
    ev = torch.empty(max(2 * n_pairs, 1), dtype=torch.uint8, device=dev)
    if fused and reads.n_reads % 2 == 0:
        conreci = map_pairs_fused(index, reads, j_index, pair_ok=pair_ok, eval_scratch=ev, stats=stats)
    else:
        check(lib().arks_pair_gate_device(pair_ok.data_ptr() if pair_ok is not None else None,
                                          reads.read_class.data_ptr(), n_pairs, ev.data_ptr(),
                                          reads.device, sp), "arks_pair_gate_device")
        conreci = map_reads_packed(index, reads, j_index, eval_mask=ev, stats=stats)
    pair = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=dev)
    check(lib().arks_pairs_device(conreci.data_ptr(),
                                  pair_ok.data_ptr() if pair_ok is not None else None,
                                  barcode_id.data_ptr() if barcode_id is not None else None,
                                  n_pairs, pair.data_ptr(), imap.handle if imap is not None else None,
                                  stored.data_ptr() if stored is not None else None,
                                  reads.device, sp), "arks_pairs_device")
    return conreci, pair[:n_pairs]


class PairStep:
    """One pass of the hot path over a resident batch with every buffer preallocated (what
    bench.py times): pair gate -> map_reads -> pair rule + imap update; the gate inside the map kernel
    (arks_map_pairs_device) unless fused=False."""

    def __init__(self, index, reads, j_index, pair_ok=None, barcode_id=None, imap=None, fused=True):
        torch = _torch()
        dev = reads.codes.device
        self.fused = bool(fused) and reads.n_reads % 2 == 0
        self.index, self.reads, self.j = index, reads, float(j_index)
        self.pair_ok, self.barcode_id, self.imap = pair_ok, barcode_id, imap
        self.n_pairs = reads.n_reads // 2
        self.eval = torch.empty(max(2 * self.n_pairs, 1), dtype=torch.uint8, device=dev)
        self.conreci = torch.empty(max(reads.n_reads, 1), dtype=torch.int32, device=dev)
        self.pair = torch.empty(max(self.n_pairs, 1), dtype=torch.int32, device=dev)

    def run(self, stats=None, stored=None, map_events=None):
        L = lib()
        r = self.reads
        sp = _stream_ptr(r.device)
        if self.fused:      # the gate worked out by the map kernel (arks_map_pairs_device)
            if map_events is not None:
                map_events[0].record()
            map_pairs_fused(self.index, r, self.j, pair_ok=self.pair_ok, eval_scratch=self.eval, stats=stats, out=self.conreci)
        else:
            check(L.arks_pair_gate_device(self.pair_ok.data_ptr() if self.pair_ok is not None else None,
                                          r.read_class.data_ptr(), self.n_pairs, self.eval.data_ptr(),
                                          r.device, sp), "arks_pair_gate_device")
            if map_events is not None:
                map_events[0].record()
            map_reads_packed(self.index, r, self.j, eval_mask=self.eval, stats=stats, out=self.conreci)
        if map_events is not None:
            map_events[1].record()
        check(L.arks_pairs_device(self.conreci.data_ptr(),
                                  self.pair_ok.data_ptr() if self.pair_ok is not None else None,
                                  self.barcode_id.data_ptr() if self.barcode_id is not None else None,
                                  self.n_pairs, self.pair.data_ptr(),
                                  self.imap.handle if self.imap is not None else None,
                                  stored.data_ptr() if stored is not None else None, r.device, sp),
              "arks_pairs_device")
