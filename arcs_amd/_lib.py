"""ctypes loader of libarks_hip.so (the C ABI declared in include/arks_hip.h).

Fails loudly: a missing library is an ImportError-like RuntimeError, never a silent fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARKS_HIP_LIB: another build of the same library (A/B runs of kernel variants: profiles/tools/ab.py)
_PATH = os.environ.get("ARKS_HIP_LIB") or os.path.join(_HERE, "lib", "libarks_hip.so")


class ArksError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        L = lib()
        msg = L.arks_strerror(status).decode()
        extra = L.arks_last_error_string().decode()
        super().__init__(f"{what}: {msg}" + (f" ({extra})" if extra else ""))


class BuildStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("total_kmers", "null_kmers", "recorded", "collisions", "removed_dup", "unique",
                 "short_ends")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class MapStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("total_valid", "bad", "found", "recorded", "dups", "reads_pass", "reads_fail",
                 "windows")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


MAP_STATS_FIELDS = [n for n, _ in MapStats._fields_]


class BuildOptions(C.Structure):
    """arks_build_options (include/arks_hip.h): the layout choices of an index build; results never depend on them"""
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in
                ("index_kind", "heavy_over", "minimizer_len", "fallback_load_inv", "shard", "n_shards",
                 "seed_rank", "seed_ranks")]


INDEX_KINDS = {"auto": 0, "hash": 1, "minimizer": 2, "seeds": 3}        # ARKS_INDEX_*

# name -> (restype, argtypes): every symbol include/arks_hip.h declares
_VP, _I, _I64, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
SYMBOLS = {
    "arks_abi_version": (_I, []),
    "arks_strerror": (C.c_char_p, [_I]),
    "arks_last_error_string": (C.c_char_p, []),
    "arks_device_count": (_I, []),
    "arks_key_bytes": (_I, [_I]),
    "arks_index_build": (_I, [C.POINTER(_VP), _I, _VP, _VP, _VP, _I64, _I, C.POINTER(BuildStats)]),
    "arks_index_build_ex": (_I, [C.POINTER(_VP), _I, _VP, _VP, _VP, _I64, _I, C.POINTER(BuildOptions), C.POINTER(BuildStats)]),
    "arks_shard_of_ends": (_I, [_VP, _I64, _I, _VP]),
    "arks_index_build_shard": (_I, [C.POINTER(_VP), _I, _VP, _VP, _VP, _I64, _I, _I, _I]),
    "arks_index_build_shard_stats": (_I, [C.POINTER(_VP), _I, _VP, _VP, _VP, _I64, _I, _I, _I, C.POINTER(BuildStats)]),
    "arks_index_build_seed_shard": (_I, [C.POINTER(_VP), _I, _VP, _VP, _VP, _I64, _I, _I, _I, C.POINTER(BuildStats)]),
    "arks_index_seed_ranks": (_I, [_VP]),
    "arks_seed_counts_device": (_I, [_VP, _VP, _VP, _I64, _VP, _VP]),
    "arks_seeds_fill_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP]),
    "arks_seeds_probe_device": (_I, [_VP, _VP, _I64, _VP, _VP]),
    "arks_map_reads_seeded_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP, _D, _VP, _VP, _VP]),
    "arks_exchange_unique_id": (_I, [_VP]),
    "arks_exchange_create": (_I, [C.POINTER(_VP), _VP, _VP, _I, _I]),
    "arks_exchange_create_local": (_I, [_VP, _VP, _I]),
    "arks_exchange_free": (_I, [_VP]),
    "arks_exchange_abort": (_I, [_VP]),
    "arks_exchange_last_stats": (_I, [_VP, _VP]),
    "arks_exchange_submit": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP]),
    "arks_exchange_submit_pairs": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP, _VP]),
    "arks_exchange_complete": (_I, [_VP]),
    "arks_exchange_complete_group": (_I, [_VP, _I]),
    "arks_map_reads_exchanged_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP]),
    "arks_index_free": (_I, [_VP]),
    "arks_index_k": (_I, [_VP]),
    "arks_index_size": (_I64, [_VP]),
    "arks_index_device_bytes": (_I64, [_VP]),
    "arks_index_fallback_size": (_I, [_VP, _VP]),
    "arks_index_kind": (_I, [_VP]),
    "arks_index_export": (_I, [_VP, _VP, _VP]),
    "arks_end_cutoff": (_I, [_I, _I, _I, C.POINTER(_I)]),
    "arks_word_offsets": (_I, [_VP, _I64, _VP]),
    "arks_pack_reads_device": (_I, [_VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP, _I, _VP]),
    "arks_pack_reads_host": (_I, [_VP, _VP, _VP, _VP, _I64, _VP, _VP, _VP]),
    "arks_map_reads_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP]),
    "arks_map_pairs_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I64, _D, _VP, _VP, _VP]),
    "arks_map_votes_device": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP]),
    "arks_votes_max_device": (_I, [_VP, _VP, _I64, _I, _VP]),
    "arks_votes_resolve_device": (_I, [_VP, _VP, _I64, _I, _D, _VP, _I, _VP]),
    "arks_votes_count_device": (_I, [_VP, _VP, _VP, _I64, _I, _D, _VP, _I, _VP]),
    "arks_map_reads": (_I, [_VP, _VP, _VP, _VP, _I64, _D, _VP, C.POINTER(MapStats)]),
    "arks_imap_create": (_I, [C.POINTER(_VP), _I64, _I]),
    "arks_imap_free": (_I, [_VP]),
    "arks_imap_size": (_I64, [_VP]),
    "arks_imap_export": (_I, [_VP, _VP]),
    "arks_imap_set_pair_base": (_I, [_VP, C.c_uint64]),
    "arks_imap_export_ordered": (_I, [_VP, _VP, _VP]),
    "arks_debug_queue_counts": (_I, [_VP, _VP]),
    "arks_exchange_debug_set_rccl": (_I, [_VP]),
    "arks_debug_set_medium_blocks": (_I, [_I]),
    "arks_pair_gate_device": (_I, [_VP, _VP, _I64, _VP, _I, _VP]),
    "arks_gate_count_device": (_I, [_VP, _VP, _I64, _VP, _I, _VP]),
    "arks_pairs_device": (_I, [_VP, _VP, _VP, _I64, _VP, _VP, _VP, _I, _VP]),
}

_lib = None


ABI_VERSION = 4          # include/arks_hip.h ARKS_ABI_VERSION


def lib_path():
    return _PATH


def lib():
    """the loaded library; raises RuntimeError when it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(
                f"{_PATH} is missing: build it with `python -m arcs_amd.build` "
                "(hipcc --offload-arch=gfx950); arcs_amd has no CPU fallback")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 /
        # libhsa-runtime64.so.1, and a second copy (ROCm's, which libarks_hip.so is linked to)
        # cannot open the device once the first has.  Importing torch first makes the dynamic
        # loader satisfy libarks_hip's DT_NEEDED sonames with the copies torch already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError when the ABI and the header disagree
            fn.restype = res
            fn.argtypes = args
        v = L.arks_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"{_PATH} reports ABI version {v}, this package is written against {ABI_VERSION}"
                               + (" (a calibration build: results wrong by design)" if v < 0 else "")
                               + ": rebuild with `python -m arcs_amd.build --force`")
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise ArksError(status, what)
