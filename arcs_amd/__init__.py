"""arcs_amd -- MI355X-native ARKS read->contig k-mer mapping (the hot path of bcgsc/arcs --arks).

The product is libarks_hip.so (hand-written HIP for gfx950 behind the C ABI of include/arks_hip.h).
This package only loads it and offers thin host-side helpers for tests, bench.py and the
multi-GPU driver; there is no CPU fallback -- on a machine without the built library or without a
gfx950 device every compute call raises.
"""
from . import api  # noqa: F401
from ._lib import ArksError, lib, lib_path  # noqa: F401
from .api import (ArksIndex, ImapAccumulator, PackedReads, PairStep, SeedExchange, contig_ends, device_count,  # noqa: F401
                  count_votes, end_cutoff, key_bytes, map_pairs_fused, map_pairs_packed, map_reads_packed, map_votes_packed, max_votes,
                  pack_reads_host, pair_gate, pairs_rule, queue_counts, resolve_votes, shard_of_ends)

__all__ = ["ArksError", "ArksIndex", "ImapAccumulator", "PackedReads", "SeedExchange", "contig_ends",
           "device_count", "end_cutoff", "key_bytes", "lib", "lib_path", "map_pairs_fused", "map_pairs_packed",
           "count_votes", "map_reads_packed", "map_votes_packed", "max_votes", "pack_reads_host", "pair_gate", "pairs_rule",
           "queue_counts", "resolve_votes", "shard_of_ends"]
