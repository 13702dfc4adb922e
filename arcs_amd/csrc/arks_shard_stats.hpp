// arks_shard_stats.hpp -- launchers (arks_build.hip) behind the build counters of a contig-sharded index
// (arks_index_build_shard_stats): a key that several shards hold is counted by the shard whose end is the smallest
// end of the whole list that visited it -- the end the serial loop of Arcs/Arcs.cpp:884-927 meets first.
#pragma once

#include "arks_device.hpp"

namespace arks {

// launch_poison that also lowers the slot's "smallest end that visited the key" to the foreign end's conreci:
// word_end[w] = 1-based index into `conreci` of the end that owns text word w of the foreign chunk
hipError_t launch_poison_min(
    int kw, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView t,
    const u32* word_end, const u32* conreci, u64* counter, hipStream_t st);
// *out += the keys of the table whose smallest end is one of this shard's (lens[end - 1] != 0: a shard sees the
// other shards' ends as empty strings)
hipError_t launch_count_first_holder(TableView t, const u32* lens, u64* out, hipStream_t st);

} // namespace arks
