// arks_shard_stats.hpp -- launchers (arks_build.hip) of the "first holder" rule of a contig-sharded index
// (arks_index_build_shard): a key that ends of several shards visit is kept, and counted, by the shard whose end is
// the smallest end of the whole list that visited it -- the end the serial loop of Arcs/Arcs.cpp:884-927 meets first.
// With every key in one shard the counters of the build AND of the read stage add up over the shards.
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
// the shards that are not a key's first holder take their visits of it out of `visited` (*counter += visits taken back)
hipError_t launch_drop_later_holders(
    int kw, const u64* codes, u32* visited, const u32* lens, u64 total_words, const KeyGeom& g, TableView t, u64* counter,
    hipStream_t st);
// the first holders' slots of `from` re-inserted into the empty table `to` (a hash-table index: the table IS the index)
hipError_t launch_keep_first_holders(int kw, TableView from, const u32* lens, TableView to, hipStream_t st);

// reads_pass / reads_fail (stats[5], stats[6] of an arks_map_stats) of votes folded over the shards (arks_shard.hip)
hipError_t launch_votes_count(
    const u64* votes, const u32* lens, const uint8_t* eval, long n_reads, int k, double j_index, u64* stats, hipStream_t st);

} // namespace arks
