// arks_kernels.hpp -- host-visible launchers of the kernels in arks_kernels.hip.
#pragma once

#include "arks_device.hpp"

namespace arks {

hipError_t launch_pack(
    const uint8_t* ascii, const u64* offsets, const u32* lens, const u64* word_off, long n_seqs,
    u64 total_words, u64* codes, u32* nmask, u32* n_count, u32* other, hipStream_t st);
hipError_t launch_read_class(
    const u32* lens, const u32* n_count, const u32* other, long n, uint8_t* out, hipStream_t st);
hipError_t launch_visit(
    const u32* nmask, const u64* word_off, const u32* lens, long n_ends, int k, u32* visited,
    u64* counters, hipStream_t st);
hipError_t launch_popcount(const u32* words, u64 n, u64* out, hipStream_t st);
hipError_t launch_insert(
    int kw, const u64* codes, const u32* visited, const u32* word_owner, long n_ends, u64 total_words,
    const KeyGeom& g, TableView t, u64* counters, hipStream_t st);
hipError_t launch_poison(
    int kw, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView t,
    u64* counter, hipStream_t st);
hipError_t launch_build_stats(
    int kw, const u64* codes, const u32* visited, const u64* word_off, long n_ends, u64 total_words,
    const KeyGeom& g, TableView t, u64* counters, hipStream_t st);
// bytes of the scratch block behind queue_count: 4 u32 counters, pad to 64, 64 rows of 8 u64 partial statistics
constexpr size_t kMapScratchBytes = 64 + 64 * 8 * sizeof(u64) + 64 * 128; // ... and up to 64 work counters 128 bytes apart
hipError_t launch_map_reads(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens,
    const uint8_t* eval, long n_reads, double j_index, const KeyGeom& g, TableView t,
    const BIndexView& bx, int* out, u64* stats, u32* queue, u32* queue_count, int n_cu,
    hipStream_t st, bool raw = false, // raw: out is u64[n_reads], the votes of put_result<true>
    const uint8_t* gate_class = nullptr, const uint8_t* gate_ok = nullptr); // the pair gate worked out by the seed tile kernel (bx.dense only)
void set_medium_blocks_cap(unsigned n); // arks_debug_set_medium_blocks (0: no cap)
hipError_t launch_seed_counts(const u32* lens, const uint8_t* eval, long n_reads, int k, int w, int* out, hipStream_t st);
hipError_t launch_seeds_fill(
    int mm, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval, long n_reads,
    int k, int w, u32 n_owners, const long* seed_off, u64* out_cm, int* out_owner, hipStream_t st);
hipError_t launch_seeds_probe(int mm, const BIndexView& bx, const u64* cm, long n, u64* ans, hipStream_t st);
hipError_t launch_map_reads_seeded(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval,
    long n_reads, double j_index, const KeyGeom& g, const BIndexView& bx, const BIndexView& bxg, const long* seed_off,
    const u64* ans, int* out, u64* stats, u32* queue, u32* queue_count, int n_cu, hipStream_t st,
    const u32* seed_slot = nullptr, const u32* chunk_off = nullptr, bool scratch_zeroed = false);
// arks_exchange: a batch's seeds listed and bucketed by owner in one launch (arks_shard.hip)
// The counters of seed_bucket_kernel.  They are never zeroed (round 5: a memset per batch was a launch of its own on
// a stream that has ~10 of them per batch): they only count up, the launch is told where they stood (SeedBucketBase:
// the host knows, the set's previous batch is complete when the next is submitted) and works with the differences.
// One counter per 128-byte line: thousands of blocks add to each of them, and atomics on one line queue up behind each
// other whatever their addresses.
constexpr int kCtlStride = 16; // u64 words between two counters
struct SeedBucketCtl
{
	u64 fill[64 * kCtlStride]; // [o * kCtlStride]: seeds asked of owner o so far (goes on counting when a region is full)
	u64 seeds[kCtlStride];     // [0]: seeds so far, sent or not (the read-major numbering)
	u64 overflow[kCtlStride];  // [0]: number (SeedBucketBase::seq) of the last launch in which a block found no room:
	                           // that batch is run again with larger regions
};
struct SeedBucketBase
{
	u64 fill[64];
	u64 seeds;
	u64 seq; // > 0, another one for every launch on this ctl
};
// the pair gate of chromiumRead (checkReadSequence of both mates and the barcode test, Arcs.cpp:1264-1292) folded into
// the bucket kernel: eval is computed from the read classes and pair_ok (arks_pair_gate_device's rule) and written to
// eval_out for the map kernels, instead of being read
struct SeedBucketGate
{
	const uint8_t* read_class = nullptr; // != NULL: compute; reads 2p, 2p+1 are mates
	const uint8_t* pair_ok = nullptr;    // may be NULL
	uint8_t* eval_out = nullptr;
};
long seed_bucket_chunks(long n_reads);
// out[0] = status, out[1 + o] = the batch's seeds for owner o (0 with a status): what a rank contributes to the counts
// all-gather of the RCCL transport, made on the device behind the bucket launch
hipError_t launch_gather_prep(const SeedBucketCtl* ctl, const SeedBucketBase& base, u64 status, int n_owners, u64* out, hipStream_t st);
// dst[i] = src[i] ^ salt ^ i (arks_exchange_create_local's check of the path between two devices)
hipError_t launch_peer_pattern(const u64* src, u64* dst, int n, u64 salt, hipStream_t st);
// zero_words / n_zero: a block of words the launch zeroes on the side (the map kernels' scratch of the same stream:
// saves that launch's memset); may be NULL
hipError_t launch_seed_bucket(
    int mm, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval, long n_reads,
    int k, int w, u32 n_owners, u64 cap, u64 slot_cap, SeedBucketCtl* ctl, const SeedBucketBase& base,
    const SeedBucketGate& gate, u32* zero_words, int n_zero, u32* chunk_off, u32* slot, u64* send, hipStream_t st);
// owner side of arks_exchange: the seeds of several askers answered in one launch; segment s holds n[s] seeds at
// src[s], their answers (16 B each) go to dst[s]
struct ProbeSegs
{
	int n_segs;
	const u64* src[65];
	u64* dst[65];
	u64 end[65]; // inclusive prefix of the segments' lengths
};
hipError_t launch_seeds_probe_segs(int mm, const BIndexView& bx, const ProbeSegs& sg, hipStream_t st, int n_cu = 256);
hipError_t launch_word_owner(const u64* word_off, long n_ends, u64 total_words, u32* owner, hipStream_t st);
hipError_t launch_bmark(
    int kw, int mm, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView full,
    int w, bool dense, u32* ambig, u32* is_min, u32* is_pal, u32* is_img, hipStream_t st);
hipError_t launch_bcount(
    int mm, const u64* codes, const u32* is_min, u64 total_words, u64* ckeys, u32* ccnts, u64 ccap, u32 own, u32 n_own,
    hipStream_t st);
hipError_t launch_bforce(
    int kw, int mm, int phase, const u64* codes, const u32* is_pal, u64 total_words, const KeyGeom& g, int w,
    bool dense, u64* ckeys, u32* ccnts, u64 ccap, u64* mtab, u64 mcap, u32 own, u32 n_own, hipStream_t st);
hipError_t launch_bfill_mtab(
    int mm, const u64* codes, const u32* is_min, u64 total_words, u64* ckeys, u32* ccnts, u64 ccap, u64* mtab,
    u64 mcap, u32* heavy_min, u32 own, u32 n_own, bool fill, u32 heavy_over, hipStream_t st);
hipError_t launch_bfallback(
    int kw, int mm, bool insert, const u64* codes, const u32* visited, const u32* ambig, const u32* is_pal,
    const u32* is_img, const u32* heavy_min, const u32* word_owner, u64 total_words, const KeyGeom& g,
    int w, bool dense, TableView fb, u64* counter, hipStream_t st);
hipError_t launch_bdilate(const u32* visited, u64 total_words, int w, u32* is_min, hipStream_t st);
hipError_t launch_bowners(
    int mm, const u64* codes, const u32* is_min, u64 total_words, u32 n_own, u64* per_owner, hipStream_t st);
hipError_t launch_btextrec(
    const u64* codes, const u32* visited, const u32* ambig, const u32* word_owner, u64 alloc_words, u64* trec,
    u32* owner_blk,
    hipStream_t st);
hipError_t launch_bexport(
    int kw, const u64* codes, const u32* visited, const u32* ambig, const u32* word_owner,
    u64 total_words, const KeyGeom& g, u64* out_keys, int* out_vals, u64* counter, hipStream_t st);
hipError_t launch_max_votes(u64* acc, const u64* in, long n, hipStream_t st);
hipError_t launch_resolve_votes(
    const u64* votes, const u32* lens, long n_reads, int k, double j_index, int* out, hipStream_t st);
hipError_t launch_pair_gate(
    const uint8_t* pair_ok, const uint8_t* read_class, long n_pairs, uint8_t* eval, hipStream_t st);
// the IndexMap accumulator (arks_imap.hip): open-addressed (barcode id << 32 | conreci) -> count, plus the
// sequence number of the first stored pair of the entry (the order in which a single-threaded reference
// run would have created it)
struct ImapView
{
	u64* keys;      // 0 = empty
	u32* counts;
	u64* first;     // ~0 until set
	u64 cap;        // slots
	u32* n_entries; // occupied slots
	u32* overflow;  // set when an insert found no slot (the growth rule of arks_pairs_device excludes it)
};
hipError_t launch_pairs(
    const int* conreci, const uint8_t* pair_ok, const u32* barcode_id, long n_pairs, int* out_pair,
    const ImapView& im, u64 seq_base, u64* stored, hipStream_t st);
hipError_t launch_imap_rehash(const ImapView& from, const ImapView& to, hipStream_t st);
hipError_t launch_imap_compact(
    const ImapView& im, u64* out_keys, u64* out_first, u32* out_counts, u32* cursor, hipStream_t st);

#ifdef ARKS_PROFILE_SECTIONS
void read_section_cycles(unsigned long long* out16);
#endif
#ifdef ARKS_MEDIUM_DIAG
void read_medium_diag(unsigned long long* out16);
#endif

} // namespace arks
