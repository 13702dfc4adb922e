/*
 * arks_hip.h -- C ABI of libarks_hip.so: the MI355X (gfx950) implementation of ARCS's ARKS-mode
 * read -> contig-end k-mer mapping path.
 *
 * The reference (bcgsc/arcs v1.2.8) has no plugin/FFI seam for this path; the seams are the plain
 * C++ functions of Arcs/Arcs.cpp that runArcs() calls (Arcs.cpp:1897,1902).  Each entry point
 * below names the reference function it replaces.  All arguments are plain pointers and sizes;
 * no C++ or torch types cross this boundary; no exception and no abort() crosses it either --
 * every function returns an ARKS_* status code (the reference prints a message and exit(1)s; the
 * host CLI maps codes back to those messages).
 *
 * Conventions
 *   - "end e" (0-based) of an index build <-> contig-end index `conreci` = e + 1
 *     (Arcs.cpp:1057-1058: head of the n-th valid contig = 2n-1, tail = 2n; 0 = "null contig").
 *   - h_ pointers are host memory, d_ pointers are device (HIP) memory on the index's device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Device entry points
 *     are asynchronous on that stream; host entry points return after their results are valid.
 *   - Handles are thread-compatible, not thread-safe (one in-flight call per handle/stream), like
 *     the per-thread ReadsProcessor of Arcs.cpp:1152-1156.
 *
 * Packed read layout (what the kernels consume; produced by arks_pack_* below)
 *   - codes: uint64 words, 32 bases per word, MSB first (base 0 of a word in bits 63:62), A=0 C=1
 *     G=2 T=3 (case-insensitive), anything else 0.  Read r starts at word d_word_off[r] and owns
 *     ceil(len/32) words; unused tail bits are 0.  The array must be followed by >= 4 readable
 *     padding words.  These are exactly the bytes of the reference's packed k-mer
 *     (Common/ReadsProcessor.cpp:376-535) so a window's key is a bit-field of the stream.
 *   - nmask: uint32 words, 1 bit per base, same indexing (bit 31 = base 0 of the word); 1 = the
 *     character was not one of ACGTacgt (such a window is the NULL k-mer, ReadsProcessor.cpp:400).
 */
#ifndef ARKS_HIP_H
#define ARKS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): arks_index_build_seed_shard took a trailing `arks_build_stats*` and arks_exchange_stats grew `reruns`
 * in round 4 without a new number; callers check arks_abi_version() == ARKS_ABI_VERSION at load (arcs_amd/_lib.py,
 * arcs_amd/host/arcs.cpp) so that a caller built against another header fails there, not in a wild write.  A
 * calibration build of the library (-DARKS_CALIBRATION_BUILD: kernels with a memory phase taken out, results wrong by
 * design, profiles/tools/) reports the NEGATIVE number and is refused by both. */
#define ARKS_ABI_VERSION 4 /* 3 (end of round 5): arks_map_pairs_device added; 4 (round 6): arks_index_build_ex +
                           * arks_build_options -- the library no longer reads the caller's environment */

/* status codes */
#define ARKS_OK 0
#define ARKS_ERR_BAD_K 1        /* k <= 3 (ReadsProcessor.cpp:25 assert) or k in {6,10} (reference UB) */
#define ARKS_ERR_K_UNSUPPORTED 2 /* k > ARKS_MAX_K */
#define ARKS_ERR_OOM 3          /* host or device allocation failed */
#define ARKS_ERR_HIP 4          /* a HIP runtime call / kernel launch failed (arks_last_error_string) */
#define ARKS_ERR_NO_DEVICE 5    /* no usable gfx950 device: the product has no CPU fallback */
#define ARKS_ERR_BAD_ARG 6      /* NULL pointer, negative count, ... */
/* 7 was never assigned */
#define ARKS_ERR_FULL 8         /* an accumulator insert found no slot (the growth rule excludes it: a bug) */

#define ARKS_MAX_K 96

int arks_abi_version(void);
const char* arks_strerror(int status);
/* text of the last HIP error seen by the calling thread (empty string when none) */
const char* arks_last_error_string(void);
/* number of visible HIP devices whose architecture is gfx950; 0 when there is none */
int arks_device_count(void);
/* bytes of one packed key, Common/ReadsProcessor.cpp:20-37 (k=60 -> 15) */
int arks_key_bytes(int k);

/* ---- counters -------------------------------------------------------------------------------- */

/* index-build counters printed by getContigKmers under -v, Arcs/Arcs.cpp:1107-1128
 * (64-bit here; the CLI narrows to the reference's 32-bit unsigned for printing). */
typedef struct
{
	uint64_t total_kmers; /* "Total number of Kmers"      sum of mapKmers() returns :1087,1093 */
	uint64_t null_kmers;  /* "Number Null Kmers"          s_numbadkmers        :924 */
	uint64_t recorded;    /* "Number Kmers Recorded"      s_numkmersmapped     :919 */
	uint64_t collisions;  /* "Number Kmer Collisions"     s_numkmercollisions  :915 */
	uint64_t removed_dup; /* "Number Times Kmers Removed" s_numkmersremdup     :909 */
	uint64_t unique;      /* "Number of unique kmers"     s_uniquedraftkmers   :911,918 */
	uint64_t short_ends;  /* ends shorter than k (the warning of :877-882), no k-mers added */
} arks_build_stats;

/* read-mapping counters printed by chromiumRead under -v, Arcs/Arcs.cpp:1329-1340 */
typedef struct
{
	uint64_t total_valid; /* s_totalnumckmers :966 */
	uint64_t bad;         /* s_numbadckmers   :991 */
	uint64_t found;       /* s_numckmersfound :987 */
	uint64_t recorded;    /* s_numckmersrec   :977 */
	uint64_t dups;        /* s_ckmersasdups   :980 */
	uint64_t reads_pass;  /* s_numreadspassingjaccard :1007 */
	uint64_t reads_fail;  /* s_numreadsfailjaccard    :1011 */
	uint64_t windows;     /* sum of totalnumkmers :962 -- the unit of the throughput metric */
} arks_map_stats;

/* ---- the contig k-mer index ------------------------------------------------------------------ */

/* Opaque device-resident replacement of `ARCS::ContigKMap kmap` (Arcs/Arcs.h:158) together with
 * the ReadsProcessor(k) geometry (Arcs.cpp:1044-1045). */
typedef struct arks_index arks_index;

/* Replaces getContigKmers' loop over mapKmers (Arcs/Arcs.cpp:869-929, 1084-1091): k-merizes every
 * end string (ASCII, h_bases + h_offsets[e], h_lens[e] characters) with the reference's visit rule
 * (a NULL k-mer jumps k positions, :922-925) and records key -> (e+1), or 0 once a key has been
 * seen from two different ends.  The caller does the head/tail split (arks_end_cutoff).
 * stats may be NULL (skips the extra pass that the "removed" counter needs). */
int arks_index_build(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int device,
    arks_build_stats* stats);

/* Every build above and below is a case of this one.  The options choose the LAYOUT of the index and of its build --
 * never a result: whatever they say, every map call returns what Arcs/Arcs.cpp:939-1014 returns.  opt == NULL or a
 * zeroed struct with struct_size set = the defaults (what arks_index_build does).  Rounds 1-5 read these choices from
 * the process environment on every build (ARKS_INDEX_KIND, ARKS_HEAVY_OVER, ARKS_MINIMIZER_LEN, ARKS_FALLBACK_LOAD):
 * a shared library whose layout follows its host's environment is no drop-in, and getenv() under a host that calls
 * setenv() from another thread is a data race; the release build of the library has no getenv in it (tests/test_abi.py). */
#define ARKS_INDEX_AUTO 0      /* locality index where k >= 20, seed table where it fits the device, else minimizers */
#define ARKS_INDEX_HASH 1      /* the exact hash table only (what k < 20 always gets) */
#define ARKS_INDEX_MINIMIZER 2 /* locality index, minimizer table (~20x smaller than the seed table) */
#define ARKS_INDEX_SEEDS 3     /* locality index, seed table (every m-mer position of a visited window) */
typedef struct
{
	uint32_t struct_size;      /* sizeof(arks_build_options) of the caller: the struct may grow at its end */
	int32_t index_kind;        /* ARKS_INDEX_*; a kind that k or the text size rules out falls back as AUTO does */
	int32_t heavy_over;        /* occurrences beyond which an m-mer is "heavy" (its windows go to the exact table):
	                              0 = the default (2), else 2..8 */
	int32_t minimizer_len;     /* 0 = by k (the long one, 21, from k = 24 on, else the short one); >= 19: the long one
	                              where k leaves room, 1..18: the short one */
	int32_t fallback_load_inv; /* slots per key of the exact table: 0 = 4 where the device has room, else 2; 2; 4 */
	int32_t shard, n_shards;   /* contig shards (arks_index_build_shard_stats): 0, 0|1 = every end */
	int32_t seed_rank, seed_ranks; /* seed-table shards (arks_index_build_seed_shard): 0, 0|1 = the whole table */
} arks_build_options;
int arks_index_build_ex(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int device,
    const arks_build_options* opt,
    arks_build_stats* stats);

/* One shard of the index for a draft whose whole index should not (or cannot) live on one GPU --
 * BASELINE config "contig k-mer index hash-sharded across 8 GPUs".  Shard `shard` of `n_shards` holds the
 * k-mers of the ends that arks_shard_of_ends assigns to it (a contig's head and tail together; contigs
 * dealt in list order to the shard with the fewest bases so far); every rank passes the SAME end list,
 * conreci numbering is that of the whole list.
 * A key that also occurs in an end of another shard reads 0, as it does in the one map of
 * Arcs/Arcs.cpp:903-920: the foreign ends are streamed through the shard's table once while it is
 * built (no exchange between ranks).  Every conreci therefore lives in exactly one shard, and the
 * winner of bestContig's walk (Arcs.cpp:998-1004) over the whole map is the maximum over shards of the
 * per-shard winners -- see arks_map_votes_device.  Such a shared key is KEPT by one shard only (round 4): its
 * first holder, the shard of the smallest end of the list that visited it -- the end the serial loop of
 * :884-927 meets first (the streaming carries the end numbers: an atomic minimum per shared key); the other
 * shards take their visits of it back.  No vote changes (the key reads 0 wherever it is), and every key of the
 * one map is in exactly one shard: the counters of the build and of the read stage add up over the shards.
 * n_shards == 1 is arks_index_build. */
int arks_shard_of_ends(const uint32_t* h_lens, int64_t n_ends, int n_shards, int32_t* h_shard);
int arks_index_build_shard(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int shard,
    int n_shards,
    int device);
/* The same with the counters of getContigKmers (Arcs/Arcs.cpp:1093-1107) -- this shard's SHARE of them: the sums
 * over the shards are exactly what arks_index_build reports for the whole list.  total_kmers, null_kmers and
 * short_ends are those of the shard's own ends; a key that several shards hold is `recorded` by the shard that
 * holds the smallest end of the list that visited it (the one the serial loop of :884-927 meets first), and every
 * other visit is a collision in the shard of its end; removed_dup counts, as in the one map, the visits after the
 * first end's; unique counts the keys no other end of the list holds.  stats == NULL is arks_index_build_shard. */
int arks_index_build_shard_stats(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int shard,
    int n_shards,
    int device,
    arks_build_stats* stats);

/* The north star's sharded configuration (BASELINE configs[3]): the SEED TABLE of the seed index -- 32 B per
 * text position, 43 GB for a 3 Gbp draft, everything else of the index is ~2 GB -- split over the ranks
 * of a node by a hash prefix of the canonical m-mer.  Every rank passes the same end list and gets: the
 * packed text, its bitmaps and the exact fallback table (replicated: identical on every rank), the seeds
 * it OWNS, and a replicated minimizer table for the general kernels (the few reads the hot kernel does not
 * finish).  A read is mapped on ONE rank (its home, reads are dealt to ranks); per batch the home
 *   1. lists its reads' seeds        arks_seed_counts_device, a prefix sum, arks_seeds_fill_device
 *   2. routes each seed (8 B) to its owner -- the all-to-all of the north star (RCCL over xGMI)
 *   3. the owner answers             arks_seeds_probe_device (16 B per seed)
 *   4. the answers travel back       the reverse all-to-all
 *   5. the home finishes             arks_map_reads_seeded_device (== arks_map_reads_device's results)
 * ~60 B per read on the wire instead of the 2.6 KB of keys that routing every k-mer would cost; the table,
 * the probes and the reads all divide by the number of ranks.  n_ranks == 1 is arks_index_build. */
int arks_index_build_seed_shard(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int rank,
    int n_ranks,
    int device,
    arks_build_stats* stats /* may be NULL; the counters of the whole draft (the same on every rank) */);
/* number of ranks the index's seed table is sharded over (1 = whole) */
int arks_index_seed_ranks(const arks_index* idx);

int arks_index_free(arks_index* idx);
int arks_index_k(const arks_index* idx);
/* number of distinct keys (== kmap.size()) */
int64_t arks_index_size(const arks_index* idx);
/* device bytes held by the index */
int64_t arks_index_device_bytes(const arks_index* idx);
/* locality indexes: keys the text cannot answer and the exact table holds instead -- palindromes, windows under a heavy
 * seed (an m-mer of a repeat family), quirk images -- and that table's bytes (part of arks_index_device_bytes); what a
 * repeat-rich draft costs shows here.  out[0] = keys, out[1] = bytes; zeros for the plain hash table (kind 0). */
int arks_index_fallback_size(const arks_index* idx, int64_t out[2]);
/* layout of the index: 0 = exact open-addressed hash table of packed keys (k < 20, or when the
 * environment says ARKS_INDEX_KIND=hash); 1 = locality index over MINIMIZERS (packed contig-end text +
 * table of the minimizer positions + exact fallback table; ARKS_INDEX_KIND=minimizer); 2 = locality
 * index over SEEDS (the same text and fallback, but the table holds every m-mer position of the text --
 * 32 B per text position -- so that the read side needs no minimizers, only one fixed-position seed per
 * k - m + 1 windows; the default whenever the table fits in half of the free device memory;
 * ARKS_INDEX_KIND=seeds).  DESIGN.md.  Results are identical, only speed and memory differ. */
int arks_index_kind(const arks_index* idx);
/* Copies every (key, value) to host: h_keys = size * arks_key_bytes(k) bytes in the reference's
 * byte order (what ReadsProcessor::getStr returns), h_vals = size int32.  Order unspecified. */
int arks_index_export(const arks_index* idx, unsigned char* h_keys, int32_t* h_vals);

/* The head/tail split of getContigKmers, Arcs/Arcs.cpp:1056,1072-1074.  Returns 1 and the length
 * of both end substrings in *cutoff, or 0 when the contig is skipped (len < min_size). */
int arks_end_cutoff(int len, int min_size, int end_length, int* cutoff);

/* ---- packing --------------------------------------------------------------------------------- */

/* Words of packed storage for n sequences: h_word_off[i] = first word of sequence i,
 * h_word_off[n] = total words (add ARKS_PAD_WORDS when allocating). */
#define ARKS_PAD_WORDS 4
int arks_word_offsets(const uint32_t* h_lens, int64_t n, uint64_t* h_word_off);

/* ASCII -> packed, on the device.  d_read_class[r] (may be NULL) receives what
 * checkReadSequence (Arcs/Arcs.cpp:366-389) decides for read r in bit 0: 1 = accepted (only ACGTN,
 * N fraction <= 0.02), 0 = rejected; bit 1 is set for an accepted read that holds ACGT only (values
 * 0, 1, 3): arks_pair_gate_device hands it on, and the map kernels do not fetch the N masks of such
 * reads. */
int arks_pack_reads_device(
    const uint8_t* d_ascii,
    const uint64_t* d_offsets,
    const uint32_t* d_lens,
    const uint64_t* d_word_off,
    int64_t n_reads,
    uint64_t* d_codes,
    uint32_t* d_nmask,
    uint8_t* d_read_class,
    int device,
    void* stream);

/* The same packing on the host (for ingest threads that ship 3 bits/base over PCIe instead of 8). */
int arks_pack_reads_host(
    const char* h_ascii,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    const uint64_t* h_word_off,
    int64_t n_reads,
    uint64_t* h_codes,
    uint32_t* h_nmask,
    uint8_t* h_read_class);

/* ---- read mapping ---------------------------------------------------------------------------- */

/* Replaces bestContig (Arcs/Arcs.cpp:939-1014) for a batch of packed reads resident on the device:
 * d_out_conreci[r] = the contig end whose k-mers dominate read r (count/total > j_index, ties to
 * the smallest index, total counts NULL windows too) or 0.  d_eval (may be NULL = all) selects the
 * reads bestContig is called for (Arcs.cpp:1268); the others get 0 and touch no counter.
 * d_stats (may be NULL) points to one arks_map_stats in device memory that is ADDED to.
 * Asynchronous on `stream`.  An index keeps one set of work queues per stream it is mapped on: calls on
 * different streams may run at the same time, calls on one stream are ordered by the stream. */
int arks_map_reads_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);

/* arks_map_reads_device for a batch of PAIRS (reads 2p, 2p + 1 are mates; n_reads even) with the gate of chromiumRead
 * (Arcs/Arcs.cpp:1264-1268) folded in: what arks_pair_gate_device would write from d_read_class (arks_pack_reads_device
 * / the host packer) and d_pair_ok (may be NULL = every pair) is worked out by the map kernel itself while it reads
 * the batch's metadata -- one launch and one pass over three arrays less per batch (round 5: 1.6 % of a step of 500 M
 * pairs).  Results are arks_pair_gate_device + arks_map_reads_device's, read for read and counter for counter.
 * d_eval_out (n_reads bytes, required) is scratch the call MAY write: an index whose layout is not the seed index
 * (arks_index_kind != 2) takes the two launches through it; with the seed index it is left untouched -- a caller that
 * wants the gate's array (arks_gate_count_device) calls arks_pair_gate_device. */
int arks_map_pairs_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_pair_ok,
    const uint8_t* d_read_class,
    uint8_t* d_eval_out,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);

/* bestContig (Arcs/Arcs.cpp:939-1004) up to, not including, the j_index test, against ONE shard:
 * d_out_votes[r] = (count << 32) | ~conreci of the end that won the walk of :998-1004 in this shard
 * (0 = nothing recorded, or d_eval[r] == 0).  The unsigned 64-bit MAXIMUM of a read's votes over all
 * shards is the vote of the whole map: the larger count wins and a tie keeps the smaller conreci
 * (:1000, strict <).  That maximum is the only data-path exchange of the sharded configuration (one
 * 8-byte all-reduce(MAX) per read; reads are replicated to every shard), arks_votes_resolve_device
 * finishes the call.  Same rule per index and stream as arks_map_reads_device.
 * No arks_map_stats here; the counters of the whole map (Arcs.cpp:1329-1340) come from arks_map_reads_device on
 * every shard -- a shard's index keeps a shared key only if the shard is the key's first holder
 * (arks_index_build_shard), so found, recorded and dups ADD UP over the shards, and total_valid, bad and windows are
 * the same in every shard (take one) -- and from arks_votes_count_device on the folded votes (reads_pass,
 * reads_fail).  `arcs --index-shards -v` does that: a second map pass per shard, for the verbose log only. */
int arks_map_votes_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    uint64_t* d_out_votes,
    void* stream);

/* ---- read mapping over a sharded seed table (arks_index_build_seed_shard) ------------------------------- */

/* d_counts[r] = seeds of read r: ceil(windows / (k - m + 1)), 0 when d_eval[r] == 0 */
int arks_seed_counts_device(
    const arks_index* idx, const uint32_t* d_lens, const uint8_t* d_eval, int64_t n_reads, int32_t* d_counts, void* stream);
/* d_seed_off = exclusive prefix sum of the counts (n_reads + 1 entries).  Seed s of the batch (read-major):
 * d_seed_mmer[s] = its canonical m-mer right-aligned (~0: it holds an invalid base, nobody has it),
 * d_seed_owner[s] = the rank whose shard holds it. */
int arks_seeds_fill_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    const int64_t* d_seed_off,
    uint64_t* d_seed_mmer,
    int32_t* d_seed_owner,
    void* stream);
/* Owner side: d_answers[2 i], [2 i + 1] = the (up to two) entries of d_mmer[i] in this rank's shard; 0 = none,
 * [2 i] == 1: more than two, [2 i] == 2: a heavy seed (the home consults the fallback table). */
int arks_seeds_probe_device(const arks_index* idx, const uint64_t* d_mmer, int64_t n, uint64_t* d_answers, void* stream);
/* arks_map_reads_device with the seed probes answered beforehand: d_answers[2 s] for the seeds in the
 * order of arks_seeds_fill_device.  Same results, same counters. */
int arks_map_reads_seeded_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    const int64_t* d_seed_off,
    const uint64_t* d_answers,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);

/* ---- the sharded seed table (the product path of BASELINE configs[3]) ------------------------------------- */

/* One arks_exchange per rank: its shard of the seed table, two sets of device buffers for the seeds that travel
 * (two batches in flight), and the transport to the other ranks:
 *   - RCCL (arks_exchange_create; one process per GPU): ncclAllGather of the per-owner counts, ncclSend / ncclRecv
 *     groups over xGMI for the seeds and for the answers; librccl is opened on demand.  The host program that starts
 *     the ranks (bench.py: torch.distributed) carries the 128-byte id of rank 0 to the others, as an MPI or NCCL
 *     program does;
 *   - direct (arks_exchange_create_local; the ranks are threads of ONE process): the devices of the ranks are the
 *     same device or peers of each other (xGMI); nothing is copied -- the owner's probe kernel reads the askers'
 *     buffers where they lie and writes its answers into theirs; host barriers and events order it.  What
 *     `arcs --index-sharded` runs, and what a single-GPU box can test. */
typedef struct arks_exchange arks_exchange;
#define ARKS_EXCHANGE_ID_BYTES 128
typedef struct
{
	uint64_t seeds;    /* seeds of the last completed batch of this rank */
	uint64_t sent;     /* ... of which asked of other ranks (8 B out, 16 B back each) */
	uint64_t received; /* seeds other ranks asked of this one */
	uint64_t reruns;   /* batches so far that did not fit the send buffer's regions and were bucketed again, larger */
	uint64_t stream_syncs; /* times the exchange has drained a stream from the host so far: a buffer of a set had to grow
	                        * (its old block may still be read), or a set was handed a batch on another stream than its
	                        * last.  A run of batches of one shape on fixed streams adds none: the host then waits for
	                        * the batch's counts (an event) and for nothing else */
} arks_exchange_stats;

/* rank 0: a fresh id for arks_exchange_create on every rank (ncclGetUniqueId) */
int arks_exchange_unique_id(unsigned char* out_id /* ARKS_EXCHANGE_ID_BYTES */);
/* shard = arks_index_build_seed_shard(..., rank, world, device); unique_id may be NULL when world == 1.  With a
 * communicator, a send to oneself of a known pattern checks the data type numbering before anything travels. */
int arks_exchange_create(arks_exchange** out, const arks_index* shard, const unsigned char* unique_id, int rank, int world);
/* out[r] for r in [0, world): the ranks of one process (shards[r] = shard r, on one device or on devices with peer
 * access to each other, which is enabled here); each rank is driven by its own host thread, and
 * arks_exchange_complete blocks until all of them have called it */
int arks_exchange_create_local(arks_exchange** out, const arks_index* const* shards, int world);
int arks_exchange_free(arks_exchange* x);
/* a local rank whose driver gives up (an error outside the library): the other ranks of its group get an error from
 * arks_exchange_complete instead of waiting for it (a rank that is missing for ten minutes has the same effect);
 * no-op for an RCCL exchange (a rank that fails inside the library aborts its communicator itself) */
int arks_exchange_abort(arks_exchange* x);
int arks_exchange_last_stats(const arks_exchange* x, arks_exchange_stats* out);

/* bestContig (Arcs/Arcs.cpp:939-1014) of this rank's reads against the sharded seed table, in two steps -- the
 * same results and counters as arks_map_reads_device against the whole index (the reads are independent,
 * Arcs.cpp:1169; the index is only read, :969-971):
 *   arks_exchange_submit    On `stream`: the batch's seeds listed and bucketed by owner (one kernel), the
 *                           per-owner counts copied to the host.  Does not wait for anybody; with the RCCL transport
 *                           it also ENQUEUES the all-gather of the counts behind the kernel (round 5; complete() did
 *                           it until then, and waited for the stream), so every rank submits its batches in the same
 *                           order -- which the collective complete() asks for anyway.  At most two batches may be
 *                           submitted and not yet completed; the arrays must stay valid until the batch's
 *                           results have been consumed.
 *   arks_exchange_complete  COLLECTIVE: every rank calls it, in the same order, for its oldest submitted batch
 *                           (which may be empty).  Counts to everybody, the seeds to their owners (8 B each),
 *                           owner-side probe, answers back (16 B each), map kernels -- all on the batch's stream;
 *                           d_out_conreci / d_stats are complete when that stream gets there.
 * Submitting batch n + 1 (on a second stream) before completing batch n keeps the device busy while the host
 * waits for counts: they have long arrived.  A rank that fails (in either call) makes the same complete() fail on
 * every rank; nobody is left waiting.  A buffer set is reused in the order of the stream it was last used on. */
int arks_exchange_submit(
    arks_exchange* x,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);
/* arks_exchange_submit for a batch of PAIRS (reads 2p, 2p + 1 are mates; n_reads even) with the pair gate of
 * chromiumRead folded in: what arks_pair_gate_device would write from d_read_class (arks_pack_reads_device / the host
 * packer) and d_pair_ok (may be NULL) is computed by the bucketing kernel itself -- one launch and one pass over an
 * array less per batch -- and written to d_eval_out (n_reads bytes), which the map kernels of arks_exchange_complete
 * read and the caller may use afterwards (arks_gate_count_device, ...). */
int arks_exchange_submit_pairs(
    arks_exchange* x,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_read_class,
    const uint8_t* d_pair_ok,
    int64_t n_reads,
    double j_index,
    uint8_t* d_eval_out,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);
int arks_exchange_complete(arks_exchange* x);
/* arks_exchange_complete for ALL ranks of a local group (arks_exchange_create_local) from ONE thread: xs[r] = rank r,
 * each with a submitted batch; the stages of the ranks are interleaved in this call instead of meeting at barriers.
 * What a host program with one consumer thread (arcs --index-sharded) calls; a group is driven either this way or
 * by a thread per rank, not both at once. */
int arks_exchange_complete_group(arks_exchange* const* xs, int world);

/* submit + complete for one batch at a time (COLLECTIVE; nothing else may be in flight) */
int arks_map_reads_exchanged_device(
    arks_exchange* x,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream);

/* d_acc[r] = max(d_acc[r], d_in[r]) (unsigned): folds another shard's votes in, for a driver that
 * moves votes between GPUs itself (hipMemcpyPeerAsync) instead of calling RCCL's all-reduce(MAX). */
int arks_votes_max_device(uint64_t* d_acc, const uint64_t* d_in, int64_t n_reads, int device, void* stream);

/* The tail of bestContig (Arcs/Arcs.cpp:996,1006-1013) over reduced votes: d_out_conreci[r] = conreci
 * if count / (len - k + 1 windows, NULL ones included, :962) > j_index in double, else 0. */
int arks_votes_resolve_device(
    const uint64_t* d_votes,
    const uint32_t* d_lens,
    int64_t n_reads,
    int k,
    double j_index,
    int32_t* d_out_conreci,
    int device,
    void* stream);

/* The same test as counters (Arcs.cpp:1006-1013), for the reads bestContig is called for (d_eval[r] != 0; NULL = all):
 * d_stats->reads_pass += reads with count / total > j_index, d_stats->reads_fail += the others. */
int arks_votes_count_device(
    const uint64_t* d_votes,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    int k,
    double j_index,
    arks_map_stats* d_stats,
    int device,
    void* stream);

/* Host convenience over the above (pack + map + copy back); mirrors calling bestContig on every
 * read of the batch.  stats may be NULL. */
int arks_map_reads(
    const arks_index* idx,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_reads,
    double j_index,
    int32_t* h_out_conreci,
    arks_map_stats* stats);

/* ---- pairs and the IndexMap ------------------------------------------------------------------ */

/* Device accumulator for `IndexMap imap` (Arcs/Arcs.h:108-113) as (barcode id, conreci) -> count;
 * the caller keeps the barcode string <-> id dictionary.  `capacity_entries` is a starting size only:
 * the table grows (arks_pairs_device rebuilds it larger before a launch could fill it beyond one half;
 * that is the one place where the call waits for the device), so no input can overflow it.
 * Calls that use one accumulator must come from one host thread at a time; they may use different
 * streams (the device side is atomic, a rebuild waits for the whole device first). */
typedef struct arks_imap arks_imap;
int arks_imap_create(arks_imap** out, int64_t capacity_entries, int device);
int arks_imap_free(arks_imap* m);
/* number of distinct (barcode id, conreci) entries, negative status on error; waits for the device */
int64_t arks_imap_size(const arks_imap* m);
/* h_triples = size * 3 uint32 (barcode id, conreci, count), sorted by (barcode id, conreci). */
int arks_imap_export(const arks_imap* m, uint32_t* h_triples);
/* The same plus, per triple, the sequence number of the FIRST stored pair that created the entry:
 * pair p of a batch has number base + p, where base is what arks_imap_set_pair_base set before the
 * batch's arks_pairs_device call (default: batches continue each other's numbering).  With numbers that
 * follow the input order, sorting the barcodes by their smallest number gives the order in which
 * chromiumRead's `imap[barcode][...]++` (Arcs.cpp:1282-1285) creates them in a single-threaded run --
 * which decides the iteration order of the reference's unordered_map and so the vertex order of
 * `.dist.gv` and the tie order of -D. */
int arks_imap_set_pair_base(arks_imap* m, uint64_t first_pair);
int arks_imap_export_ordered(const arks_imap* m, uint32_t* h_triples, uint64_t* h_first_pair);

/* The gate of chromiumRead, Arcs/Arcs.cpp:1264-1268: d_eval[2p], d_eval[2p+1] nonzero iff
 * pair_ok[p] && (class[2p] & 1) && (class[2p+1] & 1)  (goodmult is always true, :1267); a nonzero
 * d_eval[r] is 1 (ARKS_EVAL) or 3 (ARKS_EVAL_ACGT_ONLY: class bit 1, "read r holds ACGT only").
 * The d_eval contract of every mapping entry point: 0 = not evaluated; ARKS_EVAL_ACGT_ONLY (exactly 3) =
 * evaluate, and the caller VOUCHES that the read's N mask is all zero (the kernels do not fetch it); ANY
 * other nonzero value (1, 2, 0xFF ...) = evaluate, nothing known about the bases.  A caller's own array
 * must therefore not hold 3 unless it means it. */
#define ARKS_EVAL 1
#define ARKS_EVAL_ACGT_ONLY 3
int arks_pair_gate_device(
    const uint8_t* d_pair_ok,
    const uint8_t* d_read_class,
    int64_t n_pairs,
    uint8_t* d_eval,
    int device,
    void* stream);

/* *d_counter += the gated pairs one of whose mates checkReadSequence rejected (skipped_invalidreadpair,
 * Arcs/Arcs.cpp:1273-1276): d_pair_ok[p] (NULL = every pair) && !d_eval[2p], d_eval from arks_pair_gate_device.
 * For a front end that packs and classifies the reads on the device (arks_pack_reads_device) and so has no
 * read classes on the host to count with. */
int arks_gate_count_device(
    const uint8_t* d_pair_ok,
    const uint8_t* d_eval,
    int64_t n_pairs,
    uint64_t* d_counter,
    int device,
    void* stream);

/* The pair rule of chromiumRead, Arcs/Arcs.cpp:1280-1292: reads 2p, 2p+1 are mates;
 * d_out_pair[p] = c1 if (c1 != 0 && c1 == c2) else 0; for stored pairs with d_pair_ok[p] != 0
 * (NULL = all) imap[(d_barcode_id[p], c1)]++ (imap and d_barcode_id may both be NULL).
 * d_stored (may be NULL) points to one uint64 on the device that is ADDED to. */
int arks_pairs_device(
    const int32_t* d_conreci,
    const uint8_t* d_pair_ok,
    const uint32_t* d_barcode_id,
    int64_t n_pairs,
    int32_t* d_out_pair,
    arks_imap* imap,
    uint64_t* d_stored,
    int device,
    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARKS_HIP_H */
