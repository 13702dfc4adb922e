/*
 * arks_oracle.h -- CPU restatement of the ARKS read->contig k-mer mapping path of bcgsc/arcs.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker /
 * the timed CPU baseline.  The product path (arcs_amd/, include/) never links or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to the upstream
 * checkout, bcgsc/arcs v1.2.8).  Parity status: PINNED -- the key function is differentially
 * tested against the reference's own Common/ReadsProcessor.cpp compiled into oracle/_ref (see
 * oracle/Makefile, tests/test_oracle_vs_ref.py), the golden keys of SURVEY.md section 8 and the
 * index-build counters of Examples/arks_test-demo/output/..._arks.log:53-58 are reproduced
 * (tests/test_oracle_golden.py).
 */
#ifndef ARKS_ORACLE_H
#define ARKS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARKS_ORACLE_MAX_KEY_BYTES 64

/* Common/ReadsProcessor.cpp:20-37 -- bytes of one packed k-mer key (k=60 -> 15). */
int arks_oracle_key_bytes(int k);

/* Common/ReadsProcessor.cpp:376-535 (prepSeq) + :349-351 (getStr).
 * Writes arks_oracle_key_bytes(k) bytes to out and returns 1, or returns 0 for the NULL k-mer
 * (any character of seq[pos, pos+k) outside ACGTacgt).  Includes the palindrome quirk. */
int arks_oracle_key(const char* seq, size_t pos, int k, unsigned char* out);

/* Counters of the index build, Arcs/Arcs.cpp:179-180 (32-bit, like the reference). */
typedef struct
{
	uint32_t total_kmers;  /* sum of mapKmers() return values, Arcs.cpp:1087,1093 */
	uint32_t null_kmers;   /* s_numbadkmers        Arcs.cpp:924  */
	uint32_t recorded;     /* s_numkmersmapped     Arcs.cpp:919  */
	uint32_t collisions;   /* s_numkmercollisions  Arcs.cpp:915  */
	uint32_t removed_dup;  /* s_numkmersremdup     Arcs.cpp:909  */
	uint32_t unique;       /* s_uniquedraftkmers   Arcs.cpp:911,918 */
} arks_oracle_build_stats;

/* Counters of the read mapping, Arcs/Arcs.cpp:182-185. */
typedef struct
{
	uint64_t total_valid;   /* s_totalnumckmers  Arcs.cpp:966 */
	uint64_t bad;           /* s_numbadckmers    Arcs.cpp:991 */
	uint64_t found;         /* s_numckmersfound  Arcs.cpp:987 */
	uint64_t recorded;      /* s_numckmersrec    Arcs.cpp:977 */
	uint64_t dups;          /* s_ckmersasdups    Arcs.cpp:980 */
	uint64_t reads_pass;    /* s_numreadspassingjaccard Arcs.cpp:1007 */
	uint64_t reads_fail;    /* s_numreadsfailjaccard    Arcs.cpp:1011 */
	uint64_t windows;       /* sum of totalnumkmers (Arcs.cpp:962): the bench metric's unit */
} arks_oracle_map_stats;

typedef struct arks_oracle_index arks_oracle_index;

/* ContigKMap (Arcs/Arcs.h:158): packed key -> contig-end index, 0 = ambiguous. */
arks_oracle_index* arks_oracle_index_new(int k);
void arks_oracle_index_free(arks_oracle_index* idx);
size_t arks_oracle_index_size(const arks_oracle_index* idx);
/* value of key (arks_oracle_key_bytes(k) bytes) or -1 when absent */
int arks_oracle_index_get(const arks_oracle_index* idx, const unsigned char* key);
/* dump all entries: keys (size*key_bytes) and values (size); order unspecified */
void arks_oracle_index_dump(const arks_oracle_index* idx, unsigned char* keys, int32_t* vals);

/* mapKmers, Arcs/Arcs.cpp:869-929 (incl. the i += k jump on a NULL k-mer, :922-925). */
int arks_oracle_map_kmers(
    arks_oracle_index* idx,
    const char* seq,
    int len,
    int conreci,
    arks_oracle_build_stats* st);

/* The same scan (the i += k jumps are those of the whole sequence), but only the windows that start in
 * [lo, hi) are inserted: lets a test give a sub-draft index the whole draft's windows around chosen sites
 * (the (AT)n microsatellites, whose flank + repeat k-mers recur between sites all over a draft) without
 * holding the whole map.  Not a function of the reference; owner-or-0 does not depend on insertion order. */
int arks_oracle_map_kmers_range(
    arks_oracle_index* idx,
    const char* seq,
    int len,
    int conreci,
    int lo,
    int hi);

/* The head/tail split of getContigKmers, Arcs/Arcs.cpp:1056-1093: for a contig of length len
 * writes the cut-off (length of both end substrings); returns 0 when the contig is skipped
 * (len < min_size). */
int arks_oracle_end_cutoff(int len, int min_size, int end_length, int* cutoff);

/* bestContig, Arcs/Arcs.cpp:939-1014. */
int arks_oracle_best_contig(
    const arks_oracle_index* idx,
    const char* read,
    int len,
    double j_index,
    arks_oracle_map_stats* st);

/* checkReadSequence, Arcs/Arcs.cpp:366-389. */
int arks_oracle_check_read_sequence(const char* seq, int len);

/* The per-pair rule of chromiumRead, Arcs/Arcs.cpp:1264-1292, over a batch that the caller has
 * already parsed: reads 2p and 2p+1 are the mates of pair p; pair_ok[p] != 0 iff the pair is
 * `paired && validbarcode && barcode1 == barcode2` (:1264-1265).  out_conreci[r] receives
 * bestContig of read r (0 when not evaluated); out_pair[p] the agreed contig end (c1 != 0 &&
 * c1 == c2) or 0.  Returns the number of stored pairs. */
int64_t arks_oracle_map_pairs(
    const arks_oracle_index* idx,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    int64_t n_pairs,
    const uint8_t* pair_ok,
    double j_index,
    int32_t* out_conreci,
    int32_t* out_pair,
    arks_oracle_map_stats* st,
    int n_threads);

/* The CPU port end to end from a gzipped interleaved FASTQ (arks_port_fastq.c): chromiumRead's pair loop,
 * Arcs/Arcs.cpp:1169-1292, records read inside one critical section, mapped outside of it.  Returns the record
 * pairs read (-1: the file cannot be opened); *st the mapping counters, *stored_pairs the pairs whose mates agree. */
int64_t arks_oracle_map_fastq_gz(
    const arks_oracle_index* idx,
    const char* path,
    double j_index,
    int n_threads,
    arks_oracle_map_stats* st,
    int64_t* stored_pairs);

#ifdef __cplusplus
}
#endif
#endif
