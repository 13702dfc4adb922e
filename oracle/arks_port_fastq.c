/*
 * arks_port_fastq.c -- the CPU port END TO END from a gzipped FASTQ, for bench.py's cpu_baseline leg
 * ("and additionally end-to-end from .fq.gz").  TEST INFRASTRUCTURE ONLY, like
 * everything under oracle/ (see arks_oracle.h).
 *
 * Restates the record-pair loop of chromiumRead, Arcs/Arcs.cpp:1169-1292, with its parallel structure: one
 * OpenMP parallel region; every thread, in a loop, takes the next TWO records of the interleaved file inside
 * one critical section (`#pragma omp critical(checkread1or2)`, :1185 -- kseq_read over gzread there, line
 * reads over gzgets here: the same zlib inflate, serial by construction), then outside of it compares the
 * mates' names (stripReadNum, :243-254), takes the barcode from the BX:Z: tag of the comment (:1225-1251),
 * applies checkReadSequence to both mates (:1264-1268) and calls bestContig twice (:1270-1272).  The
 * barcode multiplicity lookup (:1255-1262; an unordered_map find per pair) and the IndexMap update
 * (:1282-1285; one critical section per stored pair) are left out -- both only make the reference slower.
 */
#include "arks_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LINE_MAX_BYTES (1 << 16)

typedef struct
{
	char name[512];
	char comment[1024];
	char seq[LINE_MAX_BYTES];
	int len;
} record;

/* one 4-line record; 0 at the end of the file (or at a damaged record: kseq stops there too, :1188-1205) */
static int
read_record(gzFile f, record* r, char* scratch)
{
	if (!gzgets(f, scratch, LINE_MAX_BYTES) || scratch[0] != '@')
		return 0;
	size_t n = strcspn(scratch + 1, " \t\r\n");
	if (n >= sizeof r->name)
		n = sizeof r->name - 1;
	memcpy(r->name, scratch + 1, n);
	r->name[n] = 0;
	const char* c = scratch + 1 + n;
	while (*c == ' ' || *c == '\t')
		c++;
	size_t cn = strcspn(c, "\r\n");
	if (cn >= sizeof r->comment)
		cn = sizeof r->comment - 1;
	memcpy(r->comment, c, cn);
	r->comment[cn] = 0;
	if (!gzgets(f, r->seq, LINE_MAX_BYTES))
		return 0;
	r->len = (int)strcspn(r->seq, "\r\n");
	r->seq[r->len] = 0;
	if (!gzgets(f, scratch, LINE_MAX_BYTES) || scratch[0] != '+')
		return 0;
	if (!gzgets(f, scratch, LINE_MAX_BYTES))
		return 0;
	return 1;
}

/* stripReadNum, Arcs/Arcs.cpp:243-254: "/1" or "/2" at the end of the name */
static void
strip_read_num(char* name)
{
	size_t n = strlen(name);
	if (n >= 2 && name[n - 2] == '/' && (name[n - 1] == '1' || name[n - 1] == '2'))
		name[n - 2] = 0;
}

int64_t
arks_oracle_map_fastq_gz(
    const arks_oracle_index* idx,
    const char* path,
    double j_index,
    int n_threads,
    arks_oracle_map_stats* st,
    int64_t* stored_pairs)
{
	gzFile f = gzopen(path, "rb");
	if (!f)
		return -1;
	gzbuffer(f, 1 << 20);
	int64_t pairs = 0, stored = 0;
	int done = 0;
	arks_oracle_map_stats total;
	memset(&total, 0, sizeof total);
#ifdef _OPENMP
	if (n_threads > 0)
		omp_set_num_threads(n_threads);
#else
	(void)n_threads;
#endif
#pragma omp parallel
	{
		record* r1 = (record*)malloc(sizeof(record));
		record* r2 = (record*)malloc(sizeof(record));
		char* scratch = (char*)malloc(LINE_MAX_BYTES);
		arks_oracle_map_stats loc;
		memset(&loc, 0, sizeof loc);
		int64_t loc_pairs = 0, loc_stored = 0;
		for (;;) {
			int got = 0;
#pragma omp critical(checkread1or2)
			{
				if (!done) {
					got = read_record(f, r1, scratch) && read_record(f, r2, scratch);
					if (!got)
						done = 1;
				}
			}
			if (!got)
				break;
			loc_pairs++;
			strip_read_num(r1->name);
			strip_read_num(r2->name);
			if (strcmp(r1->name, r2->name) != 0)
				continue; /* :1213-1222 */
			const char* bx = strstr(r1->comment, "BX:Z:");
			if (!bx || bx[5] == 0 || bx[5] == ' ' || bx[5] == '\t')
				continue; /* no barcode: the pair is not used, :1253 */
			if (!arks_oracle_check_read_sequence(r1->seq, r1->len) || !arks_oracle_check_read_sequence(r2->seq, r2->len))
				continue;
			const int c1 = arks_oracle_best_contig(idx, r1->seq, r1->len, j_index, &loc);
			const int c2 = arks_oracle_best_contig(idx, r2->seq, r2->len, j_index, &loc);
			if (c1 != 0 && c1 == c2)
				loc_stored++;
		}
#pragma omp critical(fold)
		{
			total.total_valid += loc.total_valid;
			total.bad += loc.bad;
			total.found += loc.found;
			total.recorded += loc.recorded;
			total.dups += loc.dups;
			total.reads_pass += loc.reads_pass;
			total.reads_fail += loc.reads_fail;
			total.windows += loc.windows;
			pairs += loc_pairs;
			stored += loc_stored;
		}
		free(r1);
		free(r2);
		free(scratch);
	}
	gzclose(f);
	if (st)
		*st = total;
	if (stored_pairs)
		*stored_pairs = stored;
	return pairs;
}
