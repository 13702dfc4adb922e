/*
 * arks_oracle.c -- CPU restatement (plain C) of the ARKS k-mer mapping path of bcgsc/arcs.
 * TEST INFRASTRUCTURE ONLY -- see arks_oracle.h.  Parity status: pinned (header of arks_oracle.h).
 *
 * Written from the behaviour of the reference, not from its text: the key function is the closed
 * form that SURVEY.md section 8 derives for Common/ReadsProcessor.cpp:376-535 and that
 * tests/test_oracle_vs_ref.py fuzzes against the compiled reference encoder.
 */
#include "arks_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- 2-bit code of a base; 0xFF = not one of ACGTacgt (ReadsProcessor.cpp:39-317 LUT rows) ---- */
static unsigned char g_code[256];
static int g_code_ready = 0;

static void
init_code(void)
{
	if (g_code_ready)
		return;
	memset(g_code, 0xFF, sizeof g_code);
	g_code['A'] = g_code['a'] = 0;
	g_code['C'] = g_code['c'] = 1;
	g_code['G'] = g_code['g'] = 2;
	g_code['T'] = g_code['t'] = 3;
	g_code_ready = 1;
}

/* Common/ReadsProcessor.cpp:20-37 */
int
arks_oracle_key_bytes(int k)
{
	return k / 4 + ((k % 4) ? 1 : 0);
}

/* 4 codes starting at c[0] packed MSB-first into one byte */
static unsigned char
pack4(const unsigned char* c)
{
	return (unsigned char)((c[0] << 6) | (c[1] << 4) | (c[2] << 2) | c[3]);
}

/* pack n codes, 4 per byte, first code in bits 7:6, last byte zero padded */
static void
pack_codes(const unsigned char* c, int n, unsigned char* out)
{
	int nbytes = arks_oracle_key_bytes(n);
	memset(out, 0, (size_t)nbytes);
	for (int i = 0; i < n; ++i)
		out[i >> 2] |= (unsigned char)(c[i] << (6 - 2 * (i & 3)));
}

/* Common/ReadsProcessor.cpp:376-535 (prepSeq) */
int
arks_oracle_key(const char* seq, size_t pos, int k, unsigned char* out)
{
	unsigned char f[4 * ARKS_ORACLE_MAX_KEY_BYTES], r[4 * ARKS_ORACLE_MAX_KEY_BYTES];
	init_code();
	/* rule 1: any character outside ACGTacgt anywhere in the window -> NULL (:397-404,:417-421) */
	for (int i = 0; i < k; ++i) {
		unsigned char c = g_code[(unsigned char)seq[pos + (size_t)i]];
		if (c == 0xFF)
			return 0;
		f[i] = c;
	}
	for (int i = 0; i < k; ++i)
		r[i] = (unsigned char)(3 - f[k - 1 - i]);
	int cmp = 0;
	for (int i = 0; i < k && cmp == 0; ++i)
		cmp = (int)f[i] - (int)r[i];
	if (cmp < 0) { /* forward strand is smaller (:427-462) */
		pack_codes(f, k, out);
		return 1;
	}
	if (cmp > 0) { /* reverse complement is smaller (:464-500) */
		pack_codes(r, k, out);
		return 1;
	}
	/* rule 3: reverse-complement palindrome, the reference's damaged branch (:503-534).
	 * bytes [0, half) are the forward bytes, byte `half` stays 0, later bytes are filled from a
	 * cursor that advances 3 bases (not 4) per byte; a hanging last byte degenerates to one base. */
	{
		const int full = k / 4, hang = k % 4, nbytes = arks_oracle_key_bytes(k);
		const int half = k / 8 + ((k % 8) ? 1 : 0);
		memset(out, 0, (size_t)nbytes);
		for (int b = 0; b < half; ++b)
			out[b] = pack4(f + 4 * b);
		int idx = 4 * half; /* cursor after the first loop (:394-396, :408-410) */
		for (int b = half + 1; b < full; ++b) { /* :505-519: three ++, then OR without ++ */
			out[b] = pack4(f + idx);
			idx += 3;
		}
		if (hang) { /* :521-533 */
			/* m_fw[out] |= fw0[s[lastPos]]; then while (idx < lastPos) { <<= 2; |= fw0[s[--lastPos]] }
			 * : every shift pushes the earlier bases out of the byte's top, so what survives is
			 * fw0 of the last base visited, i.e. s[pos + idx + 1] -- unless the loop never runs. */
			int last = k - 1;
			unsigned char v = (unsigned char)(f[last] << 6);
			for (; idx < last; --last) {
				v = (unsigned char)(v << 2);
				v |= (unsigned char)(f[last] << 6);
			}
			out[full] = v;
		}
	}
	return 1;
}

/* ---------------------------------------------------------------------------------------------
 * ContigKMap stand-in: exact open-addressed map  key bytes -> int.  Results of the path depend only
 * on exact key equality (Arcs/Arcs.h:153-156), never on the hash or the table layout.
 * ------------------------------------------------------------------------------------------- */
struct arks_oracle_index
{
	int k;
	int key_bytes;
	size_t cap;  /* power of two */
	size_t size;
	unsigned char* keys; /* cap * key_bytes */
	int32_t* vals;
	unsigned char* used;
};

static uint64_t
hash_bytes(const unsigned char* p, int n)
{
	uint64_t h = 0x9E3779B97F4A7C15ull;
	int i = 0;
	for (; i + 8 <= n; i += 8) {
		uint64_t w;
		memcpy(&w, p + i, 8);
		h = (h ^ w) * 0xff51afd7ed558ccdull;
		h ^= h >> 32;
	}
	uint64_t w = 0;
	if (i < n)
		memcpy(&w, p + i, (size_t)(n - i));
	h = (h ^ w ^ (uint64_t)n) * 0xc4ceb9fe1a85ec53ull;
	h ^= h >> 29;
	h *= 0xff51afd7ed558ccdull;
	h ^= h >> 32;
	return h;
}

arks_oracle_index*
arks_oracle_index_new(int k)
{
	if (k <= 3 || arks_oracle_key_bytes(k) > ARKS_ORACLE_MAX_KEY_BYTES) /* ReadsProcessor.cpp:25 */
		return NULL;
	arks_oracle_index* idx = (arks_oracle_index*)calloc(1, sizeof *idx);
	idx->k = k;
	idx->key_bytes = arks_oracle_key_bytes(k);
	idx->cap = 1024;
	idx->keys = (unsigned char*)malloc(idx->cap * (size_t)idx->key_bytes);
	idx->vals = (int32_t*)malloc(idx->cap * sizeof(int32_t));
	idx->used = (unsigned char*)calloc(idx->cap, 1);
	return idx;
}

void
arks_oracle_index_free(arks_oracle_index* idx)
{
	if (!idx)
		return;
	free(idx->keys);
	free(idx->vals);
	free(idx->used);
	free(idx);
}

size_t
arks_oracle_index_size(const arks_oracle_index* idx)
{
	return idx->size;
}

/* slot of key, or the empty slot where it would go */
static size_t
find_slot(const arks_oracle_index* idx, const unsigned char* key)
{
	size_t mask = idx->cap - 1;
	size_t s = (size_t)hash_bytes(key, idx->key_bytes) & mask;
	while (idx->used[s] &&
	       memcmp(idx->keys + s * (size_t)idx->key_bytes, key, (size_t)idx->key_bytes) != 0)
		s = (s + 1) & mask;
	return s;
}

static void
grow(arks_oracle_index* idx)
{
	arks_oracle_index old = *idx;
	idx->cap = old.cap * 2;
	idx->keys = (unsigned char*)malloc(idx->cap * (size_t)idx->key_bytes);
	idx->vals = (int32_t*)malloc(idx->cap * sizeof(int32_t));
	idx->used = (unsigned char*)calloc(idx->cap, 1);
	for (size_t i = 0; i < old.cap; ++i) {
		if (!old.used[i])
			continue;
		const unsigned char* key = old.keys + i * (size_t)idx->key_bytes;
		size_t s = find_slot(idx, key);
		idx->used[s] = 1;
		memcpy(idx->keys + s * (size_t)idx->key_bytes, key, (size_t)idx->key_bytes);
		idx->vals[s] = old.vals[i];
	}
	free(old.keys);
	free(old.vals);
	free(old.used);
}

int
arks_oracle_index_get(const arks_oracle_index* idx, const unsigned char* key)
{
	size_t s = find_slot(idx, key);
	return idx->used[s] ? idx->vals[s] : -1;
}

void
arks_oracle_index_dump(const arks_oracle_index* idx, unsigned char* keys, int32_t* vals)
{
	size_t n = 0;
	for (size_t i = 0; i < idx->cap; ++i) {
		if (!idx->used[i])
			continue;
		memcpy(keys + n * (size_t)idx->key_bytes, idx->keys + i * (size_t)idx->key_bytes,
		       (size_t)idx->key_bytes);
		vals[n++] = idx->vals[i];
	}
}

/* mapKmers, Arcs/Arcs.cpp:869-929 */
/* the scan of mapKmers over the whole sequence; only the windows that start in [lo, hi) are put into the map */
static int
map_kmers_scan(
    arks_oracle_index* idx,
    const char* seq,
    int len,
    int conreci,
    int lo,
    int hi,
    arks_oracle_build_stats* st)
{
	unsigned char key[ARKS_ORACLE_MAX_KEY_BYTES];
	const int k = idx->k;
	int num = 0;
	if (len < k) /* :877-882 (the reference also prints a warning) */
		return 0;
	int i = 0;
	init_code();
	if (lo > 0) {
		/* arks_oracle_map_kmers_range: the windows in front of lo only steer the walk (i + 1 after a window
		 * without an invalid character, i + k after one with: the only NULL condition of the key function),
		 * so no key is made for them */
		int next_bad = -1; /* first invalid position >= i, found lazily */
		while (i < lo && i <= len - k) {
			if (next_bad < i) {
				next_bad = i;
				while (next_bad < len && g_code[(unsigned char)seq[next_bad]] != 0xFF)
					next_bad++;
			}
			if (next_bad >= i + k)
				i = next_bad - k + 1 < lo ? next_bad - k + 1 : lo; /* every window up to there is valid */
			else
				i += k;
		}
	}
	while (i <= len - k) { /* :887 */
		if (i >= hi)
			break; /* (range variant) nothing behind hi is asked for */
		if (arks_oracle_key(seq, (size_t)i, k, key)) {
			num++;
			if ((idx->size + 1) * 2 > idx->cap)
				grow(idx);
			size_t s = find_slot(idx, key);
			if (idx->used[s]) { /* :907-915 */
				if (idx->vals[s] != conreci) {
					if (st)
						st->removed_dup++;
					if (idx->vals[s] != 0) {
						if (st)
							st->unique--;
						idx->vals[s] = 0;
					}
				}
				if (st)
					st->collisions++;
			} else { /* :916-920 */
				idx->used[s] = 1;
				memcpy(idx->keys + s * (size_t)idx->key_bytes, key, (size_t)idx->key_bytes);
				idx->vals[s] = conreci;
				idx->size++;
				if (st) {
					st->unique++;
					st->recorded++;
				}
			}
			i++;
		} else { /* :922-925 -- jumps k, not 1 */
			i += k;
			if (st)
				st->null_kmers++;
		}
	}
	if (st)
		st->total_kmers += (uint32_t)num;
	return num;
}

int
arks_oracle_map_kmers(
    arks_oracle_index* idx,
    const char* seq,
    int len,
    int conreci,
    arks_oracle_build_stats* st)
{
	return map_kmers_scan(idx, seq, len, conreci, 0, len, st);
}

int
arks_oracle_map_kmers_range(
    arks_oracle_index* idx,
    const char* seq,
    int len,
    int conreci,
    int lo,
    int hi)
{
	return map_kmers_scan(idx, seq, len, conreci, lo, hi, NULL);
}

/* Arcs/Arcs.cpp:1056, :1072-1074 */
int
arks_oracle_end_cutoff(int len, int min_size, int end_length, int* cutoff)
{
	if (len < min_size)
		return 0;
	int c = end_length;
	if (c == 0 || len <= c * 2)
		c = len / 2;
	*cutoff = c;
	return 1;
}

/* bestContig, Arcs/Arcs.cpp:939-1014 */
int
arks_oracle_best_contig(
    const arks_oracle_index* idx,
    const char* read,
    int len,
    double j_index,
    arks_oracle_map_stats* st)
{
	unsigned char key[ARKS_ORACLE_MAX_KEY_BYTES];
	const int k = idx->k;
	/* ktrack (std::map<int,int>, :946): distinct non-zero values of this read and their counts */
	int tv_small[16], tc_small[16];
	int *tv = tv_small, *tc = tc_small, tn = 0, tcap = 16;
	int total = 0;
	for (int i = 0; i <= len - k; ++i) { /* :959 */
		int ok = arks_oracle_key(read, (size_t)i, k, key);
		total++; /* :962 -- NULL windows count too */
		if (!ok) {
			if (st)
				st->bad++;
			continue;
		}
		if (st)
			st->total_valid++;
		int v = arks_oracle_index_get(idx, key);
		if (v < 0)
			continue;
		if (st)
			st->found++;
		if (v == 0) {
			if (st)
				st->dups++;
			continue;
		}
		if (st)
			st->recorded++;
		int t = 0;
		while (t < tn && tv[t] != v)
			t++;
		if (t == tn) {
			if (tn == tcap) {
				int* nv = (int*)malloc(sizeof(int) * (size_t)tcap * 2);
				int* nc = (int*)malloc(sizeof(int) * (size_t)tcap * 2);
				memcpy(nv, tv, sizeof(int) * (size_t)tn);
				memcpy(nc, tc, sizeof(int) * (size_t)tn);
				if (tv != tv_small) {
					free(tv);
					free(tc);
				}
				tv = nv;
				tc = nc;
				tcap *= 2;
			}
			tv[tn] = v;
			tc[tn] = 0;
			tn++;
		}
		tc[t]++;
	}
	if (st)
		st->windows += (uint64_t)total;
	/* :996-1004 -- ascending key order with a strict '<' => ties go to the smallest index */
	double maxj = 0;
	int best = 0;
	for (int t = 0; t < tn; ++t) {
		double jac = (double)tc[t] / (double)total;
		if (maxj < jac || (maxj == jac && best != 0 && tv[t] < best)) {
			maxj = jac;
			best = tv[t];
		}
	}
	if (tv != tv_small) {
		free(tv);
		free(tc);
	}
	if (maxj > j_index) { /* :1006 strict */
		if (st)
			st->reads_pass++;
		return best;
	}
	if (st)
		st->reads_fail++;
	return 0;
}

/* checkReadSequence, Arcs/Arcs.cpp:366-389 */
int
arks_oracle_check_read_sequence(const char* seq, int len)
{
	double ambiguity = 0;
	for (int i = 0; i < len; ++i) {
		char c = seq[i];
		if (c >= 'a' && c <= 'z')
			c = (char)(c - 'a' + 'A');
		if (c != 'A' && c != 'T' && c != 'G' && c != 'C') {
			if (c == 'N')
				ambiguity++;
			else
				return 0;
		}
	}
	double ar = ambiguity / (double)len;
	return !(ar > 0.02);
}

/* the per-pair rule of chromiumRead, Arcs/Arcs.cpp:1264-1292 */
int64_t
arks_oracle_map_pairs(
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
    int n_threads)
{
	int64_t stored = 0;
	arks_oracle_map_stats total;
	memset(&total, 0, sizeof total);
	init_code();
#ifdef _OPENMP
	if (n_threads > 0)
		omp_set_num_threads(n_threads);
#else
	(void)n_threads;
#endif
#pragma omp parallel
	{
		arks_oracle_map_stats loc;
		memset(&loc, 0, sizeof loc);
		int64_t loc_stored = 0;
#pragma omp for schedule(dynamic, 256)
		for (int64_t p = 0; p < n_pairs; ++p) {
			int c1 = 0, c2 = 0;
			if (!pair_ok || pair_ok[p]) {
				const char* r1 = bases + offsets[2 * p];
				const char* r2 = bases + offsets[2 * p + 1];
				int l1 = (int)lens[2 * p], l2 = (int)lens[2 * p + 1];
				/* :1267 goodmult is always true ("||"), so only the sequence filter gates */
				if (arks_oracle_check_read_sequence(r1, l1) &&
				    arks_oracle_check_read_sequence(r2, l2)) {
					c1 = arks_oracle_best_contig(idx, r1, l1, j_index, &loc);
					c2 = arks_oracle_best_contig(idx, r2, l2, j_index, &loc);
				}
			}
			if (out_conreci) {
				out_conreci[2 * p] = c1;
				out_conreci[2 * p + 1] = c2;
			}
			int agreed = (c1 != 0 && c1 == c2) ? c1 : 0; /* :1280 */
			if (out_pair)
				out_pair[p] = agreed;
			if (agreed)
				loc_stored++;
		}
#pragma omp critical
		{
			total.total_valid += loc.total_valid;
			total.bad += loc.bad;
			total.found += loc.found;
			total.recorded += loc.recorded;
			total.dups += loc.dups;
			total.reads_pass += loc.reads_pass;
			total.reads_fail += loc.reads_fail;
			total.windows += loc.windows;
			stored += loc_stored;
		}
	}
	if (st) {
		st->total_valid += total.total_valid;
		st->bad += total.bad;
		st->found += total.found;
		st->recorded += total.recorded;
		st->dups += total.dups;
		st->reads_pass += total.reads_pass;
		st->reads_fail += total.reads_fail;
		st->windows += total.windows;
	}
	return stored;
}
