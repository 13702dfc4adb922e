/*
 * ref_shim.cpp -- extern "C" access to the REFERENCE's own k-mer encoder, for validating the
 * oracle restatement.  TEST INFRASTRUCTURE ONLY (see arks_oracle.h).
 *
 * Built by oracle/Makefile into oracle/_ref/libarks_ref.so from this file plus the reference's
 * Common/ReadsProcessor.cpp compiled where it lies (never copied into this repo).  Only that one
 * reference translation unit is buildable in this image: Arcs/Arcs.cpp needs Boost.Graph and Google
 * sparsehash, Common/city.cc needs the autoconf-generated config.h -- none of which exist here,
 * and no stand-ins are written for them.  The control flow around the encoder (mapKmers /
 * bestContig, Arcs/Arcs.cpp:869-1014) is therefore restated below over std::unordered_map, which
 * is exact because the path's results depend only on key equality (Arcs/Arcs.h:153-156).
 */
#include "ReadsProcessor.h" /* the reference header, via -I<reference>/Common */

/* Two more std-only reference headers, for the -D distance estimates (the host graph stage's tests):
 * closestKeys and quantile as the reference defines them. */
#include <cassert>
#include "MapUtil.h"
#include "StatUtil.h"
#include <vector>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>

namespace {
struct RefIndex
{
	int k;
	ReadsProcessor proc;
	std::unordered_map<std::string, int> kmap;
	uint32_t total_kmers = 0, null_kmers = 0, recorded = 0, collisions = 0, removed_dup = 0,
	         unique = 0;
	explicit RefIndex(int k_)
	  : k(k_)
	  , proc((unsigned)k_)
	{}
};
}

extern "C" {

void*
ref_proc_new(int k)
{
	return new ReadsProcessor((unsigned)k);
}

void
ref_proc_free(void* h)
{
	delete static_cast<ReadsProcessor*>(h);
}

/* prepSeq + getStr of one window; returns key length, or 0 for the NULL k-mer */
int
ref_proc_key(void* h, const char* seq, size_t len, size_t pos, unsigned char* out)
{
	ReadsProcessor* p = static_cast<ReadsProcessor*>(h);
	std::string s(seq, len);
	const unsigned char* key = p->prepSeq(s, pos);
	if (key == NULL)
		return 0;
	std::string str = p->getStr(key);
	std::memcpy(out, str.data(), str.size());
	return (int)str.size();
}

/* every window of seq: out_keys[i*key_bytes ..], out_valid[i]; returns number of windows */
int64_t
ref_proc_keys_all(
    void* h,
    const char* seq,
    size_t len,
    int k,
    int key_bytes,
    unsigned char* out_keys,
    unsigned char* out_valid)
{
	ReadsProcessor* p = static_cast<ReadsProcessor*>(h);
	std::string s(seq, len);
	int64_t n = 0;
	for (size_t i = 0; i + (size_t)k <= len; ++i, ++n) {
		const unsigned char* key = p->prepSeq(s, i);
		out_valid[n] = key != NULL;
		if (key) {
			std::string str = p->getStr(key);
			std::memcpy(out_keys + (size_t)n * (size_t)key_bytes, str.data(), (size_t)key_bytes);
		} else
			std::memset(out_keys + (size_t)n * (size_t)key_bytes, 0, (size_t)key_bytes);
	}
	return n;
}

void*
ref_index_new(int k)
{
	return new RefIndex(k);
}

void
ref_index_free(void* h)
{
	delete static_cast<RefIndex*>(h);
}

/* control flow of mapKmers, Arcs/Arcs.cpp:869-929, over the reference encoder */
int
ref_index_map_kmers(void* h, const char* seq, int len, int conreci)
{
	RefIndex* ix = static_cast<RefIndex*>(h);
	std::string s(seq, (size_t)len);
	if (len < ix->k)
		return 0;
	int num = 0, i = 0;
	while (i <= len - ix->k) {
		const unsigned char* t = ix->proc.prepSeq(s, (size_t)i);
		if (t != NULL) {
			std::string key = ix->proc.getStr(t);
			num++;
			bool exists = ix->kmap.find(key) != ix->kmap.end();
			int already = ix->kmap[key];
			if (exists) {
				if (already != conreci) {
					ix->removed_dup++;
					if (already != 0) {
						ix->unique--;
						ix->kmap[key] = 0;
					}
				}
				ix->collisions++;
			} else {
				ix->kmap[key] = conreci;
				ix->unique++;
				ix->recorded++;
			}
			i++;
		} else {
			i += ix->k;
			ix->null_kmers++;
		}
	}
	ix->total_kmers += (uint32_t)num;
	return num;
}

void
ref_index_stats(void* h, uint32_t* out6)
{
	RefIndex* ix = static_cast<RefIndex*>(h);
	out6[0] = ix->total_kmers;
	out6[1] = ix->null_kmers;
	out6[2] = ix->recorded;
	out6[3] = ix->collisions;
	out6[4] = ix->removed_dup;
	out6[5] = ix->unique;
}

int64_t
ref_index_size(void* h)
{
	return (int64_t) static_cast<RefIndex*>(h)->kmap.size();
}

int
ref_index_get(void* h, const unsigned char* key, int key_bytes)
{
	RefIndex* ix = static_cast<RefIndex*>(h);
	auto it = ix->kmap.find(std::string(reinterpret_cast<const char*>(key), (size_t)key_bytes));
	return it == ix->kmap.end() ? -1 : it->second;
}

/* control flow of bestContig, Arcs/Arcs.cpp:939-1014, over the reference encoder.
 * counters[0..6] += total_valid, bad, found, recorded, dups, pass, fail; counters[7] += windows */
int
ref_index_best_contig(void* h, const char* read, int len, double j_index, uint64_t* counters)
{
	RefIndex* ix = static_cast<RefIndex*>(h);
	std::string s(read, (size_t)len);
	std::map<int, int> ktrack;
	int total = 0;
	for (int i = 0; i <= len - ix->k; ++i) {
		const unsigned char* t = ix->proc.prepSeq(s, (size_t)i);
		total++;
		if (t != NULL) {
			std::string key = ix->proc.getStr(t);
			if (counters)
				counters[0]++;
			auto it = ix->kmap.find(key);
			if (it != ix->kmap.end()) {
				if (it->second != 0) {
					ktrack[it->second]++;
					if (counters)
						counters[3]++;
				} else if (counters)
					counters[4]++;
				if (counters)
					counters[2]++;
			}
		} else if (counters)
			counters[1]++;
	}
	if (counters)
		counters[7] += (uint64_t)total;
	double maxj = 0;
	int best = 0;
	for (auto it = ktrack.begin(); it != ktrack.end(); ++it) {
		double jac = (double)it->second / (double)total;
		if (maxj < jac) {
			maxj = jac;
			best = it->first;
		}
	}
	if (maxj > j_index) {
		if (counters)
			counters[5]++;
		return best;
	}
	if (counters)
		counters[6]++;
	return 0;
}

/* closestKeys(map<double, int>, key, n) of Common/MapUtil.h over the sorted, distinct keys[0..n_keys):
 * the positions [first, last) of the returned iterator range */
void
ref_closest_keys(const double* keys, int n_keys, double key, int n, int* first, int* last)
{
	std::map<double, int> m;
	for (int i = 0; i < n_keys; ++i)
		m[keys[i]] = i;
	auto r = closestKeys(m, key, (size_t)n);
	*first = r.first == m.end() ? n_keys : r.first->second;
	*last = r.second == m.end() ? n_keys : r.second->second;
}

/* quantile() of Common/StatUtil.h over a sorted vector<unsigned>, as estimateDistance calls it */
double
ref_quantile(const unsigned* vals, int n, double q)
{
	std::vector<unsigned> v(vals, vals + n);
	return quantile(v.begin(), v.end(), q);
}

} /* extern "C" */
