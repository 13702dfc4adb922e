// rank_merge.hpp -- the read stage's results as data: what one rank (one process, one GPU) hands to rank 0,
// how it travels through a pipe, and how rank 0 turns the results of all ranks into the stage's log, the
// barcode multiplicities of the fused mode and the IndexMap (Arcs/Arcs.h:108-113) of every k.  No device
// code here: tests/test_host_ranks.py drives it on the CPU with two processes.
#pragma once

#include "arks_hip.h"
#include "graph_fast.hpp"
#include "ingest.hpp"

#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace arks_host {

#ifndef ARKS_PROGRAM_NAME
#define ARKS_PROGRAM_NAME "arcs"
#endif

// printf into a string: in the fused barcode mode (no -u) the stages do not run in the order the
// reference prints them, so their stdout text is collected and emitted in the reference's order
inline void
appendf(std::string& out, const char* fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	const int n = vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (n > 0)
		out.append(buf, (size_t)std::min<int>(n, (int)sizeof buf - 1));
}

// ---- the read stage in two parts, so that it can run as one process per GPU ---------------------------
// map_files: what ONE rank does -- its share of the read files through the ingest pipeline and its GPU.
// merge_results: what rank 0 does with the results of all ranks -- the stage's log in file order, the
// barcode multiplicities of the fused mode, and the IndexMap of every k.  A single process is one rank.
struct FileResult
{
	bool have = false;
	FileCounters fc;
	std::map<int64_t, std::string> messages; // by batch number within the file
	std::vector<uint64_t> stored;            // per k
	std::vector<arks_map_stats> st;          // per k
	// fused mode: what readBarcodes would have seen in this file (PrepassInfo, counts by barcode NAME id)
	bool zero_len = false;
	uint64_t pre_total = 0, pre_lead = 0;
	std::vector<uint64_t> untagged_at;
	std::vector<std::pair<uint32_t, uint32_t>> pre_counts; // (id into RankResult::names, reads)
};

struct RankResult
{
	bool redo = false;
	std::vector<FileResult> files;  // one per input file; only this rank's are filled
	std::vector<std::string> names; // barcode id -> text, the ids of the triples and of pre_counts
	// per k: (barcode id, conreci, count) x n, and the sequence number of each entry's first stored pair
	std::vector<std::vector<uint32_t>> triples;
	std::vector<std::vector<uint64_t>> first;
};

// The IndexMap entries of several lanes -- each list sorted by (barcode id, conreci), as arks_imap_export_ordered
// gives them -- into one: counts add up, the first stored pair of an entry is the earliest of the lanes'.
inline void
merge_lane_entries(
    const std::vector<std::vector<uint32_t>>& lt, const std::vector<std::vector<uint64_t>>& lf, std::vector<uint32_t>& triples,
    std::vector<uint64_t>& first)
{
	const size_t L = lt.size();
	std::vector<size_t> cur(L, 0);
	size_t total = 0;
	for (size_t l = 0; l < L; ++l)
		total += lf[l].size();
	triples.clear(), first.clear();
	triples.reserve(total * 3), first.reserve(total);
	auto key = [&](size_t l) { return ((uint64_t)lt[l][3 * cur[l]] << 32) | (uint64_t)lt[l][3 * cur[l] + 1]; };
	for (;;) {
		bool any = false;
		uint64_t best = 0;
		for (size_t l = 0; l < L; ++l)
			if (cur[l] < lf[l].size() && (!any || key(l) < best)) {
				best = key(l);
				any = true;
			}
		if (!any)
			break;
		uint64_t count = 0, fp = ~0ull;
		for (size_t l = 0; l < L; ++l)
			if (cur[l] < lf[l].size() && key(l) == best) {
				count += lt[l][3 * cur[l] + 2];
				fp = std::min(fp, lf[l][cur[l]]);
				cur[l]++;
			}
		triples.push_back((uint32_t)(best >> 32));
		triples.push_back((uint32_t)best);
		triples.push_back((uint32_t)count);
		first.push_back(fp);
	}
}

// rank results <-> a byte stream (the pipe between a worker rank and rank 0)
struct ByteSink
{
	int fd;
	std::vector<char> buf;
	void put(const void* p, size_t n)
	{
		buf.insert(buf.end(), (const char*)p, (const char*)p + n);
		if (buf.size() > (1u << 20))
			flush();
	}
	template <typename T> void pod(const T& v) { put(&v, sizeof v); }
	void str(const std::string& s)
	{
		pod<uint64_t>(s.size());
		put(s.data(), s.size());
	}
	template <typename T> void vec(const std::vector<T>& v)
	{
		pod<uint64_t>(v.size());
		if (!v.empty())
			put(v.data(), v.size() * sizeof(T));
	}
	void flush()
	{
		size_t done = 0;
		while (done < buf.size()) {
			const ssize_t w = ::write(fd, buf.data() + done, buf.size() - done);
			if (w <= 0) {
				std::cerr << ARKS_PROGRAM_NAME ": cannot send a rank's results\n";
				_exit(EXIT_FAILURE);
			}
			done += (size_t)w;
		}
		buf.clear();
	}
};

struct ByteSource
{
	int fd;
	bool ok = true;
	void get(void* p, size_t n)
	{
		size_t done = 0;
		while (ok && done < n) {
			const ssize_t r = ::read(fd, (char*)p + done, n - done);
			if (r <= 0)
				ok = false;
			else
				done += (size_t)r;
		}
	}
	template <typename T> T pod()
	{
		T v{};
		get(&v, sizeof v);
		return v;
	}
	std::string str()
	{
		std::string s((size_t)pod<uint64_t>(), '\0');
		if (ok && !s.empty())
			get(&s[0], s.size());
		return s;
	}
	template <typename T> void vec(std::vector<T>& v)
	{
		const uint64_t n = pod<uint64_t>();
		if (!ok)
			return;
		v.resize((size_t)n);
		if (n)
			get(v.data(), (size_t)n * sizeof(T));
	}
};

inline void
send_result(int fd, const RankResult& r)
{
	ByteSink o{ fd, {} };
	o.pod<uint8_t>(r.redo);
	o.pod<uint64_t>(r.files.size());
	for (const FileResult& f : r.files) {
		o.pod<uint8_t>(f.have);
		if (!f.have)
			continue;
		o.pod(f.fc);
		o.pod<uint64_t>(f.messages.size());
		for (const auto& kv : f.messages) {
			o.pod<int64_t>(kv.first);
			o.str(kv.second);
		}
		o.vec(f.stored);
		o.vec(f.st);
		o.pod<uint8_t>(f.zero_len);
		o.pod(f.pre_total);
		o.pod(f.pre_lead);
		o.vec(f.untagged_at);
		o.vec(f.pre_counts);
	}
	o.pod<uint64_t>(r.names.size());
	for (const std::string& s : r.names)
		o.str(s);
	o.pod<uint64_t>(r.triples.size());
	for (size_t k = 0; k < r.triples.size(); ++k) {
		o.vec(r.triples[k]);
		o.vec(r.first[k]);
	}
	o.flush();
}

inline bool
receive_result(int fd, RankResult& r)
{
	ByteSource in{ fd };
	r.redo = in.pod<uint8_t>() != 0;
	r.files.resize((size_t)in.pod<uint64_t>());
	for (FileResult& f : r.files) {
		f.have = in.pod<uint8_t>() != 0;
		if (!f.have || !in.ok)
			continue;
		f.fc = in.pod<FileCounters>();
		const uint64_t nm = in.pod<uint64_t>();
		for (uint64_t i = 0; i < nm && in.ok; ++i) {
			const int64_t seq = in.pod<int64_t>();
			f.messages[seq] = in.str();
		}
		in.vec(f.stored);
		in.vec(f.st);
		f.zero_len = in.pod<uint8_t>() != 0;
		f.pre_total = in.pod<uint64_t>();
		f.pre_lead = in.pod<uint64_t>();
		in.vec(f.untagged_at);
		in.vec(f.pre_counts);
	}
	r.names.resize((size_t)in.pod<uint64_t>());
	for (std::string& s : r.names)
		s = in.str();
	const size_t nk = (size_t)in.pod<uint64_t>();
	r.triples.resize(nk);
	r.first.resize(nk);
	for (size_t k = 0; k < nk && in.ok; ++k) {
		in.vec(r.triples[k]);
		in.vec(r.first[k]);
	}
	return in.ok;
}

// what the barcode pre-pass (readBarcodes, Arcs.cpp:481-547) prints, rebuilt from the per-file
// summaries of the fused pass; `added` is its global running count of tagged reads
struct MergeParams
{
	bool verbose;
	std::vector<int> k_list;
	int index_shards;
	unsigned threads = 1; // for the sorts of the number-based merge
};

inline void
prepass_log(
    const std::vector<std::string>& files, const std::vector<const FileResult*>& pre, const MergeParams& params,
    std::string& out, std::string& err)
{
	const uint64_t step = 100000000;
	uint64_t added = 0;
	for (size_t f = 0; f < files.size(); ++f) {
		if (params.verbose)
			out += "Reading chrom " + files[f] + "\n";
		err += "File " + files[f] + " opened.\n";
		if (params.verbose) {
			// the progress line appears at every record with a comment while added % step == 0
			const FileResult& pi = *pre[f];
			if (added % step == 0)
				for (uint64_t i = 0; i < pi.pre_lead; ++i)
					out += std::to_string(added) + " read with valid barcode\n";
			size_t e = 0;
			for (uint64_t x = step - added % step; x <= pi.pre_total; x += step) {
				uint64_t count = 1;
				while (e < pi.untagged_at.size() && pi.untagged_at[e] < x)
					e++;
				while (e < pi.untagged_at.size() && pi.untagged_at[e] == x)
					e++, count++;
				for (uint64_t i = 0; i < count; ++i)
					out += std::to_string(added + x) + " read with valid barcode\n";
			}
		}
		added += pre[f]->pre_total;
	}
}

// results of all ranks (every file filled by exactly one) -> the log of the stage, the multiplicities
// of the fused mode, the IndexMap of every k
inline void
merge_results(
    const std::vector<std::string>& files, const std::vector<RankResult>& ranks, std::vector<IndexMap>& imaps,
    std::unordered_map<std::string, int>& mult, const std::vector<CI>& contigRecord, bool fused, const MergeParams& params,
    std::string& out, std::string& err, std::string* pre_out, std::string* pre_err, std::vector<CompactIndex>* compact = nullptr)
{
	const size_t nf = files.size();
	const size_t nk = ranks.empty() ? 0 : ranks[0].triples.size();
	std::vector<const FileResult*> fr(nf, nullptr);
	std::vector<const RankResult*> owner(nf, nullptr);
	for (const RankResult& r : ranks)
		for (size_t f = 0; f < nf && f < r.files.size(); ++f)
			if (r.files[f].have) {
				fr[f] = &r.files[f];
				owner[f] = &r;
			}
	if (fused) {
		// reads per barcode as readBarcodes would have counted them, file by file
		for (size_t f = 0; f < nf; ++f)
			for (const auto& ic : fr[f]->pre_counts)
				mult[owner[f]->names[ic.first]] += (int)ic.second;
		prepass_log(files, fr, params, *pre_out, *pre_err);
		if (params.verbose)
			*pre_out += "Saw " + std::to_string(mult.size()) + " distinct barcode.\n";
	}
	// the log of the stage, file by file as the reference prints it (Arcs.cpp:1158-1166, 1209-1215,
	// 1321-1349); its s_* k-mer counters are process-wide, i.e. cumulative over the files
	std::vector<arks_map_stats> cum(nk);
	std::memset(cum.data(), 0, nk * sizeof(arks_map_stats));
	for (size_t f = 0; f < nf; ++f) {
		if (params.verbose)
			out += "Reading chrom " + files[f] + "\n";
		err += "File " + files[f] + " opened.\n";
		for (const auto& kv : fr[f]->messages)
			out += kv.second;
		for (size_t ki = 0; ki < nk; ++ki) {
			const arks_map_stats& s = fr[f]->st[ki];
			arks_map_stats& c = cum[ki];
			c.total_valid += s.total_valid, c.bad += s.bad, c.found += s.found, c.recorded += s.recorded,
			    c.dups += s.dups, c.reads_pass += s.reads_pass, c.reads_fail += s.reads_fail, c.windows += s.windows;
			if (!params.verbose)
				continue;
			const FileCounters& mc = fr[f]->fc;
			const uint64_t stored = fr[f]->stored[ki];
			if (nk > 1)
				appendf(out, "k = %d:\n", params.k_list[ki]);
			appendf(out, "Stored read pairs: %u\nSkipped invalid read pairs: %u\nSkipped unpaired reads: "
			       "%u\nSkipped reads pairs without a good contig: %u\n",
			       (unsigned)stored, (unsigned)mc.skipped_invalid, (unsigned)mc.skipped_unpaired,
			       (unsigned)(mc.gated - stored));
			// (--index-shards: the parts' counters folded by the read stage, arcs.cpp)
			appendf(out, "Total valid kmers: %u\nNumber invalid kmers: %u\nNumber of kmers found in ContigKmap: "
			       "%u\nNumber of kmers recorded in Ktrack: %u\nNumber of kmers found in ContigKmap but "
			       "duplicate: %u\nNumber of reads passing jaccard threshold: %u\nNumber of reads failing "
			       "jaccard threshold: %u\n",
			       (unsigned)c.total_valid, (unsigned)c.bad, (unsigned)c.found, (unsigned)c.recorded,
			       (unsigned)c.dups, (unsigned)c.reads_pass, (unsigned)c.reads_fail);
			if (mc.emptybarcode > 0)
				appendf(out, "WARNING:: Your chromium read file has %d readpairs that have an empty barcode.",
				       (int)mc.emptybarcode);
			if (mc.invalidbarcode > 0)
				appendf(out, "WARNING:: Your chromium read file has %d read pairs that have barcodes not in the "
				       "barcode multiplicity file.",
				       (int)mc.invalidbarcode);
		}
	}
	// The reference completes the IndexMap after every file (Arcs.cpp:1304-1319); the accumulators are
	// additive, so the rebuild below after the last file gives the same map.  Barcodes enter the unordered
	// IndexMap in the order of their first stored pair (pairs are numbered file, batch, pair), as in a
	// single-threaded reference run: the container's iteration order -- which -D's tie handling sees,
	// Arcs/DistanceEst.h:230-262 -- is then the reference's for the same libstdc++.
	if (compact) {
		// graph_fast.hpp: the same content as numbers (no IndexMap is built; -D needs the map itself and does
		// not come here).  Barcodes are numbered over all ranks, by name; one rank's numbers are kept as they are.
		std::vector<std::vector<uint32_t>> gid(ranks.size());
		size_t n_gid = 0;
		if (ranks.size() == 1)
			n_gid = ranks[0].names.size();
		else {
			std::unordered_map<std::string_view, uint32_t> gid_of;
			for (size_t r = 0; r < ranks.size(); ++r) {
				gid[r].resize(ranks[r].names.size());
				for (size_t i = 0; i < ranks[r].names.size(); ++i)
					gid[r][i] = gid_of.emplace(std::string_view(ranks[r].names[i]), (uint32_t)gid_of.size()).first->second;
			}
			n_gid = gid_of.size();
		}
		std::vector<int> mult_of(n_gid, 0);
		for (size_t r = 0; r < ranks.size(); ++r) {
			const std::vector<std::string>& names = ranks[r].names;
			const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(16, names.size() >> 16));
			std::vector<std::thread> th;
			for (unsigned t = 0; t < T; ++t)
				th.emplace_back([&, t] { // (finds only: the map is not changed)
					for (size_t i = names.size() * t / T; i < names.size() * (t + 1) / T; ++i) {
						const auto m = mult.find(names[i]);
						if (m != mult.end())
							mult_of[ranks.size() == 1 ? i : gid[r][i]] = m->second;
					}
				});
			for (auto& x : th)
				x.join();
		}
		compact->clear();
		for (size_t ki = 0; ki < nk; ++ki) {
			std::vector<RawEntry> raw;
			for (size_t r = 0; r < ranks.size(); ++r) {
				const std::vector<uint32_t>& t = ranks[r].triples[ki];
				raw.reserve(raw.size() + t.size() / 3);
				for (size_t i = 0; 3 * i < t.size(); ++i)
					raw.push_back(RawEntry{ ranks.size() == 1 ? t[3 * i] : gid[r][t[3 * i]], t[3 * i + 1], t[3 * i + 2] });
			}
			compact->push_back(build_compact_index(raw, contigRecord, mult_of, params.threads));
		}
		imaps.clear();
		return;
	}
	imaps.assign(nk, IndexMap());
	for (size_t ki = 0; ki < nk; ++ki) {
		struct Entry
		{
			const std::string* barcode;
			uint32_t conreci, count;
			uint64_t first;
		};
		std::vector<Entry> ent;
		for (const RankResult& r : ranks) {
			const std::vector<uint32_t>& t = r.triples[ki];
			for (size_t i = 0; 3 * i < t.size(); ++i)
				ent.push_back(Entry{ &r.names[t[3 * i]], t[3 * i + 1], t[3 * i + 2], r.first[ki][i] });
		}
		// first stored pair of every barcode over all ranks
		std::unordered_map<std::string_view, uint64_t> first_of;
		first_of.reserve(ent.size());
		for (const Entry& e : ent) {
			auto it = first_of.emplace(std::string_view(*e.barcode), e.first).first;
			it->second = std::min(it->second, e.first);
		}
		std::vector<std::pair<uint64_t, std::string_view>> order;
		order.reserve(first_of.size());
		for (const auto& kv : first_of)
			order.emplace_back(kv.second, kv.first);
		std::sort(order.begin(), order.end());
		IndexMap& imap = imaps[ki];
		for (const auto& o : order)
			imap[std::string(o.second)]; // creation order
		for (const Entry& e : ent)
			imap[*e.barcode][contigRecord[e.conreci]] += (int)e.count;
		add_opposite_ends(imap);
	}
}


} // namespace arks_host
