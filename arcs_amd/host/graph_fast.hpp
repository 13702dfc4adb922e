// graph_fast.hpp -- the stages between the read stage and the graph (IndexMap -> PairMap -> graph, _main.tsv,
// pair TSV) on integers instead of strings.
//
// graph.hpp restates the reference literally: unordered_map<barcode string, map<(contig string, bool), int>>
// and map<pair<string, string>, vector<unsigned>> (Arcs/Arcs.h:105-113, Arcs.cpp:1378-1435).  At human scale
// (6 M barcodes, 17 M (barcode, end) entries, 12 M contig pairs -- most seen by one barcode) those containers
// are minutes of single-threaded string hashing and tree walking, an order of magnitude more than the read
// stage in front of them.  Nothing in what the stages compute needs the strings: contig ids are replaced by
// their rank in std::string order (the order of the PairMap's keys, hence of every output line), barcodes
// by a number, and the maps by sorted vectors --
//   CompactIndex   per (barcode, contig): reads at the head and at the tail   = IndexMap after its post-pass
//   CompactPairs   per (contig a < contig b): barcodes per orientation        = PairMap, in its iteration order
// The output functions write byte for byte what graph.hpp's write for the same input (tests/test_host_graph.py
// runs both on the reference's demo data and on random IndexMaps).  -D (distance estimates) walks the IndexMap
// itself, in its container's order: that path keeps graph.hpp / dist_est.hpp.
#pragma once

#include "graph.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace arks_host {

// std::sort over `threads` threads: the keys are cut into ranges by sampled splitters, every range is gathered
// and sorted by a thread of its own, and the ranges are laid out one after the other -- the result is the sorted
// array (equal keys may sit in any order among themselves, as with std::sort).  Classification, counting and the
// scatter run over the threads too (every thread its stretch of the input and its own histogram: the scatter of
// stretch t into range b starts behind what the stretches in front of it put there).  Memory: a second copy of
// the array plus a byte per element while it runs.  A range that took more than half of the keys (one key
// dominating: the splitters cannot cut it) is not worth the copy: plain std::sort then.
template <typename T, typename Key>
inline void
parallel_sort_by(std::vector<T>& v, Key key, unsigned threads)
{
	const size_t n = v.size();
	auto less = [&](const T& a, const T& b) { return key(a) < key(b); };
	if (threads < 2 || n < (1u << 16)) {
		std::sort(v.begin(), v.end(), less);
		return;
	}
	const unsigned B = std::min<unsigned>(threads, 64);
	std::vector<uint64_t> sample;
	for (size_t i = 0; i < (size_t)B * 64; ++i)
		sample.push_back(key(v[(i * 2654435761ull) % n]));
	std::sort(sample.begin(), sample.end());
	std::vector<uint64_t> split; // bucket b holds keys in (split[b-1], split[b]]
	for (unsigned b = 1; b < B; ++b)
		split.push_back(sample[(size_t)b * 64]);
	auto bucket_of = [&](uint64_t k) { return (size_t)(std::lower_bound(split.begin(), split.end(), k) - split.begin()); };
	const unsigned T_ = std::min<unsigned>(threads, 64);
	auto stretch = [&](unsigned t) { return std::make_pair(n * t / T_, n * (t + 1) / T_); };
	std::vector<uint8_t> which(n);
	std::vector<std::vector<size_t>> hist(T_, std::vector<size_t>(B, 0));
	auto over_threads = [&](auto&& body) {
		std::vector<std::thread> th;
		for (unsigned t = 0; t < T_; ++t)
			th.emplace_back([&, t] { body(t); });
		for (auto& x : th)
			x.join();
	};
	over_threads([&](unsigned t) {
		const auto [lo, hi] = stretch(t);
		std::vector<size_t>& h = hist[t];
		for (size_t i = lo; i < hi; ++i) {
			which[i] = (uint8_t)bucket_of(key(v[i]));
			h[which[i]]++;
		}
	});
	std::vector<size_t> count(B + 1, 0);
	for (unsigned b = 0; b < B; ++b) {
		size_t c = 0;
		for (unsigned t = 0; t < T_; ++t)
			c += hist[t][b];
		count[b + 1] = count[b] + c;
	}
	for (unsigned b = 0; b < B; ++b)
		if (count[b + 1] - count[b] > n / 2) {
			std::sort(v.begin(), v.end(), less);
			return;
		}
	std::vector<T> out(n);
	// at[t][b]: where stretch t's elements of range b go
	std::vector<std::vector<size_t>> at(T_, std::vector<size_t>(B, 0));
	for (unsigned b = 0; b < B; ++b) {
		size_t pos = count[b];
		for (unsigned t = 0; t < T_; ++t) {
			at[t][b] = pos;
			pos += hist[t][b];
		}
	}
	over_threads([&](unsigned t) {
		const auto [lo, hi] = stretch(t);
		std::vector<size_t>& a = at[t];
		for (size_t i = lo; i < hi; ++i)
			out[a[which[i]]++] = v[i];
	});
	std::vector<std::thread> th;
	for (unsigned b = 0; b < B; ++b)
		th.emplace_back([&, b] {
			std::sort(out.begin() + (std::ptrdiff_t)count[b], out.begin() + (std::ptrdiff_t)count[b + 1], less);
		});
	for (auto& t : th)
		t.join();
	v.swap(out);
}

struct CompactEntry
{
	uint32_t barcode; // a number per barcode (any: only equality matters)
	uint32_t contig;  // rank of the contig id in std::string order
	int head, tail;   // read pairs at the two ends (0: the end the post-pass of chromiumRead would add)
};

struct CompactIndex
{
	std::vector<std::string> contig;   // rank -> id
	std::vector<CompactEntry> entries; // by (barcode, contig)
	std::vector<int> barcode_mult;     // barcode -> its multiplicity (0: not in the multiplicity map)
	size_t n_barcodes = 0;             // barcodes that have entries = IndexMap::size()
};

struct CompactPair
{
	uint32_t a, b; // contig ranks, a < b
	unsigned cnt[4]; // HH, HT, TH, TT
};
typedef std::vector<CompactPair> CompactPairs;

// One (barcode, conreci, count) entry of a read-stage result, the barcode already numbered over all ranks
struct RawEntry
{
	uint32_t barcode, conreci, count;
};

// contigRecord[conreci] = (contig id, is head) as getContigKmers fills it (Arcs.cpp:1079-1081); `mult_of`
// = multiplicity per barcode number.  Contigs are told apart by their ID, as the reference's maps do: two
// FASTA records of one name share their entries.
inline CompactIndex
build_compact_index(std::vector<RawEntry>& raw, const std::vector<CI>& contigRecord, std::vector<int> mult_of, unsigned threads = 1)
{
	CompactIndex ix;
	// rank of every contig id
	std::vector<std::string_view> ids;
	ids.reserve(contigRecord.size());
	for (size_t c = 1; c < contigRecord.size(); ++c)
		ids.push_back(contigRecord[c].first);
	std::sort(ids.begin(), ids.end());
	ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
	ix.contig.assign(ids.begin(), ids.end());
	std::vector<uint32_t> rank_of(contigRecord.size(), 0);
	for (size_t c = 1; c < contigRecord.size(); ++c)
		rank_of[c] = (uint32_t)(std::lower_bound(ids.begin(), ids.end(), std::string_view(contigRecord[c].first)) - ids.begin());
	// (barcode, contig) groups: sort, then add up heads and tails
	struct Rec
	{
		uint64_t key; // barcode << 32 | contig rank
		uint32_t count;
		uint32_t head;
	};
	std::vector<Rec> recs(raw.size());
	for (size_t i = 0; i < raw.size(); ++i)
		recs[i] = Rec{ ((uint64_t)raw[i].barcode << 32) | rank_of[raw[i].conreci], raw[i].count, contigRecord[raw[i].conreci].second ? 1u : 0u };
	std::vector<RawEntry>().swap(raw);
	parallel_sort_by(recs, [](const Rec& x) { return x.key; }, threads); // (records of one key are added up: their order does not matter)
	ix.entries.reserve(recs.size());
	for (size_t i = 0; i < recs.size();) {
		CompactEntry e{ (uint32_t)(recs[i].key >> 32), (uint32_t)recs[i].key, 0, 0 };
		size_t j = i;
		for (; j < recs.size() && recs[j].key == recs[i].key; ++j)
			(recs[j].head ? e.head : e.tail) += (int)recs[j].count;
		if (ix.entries.empty() || ix.entries.back().barcode != e.barcode)
			ix.n_barcodes++;
		ix.entries.push_back(e);
		i = j;
	}
	ix.barcode_mult = std::move(mult_of);
	return ix;
}

// pairContigs (Arcs.cpp:1378-1435) on a CompactIndex: per barcode within the multiplicity bounds, every pair of
// its contigs whose reads sit significantly at one end (head_or_tail) adds one to that orientation's count
inline CompactPairs
pair_contigs_compact(const CompactIndex& ix, const GraphParams& P, unsigned threads = 1)
{
	struct Hit
	{
		uint64_t key; // a << 32 | b
		uint32_t orientation;
	};
	std::vector<Hit> hits;
	struct Valid
	{
		uint32_t contig;
		bool head;
	};
	std::vector<Valid> valid;
	const std::vector<CompactEntry>& E = ix.entries;
	for (size_t i = 0; i < E.size();) {
		size_t j = i;
		while (j < E.size() && E[j].barcode == E[i].barcode)
			++j;
		const int m = E[i].barcode < ix.barcode_mult.size() ? ix.barcode_mult[E[i].barcode] : 0;
		if (m >= P.min_mult && m <= P.max_mult && j - i > 1) {
			valid.clear();
			for (size_t e = i; e < j; ++e) {
				const auto v = head_or_tail(E[e].head, E[e].tail, P);
				if (v.first)
					valid.push_back(Valid{ E[e].contig, v.second });
			}
			for (size_t x = 0; x < valid.size(); ++x)
				for (size_t y = x + 1; y < valid.size(); ++y) // entries are in contig order: x is `a`, y is `b`
					hits.push_back(Hit{ ((uint64_t)valid[x].contig << 32) | valid[y].contig,
					                    (valid[x].head ? 0u : 2u) + (valid[y].head ? 0u : 1u) });
		}
		i = j;
	}
	parallel_sort_by(hits, [](const Hit& x) { return x.key; }, threads); // (hits of one key are counted: any order)
	CompactPairs pairs;
	for (size_t i = 0; i < hits.size();) {
		CompactPair p{ (uint32_t)(hits[i].key >> 32), (uint32_t)hits[i].key, { 0, 0, 0, 0 } };
		size_t j = i;
		for (; j < hits.size() && hits[j].key == hits[i].key; ++j)
			p.cnt[hits[j].orientation]++;
		pairs.push_back(p);
		i = j;
	}
	return pairs;
}

// createGraph (Arcs.cpp:1475-1526)
inline void
create_graph_compact(const CompactPairs& pairs, const CompactIndex& ix, ScaffoldGraph& g, const GraphParams& P)
{
	std::vector<int> vertex(ix.contig.size(), -1);
	for (const CompactPair& p : pairs) {
		unsigned mx = 0, index = 0;
		for (unsigned i = 0; i < 4; ++i)
			if (p.cnt[i] > mx) {
				mx = p.cnt[i];
				index = i;
			}
		unsigned second = 0;
		for (unsigned i = 0; i < 4; ++i)
			if (p.cnt[i] != mx && p.cnt[i] > second)
				second = p.cnt[i];
		if (!check_significance((int)mx, (int)(mx + second), P))
			continue;
		for (const uint32_t c : { p.a, p.b })
			if (vertex[c] < 0) {
				vertex[c] = (int)g.id.size();
				g.id.push_back(ix.contig[c]);
				g.alive.push_back(true);
			}
		g.edges.push_back(Edge{ vertex[p.a], vertex[p.b], (int)index, (int)mx });
	}
}

// text is put together in a buffer of its own and written in large pieces (the TSV of a human-size run is a
// gigabyte: an ostream insertion per field is most of the time otherwise)
class TextOut
{
  public:
	explicit TextOut(std::ostream& f)
	  : f_(f)
	{
		buf_.reserve(kFlush + 4096);
	}
	~TextOut() { flush(); }
	void str(const std::string& s) { buf_.append(s); }
	void ch(char c) { buf_.push_back(c); }
	void num(unsigned long long v)
	{
		char tmp[24];
		int n = 0;
		do {
			tmp[n++] = (char)('0' + v % 10);
			v /= 10;
		} while (v);
		while (n)
			buf_.push_back(tmp[--n]);
	}
	void end_line()
	{
		buf_.push_back('\n');
		if (buf_.size() >= kFlush)
			flush();
	}
	void flush()
	{
		f_.write(buf_.data(), (std::streamsize)buf_.size());
		buf_.clear();
	}

  private:
	static constexpr size_t kFlush = 1u << 20;
	std::ostream& f_;
	std::string buf_;
};

// writeTSV (Arcs.cpp:1709-1757).  The per-end barcode counts run over EVERY barcode of the IndexMap, whatever
// its multiplicity, and over the ends the post-pass added (count 0: they pass only with -c 0).
inline void
write_tsv_compact(std::ostream& f, const CompactIndex& ix, const CompactPairs& pairs, size_t barcode_count, const GraphParams& P,
                  unsigned threads = 1, size_t block = (size_t)1 << 16 /* pairs per piece of text */)
{
	std::vector<unsigned> per_head(ix.contig.size(), 0), per_tail(ix.contig.size(), 0);
	for (const CompactEntry& e : ix.entries) {
		if (e.head >= P.min_reads)
			per_head[e.contig]++;
		if (e.tail >= P.min_reads)
			per_tail[e.contig]++;
	}
	// the lines of a range of pairs (the file of a human-size run is a gigabyte: the ranges are put into text by
	// the -t threads and written one after the other)
	auto lines = [&](size_t lo, size_t hi, std::string& text) {
		auto num = [&](unsigned long long v) {
			char tmp[24];
			int n = 0;
			do {
				tmp[n++] = (char)('0' + v % 10);
				v /= 10;
			} while (v);
			while (n)
				text.push_back(tmp[--n]);
		};
		for (size_t q = lo; q < hi; ++q) {
			const CompactPair& p = pairs[q];
			const std::string& u = ix.contig[p.a];
			const std::string& v = ix.contig[p.b];
			const unsigned mx = *std::max_element(p.cnt, p.cnt + 4);
			for (unsigned i = 0; i < 4; ++i) {
				if (p.cnt[i] == 0)
					continue;
				const bool usense = i < 2, vsense = i % 2;
				const char best = p.cnt[i] == mx ? 'T' : 'F';
				const unsigned ub = usense ? per_head[p.a] : per_tail[p.a], vb = !vsense ? per_head[p.b] : per_tail[p.b];
				text.append(u), text.push_back(usense ? '-' : '+'), text.push_back('\t'), text.append(v), text.push_back(vsense ? '-' : '+');
				text.push_back('\t'), text.push_back(best), text.push_back('\t');
				num(p.cnt[i]), text.push_back('\t'), num(ub), text.push_back('\t'), num(vb), text.push_back('\t'), num(barcode_count);
				text.push_back('\n');
				text.append(v), text.push_back(vsense ? '+' : '-'), text.push_back('\t'), text.append(u), text.push_back(usense ? '+' : '-');
				text.push_back('\t'), text.push_back(best), text.push_back('\t');
				num(p.cnt[i]), text.push_back('\t'), num(vb), text.push_back('\t'), num(ub), text.push_back('\t'), num(barcode_count);
				text.push_back('\n');
			}
		}
	};
	f << "U\tV\tBest_orientation\tShared_barcodes\tU_barcodes\tV_barcodes\tAll_barcodes\n";
	block = std::max<size_t>(block, 1);
	const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, (pairs.size() + block - 1) / block));
	for (size_t base = 0; base < pairs.size(); base += block * T) {
		std::vector<std::string> text(T);
		std::vector<std::thread> th;
		for (unsigned t = 0; t < T; ++t) {
			const size_t lo = std::min(pairs.size(), base + block * t), hi = std::min(pairs.size(), lo + block);
			if (t + 1 < T)
				th.emplace_back([&, t, lo, hi] { lines(lo, hi, text[t]); });
			else
				lines(lo, hi, text[t]);
		}
		for (auto& x : th)
			x.join();
		for (const std::string& x : text)
			f.write(x.data(), (std::streamsize)x.size());
	}
}

// the -P pair TSV (Arcs.cpp:1531-1544)
inline void
write_pair_map_compact(std::ostream& out, const CompactIndex& ix, const CompactPairs& pairs)
{
	TextOut o(out);
	for (const CompactPair& p : pairs) {
		o.str(ix.contig[p.a]), o.ch('\t'), o.str(ix.contig[p.b]);
		for (unsigned i = 0; i < 4; ++i)
			o.ch('\t'), o.num(p.cnt[i]);
		o.end_line();
	}
}

// countBarcodes (Arcs.cpp:815-830; also prints the summary line)
inline size_t
count_barcodes_compact(const CompactIndex& ix, const std::unordered_map<std::string, int>& mult, const GraphParams& P)
{
	size_t n = 0;
	for (const auto& x : mult)
		if (x.second >= P.min_mult && x.second <= P.max_mult)
			++n;
	std::cout << "{ \"All_barcodes_unfiltered\":" << mult.size() << ", \"All_barcodes_filtered\":" << n
	          << ", \"Scaffold_end_barcodes\":" << ix.n_barcodes << ", \"Min_barcode_reads_threshold\":" << P.min_mult
	          << ", \"Max_barcode_reads_threshold\":" << P.max_mult << " }\n";
	return n;
}

// an IndexMap as a CompactIndex (tests; graph_check)
inline CompactIndex
compact_from_imap(const IndexMap& imap, const std::unordered_map<std::string, int>& mult)
{
	std::vector<CI> record;
	record.push_back(CI("null contig", false));
	std::unordered_map<std::string, uint32_t> conreci_of; // id + 'H' / 'T'
	std::vector<RawEntry> raw;
	std::vector<int> mult_of;
	uint32_t b = 0;
	for (const auto& it : imap) {
		const auto m = mult.find(it.first);
		mult_of.push_back(m == mult.end() ? 0 : m->second);
		for (const auto& sc : it.second) {
			const std::string key = sc.first.first + (sc.first.second ? "\tH" : "\tT");
			auto f = conreci_of.find(key);
			if (f == conreci_of.end()) {
				f = conreci_of.emplace(key, (uint32_t)record.size()).first;
				record.push_back(sc.first);
			}
			raw.push_back(RawEntry{ b, f->second, (uint32_t)sc.second });
		}
		++b;
	}
	return build_compact_index(raw, record, std::move(mult_of));
}

} // namespace arks_host
