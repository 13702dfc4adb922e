// graph.hpp -- the stages of `arcs` after the read mapping, restated without Boost / ABySS headers
// and byte-for-byte compatible in their outputs:
//   IndexMap -> PairMap        pairContigs      Arcs/Arcs.cpp:1378-1435 (+ headOrTail :846-861,
//                                               normalEstimation :833-839)
//   PairMap  -> scaffold graph createGraph      :1475-1526 (+ checkSignificance :1459-1467)
//   degree filter              removeDegreeNodes :1569-1589
//   <base>_original.gv         writeGraph       :1549-1563 with the writers of Arcs/Arcs.h:185-229
//                                               (boost::write_graphviz layout)
//   <base>.dist.gv             createAbyssGraph :1615-1659 + write_dot Graph/DotIO.h:82-114
//   <base>_main.tsv            writeTSV         :1709-1757 ; countBarcodes :815-830
//   barcode counts / pair TSV  :1678-1700 / :1531-1544
// The container types are the reference's (Arcs/Arcs.h:105-115) wherever their iteration order
// reaches an output file.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <limits>
#include <vector>

namespace arks_host {

typedef std::pair<std::string, bool> CI; // (contig id, is head)
typedef std::map<CI, int> ScafMap;
typedef std::unordered_map<std::string, ScafMap> IndexMap;
typedef std::map<std::pair<std::string, std::string>, std::vector<unsigned>> PairMap;
typedef std::unordered_map<std::string, int> ContigToLength;

struct GraphParams
{
	int min_reads = 5;
	int min_links = 0;
	int min_mult = 50;
	int max_mult = 10000;
	int max_degree = 0;
	float error_percent = 0.05f;
	unsigned gap = 100;
	int end_length = 30000;      // -e, used by the distance estimates only
	bool dist_est = false;       // -D
	bool dist_upper = false;     // --dist_upper: the .dist.gv carries the upper bound instead of the median
	unsigned dist_bin_size = 20; // -B
};

// Arcs.cpp:833-839.  The mixed widths are part of the behaviour: mean and sd are floats, the
// division runs in double (std::sqrt(2) on an int argument is the double overload), the result
// narrows to float.
inline float
normal_estimation(int x, float p, int n)
{
	float mean = n * p;
	float sd = std::sqrt(n * p * (1 - p));
	return (float)(0.5 * (1 + std::erf((x - mean) / (sd * std::sqrt(2.0)))));
}

// Arcs.cpp:846-861
inline std::pair<bool, bool>
head_or_tail(int head, int tail, const GraphParams& P)
{
	const int mx = std::max(head, tail);
	const int sum = head + tail;
	if (sum < P.min_reads)
		return { false, false };
	const float cdf = normal_estimation(mx, 0.5f, sum);
	if (1 - cdf < P.error_percent)
		return { true, mx == head };
	return { false, false };
}

// Arcs.cpp:1378-1435
inline void
pair_contigs(IndexMap& imap, PairMap& pmap, std::unordered_map<std::string, int>& mult, const GraphParams& P)
{
	for (auto it = imap.begin(); it != imap.end(); ++it) {
		const int m = mult[it->first];
		if (m < P.min_mult || m > P.max_mult)
			continue;
		ScafMap& sm = it->second;
		for (auto o = sm.begin(); o != sm.end(); ++o) {
			for (auto p = sm.begin(); p != sm.end(); ++p) {
				const std::string& a = o->first.first;
				const std::string& b = p->first.first;
				if (!(a < b && o->first.second && p->first.second))
					continue;
				const auto va = head_or_tail(sm[CI(a, true)], sm[CI(a, false)], P);
				const auto vb = head_or_tail(sm[CI(b, true)], sm[CI(b, false)], P);
				if (!(va.first && vb.first))
					continue;
				auto& cnt = pmap[std::make_pair(a, b)];
				if (cnt.empty())
					cnt.resize(4);
				cnt[(va.second ? 0 : 2) + (vb.second ? 0 : 1)]++; // HH, HT, TH, TT
			}
		}
	}
}

struct Edge
{
	int u, v; // vertex slots
	int orientation;
	int weight;
	// distance estimate (-D, dist_est.hpp); unset as in EdgeProperties, Arcs/Arcs.h:166-183
	int min_dist = std::numeric_limits<int>::min();
	int dist = std::numeric_limits<int>::max();
	int max_dist = std::numeric_limits<int>::max();
	float jaccard = -1.0f;
};

// what boost::undirected_graph<VertexProperties, EdgeProperties> amounts to for this program:
// vertices and edges in insertion order, vertex indices handed out sequentially
struct ScaffoldGraph
{
	std::vector<std::string> id; // per vertex slot
	std::vector<bool> alive;
	std::list<Edge> edges;
};

// Arcs.cpp:1459-1467
inline bool
check_significance(int mx, int second, const GraphParams& P)
{
	if (mx < P.min_links)
		return false;
	const float cdf = normal_estimation(mx, 0.5f, second);
	return 1 - cdf < P.error_percent;
}

// Arcs.cpp:1475-1526
inline void
create_graph(const PairMap& pmap, ScaffoldGraph& g, const GraphParams& P)
{
	std::unordered_map<std::string, int> vmap;
	for (const auto& it : pmap) {
		const auto& count = it.second;
		unsigned mx = 0, index = 0;
		for (unsigned i = 0; i < count.size(); ++i)
			if (count[i] > mx) {
				mx = count[i];
				index = i;
			}
		unsigned second = 0;
		for (unsigned i = 0; i < count.size(); ++i)
			if (count[i] != mx && count[i] > second)
				second = count[i];
		if (!check_significance((int)mx, (int)(mx + second), P))
			continue;
		for (const std::string* s : { &it.first.first, &it.first.second })
			if (vmap.count(*s) == 0) {
				vmap[*s] = (int)g.id.size();
				g.id.push_back(*s);
				g.alive.push_back(true);
			}
		g.edges.push_back(Edge{ vmap[it.first.first], vmap[it.first.second], (int)index, (int)mx });
	}
}

// Arcs.cpp:1569-1589: every vertex whose degree exceeds max_degree in the ORIGINAL graph is
// collected first, then cleared and removed one by one
inline void
remove_degree_nodes(ScaffoldGraph& g, int max_degree)
{
	std::vector<int> deg(g.id.size(), 0);
	for (const Edge& e : g.edges) {
		deg[e.u]++;
		deg[e.v]++;
	}
	for (size_t v = 0; v < g.id.size(); ++v)
		if (deg[v] > max_degree) {
			g.alive[v] = false;
			g.edges.remove_if([&](const Edge& e) { return e.u == (int)v || e.v == (int)v; });
		}
}

// boost::write_graphviz with the property writers of Arcs/Arcs.h:185-229
inline void
write_graph(std::ostream& out, const ScaffoldGraph& g)
{
	std::vector<int> index(g.id.size(), -1);
	int n = 0;
	for (size_t v = 0; v < g.id.size(); ++v)
		if (g.alive[v])
			index[v] = n++;
	out << "graph G {\n";
	for (size_t v = 0; v < g.id.size(); ++v)
		if (g.alive[v])
			out << index[v] << " [id=" << g.id[v] << "];\n";
	for (const Edge& e : g.edges)
	{
		out << index[e.u] << "--" << index[e.v] << " [label=" << e.orientation << ", weight=" << e.weight;
		if (e.min_dist != std::numeric_limits<int>::min()) // Arcs.h:204-211
			out << ", d=" << e.dist << ", maxd=" << e.max_dist;
		out << "];\n";
	}
	out << "}\n";
}

// createAbyssGraph + write_dot: two vertices per contig ("id+", "id-") in the iteration order of the
// ContigToLength unordered_map, every edge with its reverse-complement twin, out-edges listed per
// vertex in insertion order.  Returns false on a duplicate edge (the reference exits).
inline bool
write_dist_graph(std::ostream& out, const ContigToLength& lengths, const ScaffoldGraph& g, unsigned gap,
                 std::string* err, bool dist_est = false, bool dist_upper = false)
{
	std::vector<std::string> names;
	std::vector<int> len;
	std::unordered_map<std::string, int> dict;
	for (const auto& it : lengths) {
		dict[it.first] = (int)names.size();
		names.push_back(it.first);
		len.push_back(it.second);
	}
	struct Out { int v; int weight; int d; };
	std::vector<std::vector<Out>> adj(2 * names.size());
	auto vname = [&](int v) { return names[(size_t)(v >> 1)] + ((v & 1) ? "-" : "+"); };
	auto add = [&](int u, int v, int w, int d) {
		for (const Out& o : adj[(size_t)u])
			if (o.v == v)
				return false;
		adj[(size_t)u].push_back(Out{ v, w, d });
		return true;
	};
	for (const Edge& e : g.edges) {
		// sense = true is the '-' vertex: u sense = orientation < 2, v sense = orientation % 2
		const int u = 2 * dict.at(g.id[(size_t)e.u]) + (e.orientation < 2 ? 1 : 0);
		const int v = 2 * dict.at(g.id[(size_t)e.v]) + (e.orientation % 2);
		// Arcs.cpp:1636-1648: with -D the estimate replaces the gap (an edge without one keeps INT_MAX)
		const int d = dist_est ? (dist_upper ? e.max_dist : e.dist) : (int)gap;
		if (!add(u, v, e.weight, d)) {
			if (err)
				*err = "error: Duplicate edge: \"" + vname(u) + "\" -> \"" + vname(v) + "\"";
			return false;
		}
		if (u != (v ^ 1))
			add(v ^ 1, u ^ 1, e.weight, d);
	}
	out << "digraph arcs {\n";
	for (size_t v = 0; v < adj.size(); ++v)
		out << '"' << vname((int)v) << "\" [l=" << len[v >> 1] << "]\n";
	for (size_t u = 0; u < adj.size(); ++u)
		for (const Out& o : adj[u])
			out << '"' << vname((int)u) << "\" -> \"" << vname(o.v) << "\" [d=" << o.d
			    << " e=" << std::fixed << std::setprecision(1) << (float)gap << " n=" << o.weight << "]\n";
	out << "}\n";
	return true;
}

struct HashScaffoldEnd
{
	size_t operator()(const CI& key) const { return std::hash<std::string>()(key.first) ^ key.second; }
};

// Arcs.cpp:815-830 (also prints the JSON-ish summary line)
inline size_t
count_barcodes(const IndexMap& imap, const std::unordered_map<std::string, int>& mult, const GraphParams& P)
{
	size_t n = 0;
	for (const auto& x : mult)
		if (x.second >= P.min_mult && x.second <= P.max_mult)
			++n;
	std::cout << "{ \"All_barcodes_unfiltered\":" << mult.size() << ", \"All_barcodes_filtered\":" << n
	          << ", \"Scaffold_end_barcodes\":" << imap.size()
	          << ", \"Min_barcode_reads_threshold\":" << P.min_mult
	          << ", \"Max_barcode_reads_threshold\":" << P.max_mult << " }\n";
	return n;
}

// Arcs.cpp:1709-1757
inline void
write_tsv(std::ostream& f, const IndexMap& imap, const PairMap& pmap, size_t barcode_count, const GraphParams& P)
{
	std::unordered_map<CI, unsigned, HashScaffoldEnd> per_end;
	for (const auto& it : imap)
		for (const auto& sc : it.second)
			if (sc.second >= P.min_reads)
				++per_end[sc.first];
	f << "U\tV\tBest_orientation\tShared_barcodes\tU_barcodes\tV_barcodes\tAll_barcodes\n";
	for (const auto& it : pmap) {
		const std::string& u = it.first.first;
		const std::string& v = it.first.second;
		const auto& counts = it.second;
		const unsigned mx = *std::max_element(counts.begin(), counts.end());
		for (unsigned i = 0; i < counts.size(); ++i) {
			if (counts[i] == 0)
				continue;
			const bool usense = i < 2, vsense = i % 2;
			const char* best = counts[i] == mx ? "T" : "F";
			const unsigned ub = per_end[std::make_pair(u, usense)], vb = per_end[std::make_pair(v, !vsense)];
			f << u << (usense ? '-' : '+') << '\t' << v << (vsense ? '-' : '+') << '\t' << best << '\t'
			  << counts[i] << '\t' << ub << '\t' << vb << '\t' << barcode_count << '\n';
			f << v << (vsense ? '+' : '-') << '\t' << u << (usense ? '+' : '-') << '\t' << best << '\t'
			  << counts[i] << '\t' << vb << '\t' << ub << '\t' << barcode_count << '\n';
		}
	}
}

// Arcs.cpp:1678-1700
inline void
write_barcode_counts(std::ostream& f, const std::unordered_map<std::string, int>& mult)
{
	// (the reference copies the map into a vector of pairs and sorts that; pointers to the keys sort the same
	// way without copying millions of strings, and the lines go out through one buffer)
	typedef std::pair<const std::string*, unsigned> Item;
	std::vector<Item> sorted;
	sorted.reserve(mult.size());
	for (const auto& kv : mult)
		sorted.emplace_back(&kv.first, (unsigned)kv.second);
	std::sort(sorted.begin(), sorted.end(), [](const Item& a, const Item& b) {
		return a.second != b.second ? a.second > b.second : *a.first < *b.first;
	});
	std::string buf;
	buf.reserve((1u << 20) + 256);
	for (const Item& x : sorted) {
		buf.append(*x.first);
		buf.push_back('\t');
		buf.append(std::to_string(x.second));
		buf.push_back('\n');
		if (buf.size() >= (1u << 20)) {
			f.write(buf.data(), (std::streamsize)buf.size());
			buf.clear();
		}
	}
	f.write(buf.data(), (std::streamsize)buf.size());
}

// Arcs.cpp:1531-1544
inline void
write_pair_map(std::ostream& out, const PairMap& pmap)
{
	for (const auto& it : pmap)
		out << it.first.first << "\t" << it.first.second << "\t" << it.second[0] << "\t" << it.second[1]
		    << "\t" << it.second[2] << "\t" << it.second[3] << std::endl;
}

// the post-pass of chromiumRead, Arcs.cpp:1304-1319: a barcode that hits only one end of a contig
// gets the other end with count 0
inline void
add_opposite_ends(IndexMap& imap)
{
	for (auto& it : imap) {
		const ScafMap copy = it.second;
		for (const auto& j : copy) {
			const CI other(j.first.first, !j.first.second);
			if (it.second.count(other) == 0)
				it.second[other] = 0;
		}
	}
}


// createIndexMultMap, Arcs.cpp:392-448: barcode -> reads from a `-u` file; the name decides the format (".tsv"
// anywhere in it: white-space separated, otherwise "barcode,reads").  Returns the number of lines read (what
// the reference reports as "distinct barcodes"); exits like the reference when the file cannot be opened.
inline size_t
read_multiplicity_file(const std::string& multfile, std::unordered_map<std::string, int>& mult)
{
	size_t numbarcodes = 0;
	const bool tsv = multfile.find(".tsv") != std::string::npos;
	std::ifstream in(multfile.c_str(), std::ios::binary);
	if (!in) {
		std::cerr << "Could not open " << multfile << ". --fatal.\n";
		exit(EXIT_FAILURE);
	}
	// the whole file at once, line by line out of memory: a stringstream per line is a microsecond, and a
	// linked-read run has millions of barcodes.  A line of the usual shape -- barcode, separator, a short run
	// of digits -- is taken apart here; any other line goes through the reference's own calls (below), so that
	// what they make of it (including the exception std::stoi throws at a line without a number) is unchanged.
	std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	mult.reserve(mult.size() + text.size() / 24);
	auto space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
	size_t at = 0;
	while (at < text.size()) {
		const char* nl = (const char*)std::memchr(text.data() + at, '\n', text.size() - at);
		const size_t begin = at, end = nl ? (size_t)(nl - text.data()) : text.size();
		const char* p = text.data() + begin;
		const char* const e = text.data() + end;
		at = end + 1;
		numbarcodes++;
		const char *b0, *b1; // barcode
		bool plain = true;
		if (tsv) {
			while (p < e && space(*p))
				++p;
			b0 = p;
			while (p < e && !space(*p))
				++p;
			b1 = p;
		} else {
			b0 = p;
			const char* comma = (const char*)std::memchr(p, ',', (size_t)(e - p));
			plain = comma != nullptr;
			b1 = p = comma ? comma : e;
			if (comma)
				++p;
		}
		while (p < e && space(*p))
			++p;
		const char* d0 = p;
		while (p < e && *p >= '0' && *p <= '9')
			++p;
		plain = plain && p > d0 && p - d0 <= 9 && (p == e || space(*p)) && b1 > b0;
		if (plain) {
			int m = 0;
			for (const char* q = d0; q < p; ++q)
				m = m * 10 + (*q - '0');
			mult[std::string(b0, b1)] = m;
			continue;
		}
		// (Arcs.cpp:404-427, literally)
		const std::string line(text, begin, end - begin);
		std::string barcode, ms;
		if (tsv) {
			std::stringstream sst(line);
			sst >> barcode >> ms;
		} else {
			std::istringstream iss(line);
			getline(iss, barcode, ',');
			iss >> ms;
		}
		const size_t m = (size_t)std::stoi(ms);
		if (!barcode.empty())
			mult[barcode] = (int)m;
		else
			std::cout << "Please check your multiplicity file." << std::endl;
	}
	return numbarcodes;
}

} // namespace arks_host
