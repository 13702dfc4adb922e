// dist_est.hpp -- `arcs -D`: gap size estimates for the edges of the scaffold graph from the barcodes that
// two contig ends share (replaces Arcs/DistanceEst.h and calcDistanceEstimates, Arcs/Arcs.cpp:1767-1808).
//
// The idea of the reference: inside ONE long contig the distance between its head and tail regions is
// known (length - 2 * end_length) and so is the Jaccard index of the barcode sets of the two regions; these
// (jaccard, distance) samples are the training data.  For an edge between two contigs the Jaccard index of
// the barcode sets of the joined ends is computed the same way, the `dist_bin_size` samples with the
// closest Jaccard index are looked up and their 1st percentile / median / 99th percentile distance become
// the edge's min / d / max.
//
// One point where this build is deliberately NOT the reference: samples with the same Jaccard index (very
// common: 0) collide in the reference's std::map<double, DistSample> and the one that was inserted first
// survives -- in the iteration order of an unordered_map of contig ids filled in the iteration order of the
// unordered IndexMap, i.e. an order that depends on the hash function and, with -t > 1, on thread timing.
// Here the sample of the smallest contig id (string order) survives and the samples file is written in
// that order too, so the output is a function of the input.
#pragma once

#include "graph.hpp"

#include <array>
#include <cmath>
#include <limits>

namespace arks_host {

// head-to-tail sample of one contig (DistanceEst.h:37-54)
struct DistSample
{
	unsigned distance = std::numeric_limits<unsigned>::max();
	unsigned barcodes_head = 0, barcodes_tail = 0, barcodes_union = 0, barcodes_intersect = 0;
};
typedef std::map<std::string, DistSample> DistSampleMap; // ordered: see the note above
typedef std::map<double, DistSample> JaccardToDist;

// shared-barcode counts of a candidate pair of contig ends (DistanceEst.h:64-80)
struct BarcodeStats
{
	unsigned barcodes1 = 0, barcodes2 = 0, barcodes_union = 0, barcodes_intersect = 0;
};
typedef std::array<BarcodeStats, 4> BarcodeStatsArray; // HH, HT, TH, TT
typedef std::map<std::pair<std::string, std::string>, BarcodeStatsArray> PairToBarcodeStats;

// an end counts for a barcode when it has at least -c read pairs and its contig is at least two end
// lengths long (DistanceEst.h:196-217)
inline bool
valid_barcode_mapping(unsigned contig_length, int pairs, const GraphParams& P)
{
	return pairs >= P.min_reads && contig_length >= unsigned(2 * P.end_length);
}

// DistanceEst.h:101-173
inline void
calc_dist_samples(
    const IndexMap& imap, const ContigToLength& lengths, const std::unordered_map<std::string, int>& mult,
    const GraphParams& P, DistSampleMap& samples)
{
	for (const auto& bc : imap) {
		const int m = mult.at(bc.first);
		if (m < P.min_mult || m > P.max_mult)
			continue;
		const ScafMap& ends = bc.second;
		for (const auto& end : ends) {
			const std::string& id = end.first.first;
			const bool is_head = end.first.second;
			if (end.second < P.min_reads)
				continue;
			const unsigned l = (unsigned)lengths.at(id);
			if (l < (unsigned)2 * P.end_length)
				continue;
			DistSample& s = samples[id];
			s.distance = l - 2 * P.end_length;
			(is_head ? s.barcodes_head : s.barcodes_tail)++;
			// does the barcode reach the other end of the contig as well?  (counted once, at the head)
			const auto other = ends.find(CI(id, !is_head));
			const bool found_other = other != ends.end() && other->second >= P.min_reads;
			if (found_other && is_head) {
				s.barcodes_intersect++;
				s.barcodes_union++;
			} else if (!found_other)
				s.barcodes_union++;
		}
	}
}

// DistanceEst.h:181-189; on equal keys the first insertion stays (std::map::insert)
inline void
build_jaccard_to_dist(const DistSampleMap& samples, JaccardToDist& out)
{
	for (const auto& it : samples)
		out.insert(JaccardToDist::value_type(
		    double(it.second.barcodes_intersect) / it.second.barcodes_union, it.second));
}

// DistanceEst.h:220-334
inline void
build_pair_to_barcode_stats(
    const IndexMap& imap, const std::unordered_map<std::string, int>& mult, const ContigToLength& lengths,
    const GraphParams& P, PairToBarcodeStats& out)
{
	std::map<CI, size_t> barcodes_of_end;
	for (const auto& bc : imap) {
		const int m = mult.at(bc.first);
		if (m < P.min_mult || m > P.max_mult)
			continue;
		const ScafMap& ends = bc.second;
		for (const auto& e1 : ends) {
			if (!valid_barcode_mapping((unsigned)lengths.at(e1.first.first), e1.second, P))
				continue;
			barcodes_of_end[e1.first]++;
			for (const auto& e2 : ends) {
				if (!valid_barcode_mapping((unsigned)lengths.at(e2.first.first), e2.second, P))
					continue;
				if (e1.first.first > e2.first.first)
					continue; // each unordered pair once (a contig with itself included)
				BarcodeStatsArray& st = out[std::make_pair(e1.first.first, e2.first.first)];
				st[(e1.first.second ? 0 : 2) + (e2.first.second ? 0 : 1)].barcodes_intersect++;
			}
		}
	}
	for (auto& it : out)
		for (int o = 0; o < 4; ++o) {
			BarcodeStats& st = it.second[(size_t)o];
			const auto c1 = barcodes_of_end.find(CI(it.first.first, o < 2));
			if (c1 == barcodes_of_end.end())
				continue;
			st.barcodes1 = (unsigned)c1->second;
			const auto c2 = barcodes_of_end.find(CI(it.first.second, o % 2 == 0));
			if (c2 == barcodes_of_end.end())
				continue;
			st.barcodes2 = (unsigned)c2->second;
			st.barcodes_union = st.barcodes1 + st.barcodes2 - st.barcodes_intersect;
		}
}

// Common/MapUtil.h:8-44: the element whose key is nearest; the lower one on equal distance
inline JaccardToDist::const_iterator
closest_key(const JaccardToDist& m, double key)
{
	if (m.empty())
		return m.end();
	auto it = m.lower_bound(key);
	if (it == m.begin())
		return it;
	if (it == m.end())
		return --it;
	auto prev = it;
	--prev;
	return std::fabs(key - prev->first) > std::fabs(key - it->first) ? it : prev;
}

// Common/MapUtil.h:47-93: grows the range around the nearest key to n elements, one neighbour at a time,
// taking the nearer of the two candidates (the upper one on equal distance)
inline std::pair<JaccardToDist::const_iterator, JaccardToDist::const_iterator>
closest_keys(const JaccardToDist& m, double key, size_t n)
{
	if (m.empty())
		return { m.end(), m.end() };
	auto first = closest_key(m, key);
	auto last = first;
	++last;
	for (size_t count = 1; count < n; ++count) {
		if (first == m.begin() && last == m.end())
			break;
		if (first == m.begin())
			++last;
		else if (last == m.end())
			--first;
		else {
			auto prev = first;
			--prev;
			if (std::fabs(key - prev->first) < std::fabs(key - last->first))
				first = prev;
			else
				++last;
		}
	}
	return { first, last };
}

// Common/StatUtil.h:8-32.  The weights are the reference's: the element BELOW the quantile position gets
// the fractional part, the one above its complement (the mirror image of the usual interpolation); the
// elements pass through size_t.
inline double
quantile(const std::vector<unsigned>& sorted, double q)
{
	const size_t last = sorted.size() - 1;
	const size_t before_pos = (size_t)std::floor(q * last), after_pos = (size_t)std::ceil(q * last);
	const size_t before = sorted[before_pos], after = sorted[after_pos];
	const double weight = (q * last - before_pos) / 1.0;
	return weight * before + (1.0 - weight) * after;
}

struct DistanceEstimate
{
	int min_dist = 0, dist = 0, max_dist = 0;
	double jaccard = 0.0;
};

// DistanceEst.h:337-389
inline bool
estimate_distance(const BarcodeStats& st, const JaccardToDist& j2d, const GraphParams& P, DistanceEstimate& out)
{
	if (j2d.empty() || st.barcodes_union == 0)
		return false;
	out.jaccard = double(st.barcodes_intersect) / st.barcodes_union;
	const auto range = closest_keys(j2d, out.jaccard, P.dist_bin_size);
	std::vector<unsigned> d;
	for (auto it = range.first; it != range.second; ++it)
		d.push_back(it->second.distance);
	std::sort(d.begin(), d.end());
	out.min_dist = (int)std::floor(quantile(d, 0.01));
	out.dist = (int)std::round(quantile(d, 0.5));
	out.max_dist = (int)std::ceil(quantile(d, 0.99));
	return true;
}

// DistanceEst.h:392-430: the pair is looked up as (source id, target id), as the edge was created
inline void
add_edge_distances(const PairToBarcodeStats& stats, const JaccardToDist& j2d, const GraphParams& P, ScaffoldGraph& g)
{
	if (j2d.empty())
		return;
	for (Edge& e : g.edges) {
		const auto it = stats.find(std::make_pair(g.id[(size_t)e.u], g.id[(size_t)e.v]));
		if (it == stats.end())
			continue;
		DistanceEstimate est;
		if (!estimate_distance(it->second[(size_t)e.orientation], j2d, P, est))
			continue;
		e.min_dist = est.min_dist;
		e.dist = est.dist;
		e.max_dist = est.max_dist;
		e.jaccard = (float)est.jaccard;
	}
}

// --dist_tsv, DistanceEst.h:433-494: two lines per edge (both reading directions)
inline void
write_dist_tsv(std::ostream& out, const PairToBarcodeStats& stats, const ScaffoldGraph& g)
{
	out << "contig1\tcontig2\tmin_dist\tdist\tmax_dist\tbarcodes1\tbarcodes2\tbarcodes_union\tbarcodes_intersect\n";
	for (const Edge& e : g.edges) {
		const std::string& id1 = g.id[(size_t)e.u];
		const std::string& id2 = g.id[(size_t)e.v];
		const auto it = stats.find(std::make_pair(id1, id2));
		if (it == stats.end())
			continue;
		const BarcodeStats& st = it->second[(size_t)e.orientation];
		const bool sense1 = e.orientation < 2, sense2 = e.orientation % 2;
		auto dists = [&]() {
			if (e.jaccard >= 0)
				out << e.min_dist << '\t' << e.dist << '\t' << e.max_dist << '\t';
			else
				out << "NA\tNA\tNA\t";
		};
		out << id1 << (sense1 ? '-' : '+') << '\t' << id2 << (sense2 ? '-' : '+') << '\t';
		dists();
		out << st.barcodes1 << '\t' << st.barcodes2 << '\t' << st.barcodes_union << '\t' << st.barcodes_intersect << '\n';
		out << id2 << (sense2 ? '+' : '-') << '\t' << id1 << (sense1 ? '+' : '-') << '\t';
		dists();
		out << st.barcodes2 << '\t' << st.barcodes1 << '\t' << st.barcodes_union << '\t' << st.barcodes_intersect << '\n';
	}
}

// --samples_tsv, DistanceEst.h:501-520
inline void
write_dist_samples_tsv(std::ostream& out, const DistSampleMap& samples)
{
	out << "contig_id\tdistance\tbarcodes_head\tbarcodes_tail\tbarcodes_union\tbarcodes_intersect\n";
	for (const auto& it : samples)
		out << it.first << '\t' << it.second.distance << '\t' << it.second.barcodes_head << '\t'
		    << it.second.barcodes_tail << '\t' << it.second.barcodes_union << '\t' << it.second.barcodes_intersect
		    << '\n';
}

} // namespace arks_host
