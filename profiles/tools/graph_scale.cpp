// graph_scale.cpp -- how long the host stages behind the read stage take at human scale (no GPU): a synthetic
// result of the read stage shaped like BASELINE configs[2] (6.25 M barcodes, ~17 M (barcode, end, count)
// entries over 60 296 contig ends) through merge_results (IndexMap), pair_contigs, create_graph and the writers.
//   g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Iarcs_amd/host profiles/tools/graph_scale.cpp -o graph_scale
//   ./graph_scale [barcodes in millions = 6.25] [literal]      literal = graph.hpp's containers (the reference's), else graph_fast.hpp
#include "graph.hpp"
#include "rank_merge.hpp"

#include <chrono>
#include <random>

using namespace arks_host;

int
main(int argc, char** argv)
{
	const size_t n_bc = (size_t)((argc > 1 ? std::atof(argv[1]) : 6.25) * 1e6), n_contigs = 30148;
	auto t0 = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		const auto t = std::chrono::steady_clock::now();
		std::printf("%-46s %8.2f s\n", what, std::chrono::duration<double>(t - t0).count());
		t0 = t;
	};
	std::mt19937_64 rng(7);
	std::vector<CI> contigRecord;
	contigRecord.push_back(CI("null contig", false));
	ContigToLength lengths;
	for (size_t c = 0; c < n_contigs; ++c) {
		const std::string id = std::to_string(c + 1);
		contigRecord.push_back(CI(id, true));
		contigRecord.push_back(CI(id, false));
		lengths[id] = 100000;
	}
	std::vector<RankResult> ranks(1);
	RankResult& r = ranks[0];
	r.files.resize(1);
	r.files[0].have = true;
	r.files[0].stored.push_back(0);
	r.files[0].st.resize(1);
	r.triples.resize(1);
	r.first.resize(1);
	std::unordered_map<std::string, int> mult;
	mult.reserve(n_bc);
	char name[32];
	uint64_t seq = 0;
	for (size_t b = 0; b < n_bc; ++b) {
		std::snprintf(name, sizeof name, "BC%08zu-1", b);
		r.names.push_back(name);
		mult[name] = 160;
		// two molecules of 50 kbp: each touches one end of a contig, now and then both ends or a neighbour's
		for (int m = 0; m < 2; ++m) {
			const uint32_t c = (uint32_t)(rng() % n_contigs);
			const uint32_t e = 2 * c + 1 + (uint32_t)(rng() & 1);
			r.triples[0].insert(r.triples[0].end(), { (uint32_t)b, e, 5 + (uint32_t)(rng() % 20) });
			r.first[0].push_back(seq++);
			if (rng() % 8 < 3) { // the molecule spans a join: the facing end of the next contig as well
				const uint32_t e2 = 2 * ((c + 1) % n_contigs) + 1 + (uint32_t)(rng() & 1);
				r.triples[0].insert(r.triples[0].end(), { (uint32_t)b, e2, 5 + (uint32_t)(rng() % 20) });
				r.first[0].push_back(seq++);
			}
		}
	}
	std::printf("%zu barcodes, %zu entries\n", n_bc, r.first[0].size());
	lap("synthetic read-stage result");
	std::vector<IndexMap> imaps;
	std::string out, err, pre_out, pre_err;
	GraphParams P;
	const bool literal = argc > 2 && std::string(argv[2]) == "literal";
	const unsigned T = argc > 3 ? (unsigned)std::atoi(argv[3]) : 1; // threads of the sorts (graph_fast.hpp)
	if (!literal) { // graph_fast.hpp
		std::vector<CompactIndex> cix;
		merge_results({ "reads.fq" }, ranks, imaps, mult, contigRecord, false, MergeParams{ false, { 60 }, 1, T }, out, err, &pre_out, &pre_err, &cix);
		lap("merge_results (CompactIndex)");
		const CompactPairs pairs = pair_contigs_compact(cix[0], P, T);
		lap("pair_contigs_compact");
		ScaffoldGraph g;
		create_graph_compact(pairs, cix[0], g, P);
		lap("create_graph_compact");
		{
			std::ofstream f("/tmp/gs/fast_original.gv");
			write_graph(f, g);
		}
		{
			std::ofstream f("/tmp/gs/fast.dist.gv");
			std::string e;
			write_dist_graph(f, lengths, g, P.gap, &e);
		}
		lap("write_graph + write_dist_graph");
		{
			const size_t nb = count_barcodes_compact(cix[0], mult, P);
			std::ofstream f("/tmp/gs/fast_main.tsv");
			write_tsv_compact(f, cix[0], pairs, nb, P, T);
		}
		lap("count_barcodes + write_tsv_compact");
		{
			std::ofstream f("/tmp/gs/fast_pair.tsv");
			write_pair_map_compact(f, cix[0], pairs);
		}
		lap("write_pair_map_compact");
		std::printf("%zu pairs, %zu edges\n", pairs.size(), g.edges.size());
		return 0;
	}
	merge_results({ "reads.fq" }, ranks, imaps, mult, contigRecord, false, MergeParams{ false, { 60 }, 1 }, out, err, &pre_out, &pre_err);
	lap("merge_results (IndexMap rebuilt on the host)");
	PairMap pmap;
	pair_contigs(imaps[0], pmap, mult, P);
	lap("pair_contigs");
	ScaffoldGraph g;
	create_graph(pmap, g, P);
	lap("create_graph");
	{
		std::ofstream f("/tmp/gs/out_original.gv");
		write_graph(f, g);
	}
	{
		std::ofstream f("/tmp/gs/out.dist.gv");
		std::string e;
		write_dist_graph(f, lengths, g, P.gap, &e);
	}
	lap("write_graph + write_dist_graph");
	{
		const size_t nb = count_barcodes(imaps[0], mult, P);
		std::ofstream f("/tmp/gs/out_main.tsv");
		write_tsv(f, imaps[0], pmap, nb, P);
	}
	lap("count_barcodes + write_tsv");
	{
		std::ofstream f("/tmp/gs/out_pair.tsv");
		write_pair_map(f, pmap);
	}
	lap("write_pair_map");
	std::printf("%zu pairs in the PairMap, %zu edges\n", pmap.size(), g.edges.size());
	return 0;
}
