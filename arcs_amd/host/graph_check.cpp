// graph_check.cpp -- test driver for graph.hpp (no GPU): runs the graph stage on an IndexMap given as
// text and writes the reference's output files.  Built and used by tests/test_host_graph.py only.
//   graph_check imap  <imap.tsv> <mult.tsv> <lengths.tsv> <out_base> c l m_lo m_hi d r gap x [e B upper]
//       imap.tsv: barcode \t contig \t H|T \t count   (the post-pass for missing ends is applied)
//       with e (end length), B (bin size) and upper (0|1): -D distance estimates as well
//       (<out_base>_dist.tsv, <out_base>_samples.tsv, d= / maxd= in the graph files)
//   graph_check imapfast ...   the same through graph_fast.hpp (numbers instead of strings; no distance estimates)
//   graph_check mult  <multiplicity file> <out.tsv>
//       the -u parser (createIndexMultMap) and the --barcode-counts writer: lines read, barcodes kept, sum of
//       the reads on stdout; the map as the reference writes it (Arcs.cpp:1684-1706) to <out.tsv>
//   graph_check gv    <original.gv> <lengths.tsv> <out.dist.gv> gap
//       rebuilds the scaffold graph from an _original.gv and writes the ABySS dist.gv for it
#include "dist_est.hpp"
#include "graph_fast.hpp"

#include <cstdio>
#include <cstring>

using namespace arks_host;

static void
read_lengths(const char* path, ContigToLength& len)
{
	std::ifstream in(path);
	std::string id;
	int l;
	while (in >> id >> l)
		len[id] = l; // insertion order = FASTA order, as in getContigKmers
}

int
main(int argc, char** argv)
{
	if (argc >= 4 && std::strcmp(argv[1], "mult") == 0) {
		std::unordered_map<std::string, int> mult;
		const size_t lines = read_multiplicity_file(argv[2], mult);
		long long total = 0;
		for (const auto& kv : mult)
			total += kv.second;
		std::printf("%zu %zu %lld\n", lines, mult.size(), total);
		std::ofstream out(argv[3]);
		write_barcode_counts(out, mult);
		return 0;
	}
	if (argc >= 6 && std::strcmp(argv[1], "gv") == 0) {
		ScaffoldGraph g;
		std::ifstream in(argv[2]);
		std::string line;
		while (std::getline(in, line)) {
			int a, b, lab, w;
			char idbuf[256];
			if (std::sscanf(line.c_str(), "%d--%d [label=%d, weight=%d];", &a, &b, &lab, &w) == 4)
				g.edges.push_back(Edge{ a, b, lab, w });
			else if (std::sscanf(line.c_str(), "%d [id=%255[^]]];", &a, idbuf) == 2) {
				g.id.push_back(idbuf);
				g.alive.push_back(true);
			}
		}
		ContigToLength len;
		read_lengths(argv[3], len);
		std::ofstream out(argv[4]);
		std::string err;
		if (!write_dist_graph(out, len, g, (unsigned)std::atoi(argv[5]), &err)) {
			std::cerr << err << "\n";
			return 1;
		}
		return 0;
	}
	if (argc >= 14 && (std::strcmp(argv[1], "imap") == 0 || std::strcmp(argv[1], "imapfast") == 0)) {
		const bool fast = std::strcmp(argv[1], "imapfast") == 0; // graph_fast.hpp instead of graph.hpp (no -D there)
		IndexMap imap;
		std::unordered_map<std::string, int> mult;
		ContigToLength len;
		{
			std::ifstream in(argv[2]);
			std::string bc, ctg, ht;
			int cnt;
			while (in >> bc >> ctg >> ht >> cnt)
				imap[bc][CI(ctg, ht == "H")] += cnt;
		}
		{
			std::ifstream in(argv[3]);
			std::string bc;
			int m;
			while (in >> bc >> m)
				mult[bc] = m;
		}
		read_lengths(argv[4], len);
		const std::string base = argv[5];
		GraphParams P;
		P.min_reads = std::atoi(argv[6]);
		P.min_links = std::atoi(argv[7]);
		P.min_mult = std::atoi(argv[8]);
		P.max_mult = std::atoi(argv[9]);
		P.max_degree = std::atoi(argv[10]);
		P.error_percent = (float)std::atof(argv[11]);
		P.gap = (unsigned)std::atoi(argv[12]);
		if (fast) {
			if (argc >= 17)
				return 2;
			const CompactIndex ix = compact_from_imap(imap, mult); // (the post-pass for missing ends is implied)
			const CompactPairs pairs = pair_contigs_compact(ix, P);
			{
				std::ofstream out(base + "_pair.tsv");
				write_pair_map_compact(out, ix, pairs);
			}
			ScaffoldGraph g;
			create_graph_compact(pairs, ix, g, P);
			if (P.max_degree != 0)
				remove_degree_nodes(g, P.max_degree);
			{
				std::ofstream out(base + "_original.gv");
				write_graph(out, g);
			}
			{
				std::ofstream out(base + ".dist.gv");
				std::string err;
				if (!write_dist_graph(out, len, g, P.gap, &err, false, false)) {
					std::cerr << err << "\n";
					return 1;
				}
			}
			{
				const size_t n = count_barcodes_compact(ix, mult, P);
				std::ofstream f(base + "_main.tsv");
				// (GRAPH_THREADS / GRAPH_TSV_BLOCK: the text of the file put together by several threads, in small pieces)
				const char* gt = std::getenv("GRAPH_THREADS");
				const char* gb = std::getenv("GRAPH_TSV_BLOCK");
				write_tsv_compact(f, ix, pairs, n, P, gt ? (unsigned)std::atoi(gt) : 1u, gb ? (size_t)std::atoll(gb) : (size_t)1 << 16);
			}
			{
				std::ofstream f(base + "_counts.tsv");
				write_barcode_counts(f, mult);
			}
			return 0;
		}
		add_opposite_ends(imap);
		PairMap pmap;
		pair_contigs(imap, pmap, mult, P);
		{
			std::ofstream out(base + "_pair.tsv");
			write_pair_map(out, pmap);
		}
		ScaffoldGraph g;
		create_graph(pmap, g, P);
		if (argc >= 17) { // calcDistanceEstimates runs on the graph before the degree filter (Arcs.cpp:1921-1931)
			P.dist_est = true;
			P.end_length = std::atoi(argv[14]);
			P.dist_bin_size = (unsigned)std::atoi(argv[15]);
			P.dist_upper = std::atoi(argv[16]) != 0;
			DistSampleMap samples;
			calc_dist_samples(imap, len, mult, P, samples);
			{
				std::ofstream f(base + "_samples.tsv");
				write_dist_samples_tsv(f, samples);
			}
			JaccardToDist j2d;
			build_jaccard_to_dist(samples, j2d);
			PairToBarcodeStats stats;
			build_pair_to_barcode_stats(imap, mult, len, P, stats);
			add_edge_distances(stats, j2d, P, g);
			std::ofstream f(base + "_dist.tsv");
			write_dist_tsv(f, stats, g);
		}
		if (P.max_degree != 0)
			remove_degree_nodes(g, P.max_degree);
		{
			std::ofstream out(base + "_original.gv");
			write_graph(out, g);
		}
		{
			std::ofstream out(base + ".dist.gv");
			std::string err;
			if (!write_dist_graph(out, len, g, P.gap, &err, P.dist_est, P.dist_upper)) {
				std::cerr << err << "\n";
				return 1;
			}
		}
		{
			const size_t n = count_barcodes(imap, mult, P);
			std::ofstream f(base + "_main.tsv");
			write_tsv(f, imap, pmap, n, P);
		}
		{
			std::ofstream f(base + "_counts.tsv");
			write_barcode_counts(f, mult);
		}
		return 0;
	}
	std::cerr << "usage: graph_check imap ... | gv ...\n";
	return 2;
}
