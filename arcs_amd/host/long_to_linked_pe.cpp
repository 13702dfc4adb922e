// long_to_linked_pe.cpp -- cuts long reads into pseudo-linked read pairs (the feeder of the
// `arcs-make arks-long` pipeline, bin/arcs-make:299-313), restating the behaviour of the
// reference's src/long-to-linked-pe.cpp:185-319 without btllib:
//   * reads shorter than max(2L, M) are dropped; every 2L bases give one pair = first L bases forward
//     + reverse complement of the next L bases; both mates carry "BX:Z:<record number + 1>";
//   * a remainder (length % 2L != 0) gives one more pair: forward = up to L bases of the remainder,
//     reverse = reverse complement of the LAST |forward| bases of the read (:255-287);
//   * --bx / --bx-only write "<record number + 1> \t <number of reads>" per kept read (:194-204);
//   * -s / -d append tigmint-long's span / dist estimates to the parameter file (:294-319).
// Record numbers restart at 0 for every input file, as btllib's reader numbers them.
#include "seqio.hpp"

#include <getopt.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

using arks_host::SeqReader;

static void
revcomp(std::string& s)
{
	std::reverse(s.begin(), s.end());
	for (char& c : s) {
		switch (c) {
		case 'A': c = 'T'; break;
		case 'C': c = 'G'; break;
		case 'G': c = 'C'; break;
		case 'T': c = 'A'; break;
		case 'a': c = 't'; break;
		case 'c': c = 'g'; break;
		case 'g': c = 'c'; break;
		case 't': c = 'a'; break;
		case 'U': c = 'A'; break;
		case 'u': c = 'a'; break;
		case 'R': c = 'Y'; break;
		case 'Y': c = 'R'; break;
		case 'K': c = 'M'; break;
		case 'M': c = 'K'; break;
		case 'B': c = 'V'; break;
		case 'V': c = 'B'; break;
		case 'D': c = 'H'; break;
		case 'H': c = 'D'; break;
		case 'r': c = 'y'; break;
		case 'y': c = 'r'; break;
		case 'k': c = 'm'; break;
		case 'm': c = 'k'; break;
		case 'b': c = 'v'; break;
		case 'v': c = 'b'; break;
		case 'd': c = 'h'; break;
		case 'h': c = 'd'; break;
		default: break; // N, S, W and anything else map to themselves
		}
	}
}

int
main(int argc, char* argv[])
{
	static int help = 0, version = 0, with_fasta = 0, with_bx = 0, with_bx_only = 0;
	bool auto_span = false, auto_dist = false, l_set = false, g_set = false;
	size_t l = 0, g = 0, m = 2000;
	unsigned threads = 1;
	double cov_to_span = 0.25, dist_read_perc = 50;
	const size_t dist_lower_bound = 1000;
	std::string configFile("tigmint-long.params.tsv"), bxFile("barcode_multiplicity.tsv");
	static const struct option longopts[] = { { "bx", no_argument, &with_bx, 1 },
		                                      { "bx-only", no_argument, &with_bx_only, 1 },
		                                      { "fasta", no_argument, &with_fasta, 1 },
		                                      { "help", no_argument, &help, 1 },
		                                      { "version", no_argument, &version, 1 },
		                                      { nullptr, 0, nullptr, 0 } };
	int c, optindex = 0;
	bool failed = false;
	while ((c = getopt_long(argc, argv, "l:g:o:c:p:sdf:t:b:m:v", longopts, &optindex)) != -1) {
		switch (c) {
		case 0: break;
		case 'l': l_set = true; l = std::stoul(optarg); break;
		case 'm': m = std::stoul(optarg); break;
		case 'g': g_set = true; g = (size_t)std::stod(optarg); break;
		case 'p': dist_read_perc = std::stod(optarg); break;
		case 'c': cov_to_span = std::stod(optarg); break;
		case 't': threads = (unsigned)std::max(1L, std::atol(optarg)); break; // inflate threads for bgzip'ed input
		case 's': auto_span = true; break;
		case 'd': auto_dist = true; break;
		case 'f': configFile = optarg; break;
		case 'b': bxFile = optarg; break;
		case 'v': break;
		default: failed = true; break;
		}
	}
	if (help || version) {
		std::cerr << "long-to-linked-pe v1.0 (MI355X build of bcgsc/arcs)\n"
		             "Usage: long-to-linked-pe -l L [-m M] [--fasta] [--bx | --bx-only] [-b FILE] [-s -g G -c C] "
		             "[-d -p P] [-f FILE] READS...\n";
		return 0;
	}
	if (!l_set) {
		std::cerr << "long-to-linked-pe v1.0: missing option -- 'l'\n";
		failed = true;
	}
	if (auto_span && !g_set) {
		std::cerr << "long-to-linked-pe v1.0: missing option -- 'g'\n";
		failed = true;
	}
	std::vector<std::string> infiles(argv + optind, argv + argc);
	if (infiles.empty()) {
		std::cerr << "long-to-linked-pe v1.0: missing file operand\n";
		failed = true;
	}
	if (failed)
		return EXIT_FAILURE;
	const char header_symbol = with_fasta ? '>' : '@';
	std::ofstream bx_ofs;
	if (with_bx || with_bx_only)
		bx_ofs.open(bxFile);
	std::vector<size_t> read_lengths;
	size_t total_bases = 0;
	std::string out;
	out.reserve(1 << 20);
	auto emit = [&](const std::string& id, int read_num, size_t num, const std::string& s, const std::string& q) {
		out += header_symbol;
		out += id;
		out += "_f";
		out += std::to_string(read_num);
		out += " BX:Z:";
		out += std::to_string(num + 1);
		out += '\n';
		out += s;
		out += '\n';
		if (!with_fasta) {
			out += "+\n";
			out += q;
			out += '\n';
		}
	};
	for (const auto& infile : infiles) {
		SeqReader rd(infile.c_str(), threads);
		if (!rd.ok()) {
			std::cerr << "long-to-linked-pe v1.0: cannot open " << infile << "\n";
			return EXIT_FAILURE;
		}
		size_t num = 0;
		for (; rd.next() >= 0; ++num) {
			const size_t step = l * 2;
			// (bytes as they are: the reference opens btllib::SeqReader with Flag::LONG_MODE alone, without
			// FOLD_CASE or TRIM_MASKED -- src/long-to-linked-pe.cpp:186-188 -- so lower-case bases stay lower case)
			const std::string& seq = rd.seq;
			const size_t n = seq.size();
			if (with_bx || with_bx_only) {
				if (step > n || m > n)
					continue;
				if (n % step != 0)
					bx_ofs << num + 1 << "\t" << (n / step + 1) * 2 << std::endl;
				else
					bx_ofs << num + 1 << "\t" << n / l << std::endl;
			}
			if (with_bx_only)
				continue;
			if (auto_dist && n > dist_lower_bound)
				read_lengths.push_back(n);
			if (auto_span)
				total_bases += n;
			if (step > n || m > n)
				continue;
			const std::string& qual = rd.qual;
			const bool has_q = !qual.empty();
			int read_num = 1;
			for (size_t i = 0; i <= n - step; i += step) {
				emit(rd.name, read_num, num, seq.substr(i, l), has_q ? qual.substr(i, l) : std::string(l, '#'));
				std::string r = seq.substr(i + l, l);
				revcomp(r);
				std::string rq = has_q ? qual.substr(i + l, l) : std::string(l, '#');
				std::reverse(rq.begin(), rq.end());
				emit(rd.name, read_num, num, r, rq);
				++read_num;
			}
			const size_t rem = n % step;
			if (rem != 0) {
				const size_t cur = n - rem;
				const std::string fwd = seq.substr(cur, l);
				emit(rd.name, read_num, num, fwd, has_q ? qual.substr(cur, l) : std::string(fwd.size(), '#'));
				std::string r = seq.substr(n - fwd.size(), fwd.size());
				revcomp(r);
				std::string rq = has_q ? qual.substr(n - fwd.size(), fwd.size()) : std::string(fwd.size(), '#');
				std::reverse(rq.begin(), rq.end());
				emit(rd.name, read_num, num, r, rq);
			}
			if (out.size() > (1 << 19)) {
				std::fwrite(out.data(), 1, out.size(), stdout);
				out.clear();
			}
		}
	}
	std::fwrite(out.data(), 1, out.size(), stdout);
	std::fflush(stdout);
	if (auto_span || auto_dist) {
		std::ofstream ofs(configFile, std::ofstream::app);
		if (auto_span)
			ofs << "span\t" << (size_t)(total_bases / g * cov_to_span) << "\n";
		if (auto_dist) {
			if (read_lengths.empty())
				std::cerr << "long-to-linked-pe: unable to estimate dist parameter due to no valid lengths" << std::endl;
			else {
				std::sort(read_lengths.begin(), read_lengths.end());
				const double index = (dist_read_perc / 100) * read_lengths.size();
				const size_t ii = (size_t)std::floor(index);
				const size_t est = std::floor(index) == index ? (read_lengths[ii - 1] + read_lengths[ii]) / 2 : read_lengths[ii];
				ofs << "read_p" << dist_read_perc << "\t" << est << "\n";
			}
		}
	}
	return 0;
}
