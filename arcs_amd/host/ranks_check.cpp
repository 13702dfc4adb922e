// ranks_check.cpp -- CPU test driver of rank_merge.hpp (tests/test_host_ranks.py): a seeded scenario of
// F read files (per file: stored pairs as (barcode, conreci, sequence number), counters, messages, the
// barcode pre-pass summary) is turned into the results of `world` ranks -- file f belongs to rank f mod
// world, every rank numbers its barcodes in its own order -- the worker ranks are real processes that send
// their result through a pipe, and rank 0 merges.  Prints the scenario (for the test's own expectation),
// the merged log, the multiplicities and the IndexMap in the container's iteration order: the output must
// not depend on `world`.
// usage: ranks_check <world> <seed> <fused 0|1>
#include "rank_merge.hpp"

#include <sstream>

#include <sys/wait.h>

#include <cstdio>
#include <random>

using namespace arks_host;

struct StoredPair
{
	std::string barcode;
	uint32_t conreci;
	uint64_t seq;
};

struct FileScenario
{
	std::vector<StoredPair> pairs;
	FileResult summary; // counters, messages, pre-pass
	std::vector<std::pair<std::string, uint32_t>> reads_per_barcode;
};

static RankResult
rank_result(const std::vector<FileScenario>& files, int rank, int world, size_t nk)
{
	RankResult r;
	r.files.resize(files.size());
	r.triples.resize(nk);
	r.first.resize(nk);
	std::unordered_map<std::string, uint32_t> id;
	auto id_of = [&](const std::string& b) {
		auto it = id.find(b);
		if (it == id.end()) {
			it = id.emplace(b, (uint32_t)r.names.size()).first;
			r.names.push_back(b);
		}
		return it->second;
	};
	// a rank's own numbering: rank r starts with r dummy names so that ids differ between ranks
	for (int x = 0; x < rank; ++x)
		id_of("unused-" + std::to_string(rank) + "-" + std::to_string(x));
	std::vector<std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint64_t>>> acc(nk);
	for (size_t f = 0; f < files.size(); ++f) {
		if ((int)(f % (size_t)world) != rank)
			continue;
		FileResult fr = files[f].summary;
		fr.have = true;
		for (const auto& bc : files[f].reads_per_barcode)
			fr.pre_counts.emplace_back(id_of(bc.first), bc.second);
		r.files[f] = fr;
		for (size_t ki = 0; ki < nk; ++ki)
			for (const StoredPair& p : files[f].pairs) {
				if (ki == 1 && (p.seq & 1)) // the second k stores a subset
					continue;
				auto& e = acc[ki][{ id_of(p.barcode), p.conreci }];
				if (e.first == 0)
					e.second = p.seq;
				e.first++;
				e.second = std::min(e.second, p.seq);
			}
	}
	for (size_t ki = 0; ki < nk; ++ki)
		for (const auto& kv : acc[ki]) {
			r.triples[ki].push_back(kv.first.first);
			r.triples[ki].push_back(kv.first.second);
			r.triples[ki].push_back(kv.second.first);
			r.first[ki].push_back(kv.second.second);
		}
	return r;
}

// `ranks_check lanes <seed>`: merge_lane_entries (the GPU lanes of one process: the same barcode ids, every lane's
// entries sorted by key) against a fold through std::map
static int
check_lanes(unsigned seed)
{
	std::mt19937_64 rng(seed);
	for (int round = 0; round < 200; ++round) {
		const size_t L = 1 + rng() % 5;
		std::vector<std::map<uint64_t, std::pair<uint32_t, uint64_t>>> lanes(L);
		std::map<uint64_t, std::pair<uint64_t, uint64_t>> want;
		const size_t n = rng() % 400;
		for (size_t i = 0; i < n; ++i) {
			const uint64_t key = ((uint64_t)(rng() % 40) << 32) | (uint64_t)(1 + rng() % 12);
			const uint32_t cnt = 1 + (uint32_t)(rng() % 9);
			const uint64_t first = rng() % 100000;
			auto& e = lanes[rng() % L][key];
			e.second = e.first ? std::min(e.second, first) : first;
			e.first += cnt;
			auto& w = want[key];
			w.second = w.first ? std::min(w.second, first) : first;
			w.first += cnt;
		}
		std::vector<std::vector<uint32_t>> lt(L);
		std::vector<std::vector<uint64_t>> lf(L);
		for (size_t l = 0; l < L; ++l)
			for (const auto& kv : lanes[l]) {
				lt[l].push_back((uint32_t)(kv.first >> 32)), lt[l].push_back((uint32_t)kv.first), lt[l].push_back(kv.second.first);
				lf[l].push_back(kv.second.second);
			}
		std::vector<uint32_t> triples;
		std::vector<uint64_t> first;
		merge_lane_entries(lt, lf, triples, first);
		if (first.size() != want.size() || triples.size() != 3 * want.size()) {
			std::printf("lanes: size mismatch in round %d\n", round);
			return 1;
		}
		size_t i = 0;
		for (const auto& kv : want) {
			if (triples[3 * i] != (uint32_t)(kv.first >> 32) || triples[3 * i + 1] != (uint32_t)kv.first ||
			    triples[3 * i + 2] != (uint32_t)kv.second.first || first[i] != kv.second.second) {
				std::printf("lanes: entry %zu differs in round %d\n", i, round);
				return 1;
			}
			++i;
		}
	}
	std::printf("lanes ok\n");
	return 0;
}

int
main(int argc, char** argv)
{
	if (argc == 3 && std::string(argv[1]) == "lanes")
		return check_lanes((unsigned)std::atoi(argv[2]));
	const int world = argc > 1 ? std::atoi(argv[1]) : 1;
	const unsigned seed = argc > 2 ? (unsigned)std::atoi(argv[2]) : 1;
	const bool fused = argc > 3 && std::atoi(argv[3]) != 0;
	const size_t nk = 2, nf = 5, n_ends = 12;
	std::mt19937_64 rng(seed);
	std::vector<std::string> names;
	std::vector<FileScenario> files(nf);
	std::vector<CI> contigRecord;
	contigRecord.push_back(CI("null contig", false));
	for (size_t e = 0; e < n_ends; ++e)
		contigRecord.push_back(CI("ctg" + std::to_string(e / 2 + 1), e % 2 == 0));
	std::unordered_map<std::string, int> mult;
	for (size_t f = 0; f < nf; ++f) {
		FileScenario& fs = files[f];
		names.push_back("reads" + std::to_string(f) + ".fq");
		const size_t np = f == 3 ? 0 : 40 + (size_t)(rng() % 60); // file 3 is empty
		std::map<std::string, uint32_t> rpb;
		for (size_t p = 0; p < np; ++p) {
			// barcodes recur across files; the first stored pair of a barcode decides its place
			const std::string bc = "BC" + std::to_string(rng() % 25);
			rpb[bc] += 2;
			if (rng() % 3 == 0)
				continue; // not stored
			fs.pairs.push_back(StoredPair{ bc, 1 + (uint32_t)(rng() % n_ends), ((uint64_t)f << 48) | ((uint64_t)(p / 16) << 24) | (p % 16) });
		}
		fs.summary.fc.gated = np - np / 7;
		fs.summary.fc.skipped_unpaired = np / 11;
		fs.summary.fc.skipped_invalid = np / 13;
		fs.summary.fc.emptybarcode = f == 1 ? 3 : 0;
		fs.summary.fc.invalidbarcode = f == 2 ? 2 : 0;
		if (np)
			fs.summary.messages[(int64_t)(rng() % 4)] = "File contains unpaired reads: a" + std::to_string(f) + " b\n";
		for (size_t ki = 0; ki < nk; ++ki) {
			uint64_t stored = 0;
			for (const StoredPair& p : fs.pairs)
				stored += !(ki == 1 && (p.seq & 1));
			fs.summary.stored.push_back(stored);
			arks_map_stats st;
			std::memset(&st, 0, sizeof st);
			st.total_valid = 1000 * (f + 1) + ki, st.bad = 7 * f, st.found = 500 * (f + 1), st.recorded = 400 * (f + 1),
			st.dups = 100 * (f + 1), st.reads_pass = 2 * np / 3, st.reads_fail = 2 * np - 2 * np / 3, st.windows = st.total_valid + st.bad;
			fs.summary.st.push_back(st);
		}
		fs.summary.pre_total = 2 * np;
		fs.summary.pre_lead = f == 0 ? 2 : 0;
		for (const auto& kv : rpb)
			fs.reads_per_barcode.emplace_back(kv.first, kv.second);
		if (!fused)
			for (const auto& kv : rpb)
				mult[kv.first] += (int)kv.second;
	}
	// the scenario, for the test's own expectation
	for (size_t f = 0; f < nf; ++f)
		for (const StoredPair& p : files[f].pairs)
			std::printf("PAIR %zu %s %u %llu\n", f, p.barcode.c_str(), p.conreci, (unsigned long long)p.seq);

	std::vector<RankResult> ranks;
	std::vector<std::pair<pid_t, int>> workers;
	std::fflush(nullptr);
	for (int r = 1; r < world; ++r) {
		int fds[2];
		if (::pipe(fds) != 0)
			return 2;
		const pid_t pid = ::fork();
		if (pid == 0) {
			::close(fds[0]);
			send_result(fds[1], rank_result(files, r, world, nk));
			::close(fds[1]);
			_exit(0);
		}
		::close(fds[1]);
		workers.emplace_back(pid, fds[0]);
	}
	ranks.push_back(rank_result(files, 0, world, nk));
	for (const auto& wk : workers) {
		ranks.emplace_back();
		if (!receive_result(wk.second, ranks.back()))
			return 3;
		::close(wk.second);
		int status = 0;
		::waitpid(wk.first, &status, 0);
		if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
			return 4;
	}
	std::vector<IndexMap> imaps;
	std::string out, err, pre_out, pre_err;
	const MergeParams mp{ true, { 40, 60 }, 1 };
	merge_results(names, ranks, imaps, mult, contigRecord, fused, mp, out, err, &pre_out, &pre_err);
	std::printf("PRE\n%s%sOUT\n%sERR\n%s", pre_out.c_str(), pre_err.c_str(), out.c_str(), err.c_str());
	std::map<std::string, int> sorted(mult.begin(), mult.end());
	for (const auto& kv : sorted)
		std::printf("MULT %s %d\n", kv.first.c_str(), kv.second);
	for (size_t ki = 0; ki < imaps.size(); ++ki)
		for (const auto& b : imaps[ki]) // iteration order of the unordered container: depends on the creation order
			for (const auto& e : b.second)
				std::printf("IMAP %zu %s %s %c %d\n", ki, b.first.c_str(), e.first.first.c_str(), e.first.second ? 'H' : 'T', e.second);
	// the same content through the number-based merge (graph_fast.hpp): what its pairing and TSV stages see must
	// not depend on the number of ranks either, and must be what the IndexMap above gives
	{
		std::vector<CompactIndex> cix;
		std::vector<IndexMap> none;
		std::unordered_map<std::string, int> mult2 = mult;
		std::string o2, e2, po2, pe2;
		merge_results(names, ranks, none, mult2, contigRecord, false, mp, o2, e2, &po2, &pe2, &cix);
		GraphParams P;
		P.min_mult = 1;
		P.min_reads = 1;
		for (size_t ki = 0; ki < cix.size(); ++ki) {
			const CompactPairs a = pair_contigs_compact(cix[ki], P), b = pair_contigs_compact(compact_from_imap(imaps[ki], mult), P);
			const CompactIndex lit = compact_from_imap(imaps[ki], mult);
			bool same = a.size() == b.size() && cix[ki].n_barcodes == imaps[ki].size() && cix[ki].entries.size() == lit.entries.size();
			for (size_t i = 0; same && i < a.size(); ++i)
				same = cix[ki].contig[a[i].a] == lit.contig[b[i].a] && cix[ki].contig[a[i].b] == lit.contig[b[i].b] &&
				       std::equal(a[i].cnt, a[i].cnt + 4, b[i].cnt);
			std::printf("COMPACT %zu %s barcodes %zu entries %zu pairs %zu\n", ki, same ? "same" : "DIFFERENT", cix[ki].n_barcodes,
			            cix[ki].entries.size(), a.size());
			std::ostringstream t1, t2;
			write_tsv_compact(t1, cix[ki], a, 7, P);
			write_tsv_compact(t2, lit, b, 7, P);
			std::printf("COMPACT %zu tsv %s %zu bytes\n", ki, t1.str() == t2.str() ? "same" : "DIFFERENT", t1.str().size());
		}
	}
	return 0;
}
