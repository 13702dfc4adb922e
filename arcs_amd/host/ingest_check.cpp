// ingest_check.cpp -- test driver for ingest.hpp (no GPU work: parse, gate and pack only).
//
//   ingest_check <threads> <batch_pairs> <multiplicity.tsv | -> <reads.fq[.gz]>...
//
// `-` instead of a multiplicity file = the fused mode (barcode pre-pass inside the mapping pass): the
// multiplicities and the per-file pre-pass summaries are printed as well.
//
// Prints, per input file in file order: the stage counters, the stdout messages of the record loop,
// and a digest of everything the GPU stage would receive (per read: length, class, packed code and
// N-mask words; per pair: gate and barcode).  The digest is independent of thread count and batch
// size; tests/test_host_ingest.py recomputes it from the FASTQ text.
#include "ingest.hpp"

#include <cstdio>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>

using namespace arks_host;

namespace {

struct Fnv
{
	uint64_t h = 1469598103934665603ull;
	void bytes(const void* p, size_t n)
	{
		const unsigned char* b = (const unsigned char*)p;
		for (size_t i = 0; i < n; ++i) {
			h ^= b[i];
			h *= 1099511628211ull;
		}
	}
	template <typename T>
	void pod(T v)
	{
		bytes(&v, sizeof v);
	}
};

} // namespace

int
main(int argc, char** argv)
{
	if (argc == 3 && std::string(argv[1]) == "helpdesk-throw") {
		// HelpDesk::parallel_for with bodies that throw on the caller and on helper threads: every body is counted
		// as done (nobody is left holding the loop's function), the first exception reaches the caller, and the desk
		// serves the next loop
		const int helpers = std::atoi(argv[2]);
		HelpDesk desk;
		std::atomic<bool> stop{ false };
		std::vector<std::thread> ts;
		for (int t = 0; t < helpers; ++t)
			ts.emplace_back([&] {
				while (!stop.load())
					if (!desk.help())
						std::this_thread::yield();
			});
		int caught = 0;
		std::atomic<long> ran{ 0 };
		for (int round = 0; round < 200; ++round) {
			try {
				desk.parallel_for(64, [&](size_t i) {
					ran.fetch_add(1);
					if (i % 7 == 3)
						throw std::bad_alloc();
				});
			} catch (const std::bad_alloc&) {
				caught++;
			}
		}
		std::atomic<long> clean{ 0 }; // (a plain long here was the test's own data race: TSan, round 6)
		desk.parallel_for(1000, [&](size_t) { clean.fetch_add(1); });
		stop.store(true);
		for (auto& t : ts)
			t.join();
		std::cout << "caught " << caught << " ran " << ran.load() << " clean " << (clean.load() > 0) << "\n";
		return caught == 200 && ran.load() == 200 * 64 ? 0 : 1;
	}
	if (argc < 5) {
		std::cerr << "usage: ingest_check <threads> <batch_pairs> <multiplicity.tsv> <reads>...\n";
		return 2;
	}
	const unsigned threads = (unsigned)std::atoi(argv[1]);
	const long batch_pairs = std::atol(argv[2]);
	std::unordered_map<std::string, int> mult;
	const bool fused = std::string(argv[3]) == "-";
	if (!fused) {
		std::ifstream in(argv[3]);
		std::string bc;
		int m;
		while (in >> bc >> m)
			mult[bc] = m;
	}
	std::vector<std::string> files(argv + 4, argv + argc);
	std::vector<std::unique_ptr<SeqReader>> readers;
	std::vector<SeqReader*> rdp;
	for (const auto& f : files) {
		readers.emplace_back(new SeqReader(f.c_str(), threads)); // bgzip'ed input is inflated in parallel
		if (!readers.back()->ok()) {
			std::cerr << "File " << f << " cannot be opened.\n";
			return 1;
		}
		rdp.push_back(readers.back().get());
	}
	const BarcodeDict dict(mult);
	IngestPipeline pipe(rdp, fused ? nullptr : &dict, batch_pairs, true, threads, HostAllocator());
	const size_t nf = files.size();
	struct PerBatch
	{
		std::vector<uint64_t> read_hash, pair_hash;
		std::string messages;
	};
	std::vector<std::map<int64_t, PerBatch>> got(nf);
	std::vector<FileCounters> fc(nf);
	std::vector<uint64_t> pairs(nf), reads(nf), batches(nf);
	const bool null_consumer = std::getenv("INGEST_NULL") != nullptr; // timing runs: parse + pack only
	const int rc = pipe.run([&](PackedBatch* pb) {
		if (null_consumer) {
			pipe.recycle(pb);
			return ARKS_OK;
		}
		const size_t f = (size_t)pb->file;
		PerBatch& out = got[f][pb->seq];
		out.messages = pb->messages;
		fc[f].skipped_unpaired += pb->fc.skipped_unpaired, fc[f].emptybarcode += pb->fc.emptybarcode,
		    fc[f].invalidbarcode += pb->fc.invalidbarcode, fc[f].gated += pb->fc.gated,
		    fc[f].skipped_invalid += pb->fc.skipped_invalid;
		pairs[f] += (uint64_t)pb->n_pairs, reads[f] += (uint64_t)pb->n_reads, batches[f]++;
		for (int64_t r = 0; r < pb->n_reads; ++r) {
			Fnv h;
			h.pod<uint32_t>(pb->len[r]);
			h.pod<uint8_t>(pb->cls[r]);
			for (uint64_t w = pb->woff[r]; w < pb->woff[r + 1]; ++w) {
				h.pod<uint64_t>(pb->codes[w]);
				h.pod<uint32_t>(pb->nmask[w]);
			}
			out.read_hash.push_back(h.h);
		}
		for (int64_t p = 0; p < pb->n_pairs; ++p) {
			Fnv h;
			h.pod<uint8_t>(pb->pair_ok[p]);
			if (pb->pair_ok[p]) {
				const std::string& name = fused ? pipe.dynamic().name(pb->barcode_id[p]) : *dict.name[pb->barcode_id[p]];
				h.bytes(name.data(), name.size());
			}
			out.pair_hash.push_back(h.h);
		}
		pipe.recycle(pb);
		return ARKS_OK;
	});
	if (rc != ARKS_OK) {
		std::cerr << "ingest failed: " << arks_strerror(rc) << "\n";
		return 1;
	}
	std::printf("threads producers=%u packers=%u\n", pipe.producers(), pipe.packers());
	for (size_t f = 0; f < nf; ++f) {
		Fnv d;
		std::string messages;
		for (const auto& kv : got[f]) {
			for (uint64_t h : kv.second.read_hash)
				d.pod(h);
			messages += kv.second.messages;
		}
		for (const auto& kv : got[f])
			for (uint64_t h : kv.second.pair_hash)
				d.pod(h);
		std::printf("file %zu pairs=%llu reads=%llu unpaired=%llu empty=%llu invalid=%llu gated=%llu "
		            "skipped_invalid=%llu digest=%016llx multibatch=%d\n",
		            f, (unsigned long long)pairs[f], (unsigned long long)reads[f],
		            (unsigned long long)fc[f].skipped_unpaired, (unsigned long long)fc[f].emptybarcode,
		            (unsigned long long)fc[f].invalidbarcode, (unsigned long long)fc[f].gated,
		            (unsigned long long)fc[f].skipped_invalid, (unsigned long long)d.h, batches[f] > 1 ? 1 : 0);
		std::fputs(messages.c_str(), stdout);
	}
	if (fused) {
		std::map<std::string, uint64_t> counts;
		for (const PrepassInfo& pi : pipe.prepass())
			for (size_t id = 0; id < pi.counts.size(); ++id)
				if (pi.counts[id])
					counts[pipe.dynamic().name((uint32_t)id)] += pi.counts[id];
		for (size_t f = 0; f < nf; ++f) {
			const PrepassInfo& pi = pipe.prepass()[f];
			std::printf("prepass %zu total=%llu lead=%llu zero_len=%d untagged_at=", f, (unsigned long long)pi.total,
			            (unsigned long long)pi.lead, pi.zero_len ? 1 : 0);
			for (uint64_t x : pi.untagged_at)
				std::printf("%llu,", (unsigned long long)x);
			std::printf("\n");
		}
		for (const auto& kv : counts)
			std::printf("mult\t%s\t%llu\n", kv.first.c_str(), (unsigned long long)kv.second);
	}
	return 0;
}
