// arcs.cpp -- host front end `arcs --arks ...` over libarks_hip.so (MI355X).
//
// Keeps the command line of the reference's `arcs` (Arcs/Arcs.cpp:105-157, main :1961-2184) and the
// files it writes (<base>_original.gv, <base>.dist.gv, <base>_main.tsv, optional barcode counts
// and pair TSV), so that bin/arcs-make (`arcs --arks -v -f draft.fa -c -m -r -e -z -j -k -t -d --gap
// -b base reads.fq.gz --barcode-counts x.tsv`, arcs-make:290) -> makeTSVfile.py -> LINKS runs
// unchanged.  The ARKS stages of runArcs (Arcs.cpp:1871-1903) run as:
//   barcode multiplicities   host  (createIndexMultMap :392-448 / readBarcodes :481-547)
//   contig ends -> index     GPU   (arks_index_build; replaces getContigKmers/mapKmers)
//   read pairs -> IndexMap   host parse (chromiumRead :1185-1262) + GPU (gate, bestContig x2,
//                            pair rule, imap accumulation)
//   graph stage and writers  host  (graph.hpp)
// Not offered: the alignment (SAM/BAM) mode (outside the ARKS k-mer path), reported as an error
// instead of being silently ignored.  -D distance estimation is offered (dist_est.hpp).
#include "arks_hip.h"
#include "dist_est.hpp"
#include "graph.hpp"
#include "ingest.hpp"
#include "rank_merge.hpp"
#include "seqio.hpp"

#include <getopt.h>
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <array>
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <ctime>
#include <map>
#include <memory>

using namespace arks_host;

#define PROGRAM "arcs"
#define PACKAGE_VERSION "1.2.8-mi355x"

namespace {

struct Params
{
	std::string file, fofName, base_name, dist_graph_name, tsv_name, barcode_counts_name, multfile;
	int seq_id = 98;
	int min_size = 500;
	int end_length = 30000;
	int verbose = 0;
	int k_value = 30;
	std::string k_arg = "30"; // as given
	std::vector<int> k_list; // -k 40,60,80 (this build only): one pass over the reads, one output set per k
	double j_index = 0.55;
	unsigned threads = 1;
	bool arks = false, output_pair = false, dist_est = false;
	unsigned dist_bin_size = 20;
	std::string dist_samples_tsv, dist_tsv;
	GraphParams g;
	long batch_pairs = 262144; // --batch-pairs (this build only): read pairs per GPU batch
	int index_shards = 1;      // --index-shards (this build only): the contig k-mer index in N parts
	int device = 0;             // --device (this build only)
	int ranks = 1;              // --ranks (this build only): N GPU ranks -- processes the read files are dealt to, and GPU
	                            // lanes inside a process when there are fewer files than ranks
	int index_sharded = 0;      // --index-sharded[=N] (this build only): the seed table hash-sharded over N ranks of this
	                            // process (arks_exchange, BASELINE configs[3]); 0 = replicas
	bool share_devices = false; // --share-devices (this build only): more GPU ranks than visible devices is not an error
	int lanes = 1;              // GPU lanes of THIS process (set by run_arks): device + 0 .. lanes - 1
};

Params params;

enum
{
	OPT_HELP = 1,
	OPT_VERSION,
	OPT_BX,
	OPT_GAP,
	OPT_TSV,
	OPT_BARCODE_COUNTS,
	OPT_SAMPLES_TSV,
	OPT_DIST_TSV,
	OPT_NO_DIST_EST,
	OPT_DIST_MEDIAN,
	OPT_DIST_UPPER,
	OPT_ARKS_METHOD,
	OPT_BATCH_PAIRS,
	OPT_DEVICE,
	OPT_INDEX_SHARDS,
	OPT_RANKS,
	OPT_INDEX_SHARDED,
	OPT_SHARE_DEVICES
};

const char shortopts[] = "f:a:B:s:c:Dl:z:b:g:m:d:e:r:vt:u:j:k:P";

const struct option longopts[] = {
	{ "file", required_argument, NULL, 'f' },
	{ "fofName", required_argument, NULL, 'a' },
	{ "bin_size", required_argument, NULL, 'B' },
	{ "bx", no_argument, NULL, OPT_BX },
	{ "samples_tsv", required_argument, NULL, OPT_SAMPLES_TSV },
	{ "dist_tsv", required_argument, NULL, OPT_DIST_TSV },
	{ "seq_id", required_argument, NULL, 's' },
	{ "min_reads", required_argument, NULL, 'c' },
	{ "dist_est", no_argument, NULL, 'D' },
	{ "no_dist_est", no_argument, NULL, OPT_NO_DIST_EST },
	{ "dist_median", no_argument, NULL, OPT_DIST_MEDIAN },
	{ "dist_upper", no_argument, NULL, OPT_DIST_UPPER },
	{ "min_links", required_argument, NULL, 'l' },
	{ "min_size", required_argument, NULL, 'z' },
	{ "base_name", required_argument, NULL, 'b' },
	{ "graph", required_argument, NULL, 'g' },
	{ "tsv", required_argument, NULL, OPT_TSV },
	{ "barcode-counts", required_argument, NULL, OPT_BARCODE_COUNTS },
	{ "gap", required_argument, NULL, OPT_GAP },
	{ "index_multiplicity", required_argument, NULL, 'm' },
	{ "max_degree", required_argument, NULL, 'd' },
	{ "end_length", required_argument, NULL, 'e' },
	{ "error_percent", required_argument, NULL, 'r' },
	{ "run_verbose", required_argument, NULL, 'v' },
	{ "version", no_argument, NULL, OPT_VERSION },
	{ "help", no_argument, NULL, OPT_HELP },
	{ "threads", required_argument, NULL, 't' },
	{ "multfile", required_argument, NULL, 'u' },
	{ "k_value", required_argument, NULL, 'k' },
	{ "j_index", required_argument, NULL, 'j' },
	{ "arks", no_argument, NULL, OPT_ARKS_METHOD },
	{ "pair", no_argument, NULL, 'P' },
	{ "batch-pairs", required_argument, NULL, OPT_BATCH_PAIRS },
	{ "device", required_argument, NULL, OPT_DEVICE },
	{ "index-shards", required_argument, NULL, OPT_INDEX_SHARDS },
	{ "ranks", required_argument, NULL, OPT_RANKS },
	{ "index-sharded", optional_argument, NULL, OPT_INDEX_SHARDED },
	{ "share-devices", no_argument, NULL, OPT_SHARE_DEVICES },
	{ NULL, 0, NULL, 0 }
};

const char USAGE[] =
    PROGRAM " " PACKAGE_VERSION "\n\n"
            "Usage: arcs [Options] --arks -f <contig sequence file> <list of linked read files>\n\n"
            "MI355X build of the ARKS method of bcgsc/arcs: same options and output files as the\n"
            "reference for --arks; the alignment (SAM/BAM) method is not part of this build.\n\n"
            "   -a, --fofName=FILE    text file listing input filenames\n"
            "   -u, --multfile        tsv or csv file listing barcode multiplicities [optional]\n"
            "   -f, --file=FILE       FASTA file of contig sequences to scaffold\n"
            "   -c, --min_reads=N     min aligned read pairs per barcode mapping [5]\n"
            "   -l, --min_links=N     min shared barcodes between contigs [0]\n"
            "   -z, --min_size=N      min contig length [500]\n"
            "   -b, --base_name=STR   output file prefix\n"
            "   -g, --graph=FILE      write the ABySS dist.gv to FILE\n"
            "       --gap=N           fixed gap size for ABySS dist.gv file [100]\n"
            "       --tsv=FILE        write graph in TSV format to FILE\n"
            "       --barcode-counts=FILE       write number of reads per barcode to FILE\n"
            "   -m, --index_multiplicity=RANGE  barcode multiplicity range [50-10000]\n"
            "   -d, --max_degree=N    max node degree in scaffold graph [0]\n"
            "   -e, --end_length=N    contig head/tail length for masking alignments [30000]\n"
            "   -r, --error_percent=N p-value for head/tail assignment and link orientation [0.05]\n"
            "   -v, --run_verbose     verbose logging\n"
            "   -k  --k_value         size of a k-mer [30]; a list (40,60,80) maps every read batch against one\n"
            "                         index per k in a single pass and writes one output set per k\n"
            "   -j  --j_index         minimum fraction of read kmers matching a contigId [0.55]\n"
            "   -t  --threads         number of host ingest threads [1] (parse / pack; the mapping runs on the GPU)\n"
            "   -P, --pair            output scaffolds pairing TSV\n"
            "   -D, --dist_est        enable distance estimation\n"
            "       --no_dist_est     disable distance estimation [default]\n"
            "       --dist_median     use median distance in ABySS dist.gv [default]\n"
            "       --dist_upper      use upper bound distance in ABySS dist.gv\n"
            "   -B, --bin_size=N      estimate distance using N closest Jaccard scores [20]\n"
            "       --dist_tsv=FILE   write min/max distance estimates to FILE\n"
            "       --samples_tsv=FILE  write intra-contig distance/barcode samples to FILE\n"
            "       --batch-pairs=N   read pairs per GPU batch [262144]\n"
            "       --index-shards=N  build and map the contig k-mer index in N parts (very large drafts) [1]\n"
            "       --ranks=N         N GPU ranks (device, device + 1, ...), an index replica each: with N or more read\n"
            "                         files N processes the files are dealt to; with fewer files (one .fq.gz) the\n"
            "                         batches of a file are dealt to the GPUs of its process; outputs as with one [1]\n"
            "       --index-sharded[=N]  the index's seed table hash-sharded over N GPUs (default: --ranks, else all)\n"
            "                         instead of replicated: a read's seeds are answered by the GPUs that own them\n"
            "       --share-devices   let GPU ranks share a device when --ranks / --index-sharded ask for more of them than\n"
            "                         there are devices (an error otherwise: ranks that share a GPU add nothing; tests)\n"
            "       --device=N        GPU ordinal [0]\n";

void
die_arks(int rc, const char* what)
{
	std::cerr << PROGRAM ": " << what << ": " << arks_strerror(rc) << " " << arks_last_error_string() << "\n";
	exit(EXIT_FAILURE);
}

void
assert_readable(const std::string& path)
{
	if (access(path.c_str(), R_OK) == -1) {
		std::cerr << "error: `" << path << "': " << strerror(errno) << std::endl;
		exit(EXIT_FAILURE);
	}
}

std::vector<std::string>
read_fof(const std::string& fof)
{
	std::vector<std::string> v;
	if (fof.empty())
		return v;
	std::ifstream in(fof.c_str());
	std::string s;
	while (in >> s)
		v.push_back(s);
	return v;
}

// Arcs.cpp:336-361
bool
check_same_format(const std::vector<std::string>& names, bool& all_alignment)
{
	int prev = 0, cur = 0;
	for (const auto& f : names) {
		cur = 0;
		if (f.find(".sam") != std::string::npos || f.find(".bam") != std::string::npos)
			cur = 1;
		if (f.find(".fastq") != std::string::npos || f.find(".fq") != std::string::npos)
			cur = 2;
		if (!cur) {
			std::cout << "Unknown type file is observed!" << std::endl;
			return false;
		}
		if (!(!prev || prev == cur))
			return false;
		prev = cur;
	}
	all_alignment = cur == 1;
	return true;
}

// Arcs.cpp:392-448 (the parser itself: graph.hpp)
void
create_index_mult_map(const std::string& multfile, std::unordered_map<std::string, int>& mult)
{
	const size_t numbarcodes = read_multiplicity_file(multfile, mult);
	if (params.verbose)
		std::cout << "Saw " << numbarcodes << "  distinct barcodes." << std::endl;
}

// Arcs.cpp:481-547: counts READS per barcode; a record of length <= 0 ends the file
void
read_barcodes(const std::vector<std::string>& files, std::unordered_map<std::string, int>& mult)
{
	size_t added = 0;
	for (const auto& f : files) {
		if (params.verbose)
			std::cout << "Reading chrom " << f << std::endl;
		SeqReader rd(f.c_str(), std::max(1u, params.threads)); // (literal two-pass flow) bgzip input inflates in parallel
		if (!rd.ok()) {
			std::cerr << "File " << f << " cannot be opened." << std::endl;
			exit(1);
		}
		std::cerr << "File " << f << " opened." << std::endl;
		for (;;) {
			const int l = rd.next();
			if (l <= 0)
				break;
			if (rd.comment.empty())
				continue;
			const size_t tag = rd.comment.find("BX:Z:");
			if (tag != std::string::npos) {
				mult[bx_barcode(rd.comment)]++;
				added++;
			}
			if (params.verbose && added % 100000000 == 0)
				std::cout << added << " read with valid barcode" << std::endl;
		}
	}
	if (params.verbose)
		std::cout << "Saw " << mult.size() << " distinct barcode." << std::endl;
}

// Arcs.cpp:317-331
bool
check_contig_sequence(const std::string& seq)
{
	// (the reference's toupper + strchr per character is 2 ns a base: 7 s of a -v run's start-up on a 3 Gbp draft,
	// profiles/r09_bench.json; the same rule as a table, 64 characters at a time, first offender printed as before)
	static const std::array<unsigned char, 256> bad = [] {
		std::array<unsigned char, 256> t{};
		for (int ch = 0; ch < 256; ++ch) {
			const char c = (char)toupper(ch);
			t[(size_t)ch] = (!strchr("ATGCNMRWSYKVHDB", c) || c == '\0') ? 1 : 0;
		}
		return t;
	}();
	const unsigned char* p = reinterpret_cast<const unsigned char*>(seq.data());
	const size_t n = seq.size();
	for (size_t i = 0; i < n; i += 64) {
		const size_t e = i + 64 < n ? i + 64 : n;
		unsigned char any = 0;
		for (size_t x = i; x < e; ++x)
			any |= bad[p[x]];
		if (any)
			for (size_t x = i; x < e; ++x)
				if (bad[p[x]]) {
					std::cout << (char)toupper(p[x]) << std::endl;
					return false;
				}
	}
	return true;
}

const char*
maybe_na(const std::string& s)
{
	return s.empty() ? "NA" : s.c_str();
}

const char*
now()
{
	static char buf[64];
	std::time_t t;
	time(&t);
	std::strncpy(buf, ctime(&t), sizeof buf - 1);
	return buf;
}

int
memory_usage()
{
	std::ifstream proc("/proc/self/status");
	std::string s;
	while (getline(proc, s))
		if (s.compare(0, 5, "VmRSS") == 0) {
			int mem = 0;
			std::stringstream ss(s.substr(s.find_last_of('\t')));
			ss >> mem;
			return mem;
		}
	return 0;
}

template <typename T>
struct DevArray
{
	T* p = nullptr;
	size_t cap = 0;
	~DevArray()
	{
		if (p)
			(void)hipFree(p);
	}
	void reserve(size_t n)
	{
		if (n <= cap)
			return;
		if (p)
			(void)hipFree(p);
		cap = n + n / 4 + 64;
		if (hipMalloc((void**)&p, cap * sizeof(T)) != hipSuccess) {
			std::cerr << PROGRAM ": out of device memory\n";
			exit(EXIT_FAILURE);
		}
	}
};

// ---- the contig index (replaces initContigArray + getContigKmers, Arcs.cpp:451-479, 1021-1129) ----
std::vector<arks_index*>
build_contig_index(std::vector<CI>& contigRecord, ContigToLength& contigToLength, std::string& log)
{
	// pass 1 of the reference only sizes contigRecord; it also filters on the IUPAC alphabet, which
	// the second pass does not (Q10 of SURVEY.md): a contig with a foreign character would leave
	// contigRecord too short in the reference.  Here the record simply grows.
	std::string bases;
	std::vector<uint64_t> off;
	std::vector<uint32_t> len;
	int total = 0, skipped = 0, valid = 0;
	contigRecord.clear();
	contigRecord.push_back(CI("null contig", false));
	SeqReader rd(params.file.c_str(), std::max(1u, params.threads));
	{
		// the ends of a draft are most of it (two ends of up to -e bases per contig): room for them at once -- a
		// string that doubles its way to gigabytes copies itself, and faults its pages in, several times over
		struct stat st;
		if (::stat(params.file.c_str(), &st) == 0 && S_ISREG(st.st_mode))
			bases.reserve(std::min<size_t>((size_t)st.st_size * (rd.serial_source() || rd.parallel_inflate() ? 4 : 1) + 1,
			                               (size_t)8 << 30));
	}
	const auto t_read0 = std::chrono::steady_clock::now();
	size_t count = 0; // what initContigArray (Arcs.cpp:451-479) counts in a pass of its own; here in the same pass
	while (rd.next() >= 0) {
		total++;
		if (params.verbose && check_contig_sequence(rd.seq) && (int)rd.seq.length() >= params.min_size)
			count++;
		int cut = 0;
		if (arks_end_cutoff((int)rd.seq.length(), params.min_size, params.end_length, &cut)) {
			contigToLength[rd.name] = (int)rd.seq.length();
			contigRecord.push_back(CI(rd.name, true));
			off.push_back(bases.size());
			len.push_back((uint32_t)cut);
			bases.append(rd.seq, 0, (size_t)cut);
			contigRecord.push_back(CI(rd.name, false));
			off.push_back(bases.size());
			len.push_back((uint32_t)cut);
			bases.append(rd.seq, rd.seq.length() - (size_t)cut, (size_t)cut);
			valid++;
			// mapKmers' warning for an end shorter than k (Arcs.cpp:877-882: unconditional, with the
			// conreci, between the progress lines); with a k list it is printed per k further down
			if (params.k_list.size() == 1 && cut < params.k_list[0])
				for (size_t c = contigRecord.size() - 2; c < contigRecord.size(); ++c)
					appendf(log, "Warning: ends of contig is shorter than k-value for contigID (no k-mers added): %zu\n", c);
		} else
			skipped++;
		if (params.verbose && total % 1000 == 0)
			appendf(log, "Finished %d Contigs...\n", total);
	}
	if (rd.failed())
		std::cerr << PROGRAM ": warning: " << params.file
		          << ": the compressed stream is damaged or truncated; the contigs before the damage were used\n";
	bases.push_back('\0');
	if (getenv("ARKS_TIMING"))
		std::cerr << "[timing]   contig index, draft read (" << bases.size() << " bases of ends): "
		          << std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_read0).count() << " ms\n";
	if (params.verbose)
		std::cerr << "Number of contigs:" << count << "\nSize of Contig Array:" << count * 2 + 1 << std::endl;
	// lane-major: the indexes of lane 0 (every k; with --index-shards every part of a k together), then lane 1's ...
	// A lane is a GPU of this process (--ranks with fewer files than ranks, --index-sharded).  Replica lanes that share a
	// device share the index (tests on a one-GPU box); the build counters come from the first build of a k.
	std::vector<arks_index*> idxs;
	const int n_shards = std::max(1, params.index_shards);
	const int n_lanes = std::max(1, params.lanes);
	int ndev = 1;
	(void)hipGetDeviceCount(&ndev);
	std::vector<std::vector<arks_index*>> by_lane((size_t)n_lanes);
	for (const int k : params.k_list) {
		arks_build_stats st;
		std::memset(&st, 0, sizeof st);
		const auto build_lane = [&](int lane) {
			const int device = (params.device + lane) % std::max(1, ndev);
			arks_index* idx = nullptr;
			if (params.index_sharded > 0) {
				// --index-sharded: lane = rank `lane` of the seed table's hash shards; text, bitmaps, fallback table whole
				const int rc = arks_index_build_seed_shard(&idx, k, bases.data(), off.data(), len.data(), (int64_t)len.size(), lane,
				                                           n_lanes, device, params.verbose && lane == 0 ? &st : nullptr);
				if (rc != ARKS_OK)
					die_arks(rc, "building a shard of the seed table");
				by_lane[(size_t)lane].push_back(idx);
				return;
			}
			int same = -1; // an earlier lane on the same device: share its replica
			for (int l2 = 0; l2 < lane && same < 0; ++l2)
				if ((params.device + l2) % std::max(1, ndev) == device)
					same = l2;
			if (same >= 0) {
				const size_t per_k = (size_t)n_shards, at = by_lane[(size_t)lane].size();
				for (size_t x = 0; x < per_k; ++x)
					by_lane[(size_t)lane].push_back(by_lane[(size_t)same][at + x]);
				return;
			}
			if (n_shards > 1) {
				// --index-shards: N indexes of 1/N of the contigs each (a draft beyond one index's 2^32 text
				// positions, or whose build scratch does not fit); the reads are mapped against each in turn
				// -v: every part reports its share of the counters, the sums are the one map's (include/arks_hip.h)
				for (int sh = 0; sh < n_shards; ++sh) {
					arks_build_stats part;
					std::memset(&part, 0, sizeof part);
					const int rc = arks_index_build_shard_stats(&idx, k, bases.data(), off.data(), len.data(), (int64_t)len.size(),
					                                            sh, n_shards, device, params.verbose && lane == 0 ? &part : nullptr);
					if (rc != ARKS_OK)
						die_arks(rc, "building a shard of the contig k-mer index");
					by_lane[(size_t)lane].push_back(idx);
					if (lane == 0) { // (only lane 0 collects counters; the lanes of other devices run in threads of their own)
						st.total_kmers += part.total_kmers, st.null_kmers += part.null_kmers, st.short_ends += part.short_ends;
						st.recorded += part.recorded, st.collisions += part.collisions, st.removed_dup += part.removed_dup;
						st.unique += part.unique;
					}
				}
				return;
			}
			const int rc = arks_index_build(&idx, k, bases.data(), off.data(), len.data(), (int64_t)len.size(), device,
			                                params.verbose && lane == 0 ? &st : nullptr);
			if (rc != ARKS_OK)
				die_arks(rc, "building the contig k-mer index");
			by_lane[(size_t)lane].push_back(idx);
		};
		// the lanes of one device one after the other (its replicas are shared, its build scratch is one device's),
		// the devices side by side: eight GPUs build their replicas or shards in the time of one
		std::map<int, std::vector<int>> lanes_of_device;
		for (int lane = 0; lane < n_lanes; ++lane)
			lanes_of_device[(params.device + lane) % std::max(1, ndev)].push_back(lane);
		if (lanes_of_device.size() == 1) {
			for (int lane = 0; lane < n_lanes; ++lane)
				build_lane(lane);
		} else {
			std::vector<std::thread> builders;
			for (const auto& dl : lanes_of_device)
				builders.emplace_back([&build_lane, &dl] {
					for (const int lane : dl.second)
						build_lane(lane);
				});
			for (std::thread& t : builders)
				t.join();
		}
		if (params.verbose) {
			if (params.k_list.size() > 1)
				appendf(log, "k = %d:\n", k);
			if (params.k_list.size() > 1)
				for (size_t e = 0; e < len.size(); ++e) // Arcs.cpp:877-882 prints one line per short end
					if ((int)len[e] < k)
						appendf(log, "Warning: ends of contig is shorter than k-value for contigID (no k-mers added): %zu\n", e + 1);
			appendf(log, "%s %u\n%s %u\n%s %u\n%s %u\n%s %u\n%s %u\n%s %u\n%s %u\n%s %u\n",
			        "Total number of contigs in draft genome: ", (unsigned)total, "Total valid contigs: ", (unsigned)valid,
			        "Total skipped contigs: ", (unsigned)skipped, "Total number of Kmers: ", (unsigned)st.total_kmers,
			        "Number Null Kmers: ", (unsigned)st.null_kmers, "Number Kmers Recorded: ", (unsigned)st.recorded,
			        "Number Kmer Collisions: ", (unsigned)st.collisions,
			        "Number Times Kmers Removed (since duplicate in different contig): ", (unsigned)st.removed_dup,
			        "Number of unique kmers (only one contig): ", (unsigned)st.unique);
		}
	}
	for (const auto& v : by_lane)
		idxs.insert(idxs.end(), v.begin(), v.end());
	(void)hipSetDevice(params.device % std::max(1, ndev));
	return idxs;
}

// ---- read mapping (replaces readChroms / chromiumRead, Arcs.cpp:1132-1370) -------------------------
// device buffers of one in-flight batch
struct DeviceSet
{
	DevArray<uint64_t> d_codes, d_woff, d_aoff;
	DevArray<uint8_t> d_ascii; // device-pack mode: the reads' bases as they are
	DevArray<uint32_t> d_nmask, d_len, d_bid;
	DevArray<uint8_t> d_class, d_ok, d_eval;
	std::vector<DevArray<int32_t>> d_conreci; // per k (a sharded exchange keeps every k's result until its round completes)
	DevArray<uint64_t> d_votes, d_votes2; // --index-shards only
	DevArray<int32_t> d_scratch;          // --index-shards -v: the counters' pass writes its per-shard results here
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;
	PackedBatch* inflight = nullptr;
};

// The GPU end of the ingest pipeline.  A LANE is one GPU of this process (--ranks with fewer files than ranks,
// --index-sharded: device, device + 1, ...; several lanes may share a device -- tests on a one-GPU box): its
// indexes (a replica per k, or its shard of every k's seed table), an IndexMap accumulator per k, per-file
// counters, and two device buffer sets on two streams, so that the copies of one batch overlap the kernels of the
// previous one.  Replicas: batch t goes to lane t mod L.  Sharded seed table (arks_exchange): L batches make a
// ROUND, one per lane; a round is submitted (upload, gate, seeds bucketed by owner) and the round before it completed
// (arks_exchange_complete_group: every lane's seeds answered by their owners, map kernels) -- two rounds in flight.
struct Lane
{
	int device = 0;
	std::vector<arks_index*> idxs;  // n_k x n_shards, the shards of one k together
	std::vector<arks_imap*> imaps;  // one per k
	std::vector<arks_exchange*> xs; // one per k (sharded seed table)
	DeviceSet sets[2];
	uint64_t* d_skipped = nullptr;     // [n_files]: skipped_invalidreadpair, counted on the device in device-pack mode
	uint64_t* d_stored = nullptr;      // [n_k][n_files]
	arks_map_stats* d_stats = nullptr; // [n_k][n_files]
	// --index-shards -v: d_stats takes the first part's counters (total_valid, bad, windows are the same in every
	// part), d_stats_rest the other parts' (found, recorded, dups add up: every key is in one part),
	// d_stats_votes reads_pass / reads_fail of the folded votes
	arks_map_stats* d_stats_rest = nullptr;
	arks_map_stats* d_stats_votes = nullptr;
};

struct Mapper
{
	std::vector<Lane> lanes;
	size_t n_shards, n_k;
	size_t turn = 0;
	size_t n_files;
	const bool sharded;
	std::vector<PackedBatch*> round; // sharded: the batches of the round being collected
	size_t rounds_submitted = 0, rounds_completed = 0;
	std::vector<size_t> global_file; // position of this rank's files in the command line (pair numbering)

	// idxs: lane-major, n_k x n_shards per lane
	Mapper(const std::vector<arks_index*>& is, int64_t imap_capacity, size_t nfiles, const std::vector<size_t>& mine)
	  : lanes((size_t)std::max(1, params.lanes))
	  , n_shards((size_t)std::max(1, params.index_shards))
	  , n_k(params.k_list.size())
	  , n_files(nfiles)
	  , sharded(params.index_sharded > 0)
	  , global_file(mine)
	{
		int ndev = 1;
		(void)hipGetDeviceCount(&ndev);
		const size_t per_lane = n_k * n_shards;
		for (size_t l = 0; l < lanes.size(); ++l) {
			Lane& ln = lanes[l];
			ln.device = (params.device + (int)l) % std::max(1, ndev);
			ln.idxs.assign(is.begin() + (long)(l * per_lane), is.begin() + (long)((l + 1) * per_lane));
			(void)hipSetDevice(ln.device);
			for (size_t ki = 0; ki < n_k; ++ki) {
				arks_imap* im = nullptr;
				const int rc = arks_imap_create(&im, imap_capacity, ln.device);
				if (rc != ARKS_OK)
					die_arks(rc, "creating the IndexMap accumulator");
				ln.imaps.push_back(im);
			}
			const size_t nc = n_k * nfiles;
			if (hipMalloc((void**)&ln.d_stored, nc * sizeof(uint64_t)) != hipSuccess ||
			    hipMalloc((void**)&ln.d_skipped, nfiles * sizeof(uint64_t)) != hipSuccess ||
			    hipMalloc((void**)&ln.d_stats, nc * sizeof(arks_map_stats)) != hipSuccess ||
			    hipMalloc((void**)&ln.d_stats_rest, nc * sizeof(arks_map_stats)) != hipSuccess ||
			    hipMalloc((void**)&ln.d_stats_votes, nc * sizeof(arks_map_stats)) != hipSuccess) {
				std::cerr << PROGRAM ": out of device memory\n";
				exit(EXIT_FAILURE);
			}
			(void)hipMemset(ln.d_stored, 0, nc * sizeof(uint64_t));
			(void)hipMemset(ln.d_skipped, 0, nfiles * sizeof(uint64_t));
			(void)hipMemset(ln.d_stats, 0, nc * sizeof(arks_map_stats));
			(void)hipMemset(ln.d_stats_rest, 0, nc * sizeof(arks_map_stats));
			(void)hipMemset(ln.d_stats_votes, 0, nc * sizeof(arks_map_stats));
			for (auto& s : ln.sets) {
				s.d_conreci.resize(n_k);
				if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
				    hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
					std::cerr << PROGRAM ": cannot create a HIP stream\n";
					exit(EXIT_FAILURE);
				}
			}
		}
		if (sharded) {
			// one exchange group per k: lane l is rank l of each
			for (size_t ki = 0; ki < n_k; ++ki) {
				std::vector<arks_exchange*> xs(lanes.size(), nullptr);
				std::vector<const arks_index*> shards;
				for (Lane& ln : lanes)
					shards.push_back(ln.idxs[ki]);
				const int rc = arks_exchange_create_local(xs.data(), shards.data(), (int)lanes.size());
				if (rc != ARKS_OK)
					die_arks(rc, "setting up the exchange of the sharded seed table");
				for (size_t l = 0; l < lanes.size(); ++l)
					lanes[l].xs.push_back(xs[l]);
			}
		}
		(void)hipSetDevice(lanes[0].device);
	}

	~Mapper()
	{
		for (Lane& ln : lanes) {
			(void)hipSetDevice(ln.device);
			for (arks_exchange* x : ln.xs)
				arks_exchange_free(x);
			for (auto& s : ln.sets) {
				if (s.done)
					(void)hipEventDestroy(s.done);
				if (s.stream)
					(void)hipStreamDestroy(s.stream);
			}
			(void)hipFree(ln.d_stored);
			(void)hipFree(ln.d_skipped);
			(void)hipFree(ln.d_stats);
			(void)hipFree(ln.d_stats_rest);
			(void)hipFree(ln.d_stats_votes);
		}
		(void)hipSetDevice(lanes[0].device);
	}

	// waits until the set's previous batch is through and gives its host buffers back
	void retire(DeviceSet& s, IngestPipeline& pipe)
	{
		if (!s.inflight)
			return;
		if (hipEventSynchronize(s.done) != hipSuccess) {
			std::cerr << PROGRAM ": device error while mapping\n";
			exit(EXIT_FAILURE);
		}
		pipe.recycle(s.inflight);
		s.inflight = nullptr;
	}

	// the batch on its way to the device (set s of lane ln) and through the gate
	int upload(Lane& ln, DeviceSet& s, PackedBatch* pb)
	{
		const int64_t n = pb->n_reads, np = pb->n_pairs;
		const size_t words = pb->words;
		s.d_codes.reserve(words);
		s.d_nmask.reserve(words);
		s.d_woff.reserve((size_t)n + 1);
		s.d_len.reserve((size_t)n);
		s.d_class.reserve((size_t)n);
		s.d_eval.reserve((size_t)n);
		for (auto& c : s.d_conreci)
			c.reserve((size_t)n);
		s.d_ok.reserve((size_t)np);
		s.d_bid.reserve((size_t)np);
		auto up = [&](void* d, const void* h, size_t bytes) {
			if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s.stream) != hipSuccess) {
				std::cerr << PROGRAM ": host to device copy failed\n";
				exit(EXIT_FAILURE);
			}
		};
		int rc = ARKS_OK;
		up(s.d_woff.p, pb->woff, ((size_t)n + 1) * sizeof(uint64_t));
		up(s.d_len.p, pb->len, (size_t)n * sizeof(uint32_t));
		up(s.d_ok.p, pb->pair_ok, (size_t)np);
		up(s.d_bid.p, pb->barcode_id, (size_t)np * sizeof(uint32_t));
		if (pb->ascii) {
			// device-pack mode: the bases come as text (8 bits each over the link instead of 3) and the 2-bit
			// packing, the N mask and checkReadSequence run here -- the host threads only gather them
			s.d_ascii.reserve(pb->ascii_bytes);
			s.d_aoff.reserve((size_t)n + 1);
			up(s.d_ascii.p, pb->ascii, pb->ascii_bytes);
			up(s.d_aoff.p, pb->aoff, ((size_t)n + 1) * sizeof(uint64_t));
			if (hipMemsetAsync(s.d_codes.p, 0, words * sizeof(uint64_t), s.stream) != hipSuccess ||
			    hipMemsetAsync(s.d_nmask.p, 0, words * sizeof(uint32_t), s.stream) != hipSuccess)
				return ARKS_ERR_HIP;
			rc = arks_pack_reads_device(s.d_ascii.p, s.d_aoff.p, s.d_len.p, s.d_woff.p, n, s.d_codes.p, s.d_nmask.p, s.d_class.p,
			                            ln.device, s.stream);
		} else {
			up(s.d_codes.p, pb->codes, words * sizeof(uint64_t));
			up(s.d_nmask.p, pb->nmask, words * sizeof(uint32_t));
			up(s.d_class.p, pb->cls, (size_t)n);
		}
		// (the copies above and the kernels below overlap the other set's: an index keeps one set of work
		// queues per stream, and the IndexMap accumulator is updated with atomics)
		if (rc == ARKS_OK)
			rc = arks_pair_gate_device(s.d_ok.p, s.d_class.p, np, s.d_eval.p, ln.device, s.stream);
		if (rc == ARKS_OK && pb->ascii)
			rc = arks_gate_count_device(s.d_ok.p, s.d_eval.p, np, ln.d_skipped + pb->file, ln.device, s.stream);
		return rc;
	}

	// pair rule + IndexMap update of k number ki for the batch of set s
	int pairs(Lane& ln, DeviceSet& s, PackedBatch* pb, size_t ki)
	{
		const size_t slot = ki * n_files + (size_t)pb->file;
		// pairs are numbered in input order (file, batch, pair): the order in which a single-threaded
		// chromiumRead creates the IndexMap's barcodes (Arcs.cpp:1282-1285)
		arks_imap_set_pair_base(ln.imaps[ki], ((uint64_t)global_file[(size_t)pb->file] << 48) | ((uint64_t)pb->seq << 24));
		return arks_pairs_device(s.d_conreci[ki].p, s.d_ok.p, s.d_bid.p, pb->n_pairs, nullptr, ln.imaps[ki], ln.d_stored + slot,
		                         ln.device, s.stream);
	}

	int submit(PackedBatch* pb, IngestPipeline& pipe)
	{
		if (pb->n_reads == 0) {
			pipe.recycle(pb);
			return ARKS_OK;
		}
		if (sharded) {
			round.push_back(pb);
			return round.size() == lanes.size() ? submit_round(pipe) : ARKS_OK;
		}
		const int64_t np = pb->n_pairs;
		Lane& ln = lanes[turn % lanes.size()];
		DeviceSet& s = ln.sets[(turn / lanes.size()) & 1];
		turn++;
		(void)hipSetDevice(ln.device);
		retire(s, pipe);
		int rc = upload(ln, s, pb);
		if (n_shards > 1) {
			s.d_votes.reserve((size_t)pb->n_reads);
			s.d_votes2.reserve((size_t)pb->n_reads);
		}
		for (size_t ki = 0; ki < n_k && rc == ARKS_OK; ++ki) { // the batch is resident: every k maps it
			const size_t slot = ki * n_files + (size_t)pb->file;
			if (n_shards > 1) {
				// per shard the votes of bestContig's walk, folded with a maximum, then the j_index test
				for (size_t sh = 0; sh < n_shards && rc == ARKS_OK; ++sh) {
					rc = arks_map_votes_device(ln.idxs[ki * n_shards + sh], s.d_codes.p, s.d_nmask.p, s.d_woff.p, s.d_len.p,
					                           s.d_eval.p, 2 * np, sh ? s.d_votes2.p : s.d_votes.p, s.stream);
					if (rc == ARKS_OK && sh)
						rc = arks_votes_max_device(s.d_votes.p, s.d_votes2.p, 2 * np, ln.device, s.stream);
				}
				if (rc == ARKS_OK)
					rc = arks_votes_resolve_device(s.d_votes.p, s.d_len.p, 2 * np, params.k_list[ki], params.j_index,
					                               s.d_conreci[ki].p, ln.device, s.stream);
				if (params.verbose) {
					// the k-mer counters of the log (Arcs.cpp:1329-1340): a second pass per part with the counters'
					// kernels -- every key is in ONE part (its first holder, arks_index_build_shard), so found,
					// recorded and duplicates add up over the parts -- and the j_index test counted on the folded votes
					s.d_scratch.reserve((size_t)pb->n_reads);
					for (size_t sh = 0; sh < n_shards && rc == ARKS_OK; ++sh)
						rc = arks_map_reads_device(ln.idxs[ki * n_shards + sh], s.d_codes.p, s.d_nmask.p, s.d_woff.p, s.d_len.p,
						                           s.d_eval.p, 2 * np, params.j_index, s.d_scratch.p,
						                           (sh ? ln.d_stats_rest : ln.d_stats) + slot, s.stream);
					if (rc == ARKS_OK)
						rc = arks_votes_count_device(s.d_votes.p, s.d_len.p, s.d_eval.p, 2 * np, params.k_list[ki], params.j_index,
						                             ln.d_stats_votes + slot, ln.device, s.stream);
				}
			} else
				rc = arks_map_reads_device(ln.idxs[ki], s.d_codes.p, s.d_nmask.p, s.d_woff.p, s.d_len.p, s.d_eval.p, 2 * np,
				                           params.j_index, s.d_conreci[ki].p, params.verbose ? ln.d_stats + slot : nullptr, s.stream);
			if (rc == ARKS_OK)
				rc = pairs(ln, s, pb, ki);
		}
		if (rc != ARKS_OK)
			return rc;
		if (hipEventRecord(s.done, s.stream) != hipSuccess)
			return ARKS_ERR_HIP;
		s.inflight = pb;
		return ARKS_OK;
	}

	// ---- sharded seed table: a round = one batch per lane (the last round of the input may have fewer: the lanes
	//      without one submit empty batches -- the exchange is collective) --------------------------------------------
	int submit_round(IngestPipeline& pipe)
	{
		const size_t set = rounds_submitted & 1;
		int rc = ARKS_OK;
		for (size_t l = 0; l < lanes.size() && rc == ARKS_OK; ++l) {
			Lane& ln = lanes[l];
			DeviceSet& s = ln.sets[set];
			(void)hipSetDevice(ln.device);
			retire(s, pipe); // (the round before last, completed in the call before this one)
			PackedBatch* pb = l < round.size() ? round[l] : nullptr;
			if (pb)
				rc = upload(ln, s, pb);
			const int64_t n = pb ? 2 * pb->n_pairs : 0;
			for (size_t ki = 0; ki < n_k && rc == ARKS_OK; ++ki)
				rc = arks_exchange_submit(ln.xs[ki], s.d_codes.p, s.d_nmask.p, s.d_woff.p, s.d_len.p, s.d_eval.p, n, params.j_index,
				                          s.d_conreci[ki].p,
				                          pb && params.verbose ? ln.d_stats + (ki * n_files + (size_t)pb->file) : nullptr, s.stream);
			s.inflight = pb;
		}
		round.clear();
		rounds_submitted++;
		if (rc == ARKS_OK && rounds_submitted - rounds_completed == 2)
			rc = complete_round();
		return rc;
	}

	int complete_round()
	{
		const size_t set = rounds_completed & 1;
		rounds_completed++;
		int rc = ARKS_OK;
		std::vector<arks_exchange*> xs(lanes.size());
		for (size_t ki = 0; ki < n_k && rc == ARKS_OK; ++ki) {
			for (size_t l = 0; l < lanes.size(); ++l)
				xs[l] = lanes[l].xs[ki];
			rc = arks_exchange_complete_group(xs.data(), (int)lanes.size());
		}
		for (size_t l = 0; l < lanes.size() && rc == ARKS_OK; ++l) {
			Lane& ln = lanes[l];
			DeviceSet& s = ln.sets[set];
			(void)hipSetDevice(ln.device);
			if (s.inflight)
				for (size_t ki = 0; ki < n_k && rc == ARKS_OK; ++ki)
					rc = pairs(ln, s, s.inflight, ki);
			if (rc == ARKS_OK && hipEventRecord(s.done, s.stream) != hipSuccess)
				rc = ARKS_ERR_HIP;
		}
		return rc;
	}

	void drain(IngestPipeline& pipe)
	{
		if (sharded) {
			int rc = ARKS_OK;
			if (!round.empty())
				rc = submit_round(pipe);
			while (rc == ARKS_OK && rounds_completed < rounds_submitted)
				rc = complete_round();
			if (rc != ARKS_OK)
				die_arks(rc, "mapping a read batch (sharded seed table)");
		}
		for (Lane& ln : lanes) {
			(void)hipSetDevice(ln.device);
			for (auto& s : ln.sets)
				retire(s, pipe);
		}
		(void)hipSetDevice(lanes[0].device);
	}
};

// Pinned host memory for the packed batches.  Pinning is slow (every page is locked and mapped for the device)
// and the driver takes its calls one at a time, so the slabs are made once, by a thread that starts before the
// draft is read (prefill), handed out to the pipeline's buffers and kept for the next pass (several k, a
// second pass over the reads); they go back to the system at exit.
class PinnedPool
{
  public:
	~PinnedPool()
	{
		wait();
		for (const Slab& s : slabs_)
			(void)hipHostFree(s.p);
	}
	void prefill(unsigned n, size_t bytes)
	{
		filler_ = std::thread([this, n, bytes] {
			(void)hipSetDevice(params.device);
			for (unsigned i = 0; i < n; ++i) {
				void* p = nullptr;
				if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
					return;
				std::lock_guard<std::mutex> lk(m_);
				slabs_.push_back(Slab{ p, bytes, false });
				cv_.notify_all();
			}
		});
	}
	void* alloc(size_t n)
	{
		{
			std::unique_lock<std::mutex> lk(m_);
			for (Slab& s : slabs_)
				if (!s.used && s.bytes >= n) {
					s.used = true;
					return s.p;
				}
		}
		void* p = nullptr;
		if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) != hipSuccess)
			return nullptr;
		std::lock_guard<std::mutex> lk(m_);
		slabs_.push_back(Slab{ p, n, true });
		return p;
	}
	void release(void* p)
	{
		std::lock_guard<std::mutex> lk(m_);
		for (Slab& s : slabs_)
			if (s.p == p)
				s.used = false;
	}
	void wait()
	{
		if (filler_.joinable())
			filler_.join();
	}

  private:
	struct Slab
	{
		void* p;
		size_t bytes;
		bool used;
	};
	std::mutex m_;
	std::condition_variable cv_;
	std::vector<Slab> slabs_;
	std::thread filler_;
};
PinnedPool g_pinned;
// ARKS_DEVICE_PACK=1: the 2-bit packing and checkReadSequence of the reads on the device (the worker threads
// only gather the bases: a third less of their work per read pair, 2.3x the bytes over the link).  Off by
// default: on the GPU box it gained 13 % at -t 1 and nothing from -t 4 on, where the producers' line scan
// bounds the stage, and the pinned pool it needs is 2.4x the size.
const bool g_device_pack = getenv("ARKS_DEVICE_PACK") != nullptr;

// batches the consumer holds beyond the two of one lane: two per further lane, and the round being collected of a
// sharded exchange
unsigned
extra_batch_buffers()
{
	const unsigned L = (unsigned)std::max(1, params.lanes);
	return 2 * (L - 1) + (params.index_sharded > 0 ? L : 0);
}

// fused == true: no multiplicity file; the reads per barcode come back in the result (pre_counts) and
// `redo` is set when the input needs the exact two-pass flow instead
RankResult
map_files(
    const std::vector<std::string>& files, const std::vector<size_t>& mine, const std::vector<arks_index*>& idxs,
    const std::unordered_map<std::string, int>& mult, bool fused, std::string* open_error)
{
	const size_t nf = files.size();
	const bool timing = getenv("ARKS_TIMING") != nullptr;
	auto t_prev = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		const auto t = std::chrono::steady_clock::now();
		if (timing)
			std::cerr << "[timing]   read stage, " << what << ": "
			          << std::chrono::duration_cast<std::chrono::microseconds>(t - t_prev).count() / 1000.0 << " ms\n";
		t_prev = t;
	};
	RankResult res;
	res.files.resize(nf);
	const size_t nk = params.k_list.size();
	res.triples.resize(nk);
	res.first.resize(nk);
	std::vector<std::unique_ptr<SeqReader>> readers;
	for (const size_t f : mine) {
		// inflate threads for bgzip'ed files (bgzf.hpp): -t over the files that are read at the same time (a
		// quarter of the threads are producers, one file each; the inflaters of a file only run while it is read)
		const unsigned concurrent = (unsigned)std::min<size_t>(std::max<size_t>(mine.size(), 1), std::max(1u, params.threads / 4));
		readers.emplace_back(new SeqReader(files[f].c_str(), std::max(1u, params.threads / concurrent)));
		if (!readers.back()->ok()) {
			*open_error = files[f];
			return res;
		}
	}
	std::unique_ptr<BarcodeDict> dict(fused ? nullptr : new BarcodeDict(mult));
	// distinct (barcode, contig end) pairs: a few per barcode; a starting size only, the accumulator grows
	const int64_t imap_cap = fused ? (int64_t)1 << 20 : std::max<int64_t>(1 << 16, (int64_t)mult.size() * 4);
	const size_t nm = std::max<size_t>(mine.size(), 1);
	Mapper mapper(idxs, imap_cap, nm, mine);
	HostAllocator pinned;
	pinned.alloc = [](size_t n) { return g_pinned.alloc(n); };
	pinned.release = [](void* p) { g_pinned.release(p); };
	std::vector<SeqReader*> rdp;
	for (auto& r : readers)
		rdp.push_back(r.get());
	IngestPipeline pipe(rdp, dict.get(), params.batch_pairs, params.verbose != 0, params.threads, pinned);
	pipe.set_device_pack(g_device_pack);
	pipe.add_buffers(extra_batch_buffers());
	g_pinned.wait();
	lap("open files, barcode dictionary, device buffers, pinned pool ready");
	const int prc = pipe.run([&](PackedBatch* pb) {
		FileResult& fr = res.files[mine[(size_t)pb->file]];
		FileCounters& f = fr.fc;
		f.skipped_unpaired += pb->fc.skipped_unpaired, f.emptybarcode += pb->fc.emptybarcode,
		    f.invalidbarcode += pb->fc.invalidbarcode, f.gated += pb->fc.gated,
		    f.skipped_invalid += pb->fc.skipped_invalid;
		if (!pb->messages.empty())
			fr.messages[pb->seq].swap(pb->messages);
		pb->messages.clear();
		return mapper.submit(pb, pipe);
	}, [&] { mapper.drain(pipe); }); // the in-flight batches are retired before their pinned buffers go
	if (timing)
		std::cerr << "[timing]   read stage threads: " << pipe.producers() << " producers + " << pipe.packers() << " workers\n";
	if (prc != ARKS_OK)
		die_arks(prc, "mapping a read batch");
	if (hipDeviceSynchronize() != hipSuccess) {
		std::cerr << PROGRAM ": device error while mapping\n";
		exit(EXIT_FAILURE);
	}
	lap("pipeline (read, parse, pack, map)");
	for (size_t i = 0; i < mine.size(); ++i)
		if (readers[i]->failed()) // (the reference's gzread + kseq stop silently at the same place)
			std::cerr << PROGRAM ": warning: " << files[mine[i]]
			          << ": the compressed stream is damaged or truncated; the records before the damage were used\n";
	if (fused) {
		for (const PrepassInfo& pi : pipe.prepass())
			if (pi.zero_len || pi.untagged_at.size() > (1u << 22)) {
				res.redo = true; // rare input shapes: let the caller run the literal two passes
				for (Lane& ln : mapper.lanes)
					for (arks_imap* im : ln.imaps)
						arks_imap_free(im);
				return res;
			}
		DynamicDict& dyn = pipe.dynamic();
		res.names.reserve(dyn.size());
		for (size_t id = 0; id < dyn.size(); ++id)
			res.names.push_back(dyn.name((uint32_t)id));
		for (size_t i = 0; i < mine.size(); ++i) {
			const PrepassInfo& pi = pipe.prepass()[i];
			FileResult& fr = res.files[mine[i]];
			fr.pre_total = pi.total, fr.pre_lead = pi.lead, fr.untagged_at = pi.untagged_at;
			for (size_t id = 0; id < pi.counts.size(); ++id)
				if (pi.counts[id])
					fr.pre_counts.emplace_back((uint32_t)id, pi.counts[id]);
		}
	} else {
		res.names.reserve(dict->name.size());
		for (const std::string* s : dict->name)
			res.names.push_back(*s);
	}
	// counters: sums over the lanes; the IndexMap: the lanes' entries merged (same barcode ids: one dictionary)
	std::vector<uint64_t> stored(nk * nm, 0), skipped(nm, 0);
	std::vector<arks_map_stats> st(nk * nm);
	std::memset(st.data(), 0, st.size() * sizeof(arks_map_stats));
	for (Lane& ln : mapper.lanes) {
		(void)hipSetDevice(ln.device);
		std::vector<uint64_t> l_stored(nk * nm), l_skipped(nm, 0);
		std::vector<arks_map_stats> l_st(nk * nm);
		(void)hipMemcpy(l_skipped.data(), ln.d_skipped, nm * sizeof(uint64_t), hipMemcpyDeviceToHost);
		(void)hipMemcpy(l_stored.data(), ln.d_stored, nk * nm * sizeof(uint64_t), hipMemcpyDeviceToHost);
		(void)hipMemcpy(l_st.data(), ln.d_stats, nk * nm * sizeof(arks_map_stats), hipMemcpyDeviceToHost);
		if (params.index_shards > 1) {
			std::vector<arks_map_stats> rest(nk * nm), votes(nk * nm);
			(void)hipMemcpy(rest.data(), ln.d_stats_rest, nk * nm * sizeof(arks_map_stats), hipMemcpyDeviceToHost);
			(void)hipMemcpy(votes.data(), ln.d_stats_votes, nk * nm * sizeof(arks_map_stats), hipMemcpyDeviceToHost);
			for (size_t i = 0; i < nk * nm; ++i) {
				l_st[i].found += rest[i].found, l_st[i].recorded += rest[i].recorded, l_st[i].dups += rest[i].dups;
				l_st[i].reads_pass = votes[i].reads_pass, l_st[i].reads_fail = votes[i].reads_fail;
			}
		}
		for (size_t i = 0; i < nm; ++i)
			skipped[i] += l_skipped[i];
		for (size_t i = 0; i < nk * nm; ++i) {
			stored[i] += l_stored[i];
			const uint64_t* a = reinterpret_cast<const uint64_t*>(&l_st[i]);
			uint64_t* acc = reinterpret_cast<uint64_t*>(&st[i]);
			for (size_t w = 0; w < sizeof(arks_map_stats) / sizeof(uint64_t); ++w)
				acc[w] += a[w];
		}
	}
	for (size_t i = 0; i < mine.size(); ++i)
		res.files[mine[i]].fc.skipped_invalid += skipped[i]; // (device-pack mode; 0 otherwise)
	for (size_t i = 0; i < mine.size(); ++i) {
		FileResult& fr = res.files[mine[i]];
		fr.have = true;
		for (size_t ki = 0; ki < nk; ++ki) {
			fr.stored.push_back(stored[ki * nm + i]);
			fr.st.push_back(st[ki * nm + i]);
		}
	}
	for (size_t ki = 0; ki < nk; ++ki) {
		std::vector<std::vector<uint32_t>> lt(mapper.lanes.size());
		std::vector<std::vector<uint64_t>> lf(mapper.lanes.size());
		for (size_t l = 0; l < mapper.lanes.size(); ++l) {
			Lane& ln = mapper.lanes[l];
			(void)hipSetDevice(ln.device);
			const int64_t n = arks_imap_size(ln.imaps[ki]);
			if (n < 0)
				die_arks((int)-n, "reading the IndexMap accumulator");
			lt[l].resize((size_t)n * 3 + 3);
			lf[l].resize((size_t)n + 1);
			const int rc = arks_imap_export_ordered(ln.imaps[ki], lt[l].data(), lf[l].data());
			if (rc != ARKS_OK)
				die_arks(rc, "exporting the IndexMap");
			lt[l].resize((size_t)n * 3);
			lf[l].resize((size_t)n);
			arks_imap_free(ln.imaps[ki]);
		}
		if (mapper.lanes.size() == 1) {
			res.triples[ki].swap(lt[0]);
			res.first[ki].swap(lf[0]);
		} else
			merge_lane_entries(lt, lf, res.triples[ki], res.first[ki]);
	}
	(void)hipSetDevice(mapper.lanes[0].device);
	lap("counters and IndexMap back to the host");
	return res;
}

// The read stage of this process and, with --ranks N, of its N - 1 worker processes (g_workers: forked
// before the first HIP call, one GPU each, they ran the stages up to here on their own and now map the
// files dealt to them -- file f goes to rank f mod N -- and send their results through their pipes).
struct Worker
{
	pid_t pid;
	int fd;
};
std::vector<Worker> g_workers;
int g_rank = 0, g_world = 1, g_result_fd = -1;

void
read_stage(
    const std::vector<std::string>& files, const std::vector<arks_index*>& idxs, std::vector<IndexMap>& imaps,
    std::unordered_map<std::string, int>& mult, const std::vector<CI>& contigRecord, bool fused, std::string& out,
    std::string& err, std::string& pre_out, std::string& pre_err, bool& redo, std::vector<CompactIndex>* compact)
{
	// after a fall-back to two passes the workers are gone: rank 0 maps every file
	const int world = g_workers.empty() && g_rank == 0 ? 1 : g_world;
	std::vector<size_t> mine;
	for (size_t f = 0; f < files.size(); ++f)
		if ((int)(f % (size_t)world) == g_rank)
			mine.push_back(f);
	std::string open_error;
	std::vector<RankResult> ranks(1);
	ranks[0] = map_files(files, mine, idxs, mult, fused, &open_error);
	if (!open_error.empty()) {
		// the reference stops at the first file it cannot open, after the log of the files before it
		// (Arcs.cpp:1158-1163); the workers' share is dropped
		for (const Worker& wk : g_workers)
			(void)::kill(wk.pid, SIGTERM);
		std::cout << out;
		if (params.verbose)
			std::cout << "Reading chrom " << open_error << std::endl;
		std::cerr << "File " << open_error << " cannot be opened." << std::endl;
		exit(1);
	}
	if (g_rank != 0) { // a worker: hand the results to rank 0 and leave
		send_result(g_result_fd, ranks[0]);
		::close(g_result_fd);
		std::fflush(nullptr);
		_exit(EXIT_SUCCESS);
	}
	for (const Worker& wk : g_workers) {
		ranks.emplace_back();
		const bool ok = receive_result(wk.fd, ranks.back());
		::close(wk.fd);
		int status = 0;
		(void)::waitpid(wk.pid, &status, 0);
		if (!ok || !WIFEXITED(status) || WEXITSTATUS(status) != 0) {
			std::cerr << PROGRAM ": a worker rank failed (pid " << wk.pid << ")\n";
			exit(EXIT_FAILURE);
		}
	}
	g_workers.clear();
	redo = false;
	for (const RankResult& r : ranks)
		redo = redo || r.redo;
	if (redo)
		return;
	merge_results(files, ranks, imaps, mult, contigRecord, fused, MergeParams{ params.verbose != 0, params.k_list, params.index_shards, params.threads }, out, err, &pre_out, &pre_err, compact);
}

// file names of one k: with a single -k exactly the reference's (Arcs.cpp:2144-2157); with a list
// every name carries its k (the default base name does already, explicit names get a _k<k> suffix)
struct OutputNames
{
	std::string base, dist, tsv;
};

std::string
with_k(const std::string& name, int k)
{
	const std::string tag = "_k" + std::to_string(k);
	const size_t dot = name.rfind('.');
	const size_t slash = name.rfind('/');
	if (dot == std::string::npos || (slash != std::string::npos && dot < slash))
		return name + tag;
	return name.substr(0, dot) + tag + name.substr(dot);
}

OutputNames
output_names(int k)
{
	const bool multi = params.k_list.size() > 1;
	OutputNames n;
	if (params.base_name.empty()) {
		std::ostringstream fn;
		fn << params.file << ".scaff"
		   << "_arks"
		   << "_c" << params.g.min_reads << "_k" << k << "_j" << params.j_index << "_l" << params.g.min_links << "_d"
		   << params.g.max_degree << "_e" << params.end_length << "_r" << params.g.error_percent;
		n.base = fn.str();
	} else
		n.base = multi ? params.base_name + "_k" + std::to_string(k) : params.base_name;
	n.dist = params.dist_graph_name.empty() ? n.base + ".dist.gv" : (multi ? with_k(params.dist_graph_name, k) : params.dist_graph_name);
	n.tsv = params.tsv_name.empty() ? n.base + "_main.tsv" : (multi ? with_k(params.tsv_name, k) : params.tsv_name);
	return n;
}

void
run_arks(const std::vector<std::string>& filenames)
{
	std::cout << "Running: " << PROGRAM << " " << PACKAGE_VERSION << "\nARKS method\n pid " << ::getpid()
	          << "\n -c " << params.g.min_reads << "\n -d " << params.g.max_degree << "\n -e " << params.end_length
	          << "\n -l " << params.g.min_links << "\n -m " << params.g.min_mult << '-' << params.g.max_mult
	          << "\n -r " << params.g.error_percent << "\n -v " << params.verbose << "\n -z " << params.min_size
	          << "\n --gap=" << params.g.gap << "\n -k " << params.k_arg << "\n -j " << params.j_index << "\n -t "
	          << params.threads << "\n -b " << maybe_na(params.base_name) << "\n -g "
	          << maybe_na(params.dist_graph_name) << "\n --barcode-counts=" << maybe_na(params.barcode_counts_name)
	          << "\n --tsv=" << maybe_na(params.tsv_name) << "\n -a " << maybe_na(params.fofName) << "\n -f "
	          << maybe_na(params.file) << "\n -u " << maybe_na(params.multfile) << '\n';
	for (const auto& f : filenames)
		std::cout << ' ' << f << '\n';
	std::cout.flush();

	// --ranks N: N - 1 worker processes, forked here -- before the first HIP call of this process -- each
	// with a GPU of its own (device + rank, modulo the devices there are, so that a test can run several
	// ranks on one GPU).  A worker runs the same stages silently up to the read stage, maps the files dealt
	// to it and sends its results to this process (read_stage); everything after that is rank 0's.
	// (a file that cannot be opened: no workers -- one process then writes the reference's log of the files in
	// front of it and stops at it, Arcs.cpp:1158-1163, instead of a worker dying with its own message after
	// rank 0 has mapped its whole share)
	// With fewer files than ranks -- the pipeline's one reads.fq.gz (bin/arcs-make:290), /dev/stdin (:305) -- a process
	// drives several GPUs: the batches of its files are dealt to its LANES (Mapper).  --index-sharded: one process,
	// every rank a lane, the seed table hash-sharded over them instead of replicated.
	bool all_open = true;
	if (params.index_sharded != 0) {
		if (params.index_sharded < 0) { // no number given: --ranks, else every visible GPU
			int ndev = 1;
			params.index_sharded = params.ranks > 1 ? params.ranks : (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0 ? ndev : 1);
		}
		params.ranks = 1;
		params.lanes = params.index_sharded;
	}
	if (params.ranks > 1 && filenames.size() > 1)
		for (const std::string& f : filenames) {
			// by stat / access, and only for regular files: opening and closing a FIFO here would block until its
			// writer attaches and then take the only reader away from it (SIGPIPE before the read stage opens it)
			struct stat sb;
			if (::stat(f.c_str(), &sb) != 0)
				all_open = false;
			else if (S_ISREG(sb.st_mode) && ::access(f.c_str(), R_OK) != 0)
				all_open = false;
		}
	if (params.ranks > 1 && filenames.size() > 1 && all_open) {
		g_world = (int)std::min<size_t>((size_t)params.ranks, filenames.size());
		std::fflush(nullptr);
		for (int r = 1; r < g_world; ++r) {
			int fds[2];
			if (::pipe(fds) != 0) {
				std::cerr << PROGRAM ": cannot create a pipe\n";
				exit(EXIT_FAILURE);
			}
			const pid_t pid = ::fork();
			if (pid < 0) {
				std::cerr << PROGRAM ": cannot fork a worker rank\n";
				exit(EXIT_FAILURE);
			}
			if (pid == 0) {
				::close(fds[0]);
				for (const Worker& wk : g_workers)
					::close(wk.fd);
				g_workers.clear();
				g_rank = r;
				g_result_fd = fds[1];
				if (!std::freopen("/dev/null", "w", stdout))
					_exit(EXIT_FAILURE);
				break;
			}
			::close(fds[1]);
			g_workers.push_back(Worker{ pid, fds[0] });
		}
	}
	if (arks_abi_version() != ARKS_ABI_VERSION) {
		// (arks_abi_version makes no HIP call: safe in front of the forks' first device use)
		std::cerr << PROGRAM ": error: libarks_hip reports ABI version " << arks_abi_version() << ", this program was built against "
		          << ARKS_ABI_VERSION << (arks_abi_version() < 0 ? " (a calibration build of the library: results wrong by design)" : "") << ".\n";
		exit(EXIT_FAILURE);
	}
	if (arks_device_count() < 1) {
		std::cerr << PROGRAM ": error: no gfx950 (MI355X) device is visible; this build has no CPU path.\n";
		exit(EXIT_FAILURE);
	}
	{
		// the GPU ranks of the run dealt to the processes: rank r of g_world gets ranks / g_world of them (the first
		// ranks % g_world one more) as its lanes, on the devices device + first lane ... (modulo the devices there are,
		// so that a test can run several ranks on one GPU)
		int first_lane = 0;
		if (params.index_sharded == 0) {
			const int n = std::max(1, params.ranks), p = std::max(1, g_world);
			params.lanes = n / p + (g_rank < n % p ? 1 : 0);
			first_lane = g_rank * (n / p) + std::min(g_rank, n % p);
		}
		int ndev = 0;
		if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
			// Which device every GPU rank of this process got, said out loud (stderr: stdout is the reference's log), and no
			// silent doubling up: N ranks on fewer than N devices map nothing faster than the devices alone would, and a run
			// that was meant for eight GPUs and found one should say so -- unless it was asked for (--share-devices: tests
			// on a one-GPU box).
			const int asked = params.index_sharded > 0 ? params.index_sharded : std::max(1, params.ranks);
			if (asked > ndev && !params.share_devices) {
				if (g_rank == 0)
					std::cerr << PROGRAM ": error: " << asked << " GPU ranks asked for ("
					          << (params.index_sharded > 0 ? "--index-sharded" : "--ranks") << "), " << ndev
					          << " gfx950 device(s) visible; ranks that share a device add nothing (--share-devices to allow it).\n";
				exit(EXIT_FAILURE);
			}
			const int base = params.device;
			params.device = (base + first_lane) % ndev;
			(void)hipSetDevice(params.device);
			if (asked > 1)
				for (int l = 0; l < std::max(1, params.lanes); ++l) {
					const int dev = (params.device + l) % ndev;
					hipDeviceProp_t pr;
					const bool named = hipGetDeviceProperties(&pr, dev) == hipSuccess;
					std::fprintf(stderr, "%s: GPU rank %d (process %d, lane %d) -> device %d%s%s%s\n", PROGRAM, first_lane + l, g_rank, l, dev,
					             named ? " (" : "", named ? pr.name : "", named ? ")" : "");
				}
		}
	}
	{
		// the pinned buffers of the read stage, made while the draft is read and indexed
		size_t n_mine = 0, words = 0, reads = 0;
		for (size_t f = 0; f < filenames.size(); ++f)
			n_mine += (int)(f % (size_t)g_world) == g_rank;
		packed_estimate(params.batch_pairs, &words, &reads);
		size_t slab = packed_slab_bytes(words + words / 8, reads + reads / 8);
		if (g_device_pack) {
			size_t bases = 0;
			raw_estimate(params.batch_pairs, &bases, &reads);
			slab = raw_slab_bytes(bases + bases / 8 + 64, reads + reads / 8 + 64);
		}
		g_pinned.prefill(IngestPipeline::buffers_for(params.threads, (unsigned)n_mine) + extra_batch_buffers(), slab);
	}

	std::vector<IndexMap> imaps;
	// the stages behind the read stage run on numbers (graph_fast.hpp) unless the distance estimates are asked
	// for, which walk the IndexMap itself (ARKS_LITERAL_GRAPH=1: graph.hpp's containers in any case)
	const bool fast_graph = !params.dist_est && getenv("ARKS_LITERAL_GRAPH") == nullptr;
	std::vector<CompactIndex> cix;
	std::unordered_map<std::string, int> mult;
	ContigToLength contigToLength;
	std::vector<CI> contigRecord;

	// ARKS_TIMING=1: wall time of each stage on stderr (this build only)
	const bool timing = getenv("ARKS_TIMING") != nullptr;
	auto t_prev = std::chrono::steady_clock::now();
	auto lap = [&](const char* what) {
		const auto t = std::chrono::steady_clock::now();
		if (timing)
			std::cerr << "[timing] " << what << ": "
			          << std::chrono::duration_cast<std::chrono::milliseconds>(t - t_prev).count() << " ms\n";
		t_prev = t;
	};
	std::cout << "\n=>Preprocessing: Gathering barcode multiplicity information..." << now();
	// Without -u the reference makes a pass of its own over the reads to count reads per barcode
	// (Arcs.cpp:1881-1888).  Here that counting rides along with the mapping pass ("fused"); the
	// text of the stages is emitted in the reference's order afterwards.  ARKS_TWO_PASS=1 forces
	// the literal flow, which is also the fallback for inputs the fused pass cannot reproduce.
	const bool fused = params.multfile.empty() && getenv("ARKS_TWO_PASS") == nullptr;
	if (!params.multfile.empty())
		create_index_mult_map(params.multfile, mult);
	else {
		std::cout << "Multiplicity information is being formed from reads as no barcode multiplicity file provided."
		          << std::endl;
		if (!fused)
			read_barcodes(filenames, mult);
	}
	lap("barcode multiplicities");
	std::string mid; // stdout of the stages between the barcode pass and the read stage
	mid += std::string("\n=>Preprocessing: Gathering draft information...") + now() + "\n";
	mid += std::string("\n=>Storing Kmers from Contig ends... ") + now() + "\n";
	const std::vector<arks_index*> idxs = build_contig_index(contigRecord, contigToLength, mid);
	lap("contig index (read draft + device build)");
	mid += std::string("\n=>Reading Chromium FASTQ file(s)... ") + now() + "\n";
	if (!fused) {
		std::cout << mid << std::flush;
		mid.clear();
	}
	{
		std::string out, err, pre_out, pre_err;
		bool redo = false;
		read_stage(filenames, idxs, imaps, mult, contigRecord, fused, out, err, pre_out, pre_err, redo, fast_graph ? &cix : nullptr);
		if (fused && redo) {
			// (a rank met an input the fused pass cannot reproduce: the literal two passes, in this process)
			mult.clear();
			imaps.clear();
			cix.clear();
			out.clear();
			err.clear();
			pre_out.clear();
			pre_err.clear();
			read_barcodes(filenames, mult);
			std::cout << mid << std::flush;
			read_stage(filenames, idxs, imaps, mult, contigRecord, false, out, err, pre_out, pre_err, redo, fast_graph ? &cix : nullptr);
		} else if (fused) {
			std::cout << pre_out;
			std::cerr << pre_err;
			std::cout << mid;
		}
		std::cout << out << std::flush;
		std::cerr << err << std::flush;
	}
	lap("read files -> IndexMap (ingest pipeline + GPU mapping)");
	{
		std::vector<arks_index*> uniq(idxs);
		std::sort(uniq.begin(), uniq.end());
		uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
		for (arks_index* idx : uniq) // (replica lanes on one device share an index)
			arks_index_free(idx);
	}
	std::cout << "Cumulative memory usage: " << memory_usage() << std::endl;

	for (size_t ki = 0; ki < params.k_list.size(); ++ki) {
		const OutputNames names = output_names(params.k_list[ki]);
		IndexMap literal_none;
		IndexMap& imap = fast_graph ? literal_none : imaps[ki];
		PairMap pmap;
		CompactPairs cpairs;
		ScaffoldGraph g;
		if (params.k_list.size() > 1)
			std::cout << "\n=> Graph stage for k = " << params.k_list[ki] << "\n";
		std::cout << "\n=> Pairing scaffolds... " << now();
		if (fast_graph)
			cpairs = pair_contigs_compact(cix[ki], params.g, params.threads);
		else
			pair_contigs(imap, pmap, mult, params.g);
		if (params.output_pair) {
			std::cout << "\n=> Outputting Pairing information... " << now();
			std::ofstream out((names.base + "_pair.tsv").c_str());
			if (fast_graph)
				write_pair_map_compact(out, cix[ki], cpairs);
			else
				write_pair_map(out, pmap);
		}
		const std::string t_graph = now(); // the reference reuses this time stamp for the next heading (Arcs.cpp:1917-1924)
		std::cout << "\n=> Creating the graph... " << t_graph;
		if (fast_graph)
			create_graph_compact(cpairs, cix[ki], g, params.g);
		else
			create_graph(pmap, g, params.g);
		if (params.dist_est) { // calcDistanceEstimates, Arcs.cpp:1767-1808
			const bool multi = params.k_list.size() > 1;
			std::cout << "\n=> Calculating distance estimates... " << t_graph;
			std::cout << "\n\t=> Measuring intra-contig distances / shared barcodes... " << now();
			DistSampleMap samples;
			calc_dist_samples(imap, contigToLength, mult, params.g, samples);
			std::cout << "\n\t=> Writing intra-contig distance samples to TSV... " << now();
			if (!params.dist_samples_tsv.empty()) {
				std::ofstream f((multi ? with_k(params.dist_samples_tsv, params.k_list[ki]) : params.dist_samples_tsv).c_str());
				write_dist_samples_tsv(f, samples);
			}
			std::cout << "\n\t=> Building Jaccard to distance map... " << now();
			JaccardToDist j2d;
			build_jaccard_to_dist(samples, j2d);
			std::cout << "\n\t=> Calculating barcode stats for scaffold pairs... " << now();
			PairToBarcodeStats pair_stats;
			build_pair_to_barcode_stats(imap, mult, contigToLength, params.g, pair_stats);
			std::cout << "\n\t=> Adding edge distances... " << now();
			add_edge_distances(pair_stats, j2d, params.g, g);
			if (!params.dist_tsv.empty()) {
				std::cout << "\n\t=> Writing distance estimates to TSV... " << now();
				std::ofstream f((multi ? with_k(params.dist_tsv, params.k_list[ki]) : params.dist_tsv).c_str());
				write_dist_tsv(f, pair_stats, g);
			}
		}
		std::cout << "\n=> Writing graph file... " << now() << "\n";
		const std::string graph_file = names.base + "_original.gv";
		if (params.g.max_degree != 0) {
			std::cout << "      Deleting nodes with degree > " << params.g.max_degree << "... \n";
			remove_degree_nodes(g, params.g.max_degree);
		} else
			std::cout << "      Max Degree (-d) set to: " << params.g.max_degree
			          << ". Will not delete any vertices from graph.\n";
		std::cout << "      Writing graph file to " << graph_file << "...\n";
		{
			std::ofstream out(graph_file.c_str());
			write_graph(out, g);
		}
		std::cout << "\n=> Creating the ABySS graph... " << now();
		std::cout << "\n=> Writing the ABySS graph file... " << now() << "\n";
		{
			std::ofstream out(names.dist.c_str());
			if (!out.good()) {
				std::cerr << "error: `" << names.dist << "': " << strerror(errno) << std::endl;
				exit(EXIT_FAILURE);
			}
			std::string err;
			if (!write_dist_graph(out, contigToLength, g, params.g.gap, &err, params.dist_est, params.g.dist_upper)) {
				std::cerr << err << std::endl;
				exit(EXIT_FAILURE);
			}
		}
		if (!names.tsv.empty()) {
			const size_t barcode_count = fast_graph ? count_barcodes_compact(cix[ki], mult, params.g) : count_barcodes(imap, mult, params.g);
			std::cout << "\n=> Writing TSV file... " << now();
			std::ofstream f(names.tsv.c_str());
			if (fast_graph)
				write_tsv_compact(f, cix[ki], cpairs, barcode_count, params.g, params.threads);
			else
				write_tsv(f, imap, pmap, barcode_count, params.g);
		}
	}
	if (!params.barcode_counts_name.empty()) {
		std::cout << "\n=> Writing reads per barcode TSV file... " << now();
		if (params.barcode_counts_name.find(".tsv") == std::string::npos)
			params.barcode_counts_name += ".tsv";
		std::ofstream f(params.barcode_counts_name.c_str());
		write_barcode_counts(f, mult);
	}
	std::cout << "\n=> Done.\n" << now();
	lap("graph stage and output files");
}

} // namespace

int
main(int argc, char** argv)
{
	printf("Reading user inputs...\n");
	bool arcsOnly = false, arksOnly = false, die = false;
	for (int c; (c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1;) {
		std::istringstream arg(optarg != NULL ? optarg : "");
		switch (c) {
		case 'u': arg >> params.multfile; break;
		case 'k': arg >> params.k_arg; arksOnly = true; break;
		case 'j': arg >> params.j_index; arksOnly = true; break;
		case 't': arg >> params.threads; arksOnly = true; break;
		case '?': die = true; break;
		case 'f': arg >> params.file; break;
		case 'a': arg >> params.fofName; break;
		case 'B': arg >> params.dist_bin_size; break;
		case 's': arg >> params.seq_id; arcsOnly = true; break;
		case 'c': arg >> params.g.min_reads; break;
		case 'P': params.output_pair = true; break;
		case 'D': params.dist_est = true; break;
		case 'l': arg >> params.g.min_links; break;
		case 'z': arg >> params.min_size; break;
		case 'b': arg >> params.base_name; break;
		case 'g': arg >> params.dist_graph_name; break;
		case OPT_TSV: arg >> params.tsv_name; break;
		case OPT_GAP: arg >> params.g.gap; break;
		case OPT_BARCODE_COUNTS: arg >> params.barcode_counts_name; break;
		case OPT_SAMPLES_TSV: arg >> params.dist_samples_tsv; break;
		case OPT_DIST_TSV: arg >> params.dist_tsv; break;
		case OPT_NO_DIST_EST: params.dist_est = false; break;
		case OPT_DIST_MEDIAN: params.g.dist_upper = false; break;
		case OPT_DIST_UPPER: params.g.dist_upper = true; break;
		case OPT_ARKS_METHOD: params.arks = true; break;
		case OPT_BATCH_PAIRS:
			arg >> params.batch_pairs;
			params.batch_pairs = std::min(params.batch_pairs, (1L << 24) - 1); // pair numbering: 24 bits per batch
			break;
		case OPT_DEVICE: arg >> params.device; break;
		case OPT_INDEX_SHARDS: arg >> params.index_shards; break;
		case OPT_RANKS: arg >> params.ranks; break;
		case OPT_SHARE_DEVICES: params.share_devices = true; break;
		case OPT_INDEX_SHARDED:
			params.index_sharded = -1; // (resolved in run_arks)
			if (optarg != NULL)
				arg >> params.index_sharded;
			break;
		case 'm': {
			std::string first, second;
			std::getline(arg, first, '-');
			std::getline(arg, second);
			std::stringstream ss;
			ss << first << "\t" << second;
			ss >> params.g.min_mult >> params.g.max_mult;
		} break;
		case 'd': arg >> params.g.max_degree; break;
		case 'e': arg >> params.end_length; break;
		case 'r': arg >> params.g.error_percent; break;
		case 'v': ++params.verbose; break;
		case OPT_HELP: std::cout << USAGE; exit(EXIT_SUCCESS);
		case OPT_VERSION: std::cout << PROGRAM " " PACKAGE_VERSION "\n"; exit(EXIT_SUCCESS);
		}
		if (optarg != NULL && (!arg.eof() || arg.fail())) {
			std::cerr << PROGRAM ": invalid option: `-" << (char)c << optarg << "'\n";
			exit(EXIT_FAILURE);
		}
	}
	if ((params.arks && arcsOnly) || (!params.arks && arksOnly)) {
		std::cerr << PROGRAM ": error: You specified an option that does not match with method "
		                     "choosen.\nCheck --help for method specific options.\n";
		die = true;
	}
	if (!params.arks) {
		std::cerr << PROGRAM ": error: this build provides the ARKS method only (--arks); the alignment "
		                     "(SAM/BAM) method is not part of it.\n";
		die = true;
	}
	std::vector<std::string> filenames(argv + optind, argv + argc);
	if (params.fofName.empty() && filenames.empty()) {
		std::cerr << PROGRAM ": error: specify input (chromium reads) or a list of files with -a option\n";
		die = true;
	}
	bool stdIn = false;
	if (!filenames.empty())
		stdIn = filenames[0] == "/dev/stdin";
	if (!params.file.empty())
		assert_readable(params.file);
	if (!params.fofName.empty())
		assert_readable(params.fofName);
	for (const auto& f : filenames)
		assert_readable(f);
	const std::vector<std::string> fof = read_fof(params.fofName);
	filenames.insert(filenames.end(), fof.begin(), fof.end());
	bool alignment = false;
	if (!stdIn && !check_same_format(filenames, alignment)) {
		std::cerr << "Input files must be all alignment or all read files." << params.file << ". Exiting... \n";
		die = true;
	}
	if (!stdIn && !(alignment ^ params.arks)) {
		std::cerr << "File type must be compatible with the method. (BAM/SAM for ARCS) or (Read file for ARKS "
		             "(--arks)). Exiting... \n";
		die = true;
	}
	{
		std::ifstream g(params.file.c_str());
		if (!g.good() && params.arks) {
			std::cerr << "Cannot find [-f] scaffold file which is required for --arks" << params.file
			          << ". Exiting... \n";
			die = true;
		}
	}
	params.g.end_length = params.end_length;
	params.g.dist_bin_size = params.dist_bin_size;
	params.g.dist_est = params.dist_est;
	if (params.index_shards < 1 || params.index_shards > 4096) {
		std::cerr << PROGRAM ": --index-shards must be between 1 and 4096\n";
		die = true;
	}
	if (params.index_sharded != 0 && (params.index_sharded < -1 || params.index_sharded > 64 || params.index_shards > 1)) {
		std::cerr << PROGRAM ": --index-sharded takes 1 to 64 ranks and does not combine with --index-shards\n";
		die = true;
	}
	if (params.ranks < 1 || params.ranks > 64) {
		std::cerr << PROGRAM ": --ranks must be between 1 and 64\n";
		die = true;
	}
	{ // -k: one value as in the reference, or a comma-separated list (this build only)
		std::istringstream ks(params.k_arg);
		std::string item;
		while (std::getline(ks, item, ',')) {
			std::istringstream one(item);
			int k = 0;
			if (!(one >> k) || !one.eof()) {
				std::cerr << PROGRAM ": invalid option: `-k" << params.k_arg << "'\n";
				exit(EXIT_FAILURE);
			}
			params.k_list.push_back(k);
		}
		if (params.k_list.empty())
			params.k_list.push_back(params.k_value);
		params.k_value = params.k_list[0];
	}
	// the run header prints -b / -g / --tsv as the reference does after filling in its defaults
	// (Arcs.cpp:2144-2157); with a k list the per-k names are derived in output_names()
	if (params.k_list.size() == 1) {
		const OutputNames n = output_names(params.k_value);
		params.base_name = n.base, params.dist_graph_name = n.dist, params.tsv_name = n.tsv;
	}
	if (die) {
		std::cerr << "Try " << PROGRAM << " --help for more information.\n";
		exit(EXIT_FAILURE);
	}
	// (the device check is the first HIP call: it waits until the worker ranks, if any, are forked -- run_arks)
	printf("%s\n", "Finished reading user inputs...entering runArcs()...");
	run_arks(filenames);
	return 0;
}
