// ingest.hpp -- pipelined read ingest for `arcs --arks`.
//
// The reference parses, filters and maps a read pair inside one OpenMP critical section per
// record pair (Arcs/Arcs.cpp:1185-1262); with the mapping on the GPU that serial parse is the whole
// run time.  Here the stage is a pipeline:
//
//   producer (one per input file, files in parallel)   gz inflate + kseq-compatible record split,
//       stripReadNum / name match / BX:Z: barcode / multiplicity-map gate (Arcs.cpp:1208-1265)
//       -> RawBatch (ASCII bases + per-pair gate + barcode id), messages kept in file order
//   packers (thread pool)                               2-bit pack + N mask + checkReadSequence
//       verdict per read (arks_pack_reads_host) into pinned host buffers -> PackedBatch
//   consumer (the caller's thread)                      H2D copies + gate / map / pair kernels,
//       double-buffered on two streams (arcs.cpp)
//
// Results do not depend on batch boundaries or on the order batches reach the GPU: the IndexMap
// accumulator is a commutative sum and every message is buffered per file and emitted in file
// order.  The record semantics are those of seqio.hpp (kseq as the reference instantiates it).
#pragma once

#include "arks_hip.h"
#include "seqio.hpp"

#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace arks_host {

// Arcs.cpp:243-254
inline void
strip_read_num(std::string& name)
{
	const size_t pos = name.rfind('/');
	if (pos == std::string::npos || pos == 0 || pos == name.length() - 1)
		return;
	if (!std::isdigit((unsigned char)name.at(pos + 1)))
		return;
	name.resize(pos);
}

// text after "BX:Z:" up to the next space (Arcs.cpp:1227-1237); empty when the tag is absent
inline std::string
bx_barcode(const std::string& comment)
{
	const size_t tag = comment.find("BX:Z:");
	if (tag == std::string::npos)
		return std::string();
	const size_t end = comment.find(' ', tag);
	return end != std::string::npos ? comment.substr(tag + 5, end - tag - 5) : comment.substr(tag + 5);
}

// the same without a copy (a view into `comment`)
inline std::string_view
bx_barcode_view(const std::string& comment)
{
	const size_t tag = comment.find("BX:Z:");
	if (tag == std::string::npos)
		return std::string_view();
	const size_t end = comment.find(' ', tag);
	return std::string_view(comment).substr(tag + 5, end != std::string::npos ? end - tag - 5 : std::string::npos);
}

template <typename T>
class BoundedQueue
{
  public:
	explicit BoundedQueue(size_t cap)
	  : cap_(cap)
	{}
	void push(T v)
	{
		std::unique_lock<std::mutex> lk(m_);
		not_full_.wait(lk, [&] { return q_.size() < cap_; });
		q_.push_back(std::move(v));
		not_empty_.notify_one();
	}
	// false once the queue is closed and drained
	bool pop(T& out)
	{
		std::unique_lock<std::mutex> lk(m_);
		not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
		if (q_.empty())
			return false;
		out = std::move(q_.front());
		q_.pop_front();
		not_full_.notify_one();
		return true;
	}
	// pop that gives up after `us` microseconds: 1 = got an element, 0 = nothing yet, -1 = closed and drained
	int pop_for(T& out, long us)
	{
		std::unique_lock<std::mutex> lk(m_);
		// (system_clock: pthread_cond_timedwait, which ThreadSanitizer follows; the steady-clock wait of wait_for
		// is pthread_cond_clockwait, which GCC 11's does not)
		not_empty_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(us),
		                      [&] { return !q_.empty() || closed_; });
		if (q_.empty())
			return closed_ ? -1 : 0;
		out = std::move(q_.front());
		q_.pop_front();
		not_full_.notify_one();
		return 1;
	}
	// like pop, but the element pushed LAST: for a pool of buffers that are allocated on first use, so
	// that a buffer that has just come back (allocated, warm) is taken before one that was never used --
	// the pool then grows to the number of buffers in flight, not to its capacity
	bool pop_newest(T& out)
	{
		std::unique_lock<std::mutex> lk(m_);
		not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
		if (q_.empty())
			return false;
		out = std::move(q_.back());
		q_.pop_back();
		not_full_.notify_one();
		return true;
	}
	// non-blocking variants
	bool try_pop(T& out)
	{
		std::lock_guard<std::mutex> lk(m_);
		if (q_.empty())
			return false;
		out = std::move(q_.front());
		q_.pop_front();
		not_full_.notify_one();
		return true;
	}
	bool try_push(T&& v)
	{
		std::lock_guard<std::mutex> lk(m_);
		if (q_.size() >= cap_)
			return false;
		q_.push_back(std::move(v));
		not_empty_.notify_one();
		return true;
	}
	void close()
	{
		std::lock_guard<std::mutex> lk(m_);
		closed_ = true;
		not_empty_.notify_all();
	}

  private:
	size_t cap_;
	std::deque<T> q_;
	bool closed_ = false;
	std::mutex m_;
	std::condition_variable not_full_, not_empty_;
};

// The parallel phases of a producer (inflate the members of a stretch, find its newlines, verify its record
// groups) are loops over chunks that any thread may take a share of: the producer publishes the loop here and
// works on it itself; workers that find no batch to parse help out (help()).  So the pipeline runs on the -t
// threads it was given, whichever stage is the slow one, and no thread is started per phase.
class HelpDesk
{
  public:
	// fn(i) for every i in [0, n), on the caller and on the threads that call help() meanwhile; returns when
	// all of them are done
	void parallel_for(size_t n, const std::function<void(size_t)>& fn)
	{
		if (n == 0)
			return;
		if (n == 1) {
			fn(0);
			return;
		}
		auto job = std::make_shared<Job>();
		job->n = n;
		job->fn = &fn;
		{
			std::lock_guard<std::mutex> lk(m_);
			jobs_.push_back(job);
		}
		open_.fetch_add(1, std::memory_order_release);
		run(*job);
		{
			std::unique_lock<std::mutex> lk(job->m);
			job->cv.wait(lk, [&] { return job->done.load() == n; });
		}
		open_.fetch_sub(1, std::memory_order_release);
		{
			std::lock_guard<std::mutex> lk(m_);
			for (size_t i = 0; i < jobs_.size(); ++i)
				if (jobs_[i] == job) {
					jobs_.erase(jobs_.begin() + (std::ptrdiff_t)i);
					break;
				}
		}
		// a body that threw (a std::bad_alloc from a buffer growing, say) -- on whichever thread -- counted as
		// done, so that nobody holds `fn` or the caller's stack any more; the first exception goes to the caller
		if (job->error)
			std::rethrow_exception(job->error);
	}
	// takes a share of a published loop, if there is one with chunks left: false when there was nothing to do
	bool help()
	{
		if (open_.load(std::memory_order_acquire) == 0)
			return false;
		std::shared_ptr<Job> job;
		{
			std::lock_guard<std::mutex> lk(m_);
			for (auto& j : jobs_)
				if (j->next.load(std::memory_order_relaxed) < j->n) {
					job = j;
					break;
				}
		}
		if (!job)
			return false;
		return run(*job);
	}

  private:
	struct Job
	{
		size_t n = 0;
		const std::function<void(size_t)>* fn = nullptr; // alive until done == n: its owner waits for that
		std::atomic<size_t> next{ 0 }, done{ 0 };
		std::mutex m;
		std::condition_variable cv;
		std::exception_ptr error; // the first exception of a body (under m)
	};
	static bool run(Job& j)
	{
		bool any = false;
		for (;;) {
			const size_t i = j.next.fetch_add(1);
			if (i >= j.n)
				return any;
			any = true;
			try {
				(*j.fn)(i);
			} catch (...) {
				std::lock_guard<std::mutex> lk(j.m);
				if (!j.error)
					j.error = std::current_exception();
			}
			if (j.done.fetch_add(1) + 1 == j.n) {
				std::lock_guard<std::mutex> lk(j.m);
				j.cv.notify_all();
			}
		}
	}
	std::mutex m_;
	std::vector<std::shared_ptr<Job>> jobs_;
	std::atomic<int> open_{ 0 };
};

// barcode -> dense id, fixed before the reads are parsed: every barcode that can pass the gate is
// a key of the multiplicity map (Arcs.cpp:1258-1262), so the producers only read this table
struct BarcodeDict
{
	std::unordered_map<std::string_view, uint32_t> id; // views into the keys of `mult` (node-stable)
	std::vector<const std::string*> name;
	explicit BarcodeDict(const std::unordered_map<std::string, int>& mult)
	{
		id.reserve(mult.size());
		name.reserve(mult.size());
		for (const auto& kv : mult) {
			id.emplace(std::string_view(kv.first), (uint32_t)name.size());
			name.push_back(&kv.first);
		}
	}
};

// Without a multiplicity file (-u) the reference first counts reads per barcode in a pass of its own
// (readBarcodes, Arcs.cpp:481-547).  The fused mode does that counting inside the mapping pass: ids
// are handed out as barcodes appear, and every barcode that reaches the gate is by construction a
// key of the map the pre-pass would have built -- unless a file holds a zero-length record, where
// the pre-pass stops but the pair loop goes on (then the caller falls back to two passes).
class DynamicDict
{
  public:
	// id of the barcode; *stored = the dictionary's own copy of its text (stable: usable as a cache key)
	uint32_t get(std::string_view barcode, std::string_view* stored = nullptr)
	{
		std::lock_guard<std::mutex> lk(m_);
		auto it = id_.find(barcode);
		if (it == id_.end()) {
			names_.emplace_back(barcode);
			it = id_.emplace(std::string_view(names_.back()), (uint32_t)(names_.size() - 1)).first;
		}
		if (stored)
			*stored = it->first;
		return it->second;
	}
	// (locked: a producer may be growing the deque's block map at the same time; the strings themselves
	// never move)
	size_t size() const
	{
		std::lock_guard<std::mutex> lk(m_);
		return names_.size();
	}
	const std::string& name(uint32_t id) const
	{
		std::lock_guard<std::mutex> lk(m_);
		return names_[id];
	}

  private:
	mutable std::mutex m_;
	std::unordered_map<std::string_view, uint32_t> id_; // views into names_ (a deque: stable)
	std::deque<std::string> names_;
};

// what readBarcodes would have seen in one file
struct PrepassInfo
{
	bool active = true;    // no record of length <= 0 yet
	bool zero_len = false; // a zero-length record: pre-pass and pair loop part ways
	uint64_t total = 0;    // reads with a BX:Z: tag
	uint64_t lead = 0;     // records with a comment but no tag before the first tagged one
	std::vector<uint64_t> untagged_at; // running tagged count at every later such record
	std::vector<uint32_t> counts;      // reads per barcode id
	std::vector<std::pair<uint32_t, uint32_t>> sparse_counts; // the same as (id, reads) pairs (a batch's share)

	void record(int l, const std::string& comment, DynamicDict& dict, std::unordered_map<std::string_view, uint32_t>& cache)
	{
		if (!active)
			return;
		if (l <= 0) {
			active = false;
			zero_len = l == 0;
			return;
		}
		if (comment.empty())
			return;
		if (comment.find("BX:Z:") == std::string::npos) {
			if (total == 0)
				lead++;
			else
				untagged_at.push_back(total);
			return;
		}
		const std::string_view bc = bx_barcode_view(comment);
		auto it = cache.find(bc);
		if (it == cache.end()) {
			std::string_view stored;
			const uint32_t id = dict.get(bc, &stored);
			it = cache.emplace(stored, id).first; // keyed by the dictionary's copy, not by the record's text
		}
		if (it->second >= counts.size())
			counts.resize((size_t)it->second + 1 + counts.size() / 2, 0);
		counts[it->second]++;
		total++;
	}
};

// what a later stretch of the same file saw (`part`, counted from zero) appended to `whole`
inline void
prepass_merge(PrepassInfo& whole, const PrepassInfo& part)
{
	if (!whole.active)
		return; // the pre-pass had stopped before
	for (uint64_t i = 0; i < part.lead; ++i) {
		if (whole.total == 0)
			whole.lead++;
		else
			whole.untagged_at.push_back(whole.total);
	}
	for (const uint64_t u : part.untagged_at)
		whole.untagged_at.push_back(whole.total + u);
	if (part.counts.size() > whole.counts.size())
		whole.counts.resize(part.counts.size(), 0);
	for (size_t i = 0; i < part.counts.size(); ++i)
		whole.counts[i] += part.counts[i];
	for (const auto& ic : part.sparse_counts) {
		if (ic.first >= whole.counts.size())
			whole.counts.resize((size_t)ic.first + 1 + whole.counts.size() / 2, 0);
		whole.counts[ic.first] += ic.second;
	}
	whole.total += part.total;
	whole.active = part.active;
	whole.zero_len = whole.zero_len || part.zero_len;
}

struct FileCounters
{
	uint64_t skipped_unpaired = 0, emptybarcode = 0, invalidbarcode = 0, gated = 0, skipped_invalid = 0;
};

// a growing byte buffer that does not clear what it hands out (a std::vector would zero 160 MB per batch)
// ARKS_INGEST_PROFILE=1: thread-seconds per stage of the pipeline, printed to stderr at the end of run()
struct IngestProfile
{
	enum Stage { SOURCE, SCAN, VERIFY, CUT, EMIT_WAIT, POP_WAIT, HELP, PARSE, BUF_WAIT, PACK, N_STAGES };
	std::atomic<int64_t> ns[N_STAGES];
	bool on = std::getenv("ARKS_INGEST_PROFILE") != nullptr;
	IngestProfile()
	{
		for (auto& x : ns)
			x = 0;
	}
	static IngestProfile& get()
	{
		static IngestProfile p;
		return p;
	}
	static int64_t now()
	{
		return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
	}
	struct Scope
	{
		IngestProfile& p;
		Stage s;
		int64_t t0;
		explicit Scope(Stage st)
		  : p(get())
		  , s(st)
		  , t0(p.on ? now() : 0)
		{}
		~Scope()
		{
			if (p.on)
				p.ns[s] += now() - t0;
		}
	};
	void print() const
	{
		if (!on)
			return;
		static const char* names[] = { "source(inflate/map)", "scan", "stitch+verify", "cut", "emit wait", "pop wait", "help producers", "parse", "buffer wait", "pack" };
		for (int i = 0; i < N_STAGES; ++i)
			std::fprintf(stderr, "ingest profile: %-20s %9.1f ms (thread time)\n", names[i], (double)ns[i].load() / 1e6);
	}
};

struct TextBuf
{
	std::unique_ptr<char[]> p;
	size_t n = 0, cap = 0;
	const char* view = nullptr; // not owned: a stretch of a mapped file or of `keep` (then p is unused)
	std::shared_ptr<const void> keep; // the buffer a view points into, when it is not a mapping
	char* data() { return view ? const_cast<char*>(view) : p.get(); }
	const char* data() const { return view ? view : p.get(); }
	size_t size() const { return n; }
	void clear()
	{
		n = 0;
		view = nullptr;
		keep.reset();
	}
	void reserve(size_t want)
	{
		if (want <= cap)
			return;
		const size_t c = std::max(want, cap + cap / 2);
		std::unique_ptr<char[]> q(new char[c]);
		if (n)
			std::memcpy(q.get(), p.get(), n);
		p.swap(q);
		cap = c;
	}
	void append(const char* src, size_t len)
	{
		reserve(n + len + 1);
		std::memcpy(p.get() + n, src, len);
		n += len;
	}
};

struct RawBatch
{
	int file = 0;
	int64_t seq = 0; // position of the batch within its file
	std::string bases;
	std::vector<uint64_t> off;
	std::vector<uint32_t> len;
	std::vector<uint8_t> pair_ok;
	std::vector<uint32_t> barcode_id;
	FileCounters fc;      // this batch's share
	std::string messages; // stdout text produced while parsing it
	bool last = false;    // last batch of its file
	// the split form (fast path): whole lines of 4-line FASTQ records, not parsed yet -- `text` holds
	// 8 * n_text_pairs lines, line[i] = offset of line i (one more entry: the end); the worker that
	// takes the batch parses it (parse_text_batch), `off` then points into `text`, `bases` stays empty
	bool is_text = false;
	TextBuf text;
	std::vector<uint32_t> line;
	int64_t n_text_pairs = 0;
	uint64_t first_pair = 0; // index of the batch's first pair in its file (progress messages)
	size_t pairs() const { return pair_ok.size(); }
	const char* base_ptr() const { return is_text ? text.data() : bases.data(); }
};

// host buffers of one packed batch; `alloc`/`release` let the front end use pinned memory
struct PackedBatch
{
	int file = 0;
	int64_t seq = 0;
	bool last = false;
	int64_t n_reads = 0, n_pairs = 0;
	size_t words = 0;
	uint64_t* codes = nullptr;
	uint32_t* nmask = nullptr;
	uint64_t* woff = nullptr;
	uint32_t* len = nullptr;
	uint8_t* cls = nullptr;
	uint8_t* pair_ok = nullptr;
	uint32_t* barcode_id = nullptr;
	size_t cap_words = 0, cap_reads = 0;
	// device-pack mode (the front end packs on the GPU, arks_pack_reads_device): the reads' bases as they are,
	// back to back, instead of codes / nmask / cls
	unsigned char* ascii = nullptr;
	uint64_t* aoff = nullptr; // first base of every read in `ascii`
	size_t ascii_bytes = 0, cap_ascii = 0;
	void* slab = nullptr; // the one allocation the arrays above live in
	FileCounters fc;
	std::string messages;
};

struct HostAllocator
{
	std::function<void*(size_t)> alloc = [](size_t n) { return std::malloc(n); };
	std::function<void(void*)> release = [](void* p) { std::free(p); };
};

inline void
packed_free(PackedBatch& pb, const HostAllocator& a)
{
	if (pb.slab)
		a.release(pb.slab);
	pb.slab = nullptr;
	pb.codes = nullptr, pb.nmask = nullptr, pb.woff = nullptr, pb.len = nullptr, pb.cls = nullptr,
	pb.pair_ok = nullptr, pb.barcode_id = nullptr, pb.cap_words = 0, pb.cap_reads = 0;
	pb.ascii = nullptr, pb.aoff = nullptr, pb.cap_ascii = 0;
}

// The arrays of a packed batch are carved out of ONE allocation (pinning host memory is slow and the driver
// serializes it: one call per buffer instead of seven).  Size of a slab for `words` and `reads`:
inline size_t
packed_slab_bytes(size_t words, size_t reads)
{
	auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
	return up(words * 8) + up(words * 4) + up((reads + 1) * 8) + up(reads * 4) + up(reads) + up(reads / 2 + 1) +
	       up((reads / 2 + 1) * 4);
}
// what a batch of `batch_pairs` pairs of short reads needs (longer reads make the buffer grow when they come)
inline void
packed_estimate(long batch_pairs, size_t* words, size_t* reads)
{
	*reads = 2 * (size_t)batch_pairs + 64;
	*words = *reads * 5 + 64; // reads of up to 128 bases after the gate's trim: 4 words + padding
}

inline bool
packed_reserve(PackedBatch& pb, size_t words, size_t reads, const HostAllocator& a)
{
	if (words <= pb.cap_words && reads <= pb.cap_reads)
		return true;
	size_t cap_w = pb.cap_words, cap_r = pb.cap_reads;
	if (reads > cap_r)
		cap_r = reads + reads / 8 + 64;
	if (words > cap_w)
		cap_w = words + words / 8 + 64;
	if (cap_w < cap_r * 5 + 64)
		cap_w = cap_r * 5 + 64; // the word count is known only after the read count: room for short reads at once
	char* slab = (char*)a.alloc(packed_slab_bytes(cap_w, cap_r));
	if (!slab)
		return false;
	auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
	char* q = slab;
	uint64_t* codes = (uint64_t*)q;
	q += up(cap_w * 8);
	uint32_t* nmask = (uint32_t*)q;
	q += up(cap_w * 4);
	uint64_t* woff = (uint64_t*)q;
	q += up((cap_r + 1) * 8);
	uint32_t* len = (uint32_t*)q;
	q += up(cap_r * 4);
	uint8_t* cls = (uint8_t*)q;
	q += up(cap_r);
	uint8_t* pair_ok = (uint8_t*)q;
	q += up(cap_r / 2 + 1);
	uint32_t* barcode_id = (uint32_t*)q;
	if (pb.slab) {
		if (pb.cap_reads) // pack_batch reserves for the words after it filled woff
			std::memcpy(woff, pb.woff, (std::min(pb.cap_reads, cap_r) + 1) * sizeof(uint64_t));
		a.release(pb.slab);
	}
	pb.slab = slab;
	pb.codes = codes, pb.nmask = nmask, pb.woff = woff, pb.len = len, pb.cls = cls, pb.pair_ok = pair_ok,
	pb.barcode_id = barcode_id;
	pb.cap_words = cap_w, pb.cap_reads = cap_r;
	return true;
}

// device-pack mode: bases (+ 64 bytes the device packer may read past the last one), offsets, word offsets,
// lengths, gate and barcode per pair in one slab
inline size_t
raw_slab_bytes(size_t bases, size_t reads)
{
	auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
	return up(bases + 64) + 2 * up((reads + 1) * 8) + up(reads * 4) + up(reads / 2 + 1) + up((reads / 2 + 1) * 4);
}
inline void
raw_estimate(long batch_pairs, size_t* bases, size_t* reads)
{
	*reads = 2 * (size_t)batch_pairs + 64;
	*bases = *reads * 160; // 10x reads: 128 + 151 bases per pair
}
inline bool
raw_reserve(PackedBatch& pb, size_t bases, size_t reads, const HostAllocator& a)
{
	if (bases <= pb.cap_ascii && reads <= pb.cap_reads && pb.ascii)
		return true;
	size_t cap_b = std::max(pb.cap_ascii, bases + bases / 8 + 64), cap_r = std::max(pb.cap_reads, reads + reads / 8 + 64);
	if (cap_b < cap_r * 160)
		cap_b = cap_r * 160;
	char* slab = (char*)a.alloc(raw_slab_bytes(cap_b, cap_r));
	if (!slab)
		return false;
	auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
	char* q = slab;
	unsigned char* ascii = (unsigned char*)q;
	q += up(cap_b + 64);
	uint64_t* aoff = (uint64_t*)q;
	q += up((cap_r + 1) * 8);
	uint64_t* woff = (uint64_t*)q;
	q += up((cap_r + 1) * 8);
	uint32_t* len = (uint32_t*)q;
	q += up(cap_r * 4);
	uint8_t* pair_ok = (uint8_t*)q;
	q += up(cap_r / 2 + 1);
	uint32_t* barcode_id = (uint32_t*)q;
	if (pb.slab)
		a.release(pb.slab); // (nothing of the old batch is kept: the caller fills everything after this)
	pb.slab = slab;
	pb.ascii = ascii, pb.aoff = aoff, pb.woff = woff, pb.len = len, pb.pair_ok = pair_ok, pb.barcode_id = barcode_id;
	pb.codes = nullptr, pb.nmask = nullptr, pb.cls = nullptr;
	pb.cap_ascii = cap_b, pb.cap_reads = cap_r, pb.cap_words = 0;
	return true;
}

// RawBatch -> PackedBatch for the device packer: the reads' bases gathered back to back, their offsets and
// word offsets; class, codes and N mask are the device's to make
inline int
gather_batch(RawBatch& rb, PackedBatch& pb, const HostAllocator& a)
{
	const int64_t n = (int64_t)rb.len.size(), np = (int64_t)rb.pairs();
	pb.file = rb.file, pb.seq = rb.seq, pb.last = rb.last, pb.n_reads = n, pb.n_pairs = np, pb.fc = rb.fc;
	pb.messages.swap(rb.messages);
	pb.words = 0;
	pb.ascii_bytes = 0;
	if (n == 0)
		return ARKS_OK;
	size_t bases = 0;
	for (int64_t r = 0; r < n; ++r)
		bases += rb.len[(size_t)r];
	if (!raw_reserve(pb, bases, (size_t)n, a))
		return ARKS_ERR_OOM;
	arks_word_offsets(rb.len.data(), n, pb.woff);
	pb.words = (size_t)pb.woff[n] + ARKS_PAD_WORDS;
	const char* base = rb.base_ptr();
	size_t pos = 0;
	for (int64_t r = 0; r < n; ++r) {
		const uint32_t l = rb.len[(size_t)r];
		pb.aoff[r] = pos;
		std::memcpy(pb.ascii + pos, base + rb.off[(size_t)r], l);
		pos += l;
	}
	pb.aoff[n] = pos;
	std::memset(pb.ascii + pos, 'N', 64);
	pb.ascii_bytes = pos + 64;
	std::memcpy(pb.len, rb.len.data(), (size_t)n * sizeof(uint32_t));
	std::memcpy(pb.pair_ok, rb.pair_ok.data(), (size_t)np);
	std::memcpy(pb.barcode_id, rb.barcode_id.data(), (size_t)np * sizeof(uint32_t));
	for (int64_t p = 0; p < np; ++p)
		pb.fc.gated += rb.pair_ok[(size_t)p] != 0; // (skipped_invalid needs the read classes: counted on the device)
	return ARKS_OK;
}

// RawBatch -> PackedBatch: word layout, 2-bit codes + N mask + read class (checkReadSequence,
// Arcs.cpp:366-389), and the two counters that need the class (Arcs.cpp:1273-1292)
inline int
pack_batch(RawBatch& rb, PackedBatch& pb, const HostAllocator& a)
{
	const int64_t n = (int64_t)rb.len.size(), np = (int64_t)rb.pairs();
	pb.file = rb.file, pb.seq = rb.seq, pb.last = rb.last, pb.n_reads = n, pb.n_pairs = np, pb.fc = rb.fc;
	pb.messages.swap(rb.messages);
	pb.words = 0;
	if (n == 0)
		return ARKS_OK;
	if (!packed_reserve(pb, 0, (size_t)n, a))
		return ARKS_ERR_OOM;
	arks_word_offsets(rb.len.data(), n, pb.woff);
	const size_t words = (size_t)pb.woff[n] + ARKS_PAD_WORDS;
	if (!packed_reserve(pb, words, (size_t)n, a))
		return ARKS_ERR_OOM;
	pb.words = words;
	std::memset(pb.codes, 0, words * sizeof(uint64_t));
	std::memset(pb.nmask, 0, words * sizeof(uint32_t));
	if (!rb.is_text)
		rb.bases.push_back('\0'); // (a text batch ends with a newline)
	const int rc = arks_pack_reads_host(rb.base_ptr(), rb.off.data(), rb.len.data(), pb.woff, n, pb.codes, pb.nmask,
	                                    pb.cls);
	if (rc != ARKS_OK)
		return rc;
	std::memcpy(pb.len, rb.len.data(), (size_t)n * sizeof(uint32_t));
	std::memcpy(pb.pair_ok, rb.pair_ok.data(), (size_t)np);
	std::memcpy(pb.barcode_id, rb.barcode_id.data(), (size_t)np * sizeof(uint32_t));
	for (int64_t p = 0; p < np; ++p)
		if (rb.pair_ok[(size_t)p]) {
			pb.fc.gated++;
			if (!(pb.cls[2 * p] && pb.cls[2 * p + 1]))
				pb.fc.skipped_invalid++;
		}
	return ARKS_OK;
}

// The record-pair loop of chromiumRead (Arcs.cpp:1185-1268) for one file, up to the point where
// the reference calls bestContig: pairs are appended to batches of `batch_pairs` and handed to
// `emit` (the last one flagged, possibly empty).
inline void
produce_file(
    SeqReader& rd, int file_idx, const BarcodeDict* dict, DynamicDict* dyn, PrepassInfo* pre, long batch_pairs,
    bool verbose, const std::function<void(RawBatch&&)>& emit,
    const std::function<bool(RawBatch&)>& recycled = nullptr, // hands back a used batch (its buffers are warm)
    int64_t first_seq = 0, uint64_t pairs_before = 0)        // a file whose head went through split_file
{
	std::unordered_map<std::string_view, uint32_t> cache; // fused mode: this producer's view of the dictionary
	rd.keep_qual = false; // the mapping needs no base qualities: measure them, do not copy them
	RawBatch b;
	int64_t seq = first_seq;
	auto reset = [&](RawBatch& x) {
		// a recycled batch keeps its capacity: a fresh 80 MB of bases costs an mmap and 20 k page faults
		if (recycled && recycled(x)) {
			x.bases.clear(), x.off.clear(), x.len.clear(), x.pair_ok.clear(), x.barcode_id.clear(), x.messages.clear();
			x.fc = FileCounters();
			x.last = false;
			x.is_text = false;
			x.text.clear(), x.line.clear();
		} else
			x = RawBatch();
		x.file = file_idx;
		x.seq = seq++;
		x.bases.reserve((size_t)batch_pairs * 300);
		x.off.reserve((size_t)batch_pairs * 2);
		x.len.reserve((size_t)batch_pairs * 2);
		x.pair_ok.reserve((size_t)batch_pairs);
		x.barcode_id.reserve((size_t)batch_pairs);
	};
	reset(b);
	size_t count = 2 * (size_t)pairs_before;
	bool stop = false;
	std::string n1, n2, c1, c2, s1, s2;
	while (!stop) {
		n1.clear(), n2.clear(), c1.clear(), c2.clear(), s1.clear(), s2.clear();
		int l = rd.next(); // Arcs.cpp:1187-1206
		if (pre)
			pre->record(l, rd.comment, *dyn, cache);
		if (l >= 0) {
			n1.swap(rd.name), c1.swap(rd.comment), s1.swap(rd.seq);
			l = rd.next();
			if (pre)
				pre->record(l, rd.comment, *dyn, cache);
			if (l >= 0)
				n2.swap(rd.name), c2.swap(rd.comment), s2.swap(rd.seq);
			else
				stop = true;
		} else
			stop = true;
		strip_read_num(n1);
		strip_read_num(n2);
		const bool paired = n1 == n2;
		if (!paired) {
			b.messages += "File contains unpaired reads: " + n1 + " " + n2 + "\n";
			b.fc.skipped_unpaired++;
		}
		count += 2;
		if (verbose && count % 10000000 == 0)
			b.messages += "Processed " + std::to_string(count) + " read pairs.\n";
		if (stop)
			break;
		const std::string_view b1 = bx_barcode_view(c1), b2 = bx_barcode_view(c2);
		bool valid = false;
		uint32_t bid = 0;
		if (b1.empty() || b2.empty())
			b.fc.emptybarcode++;
		else if (dict) {
			const auto it = dict->id.find(b1); // same key set as the multiplicity map (Arcs.cpp:1258)
			valid = it != dict->id.end();
			if (!valid)
				b.fc.invalidbarcode++;
			else
				bid = it->second;
		} else {
			// fused mode: mate 1 carries the tag, so the pre-pass has counted it: b1 is in the map
			valid = true;
			auto it = cache.find(b1);
			if (it == cache.end()) {
				std::string_view stored;
				const uint32_t id = dyn->get(b1, &stored);
				it = cache.emplace(stored, id).first;
			}
			bid = it->second;
		}
		const bool ok = paired && valid && b1 == b2; // Arcs.cpp:1264-1265 (goodmult is always true)
		if (!ok)
			bid = 0;
		b.off.push_back(b.bases.size());
		b.len.push_back((uint32_t)s1.size());
		b.bases += s1;
		b.off.push_back(b.bases.size());
		b.len.push_back((uint32_t)s2.size());
		b.bases += s2;
		b.pair_ok.push_back(ok ? 1 : 0);
		b.barcode_id.push_back(bid);
		if ((long)b.pairs() >= batch_pairs) {
			emit(std::move(b));
			reset(b);
		}
	}
	b.last = true;
	emit(std::move(b));
}

// ---- the fast path: 4-line FASTQ split by lines, parsed by the worker threads ----------------------------
// One record parser per file is what bounds the stage (~4 M pairs/s).  Nearly every read file is plain
// 4-line FASTQ, and for such text record boundaries are line boundaries: the producer of a file only finds the
// line starts of its (inflated) text, checks that each group of four lines IS a record as kseq would read it
// -- '@' header, one sequence line that does not start with '@', '+' or '>', a '+' line, one quality line of
// the sequence's length -- and hands out whole batches of lines; the workers do the parsing proper (names,
// stripReadNum, BX:Z:, multiplicity gate) and the packing.  The first group that is anything else (multi-line
// records, FASTA, a short quality line, an empty read, the unfinished lines at the end of the file) ends the
// fast path of that file: the text from there on goes back to the reader and the kseq-compatible loop
// (produce_file) takes over, so the semantics are kseq's for every input.

// positions of the first `max` newlines of p[0, n): AVX-512 where the CPU has it
__attribute__((target("avx512bw,bmi,bmi2"))) inline size_t
newline_positions_avx512(const char* p, size_t n, uint32_t base, uint32_t* out, size_t max, size_t* scanned)
{
	size_t cnt = 0, i = 0;
	const __m512i nl = _mm512_set1_epi8('\n');
	for (; i + 64 <= n && cnt + 64 <= max; i += 64) {
		unsigned long long m = _mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void*)(p + i)), nl);
		while (m) {
			out[cnt++] = base + (uint32_t)i + (uint32_t)_tzcnt_u64(m);
			m &= m - 1;
		}
	}
	for (; i < n && cnt < max; ++i)
		if (p[i] == '\n')
			out[cnt++] = base + (uint32_t)i;
	*scanned = i;
	return cnt;
}

inline size_t
newline_positions(const char* p, size_t n, uint32_t base, uint32_t* out, size_t max, size_t* scanned)
{
	static const bool wide = __builtin_cpu_supports("avx512bw") && !std::getenv("ARKS_NO_AVX512");
	if (wide)
		return newline_positions_avx512(p, n, base, out, max, scanned);
	size_t cnt = 0, i = 0;
	while (i < n && cnt < max) {
		const void* q = std::memchr(p + i, '\n', n - i);
		if (!q) {
			i = n;
			break;
		}
		i = (size_t)((const char*)q - p);
		out[cnt++] = base + (uint32_t)i;
		++i;
	}
	*scanned = i;
	return cnt;
}

// Is the line [b, e) -- e = position of its newline -- a length kseq would report for it?  (the reader drops a
// trailing carriage return of a line longer than one character)
inline uint32_t
kseq_line_len(const char* t, uint32_t b, uint32_t e)
{
	uint32_t n = e - b;
	if (n > 1 && t[e - 1] == '\r')
		n--;
	return n;
}

// the four lines from line index i on are one FASTQ record in kseq's reading (see above)
inline bool
regular_record(const char* t, const uint32_t* start, size_t i)
{
	const uint32_t h = start[i], s = start[i + 1], p = start[i + 2], q = start[i + 3], e = start[i + 4];
	if (t[h] != '@' || t[p] != '+')
		return false;
	const uint32_t sl = kseq_line_len(t, s, p - 1), ql = kseq_line_len(t, q, e - 1);
	if (sl == 0 || sl != ql)
		return false;
	const char c = t[s];
	return c != '@' && c != '+' && c != '>';
}

inline bool
kseq_space(char c)
{
	return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r';
}

// name and comment of the header line [b, e) (b = the '@'), as SeqReader::next() splits them
inline void
split_header(const char* t, uint32_t b, uint32_t e, std::string_view& name, std::string_view& comment)
{
	uint32_t i = b + 1;
	while (i < e && !kseq_space(t[i]))
		++i;
	name = std::string_view(t + b + 1, i - b - 1);
	comment = std::string_view();
	if (i < e) { // the terminator is inside the line: the rest is the comment
		uint32_t cb = i + 1, ce = e;
		if (ce - cb > 1 && t[ce - 1] == '\r')
			ce--;
		comment = std::string_view(t + cb, ce > cb ? ce - cb : 0);
	}
}

inline std::string_view
strip_read_num_view(std::string_view name)
{
	const size_t pos = name.rfind('/');
	if (pos == std::string_view::npos || pos == 0 || pos == name.length() - 1)
		return name;
	if (!std::isdigit((unsigned char)name[pos + 1]))
		return name;
	return name.substr(0, pos);
}

inline std::string_view
bx_barcode_sv(std::string_view comment)
{
	const size_t tag = comment.find("BX:Z:");
	if (tag == std::string_view::npos)
		return std::string_view();
	const size_t end = comment.find(' ', tag);
	return comment.substr(tag + 5, end != std::string_view::npos ? end - tag - 5 : std::string_view::npos);
}

// The record-pair loop of chromiumRead (Arcs.cpp:1185-1268) over a batch of regular records: fills the
// parsed fields of `b` exactly as produce_file does for the same records; `pre` (fused mode) = this batch's
// share of what the barcode pre-pass would have seen.
inline void
parse_text_batch(
    RawBatch& b, const BarcodeDict* dict, DynamicDict* dyn, std::unordered_map<std::string_view, uint32_t>& cache,
    PrepassInfo* pre, bool verbose)
{
	const char* t = b.text.data();
	const uint32_t* L = b.line.data();
	const size_t np = (size_t)b.n_text_pairs;
	b.off.resize(2 * np), b.len.resize(2 * np), b.pair_ok.resize(np), b.barcode_id.resize(np);
	std::unordered_map<uint32_t, uint32_t> seen; // fused mode: reads per barcode id in this batch
	auto record = [&](std::string_view comment) { // PrepassInfo::record for a record of positive length
		if (comment.empty())
			return;
		if (comment.find("BX:Z:") == std::string_view::npos) {
			if (pre->total == 0)
				pre->lead++;
			else
				pre->untagged_at.push_back(pre->total);
			return;
		}
		const std::string_view bc = bx_barcode_sv(comment);
		auto it = cache.find(bc);
		if (it == cache.end()) {
			std::string_view stored;
			const uint32_t id = dyn->get(bc, &stored);
			it = cache.emplace(stored, id).first;
		}
		seen[it->second]++;
		pre->total++;
	};
	bool have_last = false, last_valid = false;
	uint32_t last_id = 0;
	std::string_view last_bc; // (a view of this batch's text)
	for (size_t p = 0; p < np; ++p) {
		const uint32_t* l = L + 8 * p;
		std::string_view n1, c1, n2, c2;
		split_header(t, l[0], l[1] - 1, n1, c1);
		split_header(t, l[4], l[5] - 1, n2, c2);
		if (pre) {
			record(c1);
			record(c2);
		}
		n1 = strip_read_num_view(n1);
		n2 = strip_read_num_view(n2);
		const bool paired = n1 == n2;
		if (!paired) {
			b.messages.append("File contains unpaired reads: ").append(n1).append(" ").append(n2).append("\n");
			b.fc.skipped_unpaired++;
		}
		const uint64_t count = 2 * (b.first_pair + p + 1);
		if (verbose && count % 10000000 == 0)
			b.messages += "Processed " + std::to_string(count) + " read pairs.\n";
		const std::string_view b1 = bx_barcode_sv(c1), b2 = bx_barcode_sv(c2);
		bool valid = false;
		uint32_t bid = 0;
		if (b1.empty() || b2.empty())
			b.fc.emptybarcode++;
		else if (have_last && b1 == last_bc) {
			// linked-read files are sorted by barcode: the pair in front answers most lookups (with millions of
			// barcodes a lookup in the dictionary is a chain of cache misses)
			valid = last_valid;
			bid = last_id;
			if (!valid)
				b.fc.invalidbarcode++;
		} else if (dict) {
			const auto it = dict->id.find(b1);
			valid = it != dict->id.end();
			if (!valid)
				b.fc.invalidbarcode++;
			else
				bid = it->second;
			have_last = true, last_bc = b1, last_valid = valid, last_id = bid;
		} else {
			valid = true;
			auto it = cache.find(b1);
			if (it == cache.end()) {
				std::string_view stored;
				const uint32_t id = dyn->get(b1, &stored);
				it = cache.emplace(stored, id).first;
			}
			bid = it->second;
			have_last = true, last_bc = b1, last_valid = true, last_id = bid;
		}
		const bool ok = paired && valid && b1 == b2;
		b.off[2 * p] = l[1];
		b.len[2 * p] = kseq_line_len(t, l[1], l[2] - 1);
		b.off[2 * p + 1] = l[5];
		b.len[2 * p + 1] = kseq_line_len(t, l[5], l[6] - 1);
		b.pair_ok[p] = ok ? 1 : 0;
		b.barcode_id[p] = ok ? bid : 0;
	}
	if (pre) {
		pre->sparse_counts.assign(seen.begin(), seen.end());
	}
}

// The same over text that is in memory a stretch at a time, where the line scan is not bound to one thread
// either: `scan_threads` threads find the newlines of a slice of the stretch each, the line starts are put
// together, the slices' record groups are verified in parallel, and only then (everything before the first
// irregular group being known to be regular) the batches are cut -- views of the stretch, no copy.  Two sources:
// a plain file mapped into memory (MappedStretches), and a BGZF file whose members the same threads inflate
// into a buffer per stretch (InflatedStretches).
struct StretchSource
{
	virtual ~StretchSource() = default;
	// the next stretch, beginning at the first byte no batch has taken: false when no text is left;
	// *keep = what keeps the text alive (empty for a mapping), *last = no stretch follows this one
	virtual bool next(const char** p, size_t* n, std::shared_ptr<const void>* keep, bool* last) = 0;
	// the batches cut from that stretch end `used` bytes into it
	virtual void consumed(size_t used) = 0;
};

// a stretch = at least one batch's worth of text (so that a stretch yields full batches) and enough for every
// scan thread, at most a gigabyte (offsets within it are 32-bit); the workers parse the batches of one stretch
// while the next is scanned
inline size_t
stretch_bytes(long batch_pairs, unsigned scan_threads)
{
	if (const char* e = std::getenv("ARKS_STRETCH_BYTES")) // tests: many short stretches
		return std::max<size_t>(4096, (size_t)std::atoll(e));
	return std::min<size_t>((size_t)1 << 30, std::max<size_t>((size_t)batch_pairs * 720, (size_t)scan_threads << 25));
}

class MappedStretches : public StretchSource
{
  public:
	MappedStretches(const char* map, size_t size, size_t stretch)
	  : map_(map)
	  , size_(size)
	  , stretch_(stretch)
	{}
	bool next(const char** p, size_t* n, std::shared_ptr<const void>* keep, bool* last) override
	{
		const size_t span = std::min(size_ - at_, stretch_);
		if (span == 0)
			return false;
		*p = map_ + at_;
		*n = span;
		keep->reset();
		*last = at_ + span >= size_;
		return true;
	}
	void consumed(size_t used) override { at_ += used; }
	size_t offset() const { return at_; } // where the sequential loop reads on

  private:
	const char* map_;
	size_t size_, stretch_, at_ = 0;
};

// Buffers for inflated stretches, reused once the batches cut from them are done: one pool for all producers.
// A fresh buffer costs a page fault per page on first touch -- taken by the threads that inflate into it -- so
// the buffers are mappings of their own for which the kernel is asked for huge pages (512x fewer faults where
// transparent huge pages are available), and they are kept for the whole run.
class StretchPool
{
  public:
	struct Buf
	{
		unsigned char* p = nullptr;
		size_t cap = 0;
		Buf() = default;
		Buf(const Buf&) = delete;
		Buf& operator=(const Buf&) = delete;
		~Buf()
		{
			if (p)
				(void)::munmap(p, cap);
		}
	};
	std::shared_ptr<Buf> get(size_t want)
	{
		std::lock_guard<std::mutex> lk(m_);
		for (auto& b : bufs_)
			if (b.use_count() == 1 && b->cap >= want)
				return b;
		for (auto& b : bufs_)
			if (b.use_count() == 1) { // too small: replace it
				b = make(want);
				return b;
			}
		bufs_.push_back(make(want));
		return bufs_.back();
	}

  private:
	static std::shared_ptr<Buf> make(size_t want)
	{
		auto b = std::make_shared<Buf>();
		const size_t huge = (size_t)2 << 20;
		b->cap = (want + want / 8 + huge - 1) & ~(huge - 1);
		void* m = ::mmap(nullptr, b->cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (m == MAP_FAILED)
			throw std::bad_alloc();
		(void)::madvise(m, b->cap, MADV_HUGEPAGE);
		b->p = (unsigned char*)m;
		return b;
	}
	std::mutex m_;
	std::vector<std::shared_ptr<Buf>> bufs_;
};

// what a producer keeps from stretch to stretch and file to file (line ends per chunk, line starts): vectors
// of tens of megabytes that would otherwise be allocated -- and faulted in -- again for every file
struct SplitScratch
{
	std::vector<uint32_t> start;
	std::vector<std::vector<uint32_t>> nl;
};

class InflatedStretches : public StretchSource
{
  public:
	InflatedStretches(BgzfStretches& z, StretchPool& pool, size_t stretch, ParallelFor pf)
	  : z_(z)
	  , pool_(pool)
	  , stretch_(stretch)
	  , pf_(std::move(pf))
	{}
	bool next(const char** p, size_t* n, std::shared_ptr<const void>* keep, bool* last) override
	{
		// what the previous stretch left over goes in front (it is short: less than a group of records, unless
		// the text stopped being regular -- and then no stretch follows)
		const size_t carry = cur_ ? cur_n_ - cur_used_ : 0;
		const size_t planned = z_.at_end() ? 0 : z_.plan(stretch_);
		if (carry + planned == 0)
			return false;
		std::shared_ptr<StretchPool::Buf> nb = pool_.get(carry + planned + 64);
		if (carry)
			std::memcpy(nb->p, cur_->p + cur_used_, carry);
		const size_t good = planned ? z_.inflate(nb->p + carry, pf_) : 0;
		cur_ = std::move(nb);
		cur_n_ = carry + good;
		cur_used_ = 0;
		*p = (const char*)cur_->p;
		*n = cur_n_;
		*keep = cur_;
		*last = z_.at_end();
		return cur_n_ > 0;
	}
	void consumed(size_t used) override { cur_used_ = used; }
	// the text no batch has taken (to go back into the reader in front of what follows offset() in the file)
	const unsigned char* rest(size_t* n) const
	{
		*n = cur_ ? cur_n_ - cur_used_ : 0;
		return cur_ ? cur_->p + cur_used_ : nullptr;
	}

  private:
	BgzfStretches& z_;
	StretchPool& pool_;
	size_t stretch_;
	ParallelFor pf_;
	std::shared_ptr<StretchPool::Buf> cur_;
	size_t cur_n_ = 0, cur_used_ = 0;
};

// an ordinary gzip file: the blocks of a stretch decoded by the threads of `pf` (pgzip.hpp)
class GzTextStretches : public StretchSource
{
  public:
	GzTextStretches(GzStretches& z, StretchPool& pool, ParallelFor pf)
	  : z_(z)
	  , pool_(pool)
	  , pf_(std::move(pf))
	{}
	bool next(const char** p, size_t* n, std::shared_ptr<const void>* keep, bool* last) override
	{
		const size_t carry = cur_ ? cur_n_ - cur_used_ : 0;
		std::shared_ptr<StretchPool::Buf> nb;
		size_t got = 0;
		if (!z_.at_end())
			got = z_.next(pf_, [&](size_t bytes) {
				nb = pool_.get(carry + bytes + 64);
				return nb->p + carry;
			});
		if (carry + got == 0)
			return false;
		if (!nb)
			nb = pool_.get(carry + 64);
		if (carry)
			std::memcpy(nb->p, cur_->p + cur_used_, carry);
		cur_ = std::move(nb);
		cur_n_ = carry + got;
		cur_used_ = 0;
		*p = (const char*)cur_->p;
		*n = cur_n_;
		*keep = cur_;
		*last = z_.at_end();
		return true;
	}
	void consumed(size_t used) override { cur_used_ = used; }
	const unsigned char* rest(size_t* n) const
	{
		*n = cur_ ? cur_n_ - cur_used_ : 0;
		return cur_ ? cur_->p + cur_used_ : nullptr;
	}

  private:
	GzStretches& z_;
	StretchPool& pool_;
	ParallelFor pf_;
	std::shared_ptr<StretchPool::Buf> cur_;
	size_t cur_n_ = 0, cur_used_ = 0;
};

// Returns the batches handed out and, in *pairs_out, the pairs they held; the source says where the
// sequential loop continues.  `pf` runs the loops over chunks of the stretch.
inline int64_t
split_stretches(
    StretchSource& src, int file_idx, long batch_pairs, const std::function<void(RawBatch&&)>& emit,
    const std::function<bool(RawBatch&)>& recycled, uint64_t* pairs_out, const ParallelFor& pf, SplitScratch& scratch)
{
	int64_t seq = 0;
	uint64_t pairs_done = 0;
	std::vector<uint32_t>& start = scratch.start; // line starts of the current stretch, relative to its first byte
	std::vector<std::vector<uint32_t>>& nl = scratch.nl;
	for (;;) {
		const char* text = nullptr;
		size_t span = 0;
		std::shared_ptr<const void> keep;
		bool last = false;
		{
			IngestProfile::Scope sc(IngestProfile::SOURCE);
			if (!src.next(&text, &span, &keep, &last))
				break;
		}
		const size_t slice = (size_t)1 << 22; // 4 MB of text per chunk
		const size_t T = (span + slice - 1) / slice;
		if (nl.size() < T)
			nl.resize(T);
		{
			IngestProfile::Scope sc(IngestProfile::SCAN);
			pf(T, [&](size_t i) {
				const size_t lo = i * slice, hi = std::min(span, lo + slice);
				std::vector<uint32_t>& v = nl[i];
				v.resize(std::max(v.capacity(), (hi - lo) / 24 + 64)); // grown below when the lines are shorter than that
				size_t n = 0, pos = lo;
				while (pos < hi) {
					if (v.size() - n < 4096)
						v.resize(v.size() * 2);
					size_t adv = 0;
					n += newline_positions(text + pos, hi - pos, (uint32_t)pos, v.data() + n, v.size() - n, &adv);
					pos += adv;
				}
				v.resize(n);
			});
		}
		std::vector<size_t> first(T + 1, 0);
		for (size_t i = 0; i < T; ++i)
			first[i + 1] = first[i] + nl[i].size();
		const size_t nlines = first[T];
		const size_t full = nlines / 8;
		std::vector<size_t> bad(T, full); // first irregular pair each chunk saw
		{
			IngestProfile::Scope sc(IngestProfile::VERIFY);
			start.resize(nlines + 1);
			start[0] = 0;
			pf(T, [&](size_t i) { // line ends -> line starts, all in one array
				uint32_t* out = start.data() + 1 + first[i];
				const std::vector<uint32_t>& v = nl[i];
				for (size_t j = 0; j < v.size(); ++j)
					out[j] = v[j] + 1;
			});
			pf(T, [&](size_t i) {
				const size_t lo = full * i / T, hi = full * (i + 1) / T;
				for (size_t p = lo; p < hi; ++p)
					if (!regular_record(text, start.data(), 8 * p) || !regular_record(text, start.data(), 8 * p + 4)) {
						bad[i] = p;
						return;
					}
			});
		}
		size_t good = full;
		for (size_t i = 0; i < T; ++i)
			good = std::min(good, bad[i]);
		std::unique_ptr<IngestProfile::Scope> cut_scope(new IngestProfile::Scope(IngestProfile::CUT));
		for (size_t p0 = 0; p0 < good; p0 += (size_t)batch_pairs) {
			const size_t P = std::min<size_t>((size_t)batch_pairs, good - p0);
			RawBatch b;
			if (recycled)
				(void)recycled(b);
			b.bases.clear(), b.off.clear(), b.len.clear(), b.pair_ok.clear(), b.barcode_id.clear(), b.messages.clear();
			b.fc = FileCounters();
			b.last = false;
			b.is_text = true;
			b.file = file_idx;
			b.text.clear();
			const uint32_t s0 = start[8 * p0];
			b.line.resize(8 * P + 1);
			for (size_t j = 0; j <= 8 * P; ++j)
				b.line[j] = start[8 * p0 + j] - s0;
			b.text.view = text + s0;
			b.text.keep = keep;
			b.text.n = (size_t)(start[8 * (p0 + P)] - s0);
			b.n_text_pairs = (int64_t)P;
			b.first_pair = pairs_done;
			b.seq = seq++;
			pairs_done += P;
			cut_scope.reset();
			{
				IngestProfile::Scope sc(IngestProfile::EMIT_WAIT);
				emit(std::move(b));
			}
			cut_scope.reset(new IngestProfile::Scope(IngestProfile::CUT));
		}
		cut_scope.reset();
		src.consumed(start[8 * good]);
		if (good < full || good == 0 || last)
			break; // irregular text, or the last lines of the file: the sequential loop reads on from there
	}
	*pairs_out = pairs_done;
	return seq;
}

// The producer of one file on the fast path: batches of `batch_pairs` regular record pairs as text
// (`emit`), until the text stops being regular or ends; what is left goes back into the reader.  Returns
// the number of batches handed out (the sequential loop continues the numbering) and, in *pairs, the pairs
// they held.
inline int64_t
split_file(
    SeqReader& rd, int file_idx, long batch_pairs, const std::function<void(RawBatch&&)>& emit,
    const std::function<bool(RawBatch&)>& recycled, uint64_t* pairs_out)
{
	int64_t seq = 0;
	uint64_t pairs_done = 0;
	std::vector<char> carry; // text read beyond the batch that was cut
	const size_t want_lines = (size_t)batch_pairs * 8;
	size_t text_estimate = std::min<size_t>((size_t)batch_pairs * 680, (size_t)1 << 29); // 10x reads: ~630 B per pair
	bool eof = false, irregular = false;
	while (!eof && !irregular) {
		RawBatch b;
		if (recycled)
			(void)recycled(b);
		b.bases.clear(), b.off.clear(), b.len.clear(), b.pair_ok.clear(), b.barcode_id.clear(), b.messages.clear();
		b.fc = FileCounters();
		b.last = false;
		b.is_text = true;
		b.file = file_idx;
		b.text.clear();
		// one allocation per buffer, at the size the previous batch needed (a fresh buffer is one page fault per
		// 4 KB on first touch: growing it step by step would pay that several times over)
		b.text.reserve(std::max(text_estimate, carry.size() + ((size_t)1 << 22)) + 1);
		b.text.append(carry.data(), carry.size());
		carry.clear();
		b.line.resize(want_lines + 1);
		size_t nlines = 0, scanned = 0;
		// line ends accumulate in b.line[1..]; b.line[0] = 0 is the first line's start
		b.line[0] = 0;
		for (;;) {
			size_t adv = 0;
			nlines += newline_positions(b.text.data() + scanned, b.text.size() - scanned, (uint32_t)scanned,
			                            b.line.data() + 1 + nlines, want_lines - nlines, &adv);
			scanned += adv;
			if (nlines == want_lines || eof)
				break;
			const size_t old = b.text.size(), room = (size_t)1 << 22;
			if (old + room > 0xF0000000ull) { // offsets are 32-bit: lines this long are not FASTQ reads
				irregular = true;
				break;
			}
			b.text.reserve(old + room + 1);
			const int got = rd.read_raw((unsigned char*)b.text.data() + old, (int)room);
			b.text.n = old + (size_t)std::max(got, 0);
			if (got <= 0)
				eof = true;
		}
		// line ends -> line starts (start of line i + 1 = end of line i + 1)
		for (size_t i = 1; i <= nlines; ++i)
			b.line[i] += 1;
		// whole pairs that are regular
		size_t good = 0;
		const size_t full = nlines / 8;
		while (good < full && regular_record(b.text.data(), b.line.data(), 8 * good) &&
		       regular_record(b.text.data(), b.line.data(), 8 * good + 4))
			++good;
		if (good < full || nlines < want_lines)
			irregular = true; // (or the end of the file: the sequential loop reads the last lines)
		const size_t cut = good ? b.line[8 * good] : 0; // first byte that does not belong to the batch
		if (good == (size_t)batch_pairs)
			text_estimate = std::min<size_t>(b.text.size() + b.text.size() / 16, 0xE0000000ull);
		carry.assign(b.text.data() + cut, b.text.data() + b.text.size());
		if (good) {
			b.text.n = cut;
			b.line.resize(8 * good + 1);
			b.n_text_pairs = (int64_t)good;
			b.first_pair = pairs_done;
			b.seq = seq++;
			pairs_done += good;
			emit(std::move(b));
		}
	}
	if (!carry.empty())
		rd.unread((const unsigned char*)carry.data(), carry.size());
	*pairs_out = pairs_done;
	return seq;
}

// Runs producers (one per file, at most `n_producers` at a time) and `n_packers` packers; calls
// `consume` on the caller's thread for every packed batch (any file order; batches of one file in
// order), then `recycle`d buffers go back to the packers.  Returns the first ABI error, or ARKS_OK.
class IngestPipeline
{
  public:
	// dict != NULL: barcodes come from a multiplicity map read beforehand; dict == NULL: fused mode,
	// `dynamic()` and `prepass()` hold what the barcode pre-pass would have produced
	IngestPipeline(
	    std::vector<SeqReader*> readers, const BarcodeDict* dict, long batch_pairs, bool verbose, unsigned threads,
	    HostAllocator alloc)
	  : readers_(std::move(readers))
	  , dict_(dict)
	  , prepass_(dict ? 0 : readers_.size())
	  , tail_pre_(dict ? 0 : readers_.size())
	  , head_pre_(dict ? 0 : readers_.size())
	  , batch_pairs_(batch_pairs)
	  , verbose_(verbose)
	  , alloc_(std::move(alloc))
	  , raw_q_(4)
	  , packed_q_(4)
	  , free_q_(1u << 20)
	{
		const unsigned nf = (unsigned)readers_.size();
		// ARKS_SEQUENTIAL_INGEST=1: every file through the kseq-compatible loop alone (A/B runs, tests)
		fast_path_ = std::getenv("ARKS_SEQUENTIAL_INGEST") == nullptr;
		// a producer on the fast path only reads and finds lines (the workers parse): a quarter of the threads
		// is plenty for them; without it a producer IS a parser: half
		// ordinary gzip files: with few of them (fewer than a third of the threads; one file from two threads on)
		// each is decoded by several threads at once (pgzip.hpp: more work per byte than the one-thread inflater,
		// but it spreads: one file reads at 2.5 / 4.1 / 5.4 / 7.2 / 9.8 M pairs/s with 2 / 4 / 6 / 8 / 16 threads
		// against 2.1 for the one thread); with many, a thread per file is the better use of the threads
		unsigned n_serial = 0, n_gz = 0;
		for (SeqReader* r : readers_)
			n_gz += r->splittable_gzip();
		use_pgzip_ = fast_path_ && n_gz > 0 && (3 * n_gz < threads || (n_gz == 1 && threads >= 2)) && !std::getenv("ARKS_NO_PGZIP");
		if (const char* e = std::getenv("ARKS_PGZIP")) // 1 / 0: whatever the counts say (tests, A/B runs)
			use_pgzip_ = fast_path_ && std::atoi(e) != 0;
		for (SeqReader* r : readers_)
			n_serial += r->serial_source() || (r->splittable_gzip() && !use_pgzip_);
		split_threads(threads, nf, fast_path_, &n_producers_, &n_packers_, n_serial);
		n_buffers_ = n_packers_ + 3;
		// short bursts of line scanning per gigabyte of a mapped file: what -t leaves per producer, at most 16
		n_scan_ = std::max(1u, std::min(16u, threads / n_producers_));
	}

	// n_serial of the files are streams that one thread must inflate from end to end (ordinary .gz): their
	// producers are busy threads, not coordinators, so up to half of the threads go to them
	static void split_threads(
	    unsigned threads, unsigned n_files, bool fast_path, unsigned* producers, unsigned* packers, unsigned n_serial = 0)
	{
		unsigned want = std::max(1u, threads / (fast_path ? 4 : 2));
		if (fast_path)
			want = std::max(want, std::min(n_serial, threads / 2));
		*producers = std::max(1u, std::min(n_files, want));
		*packers = std::max(1u, threads > *producers ? threads - *producers : 1u);
	}
	// packed-batch buffers a run over n_files files with `threads` threads keeps in flight
	static unsigned buffers_for(unsigned threads, unsigned n_files)
	{
		unsigned producers = 1, packers = 1;
		split_threads(threads, n_files, std::getenv("ARKS_SEQUENTIAL_INGEST") == nullptr, &producers, &packers);
		return packers + 3;
	}

	// the batches carry the reads' bases instead of packed words: the caller packs (and classifies) on the device
	void set_device_pack(bool on) { device_pack_ = on; }
	// a consumer that keeps more than two batches in flight (several GPU lanes, a round of a sharded exchange being
	// collected) needs as many more buffers, or the packers wait for buffers the consumer will not give back
	void add_buffers(unsigned n) { n_buffers_ += n; }

	DynamicDict& dynamic() { return dynamic_; }
	const std::vector<PrepassInfo>& prepass() const { return prepass_; }
	unsigned producers() const { return n_producers_; }
	unsigned packers() const { return n_packers_; }

	// `finish` (may be empty) runs after the last batch was handed to `consume` and before the batch
	// buffers are released: a consumer that keeps batches in flight (asynchronous copies out of the
	// pinned buffers) retires them there and gives them back with recycle()
	int run(const std::function<int(PackedBatch*)>& consume, const std::function<void()>& finish = nullptr)
	{
		std::vector<PackedBatch> pool(n_buffers_);
		for (auto& pb : pool)
			free_q_.push(&pb);
		std::mutex file_m;
		size_t next_file = 0;
		std::vector<std::thread> producers, packers;
		std::mutex err_m;
		int first_err = ARKS_OK;
		for (unsigned t = 0; t < n_producers_; ++t)
			producers.emplace_back([&] {
				SplitScratch scratch;
				const ParallelFor pf = [&](size_t n, const std::function<void(size_t)>& fn) { desk_.parallel_for(n, fn); };
				for (;;) {
					size_t f;
					{
						std::lock_guard<std::mutex> lk(file_m);
						if (next_file >= readers_.size())
							return;
						f = next_file++;
					}
					// the head of the file on the fast path (batches of lines, parsed by the workers), whatever is
					// left -- at least the end of the file -- through the kseq-compatible loop
					int64_t first_seq = 0;
					uint64_t pairs_before = 0;
					const auto emit = [&](RawBatch&& rb) { raw_q_.push(std::move(rb)); };
					const auto reuse = [&](RawBatch& out) { return raw_free_.try_pop(out); };
					if (fast_path_) {
						size_t msize = 0;
						const size_t stretch = stretch_bytes(batch_pairs_, n_scan_);
						std::unique_ptr<BgzfStretches> z;
						std::unique_ptr<GzStretches> g;
						if (const char* map = readers_[f]->map_plain(&msize)) {
							MappedStretches src(map, msize, stretch);
							first_seq = split_stretches(src, (int)f, batch_pairs_, emit, reuse, &pairs_before, pf, scratch);
							readers_[f]->continue_at(src.offset());
						} else if ((z = readers_[f]->bgzf_stretches())) {
							InflatedStretches src(*z, stretch_pool_, stretch, pf);
							first_seq = split_stretches(src, (int)f, batch_pairs_, emit, reuse, &pairs_before, pf, scratch);
							size_t n_rest = 0;
							const unsigned char* rest = src.rest(&n_rest);
							readers_[f]->bgzf_continue_at(z->offset());
							if (n_rest)
								readers_[f]->unread(rest, n_rest);
						} else if (use_pgzip_ && (g = readers_[f]->gz_stretches(pgzip_chunk_, std::max(4u, std::min(64u, 4 * (n_producers_ + n_packers_)))))) {
							GzTextStretches src(*g, stretch_pool_, pf);
							first_seq = split_stretches(src, (int)f, batch_pairs_, emit, reuse, &pairs_before, pf, scratch);
							size_t n_rest = 0;
							const unsigned char* rest = src.rest(&n_rest);
							readers_[f]->gz_continue(*g);
							if (n_rest)
								readers_[f]->unread(rest, n_rest);
						} else
							first_seq = split_file(*readers_[f], (int)f, batch_pairs_, emit, reuse, &pairs_before);
					}
					produce_file(*readers_[f], (int)f, dict_, dict_ ? nullptr : &dynamic_,
					             dict_ ? nullptr : &tail_pre_[f], batch_pairs_, verbose_, emit, reuse, first_seq,
					             pairs_before);
				}
			});
		for (unsigned t = 0; t < n_packers_; ++t)
			packers.emplace_back([&] {
				RawBatch rb;
				std::unordered_map<std::string_view, uint32_t> cache; // fused mode: this worker's view of the dictionary
				for (;;) {
					{
						// a batch to parse if there is one; else a share of a producer's loop; else wait (briefly:
						// a loop may be published meanwhile)
						int got = raw_q_.try_pop(rb) ? 1 : 0;
						if (!got) {
							{
								IngestProfile::Scope sc(IngestProfile::HELP);
								if (desk_.help())
									continue;
							}
							IngestProfile::Scope sc(IngestProfile::POP_WAIT);
							got = raw_q_.pop_for(rb, 100);
						}
						if (got < 0)
							break;
						if (got == 0)
							continue;
					}
					if (rb.is_text) {
						IngestProfile::Scope sc(IngestProfile::PARSE);
						PrepassInfo part;
						parse_text_batch(rb, dict_, dict_ ? nullptr : &dynamic_, cache, dict_ ? nullptr : &part, verbose_);
						if (!dict_) {
							std::lock_guard<std::mutex> lk(err_m);
							head_pre_[(size_t)rb.file].emplace_back(rb.seq, std::move(part));
						}
					}
					PackedBatch* pb = nullptr;
					{
						IngestProfile::Scope sc(IngestProfile::BUF_WAIT);
						if (!free_q_.pop_newest(pb))
							return;
					}
					IngestProfile::Scope sc(IngestProfile::PACK);
					const int rc = device_pack_ ? gather_batch(rb, *pb, alloc_) : pack_batch(rb, *pb, alloc_);
					if (rc != ARKS_OK) {
						std::lock_guard<std::mutex> lk(err_m);
						if (first_err == ARKS_OK)
							first_err = rc;
					}
					packed_q_.push(pb);
					rb.text.clear();                   // (lets go of the stretch it was a view of)
					raw_free_.try_push(std::move(rb)); // back to the producers, buffers and all
					rb = RawBatch();
				}
			});
		std::thread closer([&] {
			for (auto& t : producers)
				t.join();
			raw_q_.close();
			for (auto& t : packers)
				t.join();
			packed_q_.close();
		});
		// the packers may finish out of order: a batch carries (file, seq) so that the caller can put
		// its messages back in file order; counters and the IndexMap are sums
		PackedBatch* pb = nullptr;
		int rc = ARKS_OK;
		while (packed_q_.pop(pb)) {
			if (rc == ARKS_OK) {
				std::lock_guard<std::mutex> lk(err_m);
				rc = first_err;
			}
			if (rc == ARKS_OK)
				rc = consume(pb);
			else
				recycle(pb);
		}
		closer.join();
		// the barcode pre-pass of every file: its fast-path batches in file order, then the sequential rest
		for (size_t f = 0; f < prepass_.size(); ++f) {
			std::sort(head_pre_[f].begin(), head_pre_[f].end(),
			          [](const std::pair<int64_t, PrepassInfo>& a, const std::pair<int64_t, PrepassInfo>& b) { return a.first < b.first; });
			prepass_[f] = PrepassInfo();
			for (const auto& part : head_pre_[f])
				prepass_merge(prepass_[f], part.second);
			prepass_merge(prepass_[f], tail_pre_[f]);
			head_pre_[f].clear();
		}
		if (finish)
			finish();
		IngestProfile::get().print();
		free_q_.close();
		for (auto& b : pool)
			packed_free(b, alloc_);
		return rc;
	}

	// hand a consumed batch's buffers back to the packers
	void recycle(PackedBatch* pb) { free_q_.push(pb); }

  private:
	std::vector<SeqReader*> readers_;
	const BarcodeDict* dict_;
	DynamicDict dynamic_;
	std::vector<PrepassInfo> prepass_, tail_pre_;
	std::vector<std::vector<std::pair<int64_t, PrepassInfo>>> head_pre_; // fast-path batches: (seq, share)
	bool fast_path_ = true;
	long batch_pairs_;
	bool verbose_;
	HostAllocator alloc_;
	unsigned n_producers_ = 1, n_packers_ = 1, n_buffers_ = 4, n_scan_ = 1;
	// compressed bytes per chunk of an ordinary gzip file decoded in parallel (ARKS_PGZIP_CHUNK: tests)
	size_t pgzip_chunk_ = std::getenv("ARKS_PGZIP_CHUNK") ? (size_t)std::atoll(std::getenv("ARKS_PGZIP_CHUNK")) : (size_t)1 << 20;
	bool use_pgzip_ = false;
	bool device_pack_ = false;
	BoundedQueue<RawBatch> raw_q_, raw_free_{ 8 };
	HelpDesk desk_;
	StretchPool stretch_pool_;
	BoundedQueue<PackedBatch*> packed_q_, free_q_;
};

} // namespace arks_host
