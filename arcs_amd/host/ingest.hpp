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

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
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
	// non-blocking variants (a recycling list: nothing waits on it)
	bool try_pop(T& out)
	{
		std::lock_guard<std::mutex> lk(m_);
		if (q_.empty())
			return false;
		out = std::move(q_.front());
		q_.pop_front();
		return true;
	}
	bool try_push(T&& v)
	{
		std::lock_guard<std::mutex> lk(m_);
		if (q_.size() >= cap_)
			return false;
		q_.push_back(std::move(v));
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

struct FileCounters
{
	uint64_t skipped_unpaired = 0, emptybarcode = 0, invalidbarcode = 0, gated = 0, skipped_invalid = 0;
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
	size_t pairs() const { return pair_ok.size(); }
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
	void* ptrs[] = { pb.codes, pb.nmask, pb.woff, pb.len, pb.cls, pb.pair_ok, pb.barcode_id };
	for (void* p : ptrs)
		if (p)
			a.release(p);
	pb.codes = nullptr, pb.nmask = nullptr, pb.woff = nullptr, pb.len = nullptr, pb.cls = nullptr,
	pb.pair_ok = nullptr, pb.barcode_id = nullptr, pb.cap_words = 0, pb.cap_reads = 0;
}

inline bool
packed_reserve(PackedBatch& pb, size_t words, size_t reads, const HostAllocator& a)
{
	if (words > pb.cap_words) {
		if (pb.codes)
			a.release(pb.codes);
		if (pb.nmask)
			a.release(pb.nmask);
		const size_t cap = words + words / 8 + 64;
		pb.codes = (uint64_t*)a.alloc(cap * sizeof(uint64_t));
		pb.nmask = (uint32_t*)a.alloc(cap * sizeof(uint32_t));
		pb.cap_words = cap;
		if (!pb.codes || !pb.nmask)
			return false;
	}
	if (reads > pb.cap_reads) {
		void* ptrs[] = { pb.woff, pb.len, pb.cls, pb.pair_ok, pb.barcode_id };
		for (void* p : ptrs)
			if (p)
				a.release(p);
		const size_t cap = reads + reads / 8 + 64;
		pb.woff = (uint64_t*)a.alloc((cap + 1) * sizeof(uint64_t));
		pb.len = (uint32_t*)a.alloc(cap * sizeof(uint32_t));
		pb.cls = (uint8_t*)a.alloc(cap);
		pb.pair_ok = (uint8_t*)a.alloc(cap / 2 + 1);
		pb.barcode_id = (uint32_t*)a.alloc((cap / 2 + 1) * sizeof(uint32_t));
		pb.cap_reads = cap;
		if (!pb.woff || !pb.len || !pb.cls || !pb.pair_ok || !pb.barcode_id)
			return false;
	}
	return true;
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
	rb.bases.push_back('\0');
	const int rc = arks_pack_reads_host(rb.bases.data(), rb.off.data(), rb.len.data(), pb.woff, n, pb.codes, pb.nmask,
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
    const std::function<bool(RawBatch&)>& recycled = nullptr) // hands back a used batch (its buffers are warm)
{
	std::unordered_map<std::string_view, uint32_t> cache; // fused mode: this producer's view of the dictionary
	rd.keep_qual = false; // the mapping needs no base qualities: measure them, do not copy them
	RawBatch b;
	int64_t seq = 0;
	auto reset = [&](RawBatch& x) {
		// a recycled batch keeps its capacity: a fresh 80 MB of bases costs an mmap and 20 k page faults
		if (recycled && recycled(x)) {
			x.bases.clear(), x.off.clear(), x.len.clear(), x.pair_ok.clear(), x.barcode_id.clear(), x.messages.clear();
			x.fc = FileCounters();
			x.last = false;
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
	size_t count = 0;
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
	  , batch_pairs_(batch_pairs)
	  , verbose_(verbose)
	  , alloc_(std::move(alloc))
	  , raw_q_(4)
	  , packed_q_(4)
	  , free_q_(1u << 20)
	{
		const unsigned nf = (unsigned)readers_.size();
		n_producers_ = std::max(1u, std::min(nf, std::max(1u, threads / 2)));
		n_packers_ = std::max(1u, threads > n_producers_ ? threads - n_producers_ : 1u);
		n_buffers_ = n_packers_ + 3;
	}

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
				for (;;) {
					size_t f;
					{
						std::lock_guard<std::mutex> lk(file_m);
						if (next_file >= readers_.size())
							return;
						f = next_file++;
					}
					produce_file(*readers_[f], (int)f, dict_, dict_ ? nullptr : &dynamic_,
					             dict_ ? nullptr : &prepass_[f], batch_pairs_, verbose_,
					             [&](RawBatch&& rb) { raw_q_.push(std::move(rb)); },
					             [&](RawBatch& out) { return raw_free_.try_pop(out); });
				}
			});
		for (unsigned t = 0; t < n_packers_; ++t)
			packers.emplace_back([&] {
				RawBatch rb;
				while (raw_q_.pop(rb)) {
					PackedBatch* pb = nullptr;
					if (!free_q_.pop_newest(pb))
						return;
					const int rc = pack_batch(rb, *pb, alloc_);
					if (rc != ARKS_OK) {
						std::lock_guard<std::mutex> lk(err_m);
						if (first_err == ARKS_OK)
							first_err = rc;
					}
					packed_q_.push(pb);
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
		if (finish)
			finish();
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
	std::vector<PrepassInfo> prepass_;
	long batch_pairs_;
	bool verbose_;
	HostAllocator alloc_;
	unsigned n_producers_ = 1, n_packers_ = 1, n_buffers_ = 4;
	BoundedQueue<RawBatch> raw_q_, raw_free_{ 8 };
	BoundedQueue<PackedBatch*> packed_q_, free_q_;
};

} // namespace arks_host
