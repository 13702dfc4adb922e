// bgzf.hpp -- parallel inflate of BGZF (blocked gzip: bgzip, htslib) input for the read ingest.
//
// A plain gzip stream is one deflate stream: it inflates on one core (~0.4 GB/s of text), which is
// what bounds `arcs --arks` on .gz input once the mapping runs on the GPU.  BGZF files are a sequence of
// independent gzip members of <= 64 KiB, each announcing its compressed size in a 'BC' extra field
// (SAM/BAM specification, section 4.1): a reader thread cuts the file into members, worker threads
// inflate them (raw deflate + CRC-32 + length check, as gzread would), the consumer takes them back in
// file order.  The bytes delivered are exactly those gzread would deliver.
#pragma once

#include "crc32_fold.hpp"

#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace arks_host {

// true when the 18 bytes start a BGZF member; *bsize = total member size
inline bool
bgzf_header(const unsigned char* h, size_t n, unsigned* bsize)
{
	if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
		return false;
	const unsigned xlen = h[10] | (h[11] << 8);
	if (xlen < 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0)
		return false;
	*bsize = (h[16] | (h[17] << 8)) + 1u;
	return *bsize >= 26;
}

// One raw deflate stream, whole in memory, into a buffer of known size: what a BGZF member is.  libdeflate's
// whole-buffer decoder where the system has the library (about twice zlib's speed on FASTQ text; found at run
// time the way htslib-based tools find it -- ARKS_ZLIB_INFLATE=1 leaves it out), zlib's inflate otherwise.
// One instance per thread.
class RawInflater
{
  public:
	RawInflater()
	{
		if (const Lib* l = lib())
			d_ = l->alloc();
		if (!d_) {
			std::memset(&z_, 0, sizeof z_);
			z_ok_ = inflateInit2(&z_, -15) == Z_OK;
		}
	}
	~RawInflater()
	{
		if (d_)
			lib()->release(d_);
		else if (z_ok_)
			inflateEnd(&z_);
	}
	RawInflater(const RawInflater&) = delete;
	RawInflater& operator=(const RawInflater&) = delete;
	static bool accelerated() { return lib() != nullptr; }

	// the stream in[0, n) into out[0, cap): the number of bytes it holds, or -1 (damaged, or longer than cap)
	long run(const unsigned char* in, size_t n, unsigned char* out, size_t cap)
	{
		if (d_) {
			size_t got = 0;
			return lib()->run(d_, in, n, out, cap, &got) == 0 ? (long)got : -1;
		}
		if (!z_ok_)
			return -1;
		inflateReset(&z_);
		z_.next_in = const_cast<unsigned char*>(in);
		z_.avail_in = (unsigned)n;
		z_.next_out = out;
		z_.avail_out = (unsigned)cap;
		const int rc = inflate(&z_, Z_FINISH);
		return rc == Z_STREAM_END ? (long)(cap - z_.avail_out) : -1;
	}

  private:
	struct Lib
	{
		void* (*alloc)();
		int (*run)(void*, const void*, size_t, void*, size_t, size_t*);
		void (*release)(void*);
	};
	static const Lib* lib()
	{
		static const Lib* found = []() -> const Lib* {
			if (std::getenv("ARKS_ZLIB_INFLATE"))
				return nullptr;
			void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
			if (!h)
				return nullptr;
			static Lib l;
			l.alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
			l.run = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
			l.release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
			return l.alloc && l.run && l.release ? &l : nullptr;
		}();
		return found;
	}
	void* d_ = nullptr;
	z_stream z_;
	bool z_ok_ = false;
};

// header, raw deflate stream and trailer of the member m[0, len): inflates it into out[0, ISIZE) and checks
// length and CRC-32 as gzread would.  False on any damage.
inline bool
bgzf_inflate_member(RawInflater& inf, const unsigned char* m, size_t len, unsigned char* out, size_t cap, uint32_t* out_len)
{
	const unsigned xlen = m[10] | (m[11] << 8);
	const unsigned hdr = 12 + xlen;
	if (len < hdr + 8)
		return false;
	const uint32_t want_crc = m[len - 8] | (m[len - 7] << 8) | (m[len - 6] << 16) | ((uint32_t)m[len - 5] << 24);
	const uint32_t want_len = m[len - 4] | (m[len - 3] << 8) | (m[len - 2] << 16) | ((uint32_t)m[len - 1] << 24);
	if (want_len > cap)
		return false;
	const long got = inf.run(m + hdr, len - hdr - 8, out, want_len);
	*out_len = got < 0 ? 0 : (uint32_t)got;
	return got == (long)want_len && crc32_fast(0u, out, want_len) == want_crc;
}

// a loop over [0, n) that its runner may spread over threads (ingest.hpp: HelpDesk::parallel_for)
using ParallelFor = std::function<void(size_t, const std::function<void(size_t)>&)>;

// A BGZF file taken a stretch of text at a time (the fast path of the read ingest): the compressed file is
// mapped, the members' headers and trailers say where every member's text goes before any of it is inflated
// (BSIZE, ISIZE), so `threads` threads inflate a stretch's members straight into one buffer -- no hand-over of
// 64 KiB pieces, no copy.  A damaged member ends the stretches before it: the caller hands the file over to
// the sequential reader (BgzfReader) at offset(), which meets the damage the way it always did.
class BgzfStretches
{
  public:
	explicit BgzfStretches(FILE* f)
	{
		struct stat st;
		if (::fstat(fileno(f), &st) != 0 || st.st_size <= 0)
			return;
		void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
		if (m == MAP_FAILED)
			return;
		(void)::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
		map_ = (const unsigned char*)m;
		size_ = (size_t)st.st_size;
	}
	~BgzfStretches()
	{
		if (map_)
			(void)::munmap(const_cast<unsigned char*>(map_), size_);
	}
	BgzfStretches(const BgzfStretches&) = delete;
	BgzfStretches& operator=(const BgzfStretches&) = delete;
	bool ok() const { return map_ != nullptr; }
	size_t offset() const { return at_; } // of the first member not delivered
	bool at_end() const { return done_; }

	// the members from offset() on whose text adds up to at least `target` bytes (fewer at the end of the file or
	// before something that is not a BGZF member): returns the size of their text
	size_t plan(size_t target)
	{
		members_.clear();
		size_t out = 0, at = at_;
		while (out < target) {
			unsigned bsize = 0;
			if (size_ - at < 18 || !bgzf_header(map_ + at, size_ - at, &bsize) || size_ - at < bsize) {
				done_ = true; // the end, or damage: the sequential reader finds out which
				break;
			}
			const unsigned char* t = map_ + at + bsize - 4;
			const uint32_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
			if (isize > (1u << 16)) {
				done_ = true;
				break;
			}
			members_.push_back(Member{ at, bsize, out, isize });
			out += isize;
			at += bsize;
		}
		planned_end_ = at;
		return out;
	}
	// inflates the planned members into dst, a chunk of members per call of the loop `pf` runs (on however many
	// threads it has); returns the bytes of text that are good (all of them, or those before the first damaged
	// member) and moves offset() behind the members they came from
	size_t inflate(unsigned char* dst, const ParallelFor& pf)
	{
		const size_t n = members_.size();
		const size_t per = 16; // members per chunk: ~1 MB of text
		std::atomic<size_t> first_bad{ n };
		pf((n + per - 1) / per, [&](size_t c) {
			static thread_local RawInflater inf;
			const size_t lo = c * per, hi = std::min(n, lo + per);
			if (lo >= first_bad.load(std::memory_order_relaxed))
				return;
			for (size_t i = lo; i < hi; ++i) {
				const Member& m = members_[i];
				uint32_t got = 0;
				if (!bgzf_inflate_member(inf, map_ + m.in, m.in_len, dst + m.out, m.out_len, &got)) {
					size_t cur = first_bad.load();
					while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {
					}
					return;
				}
			}
		});
		const size_t bad = first_bad.load();
		if (bad < n) {
			done_ = true;
			at_ = members_[bad].in;
			return members_[bad].out;
		}
		at_ = planned_end_;
		return n ? members_[n - 1].out + members_[n - 1].out_len : 0;
	}

  private:
	struct Member
	{
		size_t in;
		uint32_t in_len;
		size_t out;
		uint32_t out_len;
	};
	const unsigned char* map_ = nullptr;
	size_t size_ = 0, at_ = 0, planned_end_ = 0;
	bool done_ = false;
	std::vector<Member> members_;
};

class BgzfReader
{
  public:
	BgzfReader(FILE* f, unsigned workers)
	  : f_(f)
	  , slots_(4 * (workers ? workers : 1) + 4)
	{
		reader_ = std::thread([this] { read_loop(); });
		for (unsigned i = 0; i < (workers ? workers : 1); ++i)
			workers_.emplace_back([this] { work_loop(); });
	}
	~BgzfReader()
	{
		{
			std::lock_guard<std::mutex> lk(m_);
			stop_ = true;
		}
		cv_free_.notify_all();
		cv_work_.notify_all();
		cv_done_.notify_all();
		reader_.join();
		for (auto& t : workers_)
			t.join();
		fclose(f_);
	}
	BgzfReader(const BgzfReader&) = delete;
	BgzfReader& operator=(const BgzfReader&) = delete;

	// up to cap bytes of the inflated stream; 0 at the end, -1 on a corrupt file
	int read(unsigned char* dst, int cap)
	{
		for (;;) {
			Slot& s = slots_[next_out_ % slots_.size()];
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_done_.wait(lk, [&] { return stop_ || (s.state == DONE && s.seq == next_out_) || (eof_ && next_out_ == n_read_); });
				if (stop_)
					return -1;
				if (!(s.state == DONE && s.seq == next_out_))
					return failed_ ? -1 : 0; // every member delivered
			}
			if (s.bad)
				return -1;
			const int left = (int)s.out_len - (int)s.out_pos;
			if (left > 0) {
				const int n = left < cap ? left : cap;
				std::memcpy(dst, s.out.data() + s.out_pos, (size_t)n);
				s.out_pos += (uint32_t)n;
				return n;
			}
			{
				std::lock_guard<std::mutex> lk(m_);
				s.state = EMPTY;
				next_out_++;
			}
			cv_free_.notify_all();
		}
	}

  private:
	enum State { EMPTY, LOADED, BUSY, DONE };
	struct Slot
	{
		State state = EMPTY;
		uint64_t seq = 0;
		std::vector<unsigned char> in = std::vector<unsigned char>(1 << 16);
		std::vector<unsigned char> out = std::vector<unsigned char>(1 << 16);
		uint32_t in_len = 0, out_len = 0, out_pos = 0;
		bool bad = false;
	};

	void read_loop()
	{
		uint64_t seq = 0;
		for (;;) {
			Slot& s = slots_[seq % slots_.size()];
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_free_.wait(lk, [&] { return stop_ || s.state == EMPTY; });
				if (stop_)
					return;
			}
			unsigned char h[18];
			const size_t got = fread(h, 1, sizeof h, f_);
			unsigned bsize = 0;
			bool ok = got == sizeof h && bgzf_header(h, got, &bsize);
			if (ok) {
				std::memcpy(s.in.data(), h, sizeof h);
				ok = fread(s.in.data() + sizeof h, 1, bsize - sizeof h, f_) == bsize - sizeof h;
			}
			std::lock_guard<std::mutex> lk(m_);
			if (!ok) {
				failed_ = got != 0; // a clean end has no bytes left; anything else is a damaged file
				eof_ = true;
				n_read_ = seq;
				cv_done_.notify_all();
				return;
			}
			s.in_len = bsize;
			s.seq = seq++;
			s.out_pos = 0;
			s.bad = false;
			s.state = LOADED;
			cv_work_.notify_one();
		}
	}

	void work_loop()
	{
		RawInflater inf;
		for (;;) {
			Slot* s = nullptr;
			{
				std::unique_lock<std::mutex> lk(m_);
				// members are handed out in file order: the consumer waits for the oldest one
				cv_work_.wait(lk, [&] {
					if (stop_)
						return true;
					Slot& c = slots_[next_work_ % slots_.size()];
					if (c.state == LOADED && c.seq == next_work_) {
						s = &c;
						return true;
					}
					return false;
				});
				if (stop_)
					break;
				s->state = BUSY;
				next_work_++;
				cv_work_.notify_one(); // the next member may be loaded already
			}
			const bool bad = !bgzf_inflate_member(inf, s->in.data(), s->in_len, s->out.data(), s->out.size(), &s->out_len);
			{
				std::lock_guard<std::mutex> lk(m_);
				s->bad = bad;
				s->state = DONE;
			}
			cv_done_.notify_all();
		}
	}

	FILE* f_;
	std::vector<Slot> slots_;
	std::thread reader_;
	std::vector<std::thread> workers_;
	std::mutex m_;
	std::condition_variable cv_free_, cv_work_, cv_done_;
	bool stop_ = false, eof_ = false, failed_ = false;
	uint64_t next_out_ = 0, next_work_ = 0, n_read_ = 0;
};

} // namespace arks_host
