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

#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
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
		z_stream z;
		std::memset(&z, 0, sizeof z);
		if (inflateInit2(&z, -15) != Z_OK)
			return;
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
			const unsigned char* in = s->in.data();
			const unsigned xlen = in[10] | (in[11] << 8);
			const unsigned hdr = 12 + xlen;
			bool bad = s->in_len < hdr + 8;
			if (!bad) {
				const uint32_t want_crc = in[s->in_len - 8] | (in[s->in_len - 7] << 8) | (in[s->in_len - 6] << 16) |
				                          ((uint32_t)in[s->in_len - 5] << 24);
				const uint32_t want_len = in[s->in_len - 4] | (in[s->in_len - 3] << 8) | (in[s->in_len - 2] << 16) |
				                          ((uint32_t)in[s->in_len - 1] << 24);
				inflateReset(&z);
				z.next_in = const_cast<unsigned char*>(in) + hdr;
				z.avail_in = s->in_len - hdr - 8;
				z.next_out = s->out.data();
				z.avail_out = (unsigned)s->out.size();
				const int rc = inflate(&z, Z_FINISH);
				s->out_len = (uint32_t)(s->out.size() - z.avail_out);
				bad = rc != Z_STREAM_END || s->out_len != want_len ||
				      crc32_fast(0u, s->out.data(), s->out_len) != want_crc;
			}
			{
				std::lock_guard<std::mutex> lk(m_);
				s->bad = bad;
				s->state = DONE;
			}
			cv_done_.notify_all();
		}
		inflateEnd(&z);
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
