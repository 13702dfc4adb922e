// seqio.hpp -- gz/plain FASTA/FASTQ record reader with the record semantics of Heng Li's kseq as
// the reference uses it (Arcs/kseq.h:146-186, instantiated over gzread, Arcs.cpp:187):
//   * a record starts at the next '>' or '@'; name = header up to the first white space, comment =
//     rest of the header line; sequence lines are concatenated until a line starting with '>', '+'
//     or '@'; FASTQ quality lines are read until they cover the sequence;
//   * next() returns the sequence length, -1 at end of file, -2 when the quality string is missing
//     or its length differs from the sequence's (the callers stop at any negative value).
// Written from that behaviour; no code of kseq.h is used.
#pragma once

#include "bgzf.hpp"
#include "fast_inflate.hpp"
#include "pgzip.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <string>
#include <vector>

namespace arks_host {

class SeqReader
{
  public:
	std::string name, comment, seq, qual;
	bool keep_qual = true; // false: quality lines are only measured (length check), `qual` stays empty

	// bgzf_workers > 0: a BGZF (bgzip) file is inflated by that many threads (bgzf.hpp); an ordinary gzip file
	// by fast_inflate.hpp; plain text and pipes go through zlib's gzread as the reference's kseq does
	explicit SeqReader(const char* path, unsigned bgzf_workers = 0)
	{
		// a seekable file that starts like a gzip member: BGZF goes to the inflate threads, any other gzip file
		// to the fast single-stream inflater (fast_inflate.hpp; ARKS_ZLIB_INFLATE=1 keeps zlib's)
		// (regular files only: looking at the first bytes of a pipe -- /dev/stdin in the arks-long pipeline --
		// would take them away from the reader that follows)
		struct stat st;
		FILE* f = (::stat(path, &st) == 0 && S_ISREG(st.st_mode)) ? std::fopen(path, "rb") : nullptr;
		if (f) {
			unsigned char h[18];
			unsigned bsize = 0;
			const size_t got = std::fread(h, 1, sizeof h, f);
			const bool gz = got >= 3 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8;
			if (gz && std::fseek(f, 0, SEEK_SET) == 0) {
				if (bgzf_workers > 0 && bgzf_header(h, got, &bsize)) {
					bgzf_file_ = f; // the inflate threads start with the first read, not for every open file
					bgzf_workers_ = bgzf_workers;
				} else if (!std::getenv("ARKS_ZLIB_INFLATE")) {
					fast_.reset(new GzInflater(f)); // owns the file
					// ARKS_INFLATE_THREAD=1: inflate on a thread of its own while the records are parsed.  Off by
					// default: on the build container it gained nothing (0.31 s against 0.30 s for 600 k records
					// -- the parser then reads the text out of another core's cache), the GPU host is unmeasured
					if (bgzf_workers > 0 && std::getenv("ARKS_INFLATE_THREAD"))
						ahead_.reset(new InflateAhead(std::move(fast_)));
				}
				else
					std::fclose(f);
			} else if (!gz) {
				// plain text in a regular file: read(2) straight into the caller's buffers (zlib's transparent
				// mode would copy every byte twice more)
				std::fclose(f);
				plain_fd_ = ::open(path, O_RDONLY);
			} else
				std::fclose(f);
		}
		if (!bgzf_file_ && !fast_ && !ahead_ && plain_fd_ < 0) {
			fp_ = gzopen(path, "r");
			if (fp_)
				gzbuffer(fp_, 1u << 20);
		}
	}
	~SeqReader()
	{
		if (fp_)
			gzclose(fp_);
		if (map_)
			(void)::munmap(map_, map_size_);
		if (plain_fd_ >= 0)
			::close(plain_fd_);
		if (bgzf_file_ && !bgzf_)
			std::fclose(bgzf_file_);
	}
	SeqReader(const SeqReader&) = delete;
	SeqReader& operator=(const SeqReader&) = delete;
	bool ok() const { return fp_ != nullptr || bgzf_file_ != nullptr || fast_ != nullptr || ahead_ != nullptr || plain_fd_ >= 0; }

	// The raw (inflated) text instead of records, for a caller that splits it itself (ingest.hpp): up to
	// `cap` bytes, 0 at the end of the stream.  May be mixed with next(): both consume the same stream.
	int read_raw(unsigned char* dst, int cap)
	{
		int got = 0;
		while (got < cap) {
			if (begin_ >= end_ && plain_fd_ >= 0 && pushback_pos_ >= pushback_.size() && !eof_) {
				const ssize_t r = ::read(plain_fd_, dst + got, (size_t)(cap - got)); // no detour through buf_
				if (r <= 0) {
					eof_ = true;
					break;
				}
				got += (int)r;
				continue;
			}
			if (begin_ >= end_) {
				if (got > 0 || !fill())
					break; // (a short read once something was delivered: the next call fetches more)
			}
			const int n = std::min(cap - got, end_ - begin_);
			std::memcpy(dst + got, buf_ + begin_, (size_t)n);
			begin_ += n;
			got += n;
		}
		return got;
	}
	// A plain text file as one read-only mapping (no copy at all: the caller splits and parses it in place);
	// nullptr for any other source, or once something was read.  The mapping lives as long as the reader.
	const char* map_plain(size_t* size)
	{
		if (plain_fd_ < 0 || begin_ < end_ || eof_ || map_ || !pushback_.empty())
			return map_ ? (const char*)map_ : nullptr;
		struct stat st;
		if (::fstat(plain_fd_, &st) != 0 || st.st_size <= 0 || ::lseek(plain_fd_, 0, SEEK_CUR) != 0)
			return nullptr;
		void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, plain_fd_, 0);
		if (m == MAP_FAILED)
			return nullptr;
		(void)::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
		map_ = m;
		map_size_ = (size_t)st.st_size;
		*size = map_size_;
		return (const char*)map_;
	}
	// after map_plain: the stream (next(), read_raw()) continues at this offset of the file
	void continue_at(size_t offset)
	{
		(void)::lseek(plain_fd_, (off_t)offset, SEEK_SET);
		begin_ = end_ = 0;
		eof_ = false;
	}
	// A BGZF file nothing was read from yet, as stretches of text inflated by the caller's threads (bgzf.hpp);
	// nullptr for any other source.  bgzf_continue_at(s->offset()) hands the file back to this reader.
	std::unique_ptr<BgzfStretches> bgzf_stretches()
	{
		if (!bgzf_file_ || bgzf_ || begin_ < end_ || eof_ || !pushback_.empty())
			return nullptr;
		std::unique_ptr<BgzfStretches> s(new BgzfStretches(bgzf_file_));
		if (!s->ok())
			s.reset();
		return s;
	}
	void bgzf_continue_at(size_t offset) { (void)std::fseek(bgzf_file_, (long)offset, SEEK_SET); }
	// An ordinary gzip file nothing was read from yet, as stretches of text decoded by the caller's threads
	// (pgzip.hpp); nullptr for any other source (ARKS_NO_PGZIP=1: never).  gz_continue(*s) hands the stream
	// back to this reader's own inflater, which goes on where the stretches ended.
	std::unique_ptr<GzStretches> gz_stretches(size_t chunk_bytes, unsigned chunks_per_stretch)
	{
		if (!fast_ || begin_ < end_ || eof_ || !pushback_.empty() || gz_handed_out_ || std::getenv("ARKS_NO_PGZIP"))
			return nullptr;
		std::unique_ptr<GzStretches> s(new GzStretches(fast_->file(), chunk_bytes, chunks_per_stretch));
		if (!s->ok())
			s.reset();
		else
			gz_handed_out_ = true;
		return s;
	}
	void gz_continue(const GzStretches& s)
	{
		if (!s.started())
			return; // nothing was decoded there: the inflater starts at the head of the file as always
		const GzResumePoint& r = s.resume();
		if (!(r.member_start ? fast_->resume_member(r.bit >> 3) : fast_->resume(r.bit, r.window.data(), r.window.size(), r.crc, r.member_out))) {
			failed_ = true;
			eof_ = true;
		}
	}
	// gives `n` bytes back: they are the next the stream delivers (the text a splitting caller read ahead
	// of the point where it hands the stream over to next())
	void unread(const unsigned char* p, size_t n)
	{
		std::vector<unsigned char> rest(p, p + n);
		rest.insert(rest.end(), buf_ + begin_, buf_ + end_); // what the reader itself holds comes after
		rest.insert(rest.end(), pushback_.begin() + (std::ptrdiff_t)pushback_pos_, pushback_.end());
		pushback_.swap(rest);
		pushback_pos_ = 0;
		begin_ = end_ = 0;
		if (!pushback_.empty())
			eof_ = false;
	}
	bool parallel_inflate() const { return bgzf_file_ != nullptr; }
	// one stream that only one thread can read (an ordinary gzip file, a pipe): neither mapped nor inflated in stretches
	bool serial_source() const { return bgzf_file_ == nullptr && plain_fd_ < 0 && !fast_; }
	// an ordinary gzip file in a regular file: one thread's work, or several threads' through pgzip.hpp
	bool splittable_gzip() const { return fast_ != nullptr; }
	// the stream ended on an inflate error (damaged or truncated compressed input), not at its end
	bool failed() const { return failed_; }

	int next()
	{
		int c;
		if (last_ == 0) { // find the next header
			while ((c = getc()) != -1 && c != '>' && c != '@') {
			}
			if (c == -1)
				return -1;
			last_ = c;
		}
		comment.clear();
		seq.clear();
		qual.clear();
		name.clear();
		// name: up to white space
		bool got = false;
		int term = -1;
		while ((c = getc()) != -1) {
			got = true;
			if (c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r') {
				term = c;
				break;
			}
			name.push_back((char)c);
		}
		if (!got)
			return -1;
		if (term != '\n' && term != -1)
			read_line(comment, false);
		// sequence
		while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n')
				continue;
			seq.push_back((char)c);
			read_line(seq, true);
		}
		if (c == '>' || c == '@')
			last_ = c;
		if (c != '+')
			return (int)seq.size();
		if (!skip_line())
			return -2;
		size_t qlen = 0;
		if (keep_qual) {
			while (read_line(qual, true) >= 0 && qual.size() < seq.size()) {
			}
			qlen = qual.size();
		} else {
			while (measure_line(qlen) && qlen < seq.size()) {
			}
		}
		last_ = 0;
		if (seq.size() != qlen)
			return -2;
		return (int)seq.size();
	}

  private:
	gzFile fp_ = nullptr;
	int plain_fd_ = -1;
	void* map_ = nullptr;
	size_t map_size_ = 0;
	std::unique_ptr<BgzfReader> bgzf_;
	std::unique_ptr<GzInflater> fast_;
	std::unique_ptr<InflateAhead> ahead_;
	FILE* bgzf_file_ = nullptr;
	unsigned bgzf_workers_ = 0;
	unsigned char buf_[1 << 18];
	int begin_ = 0, end_ = 0;
	bool eof_ = false, failed_ = false, gz_handed_out_ = false;
	int last_ = 0;
	std::vector<unsigned char> pushback_;
	size_t pushback_pos_ = 0;

	bool fill()
	{
		if (pushback_pos_ < pushback_.size()) {
			const size_t n = std::min(sizeof buf_, pushback_.size() - pushback_pos_);
			std::memcpy(buf_, pushback_.data() + pushback_pos_, n);
			pushback_pos_ += n;
			begin_ = 0;
			end_ = (int)n;
			if (pushback_pos_ == pushback_.size()) {
				pushback_.clear();
				pushback_pos_ = 0;
			}
			return true;
		}
		if (eof_ || !ok())
			return false;
		begin_ = 0;
		if (bgzf_file_ && !bgzf_)
			bgzf_.reset(new BgzfReader(bgzf_file_, bgzf_workers_)); // owns the file from here on
		end_ = bgzf_ ? bgzf_->read(buf_, (int)sizeof buf_)
		             : ahead_ ? ahead_->read(buf_, (int)sizeof buf_)
		             : fast_ ? fast_->read(buf_, (int)sizeof buf_)
		             : plain_fd_ >= 0 ? (int)::read(plain_fd_, buf_, sizeof buf_) : gzread(fp_, buf_, sizeof buf_);
		if (end_ <= 0) {
			// a negative count is damage (a CRC or length mismatch, a truncated member, a BGZF file that goes
			// on as something else), not the end: zlib's gzread tells the reference's kseq the same way, and
			// the reference reads on to what it takes for the end of the file -- so does this reader, but it
			// remembers (failed()) so that the front end can say so
			if (end_ < 0)
				failed_ = true;
			end_ = 0;
			eof_ = true;
			return false;
		}
		return true;
	}

	int getc()
	{
		if (begin_ >= end_ && !fill())
			return -1;
		return buf_[begin_++];
	}

	// consumes up to and including the next newline; false when the stream ends first
	bool skip_line()
	{
		for (;;) {
			if (begin_ >= end_ && !fill())
				return false;
			const void* nl = std::memchr(buf_ + begin_, '\n', (size_t)(end_ - begin_));
			if (nl) {
				begin_ = (int)((const unsigned char*)nl - buf_) + 1;
				return true;
			}
			begin_ = end_;
		}
	}

	// read_line(s, append = true) without the copy: `total` grows by the line's length under the same
	// carriage-return rule (applied to the accumulated text); false when the stream is at its end
	bool measure_line(size_t& total)
	{
		bool got = false;
		unsigned char last = 0;
		for (;;) {
			if (begin_ >= end_ && !fill())
				break;
			got = true;
			const unsigned char* from = buf_ + begin_;
			const void* nl = std::memchr(from, '\n', (size_t)(end_ - begin_));
			const unsigned char* to = nl ? (const unsigned char*)nl : buf_ + end_;
			if (to > from) {
				total += (size_t)(to - from);
				last = to[-1];
			}
			begin_ = nl ? (int)(to - buf_) + 1 : end_;
			if (nl)
				break;
		}
		if (!got)
			return false;
		if (total > 1 && last == '\r')
			total--;
		return true;
	}

	// appends (or assigns) the rest of the current line; returns the string length, or -1 when
	// nothing could be read because the stream is at its end
	int read_line(std::string& s, bool append)
	{
		if (!append)
			s.clear();
		bool got = false;
		for (;;) {
			if (begin_ >= end_ && !fill())
				break;
			got = true;
			const unsigned char* from = buf_ + begin_;
			const void* nl = std::memchr(from, '\n', (size_t)(end_ - begin_));
			if (nl) {
				s.append((const char*)from, (const char*)nl);
				begin_ = (int)((const unsigned char*)nl - buf_) + 1;
				break;
			}
			s.append((const char*)from, (const char*)(buf_ + end_));
			begin_ = end_;
		}
		if (!got)
			return -1;
		if (s.size() > 1 && s.back() == '\r')
			s.pop_back();
		return (int)s.size();
	}
};

} // namespace arks_host
