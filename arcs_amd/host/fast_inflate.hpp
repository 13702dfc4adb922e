// fast_inflate.hpp -- a gzip reader for the read ingest that inflates about twice as fast as zlib's gzread.
//
// Ordinary .gz input is ONE deflate stream: it cannot be split over threads (bgzf.hpp does that for
// bgzip'ed files), so the speed of a single inflate loop bounds `arcs --arks` on the usual reads.fq.gz
// once the mapping runs on the GPU.  zlib's inflate keeps its state machine byte-resumable; this one owns
// its input (a FILE) and can always fetch more, so the hot loop is the plain one: a 64-bit bit buffer
// refilled eight bytes at a time, one table lookup per literal / length / distance symbol (11-bit and
// 8-bit primary tables with subtables for the rare longer codes), matches copied eight bytes at a time.
// Same contract as gzread for what the ingest needs: the concatenation of all members' data, CRC-32 and
// length of every member checked (crc32_fold.hpp), -1 on a damaged stream.  RFC 1951 / RFC 1952.
#pragma once

#include "crc32_fold.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace arks_host {

class GzInflater
{
  public:
	explicit GzInflater(FILE* f)
	  : f_(f)
	  , in_(kInSize + 8)
	  , out_(kHistory + kChunk + kSlack)
	{}
	~GzInflater()
	{
		if (f_)
			std::fclose(f_);
	}
	GzInflater(const GzInflater&) = delete;
	GzInflater& operator=(const GzInflater&) = delete;

	FILE* file() const { return f_; }
	// Goes on in the middle of the first member (nothing was read through this object yet): the block that
	// starts at bit `bit` of the file is the next to decode, `window` is the text in front of it (its last
	// 32 KiB, or all of it if there is less), `crc` / `member_out` the CRC-32 and length of the member's text so
	// far.  For pgzip.hpp, which decodes the blocks in front of that point on several threads.
	bool resume(size_t bit, const unsigned char* window, size_t window_len, uint32_t crc, uint64_t member_out)
	{
		if (std::fseek(f_, (long)(bit >> 3), SEEK_SET) != 0)
			return false;
		in_pos_ = in_end_ = 0;
		in_eof_ = false;
		bitbuf_ = 0;
		bitcnt_ = 0;
		window_len = std::min(window_len, kHistory);
		if (window_len)
			std::memcpy(out_.data() + kHistory - window_len, window, window_len);
		out_pos_ = out_read_ = kHistory;
		crc_ = crc;
		member_out_ = member_out;
		first_member_ = false;
		state_ = BLOCK_START;
		if (bit & 7) {
			if (!need((int)(bit & 7)))
				return false;
			drop((int)(bit & 7));
		}
		return true;
	}

	// Goes on at byte `offset` of the file, where a member ended: another member's header, the end of the file
	// or something else (which ends the stream as it does behind any member but the first).
	bool resume_member(size_t offset)
	{
		if (std::fseek(f_, (long)offset, SEEK_SET) != 0)
			return false;
		in_pos_ = in_end_ = 0;
		in_eof_ = false;
		bitbuf_ = 0;
		bitcnt_ = 0;
		out_pos_ = out_read_ = kHistory;
		crc_ = 0;
		member_out_ = 0;
		first_member_ = false;
		state_ = HEADER;
		return true;
	}

	// up to cap bytes of the inflated stream; 0 at the end, -1 on a damaged file
	int read(unsigned char* dst, int cap)
	{
		for (;;) {
			if (out_read_ < out_pos_) {
				const size_t n = std::min<size_t>((size_t)cap, out_pos_ - out_read_);
				std::memcpy(dst, out_.data() + out_read_, n);
				out_read_ += n;
				return (int)n;
			}
			if (state_ == DONE)
				return 0;
			if (state_ == FAILED)
				return -1;
			if (out_pos_ > kHistory + kChunk / 2) { // everything handed out: keep the window, start over
				std::memmove(out_.data(), out_.data() + out_pos_ - kHistory, kHistory);
				out_pos_ = out_read_ = kHistory;
			}
			const size_t before = out_pos_;
			if (!produce())
				state_ = FAILED;
			if (out_pos_ > before) // whatever was produced before a failure is still delivered, as gzread does
				crc_ = crc32_fast(crc_, out_.data() + before, out_pos_ - before);
			if (state_ == TRAILER_PENDING && !finish_member())
				state_ = FAILED;
		}
	}

  private:
	static constexpr size_t kInSize = 1u << 20, kHistory = 32768, kChunk = 1u << 19, kSlack = 320; // slack: one match + the literals and word copies around it
	static constexpr int kLitBits = 11, kDistBits = 8;
	enum State { HEADER, BLOCK_START, STORED, HUFFMAN, TRAILER_PENDING, DONE, FAILED };
	// table entry: bits 0-7 code length (bits to drop), 8-12 extra bits (or subtable bits), 13-15 kind,
	// 16-31 literal / base value / subtable offset
	enum Kind : uint32_t { LITERAL = 0, BASE = 1, END_OF_BLOCK = 2, SUBTABLE = 3, INVALID = 4 };
	static uint32_t entry(Kind k, uint32_t value, uint32_t extra, uint32_t len)
	{
		return (value << 16) | ((uint32_t)k << 13) | (extra << 8) | len;
	}

	FILE* f_;
	std::vector<unsigned char> in_, out_;
	size_t in_pos_ = 0, in_end_ = 0;
	bool in_eof_ = false;
	size_t out_pos_ = kHistory, out_read_ = kHistory;
	uint64_t bitbuf_ = 0;
	int bitcnt_ = 0;
	State state_ = HEADER;
	bool last_block_ = false, first_member_ = true;
	uint32_t stored_left_ = 0, crc_ = 0;
	uint64_t member_out_ = 0;
	uint32_t lit_[(1 << kLitBits) + 288 * 16], dist_[(1 << kDistBits) + 32 * 128];

	// ---- input -------------------------------------------------------------------------------------
	void refill_input()
	{
		if (in_eof_)
			return;
		const size_t left = in_end_ - in_pos_;
		if (left && in_pos_)
			std::memmove(in_.data(), in_.data() + in_pos_, left);
		in_pos_ = 0;
		in_end_ = left;
		const size_t got = std::fread(in_.data() + in_end_, 1, kInSize - in_end_, f_);
		in_end_ += got;
		if (got == 0)
			in_eof_ = true;
	}
	inline void refill_bits()
	{
		if (in_end_ - in_pos_ >= 8) {
			uint64_t v;
			std::memcpy(&v, in_.data() + in_pos_, 8); // little endian host (x86-64)
			bitbuf_ |= v << bitcnt_;
			const int n = (63 - bitcnt_) >> 3;
			in_pos_ += (size_t)n;
			bitcnt_ += n * 8;
		} else
			while (bitcnt_ <= 56 && in_pos_ < in_end_) {
				bitbuf_ |= (uint64_t)in_[in_pos_++] << bitcnt_;
				bitcnt_ += 8;
			}
	}
	// n <= 32 bits, refilling as needed; false when the input ends first
	bool need(int n)
	{
		if (bitcnt_ >= n)
			return true;
		if (in_end_ - in_pos_ < 8)
			refill_input();
		refill_bits();
		return bitcnt_ >= n;
	}
	uint32_t peek(int n) const { return (uint32_t)(bitbuf_ & ((1ull << n) - 1)); }
	void drop(int n)
	{
		bitbuf_ >>= n;
		bitcnt_ -= n;
	}
	bool bits(int n, uint32_t* v)
	{
		if (!need(n))
			return false;
		*v = peek(n);
		drop(n);
		return true;
	}
	// one byte of the byte-aligned parts (gzip header and trailer, stored blocks): first what is still in
	// the bit buffer
	int byte()
	{
		uint32_t v;
		return bits(8, &v) ? (int)v : -1;
	}

	// ---- gzip framing (RFC 1952) -------------------------------------------------------------------
	bool read_header()
	{
		if (bitcnt_ == 0 && in_pos_ == in_end_) {
			refill_input();
			if (in_pos_ == in_end_) { // clean end of the file
				state_ = first_member_ ? FAILED : DONE;
				return !first_member_;
			}
		}
		const int m0 = byte(), m1 = byte();
		if (m0 != 0x1f || m1 != 0x8b) {
			if (first_member_)
				return false;
			state_ = DONE; // zlib ignores what follows the last member when it is not another member
			return true;
		}
		const int cm = byte(), flg = byte();
		if (cm != 8 || flg < 0 || (flg & 0xe0))
			return false;
		for (int i = 0; i < 6; ++i) // MTIME, XFL, OS
			if (byte() < 0)
				return false;
		if (flg & 4) { // FEXTRA
			const int lo = byte(), hi = byte();
			if (lo < 0 || hi < 0)
				return false;
			for (int i = 0; i < (lo | (hi << 8)); ++i)
				if (byte() < 0)
					return false;
		}
		for (int bit : { 8, 16 }) // FNAME, FCOMMENT: zero-terminated
			if (flg & bit) {
				int c;
				while ((c = byte()) > 0) {
				}
				if (c < 0)
					return false;
			}
		if ((flg & 2) && (byte() < 0 || byte() < 0)) // FHCRC
			return false;
		first_member_ = false;
		crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
		member_out_ = 0;
		state_ = BLOCK_START;
		return true;
	}
	bool finish_member()
	{
		drop(bitcnt_ & 7); // the deflate stream ends inside a byte
		uint32_t v[8];
		for (int i = 0; i < 8; ++i) {
			const int b = byte();
			if (b < 0)
				return false;
			v[i] = (uint32_t)b;
		}
		const uint32_t want_crc = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
		const uint32_t want_len = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
		if (want_crc != crc_ || want_len != (uint32_t)member_out_)
			return false;
		state_ = HEADER;
		return true;
	}

	// ---- Huffman tables ----------------------------------------------------------------------------
	static uint32_t reverse(uint32_t code, int len)
	{
		uint32_t r = 0;
		for (int i = 0; i < len; ++i)
			r |= ((code >> i) & 1u) << (len - 1 - i);
		return r;
	}
	// canonical code of `lens` (RFC 1951 3.2.2) as a lookup table indexed by the next input bits; the
	// entry of symbol s is made by `make(s, code length)`.  false on an over-subscribed code.
	template <typename MakeEntry>
	static bool build_table(const uint8_t* lens, int n, int primary_bits, uint32_t* table, size_t table_size, MakeEntry make)
	{
		int count[16] = { 0 };
		for (int s = 0; s < n; ++s)
			count[lens[s]]++;
		count[0] = 0;
		uint32_t next[16], code = 0;
		long space = 1; // Kraft sum check
		for (int l = 1; l <= 15; ++l) {
			code = (code + (uint32_t)count[l - 1]) << 1;
			next[l] = code;
			space = (space << 1) - count[l];
			if (space < 0)
				return false;
		}
		const size_t primary = (size_t)1 << primary_bits;
		for (size_t i = 0; i < table_size; ++i)
			table[i] = entry(INVALID, 0, 0, 0);
		// subtable sizes: the longest code under each primary prefix
		uint8_t sub_bits[1 << kLitBits];
		std::memset(sub_bits, 0, primary);
		uint32_t codes[288];
		for (int s = 0; s < n; ++s) {
			const int l = lens[s];
			if (!l)
				continue;
			codes[s] = reverse(next[l]++, l);
			if (l > primary_bits) {
				uint8_t& b = sub_bits[codes[s] & (primary - 1)];
				if (l - primary_bits > b)
					b = (uint8_t)(l - primary_bits);
			}
		}
		size_t next_sub = primary;
		for (size_t p = 0; p < primary; ++p)
			if (sub_bits[p]) {
				if (next_sub + ((size_t)1 << sub_bits[p]) > table_size)
					return false;
				table[p] = entry(SUBTABLE, (uint32_t)next_sub, sub_bits[p], (uint32_t)primary_bits);
				next_sub += (size_t)1 << sub_bits[p];
			}
		for (int s = 0; s < n; ++s) {
			const int l = lens[s];
			if (!l)
				continue;
			if (l <= primary_bits) {
				const uint32_t e = make(s, l);
				for (size_t i = codes[s]; i < primary; i += (size_t)1 << l)
					table[i] = e;
			} else {
				const size_t p = codes[s] & (primary - 1);
				const uint32_t off = table[p] >> 16, sb = sub_bits[p];
				const uint32_t e = make(s, l - primary_bits);
				for (size_t i = codes[s] >> primary_bits; i < ((size_t)1 << sb); i += (size_t)1 << (l - primary_bits))
					table[off + i] = e;
			}
		}
		return true;
	}
	bool build_tables(const uint8_t* litlen, int n_lit, const uint8_t* dist, int n_dist)
	{
		static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
		static const uint8_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
		static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
		static const uint8_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
		if (!build_table(litlen, n_lit, kLitBits, lit_, sizeof lit_ / sizeof lit_[0], [&](int s, int l) {
			    if (s < 256)
				    return entry(LITERAL, (uint32_t)s, 0, (uint32_t)l);
			    if (s == 256)
				    return entry(END_OF_BLOCK, 0, 0, (uint32_t)l);
			    if (s > 285)
				    return entry(INVALID, 0, 0, (uint32_t)l);
			    return entry(BASE, len_base[s - 257], len_extra[s - 257], (uint32_t)l);
		    }))
			return false;
		return build_table(dist, n_dist, kDistBits, dist_, sizeof dist_ / sizeof dist_[0], [&](int s, int l) {
			if (s > 29)
				return entry(INVALID, 0, 0, (uint32_t)l);
			return entry(BASE, dist_base[s], dist_extra[s], (uint32_t)l);
		});
	}

	// ---- blocks ------------------------------------------------------------------------------------
	bool start_block()
	{
		uint32_t v;
		if (!bits(3, &v))
			return false;
		last_block_ = v & 1;
		const uint32_t type = v >> 1;
		if (type == 0) {
			drop(bitcnt_ & 7);
			uint32_t len, nlen;
			if (!bits(16, &len) || !bits(16, &nlen) || (len ^ 0xffffu) != nlen)
				return false;
			stored_left_ = len;
			state_ = STORED;
			return true;
		}
		uint8_t lens[288 + 32];
		if (type == 1) {
			int s = 0;
			for (; s < 144; ++s) lens[s] = 8;
			for (; s < 256; ++s) lens[s] = 9;
			for (; s < 280; ++s) lens[s] = 7;
			for (; s < 288; ++s) lens[s] = 8;
			for (s = 0; s < 32; ++s) lens[288 + s] = 5;
			if (!build_tables(lens, 288, lens + 288, 32))
				return false;
			state_ = HUFFMAN;
			return true;
		}
		if (type != 2)
			return false;
		uint32_t hlit, hdist, hclen;
		if (!bits(5, &hlit) || !bits(5, &hdist) || !bits(4, &hclen))
			return false;
		hlit += 257, hdist += 1, hclen += 4;
		if (hlit > 286 || hdist > 30)
			return false;
		static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
		uint8_t cl[19] = { 0 };
		for (uint32_t i = 0; i < hclen; ++i) {
			uint32_t x;
			if (!bits(3, &x))
				return false;
			cl[order[i]] = (uint8_t)x;
		}
		uint32_t cl_table[1 << 7];
		if (!build_table(cl, 19, 7, cl_table, 1 << 7, [&](int s, int l) { return entry(LITERAL, (uint32_t)s, 0, (uint32_t)l); }))
			return false;
		uint32_t i = 0;
		while (i < hlit + hdist) {
			if (!need(7 + 7)) {
				// the last code of the stream may leave fewer bits than the widest lookup wants
				if (bitcnt_ <= 0)
					return false;
			}
			const uint32_t e = cl_table[peek(7)];
			if ((e >> 13 & 7) != LITERAL || (int)(e & 0xff) > bitcnt_)
				return false;
			drop((int)(e & 0xff));
			const uint32_t sym = e >> 16;
			if (sym < 16) {
				lens[i++] = (uint8_t)sym;
				continue;
			}
			uint32_t rep, val = 0;
			if (sym == 16) {
				if (i == 0 || !bits(2, &rep))
					return false;
				rep += 3;
				val = lens[i - 1];
			} else if (sym == 17) {
				if (!bits(3, &rep))
					return false;
				rep += 3;
			} else {
				if (!bits(7, &rep))
					return false;
				rep += 11;
			}
			if (i + rep > hlit + hdist)
				return false;
			while (rep--)
				lens[i++] = (uint8_t)val;
		}
		if (lens[256] == 0)
			return false; // no end-of-block code
		uint8_t dl[32];
		std::memcpy(dl, lens + hlit, hdist);
		if (!build_tables(lens, (int)hlit, dl, (int)hdist))
			return false;
		state_ = HUFFMAN;
		return true;
	}

	bool copy_stored(size_t out_limit)
	{
		while (stored_left_ && out_pos_ < out_limit) {
			if (bitcnt_ >= 8) { // whole bytes that were already pulled into the bit buffer
				out_[out_pos_++] = (unsigned char)peek(8);
				drop(8);
				stored_left_--;
				member_out_++;
				continue;
			}
			bitbuf_ = 0; // no valid bits left: forget the partial look-ahead, the input is read directly now
			if (in_pos_ == in_end_) {
				refill_input();
				if (in_pos_ == in_end_)
					return false;
			}
			const size_t n = std::min<size_t>(std::min<size_t>(stored_left_, in_end_ - in_pos_), out_limit - out_pos_);
			std::memcpy(out_.data() + out_pos_, in_.data() + in_pos_, n);
			out_pos_ += n, in_pos_ += n, stored_left_ -= (uint32_t)n, member_out_ += n;
		}
		if (!stored_left_)
			state_ = last_block_ ? TRAILER_PENDING : BLOCK_START;
		return true;
	}

	bool inflate_huffman(size_t out_limit)
	{
		// the loop's state lives in locals: every store to `out` is a char store, which the compiler must
		// assume may alias the members
		unsigned char* const out = out_.data();
		const uint32_t* const lit = lit_;
		const uint32_t* const dist = dist_;
		const unsigned char* in = in_.data();
		size_t ip = in_pos_, ie = in_end_, op = out_pos_;
		uint64_t bb = bitbuf_;
		int bc = bitcnt_;
		const uint64_t before = member_out_; // bytes of this member in front of out_pos_
		const size_t op0 = op;
		bool ok = true;
		while (op < out_limit) {
			if (ie - ip < 16 && !in_eof_) {
				in_pos_ = ip;
				refill_input();
				ip = in_pos_, ie = in_end_;
			}
			if (ie - ip >= 8) { // refill_bits()
				uint64_t v;
				std::memcpy(&v, in + ip, 8);
				bb |= v << bc;
				const int n = (63 - bc) >> 3;
				ip += (size_t)n;
				bc += n * 8;
			} else
				while (bc <= 56 && ip < ie) {
					bb |= (uint64_t)in[ip++] << bc;
					bc += 8;
				}
			uint32_t e = lit[bb & ((1u << kLitBits) - 1)];
			if ((e >> 13 & 7) == SUBTABLE) {
				e = lit[(e >> 16) + ((bb >> kLitBits) & ((1u << (e >> 8 & 31)) - 1))];
				bb >>= kLitBits;
				bc -= kLitBits;
			}
			bb >>= (e & 0xff);
			bc -= (int)(e & 0xff);
			const uint32_t kind = e >> 13 & 7;
			if (kind == LITERAL) {
				out[op++] = (unsigned char)(e >> 16);
				// up to two more literals from the same refill (three codes are at most 45 of >= 56 bits)
				uint32_t e2 = lit[bb & ((1u << kLitBits) - 1)];
				if ((e2 >> 13 & 7) == LITERAL && bc >= (int)(e2 & 0xff)) {
					bb >>= (e2 & 0xff);
					bc -= (int)(e2 & 0xff);
					out[op++] = (unsigned char)(e2 >> 16);
					e2 = lit[bb & ((1u << kLitBits) - 1)];
					if ((e2 >> 13 & 7) == LITERAL && bc >= (int)(e2 & 0xff)) {
						bb >>= (e2 & 0xff);
						bc -= (int)(e2 & 0xff);
						out[op++] = (unsigned char)(e2 >> 16);
					}
				}
				if (bc < 0) {
					ok = false;
					break;
				}
				continue;
			}
			if (kind == END_OF_BLOCK) {
				if (bc < 0)
					ok = false;
				state_ = last_block_ ? TRAILER_PENDING : BLOCK_START;
				break;
			}
			if (kind != BASE) {
				ok = false;
				break;
			}
			const uint32_t lx = e >> 8 & 31;
			const uint32_t length = (e >> 16) + (uint32_t)(bb & ((1u << lx) - 1));
			bb >>= lx;
			bc -= (int)lx;
			uint32_t d = dist[bb & ((1u << kDistBits) - 1)];
			if ((d >> 13 & 7) == SUBTABLE) {
				d = dist[(d >> 16) + ((bb >> kDistBits) & ((1u << (d >> 8 & 31)) - 1))];
				bb >>= kDistBits;
				bc -= kDistBits;
			}
			if ((d >> 13 & 7) != BASE) {
				ok = false;
				break;
			}
			bb >>= (d & 0xff);
			bc -= (int)(d & 0xff);
			const uint32_t dx = d >> 8 & 31;
			const size_t distance = (d >> 16) + (size_t)(bb & ((1u << dx) - 1));
			bb >>= dx;
			bc -= (int)dx;
			if (bc < 0 || distance > kHistory || distance > before + (op - op0)) {
				ok = false;
				break;
			}
			unsigned char* dst = out + op;
			const unsigned char* src = dst - distance;
			op += length;
			if (distance >= 8) {
				unsigned char* const end = dst + length;
				do {
					std::memcpy(dst, src, 8);
					dst += 8, src += 8;
				} while (dst < end);
			} else if (distance == 1) {
				std::memset(dst, *src, length); // a run (base qualities)
			} else {
				// period 2..7: the first eight bytes one by one, the rest in words from a whole number of
				// periods back (>= 8 bytes away, same content)
				for (uint32_t i = 0; i < 8; ++i)
					dst[i] = src[i];
				if (length > 8) {
					const size_t back = distance * ((7 + distance) / distance);
					unsigned char* const end = dst + length;
					unsigned char* d8 = dst + 8;
					do {
						std::memcpy(d8, d8 - back, 8);
						d8 += 8;
					} while (d8 < end);
				}
			}
		}
		in_pos_ = ip;
		bitbuf_ = bb;
		bitcnt_ = bc;
		member_out_ += op - op0;
		out_pos_ = op;
		return ok;
	}

	// runs the state machine until the output chunk is full, a member ends or the stream does
	bool produce()
	{
		const size_t out_limit = kHistory + kChunk;
		for (;;) {
			switch (state_) {
			case HEADER:
				if (!read_header())
					return false;
				if (state_ == DONE)
					return true;
				break;
			case BLOCK_START:
				if (!start_block())
					return false;
				break;
			case STORED:
				if (!copy_stored(out_limit))
					return false;
				if (out_pos_ >= out_limit)
					return true;
				break;
			case HUFFMAN:
				if (!inflate_huffman(out_limit))
					return false;
				if (out_pos_ >= out_limit)
					return true;
				break;
			case TRAILER_PENDING: // read() adds the produced bytes to the CRC first
				return true;
			default:
				return true;
			}
		}
	}
};

// Runs a GzInflater one step ahead of its consumer on a thread of its own, so that the record parser and the
// inflate loop of ONE .gz file work at the same time (the file's rate becomes the slower of the two instead
// of their harmonic sum).  Four buffers in a ring; the thread starts with the first read.
class InflateAhead
{
  public:
	explicit InflateAhead(std::unique_ptr<GzInflater> src)
	  : src_(std::move(src))
	{
		for (auto& b : ring_)
			b.data.resize(kBuf);
	}
	~InflateAhead()
	{
		{
			std::lock_guard<std::mutex> lk(m_);
			stop_ = true;
		}
		cv_.notify_all();
		if (th_.joinable())
			th_.join();
	}
	InflateAhead(const InflateAhead&) = delete;
	InflateAhead& operator=(const InflateAhead&) = delete;

	int read(unsigned char* dst, int cap)
	{
		if (!th_.joinable())
			th_ = std::thread([this] { run(); });
		Buf& b = ring_[tail_ % kRing];
		if (pos_ == 0) { // wait for the next buffer
			std::unique_lock<std::mutex> lk(m_);
			cv_.wait(lk, [&] { return tail_ < head_ || stop_; });
			if (tail_ >= head_)
				return -1;
		}
		if (b.len <= 0)
			return b.len; // end of the stream (0) or a damaged one (-1): stays
		const int n = std::min(cap, b.len - pos_);
		std::memcpy(dst, b.data.data() + pos_, (size_t)n);
		pos_ += n;
		if (pos_ == b.len) {
			pos_ = 0;
			{
				std::lock_guard<std::mutex> lk(m_);
				tail_++;
			}
			cv_.notify_all();
		}
		return n;
	}

  private:
	static constexpr size_t kBuf = 1u << 18, kRing = 4;
	struct Buf
	{
		std::vector<unsigned char> data;
		int len = 0;
	};
	std::unique_ptr<GzInflater> src_;
	Buf ring_[kRing];
	uint64_t head_ = 0, tail_ = 0; // produced / consumed buffers
	int pos_ = 0;
	bool stop_ = false;
	std::mutex m_;
	std::condition_variable cv_;
	std::thread th_;

	void run()
	{
		for (;;) {
			{
				std::unique_lock<std::mutex> lk(m_);
				cv_.wait(lk, [&] { return head_ - tail_ < kRing || stop_; });
				if (stop_)
					return;
			}
			Buf& b = ring_[head_ % kRing];
			int got = 0, n = 0; // fill the buffer: the source hands out at most what it has decoded
			while (got < (int)kBuf && (n = src_->read(b.data.data() + got, (int)kBuf - got)) > 0)
				got += n;
			const bool last = n <= 0;
			b.len = got > 0 ? got : n;
			{
				std::lock_guard<std::mutex> lk(m_);
				head_++;
			}
			cv_.notify_all();
			if (last) {
				if (got > 0) { // the end marker (or the error) goes into a buffer of its own
					{
						std::unique_lock<std::mutex> lk(m_);
						cv_.wait(lk, [&] { return head_ - tail_ < kRing || stop_; });
						if (stop_)
							return;
					}
					ring_[head_ % kRing].len = n;
					{
						std::lock_guard<std::mutex> lk(m_);
						head_++;
					}
					cv_.notify_all();
				}
				return;
			}
		}
	}
};

} // namespace arks_host
