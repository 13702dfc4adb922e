// pgzip.hpp -- an ordinary (single-stream) gzip file inflated by several threads, for the read ingest.
//
// A gzip member is one deflate stream: block n+1 may copy from the 32 KiB of text in front of it, so the
// stream is usually decoded front to back by one thread (fast_inflate.hpp: ~1.4 GB/s of text) -- which is
// what bounds `arcs --arks` on the one big reads.fastq.gz that linked-read pipelines hand over.  The way
// around it (as in pugz / rapidgzip): deflate BLOCKS can be decoded from any block boundary if the text in
// front is treated as unknown --
//   1. the compressed file (mapped) is cut into chunks; for every chunk but the first a thread looks for
//      the first position at which a dynamic-Huffman block header parses, the block decodes to printable
//      text and another block header follows (find_block_start);
//   2. every chunk is decoded from its start to the next chunk's start into 16-bit symbols: a byte, or
//      "byte j of the 32 KiB in front of this chunk" (a marker).  Matches copy symbols, so markers
//      propagate exactly as the bytes would have;
//   3. the 32 KiB windows are resolved chunk after chunk (only the tail of each chunk is needed: cheap),
//      then every chunk's markers are replaced and its text written, in parallel again.
// A chunk's result is used only if the chunk in front of it ended EXACTLY on its start (then the two
// decodes are the sequential decode, cut at a block boundary); at the first chunk where that fails, at any
// damage, at the last block of the member -- anything but plain progress -- the stream is handed back to
// the sequential inflater at a block boundary, with the window, CRC and length so far
// (GzInflater::resume), so the bytes delivered are the sequential decoder's in every case.
// RFC 1951, RFC 1952.
#pragma once

#include "bgzf.hpp"
#include "crc32_fold.hpp"

#include <immintrin.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

namespace arks_host {
namespace pgz {

constexpr size_t kWindow = 32768;
constexpr size_t kNone = ~(size_t)0;

// bits of in[0, size), least significant first
struct BitIn
{
	const unsigned char* in;
	size_t size, ip = 0;
	uint64_t bb = 0;
	int bc = 0;
	BitIn(const unsigned char* p, size_t n)
	  : in(p)
	  , size(n)
	{}
	void seek_bit(size_t bit)
	{
		ip = std::min(size, bit >> 3);
		bb = 0;
		bc = 0;
		refill();
		const int skip = (int)(bit & 7);
		if (bc >= skip) {
			bb >>= skip;
			bc -= skip;
		} else
			bc = 0;
	}
	size_t bit_position() const { return ip * 8 - (size_t)bc; }
	inline void refill()
	{
		if (size - ip >= 8) {
			uint64_t v;
			std::memcpy(&v, in + ip, 8); // little endian host (x86-64)
			bb |= v << bc;
			const int n = (63 - bc) >> 3;
			ip += (size_t)n;
			bc += n * 8;
		} else
			while (bc <= 56 && ip < size) {
				bb |= (uint64_t)in[ip++] << bc;
				bc += 8;
			}
	}
	uint32_t peek(int n) const { return (uint32_t)(bb & ((1ull << n) - 1)); }
	void drop(int n)
	{
		bb >>= n;
		bc -= n;
	}
	// n <= 32
	bool bits(int n, uint32_t* v)
	{
		if (bc < n) {
			refill();
			if (bc < n)
				return false;
		}
		*v = peek(n);
		drop(n);
		return true;
	}
};

// table entry: bits 0-7 code length (bits to drop), 8-12 extra bits (or subtable bits), 13-15 kind,
// 16-31 literal / base value / subtable offset
enum Kind : uint32_t { LITERAL = 0, BASE = 1, END_OF_BLOCK = 2, SUBTABLE = 3, INVALID = 4 };
constexpr int kLitBits = 11, kDistBits = 8;
inline uint32_t
entry(Kind k, uint32_t value, uint32_t extra, uint32_t len)
{
	return (value << 16) | ((uint32_t)k << 13) | (extra << 8) | len;
}

struct Tables
{
	uint32_t lit[(1 << kLitBits) + 288 * 16], dist[(1 << kDistBits) + 32 * 128];
};

// canonical code of `lens` (RFC 1951 3.2.2) as a lookup table indexed by the next input bits.  Returns the
// unused code space (0 = a complete code), or -1 for an over-subscribed one.
template <typename MakeEntry>
inline long
build_table(const uint8_t* lens, int n, int primary_bits, uint32_t* table, size_t table_size, MakeEntry make)
{
	int count[16] = { 0 };
	for (int s = 0; s < n; ++s)
		count[lens[s]]++;
	count[0] = 0;
	uint32_t next[16], code = 0;
	long space = 1;
	for (int l = 1; l <= 15; ++l) {
		code = (code + (uint32_t)count[l - 1]) << 1;
		next[l] = code;
		space = (space << 1) - count[l];
		if (space < 0)
			return -1;
	}
	const size_t primary = (size_t)1 << primary_bits;
	for (size_t i = 0; i < table_size; ++i)
		table[i] = entry(INVALID, 0, 0, 0);
	uint8_t sub_bits[1 << kLitBits];
	std::memset(sub_bits, 0, primary);
	uint32_t codes[288];
	for (int s = 0; s < n; ++s) {
		const int l = lens[s];
		if (!l)
			continue;
		uint32_t c = next[l]++, r = 0;
		for (int i = 0; i < l; ++i)
			r |= ((c >> i) & 1u) << (l - 1 - i);
		codes[s] = r;
		if (l > primary_bits) {
			uint8_t& b = sub_bits[r & (primary - 1)];
			if (l - primary_bits > b)
				b = (uint8_t)(l - primary_bits);
		}
	}
	size_t next_sub = primary;
	for (size_t p = 0; p < primary; ++p)
		if (sub_bits[p]) {
			if (next_sub + ((size_t)1 << sub_bits[p]) > table_size)
				return -1;
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
	return space;
}

// strict: what a compressor writes -- complete codes (a distance code may also be a single code, or none).
// Otherwise what an inflater must take: anything that is not over-subscribed (a code that is not there is
// an INVALID entry and fails when it is met).
// unused code space of the canonical code `lens` (0 = complete, < 0 = over-subscribed)
inline long
code_space(const uint8_t* lens, int n)
{
	int count[16] = { 0 };
	for (int s = 0; s < n; ++s)
		count[lens[s]]++;
	long space = 1;
	for (int l = 1; l <= 15; ++l) {
		space = (space << 1) - count[l];
		if (space < 0)
			return -1;
	}
	return space;
}

inline bool
build_tables(Tables& t, const uint8_t* litlen, int n_lit, const uint8_t* dist, int n_dist, bool strict)
{
	// (the search for block starts comes here for every bit position that looks like a header: the codes are
	// weighed before 40 KB of tables are filled for them)
	if (strict && (code_space(litlen, n_lit) != 0 || code_space(dist, n_dist) < 0))
		return false;
	static const uint16_t len_base[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
	static const uint8_t len_extra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	static const uint16_t dist_base[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
	static const uint8_t dist_extra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
	const long s1 = build_table(litlen, n_lit, kLitBits, t.lit, sizeof t.lit / sizeof t.lit[0], [&](int s, int l) {
		if (s < 256)
			return entry(LITERAL, (uint32_t)s, 0, (uint32_t)l);
		if (s == 256)
			return entry(END_OF_BLOCK, 0, 0, (uint32_t)l);
		if (s > 285)
			return entry(INVALID, 0, 0, (uint32_t)l);
		return entry(BASE, len_base[s - 257], len_extra[s - 257], (uint32_t)l);
	});
	if (s1 < 0 || (strict && s1 != 0))
		return false;
	const long s2 = build_table(dist, n_dist, kDistBits, t.dist, sizeof t.dist / sizeof t.dist[0], [&](int s, int l) {
		if (s > 29)
			return entry(INVALID, 0, 0, (uint32_t)l);
		return entry(BASE, dist_base[s], dist_extra[s], (uint32_t)l);
	});
	if (s2 < 0)
		return false;
	if (strict && s2 != 0) {
		int used = 0, longest = 0;
		for (int s = 0; s < n_dist; ++s)
			if (dist[s]) {
				used++;
				longest = std::max<int>(longest, dist[s]);
			}
		if (!(used == 0 || (used == 1 && longest == 1)))
			return false;
	}
	return true;
}

// the header of the block at the reader's position: *type = 0 stored (the reader then stands on LEN), 1 or 2
// (tables built).  False on anything an inflater refuses.
inline bool
read_block_header(BitIn& b, Tables& t, bool* last, int* type, bool strict)
{
	uint32_t v;
	if (!b.bits(3, &v))
		return false;
	*last = v & 1;
	*type = (int)(v >> 1);
	if (*type == 0)
		return true;
	uint8_t lens[288 + 32];
	if (*type == 1) {
		int s = 0;
		for (; s < 144; ++s) lens[s] = 8;
		for (; s < 256; ++s) lens[s] = 9;
		for (; s < 280; ++s) lens[s] = 7;
		for (; s < 288; ++s) lens[s] = 8;
		for (s = 0; s < 32; ++s) lens[288 + s] = 5;
		return build_tables(t, lens, 288, lens + 288, 32, false);
	}
	if (*type != 2)
		return false;
	uint32_t hlit, hdist, hclen;
	if (!b.bits(5, &hlit) || !b.bits(5, &hdist) || !b.bits(4, &hclen))
		return false;
	hlit += 257, hdist += 1, hclen += 4;
	if (hlit > 286 || hdist > 30)
		return false;
	static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	uint8_t cl[19] = { 0 };
	for (uint32_t i = 0; i < hclen; ++i) {
		uint32_t x;
		if (!b.bits(3, &x))
			return false;
		cl[order[i]] = (uint8_t)x;
	}
	uint32_t cl_table[1 << 7];
	const long cs = build_table(cl, 19, 7, cl_table, 1 << 7, [&](int s, int l) { return entry(LITERAL, (uint32_t)s, 0, (uint32_t)l); });
	if (cs < 0 || (strict && cs != 0))
		return false;
	uint32_t i = 0;
	while (i < hlit + hdist) {
		if (b.bc < 14)
			b.refill();
		if (b.bc <= 0)
			return false;
		const uint32_t e = cl_table[b.peek(7)];
		if ((e >> 13 & 7) != LITERAL || (int)(e & 0xff) > b.bc)
			return false;
		b.drop((int)(e & 0xff));
		const uint32_t sym = e >> 16;
		if (sym < 16) {
			lens[i++] = (uint8_t)sym;
			continue;
		}
		uint32_t rep, val = 0;
		if (sym == 16) {
			if (i == 0 || !b.bits(2, &rep))
				return false;
			rep += 3;
			val = lens[i - 1];
		} else if (sym == 17) {
			if (!b.bits(3, &rep))
				return false;
			rep += 3;
		} else {
			if (!b.bits(7, &rep))
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
	return build_tables(t, lens, (int)hlit, dl, (int)hdist, strict);
}

// what the text of a FASTA / FASTQ file is made of (the test that makes a false block start unlikely)
inline bool
text_byte(unsigned c)
{
	return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t';
}

// Symbols of one chunk: sym[-kWindow .. -1] are the markers of the unknown window in front of the chunk
// (256 + j = its byte j), sym[0 .. n) what was decoded.  Grows as needed.
struct Symbols
{
	uint16_t* p = nullptr; // (malloc: grown with realloc, never zero-filled)
	size_t cap = 0;        // symbols there is room for behind the markers
	size_t n = 0;
	explicit Symbols(size_t expect = 0)
	{
		cap = expect + 512;
		p = (uint16_t*)std::malloc((kWindow + cap) * sizeof(uint16_t));
		if (!p)
			throw std::bad_alloc();
		for (size_t j = 0; j < kWindow; ++j)
			p[j] = (uint16_t)(256 + j);
	}
	~Symbols() { std::free(p); }
	Symbols(const Symbols&) = delete;
	Symbols& operator=(const Symbols&) = delete;
	uint16_t* sym() { return p + kWindow; }
	const uint16_t* sym() const { return p + kWindow; }
	void room(size_t more)
	{
		if (n + more + 8 <= cap)
			return;
		const size_t want = std::max(cap + cap / 2, n + more + 8);
		uint16_t* q = (uint16_t*)std::realloc(p, (kWindow + want) * sizeof(uint16_t));
		if (!q)
			throw std::bad_alloc();
		p = q;
		cap = want;
	}
};

__attribute__((target("avx2"))) inline void
copy_symbols_avx2(uint16_t* to, const uint16_t* from, const uint16_t* end)
{
	do {
		_mm256_storeu_si256((__m256i*)to, _mm256_loadu_si256((const __m256i*)from));
		to += 16, from += 16;
	} while (to < end);
}
__attribute__((target("avx2"))) inline void
fill_symbols_avx2(uint16_t* to, uint16_t v, const uint16_t* end)
{
	const __m256i x = _mm256_set1_epi16((short)v);
	do {
		_mm256_storeu_si256((__m256i*)to, x);
		to += 16;
	} while (to < end);
}

// the body of a Huffman block (the header was read) appended to `out`; kText: every literal must be text
template <bool kText>
inline bool
huffman_block(BitIn& b, const Tables& t, Symbols& out)
{
	// the loop's state lives in locals (the reader's members would be reloaded around every store)
	size_t n = out.n, cap = out.cap;
	uint16_t* sym = out.sym();
	const unsigned char* const in = b.in;
	const size_t size = b.size;
	size_t ip = b.ip;
	uint64_t bb = b.bb;
	int bc = b.bc;
	const uint32_t* const lit = t.lit;
	const uint32_t* const dist = t.dist;
	static const bool wide = __builtin_cpu_supports("avx2");
	bool ok = false;
	for (;;) {
		if (n + 296 > cap) {
			out.n = n;
			out.room(1 << 16);
			sym = out.sym();
			cap = out.cap;
		}
		if (size - ip >= 8) { // BitIn::refill
			uint64_t v;
			std::memcpy(&v, in + ip, 8);
			bb |= v << bc;
			const int k = (63 - bc) >> 3;
			ip += (size_t)k;
			bc += k * 8;
		} else
			while (bc <= 56 && ip < size) {
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
		if (bc < 0)
			break; // the input ends inside the block
		const uint32_t kind = e >> 13 & 7;
		if (kind == LITERAL) {
			if (kText && !text_byte(e >> 16))
				break;
			sym[n++] = (uint16_t)(e >> 16);
			// up to two more literals from the same refill (three codes are at most 45 of >= 56 bits)
			uint32_t e2 = lit[bb & ((1u << kLitBits) - 1)];
			if ((e2 >> 13 & 7) == LITERAL && bc >= (int)(e2 & 0xff) && (!kText || text_byte(e2 >> 16))) {
				bb >>= (e2 & 0xff);
				bc -= (int)(e2 & 0xff);
				sym[n++] = (uint16_t)(e2 >> 16);
				e2 = lit[bb & ((1u << kLitBits) - 1)];
				if ((e2 >> 13 & 7) == LITERAL && bc >= (int)(e2 & 0xff) && (!kText || text_byte(e2 >> 16))) {
					bb >>= (e2 & 0xff);
					bc -= (int)(e2 & 0xff);
					sym[n++] = (uint16_t)(e2 >> 16);
				}
			}
			continue;
		}
		if (kind == END_OF_BLOCK) {
			ok = true;
			break;
		}
		if (kind != BASE)
			break;
		const uint32_t xl = e >> 8 & 31;
		const uint32_t length = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1));
		bb >>= xl;
		bc -= (int)xl;
		uint32_t d = dist[bb & ((1u << kDistBits) - 1)];
		if ((d >> 13 & 7) == SUBTABLE) {
			d = dist[(d >> 16) + ((bb >> kDistBits) & ((1u << (d >> 8 & 31)) - 1))];
			bb >>= kDistBits;
			bc -= kDistBits;
		}
		if ((d >> 13 & 7) != BASE)
			break;
		bb >>= (d & 0xff);
		bc -= (int)(d & 0xff);
		const uint32_t xd = d >> 8 & 31;
		const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1));
		bb >>= xd;
		bc -= (int)xd;
		if (bc < 0 || distance > kWindow) // (no valid code gives more than 32768)
			break;
		const uint16_t* from = sym + n - distance; // may reach into the markers in front of the chunk
		uint16_t* to = sym + n;
		n += length;
		uint16_t* const end = to + length;
		if (distance >= 16 && wide) { // sixteen symbols at a time (the room asked for above covers the overshoot)
			copy_symbols_avx2(to, from, end);
		} else if (distance >= 8) {
			do {
				_mm_storeu_si128((__m128i*)to, _mm_loadu_si128((const __m128i*)from));
				to += 8, from += 8;
			} while (to < end);
		} else if (distance >= 4) {
			do {
				std::memcpy(to, from, 8);
				to += 4, from += 4;
			} while (to < end);
		} else if (distance == 1) { // a run (base qualities)
			if (wide)
				fill_symbols_avx2(to, *from, end);
			else {
				const __m128i v = _mm_set1_epi16((short)*from);
				do {
					_mm_storeu_si128((__m128i*)to, v);
					to += 8;
				} while (to < end);
			}
		} else
			for (uint32_t i = 0; i < length; ++i)
				to[i] = from[i];
	}
	out.n = n;
	b.ip = ip;
	b.bb = bb;
	b.bc = bc < 0 ? 0 : bc;
	return ok;
}

inline bool
stored_block(BitIn& b, Symbols& out)
{
	b.drop(b.bc & 7);
	uint32_t len, nlen;
	if (!b.bits(16, &len) || !b.bits(16, &nlen) || (len ^ 0xffffu) != nlen)
		return false;
	out.room(len);
	uint16_t* sym = out.sym();
	for (uint32_t i = 0; i < len; ++i) {
		uint32_t c;
		if (!b.bits(8, &c))
			return false;
		sym[out.n++] = (uint16_t)c;
	}
	return true;
}

// The first bit position in [from_bit, to_bit) at which a dynamic block that is not the last one starts:
// header complete and consistent as a compressor writes it, the block decodes to text, and a plausible
// block header follows.  kNone if there is none.
inline size_t
find_block_start(const unsigned char* in, size_t size, size_t from_bit, size_t to_bit)
{
	std::unique_ptr<Tables> t(new Tables), t2(new Tables);
	Symbols scratch(1 << 20);
	BitIn b(in, size);
	for (size_t p = from_bit; p < to_bit; ++p) {
		// BFINAL = 0, BTYPE = 10 (dynamic): bits 0, 0, 1 in stream order
		const size_t byte = p >> 3;
		if (byte + 1 >= size)
			return kNone;
		const unsigned w = (unsigned)in[byte] | ((unsigned)in[byte + 1] << 8);
		if (((w >> (p & 7)) & 7u) != 4u)
			continue;
		if (byte + 16 <= size) {
			// the fixed part of the header out of two words, before anything is built: HLIT and HDIST in range, and
			// the code of the code lengths (HCLEN + 4 lengths of 3 bits) complete -- one position in hundreds is
			unsigned __int128 v;
			std::memcpy(&v, in + byte, 16);
			v >>= (p & 7);
			const unsigned hlit = (unsigned)(v >> 3) & 31u, hdist = (unsigned)(v >> 8) & 31u, hclen = ((unsigned)(v >> 13) & 15u) + 4u;
			if (hlit > 29u || hdist > 29u)
				continue;
			v >>= 17;
			int space = 128;
			for (unsigned i = 0; i < hclen; ++i) {
				const unsigned l = (unsigned)v & 7u;
				v >>= 3;
				if (l)
					space -= 128 >> l;
			}
			if (space != 0)
				continue;
		}
		b.seek_bit(p);
		bool last = false;
		int type = 0;
		if (!read_block_header(b, *t, &last, &type, true) || last || type != 2)
			continue;
		scratch.n = 0;
		if (!huffman_block<true>(b, *t, scratch) || scratch.n == 0)
			continue;
		// what follows must be the header of another block
		bool last2 = false;
		int type2 = 0;
		BitIn nb = b;
		if (!read_block_header(nb, *t2, &last2, &type2, true))
			continue;
		if (type2 == 0) { // stored: LEN and its complement
			nb.drop(nb.bc & 7);
			uint32_t len, nlen;
			if (!nb.bits(16, &len) || !nb.bits(16, &nlen) || (len ^ 0xffffu) != nlen)
				continue;
		}
		return p;
	}
	return kNone;
}

// sixteen symbols -> sixteen bytes if every one of them is a byte
__attribute__((target("sse4.1"))) inline bool
pack16(const uint16_t* sym, unsigned char* out)
{
	const __m128i a = _mm_loadu_si128((const __m128i*)sym), b = _mm_loadu_si128((const __m128i*)(sym + 8));
	const __m128i hi = _mm_set1_epi16((short)0xff00);
	if (!_mm_testz_si128(_mm_or_si128(a, b), hi))
		return false;
	_mm_storeu_si128((__m128i*)out, _mm_packus_epi16(a, b));
	return true;
}

struct ChunkResult
{
	Symbols out;
	size_t start_bit = 0, end_bit = 0; // end: the boundary the decode stopped at (in front of a block)
	bool error = false;                // the block at end_bit did not decode
	bool final_ahead = false;          // the block at end_bit is the member's last
	explicit ChunkResult(size_t expect)
	  : out(expect)
	{}
	void reset()
	{
		out.n = 0;
		start_bit = end_bit = 0;
		error = final_ahead = false;
	}
};

// A chunk's decode also ends -- at a block boundary, like at stop_bit -- once it has produced this many symbols:
// the output of 1 MB of compressed text has no bound of its own (a run of one base compresses 1000 : 1), and
// a stretch of 64 such chunks would otherwise grow to tens of GB of 16-bit symbols and overflow the 32-bit line
// offsets of the record splitter.  The chunk behind it then does not start where this one ended: the chain breaks
// there, the next stretch takes over at this chunk's end, and a file that keeps doing so goes to the one-thread
// inflater (constant memory) by the poor-stretch rule below.
constexpr size_t kChunkOutCap = (size_t)48 << 20;   // symbols per chunk (96 MB)
constexpr size_t kStretchOutCap = (size_t)1 << 30;  // bytes of text per stretch

// blocks from start_bit on until a block boundary at or beyond stop_bit, the member's last block, or damage
inline void
decode_chunk(const unsigned char* in, size_t size, size_t start_bit, size_t stop_bit, ChunkResult& r, size_t out_cap = kChunkOutCap)
{
	std::unique_ptr<Tables> t(new Tables);
	BitIn b(in, size);
	size_t pos = start_bit;
	r.start_bit = start_bit;
	for (;;) {
		r.end_bit = pos;
		b.seek_bit(pos);
		if (b.bc < 3) {
			r.error = pos < stop_bit;
			return;
		}
		if (b.peek(1)) {
			r.final_ahead = true;
			return;
		}
		// behind the stop: up to the next block of the kind a chunk can start with (find_block_start: dynamic,
		// not the last) -- the empty stored blocks of a flushed stream (pigz) and fixed blocks belong to this chunk
		if (pos >= stop_bit && b.peek(3) == 4u)
			return;
		if (r.out.n >= out_cap && pos > start_bit)
			return; // (see kChunkOutCap)
		bool last = false;
		int type = 0;
		const size_t n0 = r.out.n;
		bool ok = read_block_header(b, *t, &last, &type, false);
		if (ok)
			ok = type == 0 ? stored_block(b, r.out) : huffman_block<false>(b, *t, r.out);
		if (!ok) {
			r.out.n = n0;
			r.error = true;
			return;
		}
		pos = b.bit_position();
	}
}

// the member's last block (BFINAL = 1) at start_bit: r.end_bit = the first bit behind it
inline void
decode_last_block(const unsigned char* in, size_t size, size_t start_bit, ChunkResult& r)
{
	std::unique_ptr<Tables> t(new Tables);
	BitIn b(in, size);
	b.seek_bit(start_bit);
	r.start_bit = r.end_bit = start_bit;
	bool last = false;
	int type = 0;
	bool ok = b.bc >= 3 && read_block_header(b, *t, &last, &type, false) && last;
	if (ok)
		ok = type == 0 ? stored_block(b, r.out) : huffman_block<false>(b, *t, r.out);
	r.error = !ok;
	if (ok)
		r.end_bit = b.bit_position();
}

// the start of the deflate stream of the gzip member at in[at ...] (RFC 1952 2.3), or kNone
inline size_t
member_data_offset(const unsigned char* in, size_t size, size_t at)
{
	if (size - at < 18 || in[at] != 0x1f || in[at + 1] != 0x8b || in[at + 2] != 8 || (in[at + 3] & 0xe0))
		return kNone;
	const unsigned flg = in[at + 3];
	size_t p = at + 10;
	if (flg & 4) {
		if (p + 2 > size)
			return kNone;
		p += 2 + ((size_t)in[p] | ((size_t)in[p + 1] << 8));
	}
	for (int k = 0; k < 2; ++k)
		if (flg & (k == 0 ? 8 : 16)) {
			while (p < size && in[p])
				++p;
			++p;
		}
	if (flg & 2)
		p += 2;
	return p < size ? p : kNone;
}

} // namespace pgz

// Where the sequential inflater takes over: a block boundary of the first member, and what it needs to go on
struct GzResumePoint
{
	size_t bit = 0;            // position of the block's first bit in the file
	std::vector<unsigned char> window; // the (up to) 32 KiB of text in front of it
	uint32_t crc = 0;          // CRC-32 of the member's text so far
	uint64_t member_out = 0;   // its length
	bool member_start = false; // instead: `bit` is the first bit behind a complete member (GzInflater::resume_member)
};

// One gzip file taken a stretch of text at a time by the threads of `pf` (see the head of this file).
class GzStretches
{
  public:
	// `chunk`: compressed bytes per chunk; a stretch is up to `chunks_per_stretch` of them
	GzStretches(FILE* f, size_t chunk, unsigned chunks_per_stretch)
	  : chunk_(std::max<size_t>(chunk, 1 << 12))
	  , per_stretch_(std::max(2u, chunks_per_stretch))
	{
		struct stat st;
		if (::fstat(fileno(f), &st) != 0 || st.st_size < 64)
			return;
		void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
		if (m == MAP_FAILED)
			return;
		(void)::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
		map_ = (const unsigned char*)m;
		size_ = (size_t)st.st_size;
		const size_t data = pgz::member_data_offset(map_, size_, 0);
		if (data == pgz::kNone) {
			done_ = true;
			return;
		}
		resume_.bit = data * 8;
		// ARKS_PGZIP_OUT_CAP=<bytes>: the cap on a stretch's text (tests; a chunk's cap is a sixteenth of it)
		if (const char* e = std::getenv("ARKS_PGZIP_OUT_CAP")) {
			const size_t v = (size_t)std::strtoull(e, nullptr, 10);
			if (v >= 4096) {
				stretch_out_cap_ = v;
				chunk_out_cap_ = std::max<size_t>(v / 16, 1024);
			}
		}
	}
	~GzStretches()
	{
		if (map_)
			(void)::munmap(const_cast<unsigned char*>(map_), size_);
	}
	GzStretches(const GzStretches&) = delete;
	GzStretches& operator=(const GzStretches&) = delete;
	bool ok() const { return map_ != nullptr; }
	bool at_end() const { return done_; } // no further stretch: the sequential inflater goes on at resume()
	const GzResumePoint& resume() const { return resume_; }
	bool started() const { return started_; } // false: nothing was decoded here, the reader may start afresh

	// Decodes the next stretch.  Returns the bytes of text it holds; `place(n)` is called once with that
	// number and returns where they go.
	size_t next(const ParallelFor& pf, const std::function<unsigned char*(size_t)>& place)
	{
		if (done_)
			return 0;
		const bool prof = std::getenv("ARKS_PGZIP_PROFILE") != nullptr;
		auto clk = [] { return std::chrono::steady_clock::now(); };
		auto t0 = clk();
		auto lap = [&](const char* what) {
			if (prof)
				std::fprintf(stderr, "pgzip: %-8s %.1f ms\n", what, std::chrono::duration<double>(clk() - t0).count() * 1e3);
			t0 = clk();
		};
		const size_t first_byte = resume_.bit >> 3;
		// chunk i covers the compressed bytes from border[i] on: the first starts at the known block boundary,
		// every other at the first block start that find_block_start sees behind its border, and each decodes up
		// to the first block boundary at or behind the next border -- which is the next chunk's start unless a
		// block of another kind (stored, fixed, the last one) or a false start comes between
		std::vector<size_t> border;
		for (unsigned i = 0; i <= per_stretch_; ++i) {
			const size_t at = std::min(size_, first_byte + (size_t)i * chunk_);
			border.push_back(at);
			if (at == size_)
				break;
		}
		const size_t nc = border.size() - 1;
		std::vector<std::unique_ptr<pgz::ChunkResult>>& res = results_; // (kept from stretch to stretch: no new pages)
		if (res.size() < nc)
			res.resize(nc);
		std::vector<size_t> begin(nc, pgz::kNone);
		pf(nc, [&](size_t i) {
			if (!res[i])
				res[i].reset(new pgz::ChunkResult(chunk_ * 6));
			res[i]->reset();
			begin[i] = i == 0 ? resume_.bit : pgz::find_block_start(map_, size_, border[i] * 8, border[i + 1] * 8);
			if (begin[i] != pgz::kNone)
				pgz::decode_chunk(map_, size_, begin[i], std::max(begin[i] + 1, border[i + 1] * 8), *res[i], chunk_out_cap_);
		});
		lap("decode");
		// the chunks whose results stand: every one that starts where the one in front of it ended (a chunk in
		// whose range no block starts -- a block longer than a chunk -- is simply passed over by that one)
		std::vector<const pgz::ChunkResult*> acc;
		bool stop_here = false, chain_broke = false;
		for (size_t i = 0; i < nc && !stop_here && !chain_broke;) {
			const pgz::ChunkResult& r = *res[i];
			acc.push_back(&r);
			if (r.error || r.final_ahead) {
				stop_here = true; // what the chunk holds in front of that block is good
				break;
			}
			size_t j = i + 1;
			while (j < nc && (begin[j] == pgz::kNone || begin[j] < r.end_bit))
				++j;
			if (j < nc && begin[j] != r.end_bit)
				chain_broke = true; // the next stretch starts where this chunk ended
			i = j;
		}
		size_t good = acc.size();
		// text that keeps breaking the chain (stored blocks, something that is not text): one thread does better
		if (chain_broke && nc >= 4 && good < std::max<size_t>(2, nc / 4)) {
			if (++poor_stretches_ >= 2)
				stop_here = true;
		} else if (good >= nc / 2)
			poor_stretches_ = 0;
		// windows, chunk after chunk; a reference in front of the member's first byte is damage
		std::vector<std::vector<unsigned char>> win(good + 1);
		win[0] = resume_.window;
		std::vector<size_t> off(good + 1, 0);
		size_t usable = good;
		for (size_t i = 0; i < good; ++i) {
			const pgz::Symbols& s = acc[i]->out;
			off[i + 1] = off[i] + s.n;
			if (!tail_window(s, win[i], win[i + 1])) {
				usable = i; // chunk i is the sequential decoder's to report
				break;
			}
		}
		if (usable < good) {
			good = usable;
			stop_here = true;
		}
		// no more than kStretchOutCap bytes of text per stretch (but at least one chunk): the rest of the chunks
		// are decoded again by the next stretch, which starts where the last chunk taken ended
		for (size_t i = 1; i < good; ++i)
			if (off[i + 1] > stretch_out_cap_) {
				good = i;
				break;
			}
		lap("windows");
		// the member's last block, if it comes next: decoded here too, so that a file of several members (lanes
		// put together with cat) stays on this path; taken only if the member's CRC and length then agree
		bool have_fin = false;
		if (good > 0 && usable == good && acc[good - 1]->final_ahead && !acc[good - 1]->error) {
			if (!fin_)
				fin_.reset(new pgz::ChunkResult(1 << 20));
			fin_->reset();
			pgz::decode_last_block(map_, size_, acc[good - 1]->end_bit, *fin_);
			have_fin = !fin_->error && (fin_->end_bit + 7) / 8 + 8 <= size_;
		}
		const size_t total = off[good] + (have_fin ? fin_->out.n : 0);
		unsigned char* dst = total ? place(total) : nullptr;
		lap("place");
		std::vector<uint32_t> crc(good + 1, 0);
		std::vector<char> bad(good + 1, 0);
		pf(good + (have_fin ? 1 : 0), [&](size_t i) {
			const pgz::Symbols& s = i < good ? acc[i]->out : fin_->out;
			unsigned char* o = dst + off[i];
			const uint16_t* sym = s.sym();
			const std::vector<unsigned char>& w = win[i];
			const size_t lack = pgz::kWindow - w.size();
			// symbol -> byte through a table (a byte is itself, marker j is byte j of the window): in FASTQ text
			// markers do not thin out -- every quality line and header copies from the one before it, back to the
			// unknown window -- so the table is the common path, sixteen plain bytes at a time the fast one
			std::vector<unsigned char> lut(256 + pgz::kWindow, 0);
			for (unsigned v = 0; v < 256; ++v)
				lut[v] = (unsigned char)v;
			if (!w.empty())
				std::memcpy(lut.data() + 256 + lack, w.data(), w.size());
			size_t k = 0;
			for (; k + 16 <= s.n; k += 16)
				if (!pgz::pack16(sym + k, o + k))
					for (size_t q = k; q < k + 16; ++q)
						o[q] = lut[sym[q]];
			for (; k < s.n; ++k)
				o[k] = lut[sym[k]];
			if (lack) // (the first 32 KiB of a member only) a marker in front of the member's first byte is damage
				for (size_t q = 0; q < s.n; ++q)
					if (sym[q] >= 256 && (size_t)(sym[q] - 256) < lack) {
						bad[i] = 1;
						break;
					}
			crc[i] = crc32_fast(0u, o, s.n);
		});
		lap("resolve");
		size_t kept = good;
		for (size_t i = 0; i < good; ++i)
			if (bad[i]) {
				kept = i;
				break;
			}
		have_fin = have_fin && kept == good && !bad[good];
		const uint32_t fin_crc = crc[good];
		if (kept < good) {
			good = kept;
			stop_here = true;
		}
		for (size_t i = 0; i < good; ++i) {
			resume_.crc = (uint32_t)crc32_combine(resume_.crc, crc[i], (z_off_t)acc[i]->out.n);
			resume_.member_out += acc[i]->out.n;
		}
		if (good) {
			resume_.bit = acc[good - 1]->end_bit;
			resume_.window = win[good];
			started_ = true;
		}
		if (have_fin) {
			const size_t trailer = (fin_->end_bit + 7) / 8;
			const unsigned char* t = map_ + trailer;
			const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
			const uint32_t want_len = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
			const uint32_t all_crc = (uint32_t)crc32_combine(resume_.crc, fin_crc, (z_off_t)fin_->out.n);
			if (all_crc == want_crc && (uint32_t)(resume_.member_out + fin_->out.n) == want_len) {
				// the member is complete: the next one (if the bytes behind it are a member's head) starts a text
				// of its own; anything else is the sequential inflater's to judge
				const size_t next_member = trailer + 8;
				const size_t data = next_member < size_ ? pgz::member_data_offset(map_, size_, next_member) : pgz::kNone;
				resume_ = GzResumePoint();
				// (members of a few chunks each -- a blocked gzip without the BGZF field -- leave nothing to spread)
				const bool small = next_member - member_begin_ < 4 * chunk_;
				tiny_members_ = small ? tiny_members_ + 1 : 0;
				const bool tiny = tiny_members_ >= 2;
				member_begin_ = next_member;
				if (data != pgz::kNone && !tiny) {
					resume_.bit = data * 8;
					done_ = false;
				} else {
					resume_.bit = next_member * 8;
					resume_.member_start = true;
					done_ = true;
				}
				return total;
			}
			// (damage somewhere in the member: the sequential inflater decodes the last block again and reports)
		}
		if (stop_here || good == 0 || resume_.bit >= size_ * 8)
			done_ = true;
		return off[good];
	}

  private:
	// the last 32 KiB of (window `w` + the chunk's symbols), markers replaced; false when a marker points in
	// front of the text there is
	static bool tail_window(const pgz::Symbols& s, const std::vector<unsigned char>& w, std::vector<unsigned char>& out)
	{
		const size_t have = w.size() + s.n, keep = std::min(have, pgz::kWindow);
		out.resize(keep);
		const size_t lack = pgz::kWindow - w.size();
		const uint16_t* sym = s.sym();
		for (size_t k = 0; k < keep; ++k) {
			const size_t pos = have - keep + k; // in window + symbols
			if (pos < w.size()) {
				out[k] = w[pos];
				continue;
			}
			const unsigned v = sym[pos - w.size()];
			if (v < 256)
				out[k] = (unsigned char)v;
			else if (v - 256 >= lack)
				out[k] = w[v - 256 - lack];
			else
				return false;
		}
		return true;
	}

	const unsigned char* map_ = nullptr;
	size_t size_ = 0, chunk_;
	unsigned per_stretch_;
	size_t chunk_out_cap_ = pgz::kChunkOutCap, stretch_out_cap_ = pgz::kStretchOutCap;
	bool done_ = false, started_ = false;
	unsigned poor_stretches_ = 0, tiny_members_ = 0;
	size_t member_begin_ = 0;
	GzResumePoint resume_;
	std::vector<std::unique_ptr<pgz::ChunkResult>> results_;
	std::unique_ptr<pgz::ChunkResult> fin_;
};

} // namespace arks_host
