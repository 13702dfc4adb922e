// crc32_fold.hpp -- CRC-32 (the gzip polynomial, reflected 0xEDB88320) by carry-less multiplication.
//
// The inflate of fast_inflate.hpp runs at ~0.8 GB/s per core; zlib 1.2.11's table-driven crc32() at
// ~1 GB/s would take as long again.  Folding four 128-bit lanes with PCLMULQDQ (Gopal et al., "Fast CRC
// Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009: fold by x^(512+64) /
// x^512, then x^(128+64) / x^128, reduce 128 -> 64 -> 32 bits with a Barrett step) runs at > 10 GB/s.
// Checked against zlib's crc32 in tests/test_host_ingest.py; falls back to zlib where the CPU lacks the
// instruction or for short pieces.
#pragma once

#include <zlib.h>

#include <cstddef>
#include <cstdint>
#include <immintrin.h>

namespace arks_host {

// len >= 64 and a multiple of 16
__attribute__((target("pclmul,sse4.1"))) inline uint32_t
crc32_pclmul_blocks(uint32_t crc, const unsigned char* buf, size_t len)
{
	// x^n mod P (bit-reflected) for the fold distances: 512+64, 512, 128+64, 128, 64; then P and mu
	alignas(16) static const uint64_t k1k2[2] = { 0x0154442bd4ull, 0x01c6e41596ull };
	alignas(16) static const uint64_t k3k4[2] = { 0x01751997d0ull, 0x00ccaa009eull };
	alignas(16) static const uint64_t k5k0[2] = { 0x0163cd6124ull, 0x0000000000ull };
	alignas(16) static const uint64_t poly[2] = { 0x01db710641ull, 0x01f7011641ull };
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
	x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
	x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
	x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
	x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
	x0 = _mm_load_si128((const __m128i*)k1k2);
	buf += 64;
	len -= 64;
	while (len >= 64) { // four lanes, each folded 512 bits ahead
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
		x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
		x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
		y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
		y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
		y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
		x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
		x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
		buf += 64;
		len -= 64;
	}
	x0 = _mm_load_si128((const __m128i*)k3k4); // the four lanes into one, 128 bits at a time
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
	x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while (len >= 16) {
		x2 = _mm_loadu_si128((const __m128i*)buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16;
		len -= 16;
	}
	// 128 -> 64 bits
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_loadl_epi64((const __m128i*)k5k0);
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	// Barrett reduction to 32 bits
	x0 = _mm_load_si128((const __m128i*)poly);
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	return (uint32_t)_mm_extract_epi32(x1, 1);
}

// drop-in for zlib's crc32(crc, buf, len) (same pre/post conditioning)
inline uint32_t
crc32_fast(uint32_t crc, const unsigned char* buf, size_t len)
{
	static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
	if (have && len >= 64) {
		const size_t body = len & ~(size_t)15;
		crc = ~crc32_pclmul_blocks(~crc, buf, body);
		buf += body;
		len -= body;
	}
	while (len) { // zlib takes uInt lengths
		const size_t n = len < (1u << 30) ? len : (1u << 30);
		crc = (uint32_t)crc32(crc, buf, (uInt)n);
		buf += n;
		len -= n;
	}
	return crc;
}

} // namespace arks_host
