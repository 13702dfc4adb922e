// arks_capi.hip -- the C ABI of libarks_hip.so (include/arks_hip.h) over the kernels of
// arks_kernels.hip.  No CPU fallback exists: without a gfx950 device every compute entry point
// returns ARKS_ERR_NO_DEVICE.
#include "arks_hip.h"
#include "arks_hip_debug.h"
#include "arks_kernels.hpp"
#include "arks_shard_stats.hpp"

// minimizer-table slots per entry: the map kernel's probe is a dependent HBM round trip per group of 4
// entries, so what matters is how often a run needs a second one (measured at C2: 2 -> 4 slots per
// entry +6 %, 6: +8 %); 8 B per slot
#ifndef ARKS_MTAB_LOAD_INV
#define ARKS_MTAB_LOAD_INV 6
#endif
// the same for the seed index, whose table holds every m-mer position of the text (8 B per slot:
// 32 B per text position at 4) and is probed by far fewer lanes per tile
#ifndef ARKS_SEED_LOAD_INV
#define ARKS_SEED_LOAD_INV 4
#endif

#include <immintrin.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace arks {
hipError_t launch_gate_count(const uint8_t* pair_ok, const uint8_t* eval, long n_pairs, u64* counter, hipStream_t st); // arks_imap.hip
}
using namespace arks;

// ARKS_DEBUG_POISON_LDS=<hex word>: before every map launch the LDS of every CU is filled with that word (LDS is
// not cleared between kernels: a workgroup sees what the last one on its CU left, of this process or another's).
// A read of LDS the kernel did not write then meets the same garbage on every run -- the debugging aid for the
// kind of fault that only shows with other processes on the GPU.
__global__ void
poison_lds_kernel(unsigned word)
{
	extern __shared__ unsigned poison_words[];
	for (unsigned i = threadIdx.x; i < 16384u; i += blockDim.x)
		poison_words[i] = word;
	__syncthreads();
	if (poison_words[(threadIdx.x * 7u) & 16383u] != word) // (keep the stores)
		__builtin_trap();
}
static hipError_t
poison_lds(int n_cu, hipStream_t st)
{
#ifdef ARKS_DEBUG_KNOBS
	static const char* e = std::getenv("ARKS_DEBUG_POISON_LDS");
#else
	static const char* e = nullptr; // the release build reads no environment (include/arks_hip.h, arks_build_options)
#endif
	if (!e)
		return hipSuccess;
	const unsigned word = (unsigned)std::strtoul(e, nullptr, 16);
	poison_lds_kernel<<<(unsigned)(n_cu > 0 ? n_cu : 256) * 12u, 256, 65536, st>>>(word);
	// ... and more of it on a stream of its own, to get between the launches of the call (hot, medium, slow
	// kernel) the way another process's workgroups would
	static hipStream_t side = nullptr;
	if (!side && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess)
		return hipGetLastError();
	for (int i = 0; i < 6; ++i)
		poison_lds_kernel<<<(unsigned)(n_cu > 0 ? n_cu : 256) * 4u, 256, 65536, side>>>(word ^ (unsigned)i * 0x01010101u);
	return hipGetLastError();
}

namespace {

thread_local std::string g_last_error;

int
fail_hip(hipError_t e, const char* what)
{
	g_last_error = std::string(what) + ": " + hipGetErrorString(e);
	(void)hipGetLastError(); // clear the sticky launch error, if any
	return e == hipErrorOutOfMemory ? ARKS_ERR_OOM : ARKS_ERR_HIP;
}

#ifdef ARKS_DEBUG_KNOBS
static bool g_trace = std::getenv("ARKS_TRACE") != nullptr;
#else
constexpr bool g_trace = false;
#endif
static double
trace_ms()
{
	static const auto t0 = std::chrono::steady_clock::now();
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define ARKS_TRACE_STEP(name)                                                                      \
	do {                                                                                           \
		if (g_trace) {                                                                             \
			hipError_t te_ = hipDeviceSynchronize();                                               \
			std::fprintf(stderr, "[arks] %10.1f ms  %s: %s\n", trace_ms(), name, hipGetErrorString(te_)); \
			std::fflush(stderr);                                                                   \
		}                                                                                          \
	} while (0)

#define HIP_TRY(expr)                                                                              \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess) {                                                                    \
			rc = fail_hip(e_, #expr);                                                              \
			goto done;                                                                             \
		}                                                                                          \
	} while (0)

bool
device_is_gfx950(int dev)
{
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, dev) != hipSuccess)
		return false;
	return std::strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

int
check_k(int k)
{
	if (k <= 3 || k == 6 || k == 10)
		return ARKS_ERR_BAD_K;
	if (k > ARKS_MAX_K)
		return ARKS_ERR_K_UNSUPPORTED;
	return ARKS_OK;
}

// RAII device buffer for temporaries
struct DevBuf
{
	void* p = nullptr;
	~DevBuf()
	{
		if (p)
			(void)hipFree(p);
	}
	hipError_t alloc(size_t bytes) // a second call replaces the buffer
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
		return hipMalloc(&p, bytes ? bytes : 1);
	}
	template <typename T>
	T* as() const
	{
		return static_cast<T*>(p);
	}
};

// The big buffers of an index build behind ONE address range (HIP's virtual-memory API): the exact table of the build
// (64 B per visited window: 90 GB for a 3 Gbp draft) dies before the seed table (32 B per position, 45 GB, kept) and its
// count table (24 B per position, scratch) are made.  Freed and allocated anew, those 79 GB cost seconds -- a large
// hipMalloc that follows a large hipFree waits for the driver's wipe of released VRAM (profiles/r13_alloc_cost.txt:
// 4-6 s for 45-90 GB; the CLI's index build spent 5.5 of its 6.7 s in such calls) -- so the range is made of two
// physical blocks: HEAD, sized for the seed table, which the index keeps, and TAIL, which takes the count table and goes
// back at the end of the build.  Any call of the API failing = the classic path (hipMalloc per buffer).
struct VmArena
{
	char* va = nullptr;
	size_t head = 0, tail = 0; // bytes of the two blocks (multiples of kGran)
	hipMemGenericAllocationHandle_t h[2] = {};
	bool have[2] = { false, false }, mapped[2] = { false, false };
	static constexpr size_t kGran = (size_t)2 << 20;
	static size_t up(size_t b) { return (b + kGran - 1) / kGran * kGran; }
	bool ok() const { return va != nullptr; }
	// head_bytes + tail_bytes >= what the exact table needs; false = nothing is held
	bool create(size_t head_bytes, size_t tail_bytes, int device)
	{
		head = up(head_bytes), tail = up(tail_bytes);
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = device;
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		void* p = nullptr;
		bool good = hipMemAddressReserve(&p, head + tail, kGran, nullptr, 0) == hipSuccess && p;
		va = good ? static_cast<char*>(p) : nullptr;
		const size_t sz[2] = { head, tail };
		for (int b = 0; b < 2 && good; ++b) {
			good = hipMemCreate(&h[b], sz[b], &prop, 0) == hipSuccess;
			have[b] = good;
			if (good) {
				good = hipMemMap(va + (b ? head : 0), sz[b], 0, h[b], 0) == hipSuccess;
				mapped[b] = good;
			}
		}
		good = good && hipMemSetAccess(va, head + tail, &acc, 1) == hipSuccess;
		if (!good) {
			(void)hipGetLastError();
			release_all();
		}
		return good;
	}
	void release_block(int b)
	{
		if (mapped[b])
			(void)hipMemUnmap(va + (b ? head : 0), b ? tail : head);
		if (have[b])
			(void)hipMemRelease(h[b]);
		mapped[b] = have[b] = false;
	}
	void release_all()
	{
		release_block(1);
		release_block(0);
		if (va)
			(void)hipMemAddressFree(va, head + tail);
		va = nullptr;
	}
	~VmArena() { release_all(); }
};

struct DeviceGuard
{
	int prev = -1;
	explicit DeviceGuard(int dev)
	{
		(void)hipGetDevice(&prev);
		if (prev != dev)
			(void)hipSetDevice(dev);
		else
			prev = -1;
	}
	~DeviceGuard()
	{
		if (prev >= 0)
			(void)hipSetDevice(prev);
	}
};

} // namespace

struct arks_index
{
	int k = 0;
	int kw = 0;
	int device = 0;
	int n_cu = 256;
	int kind = 0; // 0 = plain hash table, 1 = locality index (text + minimizer table)
	KeyGeom geom;
	TableView table{ nullptr, 0 }; // kind 0: the index
	BIndexView bx{};               // kind 1: the index (views into the buffers below)
	u64* codes = nullptr;
	u32* visited = nullptr;
	u32* ambig = nullptr;
	u32* word_owner = nullptr;
	u64* mtab = nullptr;
	// mtab lies in the head block of the build's arena (VmArena) instead of in a hipMalloc block of its own: the
	// address range [mtab_va, + mtab_va_bytes) is reserved, its first mtab_map_bytes are mapped to mtab_handle
	char* mtab_va = nullptr;
	size_t mtab_va_bytes = 0, mtab_map_bytes = 0;
	hipMemGenericAllocationHandle_t mtab_handle = {};
	u64* trec = nullptr; // seed index: text records (BIndexView::trec)
	// seed table sharded over ranks (arks_index_build_seed_shard): `mtab` holds the seeds this rank owns, the
	// hot kernel gets its probes answered by their owners; the general kernels (medium, slow) work from a
	// replicated minimizer table (bxg), ~20x smaller than the whole seed table
	int seed_rank = 0, seed_ranks = 1;
	u64* mtab_gen = nullptr;
	BIndexView bxg{};
	u64 text_words = 0;  // words of text incl. front padding (excl. back padding)
	u64 alloc_words = 0;
	int64_t n_keys = 0;
	int64_t n_visited = 0;
	int64_t n_minimizers = 0, n_fallback = 0;
	// Work queues of the map kernels (indices of the reads that take the medium / slow path, their lengths,
	// the work counters and the partial statistics): one set per stream that maps against this index, so that
	// map calls on different streams may run at the same time (calls on one stream are ordered by the stream).
	struct QueueSet
	{
		u32* queue = nullptr;
		u32* queue_count = nullptr;
		int64_t cap = 0;
		u64 gen = 0; // ensure_queue calls on this set so far: who holds a copy sees whether anybody else has used it since
	};
	mutable std::mutex queue_m;
	mutable std::vector<std::pair<void*, QueueSet>> queues; // (stream, its set): a handful at most
	mutable void* last_stream = nullptr;                   // for arks_debug_queue_counts
};

struct arks_imap
{
	int device = 0;
	ImapView v{ nullptr, nullptr, nullptr, 0, nullptr, nullptr };
	// host-side bound of the occupied slots: the last exact count plus the pairs submitted since (a pair
	// adds at most one entry); the table is rebuilt larger before the bound could pass half the slots
	u64 bound = 0;
	u64 seq_base = 0; // arks_imap_set_pair_base
};

extern "C" {

int
arks_abi_version(void)
{
#ifdef ARKS_CALIBRATION_BUILD
	return -ARKS_ABI_VERSION; /* results wrong by design: no product caller accepts this library */
#else
	return ARKS_ABI_VERSION;
#endif
}

const char*
arks_strerror(int status)
{
	switch (status) {
	case ARKS_OK: return "ok";
	case ARKS_ERR_BAD_K: return "k-mer size must be > 3 (and not 6 or 10)";
	case ARKS_ERR_K_UNSUPPORTED: return "k-mer size above ARKS_MAX_K";
	case ARKS_ERR_OOM: return "out of memory";
	case ARKS_ERR_HIP: return "HIP runtime error";
	case ARKS_ERR_NO_DEVICE: return "no gfx950 (MI355X) device available";
	case ARKS_ERR_BAD_ARG: return "bad argument";
	case ARKS_ERR_FULL: return "accumulator table full";
	default: return "unknown status";
	}
}

const char*
arks_last_error_string(void)
{
	return g_last_error.c_str();
}

int
arks_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	int ok = 0;
	for (int d = 0; d < n; ++d)
		ok += device_is_gfx950(d);
	return ok;
}

int
arks_key_bytes(int k)
{
	return k / 4 + ((k % 4) ? 1 : 0);
}

int
arks_end_cutoff(int len, int min_size, int end_length, int* cutoff)
{
	if (len < min_size)
		return 0;
	int c = end_length;
	if (c == 0 || len <= c * 2)
		c = len / 2;
	if (cutoff)
		*cutoff = c;
	return 1;
}

int
arks_word_offsets(const uint32_t* h_lens, int64_t n, uint64_t* h_word_off)
{
	if (n < 0 || (n > 0 && (!h_lens || !h_word_off)) || !h_word_off)
		return ARKS_ERR_BAD_ARG;
	uint64_t acc = 0;
	for (int64_t i = 0; i < n; ++i) {
		h_word_off[i] = acc;
		acc += ((uint64_t)h_lens[i] + 31) / 32;
	}
	h_word_off[n] = acc;
	return ARKS_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* packing                                                                                        */
/* ---------------------------------------------------------------------------------------------- */

static int
require_device(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
		(void)hipGetLastError();
		g_last_error = "no HIP device visible";
		return ARKS_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= n)
		return ARKS_ERR_BAD_ARG;
	if (!device_is_gfx950(device)) {
		g_last_error = "device is not gfx950";
		return ARKS_ERR_NO_DEVICE;
	}
	return ARKS_OK;
}

int
arks_pack_reads_device(
    const uint8_t* d_ascii,
    const uint64_t* d_offsets,
    const uint32_t* d_lens,
    const uint64_t* d_word_off,
    int64_t n_reads,
    uint64_t* d_codes,
    uint32_t* d_nmask,
    uint8_t* d_read_class,
    int device,
    void* stream)
{
	if (n_reads < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_ascii || !d_offsets || !d_lens || !d_word_off || !d_codes || !d_nmask)
		return ARKS_ERR_BAD_ARG;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	hipStream_t st = static_cast<hipStream_t>(stream);
	DevBuf ncount, other;
	u64 total_words = 0;
	HIP_TRY(hipMemcpyAsync(&total_words, d_word_off + n_reads, sizeof(u64), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (d_read_class) {
		HIP_TRY(ncount.alloc(sizeof(u32) * (size_t)n_reads));
		HIP_TRY(other.alloc(sizeof(u32) * (size_t)n_reads));
		HIP_TRY(hipMemsetAsync(ncount.p, 0, sizeof(u32) * (size_t)n_reads, st));
		HIP_TRY(hipMemsetAsync(other.p, 0, sizeof(u32) * (size_t)n_reads, st));
	}
	HIP_TRY(launch_pack(
	    d_ascii, (const u64*)d_offsets, d_lens, (const u64*)d_word_off, (long)n_reads, total_words,
	    (u64*)d_codes, d_nmask, ncount.as<u32>(), other.as<u32>(), st));
	if (d_read_class) {
		HIP_TRY(launch_read_class(d_lens, ncount.as<u32>(), other.as<u32>(), (long)n_reads, d_read_class, st));
		HIP_TRY(hipStreamSynchronize(st)); // temporaries are freed on return
	}
done:
	return rc;
}

// 8 bases at a time for the host packer: 2-bit codes by ((c >> 1) ^ (c >> 2)) & 3 (A, C, G, T in either
// case -> 0..3), validity by byte-wise equality tests, the per-byte bits gathered with PEXT after a
// byte swap so that the first base lands in the most significant position.  *codes16 = 8 codes,
// *bad8 = bit (7 - j) set when base j is not one of ACGTacgt, *n8 = ... is N or n.
__attribute__((target("bmi2"))) static inline void
pack8_bmi2(const unsigned char* p, uint32_t* codes16, uint32_t* bad8, uint32_t* n8)
{
	uint64_t x;
	std::memcpy(&x, p, 8);
	x = __builtin_bswap64(x);
	const uint64_t k01 = 0x0101010101010101ull, k7f = 0x7F7F7F7F7F7F7F7Full, k80 = 0x8080808080808080ull;
	const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull; // upper case
	auto eq = [&](unsigned char ch) { // 0x80 in every byte of u that equals ch (exact, no carries across bytes)
		const uint64_t v = u ^ (k01 * ch);
		return ~(((v & k7f) + k7f) | v | k7f);
	};
	const uint64_t valid = eq('A') | eq('C') | eq('G') | eq('T');
	const uint64_t isn = eq('N');
	const uint64_t two = ((x >> 1) ^ (x >> 2)) & (k01 * 3) & ((valid >> 7) * 3);
	*codes16 = (uint32_t)__builtin_ia32_pext_di(two, k01 * 3);
	*bad8 = (uint32_t)__builtin_ia32_pext_di(~valid & k80, k80);
	*n8 = (uint32_t)__builtin_ia32_pext_di(isn, k80);
}

// 32 bases -> one code word, one N-mask word, the invalid / N byte masks (bit j = base j): AVX2
__attribute__((target("avx2"))) static inline void
pack32_avx2(const unsigned char* p, uint64_t* codes, uint32_t* bad, uint32_t* isn)
{
	const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
	const __m256i u = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF)); // upper case
	const __m256i va = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('A')), vc = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('C'));
	const __m256i vg = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('G')), vt = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('T'));
	const __m256i valid = _mm256_or_si256(_mm256_or_si256(va, vc), _mm256_or_si256(vg, vt));
	const __m256i vn = _mm256_cmpeq_epi8(u, _mm256_set1_epi8('N'));
	// A C G T -> 0 1 2 3: ((c >> 1) ^ (c >> 2)) & 3 (16-bit shifts: the bits that cross bytes are masked off)
	__m256i two = _mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2));
	two = _mm256_and_si256(_mm256_and_si256(two, _mm256_set1_epi8(3)), valid);
	// four codes per byte, first base in the top bits: (b0 * 4 + b1) per 16-bit lane, then (.. * 16 + ..) per 32-bit lane
	const __m256i p16 = _mm256_maddubs_epi16(two, _mm256_set1_epi16(0x0104));
	const __m256i p32 = _mm256_madd_epi16(p16, _mm256_set1_epi32(0x00010010));
	// the low byte of the eight 32-bit lanes, lane 0 (bases 0..3) most significant
	const __m256i sh = _mm256_shuffle_epi8(
	    p32, _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 12, 8, 4, 0, -1, -1, -1, -1, -1, -1,
	                          -1, -1, -1, -1, -1, -1));
	const uint32_t lo = (uint32_t)_mm256_extract_epi32(sh, 0); // bases 0..15: byte 3 = bases 0..3
	const uint32_t hi = (uint32_t)_mm256_extract_epi32(sh, 4); // bases 16..31
	*codes = ((uint64_t)lo << 32) | hi;
	*bad = ~(uint32_t)_mm256_movemask_epi8(valid);
	*isn = (uint32_t)_mm256_movemask_epi8(vn);
}

static inline uint32_t
bitrev32(uint32_t v)
{
	v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
	v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
	v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
	return __builtin_bswap32(v);
}

int
arks_pack_reads_host(
    const char* h_ascii,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    const uint64_t* h_word_off,
    int64_t n_reads,
    uint64_t* h_codes,
    uint32_t* h_nmask,
    uint8_t* h_read_class)
{
	if (n_reads < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!h_ascii || !h_offsets || !h_lens || !h_word_off || !h_codes || !h_nmask)
		return ARKS_ERR_BAD_ARG;
	static uint8_t cls[256];
	static bool ready = false;
	if (!ready) {
		std::memset(cls, 5, sizeof cls);
		cls['A'] = cls['a'] = 0;
		cls['C'] = cls['c'] = 1;
		cls['G'] = cls['g'] = 2;
		cls['T'] = cls['t'] = 3;
		cls['N'] = cls['n'] = 4;
		ready = true;
	}
	const bool fast = __builtin_cpu_supports("bmi2");
	const bool wide = __builtin_cpu_supports("avx2");
	for (int64_t r = 0; r < n_reads; ++r) {
		const unsigned char* s = reinterpret_cast<const unsigned char*>(h_ascii) + h_offsets[r];
		const uint32_t len = h_lens[r];
		uint64_t* cw = h_codes + h_word_off[r];
		uint32_t* mw = h_nmask + h_word_off[r];
		const uint64_t nw = ((uint64_t)len + 31) / 32;
		uint32_t nn = 0, other = 0;
		for (uint64_t w = 0; w < nw; ++w) {
			uint64_t c = 0;
			uint32_t m = 0;
			const uint32_t n = std::min<uint32_t>(32, len - (uint32_t)(w * 32));
			uint32_t i = 0;
			if (wide && n == 32) { // a whole word at once
				uint32_t bad, isn;
				pack32_avx2(s + w * 32, &c, &bad, &isn);
				cw[w] = c;
				mw[w] = bitrev32(bad); // bit 31 = the word's first base
				nn += (uint32_t)__builtin_popcount(isn);
				other |= bad & ~isn;
				continue;
			}
			if (wide && r + 1 < n_reads && h_offsets[r + 1] >= h_offsets[r] + w * 32 + 32) {
				// the last, partial word the same way: the 32 bytes lie in front of the next read's text, i.e.
				// inside the caller's buffer; what is behind the read's end is masked out
				uint32_t bad, isn;
				pack32_avx2(s + w * 32, &c, &bad, &isn);
				const uint32_t in = (1u << n) - 1u; // n < 32 here
				cw[w] = c & ~(~0ull >> (2 * n));
				mw[w] = bitrev32(bad & in);
				nn += (uint32_t)__builtin_popcount(isn & in);
				other |= bad & ~isn & in;
				continue;
			}
			if (fast)
				for (; i + 8 <= n; i += 8) {
					uint32_t codes16, bad8, n8;
					pack8_bmi2(s + w * 32 + i, &codes16, &bad8, &n8);
					c |= (uint64_t)codes16 << (48 - 2 * i);
					m |= bad8 << (24 - i);
					nn += (uint32_t)__builtin_popcount(n8);
					other |= bad8 & ~n8;
				}
			for (; i < n; ++i) {
				const uint32_t x = cls[s[w * 32 + i]];
				c |= (uint64_t)(x < 4 ? x : 0) << (62 - 2 * i);
				m |= (uint32_t)(x >= 4) << (31 - i);
				nn += x == 4;
				other |= x == 5;
			}
			cw[w] = c;
			mw[w] = m;
		}
		if (h_read_class) { // checkReadSequence, Arcs/Arcs.cpp:366-389
			const double ar = (double)nn / (double)len;
			const uint8_t ok = (other == 0 && !(ar > 0.02)) ? 1 : 0;
			h_read_class[r] = (uint8_t)(ok | ((ok && nn == 0) ? 2 : 0)); // bit 1: ACGT only
		}
	}
	return ARKS_OK;
}

/* ---------------------------------------------------------------------------------------------- */
/* index                                                                                          */
/* ---------------------------------------------------------------------------------------------- */

// The layout choices of a build (include/arks_hip.h arks_build_options), validated.  Results never depend on them.
struct BuildChoice
{
	int kind = ARKS_INDEX_AUTO;
	u32 heavy = (u32)kHeavy; // occurrences of an m-mer beyond which it is heavy (arks_device.hpp kHeavy)
	int mlen = 0;            // 0: by k
	int fb_load_inv = 0;     // 0: by free memory
};

static int
want_locality(int k, const BuildChoice& c)
{
	if (c.kind == ARKS_INDEX_HASH)
		return 0;
	return k >= 20; // below that the minimizer window degenerates; the hash table serves
}

// Seed index (every m-mer position in the table, fixed-position seeds on the read side) or minimizer
// index (minimizer positions only, ~20x smaller table, the read side computes its minimizers)?  The
// seed index is the fast one; it is chosen when its table fits comfortably, unless the caller's options say which.
static bool
want_seeds(u64 text_positions, int device, const BuildChoice& c)
{
	if (c.kind == ARKS_INDEX_MINIMIZER)
		return false;
	if (c.kind == ARKS_INDEX_SEEDS)
		return true;
	size_t free_b = 0, total_b = 0;
	(void)device;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
		return false;
	// the table (ARKS_SEED_LOAD_INV slots of 8 B per position) plus the count table of the build (24 B per
	// position) within half of what is free now
	return (double)text_positions * (8.0 * ARKS_SEED_LOAD_INV + 24.0) < 0.5 * (double)free_b;
}

/* Which shard holds which contig end (arks_shard_of_ends): the head and the tail of a contig (ends 2i and
 * 2i + 1, conreci 2i + 1 and 2i + 2, Arcs.cpp:1079-1081) stay together; contigs are dealt in list order,
 * each to the shard that holds the fewest bases so far (ties: the lowest shard) -- the shards differ by
 * less than one contig whatever the order of the draft, and every rank derives the same assignment. */
static void
assign_shards(const uint32_t* h_lens, int64_t n_ends, int n_shards, std::vector<int32_t>& shard_of)
{
	shard_of.assign((size_t)n_ends, 0);
	if (n_shards <= 1)
		return;
	std::vector<uint64_t> load((size_t)n_shards, 0);
	for (int64_t e = 0; e < n_ends; e += 2) {
		int best = 0;
		for (int s = 1; s < n_shards; ++s)
			if (load[(size_t)s] < load[(size_t)best])
				best = s;
		load[(size_t)best] += h_lens[e];
		shard_of[(size_t)e] = best;
		if (e + 1 < n_ends) {
			load[(size_t)best] += h_lens[e + 1];
			shard_of[(size_t)e + 1] = best;
		}
	}
}

/* The ends of the OTHER shards take away the owner of every key they share with this shard's table
 * (launch_poison): packed and scanned with the same kernels as the shard's own text, a bounded number
 * of bases at a time, nothing of them is kept. */
static int
poison_with_foreign_ends(
    const arks_index* idx, const char* h_bases, const uint64_t* h_offsets, const uint32_t* h_lens,
    int64_t n_ends, int shard, const std::vector<int32_t>& shard_of, TableView full, u64* d_counter,
    bool first_holder, hipStream_t st)
{
	int rc = ARKS_OK;
	const uint64_t kChunkBases = 1ull << 28;
	const int kBackPad = 16;
	std::vector<uint64_t> woff, offs, src;
	std::vector<uint32_t> lens, conreci;
	DevBuf d_ascii, d_offs, d_lens, d_woff, d_codes, d_nmask, d_visited, d_scratch, d_wend, d_conreci;
	HIP_TRY(d_scratch.alloc(sizeof(u64) * 8));
	for (int64_t e = 0; e < n_ends;) {
		woff.clear(), offs.clear(), src.clear(), lens.clear(), conreci.clear();
		uint64_t acc = kFrontPadWords, bacc = 0;
		for (; e < n_ends && (bacc == 0 || bacc + h_lens[e] <= kChunkBases); ++e) {
			if (shard_of[(size_t)e] == shard || h_lens[e] == 0)
				continue;
			woff.push_back(acc), offs.push_back(bacc), src.push_back(h_offsets[e]), lens.push_back(h_lens[e]);
			conreci.push_back((uint32_t)e + 1u);
			acc += ((uint64_t)h_lens[e] + 31) / 32;
			bacc += h_lens[e];
		}
		const size_t n = lens.size();
		if (n == 0)
			continue;
		woff.push_back(acc), offs.push_back(bacc);
		const u64 text_words = acc, alloc_words = acc + kBackPad;
		HIP_TRY(d_ascii.alloc(bacc + 64));
		for (size_t i = 0; i < n;) { // one copy per run of ends that are adjacent in the source
			size_t last = i;
			while (last + 1 < n && src[last + 1] == src[last] + lens[last])
				++last;
			HIP_TRY(hipMemcpy(
			    d_ascii.as<char>() + offs[i], h_bases + src[i], offs[last] + lens[last] - offs[i],
			    hipMemcpyHostToDevice));
			i = last + 1;
		}
		lens.push_back(0);
		HIP_TRY(d_offs.alloc(sizeof(u64) * (n + 1)));
		HIP_TRY(d_lens.alloc(sizeof(u32) * (n + 1)));
		HIP_TRY(d_woff.alloc(sizeof(u64) * (n + 1)));
		HIP_TRY(d_codes.alloc(sizeof(u64) * alloc_words));
		HIP_TRY(d_nmask.alloc(sizeof(u32) * alloc_words));
		HIP_TRY(d_visited.alloc(sizeof(u32) * alloc_words));
		HIP_TRY(hipMemcpy(d_offs.p, offs.data(), sizeof(u64) * (n + 1), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(d_lens.p, lens.data(), sizeof(u32) * (n + 1), hipMemcpyHostToDevice));
		HIP_TRY(hipMemcpy(d_woff.p, woff.data(), sizeof(u64) * (n + 1), hipMemcpyHostToDevice));
		HIP_TRY(hipMemsetAsync(d_codes.p, 0, sizeof(u64) * alloc_words, st));
		HIP_TRY(hipMemsetAsync(d_nmask.p, 0, sizeof(u32) * alloc_words, st));
		HIP_TRY(hipMemsetAsync(d_visited.p, 0, sizeof(u32) * alloc_words, st));
		HIP_TRY(hipMemsetAsync(d_scratch.p, 0, sizeof(u64) * 8, st));
		HIP_TRY(launch_pack(
		    d_ascii.as<uint8_t>(), d_offs.as<u64>(), d_lens.as<u32>(), d_woff.as<u64>(), (long)n, text_words,
		    d_codes.as<u64>(), d_nmask.as<u32>(), nullptr, nullptr, st));
		HIP_TRY(launch_visit(
		    d_nmask.as<u32>(), d_woff.as<u64>(), d_lens.as<u32>(), (long)n, idx->k, d_visited.as<u32>(),
		    d_scratch.as<u64>(), st));
		if (first_holder) {
			// counters wanted: the key also learns the smallest foreign end that holds it (arks_shard_stats.hpp)
			HIP_TRY(d_wend.alloc(sizeof(u32) * alloc_words));
			HIP_TRY(d_conreci.alloc(sizeof(u32) * n));
			HIP_TRY(hipMemcpy(d_conreci.p, conreci.data(), sizeof(u32) * n, hipMemcpyHostToDevice));
			HIP_TRY(launch_word_owner(d_woff.as<u64>(), (long)n, alloc_words, d_wend.as<u32>(), st));
			HIP_TRY(launch_poison_min(
			    idx->kw, d_codes.as<u64>(), d_visited.as<u32>(), text_words, idx->geom, full, d_wend.as<u32>(),
			    d_conreci.as<u32>(), d_counter, st));
		} else
			HIP_TRY(launch_poison(
			    idx->kw, d_codes.as<u64>(), d_visited.as<u32>(), text_words, idx->geom, full, d_counter, st));
		HIP_TRY(hipStreamSynchronize(st)); // the buffers are reused (or freed) next
	}
done:
	return rc;
}

static int
index_build_impl(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_all_lens,
    int64_t n_ends,
    int shard,
    int n_shards,
    int device,
    arks_build_stats* stats,
    int seed_rank = 0,
    int seed_ranks = 1,
    const BuildChoice& choice = BuildChoice())
{
	if (!out || n_ends < 0 || (n_ends > 0 && (!h_bases || !h_offsets || !h_all_lens)) || n_shards < 1 ||
	    shard < 0 || shard >= n_shards || seed_ranks < 1 || seed_rank < 0 || seed_rank >= seed_ranks)
		return ARKS_ERR_BAD_ARG;
	*out = nullptr;
	// a shard sees the other shards' ends as empty strings: same conreci numbering, none of their k-mers
	std::vector<uint32_t> own_lens;
	std::vector<int32_t> shard_of;
	const uint32_t* h_lens = h_all_lens;
	if (n_shards > 1) {
		assign_shards(h_all_lens, n_ends, n_shards, shard_of);
		own_lens.assign(h_all_lens, h_all_lens + n_ends);
		for (int64_t e = 0; e < n_ends; ++e)
			if (shard_of[(size_t)e] != shard)
				own_lens[(size_t)e] = 0;
		h_lens = own_lens.data();
	}
	int rc = check_k(k);
	if (rc != ARKS_OK)
		return rc;
	rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);

	arks_index* idx = new (std::nothrow) arks_index;
	if (!idx)
		return ARKS_ERR_OOM;
	idx->k = k;
	idx->kw = key_words_for_k(k);
	idx->device = device;
	idx->geom = make_geom(k, idx->kw);
	{
		hipDeviceProp_t p;
		if (hipGetDeviceProperties(&p, device) == hipSuccess)
			idx->n_cu = p.multiProcessorCount;
	}

	hipStream_t st = nullptr;
	const int kBackPad = 16;
	std::vector<uint64_t> word_off((size_t)n_ends + 1, 0), offs((size_t)n_ends + 1, 0);
	uint64_t total_bases = 0;
	DevBuf d_ascii, d_offs, d_lens, d_woff, d_nmask, d_counters, d_full;
	VmArena arena; // (locality index with the whole seed table: the exact table, then the seed table + its count table)
	DevBuf d_ismin, d_ispal, d_isimg, d_heavy, d_ckeys, d_ccnts, d_wown;
	u64 text_words = 0, alloc_words = 0, counters[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	u64 visited_total = 0, n_min = 0, n_pal = 0, n_fb = 0, ccap = 0, mcap = 0;
	TableView full{ nullptr, 0 };
	int mm = kMShort, w = 0;
	bool locality = false, dense = false;
	size_t bm_bytes = 0;

	{
		uint64_t acc = kFrontPadWords, bacc = 0; // the text starts kFrontPadWords into its arrays
		for (int64_t e = 0; e < n_ends; ++e) {
			word_off[(size_t)e] = acc;
			offs[(size_t)e] = bacc;
			acc += ((uint64_t)h_lens[e] + 31) / 32;
			bacc += h_lens[e];
		}
		word_off[(size_t)n_ends] = acc;
		offs[(size_t)n_ends] = bacc;
		total_bases = bacc;
		text_words = acc;
		alloc_words = acc + kBackPad;
	}
	// minimizer length: 21-mers stay specific at any text size (a 15-mer sees ~ text / 5.4e8 chance
	// occurrences, and every chance occurrence is a third diagonal for the hot kernel); the short one
	// (17) only where k leaves no room (k < 24).  arks_build_options.minimizer_len overrides (tests).
	mm = k >= kMLong + 3 ? kMLong : kMShort; // measured: from k = 24 on the 21-mer wins even with a 4-position window
	if (choice.mlen)
		mm = (choice.mlen >= 19 && k >= kMLong + 2) ? kMLong : kMShort;
	// text positions are 32-bit in the minimizer table; and with the short minimizer (some 60 runs per
	// read at k = 20) a text beyond ~2.5e8 positions proposes chance diagonals for nearly every read:
	// measured at 1 Gbp, k = 20: 86 ms per 4 M pairs against 52 ms for the plain hash table
	locality = want_locality(k, choice) && alloc_words * 32ull < 0xFFFF0000ull &&
	           !(mm == kMShort && alloc_words * 32ull > 250000000ull && !choice.mlen);
	w = k - mm + 1;
	bm_bytes = sizeof(u32) * alloc_words;

	ARKS_TRACE_STEP("index build: start");
	HIP_TRY(d_ascii.alloc(total_bases + 64));
	// the source need not be contiguous: one copy per run of ends that are adjacent in it (a caller that
	// concatenates its ends gets a single copy instead of one per end)
	for (int64_t e = 0; e < n_ends;) {
		int64_t last = e;
		while (last + 1 < n_ends && h_offsets[last + 1] == h_offsets[last] + h_lens[last])
			++last;
		const uint64_t bytes = offs[(size_t)last] + h_lens[last] - offs[(size_t)e];
		if (bytes)
			HIP_TRY(hipMemcpy(
			    d_ascii.as<char>() + offs[(size_t)e], h_bases + h_offsets[e], bytes, hipMemcpyHostToDevice));
		e = last + 1;
	}
	ARKS_TRACE_STEP("contig ends uploaded");
	HIP_TRY(d_offs.alloc(sizeof(u64) * ((size_t)n_ends + 1)));
	HIP_TRY(d_lens.alloc(sizeof(u32) * ((size_t)n_ends + 1)));
	HIP_TRY(d_woff.alloc(sizeof(u64) * ((size_t)n_ends + 1)));
	HIP_TRY(hipMemcpy(d_offs.p, offs.data(), sizeof(u64) * ((size_t)n_ends + 1), hipMemcpyHostToDevice));
	if (n_ends)
		HIP_TRY(hipMemcpy(d_lens.p, h_lens, sizeof(u32) * (size_t)n_ends, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_woff.p, word_off.data(), sizeof(u64) * ((size_t)n_ends + 1), hipMemcpyHostToDevice));
	{
		void* p = nullptr;
		HIP_TRY(hipMalloc(&p, sizeof(u64) * alloc_words));
		idx->codes = static_cast<u64*>(p);
		HIP_TRY(hipMalloc(&p, bm_bytes));
		idx->visited = static_cast<u32*>(p);
	}
	HIP_TRY(d_nmask.alloc(bm_bytes));
	HIP_TRY(d_counters.alloc(sizeof(counters)));
	HIP_TRY(hipMemsetAsync(idx->codes, 0, sizeof(u64) * alloc_words, st));
	HIP_TRY(hipMemsetAsync(d_nmask.p, 0, bm_bytes, st));
	HIP_TRY(hipMemsetAsync(idx->visited, 0, bm_bytes, st));
	HIP_TRY(hipMemsetAsync(d_counters.p, 0, sizeof(counters), st));

	HIP_TRY(launch_pack(
	    d_ascii.as<uint8_t>(), d_offs.as<u64>(), d_lens.as<u32>(), d_woff.as<u64>(), (long)n_ends,
	    text_words, idx->codes, d_nmask.as<u32>(), nullptr, nullptr, st));
	ARKS_TRACE_STEP("launch_pack");
	HIP_TRY(launch_visit(
	    d_nmask.as<u32>(), d_woff.as<u64>(), d_lens.as<u32>(), (long)n_ends, k, idx->visited,
	    d_counters.as<u64>(), st));
	ARKS_TRACE_STEP("launch_visit");
	HIP_TRY(launch_popcount(idx->visited, text_words, d_counters.as<u64>() + 6, st));
	ARKS_TRACE_STEP("launch_popcount");
	HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	visited_total = counters[6];

	// the full exact table: load factor <= 0.5 over the visited windows (>= the distinct keys)
	full.cap = std::max<u64>(1024, visited_total * 2 + 64);
	{
		const size_t full_bytes = full.cap * kSlotWords * sizeof(u64);
		// head = what the seed table may need (every m-mer position of the visited windows: at most one per visited
		// window + w per end, palindromes' extras in the slack); a table that turns out larger gets a block of its own
		const size_t seed_bytes = sizeof(u64) * (size_t)ARKS_SEED_LOAD_INV * (size_t)(visited_total + (u64)(k - kMShort + 1) * (u64)n_ends);
		const size_t head_want = seed_bytes + seed_bytes / 64 + ((size_t)4 << 20);
		if (locality && seed_ranks == 1 && n_shards == 1 && full_bytes >= ((size_t)8 << 20) && head_want < full_bytes)
			(void)arena.create(head_want, full_bytes - VmArena::up(head_want) + VmArena::kGran, device);
		if (arena.ok())
			full.slots = reinterpret_cast<u64*>(arena.va);
		else {
			HIP_TRY(d_full.alloc(full_bytes));
			full.slots = d_full.as<u64>();
		}
	}
	ARKS_TRACE_STEP("exact table allocated");
	HIP_TRY(hipMemsetAsync(full.slots, 0, full.cap * kSlotWords * sizeof(u64), st));
	ARKS_TRACE_STEP("exact table cleared");
	HIP_TRY(d_wown.alloc(bm_bytes));
	HIP_TRY(launch_word_owner(d_woff.as<u64>(), (long)n_ends, alloc_words, d_wown.as<u32>(), st));
	ARKS_TRACE_STEP("launch_word_owner");
	HIP_TRY(launch_insert(
	    idx->kw, idx->codes, idx->visited, d_wown.as<u32>(), (long)n_ends, text_words, idx->geom, full,
	    d_counters.as<u64>(), st));
	ARKS_TRACE_STEP("launch_insert");
	if (n_shards > 1) {
		rc = poison_with_foreign_ends(
		    idx, h_bases, h_offsets, h_all_lens, n_ends, shard, shard_of, full, d_counters.as<u64>() + 7, true, st);
		if (rc != ARKS_OK)
			goto done;
		ARKS_TRACE_STEP("foreign ends");
	}
	if (stats)
		HIP_TRY(launch_build_stats(
		    idx->kw, idx->codes, idx->visited, d_woff.as<u64>(), (long)n_ends, text_words, idx->geom,
		    full, d_counters.as<u64>(), st));
	if (n_shards > 1)
		HIP_TRY(launch_count_first_holder(full, d_lens.as<u32>(), d_counters.as<u64>() + 8, st));
	ARKS_TRACE_STEP("launch_build_stats");
	HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	idx->n_keys = (int64_t)counters[2];
	idx->n_visited = (int64_t)visited_total;
	idx->text_words = text_words;
	idx->alloc_words = alloc_words;
	if (stats) {
		stats->total_kmers = visited_total;
		stats->null_kmers = counters[0];
		stats->short_ends = counters[1];
		if (n_shards > 1) {
			// this shard's share (include/arks_hip.h: the sums over the shards are the counters of the one map):
			// a key is recorded by the shard that holds the smallest end that visited it, the other ends' visits
			// are collisions wherever they are
			int64_t foreign = 0;
			for (int64_t e = 0; e < n_ends; ++e)
				foreign += shard_of[(size_t)e] != shard;
			stats->short_ends = counters[1] - (u64)foreign; // (the foreign ends were visited as empty strings)
			counters[2] = counters[8];
		}
		stats->recorded = counters[2];
		stats->collisions = visited_total - counters[2];
		stats->removed_dup = visited_total - counters[4];
		stats->unique = counters[5];
	}
	if (n_shards > 1) {
		// FIRST HOLDERS ONLY: a key that ends of other shards visit too is kept by the shard of the smallest end of
		// the list that visited it (the slot's smallest-end field, lowered by the foreign ends that streamed through);
		// the other shards take their visits of it back.  It reads 0 wherever it is, so no vote changes; with it in
		// one shard only, the found / duplicate counters of the read stage add up over the shards (arks_hip.h).
		HIP_TRY(hipMemsetAsync(d_counters.as<u64>() + 6, 0, sizeof(u64), st));
		HIP_TRY(launch_drop_later_holders(
		    idx->kw, idx->codes, idx->visited, d_lens.as<u32>(), text_words, idx->geom, full, d_counters.as<u64>() + 9, st));
		HIP_TRY(launch_popcount(idx->visited, text_words, d_counters.as<u64>() + 6, st));
		HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (counters[6] + counters[9] != visited_total) {
			g_last_error = "first holders: the visits kept and taken back do not add up";
			rc = ARKS_ERR_HIP;
			goto done;
		}
		visited_total = counters[6];
		idx->n_visited = (int64_t)visited_total;
		idx->n_keys = (int64_t)counters[8];
		ARKS_TRACE_STEP("first holders");
		if (!locality) {
			// the table is the index: the first holders' slots move to a table of their own
			TableView kept{ nullptr, std::max<u64>(1024, counters[8] * 2 + 64) };
			DevBuf d_kept;
			HIP_TRY(d_kept.alloc(kept.cap * kSlotWords * sizeof(u64)));
			kept.slots = d_kept.as<u64>();
			HIP_TRY(hipMemsetAsync(kept.slots, 0, kept.cap * kSlotWords * sizeof(u64), st));
			HIP_TRY(launch_keep_first_holders(idx->kw, full, d_lens.as<u32>(), kept, st));
			HIP_TRY(hipStreamSynchronize(st));
			std::swap(d_full.p, d_kept.p); // (d_kept frees the old table)
			full = kept;
		}
	}

	if (!locality) {
		idx->kind = 0;
		idx->table = full;
		d_full.p = nullptr; // ownership moves to the index
		(void)hipFree(idx->codes);
		(void)hipFree(idx->visited);
		idx->codes = nullptr;
		idx->visited = nullptr;
	} else {
		// (decided here, with the exact table of the build -- 64 B per visited window -- still allocated: what
		// is free now is a lower bound of what the index may use)
		dense = seed_ranks > 1 || want_seeds(visited_total + (u64)w * (u64)n_ends, device, choice);
		idx->kind = dense ? 2 : 1;
		idx->seed_rank = seed_rank;
		idx->seed_ranks = seed_ranks;
		{
			void* p = nullptr;
			HIP_TRY(hipMalloc(&p, bm_bytes));
			idx->ambig = static_cast<u32*>(p);
			HIP_TRY(hipMalloc(&p, bm_bytes));
			idx->word_owner = static_cast<u32*>(p);
		}
		HIP_TRY(d_ismin.alloc(bm_bytes));
		HIP_TRY(d_ispal.alloc(bm_bytes));
		HIP_TRY(d_isimg.alloc(bm_bytes));
		HIP_TRY(d_heavy.alloc(bm_bytes));
		HIP_TRY(hipMemsetAsync(idx->ambig, 0, bm_bytes, st));
		HIP_TRY(hipMemsetAsync(d_ismin.p, 0, bm_bytes, st));
		HIP_TRY(hipMemsetAsync(d_ispal.p, 0, bm_bytes, st));
		HIP_TRY(hipMemsetAsync(d_isimg.p, 0, bm_bytes, st));
		HIP_TRY(hipMemsetAsync(d_heavy.p, 0, bm_bytes, st));
		HIP_TRY(hipMemsetAsync(d_counters.p, 0, sizeof(counters), st));
		HIP_TRY(hipMemcpyAsync(idx->word_owner, d_wown.p, bm_bytes, hipMemcpyDeviceToDevice, st));
		HIP_TRY(launch_bmark(
		    idx->kw, mm, idx->codes, idx->visited, text_words, idx->geom, full, w, dense && seed_ranks == 1, idx->ambig,
		    d_ismin.as<u32>(), d_ispal.as<u32>(), d_isimg.as<u32>(), st));
	ARKS_TRACE_STEP("launch_bmark");
		HIP_TRY(launch_popcount(d_ismin.as<u32>(), text_words, d_counters.as<u64>() + 0, st));
	ARKS_TRACE_STEP("launch_popcount");
		HIP_TRY(launch_popcount(d_ispal.as<u32>(), text_words, d_counters.as<u64>() + 1, st));
		HIP_TRY(launch_popcount(d_isimg.as<u32>(), text_words, d_counters.as<u64>() + 2, st));
	ARKS_TRACE_STEP("launch_popcount");
		HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		n_min = counters[0];
		n_pal = counters[1];
		idx->bx.has_img = counters[2] != 0;
		// the full table is no longer needed: every position now carries its value bits (in the arena its memory is taken
		// over by the tables below; a block of its own goes back now)
		if (d_full.p)
			(void)hipFree(d_full.p);
		d_full.p = nullptr;
		// One table = count the occurrences of every registered m-mer (heavy ones get one marker instead
		// of their positions), decree the m-mers of the quirk images heavy, fill.  With the seed table
		// sharded over ranks the counting runs once per owner -- every rank needs the heavy bits of ALL
		// seeds, because the fallback table (the same on every rank) holds the windows under heavy seeds --
		// but only this rank's seeds are filled in.
		auto build_table = [&](bool dense_t, int own_n, int own_r, u64 n_pos_total, const u64* pos_per_owner,
		                       u64** out_tab, u64* out_cap) -> int {
			int rc = ARKS_OK;
			const u64 pal_extra = (u64)(dense_t ? w : 4) * n_pal;
			for (int p_own = 0; p_own < own_n; ++p_own) {
				const u64 npos = own_n > 1 ? pos_per_owner[p_own] : n_pos_total;
				const bool mine = own_n == 1 || p_own == own_r;
				const u64 ccap_t = 2 * (npos + pal_extra) + 64;
				u64 mcap_t = 0;
				if (mine)
					mcap_t = ((u64)(dense_t ? ARKS_SEED_LOAD_INV : ARKS_MTAB_LOAD_INV) * (npos + pal_extra) + 64 + 3) & ~3ull; // whole groups of 4 (mtab_home)
				// in the arena (own_n == 1 there): the table in the head block -- which the index keeps --, the count table in
				// the tail block; whichever does not fit gets a block of its own
				const size_t ck_bytes = (sizeof(u64) * ccap_t + 255) & ~(size_t)255, cc_bytes = sizeof(u32) * ccap_t;
				// (the head block was sized for a seed table: a minimizer table is 20x smaller and must not pin it)
				const bool tab_in_arena = arena.ok() && mine && dense_t && out_tab == &idx->mtab && sizeof(u64) * mcap_t <= arena.head &&
				                          2 * sizeof(u64) * mcap_t >= arena.head;
				const bool cnt_in_arena = arena.ok() && ck_bytes + cc_bytes <= arena.tail;
				u64* ck = nullptr;
				u32* cc = nullptr;
				if (cnt_in_arena) {
					ck = reinterpret_cast<u64*>(arena.va + arena.head);
					cc = reinterpret_cast<u32*>(arena.va + arena.head + ck_bytes);
				} else {
					HIP_TRY(d_ckeys.alloc(sizeof(u64) * ccap_t));
					HIP_TRY(d_ccnts.alloc(sizeof(u32) * ccap_t));
					ck = d_ckeys.as<u64>(), cc = d_ccnts.as<u32>();
				}
				HIP_TRY(hipMemsetAsync(ck, 0, sizeof(u64) * ccap_t, st));
				HIP_TRY(hipMemsetAsync(cc, 0, sizeof(u32) * ccap_t, st));
				if (mine) {
					if (tab_in_arena)
						*out_tab = reinterpret_cast<u64*>(arena.va);
					else {
						void* p = nullptr;
						HIP_TRY(hipMalloc(&p, sizeof(u64) * mcap_t));
						*out_tab = static_cast<u64*>(p);
					}
					*out_cap = mcap_t;
					HIP_TRY(hipMemsetAsync(*out_tab, 0, sizeof(u64) * mcap_t, st));
				}
				HIP_TRY(launch_bcount(mm, idx->codes, d_ismin.as<u32>(), text_words, ck, cc, ccap_t, (u32)p_own, (u32)own_n, st));
				HIP_TRY(launch_bforce(idx->kw, mm, 0, idx->codes, d_ispal.as<u32>(), text_words, idx->geom, w, dense_t, ck, cc, ccap_t,
				                      nullptr, 0, (u32)p_own, (u32)own_n, st));
				HIP_TRY(launch_bfill_mtab(mm, idx->codes, d_ismin.as<u32>(), text_words, ck, cc, ccap_t, mine ? *out_tab : nullptr,
				                          mcap_t, d_heavy.as<u32>(), (u32)p_own, (u32)own_n, mine, choice.heavy, st));
				if (mine)
					HIP_TRY(launch_bforce(idx->kw, mm, 1, idx->codes, d_ispal.as<u32>(), text_words, idx->geom, w, dense_t, ck, cc, ccap_t,
					                      *out_tab, mcap_t, (u32)p_own, (u32)own_n, st));
				HIP_TRY(hipStreamSynchronize(st));
				if (tab_in_arena) {
					// the head block is the index's from here on: the tail goes back, the range stays reserved until the index is freed
					arena.release_block(1);
					idx->mtab_va = arena.va, idx->mtab_va_bytes = arena.head + arena.tail, idx->mtab_map_bytes = arena.head;
					idx->mtab_handle = arena.h[0];
					arena.va = nullptr, arena.have[0] = arena.mapped[0] = false; // (ownership moved: arks_index_free unmaps and releases)
				}
			}
		done:
			// (a table that lies in the arena and was not handed over: the arena releases it, the index must not hipFree it)
			if (rc != ARKS_OK && arena.ok() && *out_tab == reinterpret_cast<u64*>(arena.va))
				*out_tab = nullptr;
			return rc;
		};
		if (seed_ranks > 1) {
			// the general kernels' minimizer table first (is_min = minimizer positions at this point) ...
			u64 gcap = 0;
			rc = build_table(false, 1, 0, n_min, nullptr, &idx->mtab_gen, &gcap);
			if (rc != ARKS_OK)
				goto done;
			idx->bxg.mtab = idx->mtab_gen;
			idx->bxg.mtab_cap = gcap;
			ARKS_TRACE_STEP("general (minimizer) table");
			// ... then every m-mer position, and how many of them each rank owns
			HIP_TRY(hipMemsetAsync(d_heavy.p, 0, bm_bytes, st));
			HIP_TRY(launch_bdilate(idx->visited, text_words, w, d_ismin.as<u32>(), st));
			HIP_TRY(hipMemsetAsync(d_counters.p, 0, sizeof(counters), st));
			HIP_TRY(launch_popcount(d_ismin.as<u32>(), text_words, d_counters.as<u64>() + 0, st));
			HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			n_min = counters[0];
			std::vector<u64> per_owner((size_t)seed_ranks, 0);
			{
				DevBuf d_po;
				HIP_TRY(d_po.alloc(sizeof(u64) * (size_t)seed_ranks));
				HIP_TRY(hipMemsetAsync(d_po.p, 0, sizeof(u64) * (size_t)seed_ranks, st));
				HIP_TRY(launch_bowners(mm, idx->codes, d_ismin.as<u32>(), text_words, (u32)seed_ranks, d_po.as<u64>(), st));
				HIP_TRY(hipMemcpyAsync(per_owner.data(), d_po.p, sizeof(u64) * (size_t)seed_ranks, hipMemcpyDeviceToHost, st));
				HIP_TRY(hipStreamSynchronize(st));
			}
			rc = build_table(true, seed_ranks, seed_rank, n_min, per_owner.data(), &idx->mtab, &mcap);
			if (rc != ARKS_OK)
				goto done;
		} else {
			rc = build_table(dense, 1, 0, n_min, nullptr, &idx->mtab, &mcap);
			if (rc != ARKS_OK)
				goto done;
		}
		arena.release_all(); // (whatever of it the index did not take over)
		ARKS_TRACE_STEP("table");
		HIP_TRY(hipMemsetAsync(d_counters.p, 0, sizeof(counters), st));
		HIP_TRY(launch_bfallback(
		    idx->kw, mm, false, idx->codes, idx->visited, idx->ambig, d_ispal.as<u32>(), d_isimg.as<u32>(),
		    d_heavy.as<u32>(), idx->word_owner, text_words, idx->geom, w, dense, idx->table,
		    d_counters.as<u64>(), st));
	ARKS_TRACE_STEP("launch_bfallback");
		HIP_TRY(hipMemcpyAsync(counters, d_counters.p, sizeof(counters), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		n_fb = counters[0];
		// load 1/4 where the device has the room, 1/2 (rounds 1-4) where it has not: the medium kernel's exact-key probes
		// run at the part's random-access rate (human-like draft: 2.8e8 of them per 20 M pairs, 17 ms), and a search of a
		// linear-probing table at load 1/2 touches 2.5 slots when it misses and 1.5 when it hits -- 1.4 and 1.2 at 1/4
		{
			size_t free_b = 0, total_b = 0;
			const u64 want = 4 * n_fb + 64;
			const bool roomy = hipMemGetInfo(&free_b, &total_b) == hipSuccess && want * 32ull < (u64)free_b / 4;
			// (arks_build_options.fallback_load_inv = 2: the old load, for A/B runs; = 4: whatever the room)
			idx->table.cap = (choice.fb_load_inv == 4 || (roomy && choice.fb_load_inv != 2)) ? want : 2 * n_fb + 64;
		}
		{
			void* p = nullptr;
			HIP_TRY(hipMalloc(&p, idx->table.cap * kSlotWords * sizeof(u64)));
			idx->table.slots = static_cast<u64*>(p);
		}
		HIP_TRY(hipMemsetAsync(idx->table.slots, 0, idx->table.cap * kSlotWords * sizeof(u64), st));
		HIP_TRY(launch_bfallback(
		    idx->kw, mm, true, idx->codes, idx->visited, idx->ambig, d_ispal.as<u32>(), d_isimg.as<u32>(),
		    d_heavy.as<u32>(), idx->word_owner, text_words, idx->geom, w, dense, idx->table,
		    d_counters.as<u64>(), st));
	ARKS_TRACE_STEP("launch_bfallback");
		HIP_TRY(hipStreamSynchronize(st));
		idx->n_minimizers = (int64_t)n_min;
		idx->n_fallback = (int64_t)n_fb;
		idx->bx.codes = idx->codes;
		idx->bx.visited = idx->visited;
		idx->bx.ambig = idx->ambig;
		idx->bx.word_owner = idx->word_owner;
		idx->bx.mtab = idx->mtab;
		idx->bx.mtab_cap = mcap;
		idx->bx.fallback = idx->table;
		idx->bx.m = mm;
		idx->bx.w = w;
		idx->bx.enabled = 1;
		idx->bx.dense = dense ? 1 : 0;
		if (seed_ranks > 1) { // the general kernels' view: same text and fallback, the replicated minimizer table
			u64* gen_tab = idx->mtab_gen;
			const u64 gen_cap = idx->bxg.mtab_cap;
			idx->bxg = idx->bx;
			idx->bxg.mtab = gen_tab;
			idx->bxg.mtab_cap = gen_cap;
			idx->bxg.dense = 0;
		} else
			idx->bxg = idx->bx;
		if (dense) {
			void* p = nullptr;
			// 16-byte records, and behind them the owner per block of 32 words
			HIP_TRY(hipMalloc(&p, 2 * sizeof(u64) * alloc_words + sizeof(u32) * (alloc_words / 32 + 1)));
			idx->trec = static_cast<u64*>(p);
			u32* const owner_blk = reinterpret_cast<u32*>(idx->trec + 2 * alloc_words);
			// (word_owner is 0 in the padding, visited / ambig too: a diagonal that runs off the text matches nothing)
			HIP_TRY(launch_btextrec(idx->codes, idx->visited, idx->ambig, idx->word_owner, alloc_words, idx->trec, owner_blk, st));
			HIP_TRY(hipStreamSynchronize(st));
			idx->bx.trec = idx->trec;
			idx->bx.owner_blk = owner_blk;
		}
	}
	*out = idx;
	idx = nullptr;
done:
	if (idx)
		arks_index_free(idx);
	return rc;
}

int
arks_index_build(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int device,
    arks_build_stats* stats)
{
	return index_build_impl(out, k, h_bases, h_offsets, h_lens, n_ends, 0, 1, device, stats);
}

int
arks_index_build_ex(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int device,
    const arks_build_options* opt,
    arks_build_stats* stats)
{
	arks_build_options o;
	std::memset(&o, 0, sizeof o);
	if (opt) {
		// the caller's struct may be shorter (an older header) or longer (a newer one): the common prefix counts
		if (opt->struct_size < 2 * sizeof(uint32_t))
			return ARKS_ERR_BAD_ARG;
		std::memcpy(&o, opt, opt->struct_size < sizeof o ? opt->struct_size : sizeof o);
	}
	BuildChoice c;
	if (o.index_kind < ARKS_INDEX_AUTO || o.index_kind > ARKS_INDEX_SEEDS)
		return ARKS_ERR_BAD_ARG;
	c.kind = o.index_kind;
	if (o.heavy_over) {
		if (o.heavy_over < 2 || o.heavy_over > 8)
			return ARKS_ERR_BAD_ARG;
		c.heavy = (u32)o.heavy_over;
	}
	if (o.minimizer_len < 0 || o.minimizer_len > 32)
		return ARKS_ERR_BAD_ARG;
	c.mlen = o.minimizer_len;
	if (o.fallback_load_inv && o.fallback_load_inv != 2 && o.fallback_load_inv != 4)
		return ARKS_ERR_BAD_ARG;
	c.fb_load_inv = o.fallback_load_inv;
	const int n_shards = o.n_shards ? o.n_shards : 1, seed_ranks = o.seed_ranks ? o.seed_ranks : 1;
	if (n_shards > 1 && seed_ranks > 1)
		return ARKS_ERR_BAD_ARG; // one way of sharding at a time
	const int rc = index_build_impl(
	    out, k, h_bases, h_offsets, h_lens, n_ends, o.shard, n_shards, device, stats, o.seed_rank, seed_ranks, c);
	if (rc == ARKS_OK && seed_ranks > 1 && (*out)->kind != 2) { // k < 20: no seed table to shard
		arks_index_free(*out);
		*out = nullptr;
		return ARKS_ERR_K_UNSUPPORTED;
	}
	return rc;
}

int
arks_shard_of_ends(const uint32_t* h_lens, int64_t n_ends, int n_shards, int32_t* h_shard)
{
	if (n_ends < 0 || n_shards < 1 || (n_ends > 0 && (!h_lens || !h_shard)))
		return ARKS_ERR_BAD_ARG;
	std::vector<int32_t> shard_of;
	assign_shards(h_lens, n_ends, n_shards, shard_of);
	for (int64_t e = 0; e < n_ends; ++e)
		h_shard[e] = shard_of[(size_t)e];
	return ARKS_OK;
}

int
arks_index_build_shard(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int shard,
    int n_shards,
    int device)
{
	return index_build_impl(out, k, h_bases, h_offsets, h_lens, n_ends, shard, n_shards, device, nullptr);
}

int
arks_index_build_shard_stats(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int shard,
    int n_shards,
    int device,
    arks_build_stats* stats)
{
	return index_build_impl(out, k, h_bases, h_offsets, h_lens, n_ends, shard, n_shards, device, stats);
}

int
arks_index_build_seed_shard(
    arks_index** out,
    int k,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_ends,
    int rank,
    int n_ranks,
    int device,
    arks_build_stats* stats)
{
	if (n_ranks < 1 || rank < 0 || rank >= n_ranks)
		return ARKS_ERR_BAD_ARG;
	const int rc = index_build_impl(out, k, h_bases, h_offsets, h_lens, n_ends, 0, 1, device, stats, rank, n_ranks);
	if (rc == ARKS_OK && n_ranks > 1 && (*out)->kind != 2) { // k < 20: no seed table to shard
		arks_index_free(*out);
		*out = nullptr;
		return ARKS_ERR_K_UNSUPPORTED;
	}
	return rc;
}

int
arks_index_seed_ranks(const arks_index* idx)
{
	return idx ? idx->seed_ranks : 0;
}

int
arks_index_free(arks_index* idx)
{
	if (!idx)
		return ARKS_OK;
	DeviceGuard guard(idx->device);
	if (idx->table.slots)
		(void)hipFree(idx->table.slots);
	if (idx->codes)
		(void)hipFree(idx->codes);
	if (idx->visited)
		(void)hipFree(idx->visited);
	if (idx->ambig)
		(void)hipFree(idx->ambig);
	if (idx->word_owner)
		(void)hipFree(idx->word_owner);
	if (idx->mtab_va) {
		(void)hipMemUnmap(idx->mtab_va, idx->mtab_map_bytes);
		(void)hipMemRelease(idx->mtab_handle);
		(void)hipMemAddressFree(idx->mtab_va, idx->mtab_va_bytes);
	} else if (idx->mtab)
		(void)hipFree(idx->mtab);
	if (idx->trec)
		(void)hipFree(idx->trec);
	if (idx->mtab_gen)
		(void)hipFree(idx->mtab_gen);
	for (auto& q : idx->queues) {
		if (q.second.queue)
			(void)hipFree(q.second.queue);
		if (q.second.queue_count)
			(void)hipFree(q.second.queue_count);
	}
	delete idx;
	return ARKS_OK;
}

int
arks_index_k(const arks_index* idx)
{
	return idx ? idx->k : 0;
}

int64_t
arks_index_size(const arks_index* idx)
{
	return idx ? idx->n_keys : 0;
}

int64_t
arks_index_device_bytes(const arks_index* idx)
{
	if (!idx)
		return 0;
	int64_t b = (int64_t)(idx->table.cap * kSlotWords * sizeof(u64));
	{
		std::lock_guard<std::mutex> lk(idx->queue_m);
		for (const auto& q : idx->queues)
			b += 2 * q.second.cap * (int64_t)sizeof(u32) + (int64_t)kMapScratchBytes;
	}
	if (idx->kind >= 1)
		b += (int64_t)(idx->alloc_words * (sizeof(u64) + 3 * sizeof(u32))) + (int64_t)(idx->bx.mtab_cap * sizeof(u64)) +
		     (idx->trec ? (int64_t)(idx->alloc_words * 2 * sizeof(u64) + (idx->alloc_words / 32 + 1) * sizeof(u32)) : 0) +
		     (idx->mtab_gen ? (int64_t)(idx->bxg.mtab_cap * sizeof(u64)) : 0);
	return b;
}

int
arks_index_kind(const arks_index* idx)
{
	return idx ? idx->kind : -1;
}

int
arks_index_fallback_size(const arks_index* idx, int64_t out[2])
{
	if (!idx || !out)
		return ARKS_ERR_BAD_ARG;
	out[0] = idx->kind >= 1 ? idx->n_fallback : 0;
	out[1] = idx->kind >= 1 ? (int64_t)(idx->table.cap * kSlotWords * sizeof(u64)) : 0;
	return ARKS_OK;
}

int
arks_index_export(const arks_index* idx, unsigned char* h_keys, int32_t* h_vals)
{
	if (!idx || !h_keys || !h_vals)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	int rc = ARKS_OK;
	const int kb = arks_key_bytes(idx->k);
	const int kw = idx->kw;
	int64_t n = 0;
	std::vector<u64> host;
	std::vector<int32_t> hvals;
	std::vector<size_t> order;
	DevBuf d_keys, d_vals, d_cnt;
	auto emit = [&](const u64* words, int32_t val) {
		unsigned char* kout = h_keys + (size_t)n * (size_t)kb;
		for (int b = 0; b < kb; ++b)
			kout[b] = (unsigned char)(words[b >> 3] >> (56 - 8 * (b & 7)));
		h_vals[n++] = val;
	};
	try {
		if (idx->kind == 0) {
			host.resize(idx->table.cap * kSlotWords);
			HIP_TRY(hipMemcpy(host.data(), idx->table.slots, host.size() * sizeof(u64), hipMemcpyDeviceToHost));
			for (u64 s = 0; s < idx->table.cap && n < idx->n_keys; ++s) {
				const u64* slot = host.data() + s * kSlotWords;
				const u32 st = (u32)slot[3];
				if (st != kEmpty)
					emit(slot, (int32_t)(st - 1u));
			}
		} else {
			// one record per visited window, duplicates removed on the host
			const size_t nv = (size_t)idx->n_visited;
			host.resize(nv * (size_t)kw + 1);
			hvals.resize(nv + 1);
			HIP_TRY(d_keys.alloc(sizeof(u64) * (nv * (size_t)kw + 1)));
			HIP_TRY(d_vals.alloc(sizeof(int32_t) * (nv + 1)));
			HIP_TRY(d_cnt.alloc(sizeof(u64)));
			HIP_TRY(hipMemset(d_cnt.p, 0, sizeof(u64)));
			HIP_TRY(launch_bexport(
			    kw, idx->codes, idx->visited, idx->ambig, idx->word_owner, idx->text_words, idx->geom,
			    d_keys.as<u64>(), d_vals.as<int>(), d_cnt.as<u64>(), nullptr));
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemcpy(host.data(), d_keys.p, sizeof(u64) * nv * (size_t)kw, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(hvals.data(), d_vals.p, sizeof(int32_t) * nv, hipMemcpyDeviceToHost));
			order.resize(nv);
			for (size_t i = 0; i < nv; ++i)
				order[i] = i;
			std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
				for (int j = 0; j < kw; ++j)
					if (host[a * kw + j] != host[b * kw + j])
						return host[a * kw + j] < host[b * kw + j];
				return false;
			});
			for (size_t i = 0; i < nv && n < idx->n_keys; ++i) {
				const size_t a = order[i];
				if (i > 0) {
					const size_t b = order[i - 1];
					bool same = true;
					for (int j = 0; j < kw; ++j)
						same = same && host[a * kw + j] == host[b * kw + j];
					if (same)
						continue;
				}
				emit(host.data() + a * kw, hvals[a]);
			}
		}
	} catch (const std::bad_alloc&) {
		return ARKS_ERR_OOM;
	}
done:
	return rc;
}

/* ---------------------------------------------------------------------------------------------- */
/* mapping                                                                                        */
/* ---------------------------------------------------------------------------------------------- */

// the queue set of `stream` for this index, large enough for n_reads (created / grown on demand)
static int
ensure_queue(const arks_index* idx, void* stream, int64_t n_reads, arks_index::QueueSet* out)
{
	std::lock_guard<std::mutex> lk(idx->queue_m);
	arks_index::QueueSet* qs = nullptr;
	for (auto& q : idx->queues)
		if (q.first == stream)
			qs = &q.second;
	if (!qs) {
		idx->queues.emplace_back(stream, arks_index::QueueSet());
		qs = &idx->queues.back().second;
	}
	idx->last_stream = stream;
	if (!qs->queue_count) {
		void* p = nullptr;
		// slow-queue length, work counter, medium-queue length, its work counter; partial statistics; work counters
		hipError_t e = hipMalloc(&p, kMapScratchBytes);
		if (e != hipSuccess)
			return fail_hip(e, "hipMalloc(map scratch)");
		qs->queue_count = static_cast<u32*>(p);
	}
	if (qs->cap < n_reads) {
		if (qs->queue) {
			(void)hipStreamSynchronize(static_cast<hipStream_t>(stream)); // the stream's earlier call may still use it
			(void)hipFree(qs->queue);
			qs->queue = nullptr;
			qs->cap = 0;
		}
		void* p = nullptr;
		const int64_t cap = n_reads + n_reads / 4 + 1024;
		hipError_t e = hipMalloc(&p, 2 * sizeof(u32) * (size_t)cap); // slow queue + medium queue
		if (e != hipSuccess)
			return fail_hip(e, "hipMalloc(redo queues)");
		qs->queue = static_cast<u32*>(p);
		qs->cap = cap;
	}
	++qs->gen;
	*out = *qs;
	return ARKS_OK;
}

int
arks_map_reads_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream)
{
	if (!idx || n_reads < 0 || n_reads > 0xFFFFFFFFll)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_out_conreci)
		return ARKS_ERR_BAD_ARG;
	if (idx->seed_ranks > 1)
		return ARKS_ERR_BAD_ARG; // this rank holds a shard of the seed table: arks_map_reads_seeded_device
	DeviceGuard guard(idx->device);
	arks_index::QueueSet qs;
	int rc = ensure_queue(idx, stream, n_reads, &qs);
	if (rc != ARKS_OK)
		return rc;
	HIP_TRY(poison_lds(idx->n_cu, static_cast<hipStream_t>(stream)));
	HIP_TRY(launch_map_reads(
	    idx->kw, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads,
	    j_index, idx->geom, idx->table, idx->bx, d_out_conreci, reinterpret_cast<u64*>(d_stats),
	    qs.queue, qs.queue_count, idx->n_cu, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_map_pairs_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_pair_ok,
    const uint8_t* d_read_class,
    uint8_t* d_eval_out,
    int64_t n_reads,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream)
{
	if (!idx || n_reads < 0 || n_reads > 0xFFFFFFFFll || (n_reads & 1))
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_read_class || !d_eval_out || !d_out_conreci)
		return ARKS_ERR_BAD_ARG;
	if (idx->seed_ranks > 1)
		return ARKS_ERR_BAD_ARG; // a shard of the seed table: arks_exchange_submit_pairs
	if (idx->kind != 2) {
		// only the seed index's tile kernel works the gate out itself: the other layouts take the two launches
		int rc2 = arks_pair_gate_device(d_pair_ok, d_read_class, n_reads / 2, d_eval_out, idx->device, stream);
		if (rc2 != ARKS_OK)
			return rc2;
		return arks_map_reads_device(idx, d_codes, d_nmask, d_word_off, d_lens, d_eval_out, n_reads, j_index, d_out_conreci,
		                             d_stats, stream);
	}
	DeviceGuard guard(idx->device);
	arks_index::QueueSet qs;
	int rc = ensure_queue(idx, stream, n_reads, &qs);
	if (rc != ARKS_OK)
		return rc;
	HIP_TRY(poison_lds(idx->n_cu, static_cast<hipStream_t>(stream)));
	HIP_TRY(launch_map_reads(
	    idx->kw, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, nullptr, (long)n_reads,
	    j_index, idx->geom, idx->table, idx->bx, d_out_conreci, reinterpret_cast<u64*>(d_stats),
	    qs.queue, qs.queue_count, idx->n_cu, static_cast<hipStream_t>(stream), false, d_read_class, d_pair_ok));
done:
	return rc;
}

int
arks_map_votes_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    uint64_t* d_out_votes,
    void* stream)
{
	if (!idx || n_reads < 0 || n_reads > 0xFFFFFFFFll)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_out_votes)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	arks_index::QueueSet qs;
	int rc = ensure_queue(idx, stream, n_reads, &qs);
	if (rc != ARKS_OK)
		return rc;
	HIP_TRY(launch_map_reads(
	    idx->kw, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, 0.0,
	    idx->geom, idx->table, idx->bx, reinterpret_cast<int*>(d_out_votes), nullptr, qs.queue,
	    qs.queue_count, idx->n_cu, static_cast<hipStream_t>(stream), true));
done:
	return rc;
}

int
arks_seed_counts_device(
    const arks_index* idx, const uint32_t* d_lens, const uint8_t* d_eval, int64_t n_reads, int32_t* d_counts, void* stream)
{
	if (!idx || n_reads < 0 || idx->kind != 2)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_lens || !d_counts)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	int rc = ARKS_OK;
	HIP_TRY(launch_seed_counts(d_lens, d_eval, (long)n_reads, idx->k, idx->bx.w, d_counts, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_seeds_fill_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    const int64_t* d_seed_off,
    uint64_t* d_seed_mmer,
    int32_t* d_seed_owner,
    void* stream)
{
	if (!idx || n_reads < 0 || idx->kind != 2)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_seed_off || !d_seed_mmer || !d_seed_owner)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	int rc = ARKS_OK;
	HIP_TRY(launch_seeds_fill(
	    idx->bx.m, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, idx->k, idx->bx.w,
	    (u32)idx->seed_ranks, (const long*)d_seed_off, (u64*)d_seed_mmer, d_seed_owner, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_seeds_probe_device(const arks_index* idx, const uint64_t* d_mmer, int64_t n, uint64_t* d_answers, void* stream)
{
	if (!idx || n < 0 || idx->kind != 2)
		return ARKS_ERR_BAD_ARG;
	if (n == 0)
		return ARKS_OK;
	if (!d_mmer || !d_answers)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	int rc = ARKS_OK;
	HIP_TRY(launch_seeds_probe(idx->bx.m, idx->bx, (const u64*)d_mmer, (long)n, (u64*)d_answers, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_map_reads_seeded_device(
    const arks_index* idx,
    const uint64_t* d_codes,
    const uint32_t* d_nmask,
    const uint64_t* d_word_off,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    const int64_t* d_seed_off,
    const uint64_t* d_answers,
    double j_index,
    int32_t* d_out_conreci,
    arks_map_stats* d_stats,
    void* stream)
{
	if (!idx || n_reads < 0 || n_reads > 0xFFFFFFFFll || idx->kind != 2)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!d_codes || !d_nmask || !d_word_off || !d_lens || !d_out_conreci || !d_seed_off || !d_answers)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	arks_index::QueueSet qs;
	int rc = ensure_queue(idx, stream, n_reads, &qs);
	if (rc != ARKS_OK)
		return rc;
	HIP_TRY(launch_map_reads_seeded(
	    idx->kw, (const u64*)d_codes, d_nmask, (const u64*)d_word_off, d_lens, d_eval, (long)n_reads, j_index, idx->geom,
	    idx->bx, idx->bxg, (const long*)d_seed_off, (const u64*)d_answers, d_out_conreci, reinterpret_cast<u64*>(d_stats),
	    qs.queue, qs.queue_count, idx->n_cu, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_votes_max_device(uint64_t* d_acc, const uint64_t* d_in, int64_t n_reads, int device, void* stream)
{
	if (n_reads < 0 || (n_reads > 0 && (!d_acc || !d_in)))
		return ARKS_ERR_BAD_ARG;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	HIP_TRY(launch_max_votes((u64*)d_acc, (const u64*)d_in, (long)n_reads, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_votes_resolve_device(
    const uint64_t* d_votes,
    const uint32_t* d_lens,
    int64_t n_reads,
    int k,
    double j_index,
    int32_t* d_out_conreci,
    int device,
    void* stream)
{
	if (n_reads < 0 || (n_reads > 0 && (!d_votes || !d_lens || !d_out_conreci)))
		return ARKS_ERR_BAD_ARG;
	int rc = check_k(k);
	if (rc != ARKS_OK)
		return rc;
	rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	HIP_TRY(launch_resolve_votes(
	    (const u64*)d_votes, d_lens, (long)n_reads, k, j_index, d_out_conreci, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_votes_count_device(
    const uint64_t* d_votes,
    const uint32_t* d_lens,
    const uint8_t* d_eval,
    int64_t n_reads,
    int k,
    double j_index,
    arks_map_stats* d_stats,
    int device,
    void* stream)
{
	if (n_reads < 0 || !d_stats || (n_reads > 0 && (!d_votes || !d_lens)))
		return ARKS_ERR_BAD_ARG;
	static_assert(offsetof(arks_map_stats, reads_pass) == 5 * sizeof(uint64_t) &&
	                  offsetof(arks_map_stats, reads_fail) == 6 * sizeof(uint64_t),
	              "votes_count_kernel adds to words 5 and 6");
	int rc = check_k(k);
	if (rc != ARKS_OK)
		return rc;
	rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	HIP_TRY(launch_votes_count(
	    (const u64*)d_votes, d_lens, d_eval, (long)n_reads, k, j_index, reinterpret_cast<u64*>(d_stats),
	    static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_map_reads(
    const arks_index* idx,
    const char* h_bases,
    const uint64_t* h_offsets,
    const uint32_t* h_lens,
    int64_t n_reads,
    double j_index,
    int32_t* h_out_conreci,
    arks_map_stats* stats)
{
	if (!idx || n_reads < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_reads == 0)
		return ARKS_OK;
	if (!h_bases || !h_offsets || !h_lens || !h_out_conreci)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	int rc = ARKS_OK;
	hipStream_t st = nullptr;
	std::vector<uint64_t> word_off((size_t)n_reads + 1), offs((size_t)n_reads + 1);
	arks_word_offsets(h_lens, n_reads, word_off.data());
	const u64 total_words = word_off[(size_t)n_reads];
	uint64_t total_bases = 0;
	for (int64_t r = 0; r < n_reads; ++r) {
		offs[(size_t)r] = total_bases;
		total_bases += h_lens[r];
	}
	offs[(size_t)n_reads] = total_bases;
	std::vector<char> staged;
	try {
		staged.resize(total_bases + 64);
	} catch (const std::bad_alloc&) {
		return ARKS_ERR_OOM;
	}
	for (int64_t r = 0; r < n_reads; ++r)
		std::memcpy(staged.data() + offs[(size_t)r], h_bases + h_offsets[r], h_lens[r]);
	DevBuf d_ascii, d_offs, d_lens, d_woff, d_codes, d_nmask, d_out, d_st;
	arks_map_stats zero;
	std::memset(&zero, 0, sizeof zero);
	HIP_TRY(d_ascii.alloc(total_bases + 64));
	HIP_TRY(d_offs.alloc(sizeof(u64) * ((size_t)n_reads + 1)));
	HIP_TRY(d_lens.alloc(sizeof(u32) * (size_t)n_reads));
	HIP_TRY(d_woff.alloc(sizeof(u64) * ((size_t)n_reads + 1)));
	HIP_TRY(d_codes.alloc(sizeof(u64) * (total_words + ARKS_PAD_WORDS)));
	HIP_TRY(d_nmask.alloc(sizeof(u32) * (total_words + ARKS_PAD_WORDS)));
	HIP_TRY(d_out.alloc(sizeof(int32_t) * (size_t)n_reads));
	HIP_TRY(d_st.alloc(sizeof(arks_map_stats)));
	HIP_TRY(hipMemcpy(d_ascii.p, staged.data(), total_bases, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_offs.p, offs.data(), sizeof(u64) * ((size_t)n_reads + 1), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_lens.p, h_lens, sizeof(u32) * (size_t)n_reads, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_woff.p, word_off.data(), sizeof(u64) * ((size_t)n_reads + 1), hipMemcpyHostToDevice));
	HIP_TRY(hipMemsetAsync(d_codes.p, 0, sizeof(u64) * (total_words + ARKS_PAD_WORDS), st));
	HIP_TRY(hipMemsetAsync(d_nmask.p, 0, sizeof(u32) * (total_words + ARKS_PAD_WORDS), st));
	HIP_TRY(hipMemcpy(d_st.p, &zero, sizeof zero, hipMemcpyHostToDevice));
	HIP_TRY(launch_pack(
	    d_ascii.as<uint8_t>(), d_offs.as<u64>(), d_lens.as<u32>(), d_woff.as<u64>(), (long)n_reads,
	    total_words, d_codes.as<u64>(), d_nmask.as<u32>(), nullptr, nullptr, st));
	rc = arks_map_reads_device(
	    idx, d_codes.as<uint64_t>(), d_nmask.as<u32>(), d_woff.as<uint64_t>(), d_lens.as<u32>(), nullptr,
	    n_reads, j_index, d_out.as<int32_t>(), stats ? d_st.as<arks_map_stats>() : nullptr, st);
	if (rc != ARKS_OK)
		goto done;
	HIP_TRY(hipMemcpy(h_out_conreci, d_out.p, sizeof(int32_t) * (size_t)n_reads, hipMemcpyDeviceToHost));
	if (stats) {
		arks_map_stats got;
		HIP_TRY(hipMemcpy(&got, d_st.p, sizeof got, hipMemcpyDeviceToHost));
		stats->total_valid += got.total_valid;
		stats->bad += got.bad;
		stats->found += got.found;
		stats->recorded += got.recorded;
		stats->dups += got.dups;
		stats->reads_pass += got.reads_pass;
		stats->reads_fail += got.reads_fail;
		stats->windows += got.windows;
	}
done:
	return rc;
}

/* ---------------------------------------------------------------------------------------------- */
/* pairs and the IndexMap accumulator                                                             */
/* ---------------------------------------------------------------------------------------------- */

static void
imap_release(ImapView& v)
{
	if (v.keys)
		(void)hipFree(v.keys);
	if (v.counts)
		(void)hipFree(v.counts);
	if (v.first)
		(void)hipFree(v.first);
	if (v.n_entries)
		(void)hipFree(v.n_entries);
	v = ImapView{ nullptr, nullptr, nullptr, 0, nullptr, nullptr };
}

// an empty table of `slots` slots (n_entries and overflow share one allocation)
static int
imap_alloc(ImapView& v, u64 slots)
{
	int rc = ARKS_OK;
	void* p = nullptr;
	v = ImapView{ nullptr, nullptr, nullptr, slots, nullptr, nullptr };
	HIP_TRY(hipMalloc(&p, slots * sizeof(u64)));
	v.keys = static_cast<u64*>(p);
	HIP_TRY(hipMalloc(&p, slots * sizeof(u32)));
	v.counts = static_cast<u32*>(p);
	HIP_TRY(hipMalloc(&p, slots * sizeof(u64)));
	v.first = static_cast<u64*>(p);
	HIP_TRY(hipMalloc(&p, 2 * sizeof(u32)));
	v.n_entries = static_cast<u32*>(p);
	v.overflow = v.n_entries + 1;
	HIP_TRY(hipMemset(v.keys, 0, slots * sizeof(u64)));
	HIP_TRY(hipMemset(v.counts, 0, slots * sizeof(u32)));
	HIP_TRY(hipMemset(v.first, 0xFF, slots * sizeof(u64)));
	HIP_TRY(hipMemset(v.n_entries, 0, 2 * sizeof(u32)));
	HIP_TRY(hipDeviceSynchronize()); // the caller's streams need not be ordered behind the null stream
done:
	if (rc != ARKS_OK)
		imap_release(v);
	return rc;
}

// exact number of entries (waits for the device)
static int
imap_count(const arks_imap* m, u64* n)
{
	int rc = ARKS_OK;
	u32 h[2] = { 0, 0 };
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(h, m->v.n_entries, sizeof h, hipMemcpyDeviceToHost));
	if (h[1])
		return ARKS_ERR_FULL;
	*n = h[0];
done:
	return rc;
}

// makes room for `incoming` more pairs: load <= 1/2 afterwards whatever they hold
static int
imap_reserve(arks_imap* m, u64 incoming)
{
	if ((m->bound + incoming) * 2 <= m->v.cap)
		return ARKS_OK;
	u64 n = 0;
	int rc = imap_count(m, &n); // the bound counted every pair as a new entry: take the exact figure
	if (rc != ARKS_OK)
		return rc;
	m->bound = n;
	if ((n + 8 * incoming) * 2 <= m->v.cap) // room for at least eight more launches of this size
		return ARKS_OK;
	// Room for eight more launches where the device has it (an exact count is a device-wide wait: the fewer the better),
	// for four, two or just this one where it does not: a table takes 20 bytes per slot, and with launches of 10^7-10^8
	// pairs "eight launches" is 5-40 GB per accumulator -- eight local ranks on one device (bench.py --sharded-index,
	// arcs --index-sharded) ran out of memory over it in round 5.  No more than an eighth of what is free now.
	u64 ahead = 8;
	{
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
			free_b = 0, (void)hipGetLastError();
		auto slots_for = [&](u64 a) {
			u64 sl = m->v.cap;
			while ((n + a * incoming) * 2 > sl)
				sl *= 2;
			return sl;
		};
		while (ahead > 1 && slots_for(ahead) * 20ull > (u64)free_b / 8)
			ahead /= 2;
		if ((n + ahead * incoming) * 2 <= m->v.cap)
			return ARKS_OK;
	}
	u64 slots = m->v.cap;
	while ((n + ahead * incoming) * 2 > slots)
		slots *= 2;
	ImapView to;
	rc = imap_alloc(to, slots);
	if (rc != ARKS_OK) { // no room for the larger table: go on while the exact bound allows it
		return (n + incoming) * 2 <= m->v.cap ? ARKS_OK : rc;
	}
	HIP_TRY(launch_imap_rehash(m->v, to, nullptr));
	HIP_TRY(hipDeviceSynchronize());
	imap_release(m->v);
	m->v = to;
	return ARKS_OK;
done:
	imap_release(to);
	return rc;
}

int
arks_imap_create(arks_imap** out, int64_t capacity_entries, int device)
{
	if (!out || capacity_entries <= 0)
		return ARKS_ERR_BAD_ARG;
	*out = nullptr;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	arks_imap* m = new (std::nothrow) arks_imap;
	if (!m)
		return ARKS_ERR_OOM;
	m->device = device;
	// a starting size only (the table grows): at most 2^24 slots up front
	u64 slots = 1024;
	while (slots < (u64)capacity_entries * 2 && slots < ((u64)1 << 24))
		slots *= 2;
	rc = imap_alloc(m->v, slots);
	if (rc != ARKS_OK) {
		delete m;
		return rc;
	}
	*out = m;
	return ARKS_OK;
}

int
arks_imap_free(arks_imap* m)
{
	if (!m)
		return ARKS_OK;
	DeviceGuard guard(m->device);
	imap_release(m->v);
	delete m;
	return ARKS_OK;
}

int
arks_imap_set_pair_base(arks_imap* m, uint64_t first_pair)
{
	if (!m)
		return ARKS_ERR_BAD_ARG;
	m->seq_base = first_pair;
	return ARKS_OK;
}

int64_t
arks_imap_size(const arks_imap* m)
{
	if (!m)
		return 0;
	DeviceGuard guard(m->device);
	u64 n = 0;
	const int rc = imap_count(m, &n);
	return rc == ARKS_OK ? (int64_t)n : -(int64_t)rc;
}

// the entries sorted by (barcode id, conreci): compacted on the device, sorted on the host
static int
imap_download(const arks_imap* m, uint32_t* h_triples, uint64_t* h_first)
{
	int rc = ARKS_OK;
	u64 n = 0;
	rc = imap_count(m, &n);
	if (rc != ARKS_OK || n == 0)
		return rc;
	DevBuf dk, df, dc, cur;
	std::vector<u64> keys, first;
	std::vector<u32> counts, order;
	try {
		keys.resize(n), first.resize(n), counts.resize(n), order.resize(n);
	} catch (const std::bad_alloc&) {
		return ARKS_ERR_OOM;
	}
	HIP_TRY(dk.alloc(n * sizeof(u64)));
	HIP_TRY(df.alloc(n * sizeof(u64)));
	HIP_TRY(dc.alloc(n * sizeof(u32)));
	HIP_TRY(cur.alloc(sizeof(u32)));
	HIP_TRY(hipMemset(cur.p, 0, sizeof(u32)));
	HIP_TRY(launch_imap_compact(m->v, dk.as<u64>(), df.as<u64>(), dc.as<u32>(), cur.as<u32>(), nullptr));
	HIP_TRY(hipMemcpy(keys.data(), dk.p, n * sizeof(u64), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(first.data(), df.p, n * sizeof(u64), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(counts.data(), dc.p, n * sizeof(u32), hipMemcpyDeviceToHost));
	for (u64 i = 0; i < n; ++i)
		order[i] = (u32)i;
	std::sort(order.begin(), order.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });
	for (u64 i = 0; i < n; ++i) {
		const u32 s = order[i];
		if (h_triples) {
			h_triples[3 * i + 0] = (uint32_t)(keys[s] >> 32);
			h_triples[3 * i + 1] = (uint32_t)keys[s];
			h_triples[3 * i + 2] = counts[s];
		}
		if (h_first)
			h_first[i] = first[s];
	}
done:
	return rc;
}

int
arks_imap_export(const arks_imap* m, uint32_t* h_triples)
{
	if (!m || !h_triples)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(m->device);
	return imap_download(m, h_triples, nullptr);
}

int
arks_imap_export_ordered(const arks_imap* m, uint32_t* h_triples, uint64_t* h_first_pair)
{
	if (!m || !h_triples || !h_first_pair)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(m->device);
	return imap_download(m, h_triples, h_first_pair);
}

int
arks_pair_gate_device(
    const uint8_t* d_pair_ok,
    const uint8_t* d_read_class,
    int64_t n_pairs,
    uint8_t* d_eval,
    int device,
    void* stream)
{
	if (n_pairs < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_pairs == 0)
		return ARKS_OK;
	if (!d_read_class || !d_eval)
		return ARKS_ERR_BAD_ARG;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	HIP_TRY(launch_pair_gate(d_pair_ok, d_read_class, (long)n_pairs, d_eval, static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_gate_count_device(const uint8_t* d_pair_ok, const uint8_t* d_eval, int64_t n_pairs, uint64_t* d_counter, int device, void* stream)
{
	if (n_pairs < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_pairs == 0)
		return ARKS_OK;
	if (!d_eval || !d_counter)
		return ARKS_ERR_BAD_ARG;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	HIP_TRY(arks::launch_gate_count(d_pair_ok, d_eval, (long)n_pairs, reinterpret_cast<u64*>(d_counter), static_cast<hipStream_t>(stream)));
done:
	return rc;
}

int
arks_pairs_device(
    const int32_t* d_conreci,
    const uint8_t* d_pair_ok,
    const uint32_t* d_barcode_id,
    int64_t n_pairs,
    int32_t* d_out_pair,
    arks_imap* imap,
    uint64_t* d_stored,
    int device,
    void* stream)
{
	if (n_pairs < 0)
		return ARKS_ERR_BAD_ARG;
	if (n_pairs == 0)
		return ARKS_OK;
	if (!d_conreci || (imap && !d_barcode_id))
		return ARKS_ERR_BAD_ARG;
	int rc = require_device(device);
	if (rc != ARKS_OK)
		return rc;
	DeviceGuard guard(device);
	{
		ImapView none{ nullptr, nullptr, nullptr, 0, nullptr, nullptr };
		if (imap) {
			rc = imap_reserve(imap, (u64)n_pairs);
			if (rc != ARKS_OK)
				return rc;
			imap->bound += (u64)n_pairs;
		}
		HIP_TRY(launch_pairs(
		    d_conreci, d_pair_ok, d_barcode_id, (long)n_pairs, d_out_pair, imap ? imap->v : none,
		    imap ? imap->seq_base : 0, (u64*)d_stored, static_cast<hipStream_t>(stream)));
		if (imap)
			imap->seq_base += (u64)n_pairs; // the next batch continues the numbering unless the caller sets it
	}
done:
	return rc;
}

/* debugging aid (not part of the ABI): lengths of the slow / medium queues after the last map call */
int
arks_debug_queue_counts(const arks_index* idx, unsigned* out4)
{
	if (!idx || !out4)
		return ARKS_ERR_BAD_ARG;
	DeviceGuard guard(idx->device);
	(void)hipDeviceSynchronize();
	std::lock_guard<std::mutex> lk(idx->queue_m);
	for (const auto& q : idx->queues)
		if (q.first == idx->last_stream && q.second.queue_count)
			return hipMemcpy(out4, q.second.queue_count, 4 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? ARKS_OK : ARKS_ERR_HIP;
	out4[0] = out4[1] = out4[2] = out4[3] = 0;
	return ARKS_OK;
}

int
arks_debug_set_medium_blocks(int n)
{
	if (n < 0)
		return ARKS_ERR_BAD_ARG;
	arks::set_medium_blocks_cap((unsigned)n);
	return ARKS_OK;
}

#ifdef ARKS_MEDIUM_DIAG
int
arks_debug_medium_diag(unsigned long long* out16)
{
	(void)hipDeviceSynchronize();
	arks::read_medium_diag(out16);
	return ARKS_OK;
}
#endif

#ifdef ARKS_PROFILE_SECTIONS
int
arks_debug_section_cycles(unsigned long long* out16)
{
	(void)hipDeviceSynchronize();
	arks::read_section_cycles(out16);
	return ARKS_OK;
}
#endif

} /* extern "C" */

#include "arks_exchange.hpp"
