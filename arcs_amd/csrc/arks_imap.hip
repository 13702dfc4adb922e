// arks_imap.hip -- the pair rule of chromiumRead (Arcs/Arcs.cpp:1280-1292) and the device accumulator
// behind `IndexMap imap` (Arcs/Arcs.h:108-113): an open-addressed table (barcode id, conreci) ->
// (count, sequence number of the first stored pair).  The table grows: the host side
// (arks_pairs_device) keeps an upper bound of the entries -- a pair adds at most one -- and rebuilds
// the table into a larger one before a launch could push the load beyond 1/2, so an insert never fails.
#include "arks_kernels.hpp"

namespace arks {

__device__ __forceinline__ u64
mix64(u64 x)
{
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return x;
}

// slot of `key`, claiming an empty one when the key is new (keys are never 0: conreci >= 1); ~0 when the
// table is full, which the growth rule of the host side excludes
__device__ inline u64
imap_slot(u64* keys, u64 cap, u64 key, u32* n_entries)
{
	u64 s = mulhi64(mix64(key), cap);
	for (u64 probes = 0; probes < cap; ++probes) {
		u64 cur = __hip_atomic_load(keys + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			u64 expect = 0;
			if (__hip_atomic_compare_exchange_strong(
			        keys + s, &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
				cur = key;
				atomicAdd(n_entries, 1u);
			} else
				cur = expect;
		}
		if (cur == key)
			return s;
		s = (s + 1 == cap) ? 0 : s + 1;
	}
	return ~0ull;
}

// One thread per pair.  Runs of equal (barcode, conreci) in adjacent lanes -- the normal case,
// linked-read files are grouped by barcode -- are folded with a ballot before touching the table.
__global__ void
pairs_kernel(
    const int* __restrict__ conreci,
    const uint8_t* __restrict__ pair_ok,
    const u32* __restrict__ barcode_id,
    long n_pairs,
    int* __restrict__ out_pair,
    ImapView im,
    u64 seq_base, // sequence number of pair 0 of this batch (order of the pairs in the input)
    u64* __restrict__ stored)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	int agreed = 0;
	bool ok = false;
	if (p < n_pairs) {
		const int c1 = conreci[2 * p], c2 = conreci[2 * p + 1];
		agreed = (c1 != 0 && c1 == c2) ? c1 : 0; // Arcs.cpp:1280
		if (out_pair)
			out_pair[p] = agreed;
		ok = agreed != 0 && (pair_ok ? pair_ok[p] != 0 : true);
	}
	const u64 okmask = __ballot(ok);
	if (stored && lane == 0 && okmask)
		atomicAdd(stored, (u64)__popcll(okmask));
	if (im.keys == nullptr)
		return;
	const u64 key = ok ? (((u64)barcode_id[p] << 32) | (u32)agreed) : 0ull;
	// One table update per DISTINCT key of the wave, not per pair: linked reads come sorted by barcode, so the 64
	// pairs of a wave hold a few barcodes and a few contig ends -- a handful of keys -- but in any order and
	// with unstored pairs in between (runs of equal neighbours are short).  The lowest lane of every key leads:
	// it adds the number of lanes that hold the key, and its pair is the key's first of the wave.
	bool leader = false;
	u32 group = 0;
	u64 remaining = __ballot(key != 0);
	while (remaining) { // (wave-uniform: at most one round per distinct key, no memory traffic inside)
		const int first_lane = __ffsll((long long)remaining) - 1;
		const u32 klo = (u32)__builtin_amdgcn_readlane((int)(u32)key, first_lane);
		const u32 khi = (u32)__builtin_amdgcn_readlane((int)(u32)(key >> 32), first_lane);
		const u64 same = __ballot(key == (((u64)khi << 32) | klo));
		if (lane == first_lane) {
			leader = true;
			group = (u32)__popcll(same);
		}
		remaining &= ~same;
	}
	if (leader) {
		const u64 s = imap_slot(im.keys, im.cap, key, im.n_entries);
		if (s == ~0ull) {
			atomicOr(im.overflow, 1u);
		} else {
			atomicAdd(im.counts + s, group);
			atomicMin(reinterpret_cast<unsigned long long*>(im.first + s), (unsigned long long)(seq_base + (u64)p));
		}
	}
}

// every entry of `from` into the (empty, larger) table `to`
__global__ void
imap_rehash_kernel(ImapView from, ImapView to)
{
	const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= from.cap)
		return;
	const u64 key = from.keys[s];
	if (key == 0)
		return;
	const u64 d = imap_slot(to.keys, to.cap, key, to.n_entries);
	if (d == ~0ull) {
		atomicOr(to.overflow, 1u);
		return;
	}
	to.counts[d] = from.counts[s]; // keys are distinct in `from`: no two threads share d
	to.first[d] = from.first[s];
}

// the entries, in any order, as (key, first << 0, count) records: out_keys[i], out_first[i], out_counts[i]
__global__ void
imap_compact_kernel(ImapView im, u64* out_keys, u64* out_first, u32* out_counts, u32* cursor)
{
	const u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const bool has = s < im.cap && im.keys[s] != 0;
	const u64 m = __ballot(has);
	if (m == 0)
		return;
	const int lane = threadIdx.x & 63;
	u32 base = 0;
	if (lane == 0)
		base = atomicAdd(cursor, (u32)__popcll(m));
	base = (u32)__builtin_amdgcn_readfirstlane((int)base);
	if (has) {
		const u32 i = base + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
		out_keys[i] = im.keys[s];
		out_first[i] = im.first[s];
		out_counts[i] = im.counts[s];
	}
}

#define ARKS_LAUNCH_CHECK()                                                                        \
	do {                                                                                           \
		hipError_t e_ = hipGetLastError();                                                         \
		if (e_ != hipSuccess)                                                                      \
			return e_;                                                                             \
	} while (0)

static inline unsigned
blocks_for(u64 n, unsigned bs)
{
	u64 b = (n + bs - 1) / bs;
	return (unsigned)(b ? b : 1);
}

hipError_t
launch_pairs(
    const int* conreci, const uint8_t* pair_ok, const u32* barcode_id, long n_pairs, int* out_pair,
    const ImapView& im, u64 seq_base, u64* stored, hipStream_t st)
{
	if (n_pairs <= 0)
		return hipSuccess;
	pairs_kernel<<<blocks_for((u64)n_pairs, 256), 256, 0, st>>>(
	    conreci, pair_ok, barcode_id, n_pairs, out_pair, im, seq_base, stored);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_imap_rehash(const ImapView& from, const ImapView& to, hipStream_t st)
{
	imap_rehash_kernel<<<blocks_for(from.cap, 256), 256, 0, st>>>(from, to);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_imap_compact(const ImapView& im, u64* out_keys, u64* out_first, u32* out_counts, u32* cursor, hipStream_t st)
{
	imap_compact_kernel<<<blocks_for(im.cap, 256), 256, 0, st>>>(im, out_keys, out_first, out_counts, cursor);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

// gated pairs one of whose mates checkReadSequence rejects (Arcs.cpp:1273-1276: skipped_invalidreadpair):
// pair_ok[p] set (NULL = every pair) and eval[2p] clear, for a front end that packs -- and classifies -- the
// reads on the device and so cannot count them itself
__global__ void
gate_count_kernel(const uint8_t* __restrict__ pair_ok, const uint8_t* __restrict__ eval, long n_pairs, u64* __restrict__ counter)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const bool hit = p < n_pairs && (pair_ok ? pair_ok[p] != 0 : true) && eval[2 * p] == 0;
	const u64 m = __ballot(hit);
	if ((threadIdx.x & 63) == 0 && m)
		atomicAdd(counter, (u64)__popcll(m));
}

hipError_t
launch_gate_count(const uint8_t* pair_ok, const uint8_t* eval, long n_pairs, u64* counter, hipStream_t st)
{
	gate_count_kernel<<<blocks_for((u64)n_pairs, 256), 256, 0, st>>>(pair_ok, eval, n_pairs, counter);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

} // namespace arks
