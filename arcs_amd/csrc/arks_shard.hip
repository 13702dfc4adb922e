// arks_shard.hip -- the sharded seed table's own kernels (arks_exchange, BASELINE configs[3]): a batch's seeds listed and
// bucketed by owner on the device.  (The owner-side probe and the REMOTE instantiation of the seed tile kernel are
// arks_map.hip's; this file is kept apart so that work on the exchange does not touch the sources of the hot kernel --
// bench.py's kernel_build_id is a digest of those.)
#include "arks_device.hpp"
#include "arks_kernels.hpp"
#include <cstddef>
#include <cstdint>

namespace arks {

// lanes of one wave communicate through LDS: order the compiler's view of it
#define ARKS_WAVE_SYNC()                                                                           \
	do {                                                                                           \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
		__builtin_amdgcn_wave_barrier();                                                           \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
	} while (0)
#define ARKS_LAUNCH_CHECK()                                                                        \
	do {                                                                                           \
		hipError_t e_ = hipGetLastError();                                                         \
		if (e_ != hipSuccess)                                                                      \
			return e_;                                                                             \
	} while (0)

// lanes below this one whose bit is set in m
__device__ __forceinline__ u32
mask_below(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// ---- the sharded seed table, product path (arks_exchange): seeds listed AND bucketed by owner on the device ----
// Three launches over blocks of kBucketReads reads, no library calls:
//   count  per block: seeds of its reads (read-major numbering) and, per owner, the seeds that are sent
//   scan   exclusive prefix over the blocks of every one of those columns (one workgroup per column)
//   fill   d_seed_off[r] (read-major), and for every seed its slot in the send buffer -- the seeds of owner o
//          lie together, blocks in order -- or ~0 for a seed that holds an invalid base (nobody is asked; the map
//          kernel never looks at its answer); the canonical m-mer goes to send[slot]
// A seed's answer comes back at the same slot, so the map kernel reads ans[2 * slot[s]] and no pass puts the
// answers "back into seed order".
constexpr int kBucketReads = 256; // reads per block = threads per block
constexpr int kMaxOwners = 64;

template <int MM>
__device__ __forceinline__ int
bucket_read_seeds(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, const u64* __restrict__ word_off,
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long r, long n_reads, int k, int w, u32 n_owners,
    u64 cm[], u32 own[], int max_inline)
{
	// seeds of read r; the first max_inline of them are returned in cm / own (own = ~0: not sent)
	typedef typename Mmer<MM>::type mm_t;
	if (r >= n_reads)
		return 0;
	const int nwin = (eval && !eval[r]) ? 0 : (int)lens[r] - k + 1;
	const int G = nwin > 0 ? (nwin + w - 1) / w : 0;
	if (G == 0)
		return 0;
	const u64 wb = word_off[r];
	for (int gi = 0; gi < G && gi < max_inline; ++gi) {
		int q = (gi + 1) * w - 1;
		q = q < nwin - 1 ? q : nwin - 1;
		const u64 pos = wb * 32ull + (u64)q;
		const u32* nm = nmask + (pos >> 5);
		const u64 two = ((u64)nm[0] << 32) | (u64)nm[1];
		cm[gi] = ~0ull;
		own[gi] = ~0u;
		if (((two << (pos & 31)) >> (64 - MM)) == 0) {
			const mm_t mf = mmer_fw<MM>(codes, pos), mr = mmer_rc<MM>(mf);
			cm[gi] = (u64)(mf < mr ? mf : mr);
			own[gi] = seed_owner<MM>((mm_t)cm[gi], n_owners);
		}
	}
	return G;
}

// one seed of a read beyond the inline ones (long reads): recomputed where it is needed
template <int MM>
__device__ __forceinline__ void
bucket_one_seed(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wb, int nwin, int w, int gi, u32 n_owners, u64& cm,
    u32& own)
{
	typedef typename Mmer<MM>::type mm_t;
	int q = (gi + 1) * w - 1;
	q = q < nwin - 1 ? q : nwin - 1;
	const u64 pos = wb * 32ull + (u64)q;
	const u32* nm = nmask + (pos >> 5);
	const u64 two = ((u64)nm[0] << 32) | (u64)nm[1];
	cm = ~0ull;
	own = ~0u;
	if (((two << (pos & 31)) >> (64 - MM)) == 0) {
		const mm_t mf = mmer_fw<MM>(codes, pos), mr = mmer_rc<MM>(mf);
		cm = (u64)(mf < mr ? mf : mr);
		own = seed_owner<MM>((mm_t)cm, n_owners);
	}
}

constexpr int kBucketInline = 4; // seeds of a read kept in registers (a 10x pair has 2 + 3)

// columns: [0] seeds of the block, [1 + o] seeds of the block that owner o is asked; cols[c * n_blocks + block]
template <int MM>
__global__ void __launch_bounds__(kBucketReads)
seed_bucket_count_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, const u64* __restrict__ word_off,
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k, int w, u32 n_owners,
    long n_blocks, u32* __restrict__ cols)
{
	__shared__ u32 cnt[kMaxOwners + 1];
	if (threadIdx.x <= n_owners)
		cnt[threadIdx.x] = 0;
	__syncthreads();
	const long r = (long)blockIdx.x * kBucketReads + threadIdx.x;
	u64 cm[kBucketInline];
	u32 own[kBucketInline];
	const int G = bucket_read_seeds<MM>(codes, nmask, word_off, lens, eval, r, n_reads, k, w, n_owners, cm, own, kBucketInline);
	if (G) {
		atomicAdd(&cnt[0], (u32)G);
		for (int gi = 0; gi < G; ++gi) {
			u32 o;
			if (gi < kBucketInline)
				o = own[gi];
			else {
				u64 c;
				bucket_one_seed<MM>(codes, nmask, word_off[r], (int)lens[r] - k + 1, w, gi, n_owners, c, o);
			}
			if (o != ~0u)
				atomicAdd(&cnt[1 + o], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x <= n_owners)
		cols[(long)threadIdx.x * n_blocks + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive prefix of every column over the blocks, in place (u32: a launch holds < 2^32 seeds); totals[c] = its sum.
// One workgroup of 1024 threads per column: thread t sums its stretch, the stretch sums are scanned through LDS,
// the stretch is rewritten.
__global__ void __launch_bounds__(1024)
seed_bucket_scan_kernel(long n_blocks, u32* __restrict__ cols, u64* __restrict__ totals)
{
	__shared__ u64 part[1024];
	u32* col = cols + (long)blockIdx.x * n_blocks;
	const long per = (n_blocks + 1023) / 1024;
	const long lo = (long)threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
	u64 sum = 0;
	for (long i = lo; i < hi; ++i)
		sum += col[i];
	part[threadIdx.x] = sum;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
		const u64 v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	u64 run = part[threadIdx.x] - sum;
	for (long i = lo; i < hi; ++i) {
		const u32 v = col[i];
		col[i] = (u32)run;
		run += v;
	}
	if (threadIdx.x == 1023)
		totals[blockIdx.x] = part[1023];
}

template <int MM>
__global__ void __launch_bounds__(kBucketReads)
seed_bucket_fill_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, const u64* __restrict__ word_off,
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k, int w, u32 n_owners,
    long n_blocks, const u32* __restrict__ cols, const u64* __restrict__ totals, long* __restrict__ seed_off,
    u32* __restrict__ slot, u64* __restrict__ send)
{
	// (slots within a block are handed out by an LDS atomic per seed, in no particular order.  Handing them out in
	// order -- wave, seed number, lane: ballot + mbcnt -- so that a tile's answers lie in fewer lines was tried:
	// the fill took 1.31 instead of 1.08 ms per 25 M pairs and the map kernel behind it the same 3.5 ms,
	// profiles/r03p_sharded1_kernel_stats.csv against r03o's)
	__shared__ u32 cnt[kMaxOwners + 1];
	__shared__ u32 base[kMaxOwners + 1]; // [0]: first seed of the block (read-major); [1 + o]: first send slot of the block's seeds for owner o
	__shared__ u32 wsum[kBucketReads / 64];
	if (threadIdx.x <= n_owners) {
		cnt[threadIdx.x] = 0;
		u64 b = cols[(long)threadIdx.x * n_blocks + blockIdx.x];
		if (threadIdx.x >= 1)
			for (u32 o = 0; o + 1 < threadIdx.x; ++o) // owners in front of this one in the send buffer
				b += totals[1 + o];
		base[threadIdx.x] = (u32)b;
	}
	const long r = (long)blockIdx.x * kBucketReads + threadIdx.x;
	u64 cm[kBucketInline];
	u32 own[kBucketInline];
	const int G = bucket_read_seeds<MM>(codes, nmask, word_off, lens, eval, r, n_reads, k, w, n_owners, cm, own, kBucketInline);
	// read-major number of the read's first seed: exclusive scan of G over the block's threads
	int incl = G;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const int o = __shfl_up(incl, d);
		incl += (int)(threadIdx.x & 63) >= d ? o : 0;
	}
	if ((threadIdx.x & 63) == 63)
		wsum[threadIdx.x >> 6] = (u32)incl;
	__syncthreads();
	u32 before = 0;
	for (unsigned wv = 0; wv < (threadIdx.x >> 6); ++wv)
		before += wsum[wv];
	const long first = (long)base[0] + (long)before + (long)(incl - G);
	if (r < n_reads)
		seed_off[r] = first;
	if (r == n_reads - 1)
		seed_off[n_reads] = first + G;
	for (int gi = 0; gi < G; ++gi) {
		u64 c;
		u32 o;
		if (gi < kBucketInline)
			c = cm[gi], o = own[gi];
		else
			bucket_one_seed<MM>(codes, nmask, word_off[r], (int)lens[r] - k + 1, w, gi, n_owners, c, o);
		u32 sl = ~0u;
		if (o != ~0u) {
			sl = base[1 + o] + atomicAdd(&cnt[1 + o], 1u);
			send[sl] = c;
		}
		slot[first + gi] = sl;
	}
}

// seeds of a batch listed and bucketed by owner (see seed_bucket_count_kernel): cols = (1 + n_owners) * n_blocks u32 of
// scratch, totals = 1 + n_owners u64 ([0] seeds of the batch, [1 + o] seeds owner o is asked)
long
seed_bucket_blocks(long n_reads)
{
	return (n_reads + kBucketReads - 1) / kBucketReads;
}

hipError_t
launch_seed_buckets(
    int mm, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval, long n_reads,
    int k, int w, u32 n_owners, u32* cols, u64* totals, long* seed_off, u32* slot, u64* send, int phase, hipStream_t st)
{
	if (n_owners < 1 || n_owners > (u32)kMaxOwners)
		return hipErrorInvalidValue;
	const long nb = seed_bucket_blocks(n_reads);
	if (phase == 0) { // count + scan: totals are valid when the stream gets here
		if (n_reads <= 0)
			return hipMemsetAsync(totals, 0, sizeof(u64) * (1 + n_owners), st);
		if (mm == kMShort)
			seed_bucket_count_kernel<kMShort><<<(unsigned)nb, kBucketReads, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, nb, cols);
		else
			seed_bucket_count_kernel<kMLong><<<(unsigned)nb, kBucketReads, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, nb, cols);
		seed_bucket_scan_kernel<<<1 + n_owners, 1024, 0, st>>>(nb, cols, totals);
	} else { // fill
		if (n_reads <= 0)
			return hipMemsetAsync(seed_off, 0, sizeof(long), st);
		if (mm == kMShort)
			seed_bucket_fill_kernel<kMShort><<<(unsigned)nb, kBucketReads, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, nb, cols, totals, seed_off, slot, send);
		else
			seed_bucket_fill_kernel<kMLong><<<(unsigned)nb, kBucketReads, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, nb, cols, totals, seed_off, slot, send);
	}
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

} // namespace arks
