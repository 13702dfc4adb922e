// arks_shard.hip -- the sharded seed table's own kernels (arks_exchange, BASELINE configs[3]): a batch's seeds listed and
// bucketed by owner on the device.  (The owner-side probe and the REMOTE instantiation of the seed tile kernel are
// arks_map.hip's; this file is kept apart so that work on the exchange does not touch the sources of the hot kernel --
// bench.py's kernel_build_id is a digest of those.)
#include "arks_device.hpp"
#include "arks_kernels.hpp"
#include "arks_shard_stats.hpp"
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
// ONE launch (round 4; rounds 2-3 counted, scanned and filled in three, with a host wait between the second and the
// third): the send buffer is W regions of `cap` seeds, one per owner; a block of kBkReads reads counts its seeds per
// owner in LDS, reserves its stretch of every region with one global atomic per owner (ctl->fill[o]) and a stretch
// of the read-major seed numbering (ctl->seeds), and writes:
//   send[o * cap + ...]     the canonical m-mers asked of owner o (the seeds of an owner lie together, no sort)
//   slot[seed]              where seed `seed` went (its answer comes back at the same place), ~0 for a seed that
//                           holds an invalid base (nobody is asked; the map kernel never looks at its answer)
//   chunk_off[c]            number of the first seed of the map kernel's chunk c (sChunk = 56 reads): a block is a
//                           whole number of chunks, and a chunk's seeds are numbered read by read
// The counts reach the host with the batch's other results, one step behind the device (arks_exchange_complete);
// nothing waits for them.  A block that does not fit marks ctl->overflow and writes nothing: the counters go on
// counting, so the host knows what the batch needs and runs it again with larger regions (first batch of a shape).
// Round 5: the counters only count up and the launch is told where they stood (no memset per batch); the pair gate
// can be computed here instead of read (SeedBucketGate); block 0 zeroes the map kernels' scratch on the side; the N
// masks of reads the gate knows to hold ACGT only are not fetched.
constexpr int kMaxOwners = 64;
constexpr int kBkChunk = 56;                  // = sChunk of map_reads_s_kernel (arks_map.hip; checked at its launch)
constexpr int kBkWaves = 8;
constexpr int kBkChunksPerWave = 2;
constexpr int kBkChunks = kBkWaves * kBkChunksPerWave; // 16 chunks = 896 reads per block: ~56 k blocks per 50 M reads, i.e.
constexpr int kBkReads = kBkChunks * kBkChunk;         // ~0.7 ms of same-address atomics per counter (in parallel); 512
                                                       // threads at <= 64 VGPRs: four blocks per CU, so that the three
                                                       // dependent round trips of a block (read, reserve, write) overlap
                                                       // other blocks' (1024 threads at 83 VGPRs, one block per CU: 6.9 ms
                                                       // per 25 M pairs instead of ~1, profiles/r04e_sharded1_kernel_stats.csv)

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
    u32& own, bool may_n = true)
{
	typedef typename Mmer<MM>::type mm_t;
	int q = (gi + 1) * w - 1;
	q = q < nwin - 1 ? q : nwin - 1;
	const u64 pos = wb * 32ull + (u64)q;
	// (may_n = false: the pair gate knows the read to hold ACGT only -- eval == ARKS_EVAL_ACGT_ONLY --, its N masks, a
	// third of the read stream and zero for > 98 % of the reads, are not fetched; the map kernels do the same)
	u64 two = 0;
	if (may_n) {
		const u32* nm = nmask + (pos >> 5);
		two = ((u64)nm[0] << 32) | (u64)nm[1];
	}
	cm = ~0ull;
	own = ~0u;
	if (((two << (pos & 31)) >> (64 - MM)) == 0) {
		const mm_t mf = mmer_fw<MM>(codes, pos), mr = mmer_rc<MM>(mf);
		cm = (u64)(mf < mr ? mf : mr);
		own = seed_owner<MM>((mm_t)cm, n_owners);
	}
}

constexpr int kBucketInline = 4; // seeds of a read kept in registers (a 10x pair has 2 + 3)

// inclusive prefix sum over the lanes of a wave by DPP (see arks_map.hip)
__device__ __forceinline__ int
bk_wave_incl_scan(int v)
{
#define ARKS_SCAN_ADD(ctrl, rows) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xF, true)
	ARKS_SCAN_ADD(0x111, 0xF);
	ARKS_SCAN_ADD(0x112, 0xF);
	ARKS_SCAN_ADD(0x114, 0xF);
	ARKS_SCAN_ADD(0x118, 0xF);
	ARKS_SCAN_ADD(0x142, 0xA);
	ARKS_SCAN_ADD(0x143, 0xC);
#undef ARKS_SCAN_ADD
	return v;
}

// seed gi of read r for the bucket kernel: canonical m-mer | owner << 56, ~0 = it holds an invalid base (not sent)
template <int MM>
__device__ __forceinline__ u64
bucket_seed(const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wb, int nwin, int w, int gi, u32 n_owners, bool may_n)
{
	u64 c;
	u32 o;
	bucket_one_seed<MM>(codes, nmask, wb, nwin, w, gi, n_owners, c, o, may_n);
	return o == ~0u ? ~0ull : (c | ((u64)o << 56));
}

template <int MM>
__global__ void __launch_bounds__(kBkWaves * 64) __attribute__((amdgpu_waves_per_eu(8)))
seed_bucket_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, const u64* __restrict__ word_off,
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k, int w, u32 n_owners, u64 cap,
    u64 slot_cap, SeedBucketCtl* __restrict__ ctl, SeedBucketBase cb, SeedBucketGate gate, u32* __restrict__ zero_words,
    int n_zero, u32* __restrict__ chunk_off, u32* __restrict__ slot, u64* __restrict__ send)
{
	static_assert(2 * MM <= 56, "the owner rides in the top byte of the m-mer");
	__shared__ u32 cnt[kMaxOwners];
	__shared__ u32 base[kMaxOwners];
	__shared__ u32 chunk_cnt[kBkChunks];
	__shared__ u32 chunk_base[kBkChunks];
	__shared__ u32 bad;
	const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
	if (threadIdx.x < kMaxOwners)
		cnt[threadIdx.x] = 0;
	if (threadIdx.x == 0)
		bad = 0;
	if (blockIdx.x == 0 && zero_words)
		for (int x = (int)threadIdx.x; x < n_zero; x += kBkWaves * 64)
			zero_words[x] = 0u;
	__syncthreads();
	constexpr int kInline = 3; // seeds of a read kept in registers (a 10x pair has 2 + 3); others are made again
	u64 sd[kBkChunksPerWave][kInline];
	u64 wb[kBkChunksPerWave];
	int G[kBkChunksPerWave], pre[kBkChunksPerWave], nwin[kBkChunksPerWave];
	bool mn[kBkChunksPerWave];
#pragma unroll
	for (int it = 0; it < kBkChunksPerWave; ++it) {
		const int ci = wave * kBkChunksPerWave + it;
		const long r = ((long)blockIdx.x * kBkChunks + ci) * kBkChunk + lane;
		nwin[it] = 0, wb[it] = 0, mn[it] = true;
		if (lane < kBkChunk && r < n_reads) {
			uint8_t ev = 1;
			if (gate.read_class) {
				// arks_pair_gate_device's rule: both mates pass checkReadSequence (class bit 0) and the pair's barcode
				// test; bit 1 of the class = the read holds ACGT only
				const uint8_t c0 = gate.read_class[r], c1 = gate.read_class[r ^ 1];
				const bool e = (gate.pair_ok ? gate.pair_ok[r >> 1] != 0 : true) && (c0 & 1) && (c1 & 1);
				ev = e ? (uint8_t)(1 | (c0 & 2)) : (uint8_t)0;
				gate.eval_out[r] = ev;
			} else if (eval)
				ev = eval[r];
			nwin[it] = !ev ? 0 : (int)lens[r] - k + 1;
			mn[it] = ev != 3; // (exactly ARKS_EVAL_ACGT_ONLY: include/arks_hip.h)
			wb[it] = word_off[r];
		}
		G[it] = nwin[it] > 0 ? (nwin[it] + w - 1) / w : 0;
		const int incl = bk_wave_incl_scan(G[it]);
		pre[it] = incl - G[it];
		if (lane == 63)
			chunk_cnt[ci] = (u32)incl;
#pragma unroll
		for (int gi = 0; gi < kInline; ++gi)
			sd[it][gi] = gi < G[it] ? bucket_seed<MM>(codes, nmask, wb[it], nwin[it], w, gi, n_owners, mn[it]) : ~0ull;
		for (int gi = 0; gi < G[it]; ++gi) {
			const u64 v = gi < kInline ? (gi == 0 ? sd[it][0] : gi == 1 ? sd[it][1] : sd[it][2])
			                           : bucket_seed<MM>(codes, nmask, wb[it], nwin[it], w, gi, n_owners, mn[it]);
			if (v != ~0ull)
				atomicAdd(&cnt[(u32)(v >> 56)], 1u);
		}
	}
	__syncthreads();
	// one stretch of every owner's region, one of the seed numbering: a global atomic each
	if (threadIdx.x < n_owners) {
		const u32 c = cnt[threadIdx.x];
		u64 b = 0;
		if (c) {
			b = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->fill[threadIdx.x * kCtlStride]), (unsigned long long)c) -
			    cb.fill[threadIdx.x];
			if (b + c > cap)
				bad = 1;
		}
		base[threadIdx.x] = (u32)b;
		cnt[threadIdx.x] = 0; // handed out again below, seed by seed
	}
	if (threadIdx.x == 64) {
		u32 total = 0;
		for (int c = 0; c < kBkChunks; ++c)
			total += chunk_cnt[c];
		const u64 sb = atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->seeds[0]), (unsigned long long)total) - cb.seeds;
		if (sb + total > slot_cap)
			bad = 1;
		u32 run = (u32)sb;
		for (int c = 0; c < kBkChunks; ++c) {
			chunk_base[c] = run;
			run += chunk_cnt[c];
		}
	}
	__syncthreads();
	if (bad) { // the batch is run again with larger regions (the counters above say how large)
		if (threadIdx.x == 0)
			ctl->overflow[0] = cb.seq; // (every block that writes it writes the same number)
		return;
	}
	if (threadIdx.x < kBkChunks) {
		const long c = (long)blockIdx.x * kBkChunks + threadIdx.x;
		if (c * kBkChunk < n_reads)
			chunk_off[c] = chunk_base[threadIdx.x];
	}
#pragma unroll
	for (int it = 0; it < kBkChunksPerWave; ++it) {
		const int ci = wave * kBkChunksPerWave + it;
		const u32 first = chunk_base[ci] + (u32)pre[it];
		for (int gi = 0; gi < G[it]; ++gi) {
			const u64 v = gi < kInline ? (gi == 0 ? sd[it][0] : gi == 1 ? sd[it][1] : sd[it][2])
			                           : bucket_seed<MM>(codes, nmask, wb[it], nwin[it], w, gi, n_owners, mn[it]);
			u32 sl = ~0u;
			if (v != ~0ull) {
				const u32 o = (u32)(v >> 56);
				sl = (u32)((u64)o * cap) + base[o] + atomicAdd(&cnt[o], 1u);
				send[sl] = v & ((1ull << 56) - 1ull);
			}
			slot[first + gi] = sl;
		}
	}
}

__global__ void
gather_prep_kernel(const SeedBucketCtl* __restrict__ ctl, SeedBucketBase cb, u64 status, int n_owners, u64* __restrict__ out)
{
	const int o = (int)threadIdx.x;
	if (o == 0)
		out[0] = status;
	if (o < n_owners)
		out[1 + o] = status ? 0ull : ctl->fill[o * kCtlStride] - cb.fill[o];
}

hipError_t
launch_gather_prep(const SeedBucketCtl* ctl, const SeedBucketBase& base, u64 status, int n_owners, u64* out, hipStream_t st)
{
	gather_prep_kernel<<<1, 64, 0, st>>>(ctl, base, status, n_owners, out);
	return hipGetLastError();
}

__global__ void
peer_pattern_kernel(const u64* __restrict__ src, u64* __restrict__ dst, int n, u64 salt)
{
	const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (i < n)
		dst[i] = src[i] ^ salt ^ (u64)i;
}

hipError_t
launch_peer_pattern(const u64* src, u64* dst, int n, u64 salt, hipStream_t st)
{
	peer_pattern_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, dst, n, salt);
	return hipGetLastError();
}

long
seed_bucket_chunks(long n_reads)
{
	return (n_reads + kBkChunk - 1) / kBkChunk;
}

// seeds of a batch listed and bucketed by owner; ctl's counters stand at `base` (zero after creation) and count on.
// cap * n_owners <= 0xFFFFFFFE (slots are 32-bit).  A batch without reads still makes one (empty) launch when there is
// scratch to zero.
hipError_t
launch_seed_bucket(
    int mm, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval, long n_reads,
    int k, int w, u32 n_owners, u64 cap, u64 slot_cap, SeedBucketCtl* ctl, const SeedBucketBase& base,
    const SeedBucketGate& gate, u32* zero_words, int n_zero, u32* chunk_off, u32* slot, u64* send, hipStream_t st)
{
	if (n_owners < 1 || n_owners > (u32)kMaxOwners || cap * (u64)n_owners > 0xFFFFFFFEull || slot_cap > 0xFFFFFFFEull)
		return hipErrorInvalidValue;
	if (gate.read_class && (!gate.eval_out || (n_reads & 1)))
		return hipErrorInvalidValue;
	if (n_reads <= 0 && !zero_words)
		return hipSuccess;
	const unsigned nb = n_reads > 0 ? (unsigned)((n_reads + kBkReads - 1) / kBkReads) : 1u;
	if (mm == kMShort)
		seed_bucket_kernel<kMShort><<<nb, kBkWaves * 64, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, cap, slot_cap, ctl, base, gate, zero_words, n_zero, chunk_off, slot, send);
	else
		seed_bucket_kernel<kMLong><<<nb, kBkWaves * 64, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, cap, slot_cap, ctl, base, gate, zero_words, n_zero, chunk_off, slot, send);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

// bestContig's last step (Arcs/Arcs.cpp:1006-1013) as counters, for votes folded over the shards of a contig-sharded
// index: stats[5] (reads_pass) += reads with count / total > j_index, stats[6] (reads_fail) += the others -- of the
// reads bestContig is called for (eval != 0); the same arithmetic as resolve_votes_kernel
__global__ void
votes_count_kernel(
    const u64* __restrict__ votes, const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k,
    double j_index, u64* __restrict__ stats)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	bool called = r < n_reads;
	if (called && eval)
		called = eval[r] != 0;
	bool pass = false;
	if (called) {
		const int best_cnt = (int)(votes[r] >> 32);
		const int nwin = (int)lens[r] - k + 1;
		const int total = nwin > 0 ? nwin : 0;
		const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
		pass = maxj > j_index;
	}
	const u64 bp = __ballot(called && pass), bf = __ballot(called && !pass);
	if ((threadIdx.x & 63) == 0) {
		if (bp)
			atomicAdd(stats + 5, (u64)__popcll(bp));
		if (bf)
			atomicAdd(stats + 6, (u64)__popcll(bf));
	}
}

hipError_t
launch_votes_count(
    const u64* votes, const u32* lens, const uint8_t* eval, long n_reads, int k, double j_index, u64* stats, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	votes_count_kernel<<<(unsigned)((n_reads + 255) / 256), 256, 0, st>>>(votes, lens, eval, n_reads, k, j_index, stats);
	return hipGetLastError();
}

} // namespace arks
