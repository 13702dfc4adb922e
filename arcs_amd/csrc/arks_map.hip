// arks_map.hip -- read-mapping kernels of libarks_hip (gfx950 / CDNA4, wave64): bestContig
// (Arcs/Arcs.cpp:939-1014) for a batch, the pair rule of chromiumRead (:1264-1292) and the
// (barcode, contig end) accumulation.  Reference behaviour restated, never its code.
#include "arks_kernels.hpp"
#include <atomic>
#include <cstddef>

#include <cstdio>
#include <cstdlib>

namespace arks {

// ------------------------------------------------------------------------------------------------
// K3: read mapping -- bestContig (Arcs/Arcs.cpp:939-1014) for a batch.
// One wave per read at a time (grid-stride over reads), lanes = k-mer windows, up to kMaxPass
// passes of 64 windows.  Every lane builds its window key from the packed stream (no rolling
// state), canonicalises, hashes, and walks the open-addressed table; the per-read vote is a
// wave-level "smallest remaining value" loop, which reproduces the ascending std::map walk with its
// strict '<' (ties -> smallest contig-end index).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxPass = 4; // reads with <= 256 windows keep their window values in registers

__device__ __forceinline__ int
wave_min_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const int o = __shfl_xor(v, off);
		v = o < v ? o : v;
	}
	return v;
}

__device__ __forceinline__ int
wave_sum_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_xor(v, off);
	return v;
}

// number of set bits of m below this lane's own bit
__device__ __forceinline__ u32
mask_below(u64 m)
{
	return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// value of lane l (wave-uniform l) as a scalar: stays in SGPRs, unlike the result of __shfl
// inclusive prefix sum over the lanes of a wave by DPP (row shifts, then the row broadcasts of gfx9): no lane-index
// registers -- the shuffle form keeps six of them alive as loop invariants of the chunk loop (spilled at 64 VGPRs)
__device__ __forceinline__ int
wave_incl_scan_i32(int v)
{
#define ARKS_SCAN_ADD(ctrl, rows) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xF, true)
	ARKS_SCAN_ADD(0x111, 0xF); // row_shr:1
	ARKS_SCAN_ADD(0x112, 0xF); // row_shr:2
	ARKS_SCAN_ADD(0x114, 0xF); // row_shr:4
	ARKS_SCAN_ADD(0x118, 0xF); // row_shr:8
	ARKS_SCAN_ADD(0x142, 0xA); // row_bcast:15 -> rows 1 and 3
	ARKS_SCAN_ADD(0x143, 0xC); // row_bcast:31 -> rows 2 and 3
#undef ARKS_SCAN_ADD
	return v;
}

// the value of the next lane (0 for lane 63): wave_shl:1, a gfx9 DPP control -- no lane-index register
__device__ __forceinline__ u64
wave_next_u64(u64 v)
{
	const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x130, 0xF, 0xF, true);
	const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x130, 0xF, 0xF, true);
	return ((u64)hi << 32) | (u64)lo;
}

__device__ __forceinline__ u64
lane_value_u64(u64 v, int l)
{
	const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
	const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
	return ((u64)hi << 32) | lo;
}

// The waves of a launch add their counters into kStatRows partial rows (by block index): thousands of
// waves adding into the same eight words serialise at the L2; a one-wave kernel folds the rows into
// the caller's arks_map_stats afterwards.
constexpr int kStatRows = 64; // power of two
constexpr int kStatRow = 8;   // words per row
// work counters of the tile kernel behind those rows, one cache line apart
#ifndef ARKS_COUNTERS
#define ARKS_COUNTERS 8
#endif
constexpr int kCounters = ARKS_COUNTERS; // power of two
constexpr int kCounterStride = 32;       // u32 words between two counters (128 bytes)
constexpr size_t kWorkCtrOffset = 64 + (size_t)kStatRows * kStatRow * sizeof(u64);
static_assert(kWorkCtrOffset + (size_t)kCounters * kCounterStride * sizeof(u32) <= kMapScratchBytes, "scratch block");

// per-wave counters of arks_map_stats (uniform across the lanes of a wave)
struct WaveStats
{
	u64 valid, bad, found, rec, dup, pass, fail, win;
};

// value of window p of a read: -2 = NULL k-mer, -1 = absent, 0 = ambiguous, > 0 = contig end;
// with QUIRK = false a palindromic window is not resolved but reported as -4.
// BMODE = true: the locality index (serial lookup) instead of the plain hash table.
template <int KW, bool QUIRK, bool BMODE, int MM>
__device__ __forceinline__ int
window_value(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wbase, int p,
    const KeyGeom& g, const TableView& t, const BIndexView& bx)
{
	if (window_has_invalid<KW>(nmask, wbase, p, g.k))
		return -2;
	const Key<KW> f = window_key<KW>(codes, wbase, p, g);
	const Key<KW> r = key_revcomp(f, g);
	if (key_eq(f, r)) {
		if (!QUIRK)
			return -4;
		const Key<KW> c = key_palindrome_quirk(f, g);
		return BMODE ? fallback_lookup<KW>(bx, c) : table_lookup<KW>(t, c);
	}
	if (BMODE)
		return bindex_lookup_serial<KW, MM>(bx, g, codes, wbase * 32ull + (u64)p, f, r);
	Key<KW> c;
	const bool lt = key_less(f, r);
#pragma unroll
	for (int j = 0; j < KW; ++j)
		c.w[j] = lt ? f.w[j] : r.w[j];
	return table_lookup<KW>(t, c);
}

// Slow path only.  Reads with more than 64 * kMaxPass windows (not produced by the linked-read
// pipelines, but bestContig accepts any length): instead of holding the window values, re-scan the
// read once per distinct value in ascending order.
template <int KW, bool STATS, bool BMODE, int MM>
__device__ __forceinline__ void
vote_long_read(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wbase, int nwin,
    const KeyGeom& g, const TableView& t, const BIndexView& bx, int lane, WaveStats& ws, int& best,
    int& best_cnt)
{
	best = 0;
	best_cnt = 0;
	int prev = 0;
	bool first = true;
	for (;;) {
		int m = 0x7FFFFFFF, cnt = 0;
		for (int base = 0; base < nwin; base += 64) {
			const int p = base + lane;
			const int v = p < nwin ? window_value<KW, true, BMODE, MM>(codes, nmask, wbase, p, g, t, bx) : -3;
			if (STATS && first) {
				ws.bad += __popcll(__ballot(v == -2));
				ws.valid += __popcll(__ballot(v >= -1));
				ws.found += __popcll(__ballot(v >= 0));
				ws.rec += __popcll(__ballot(v > 0));
				ws.dup += __popcll(__ballot(v == 0));
			}
			if (v > prev) {
				if (v < m) {
					m = v;
					cnt = 1;
				} else if (v == m)
					cnt++;
			}
		}
		first = false;
		const int wm = wave_min_i32(m);
		if (wm == 0x7FFFFFFF)
			break;
		const int wc = wave_sum_i32(m == wm ? cnt : 0);
		if (wc > best_cnt) {
			best_cnt = wc;
			best = wm;
		}
		prev = wm;
	}
}

// What a mapping kernel writes for read r.  RAW = false: bestContig's return value (Arcs.cpp:1006-1013).
// RAW = true (sharded index, arks_map_votes_device): the winner of the walk of Arcs.cpp:998-1004 before
// the j_index test, as one u64 whose unsigned maximum over index shards is the winner over all of them:
// count in the high word, ~conreci in the low word (a tie keeps the smaller conreci); 0 = nothing recorded.
template <bool RAW>
__device__ __forceinline__ void
put_result(int* __restrict__ out, long r, int best, int best_cnt, bool pass)
{
	if (RAW)
		reinterpret_cast<u64*>(out)[r] = best_cnt > 0 ? (((u64)(u32)best_cnt << 32) | (u64)(~(u32)best)) : 0ull;
	else
		out[r] = pass ? best : 0;
}
template <bool RAW>
__device__ __forceinline__ void
put_none(int* __restrict__ out, long r)
{
	if (RAW)
		reinterpret_cast<u64*>(out)[r] = 0ull;
	else
		out[r] = 0;
}

// FAST = true : the hot kernel.  Grid-stride over all reads; a read that needs one of the rare
//               paths (a reverse-complement palindrome window, whose key takes the reference's
//               damaged branch; or more than 64 * kMaxPass windows) is appended to `queue`
//               untouched, which keeps those paths' registers out of this kernel.
// FAST = false: the same algorithm with every path, over the reads listed in `queue`.
template <int KW, bool STATS, bool FAST, bool BMODE, int MM, bool RAW = false>
__global__ void __launch_bounds__(256)
map_reads_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ nmask,
    const u64* __restrict__ word_off,
    const u32* __restrict__ lens,
    const uint8_t* __restrict__ eval, // may be NULL
    long n_reads,
    double j_index,
    KeyGeom g,
    TableView t,
    BIndexView bx,
    int* __restrict__ out_conreci,
    u64* __restrict__ stats, // arks_map_stats layout, may be NULL when !STATS
    u32* __restrict__ queue,
    u32* __restrict__ queue_count)
{
	const int lane = threadIdx.x & 63;
	const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const long n_waves = ((long)gridDim.x * blockDim.x) >> 6;
	WaveStats ws = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const long n_items = FAST ? n_reads : (long)*queue_count;

	for (long it = wave; it < n_items; it += n_waves) {
		const long r = FAST ? it : (long)queue[it];
		if (FAST && eval && !eval[r]) {
			if (lane == 0)
				put_none<RAW>(out_conreci, r);
			continue;
		}
		const int nwin = (int)lens[r] - g.k + 1; // <= 0: the loop of Arcs.cpp:959 never runs
		const u64 wbase = word_off[r];
		int best = 0, best_cnt = 0;
		WaveStats rs = { 0, 0, 0, 0, 0, 0, 0, 0 }; // this read's window counters
		bool redo = false;
		if (nwin > 64 * kMaxPass) {
			if (FAST)
				redo = true;
			else
				vote_long_read<KW, STATS, BMODE, MM>(codes, nmask, wbase, nwin, g, t, bx, lane, rs, best, best_cnt);
		} else {
			int vals[kMaxPass];
#pragma unroll
			for (int ps = 0; ps < kMaxPass; ++ps) {
				int v = -3;
				if (ps * 64 < nwin) { // wave-uniform
					const int p = ps * 64 + lane;
					if (p < nwin)
						v = window_value<KW, !FAST, BMODE, MM>(codes, nmask, wbase, p, g, t, bx);
					if (FAST)
						redo = redo || __ballot(v == -4) != 0;
					if (STATS) {
						rs.bad += __popcll(__ballot(v == -2));
						rs.valid += __popcll(__ballot(v >= -1));
						rs.found += __popcll(__ballot(v >= 0));
						rs.rec += __popcll(__ballot(v > 0));
						rs.dup += __popcll(__ballot(v == 0));
					}
				}
				vals[ps] = v > 0 ? v : 0; // only real contig ends are counted (Arcs.cpp:972-973)
			}
			// ---- vote: ascending walk over the distinct non-zero values (Arcs.cpp:996-1004) ----
			if (!redo) {
				for (;;) {
					int m = 0x7FFFFFFF;
#pragma unroll
					for (int ps = 0; ps < kMaxPass; ++ps)
						m = (vals[ps] != 0 && vals[ps] < m) ? vals[ps] : m;
					m = wave_min_i32(m);
					if (m == 0x7FFFFFFF)
						break;
					int cnt = 0;
#pragma unroll
					for (int ps = 0; ps < kMaxPass; ++ps) {
						const bool is = vals[ps] == m;
						cnt += __popcll(__ballot(is));
						vals[ps] = is ? 0 : vals[ps];
					}
					if (cnt > best_cnt) { // strict: the first (smallest) value keeps a tie
						best_cnt = cnt;
						best = m;
					}
				}
			}
		}
		if (FAST && redo) {
			if (lane == 0)
				queue[atomicAdd(queue_count, 1u)] = (u32)r;
			continue;
		}
		// maxjaccardindex > j_index with maxjaccardindex = (double)count / (double)total, or 0
		// when nothing was recorded (Arcs.cpp:996,1006); total counts NULL windows too (:962)
		const int total = nwin > 0 ? nwin : 0;
		const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
		const bool pass = maxj > j_index;
		if (lane == 0)
			put_result<RAW>(out_conreci, r, best, best_cnt, pass);
		if (STATS) {
			ws.valid += rs.valid;
			ws.bad += rs.bad;
			ws.found += rs.found;
			ws.rec += rs.rec;
			ws.dup += rs.dup;
			ws.pass += pass;
			ws.fail += !pass;
			ws.win += (u64)total;
		}
	}
	if (STATS && lane == 0) {
		if (ws.valid) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 0, ws.valid);
		if (ws.bad) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 1, ws.bad);
		if (ws.found) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 2, ws.found);
		if (ws.rec) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 3, ws.rec);
		if (ws.dup) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 4, ws.dup);
		if (ws.pass) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 5, ws.pass);
		if (ws.fail) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 6, ws.fail);
		if (ws.win) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 7, ws.win);
	}
}

// ------------------------------------------------------------------------------------------------
// K3b: read mapping over the locality index -- the hot kernel.
//
// One wave (= one 64-thread workgroup) processes a TILE: consecutive reads in up to kTW packed words,
// laid out as one position space in LDS.  The minimizer front half (T2-T4) runs once per PASS of
// kSW words (64 lanes x 8 positions), the back half (T5-T7) once per tile so that its lanes
// (run heads, 32-base words x 2 diagonals) are full.  FULL = false is the hot instantiation, FULL =
// true the medium one (one queued read per tile, per-window records, general verification).
//   T1  stage the tile's code / N-mask words in LDS (every later access to the reads is LDS);
//   T2  lane l owns positions 8l..8l+7: one funnel shift gives the <= 32 bases its m-mers span,
//       every forward / reverse-complement m-mer is a shift + mask of that register pair; order
//       value = 20-bit hash (two 24-bit multiply-adds) above position and strand bit;
//   T3  sliding-window minimum: suffix minima of the lane's own 8 values (registers) + block minima
//       of the whole lanes in between + prefix minima of a later lane (one LDS round trip);
//   T4  run heads (first window of a run sharing a minimizer), numbered by ballot + mbcnt;
//       reverse-complement palindromes flagged through their mirrored minimizer when the index
//       holds quirk images (-> slow kernel, which evaluates the reference's damaged key exactly);
//   T5  one lane per run head walks the minimizer table (~5 runs per 151-bp read), 4 entries per
//       round trip, and publishes <= 2 matching entries (text position, strand);
//   T6a lanes = run heads: the read's two diagonals (LDS atomic minima keyed by run index); a read
//       with a heavy / overflowing run or a third diagonal goes to the medium queue;
//   T6b lanes = read words x 2 diagonals: text / visited / ambiguous / owner words staged in one
//       round trip, read XOR text -> one mismatch bit per base;
//   T6c lanes = 32-window words x 2 diagonals, one bit per window: exists & no invalid base, k clear
//       mismatch bits (constant time), visited, ambiguous; popcounts -> per-read counters (LDS
//       atomics).  Exact: membership is still key equality (Arcs/Arcs.h:153-156), the value comes
//       from the position's visited / ambiguous bits and owner word;
//   T7  lane = read: vote exactly as bestContig does (Arcs.cpp:996-1013) -- at most two distinct
//       positive values (one per diagonal), so the ordered walk is a compare.
// HBM traffic per read: ~5 random 8-B minimizer entries + a few text / bit words, instead of ~100
// random 64-B lines of the hash-table design.
// ------------------------------------------------------------------------------------------------
constexpr int kSW = 16;          // words per minimizer pass (64 lanes x 8 positions = 512 bases)
constexpr int kTP = kSW * 32;    // ... in base positions
constexpr int kTW = 32;          // tile capacity in packed words: two passes, one joint back half
constexpr int kTR = 16;          // reads per tile
constexpr int kNH = 128;         // run heads per tile that get a published entry list
// waves per SIMD of the seed index's medium kernel at k <= 64 (7.7 KB of LDS, 96 VGPRs): 5; 4 until the end of
// round 5 (9.3 KB, 112 VGPRs: 6.21 against 5.72 ms per 20 M pairs of the human-like draft) and still 4 for
// longer keys (120-124 VGPRs)
#ifndef ARKS_MEDIUM_WAVES
#define ARKS_MEDIUM_WAVES 5
#endif
constexpr int kNHd = 64;         // ... in the medium kernel of the seed index (group seeds + extra seeds: one probe round)
constexpr int kGrabF = 12;       // medium kernel: queued reads per grab at most (four tiles of 10x reads; 8 until round 5: tiles of 3 + 3 + 2)
constexpr int kSettleMargin = 2; // medium kernel without counters: open windows probed beyond the number that settles a failing vote
constexpr int kExtraSeeds = 12;  // medium kernel, seed index: m-mers per read probed beside its group seeds (8: fewer proofs; 18: the same time)
constexpr int kChunk = 48;       // reads handed out per grab of the work counter (lane l holds read l's metadata):
                                 // a multiple of the usual reads per tile (4, 6, 8, 12), small enough for an even
                                 // finish, large enough for the counter (same-address atomics serialise at ~12 ns:
                                 // 24 reads per grab made the COUNTER the kernel's run time, 20 ms at C2)
constexpr u32 kHnHeavy = 255, kHnOverflow = 254;
constexpr int kRecFallback = -4; // window record of the medium kernel: "probe the fallback table" (between T6c and T6d)
constexpr int kRecOverflow = -5; // likewise: "walk the entries of the window's seed" (a seed with 3-8 entries)

// Bit b of the result: none of the positions [b, b + k) of the 128-bit vector m3:m2:m1:m0 (m0 = positions
// 0..31) is set, for b in [0, 32) and k in [32, 96].  Such a span always reaches the end of m0, so: b lies
// above the highest set bit of m0, and b + k does not pass the first set bit of the later words.
__device__ __forceinline__ u32
clear_spans32(u32 m0, u32 m1, u32 m2, u32 m3, int k)
{
	const int hi = 32 - __clz((int)m0); // clz(0) = 32
	const u32 r0 = hi == 32 ? 0u : (0xFFFFFFFFu << hi);
	int f = 128;
	f = m3 ? 95 + __ffs((int)m3) : f;
	f = m2 ? 63 + __ffs((int)m2) : f;
	f = m1 ? 31 + __ffs((int)m1) : f;
	const int top = f - k; // the highest admissible b
	const u32 r1 = top >= 31 ? 0xFFFFFFFFu : (top < 0 ? 0u : ((2u << top) - 1u));
	return r0 & r1;
}

#ifndef ARKS_TILE_WAVES
#define ARKS_TILE_WAVES 7 // 72 VGPRs: at 8 waves (64) the kernel keeps two or three registers in scratch memory -- no kernel of the library uses scratch (tests/test_abi.py)
#endif

// DMED = the medium kernel of the SEED index (FULL && DENSE): no sliding minimum (S.a holds the seeds' entry lists and
// the per-word match masks only), seeds capped at kNHd per tile -- 7.7 KB instead of 9.3, which is what five waves per
// SIMD need (round 5: the kernel's time is LDS-latency chains at four)
template <bool FULL, bool DMED = false>
struct TileLds
{
	u32 a[DMED ? 4 * kNHd + 96 : kTP + 96];
	// hot instantiation: the block minima of T3, then the per-word result bits (no window records)
	u32 b[DMED ? kTP : (FULL ? kTP + 96 : 256)];
	u64 cw[kTW + 4];
	u32 nm[kTW + 4];
	unsigned short heads[DMED ? kNHd : kNH]; // run heads: [0] minimizer strand, [11:1] its position, [15:12] read of the tile
	unsigned char hn[DMED ? kNHd : kNH];
	unsigned char wread[kTW + 4];
	u32 wmeta[kTW + 4]; // per word: read index << 16 | local end position of that read (0 = none)
	int rstart[kTR + 1];
	int rlen[kTR];
	u64 pdiag[kTR][2]; // the read's two staged diagonals: [39:0] D, [40] same strand, [41] valid
	u32 tfirst[kTR][2];             // first text word staged for read j on diagonal d
	// text words along the diagonals (read j: slots from (rstart[j] >> 5) + j): the hot instantiation
	// keeps them in storage that is free by then (S.a, S.b; see the kernel), the medium one here
	u64 tcodes_f[FULL ? 2 * (kTW + kTR + 2) : 1];
	u32 tvis_f[FULL ? 2 * (kTW + kTR + 2) : 1];
	u32 tamb_f[FULL ? 2 * (kTW + kTR + 2) : 1];
	u32 town_f[FULL ? 2 * (kTW + kTR + 2) : 1];
	u32 mm32_f[FULL ? 2 * (kTW + 8) : 1]; // mismatch bit per base along the diagonals
	unsigned char sread[kTW + kTR + 2];
	int hbase[FULL ? kTR : 1]; // seed index, medium kernel: first seed (head) of every read
	// medium kernel: the reads of a tile are queue entries, anywhere in the batch: word in the batch's packed arrays
	// of every read's first word, and the read's number
	u64 rbase[FULL ? kTR : 1];
	u32 rid[FULL ? kTR : 1];
	// medium kernel, seed index: one bit per tile position -- the window there holds an m-mer that is in no visited
	// window of the text (T5's seeds without an entry, T6e's probes), so it is absent whatever its own seed says
	// absent[1]: ... unless it matches the text on one of the read's staged diagonals (a seed all of whose one or two
	// entries lie there: looked at only for windows that were compared on both and did not match)
	u32 absent[2][FULL ? kTP / 32 + 4 : 1];
	int vfail[FULL ? kTR : 1]; // medium kernel without counters: the largest vote count with which read j fails (T6d)
	u32 redo;
	u32 redo2; // reads that need the general verification (hot instantiation only)
	u64 wstats[8]; // hot instantiation: this wave's arks_map_stats counters (registers are scarce there)
};

// lanes of one wave communicate through LDS: order the compiler's view of it
#define ARKS_WAVE_SYNC()                                                                           \
	do {                                                                                           \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
		__builtin_amdgcn_wave_barrier();                                                           \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
	} while (0)

// forward key of the window at local position i of the staged tile words
template <int KW>
__device__ __forceinline__ Key<KW>
tile_window_key(const u64* cw, int i, const KeyGeom& g)
{
	const int wi = i >> 5, sft = (i & 31) * 2;
	u64 w[KW + 1];
#pragma unroll
	for (int j = 0; j <= KW; ++j)
		w[j] = cw[wi + j];
	Key<KW> f;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		f.w[j] = funnel_l(w[j], w[j + 1], sft) & g.mask[j];
	return f;
}

#ifdef ARKS_MEDIUM_DIAG
// quarantined diagnostic (-DARKS_MEDIUM_DIAG; arks_debug_medium_diag, printed by profiles/tools/ab.py): counters of the
// medium kernel's tiles -- 0 tiles, 1 reads, 2 reads with a diagonal, 9 seeds probed, 11 tiles with a second diagonal,
// 6 windows proven absent (high word: by a seed without entries; low: by a seed whose entries are all staged), 3 windows
// left to T6d (4: of reads without a diagonal, 5: of reads whose diagonal A differs in > 8 bases), 7 windows listed,
// 8 probe rounds, 12 slot reads, 13 reads settled before any lookup, 14 reads with a first round, 10 settled after it.
// The counters cost half of the kernel's time: shares, not times.
__device__ unsigned long long g_med_diag[16];
#define ARKS_MD(slot, v) md[slot] += (unsigned long long)(v)
#else
#define ARKS_MD(slot, v) ((void)0)
#endif
#ifdef ARKS_PROFILE_SECTIONS
__device__ unsigned long long g_sec_cycles[16];
#define ARKS_SEC(nsec)                                                                             \
	do {                                                                                           \
		const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                               \
		sec_acc[nsec] += t1_ - sec_t0;                                                             \
		sec_t0 = t1_;                                                                              \
	} while (0)
#else
#define ARKS_SEC(nsec)                                                                             \
	do {                                                                                           \
	} while (0)
#endif

// forward m-mer at local position i of the staged tile words, right-aligned in 2*MM bits
template <int MM>
__device__ __forceinline__ typename Mmer<MM>::type
tile_mmer(const u64* cw, int i)
{
	return (typename Mmer<MM>::type)(funnel_l(cw[i >> 5], cw[(i >> 5) + 1], (i & 31) * 2) >> (64 - 2 * MM));
}

template <int MM>
__device__ __forceinline__ typename Mmer<MM>::type
tile_canonical_mmer(const u64* cw, int i)
{
	const typename Mmer<MM>::type mf = tile_mmer<MM>(cw, i);
	const typename Mmer<MM>::type mr = mmer_rc<MM>(mf);
	return mf < mr ? mf : mr;
}

// 128-bit helpers for the bit-parallel window tests (bit b of the pair = position b)
struct U128
{
	u64 lo, hi;
};

__device__ __forceinline__ U128
shr128(U128 v, int s) // s in [0, 127]
{
	U128 r;
	if (s >= 64) {
		r.lo = v.hi >> (s - 64);
		r.hi = 0;
	} else {
		r.lo = funnel_r(v.lo, v.hi, s);
		r.hi = v.hi >> s;
	}
	return r;
}

// bit b of the result = AND of bits [b, b + k) of v   (k >= 1; bits beyond 127 count as 1)
__device__ __forceinline__ U128
and_window128(U128 v, int k)
{
	// p = AND over `have` consecutive bits, doubled while it fits; the remainder is combined from
	// the powers of two in k
	U128 res = { ~0ull, ~0ull };
	U128 p = v;
	int have = 1, off = 0;
	for (int bit = 0; bit < 7; ++bit) {
		if (k & have) {
			const U128 sh = shr128(p, off);
			// bits shifted in from beyond 127 are zeros: make them ones
			res.lo &= sh.lo | (off >= 64 ? (off - 64 >= 64 ? ~0ull : ~(~0ull >> (off - 64))) : 0ull);
			res.hi &= sh.hi | (off == 0 ? 0ull : (off >= 64 ? ~0ull : ~(~0ull >> off)));
			off += have;
		}
		if (2 * have > k)
			break;
		const U128 sh = shr128(p, have);
		p.lo &= sh.lo;
		p.hi &= sh.hi | ~(~0ull >> have);
		have *= 2;
	}
	return res;
}

// FULL = false: the hot instantiation.  A window that needs the general verification (an entry off
//                both staged diagonals of its read, a heavy minimizer, a run with more than two
//                entries) sends its read to the "medium" queue instead, which keeps that code's
//                registers out of the hot kernel.
// FULL = true : the same kernel over the medium queue, one queued read per tile, every path.
// ---- T5 helper: the text occurrences of the canonical m-mer cm in the minimizer table: up to two
//      entries are written to `entries`; returns their number, kHnOverflow for more, kHnHeavy when the
//      table holds the "heavy: ask the fallback table" marker.  Home slots and the capacity are multiples
//      of 4 (mtab_home): every round trip reads one aligned group of four entries. -------------------
// HALF16: two entries (16 bytes, one dwordx4) per round trip instead of the aligned group of four.  A random 16-byte
// read costs the memory system less than a 32-byte one (profiles/r04j_gather_width2.txt: 4.9e10 against 3.9e10 per
// second over a 32 GiB table), and at the table's load three probe sequences of four end within two slots; the others
// find the second half of the group in the L2.  Pays where probing is ALL a kernel does (the owner-side
// seeds_probe_segs_kernel of the sharded seed table); in the tile kernels it was the same time within the noise
// (profiles/r04k_ab_probe16.txt) and is not used there.
template <int MM, bool HALF16 = false>
__device__ __forceinline__ u32
probe_minimizer_table(const BIndexView& bx, typename Mmer<MM>::type cm, u64* entries)
{
	const u32 fp = mmer_fp<MM>(cm);
	u64 slot = mtab_home<MM>(cm, bx.mtab_cap);
	u32 cnt = 0;
	bool end = false;
	while (HALF16 && !end) {
		// (a non-temporal load here -- 5.5e10 against 4.9e10 16-byte reads per second in the microbenchmark -- makes the
		// second half of a group miss as well: 4.6 instead of 3.5 ms per 25 M pairs, profiles/r04x_probe_nt.txt)
		const ulonglong2 h = *reinterpret_cast<const ulonglong2*>(bx.mtab + slot);
		const u64 ev[2] = { h.x, h.y };
#pragma unroll
		for (int x = 0; x < 2; ++x) {
			const u64 e = ev[x];
			if (end)
				continue;
			if (!(e >> 63)) {
				end = true;
				continue;
			}
			if (((u32)(e >> 32) & kFpMask) != fp)
				continue;
			if ((u32)e == kHeavyPos) {
				cnt = kHnHeavy;
				end = true;
				continue;
			}
			if (cnt == 0)
				entries[0] = e;
			if (cnt == 1)
				entries[1] = e;
			cnt = cnt < 2 ? cnt + 1 : kHnOverflow;
			if (cnt == kHnOverflow)
				end = true;
		}
		slot += 2;
		if (slot >= bx.mtab_cap)
			slot = 0;
	}
	while (!end) {
		u64 ev[4];
#pragma unroll
		for (int x = 0; x < 4; ++x)
			ev[x] = bx.mtab[slot + x];
#pragma unroll
		for (int x = 0; x < 4; ++x) {
			const u64 e = ev[x];
			if (end)
				continue;
			if (!(e >> 63)) { // an empty slot ends the probe sequence
				end = true;
				continue;
			}
			if (((u32)(e >> 32) & kFpMask) != fp)
				continue;
			if ((u32)e == kHeavyPos) {
				cnt = kHnHeavy;
				end = true;
				continue;
			}
			// (no dynamic index: `entries` may be a pair of registers)
			if (cnt == 0)
				entries[0] = e;
			if (cnt == 1)
				entries[1] = e;
			cnt = cnt < 2 ? cnt + 1 : kHnOverflow;
			if (cnt == kHnOverflow)
				end = true;
		}
		slot += 4;
		if (slot >= bx.mtab_cap)
			slot = 0;
	}
	return cnt;
}

// ---- T6b helper: the 32 bases of one packed read word against the text they face on the diagonal pdv
//      ([39:0] D, [40] same strand): bit b of the result = base b of the word differs.  x0 = read offset of
//      the word's first base, sb = first staging slot of the read, tfirst = text word staged in slot sb,
//      tcodes = the staged text words of this diagonal. -------------------------------------------------
__device__ __forceinline__ u32
word_mismatch_bits(u64 code_word, u64 pdv, const u64* tcodes, int x0, int sb, u32 tfirst)
{
	const bool same = (pdv >> 40) & 1ull;
	const u64 D = pdv & 0xFFFFFFFFFFull;
	// text position of the lowest-addressed base this word faces
	const u64 tlo = same ? D + (u64)x0 : D - (u64)(x0 + 31);
	const int slot = sb + (int)((u32)(tlo >> 5) - tfirst);
	const u64 t0 = slot >= sb ? tcodes[slot] : 0ull;
	const u64 t32 = funnel_l(t0, tcodes[slot + 1], (int)(tlo & 31) * 2);
	const u64 face = same ? t32 : ~rev_groups(t32);
	u64 x = code_word ^ face;
	// one bit per base: OR the two bits of every group, gather the even bits
	x = (x | (x >> 1)) & 0x5555555555555555ull;
	x = (x | (x >> 1)) & 0x3333333333333333ull;
	x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
	x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
	x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
	return __brev((u32)x); // base 0 of the word -> bit 0
}

// ---- T6c' helpers: 32 window starts (one packed word of a read) at a time, one bit per window ---------
// windows of the word that exist (`left` = windows of the read from the word's first position on) and hold
// no invalid base (nm = the tile's N masks, bit 31 = first base of a word)
__device__ __forceinline__ u32
word_valid_windows(const u32* nm, int wl, int left, int k, bool has_n)
{
	const int cexist = left < 0 ? 0 : (left > 32 ? 32 : left);
	u32 valid = cexist == 32 ? 0xFFFFFFFFu : ((1u << cexist) - 1u);
	if (has_n && valid) {
		if (k >= 32) {
			valid &= clear_spans32(__brev(nm[wl]), __brev(nm[wl + 1]), __brev(nm[wl + 2]), __brev(nm[wl + 3]), k);
		} else {
			U128 nf;
			nf.lo = ~((u64)__brev(nm[wl]) | ((u64)__brev(nm[wl + 1]) << 32));
			nf.hi = ~((u64)__brev(nm[wl + 2]) | ((u64)__brev(nm[wl + 3]) << 32));
			valid &= (u32)and_window128(nf, k).lo;
		}
	}
	return valid;
}

// Windows of word wl that match the text on the diagonal pdv ([39:0] D, [40] same strand): their k mismatch
// bits (mm: one bit per base along the diagonal) are all clear, and the text position they map to is
// `visited` (tvis) -- `ok`; of those, the ones whose position is `ambig` (tamb, value 0) -- `amb`; `own` =
// contig end (town) of the others, ~0 if they span two.  p0 = first window of the word relative to the
// read, sb = first staging slot of the read, tfirst = text word staged in slot sb.
__device__ __forceinline__ void
word_match(
    u64 pdv, const u32* mm, const u32* tvis, const u32* tamb, const u32* town, int wl, int p0, int sb, u32 tfirst,
    int k, u32 valid, u32& ok, u32& amb, u32& own)
{
	if (k >= 32) {
		ok = clear_spans32(mm[wl], mm[wl + 1], mm[wl + 2], mm[wl + 3], k) & valid;
	} else {
		U128 z; // match bit per base from this word on
		z.lo = ~((u64)mm[wl] | ((u64)mm[wl + 1] << 32));
		z.hi = ~((u64)mm[wl + 2] | ((u64)mm[wl + 3] << 32));
		ok = (u32)and_window128(z, k).lo & valid;
	}
	if (!ok)
		return;
	const bool same = (pdv >> 40) & 1ull;
	const u64 D = pdv & 0xFFFFFFFFFFull;
	// text positions of the 32 window starts: same strand D + p0 + b, opposite strand (D - k + 1 - p0) - b;
	// `lo` = the lowest of them
	const u64 lo = same ? D + (u64)p0 : D - (u64)(k - 1 + p0 + 31);
	const int slot = sb + (int)((u32)(lo >> 5) - tfirst);
	const int sh = (int)(lo & 31);
	// (no range checks: a slot outside the read's staged slots can only feed bits of windows that do not exist,
	// and `ok` has lost those already; the reads stay inside the staging storage)
	const u32 v0 = tvis[slot], v1 = tvis[slot + 1];
	const u32 a0 = tamb[slot], a1 = tamb[slot + 1];
	// 32 bits from text position lo on, most significant = lo
	u32 vis = sh ? ((v0 << sh) | (v1 >> (32 - sh))) : v0;
	u32 am = sh ? ((a0 << sh) | (a1 >> (32 - sh))) : a0;
	if (same) { // bit b must be position lo + b
		vis = __brev(vis);
		am = __brev(am);
	} // opposite strand: bit b is position lo + 31 - b already
	ok &= vis;
	amb = ok & am;
	// contig end of the matched, unambiguous windows: the word(s) their text positions fall in.
	// `inlo` = window bits whose position lies in `slot`.
	const u32 recm = ok & ~amb;
	const u32 inlo = same ? (sh ? (1u << (32 - sh)) - 1u : 0xFFFFFFFFu) : (0xFFFFFFFFu << sh);
	const u32 o0 = (recm & inlo) ? town[slot] : 0u;
	const u32 o1 = (recm & ~inlo) ? town[slot + 1] : 0u;
	own = o0 ? o0 : o1;
	if (o0 && o1 && o0 != o1)
		own = 0xFFFFFFFFu;
}

// does one of the `len` (<= 32) bases from tile position q on hold an invalid character?
__device__ __forceinline__ bool
tile_span_has_n(const u32* nm, int q, int len)
{
	const u64 two = ((u64)nm[q >> 5] << 32) | (u64)nm[(q >> 5) + 1];
	return ((two << (q & 31)) >> (64 - len)) != 0;
}

// ---- T2: order values of the 8 positions a lane owns (tile positions i0 .. i0 + 7, all in one packed
//      word, i.e. one read).  The 8 + MM - 1 <= 32 bases their m-mers span are ONE funnel shift of two
//      staged words; every forward m-mer is a shift + mask of that register pair, every reverse
//      complement a shift + mask of its complement.  v[t] = mmer_order << 12 | position << 1 | strand
//      (1 = the forward m-mer is the canonical one), or ~0 when the m-mer leaves the read (rem0 = bases
//      of the read from i0 on) or holds an invalid base. ------------------------------------------------
template <int MM>
__device__ __forceinline__ void
tile_order_values(const u64* cw, const u32* nm, int i0, int lane, int rem0, bool has_n, u32 (&v)[8])
{
	typedef typename Mmer<MM>::type mm_t;
	const int wq = i0 >> 5, sft = (lane & 3) * 16;
	const u64 x = funnel_l(cw[wq], cw[wq + 1], sft);
	const u64 xr = ~rev_groups(x);
	u32 nb = 0;
	if (has_n)
		nb = (nm[wq] << (sft >> 1)) | ((nm[wq + 1] >> 1) >> (31 - (sft >> 1)));
	const mm_t mmask = (mm_t)((1ull << (2 * MM)) - 1ull);
	// positions whose m-mer lies inside the read (t <= rem0 - MM) and holds no invalid base
	const int tmax = rem0 - MM;
	u32 okmask = tmax >= 7 ? 0xFFu : (tmax < 0 ? 0u : ((2u << tmax) - 1u));
	if (nb) {
#pragma unroll
		for (int t = 0; t < 8; ++t)
			okmask &= ((nb << t) >> (32 - MM)) ? ~(1u << t) : 0xFFFFFFFFu;
	}
#pragma unroll
	for (int t = 0; t < 8; ++t) {
		v[t] = 0xFFFFFFFFu;
		if ((okmask >> t) & 1u) {
			const mm_t mf = (mm_t)(x >> (64 - 2 * MM - 2 * t)) & mmask;
			const mm_t mr = (mm_t)(xr >> (2 * t)) & mmask;
			const mm_t cm = mf < mr ? mf : mr;
			v[t] = (mmer_order<MM>(cm) << 12) | ((u32)(i0 + t) << 1) | (mf < mr ? 1u : 0u);
		}
	}
}

// ---- T3: sliding minimum over w positions for the 8 windows a lane owns.  [i, i + w) = a suffix of the
//      lane's own 8 values, whole lanes in between, a prefix of a later lane's 8: prefix minima go through
//      LDS (pre: kTP + pad words, the pad holds ~0), the lanes' block minima too (blk: 64 + 16 words);
//      everything else is registers.  l0 = 8 * lane.  Ends behind a wave sync of its own; the caller must
//      sync before anything else overwrites pre / blk. ---------------------------------------------------
__device__ __forceinline__ void
tile_sliding_min(u32* pre_lds, u32* blk_lds, int l0, int lane, int w, const u32 (&v)[8], u32 (&wmin)[8])
{
	if (w >= 9) {
		u32 pre[8], suf[8];
		pre[0] = v[0];
#pragma unroll
		for (int t = 1; t < 8; ++t)
			pre[t] = v[t] < pre[t - 1] ? v[t] : pre[t - 1];
		suf[7] = v[7];
#pragma unroll
		for (int t = 6; t >= 0; --t)
			suf[t] = v[t] < suf[t + 1] ? v[t] : suf[t + 1];
		uint4* pa = reinterpret_cast<uint4*>(pre_lds + l0);
		pa[0] = make_uint4(pre[0], pre[1], pre[2], pre[3]);
		pa[1] = make_uint4(pre[4], pre[5], pre[6], pre[7]);
		blk_lds[lane] = pre[7];
		if (lane < 16)
			blk_lds[64 + lane] = 0xFFFFFFFFu;
		ARKS_WAVE_SYNC();
		const int e0 = (w - 1) >> 3, tb = 8 - ((w - 1) & 7);
		u32 fa = 0xFFFFFFFFu;
		for (int x = 1; x < e0; ++x) {
			const u32 y = blk_lds[lane + x];
			fa = y < fa ? y : fa;
		}
		u32 fb = blk_lds[lane + e0];
		fb = fb < fa ? fb : fa;
#pragma unroll
		for (int t = 0; t < 8; ++t) {
			const u32 y = pre_lds[l0 + t + w - 1];
			const u32 f = t < tb ? fa : fb;
			u32 m = suf[t] < y ? suf[t] : y;
			wmin[t] = f < m ? f : m;
		}
	} else { // short windows: each window reads its w values
		uint4* pa = reinterpret_cast<uint4*>(pre_lds + l0);
		pa[0] = make_uint4(v[0], v[1], v[2], v[3]);
		pa[1] = make_uint4(v[4], v[5], v[6], v[7]);
		ARKS_WAVE_SYNC();
#pragma unroll
		for (int t = 0; t < 8; ++t) {
			u32 m = v[t];
			for (int o = 1; o < w; ++o) {
				const u32 y = pre_lds[l0 + t + o];
				m = y < m ? y : m;
			}
			wmin[t] = m;
		}
	}
}

// DENSE = true: the seed index (BIndexView::dense).  Every m-mer position of the text is in the table, so
//                a window may be looked up through ANY m-mer it contains: the windows of a read are cut
//                into groups of w = k - MM + 1 consecutive ones, the m-mer at the start of a group's last
//                window lies inside every window of the group and is the group's seed (its "run head").
//                No order values, no sliding minimum, no run detection (T2-T4 -- 44 % of the minimizer
//                kernel's time); 2-3 probes per 10x read instead of 5-6.  Exact for the same reason: a
//                window that is in the index brings its seed's text position into the seed's entry list.
template <int KW, bool STATS, bool FULL, int MM, bool RAW = false, bool DENSE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FULL ? (DENSE && KW <= 2 ? ARKS_MEDIUM_WAVES : 4) : ARKS_TILE_WAVES)))
map_reads_b_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ nmask,
    const u64* __restrict__ word_off,
    const u32* __restrict__ lens,
    const uint8_t* __restrict__ eval, // may be NULL
    long n_reads,
    double j_index,
    KeyGeom g,
    BIndexView bx,
    int* __restrict__ out_conreci,
    u64* __restrict__ stats,
    u32* __restrict__ queue,       // slow queue (palindromes next to quirk images, reads longer than a tile)
    u32* __restrict__ mqueue,      // medium queue
    u32* __restrict__ queue_count) // [0] slow length, [1] work counter, [2] medium length, [3] medium work counter
{
	constexpr bool kDMed = FULL && DENSE;
	constexpr int kNHx = kDMed ? kNHd : kNH; // seeds (run heads) of a tile that get an entry list
	__shared__ TileLds<FULL, kDMed> S;
	// entry lists of the run heads (T5 .. T6a; the medium kernel reads them again in T6c): S.a is free
	// once the window minimizers are taken
	u64 (*const hc)[2] = reinterpret_cast<u64(*)[2]>(S.a);
	// staged text words of T6 (written after T6a, when the entry lists are done with).  Hot: codes, visited /
	// ambiguous words and the mismatch bits of T6b all over S.a, owners over the block minima of S.b.
	constexpr int kSlots = kTW + kTR + 2;
	u64 (*const tcodes)[kSlots] = reinterpret_cast<u64(*)[kSlots]>(FULL ? (void*)S.tcodes_f : (void*)S.a);
	u32 (*const tvis)[kSlots] = reinterpret_cast<u32(*)[kSlots]>(FULL ? S.tvis_f : S.a + 4 * kSlots);
	u32 (*const tamb)[kSlots] = reinterpret_cast<u32(*)[kSlots]>(FULL ? S.tamb_f : S.a + 6 * kSlots);
	u32 (*const mm32)[kTW + 8] = reinterpret_cast<u32(*)[kTW + 8]>(FULL ? S.mm32_f : S.a + 8 * kSlots);
	u32 (*const town)[kSlots] = reinterpret_cast<u32(*)[kSlots]>(FULL ? S.town_f : S.b);
	static_assert(8 * kSlots + 2 * (kTW + 8) <= kTP, "aliases must stay clear of the pad of S.a");
	static_assert(2 * kSlots <= 128, "owners must stay clear of the per-read counters in S.b");
	const int lane_id = threadIdx.x;
	WaveStats ws = { 0, 0, 0, 0, 0, 0, 0, 0 };
	if (STATS && lane_id < 8)
		S.wstats[lane_id] = 0;
	const int k = g.k, w = bx.w;
	// the sliding minimum reads up to w - 1 + 7 positions past the tile: a constant pad
	if (!kDMed)
		for (int x = lane_id; x < 96; x += 64)
			S.a[kTP + x] = 0xFFFFFFFFu;
#ifdef ARKS_PROFILE_SECTIONS
	unsigned long long sec_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long sec_t0 = __builtin_amdgcn_s_memtime();
#endif

#ifdef ARKS_MEDIUM_DIAG
	unsigned long long md[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
	// F(windows) = the largest vote count that FAILS `(double)count / (double)windows > j_index` (Arcs.cpp:1006): the
	// quotient grows with the count, so `count > F` is that test, without a double-precision division per read -- the
	// two read lengths of a linked-read library are remembered
	int fc_n0 = -1, fc_f0 = 0, fc_n1 = -1, fc_f1 = 0;
	auto fail_count = [&](int nwin) -> int {
		if (nwin == fc_n0)
			return fc_f0;
		if (nwin == fc_n1)
			return fc_f1;
		int F = (int)(j_index * (double)nwin);
		F = F < 0 ? 0 : (F > nwin ? nwin : F);
		while (F < nwin && !((double)(F + 1) / (double)nwin > j_index))
			++F;
		while (F > 0 && (double)F / (double)nwin > j_index)
			--F;
		fc_n1 = fc_n0, fc_f1 = fc_f0;
		fc_n0 = nwin, fc_f0 = F;
		return F;
	};
	const u32 n_medium = FULL ? __hip_atomic_load(queue_count + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
	bool first_grab = true;
	u32* const work_ctr = reinterpret_cast<u32*>(reinterpret_cast<char*>(queue_count) + kWorkCtrOffset);
	u32 ctr = blockIdx.x & (u32)(kCounters - 1), misses = 0;
	for (;;) {
		long c0 = 0;
		ARKS_SEC(9);
		int nchunk;
		if (FULL) { // `grab` queued reads per grab: c0 = index of the first one in the medium queue
			// A tile holds as many of them as fit its 16 words (three or four 10x reads), gathered from wherever
			// they lie in the batch: the dependent round trips of a tile (words, seed probes, text along the
			// diagonals, fallback table) are shared by its reads -- one read per wave at a time made the medium
			// kernel the launch on a repeat-rich draft (12 ns per queued read; VERDICT r2 "what's weak" 7).  A short
			// queue is still spread one read per wave (latency, not throughput, bounds it then).
			// the first grab of a wave is the one with its block index: no atomic, and the (many) waves
			// beyond the queue length leave without touching the shared counter -- thousands of idle
			// waves queueing one atomic each on the same word cost 0.1 ms
			// (the grid size as an opaque value: `gridDim.x * grab` is otherwise an invariant of the loop that the
			// compiler forms once and, at five waves per SIMD, keeps in scratch memory)
			u32 n_blocks = gridDim.x;
			asm volatile("" : "+s"(n_blocks));
			const u32 per_wave = n_medium / n_blocks;
			const u32 grab = per_wave >= (u32)kGrabF ? (u32)kGrabF : (per_wave > 1u ? per_wave : 1u);
			u32 qi = blockIdx.x * grab;
			if (!first_grab) {
				if (lane_id == 0)
					qi = n_blocks * grab + atomicAdd(queue_count + 3, grab);
				qi = (u32)__builtin_amdgcn_readfirstlane((int)qi);
			}
			first_grab = false;
			if (qi >= n_medium)
				break;
			c0 = (long)qi;
			nchunk = (int)(n_medium - qi < grab ? n_medium - qi : grab);
		} else {
			// likewise the first chunk of a wave is the one with its block index (no start-up queue at
			// the counter: same-address atomics serialise at ~14 ns each); later chunks are handed out
			// dynamically from the end of that static round on
			// The chunks after that static round are dealt out by kCounters counters: counter j hands out
			// chunks j, j + kCounters, j + 2 kCounters, ... (so the waves still advance through the reads
			// as one front) and a wave uses counter (block index mod kCounters) until that one runs past
			// the end, then looks at the others.  One counter serialises at ~12 ns per grab -- 10 ms of
			// the 12.4 ms the kernel takes at C2.
			if (first_grab) {
				c0 = (long)blockIdx.x * kChunk;
				first_grab = false;
				if (c0 >= n_reads)
					break;
			} else {
				u32 cnt = 0;
				if (lane_id == 0)
					cnt = atomicAdd(work_ctr + ctr * kCounterStride, 1u);
				cnt = (u32)__builtin_amdgcn_readfirstlane((int)cnt);
				c0 = ((long)gridDim.x + (long)ctr + (long)cnt * kCounters) * kChunk;
				if (c0 >= n_reads) {
					ctr = (ctr + 1u) & (u32)(kCounters - 1);
					if (++misses == (u32)kCounters)
						break;
					continue;
				}
				misses = 0;
			}
			nchunk = (int)((c0 + kChunk < n_reads ? c0 + kChunk : n_reads) - c0);
		}
		// lane_id l holds the metadata of read c0 + l (lane_id nchunk: the end offset)
		u64 wo = 0;
		int rl = 0;
		u64 wreal = 0; // FULL: the read's first word in the batch (wo is then its first word in a virtual
		u32 rid = 0;   // concatenation of the grabbed reads); its number
		{
			// (the lane number as an opaque value: the addresses are formed per chunk, not kept -- and spilled to
			// scratch memory -- as invariants of the chunk loop; see map_reads_s_kernel)
			int cl = lane_id;
			asm volatile("" : "+v"(cl));
			if (FULL) {
				u64 nw = 0;
				if (cl < nchunk) {
					rid = mqueue[c0 + cl];
					wreal = word_off[rid];
					nw = word_off[(long)rid + 1] - wreal;
					rl = (int)lens[rid];
				}
				// wo = exclusive prefix of the reads' word counts (lane nchunk: their sum); nchunk <= kGrabF <= 16
				u64 incl = nw;
#pragma unroll
				for (int d = 1; d < 2 * kGrabF; d <<= 1) { // (lane nchunk <= kGrabF sums every read)
					const u64 o = __shfl_up(incl, d);
					incl += cl >= d ? o : 0;
				}
				wo = incl - nw;
			} else {
				if (cl <= nchunk)
					wo = word_off[c0 + cl];
				if (cl < nchunk) {
					rl = (int)lens[c0 + cl];
					if (eval && !eval[c0 + cl])
						rl = -1; // not evaluated: output 0, no counters
				}
			}
		}
		int cur = 0;
		while (cur < nchunk) {
			// re-derive the lane index inside the tile loop: everything computed from it is a couple of
			// VALU ops, cheaper to redo per tile than to keep (or spill) as loop invariants
			int lane = lane_id;
			asm volatile("" : "+v"(lane));
			// ---- tile = reads [cur, nxt): as many as fit --------------------------------------
			const u64 base_w = lane_value_u64(wo, cur);
			// first pass: reads [cur, mid) within kSW words; second pass: [mid, nxt) likewise
			// short sliding windows mean many runs per base: keep the tile's expected run count
			// (64 / (w + 1) per word) within kNH by capping its words (never below one read)
			// (seed index: one head per w windows and one more per read -- 7 w / 2 words stay within kNH)
			// (its medium kernel: kNHd seeds -- 3 w / 2 words of windows and sixteen reads)
			const u64 wcap = kDMed ? (u64)(3 * w / 2 < kTW ? (3 * w / 2 > 0 ? 3 * w / 2 : 1) : kTW)
			               : DENSE ? (u64)(7 * w / 2 < kTW ? 7 * w / 2 : kTW)
			                       : (u64)(7 * (w + 1) / 4 < kTW ? 7 * (w + 1) / 4 : kTW);
			const u64 fit = __ballot(
			    lane > cur && lane <= nchunk && wo - base_w <= (u64)kSW && lane - cur <= kTR &&
			    (lane == cur + 1 || wo - base_w <= wcap));
			if (fit == 0) { // a single read longer than a pass: slow kernel
				if (lane == cur) {
					const long r_one = FULL ? (long)rid : c0 + cur;
					if (rl >= 0)
						queue[atomicAdd(queue_count, 1u)] = (u32)r_one;
					else
						put_none<RAW>(out_conreci, r_one);
				}
				cur++;
				continue;
			}
			const int mid = 63 - __clzll((long long)fit);
			const u64 mid_w = lane_value_u64(wo, mid);
			int nxt = mid;
			if (!FULL) {
				const u64 fit2 = __ballot(
				    lane > mid && lane <= nchunk && wo - mid_w <= (u64)kSW && lane - cur <= kTR && wo - base_w <= wcap);
				if (fit2)
					nxt = 63 - __clzll((long long)fit2);
			}
			const int nr = nxt - cur;
			const int tw0 = (int)(mid_w - base_w);          // words of the first pass
			const int tw = (int)(lane_value_u64(wo, nxt) - base_w); // words of the tile
			const int n = tw * 32;
			ARKS_SEC(0);
			// ---- T0/T1: per-read metadata and the tile's words into LDS ------------------------
			if (lane >= cur && lane <= nxt)
				S.rstart[lane - cur] = (int)(wo - base_w) * 32;
			if (lane >= cur && lane < nxt) {
				S.rlen[lane - cur] = rl;
				if (FULL) {
					S.rbase[lane - cur] = wreal;
					S.rid[lane - cur] = rid;
				}
			}
			if (lane == 0) {
				u32 z; // (made here: as a loop-invariant constant pair the zero was spilled to scratch memory)
				asm volatile("v_mov_b32 %0, 0" : "=v"(z));
				S.redo = z;
				S.redo2 = z;
			}
			if (FULL && DENSE && lane < 2 * (kTP / 32 + 4))
				(&S.absent[0][0])[lane] = 0u;
			if (!FULL) { // per-read counters of T6c' (S.b + 128 ..: clear of the T3 block minima and the owners)
				S.b[128 + lane] = lane < 32 ? 0u : 0xFFFFFFFFu; // rcnt, rmin
				if (lane < 32)
					S.b[192 + lane] = 0u; // rmax
			}
			auto word_maps = [&]() { // per word of the tile: its read (lanes = reads)
				if (lane < nr) {
					const int w0 = S.rstart[lane] >> 5, w1 = S.rstart[lane + 1] >> 5;
					const int rend = S.rlen[lane] > 0 ? S.rstart[lane] + S.rlen[lane] : 0;
					for (int x = w0; x < w1; ++x) {
						S.wread[x] = (unsigned char)lane;
						S.wmeta[x] = ((u32)lane << 16) | (u32)rend;
					}
					for (int x = w0; x <= w1; ++x) // one staging slot more than the read has words
						S.sread[x + lane] = (unsigned char)lane;
				}
			};
			if (FULL) { // the tile's reads are gathered: word x lies where its read does
				ARKS_WAVE_SYNC();
				word_maps();
				ARKS_WAVE_SYNC();
			}
			{
				// both loads in flight before either is stored: left alone the compiler reuses one register
				// and serialises them -- two HBM round trips at the head of every tile instead of one
				u64 c_in = 0;
				u32 m_in = 0;
				if (lane < tw + 4) {
					u64 src_w = base_w + (u64)lane;
					if (FULL) { // (the four words past the tile: what follows the last read where that one lies)
						const int j = lane < tw ? (int)S.wread[lane] : nr - 1;
						src_w = S.rbase[j] + (u64)(lane - (S.rstart[j] >> 5));
					}
					c_in = codes[src_w];
					m_in = nmask[src_w];
				}
				asm volatile("" : "+v"(c_in), "+v"(m_in));
				if (lane < tw + 4) {
					S.cw[lane] = c_in;
					S.nm[lane] = m_in;
				}
			}
			ARKS_WAVE_SYNC();
			if (!FULL)
				word_maps();
			const bool has_n = __ballot(lane < tw && S.nm[lane] != 0) != 0;
			ARKS_WAVE_SYNC();
			ARKS_SEC(1);
			u32* src = S.a;
			u32* dst = S.b;
			// wmin[t] = minimizer of window i0 + t (bits [11:1]: its position); dst becomes the window
			// record: >= 0 value, -1 absent, -2 NULL window, -3 no window, <= -16 pending (q, run)
			int* rec = reinterpret_cast<int*>(dst);
			// hot instantiation: no window records; the same storage holds per-read counters instead, summed
			// by the word lanes of T6c': rcnt[j][d] = matched unambiguous windows on diagonal d | ambiguous
			// ones << 10 | (d = 0: windows that exist and hold no invalid base) << 20; rmin/rmax[j][d] =
			// smallest / largest contig end among the former (they differ iff more than one)
			u32* const rcnt = dst + 128; // [kTR][2]
			u32* const rmin = dst + 160; // [kTR][2]
			u32* const rmax = dst + 192; // [kTR][2]
			static_assert(kTR * 2 <= 32, "per-read counters");
			int nheads = 0;
			if (DENSE) {
				// ---- seeds: lane j = read j: groups of w windows, one seed each (the m-mer at the start of the
				//      group's last window); heads are numbered read-major ---------------------------------
				const u32 wrecip = (65536u + (u32)w - 1u) / (u32)w; // x / w == (x * wrecip) >> 16 for x < 65536 / w
				int G = 0, nwin = 0, rs = 0;
				if (lane < nr) {
					nwin = S.rlen[lane] - k + 1; // a read that is not evaluated has rlen < 0
					G = nwin > 0 ? (int)(((u32)(nwin + w - 1) * wrecip) >> 16) : 0;
					rs = S.rstart[lane];
				}
				int incl = G;
#define ARKS_ROW_ADD(v, ctrl) v += __builtin_amdgcn_update_dpp(0, (v), ctrl, 0xF, 0xF, true)
				ARKS_ROW_ADD(incl, 0x111); // row_shr:1 (kTR <= 16 reads: one row)
				ARKS_ROW_ADD(incl, 0x112);
				ARKS_ROW_ADD(incl, 0x114);
				ARKS_ROW_ADD(incl, 0x118);
				nheads = __builtin_amdgcn_readlane(incl, nr - 1);
				const int hb = incl - G;
				if (FULL && lane < nr)
					S.hbase[lane] = hb;
				for (int gi = 0; __ballot(gi < G) != 0; ++gi) {
					if (gi < G && hb + gi < kNHx) {
						int q = (gi + 1) * w - 1;
						q = q < nwin - 1 ? q : nwin - 1;
						S.heads[hb + gi] = (unsigned short)(((u32)(rs + q) << 1) | ((u32)lane << 12));
					}
				}
				if (FULL && kExtraSeeds > 0) {
					// Medium kernel, round 5: EXTRA seeds.  A read is here mostly because its group seeds are heavy or
					// have more than two entries (a read inside a copy of a repeat family): they propose no diagonal,
					// and without one every window under a heavy seed is an exact-key probe of the fallback table
					// (~70 per read).  The table holds EVERY m-mer position of the visited windows, so any other m-mer
					// of the read may propose the diagonal just as well -- and inside an old enough copy a third of
					// them have one or two entries.  Up to kExtraSeeds more m-mers per read, evenly spaced, are probed
					// in the same round trip as the group seeds (the lanes beyond the 6-12 group seeds of a tile were
					// idle); they only take part in the election of the diagonals (T6a), after the group seeds, and
					// the windows are verified base for base against the text as before: any diagonal is exact.
					// On the human-like draft four of five reads without a diagonal get one (profiles/tools/seedstat.py).
					int E = nheads < 64 ? (64 - nheads) / nr : 0;
					E = E > kExtraSeeds ? kExtraSeeds : E;
					const int Ej = (lane < nr && nwin > 0) ? E : 0;
					int xin = Ej;
					ARKS_ROW_ADD(xin, 0x111);
					ARKS_ROW_ADD(xin, 0x112);
					ARKS_ROW_ADD(xin, 0x114);
					ARKS_ROW_ADD(xin, 0x118);
					const int xtot = __builtin_amdgcn_readlane(xin, nr - 1);
					const int xb = nheads + xin - Ej;
					// offsets (2 e + 1) * (L - MM) / (2 E) into the read, in 16.16 fixed point
					const u32 step = Ej ? ((u32)(nwin - 1 + k - MM) << 16) / (u32)(2 * E) : 0u;
					for (int e = 0; e < E; ++e)
						if (e < Ej) {
							const int o = (int)(((u32)(2 * e + 1) * step) >> 16);
							S.heads[xb + e] = (unsigned short)(((u32)(rs + o) << 1) | ((u32)lane << 12));
						}
					nheads += xtot;
				}
#undef ARKS_ROW_ADD
				if (FULL) {
					// window records of the medium path: pending (seed position, seed strand, head)
					for (int base = 0; base < n; base += 64) {
						const int i = base + lane;
						// (n is a multiple of 32, not of 64: positions beyond the tile see stale metadata)
						const u32 wm = i < n ? S.wmeta[i >> 5] : 0u;
						const int j = (int)(wm >> 16);
						const int rem = (int)(wm & 0xFFFFu) - i;
						const bool is_win = i < n && rem >= k;
						bool bad = false;
						if (has_n && is_win) {
							const int tn = i & 31, e = tn + k;
							u32 any = 0;
#pragma unroll
							for (int x = 0; x <= KW; ++x) {
								int lo = tn - 32 * x, hi = e - 32 * x;
								lo = lo < 0 ? 0 : lo;
								hi = hi > 32 ? 32 : hi;
								if (lo < hi)
									any |= S.nm[(i >> 5) + x] & (0xFFFFFFFFu >> lo) & ~(hi == 32 ? 0u : (0xFFFFFFFFu >> hi));
							}
							bad = any != 0;
						}
						int rv = is_win ? -2 : -3;
						if (is_win && !bad) {
							const int p = i - S.rstart[j];
							const int nw = S.rlen[j] - k + 1;
							const int gi = (int)(((u32)p * wrecip) >> 16);
							int qr = (gi + 1) * w - 1;
							qr = qr < nw - 1 ? qr : nw - 1;
							const int q = S.rstart[j] + qr;
							const int hidx = S.hbase[j] + gi;
							// (the seed's strand is decided once per seed, in T5, and kept in S.heads: T6c reads it there)
							rv = -16 - (int)((u32)q | ((u32)hidx << 12));
							if (bx.has_img && !(k & 1)) { // see the minimizer path below: a necessary condition
								const int qm = 2 * i + (k - MM) - q;
								if (tile_canonical_mmer<MM>(S.cw, qm) == tile_canonical_mmer<MM>(S.cw, q))
									atomicOr(&S.redo, 1u << j);
							}
						}
						reinterpret_cast<int*>(S.b)[i] = rv;
					}
				} else if (bx.has_img && !(k & 1)) {
					// a reverse-complement palindrome carries its seed twice, mirrored about its centre: necessary
					// condition for a palindromic window (the slow kernel decides exactly), only looked for when
					// the index holds quirk images
					for (int base = 0; base < n; base += 64) {
						const int i = base + lane;
						const u32 wm = i < n ? S.wmeta[i >> 5] : 0u;
						const int j = (int)(wm >> 16);
						if (i < n && (int)(wm & 0xFFFFu) - i >= k) {
							const int p = i - S.rstart[j];
							const int nw = S.rlen[j] - k + 1;
							const int gi = (int)(((u32)p * wrecip) >> 16);
							int qr = (gi + 1) * w - 1;
							qr = qr < nw - 1 ? qr : nw - 1;
							const int q = S.rstart[j] + qr;
							const int qm = 2 * i + (k - MM) - q;
							if (tile_canonical_mmer<MM>(S.cw, qm) == tile_canonical_mmer<MM>(S.cw, q))
								atomicOr(&S.redo, 1u << j);
						}
					}
				}
			}
			// ---- T2 .. T4 once per pass of <= kSW words; the medium kernel's single read is one pass ---------
			const int npass = DENSE ? 0 : ((FULL || tw0 == tw) ? 1 : 2);
			for (int ps = 0; ps < npass; ++ps) {
				// ---- T2 / T3: lane l owns the 8 positions 8l .. 8l+7 of the pass (16 words = 512 positions) -------
				const int l0 = lane * 8;                      // index into the pass-local minimum arrays
				const int i0 = (ps ? tw0 * 32 : 0) + l0;      // tile position
				const bool in_tile = i0 < (ps ? n : tw0 * 32);
				const u32 wm0 = in_tile ? S.wmeta[i0 >> 5] : 0u; // the lane's 8 positions lie in one word, i.e. one read
				const int rem0 = in_tile ? (int)(wm0 & 0xFFFFu) - i0 : 0; // bases of the read from i0 on
				u32 wmin[8]; // minimizer (order value) of the window starting at each of the 8 positions
				{
					u32 v[8];
					tile_order_values<MM>(S.cw, S.nm, i0, lane, rem0, has_n, v);
					ARKS_SEC(2);
					tile_sliding_min(S.a, S.b, l0, lane, w, v, wmin);
				}
				ARKS_SEC(3);
				// ---- T4: windows, run heads --------------------------------------------------------------
				if (!FULL) {
					// hot path: only the run heads; window validity and values are worked out per 32-window
					// word further down
					// wc[t] = minimizer of window t when that window exists (and has any valid m-mer), else ~0
					const int tc = rem0 - k; // windows t <= tc exist
					u32 wc[8];
#pragma unroll
					for (int t = 0; t < 8; ++t)
						wc[t] = t <= tc ? wmin[t] : 0xFFFFFFFFu;
					// the previous lane's last window: one DPP wave shift (lane 0 keeps the "no window" value)
					const u32 wprev = (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)wc[7], 0x138, 0xF, 0xF, false);
					// a run head = a window whose minimizer (value includes its position) differs from its
					// predecessor's.  Heads are numbered position-class-major (all t = 0 heads of the pass,
					// then t = 1, ...): a ballot + mbcnt per class, no per-lane serial numbering.  Which run
					// comes first only decides which two diagonals get staged -- any choice is exact.
#pragma unroll
					for (int t = 0; t < 8; ++t) {
						const bool head = wc[t] != 0xFFFFFFFFu && wc[t] != (t ? wc[t - 1] : wprev);
						const u64 hb = __ballot(head);
						const u32 rank = (u32)nheads + mask_below(hb);
						if (head && rank < (u32)kNH)
							S.heads[rank] = (unsigned short)((wc[t] & 0xFFFu) | ((wm0 >> 16) << 12));
						nheads += __popcll(hb);
					}
					if (bx.has_img && !(k & 1)) { // see the comment in the FULL path below
#pragma unroll
						for (int t = 0; t < 8; ++t) {
							if (wc[t] != 0xFFFFFFFFu) {
								const int q = (int)((wc[t] >> 1) & 2047u);
								const int qm = 2 * (i0 + t) + (k - MM) - q;
								if (tile_canonical_mmer<MM>(S.cw, qm) == tile_canonical_mmer<MM>(S.cw, q))
									atomicOr(&S.redo, 1u << (wm0 >> 16));
							}
						}
					}
				} else {
					u32 carry = 0xFFFFu;
					ARKS_WAVE_SYNC();
					{
						uint4* pa = reinterpret_cast<uint4*>(S.a + l0);
						pa[0] = make_uint4(wmin[0], wmin[1], wmin[2], wmin[3]);
						pa[1] = make_uint4(wmin[4], wmin[5], wmin[6], wmin[7]);
					}
					ARKS_WAVE_SYNC();
					for (int base = 0; base < n; base += 64) {
						const int i = base + lane;
						const u32 wm = i < n ? S.wmeta[i >> 5] : 0u; // n is a multiple of 32, not of 64
						const int j = (int)(wm >> 16);
						const int rem = (int)(wm & 0xFFFFu) - i;
						const bool is_win = i < n && rem >= k;
						bool bad = false;
						if (has_n && is_win) {
							const int tn = i & 31, e = tn + k;
							u32 any = 0;
		#pragma unroll
							for (int x = 0; x <= KW; ++x) {
								int lo = tn - 32 * x, hi = e - 32 * x;
								lo = lo < 0 ? 0 : lo;
								hi = hi > 32 ? 32 : hi;
								if (lo < hi)
									any |= S.nm[(i >> 5) + x] & (0xFFFFFFFFu >> lo) & ~(hi == 32 ? 0u : (0xFFFFFFFFu >> hi));
							}
							bad = any != 0;
						}
						const bool ok = is_win && !bad;
						const u32 sv = src[i];
						const u32 q = ok ? ((sv >> 1) & 2047u) : 0xFFFFu;
						u32 qprev = __shfl_up(q, 1);
						if (lane == 0)
							qprev = carry;
						carry = __shfl(q, 63);
						const bool head = ok && q != qprev;
						const u64 hb = __ballot(head);
						const int hidx = nheads + (int)mask_below(hb) + (head ? 1 : 0) - 1; // run of this window
						if (head && hidx < kNH)
							S.heads[hidx] = (unsigned short)((sv & 0xFFFu) | ((u32)j << 12));
						nheads += __popcll(hb);
						int rv = is_win ? -2 : -3;
						if (ok) {
							rv = -16 - (int)(q | ((sv & 1u) << 11) | ((u32)hidx << 12)); // position, strand, run
							if (bx.has_img && !(k & 1)) {
								// Only when the index holds quirk images can a palindromic window have a key
								// that the text path would miss (otherwise it is either in the text, where its
								// position carries the value of its damaged key, or absent).  A reverse-
								// complement palindrome carries its minimizer twice, mirrored about its centre:
								// necessary condition; the slow kernel decides exactly.
								const int qm = 2 * i + (k - MM) - (int)q;
								if (tile_canonical_mmer<MM>(S.cw, qm) == tile_canonical_mmer<MM>(S.cw, (int)q))
									atomicOr(&S.redo, 1u << j);
							}
						}
						rec[i] = rv;
					}
				}
				if (!FULL)
					ARKS_WAVE_SYNC(); // S.a / S.b are rewritten by the next pass
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(4);
			if (FULL) {
				ARKS_MD(0, 1);
				ARKS_MD(1, nr);
				ARKS_MD(9, nheads);
			}
			// ---- T5: run heads walk the minimizer table ------------------------------------------------
			const int nh = nheads < kNHx ? nheads : kNHx;
			for (int h = lane; h < nh; h += 64) {
				const u32 q = ((u32)S.heads[h] >> 1) & 2047u;
				if (DENSE) {
					// the seed's strand is decided here; a seed that holds an invalid base has no entries (every
					// window of its group holds that base too: all NULL)
					const typename Mmer<MM>::type mf = tile_mmer<MM>(S.cw, (int)q), mr = mmer_rc<MM>(mf);
					u32 cnt = 0;
					if (!(has_n && tile_span_has_n(S.nm, (int)q, MM)))
						cnt = probe_minimizer_table<MM>(bx, mf < mr ? mf : mr, hc[h]);
					S.heads[h] = (unsigned short)(S.heads[h] | (mf < mr ? 1u : 0u));
					S.hn[h] = (unsigned char)cnt;
				} else {
					const u32 cnt = probe_minimizer_table<MM>(bx, tile_canonical_mmer<MM>(S.cw, (int)q), hc[h]);
					S.hn[h] = (unsigned char)cnt;
				}
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(5);
			// ---- T6a: up to two diagonals per read: A = entry 0 of its first run that has entries, B = the
			//      first entry (in run order) on another diagonal (a duplicated segment, a chance m-mer
			//      match).  Lanes = run heads; the "first" is an LDS atomic minimum keyed by the run index. --
			bool any_b = false; // some read of the tile has a second diagonal
			{
				constexpr u64 kDiagMask = (1ull << 42) - 1ull; // [39:0] D, [40] same strand, [41] valid
				// the diagonals run h proposes (0 = none), its read, and whether it must go the general way
				auto proposals = [&](int h, int& jh, u64& dk0, u64& dk1, bool& off) {
					const u32 cnt = S.hn[h];
					const u32 hv = S.heads[h];
					jh = (int)(hv >> 12);
					off = cnt == kHnHeavy || cnt == kHnOverflow;
					dk0 = 0, dk1 = 0;
					if (cnt >= 1 && cnt <= 2) {
						const int o = (int)((hv >> 1) & 2047u) - S.rstart[jh]; // offset of the minimizer in the read
						const u32 rstrand = hv & 1u;
						// same strand: read base x <-> text D + x ; opposite: read base x <-> text D - x
						const u64 e0 = hc[h][0];
						const bool s0 = ((u32)(e0 >> 62) & 1u) == rstrand;
						dk0 = (s0 ? (u64)(u32)e0 - (u64)o : (u64)(u32)e0 + (u64)(MM - 1 + o)) |
						      ((u64)s0 << 40) | (1ull << 41);
						if (cnt == 2) {
							const u64 e1 = hc[h][1];
							const bool s1 = ((u32)(e1 >> 62) & 1u) == rstrand;
							dk1 = (s1 ? (u64)(u32)e1 - (u64)o : (u64)(u32)e1 + (u64)(MM - 1 + o)) |
							      ((u64)s1 << 40) | (1ull << 41);
						}
					}
				};
				if (lane < 2 * nr)
					(&S.pdiag[0][0])[lane] = ~0ull;
				ARKS_WAVE_SYNC();
				// the first 64 runs keep their proposals in registers across the three phases; further
				// runs (short sliding windows only) recompute them
				int jh0 = 0;
				u64 p0 = 0, p1 = 0;
				bool off0 = false;
				if (lane < nh)
					proposals(lane, jh0, p0, p1, off0);
				auto elect_a = [&](int h, int jh, u64 dk0) {
					if (dk0)
						atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][0]),
						          (unsigned long long)(((u64)h << 42) | dk0));
				};
				auto elect_b = [&](int h, int jh, u64 dk0, u64 dk1) {
					if (dk0) {
						const u64 dA = S.pdiag[jh][0] & kDiagMask;
						if (dk0 != dA)
							atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][1]),
							          (unsigned long long)(((u64)h << 43) | dk0));
						else if (dk1 && dk1 != dA)
							atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][1]),
							          (unsigned long long)(((u64)h << 43) | (1ull << 42) | dk1));
					}
				};
				auto flag = [&](int jh, u64 dk0, u64 dk1, bool off) {
					if (dk0) {
						const u64 dA = S.pdiag[jh][0] & kDiagMask;
						const u64 rb = S.pdiag[jh][1];
						const u64 dB = rb == ~0ull ? 0ull : (rb & kDiagMask);
						off = off || (dk0 != dA && dk0 != dB) || (dk1 && dk1 != dA && dk1 != dB);
					}
					if (off)
						atomicOr(&S.redo2, 1u << jh);
				};
				elect_a(lane, jh0, p0);
				for (int h = lane + 64; h < nh; h += 64) {
					int jh;
					u64 dk0, dk1;
					bool off;
					proposals(h, jh, dk0, dk1, off);
					elect_a(h, jh, dk0);
				}
				ARKS_WAVE_SYNC();
				elect_b(lane, jh0, p0, p1);
				for (int h = lane + 64; h < nh; h += 64) {
					int jh;
					u64 dk0, dk1;
					bool off;
					proposals(h, jh, dk0, dk1, off);
					elect_b(h, jh, dk0, dk1);
				}
				ARKS_WAVE_SYNC();
				if (!FULL) {
					// hot path: a read with a run that proposes a third diagonal, sits under a heavy
					// minimizer or had more than two entries goes to the medium queue as a whole
					if (lane < nh)
						flag(jh0, p0, p1, off0);
					for (int h = lane + 64; h < nh; h += 64) {
						int jh;
						u64 dk0, dk1;
						bool off;
						proposals(h, jh, dk0, dk1, off);
						flag(jh, dk0, dk1, off);
					}
					if (nheads > kNHx && lane == 0) // more runs than the tile publishes: every read
						atomicOr(&S.redo2, 0xFFFFFFFFu);
					ARKS_WAVE_SYNC();
				}
				bool has_b = false;
				if (lane < nr) {
					// strip the run index; first text word of the span [lo, lo + L) the read covers
#pragma unroll
					for (int d = 0; d < 2; ++d) {
						const u64 raw = S.pdiag[lane][d];
						const u64 pdv = raw == ~0ull ? 0ull : (raw & kDiagMask);
						S.pdiag[lane][d] = pdv;
						const u64 Dv = pdv & 0xFFFFFFFFFFull;
						const u64 lo = ((pdv >> 40) & 1ull) ? Dv : Dv - (u64)(S.rlen[lane] - 1);
						S.tfirst[lane][d] = (u32)(lo >> 5);
						has_b = d == 1 && pdv != 0;
					}
				}
				any_b = __ballot(has_b) != 0;
				if (FULL) {
					ARKS_MD(2, __popcll(__ballot(lane < nr && (S.pdiag[lane][0] >> 41))));
					ARKS_MD(11, any_b ? 1 : 0);
				}
				if (FULL && DENSE) {
					// Proofs of absence (round 5).  The seed table holds EVERY m-mer position of the visited windows (a
					// heavy m-mer as a marker), so a window one of whose m-mers has NO entry is in no visited window of
					// the text -- the rule T6c applies to a group's own seed holds for every seed probed in T5, the
					// extra ones included.  Half of the reads that come here were drawn from a copy of a repeat family
					// that lies OUTSIDE the indexed contig ends: every seed heavy or empty, no diagonal, none of their
					// windows in the index -- each of them was an exact-key probe of the fallback table (~60 per read);
					// now nine in ten of those windows hold an empty seed.  A seed with one or two entries proves the
					// same for the windows that do not match the text there, when both entries lie on staged diagonals.
					ARKS_WAVE_SYNC(); // (the stripped diagonals)
					for (int h = lane; h < nh; h += 64) {
						const u32 cnt = S.hn[h];
						const u32 hv = S.heads[h];
						const int jh = (int)(hv >> 12), q = (int)((hv >> 1) & 2047u);
						bool proves = cnt == 0;
						if (cnt == 1 || cnt == 2) {
							int jx;
							u64 dk0, dk1;
							bool off;
							proposals(h, jx, dk0, dk1, off);
							const u64 dA = S.pdiag[jh][0], dB = S.pdiag[jh][1];
							proves = (dk0 == dA || dk0 == dB) && (cnt == 1 || dk1 == dA || dk1 == dB);
						}
						if (proves) { // windows [a, b] of the tile hold the seed
							const int rs = S.rstart[jh], rend = rs + S.rlen[jh];
							int a = q - (k - MM), b = q;
							a = a > rs ? a : rs;
							b = b < rend - k ? b : rend - k;
							for (int x = a >> 5; x <= (b >> 5); ++x) {
								const int lo = a > 32 * x ? a - 32 * x : 0, hi = b < 32 * x + 31 ? b - 32 * x : 31;
								atomicOr(&S.absent[cnt ? 1 : 0][x], (0xFFFFFFFFu >> (31 - hi)) & (0xFFFFFFFFu << lo));
							}
						}
					}
				}
			}
			ARKS_WAVE_SYNC();
			// stage the text words and their visited / ambiguous / owner words (one round trip): one lane
			// per slot of the first diagonals; second diagonals are rare -- their pass only runs for a tile
			// that has one (lanes = slots x 2 needed a second, almost empty trip through this code)
			{
				const int ns = tw + nr;
				if (FULL) {
					// medium kernel: <= kSW words + kTR reads = 32 slots per diagonal -- both diagonals in ONE round trip,
					// lanes = slots x 2 (with the extra seeds of round 5 three tiles in four have a second diagonal; as two
					// passes that was a dependent round trip per tile)
					static_assert(kSW + kTR <= 32, "staging slots of a medium tile");
					const int d = lane >> 5, sl = lane & 31;
					if (sl < ns && (d == 0 || any_b)) {
						const int j = S.sread[sl];
						if (S.pdiag[j][d] >> 41) {
							const u64 tw_idx = (u64)S.tfirst[j][d] + (u64)(sl - ((S.rstart[j] >> 5) + j));
							tcodes[d][sl] = bx.codes[tw_idx];
							tvis[d][sl] = bx.visited[tw_idx];
							tamb[d][sl] = bx.ambig[tw_idx];
							town[d][sl] = bx.word_owner[tw_idx];
						}
					}
				} else
				for (int d = 0; d < (any_b ? 2 : 1); ++d)
					for (int sl = lane; sl < ns; sl += 64) {
						const int j = S.sread[sl];
						if (S.pdiag[j][d] >> 41) {
							const u64 tw_idx = (u64)S.tfirst[j][d] + (u64)(sl - ((S.rstart[j] >> 5) + j));
							tcodes[d][sl] = bx.codes[tw_idx];
							tvis[d][sl] = bx.visited[tw_idx];
							tamb[d][sl] = bx.ambig[tw_idx];
							town[d][sl] = bx.word_owner[tw_idx];
						}
					}
			}
			ARKS_WAVE_SYNC();
			// ---- T6b: lanes = read words (x2 diagonals): XOR each word of the read with the 32 text bases
			//      it faces -> one mismatch bit per base (bit b of mm32[d][word]) -------------------------
			{
				const int d = lane >= 32 ? 1 : 0, wl = lane & 31;
				if (lane < 16) // the spans of the last words read past the tile: no mismatch there
					mm32[lane >> 3][tw + (lane & 7)] = 0u;
				if (wl < tw) {
					const int j = S.wread[wl];
					const u64 pdv = S.pdiag[j][d];
					const u32 mbits = (pdv >> 41) ? word_mismatch_bits(S.cw[wl], pdv, tcodes[d], wl * 32 - S.rstart[j],
					                                                    (S.rstart[j] >> 5) + j, S.tfirst[j][d])
					                              : 0u;
					mm32[d][wl] = mbits;
				}
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(6);
			if (!FULL) {
				// ---- T6c': lanes = 32-window words (x2 diagonals).  Window p of the tile matches the text on
				//      diagonal d iff its k mismatch bits are all clear (AND over a sliding span, done on the
				//      bit stream); it is in the index iff the text position it maps to is `visited`; its
				//      value is 0 iff that position is `ambig`.  Everything stays a bit per window.
				const int d = lane >= 32 ? 1 : 0, wl = lane & 31;
				u32 ok = 0, amb = 0, own = 0, valid = 0;
				int j = 0;
				if (wl < tw) {
					j = S.wread[wl];
					valid = word_valid_windows(S.nm, wl, S.rlen[j] - k + 1 - (wl * 32 - S.rstart[j]), k, has_n);
					if ((S.pdiag[j][d] >> 41) && valid)
						word_match(S.pdiag[j][d], mm32[d], tvis[d], tamb[d], town[d], wl, wl * 32 - S.rstart[j],
						           (S.rstart[j] >> 5) + j, S.tfirst[j][d], k, valid, ok, amb, own);
				}
				{
					// a window matched on both diagonals counts once (on A; the value is the same)
					// the first-diagonal lane of the same word is 32 lanes down: v_permlane32_swap hands the upper
					// half of the wave the lower half's values (one VALU op; __shfl_xor is an LDS crossbar trip)
					const u32 other = __builtin_amdgcn_permlane32_swap(ok, ok, false, false)[0];
					if (d) {
						ok &= ~other;
						amb &= ok;
					}
					const u32 recm = ok & ~amb;
					const u32 pack = (u32)__popc(recm) | ((u32)__popc(amb) << 10) | (d ? 0u : ((u32)__popc(valid) << 20));
					if (pack)
						atomicAdd(&rcnt[j * 2 + d], pack);
					if (recm) {
						atomicMin(&rmin[j * 2 + d], own == 0xFFFFFFFFu ? 1u : own); // mixed word: min != max
						atomicMax(&rmax[j * 2 + d], own);
					}
				}
				ARKS_WAVE_SYNC();
			}
			u32 failmask = 0; // medium kernel without counters: reads whose vote fails whatever their open windows hold (T6d)
			// medium kernel, seed index: the match of every 32-window word on both diagonals, one bit per window, as the
			// hot kernels work it out (lanes = words x 2 diagonals, one pass) -- T6c then TESTS A BIT per window where it
			// shifted four words of mismatch bits per window and diagonal (a fifth of the kernel's time on the human-like
			// draft).  In the pad of S.a, which only the minimizer index's sliding minimum reads.
			constexpr int kMaskAt = kDMed ? 4 * kNHd : kTP; // (behind the entry lists)
			u32* const wok = S.a + kMaskAt;       // [2][kSW] windows that face a visited text window, base for base
			u32* const wamb = S.a + kMaskAt + 32; // ... whose key is ambiguous (value 0)
			u32* const wown = S.a + kMaskAt + 64; // contig end of the others, ~0 when the word's windows span two
			static_assert(2 * kSW <= 32, "word masks of a medium tile");
			if (FULL && DENSE) {
				const int d = lane >= 32 ? 1 : 0, wl = lane & 31;
				if (wl < kSW) {
					u32 ok = 0, amb = 0, own = 0;
					if (wl < tw) {
						const int j = S.wread[wl];
						if (S.pdiag[j][d] >> 41) {
							const u32 valid = word_valid_windows(S.nm, wl, S.rlen[j] - k + 1 - (wl * 32 - S.rstart[j]), k, has_n);
							if (valid)
								word_match(S.pdiag[j][d], mm32[d], tvis[d], tamb[d], town[d], wl, wl * 32 - S.rstart[j],
								           (S.rstart[j] >> 5) + j, S.tfirst[j][d], k, valid, ok, amb, own);
						}
					}
					wok[d * kSW + wl] = ok;
					wamb[d * kSW + wl] = amb;
					wown[d * kSW + wl] = own;
				}
				ARKS_WAVE_SYNC();
			}
			if (FULL) {
			// ---- T6c: lanes = windows: an entry on one of the read's two staged diagonals only tests the
			//      window's k mismatch bits; anything else needs the general verification ----------------
			for (int base = 0; base < n; base += 64) {
				const int i = base + lane;
				const int rv = rec[i];
				bool pending = rv <= -16;
				if (DENSE && pending && ((S.absent[0][i >> 5] >> (i & 31)) & 1u)) {
					rec[i] = -1; // holds a seed that is in no visited window
					pending = false;
					ARKS_MD(6, 0x100000000ull);
				}
				if (__ballot(pending) == 0)
					continue;
				if (pending) {
					const u32 pay = (u32)(-16 - rv);
					const int q = (int)(pay & 2047u), hidx = (int)(pay >> 12);
					const u32 rstrand = DENSE ? (hidx < kNHx ? (u32)S.heads[hidx] & 1u : 0u) : (pay >> 11) & 1u;
					const u32 hn = hidx < kNHx ? S.hn[hidx] : kHnOverflow;
					int val = -1;
					bool full = hn == kHnHeavy || hn == kHnOverflow;
					if (full && FULL) {
						// the window's own seed says nothing (heavy, or more than two entries) -- but the read's staged
						// diagonals come from its OTHER seeds: a window that faces an indexed text window there, base for
						// base, has that window's key, and the key's value is what the exact-key probe would return.  Only
						// the windows that match on neither diagonal go to the fallback table (a read inside a repeat copy
						// sent ~70 windows there; an error-free one sends none now).
						const int j = S.wread[i >> 5];
						const int p = i - S.rstart[j];
						const int l = i & 31;
						for (int d = 0; d < 2 && val < 0; ++d) {
							if (DENSE) { // (the word masks above)
								const int wx = d * kSW + (i >> 5);
								if ((wok[wx] >> l) & 1u) {
									const u32 own = wown[wx];
									if ((wamb[wx] >> l) & 1u)
										val = 0;
									else if (own != 0xFFFFFFFFu)
										val = (int)own;
									else { // the word's windows span two contig ends: this window's text position says which
										const u64 dk = S.pdiag[j][d];
										const u64 D = dk & 0xFFFFFFFFFFull;
										const u64 t = ((dk >> 40) & 1ull) ? D + (u64)p : D - (u64)(p + k - 1);
										val = (int)town[d][(S.rstart[j] >> 5) + j + (int)((u32)(t >> 5) - S.tfirst[j][d])];
									}
								}
								continue;
							}
							const u64 dk = S.pdiag[j][d];
							if (!(dk >> 41))
								continue;
							const bool same = ((dk >> 40) & 1ull) != 0;
							const u64 D = dk & 0xFFFFFFFFFFull;
							const u32* mw = mm32[d] + (i >> 5);
							const u64 m01 = (u64)mw[0] | ((u64)mw[1] << 32);
							const u64 m23 = (u64)mw[2] | ((u64)mw[3] << 32);
							u64 bits = funnel_r(m01, m23, l); // 64 bases from i on
							if (k < 64)
								bits &= (1ull << k) - 1ull;
							if (KW > 2 && k > 64) {
								const u64 m45 = (u64)mw[4] | ((u64)mw[5] << 32);
								bits |= funnel_r(m23, m45, l) & ((1ull << (k - 64)) - 1ull);
							}
							if (bits == 0) {
								const u64 t = same ? D + (u64)p : D - (u64)(p + k - 1);
								const int slot = (S.rstart[j] >> 5) + j + (int)((u32)(t >> 5) - S.tfirst[j][d]);
								const u32 sh = 31 - (u32)(t & 31);
								if ((tvis[d][slot] >> sh) & 1u)
									val = ((tamb[d][slot] >> sh) & 1u) ? 0 : (int)town[d][slot];
							}
						}
						if (val >= 0)
							full = false;
						else if (DENSE && ((S.absent[1][i >> 5] >> (i & 31)) & 1u)) {
							full = false; // one of its m-mers occurs on the staged diagonals only, and the window did not match there
							ARKS_MD(6, 1);
						}
						// (Tried on top, for the windows still open: one m-mer of the window, laid over its first mismatch on the
						// staged diagonal, through the seed table -- no entry proves the window absent without an exact-key
						// probe.  Correct, and 6.2 instead of 5.9 ms per 100 M pairs on the repeat-rich draft: inside a repeat
						// copy the m-mers with one more mutation are mostly heavy themselves, and the extra round trip is paid
						// by every tile.  profiles/r05_repeats.txt)
					}
					if (hn == 1 || hn == 2) {
						const int j = S.wread[i >> 5];
						const int p = i - S.rstart[j];
						const int o = q - S.rstart[j];
						const int l = i & 31;
						for (u32 c = 0; c < hn && val < 0; ++c) {
							const u64 e = hc[hidx][c];
							const bool same = ((u32)(e >> 62) & 1u) == rstrand;
							const u64 D = same ? (u64)(u32)e - (u64)o : (u64)(u32)e + (u64)(MM - 1 + o);
							const u64 dk = D | ((u64)same << 40) | (1ull << 41);
							int d = -1;
							d = dk == S.pdiag[j][1] ? 1 : d;
							d = dk == S.pdiag[j][0] ? 0 : d;
							if (d < 0) {
								full = true;
								continue;
							}
							if (DENSE) { // (the word masks above)
								const int wx = d * kSW + (i >> 5);
								if ((wok[wx] >> l) & 1u) {
									const u32 own = wown[wx];
									if ((wamb[wx] >> l) & 1u)
										val = 0;
									else if (own != 0xFFFFFFFFu)
										val = (int)own;
									else {
										const u64 t = same ? D + (u64)p : D - (u64)(p + k - 1);
										val = (int)town[d][(S.rstart[j] >> 5) + j + (int)((u32)(t >> 5) - S.tfirst[j][d])];
									}
								}
								continue;
							}
							const u32* mw = mm32[d] + (i >> 5);
							const u64 m01 = (u64)mw[0] | ((u64)mw[1] << 32);
							const u64 m23 = (u64)mw[2] | ((u64)mw[3] << 32);
							u64 bits = funnel_r(m01, m23, l); // 64 bases from i on
							if (k < 64)
								bits &= (1ull << k) - 1ull;
							if (KW > 2 && k > 64) {
								const u64 m45 = (u64)mw[4] | ((u64)mw[5] << 32);
								bits |= funnel_r(m23, m45, l) & ((1ull << (k - 64)) - 1ull);
							}
							if (bits == 0) {
								const u64 t = same ? D + (u64)p : D - (u64)(p + k - 1);
								const int slot = (S.rstart[j] >> 5) + j + (int)((u32)(t >> 5) - S.tfirst[j][d]);
								const u32 sh = 31 - (u32)(t & 31);
								if ((tvis[d][slot] >> sh) & 1u)
									val = ((tamb[d][slot] >> sh) & 1u) ? 0 : (int)town[d][slot];
							}
						}
						if (val >= 0)
							full = false;
					}
					if (full && !FULL)
						atomicOr(&S.redo2, 1u << S.wread[i >> 5]);
					if (full && FULL) {
						if (hn == kHnHeavy) {
							// an exact-key probe of the fallback table: not here, where every batch of 64 windows
							// would wait for its own probes -- the windows are listed and probed together below
							val = kRecFallback;
						} else {
							// the seed has more than two entries (only fingerprint collisions since round 5; 3-8 occurrences with
							// ARKS_HEAVY_OVER=8), or one or two of which one lies on a diagonal that is not staged: the walk over
							// the entries and the text behind each -- a chain of dependent global reads.  Not here, where the
							// other 63 positions of the batch would wait for it (a third of the kernel's time on the human-like
							// draft when it was), but with the exact-key probes of T6d, all in flight together, and only for the
							// windows the vote still needs
							val = kRecOverflow;
						}
					}
					rec[i] = val;
				}
#ifdef ARKS_MEDIUM_DIAG
				ARKS_MD(3, __popcll(__ballot(pending && rec[i] == kRecFallback)));
				// ... of them in reads without a diagonal; in reads whose diagonal A differs from the read in > 8 bases
				ARKS_MD(4, __popcll(__ballot(pending && rec[i] == kRecFallback && !(S.pdiag[S.wread[i >> 5]][0] >> 41))));
				{
					bool far = false;
					if (pending && rec[i] == kRecFallback && (S.pdiag[S.wread[i >> 5]][0] >> 41)) {
						const int j = S.wread[i >> 5];
						int dif = 0;
						for (int x = S.rstart[j] >> 5; x < (S.rstart[j + 1] >> 5); ++x) {
							const int nb = S.rstart[j] + S.rlen[j] - 32 * x;
							dif += __popc(mm32[0][x] & (nb >= 32 ? 0xFFFFFFFFu : (nb > 0 ? (1u << nb) - 1u : 0u)));
						}
						far = dif > 8;
					}
					ARKS_MD(5, __popcll(__ballot(far)));
				}
#endif
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(10);
			// ---- T6d: the windows under heavy seeds: exact keys into the fallback table.  Their positions are
			//      compacted (ballot + mbcnt) into the storage of the staged text, which is dead by now, and every
			//      lane keeps TWO probes in flight per round trip (a repeat-rich draft sends ~70 windows of a read
			//      here: eight dependent probe rounds per tile before, two or three now) -------------------------
			{
				unsigned short* const flist = reinterpret_cast<unsigned short*>(S.tcodes_f);
				// (tcodes_f, tvis_f, tamb_f, town_f lie one behind the other: 2000 bytes for <= kTP positions)
				typedef TileLds<FULL, kDMed> Lds_t;
				static_assert(!FULL || offsetof(Lds_t, mm32_f) - offsetof(Lds_t, tcodes_f) >=
				                           sizeof(unsigned short) * (size_t)kTP, "window list");
				static_assert(kTP <= 0x8000, "a list entry: position | entry walk << 15");
				// the probes of a list of nlist windows
				auto probe_list = [&](int nlist) {
					for (int base = 0; base < nlist; base += 128) {
						int wi[2];
						bool act[2], ov[2];
						Key<KW> c[2];
						u64 sl[2];
						int val[2] = { -1, -1 };
#pragma unroll
						for (int u = 0; u < 2; ++u) {
							const int e = base + 64 * u + lane;
							const u32 raw = e < nlist ? (u32)flist[e] : 0u;
							ov[u] = (raw >> 15) != 0; // an entry walk, below
							act[u] = e < nlist && !ov[u];
							wi[u] = (int)(raw & 0x7FFFu);
							const Key<KW> f = tile_window_key<KW>(S.cw, wi[u], g);
							const Key<KW> r = key_revcomp(f, g);
							const bool lt = key_less(f, r);
#pragma unroll
							for (int x = 0; x < KW; ++x)
								c[u].w[x] = lt ? f.w[x] : r.w[x];
							if (key_eq(f, r)) // a palindrome lives in the fallback table under its damaged key
								c[u] = key_palindrome_quirk(f, g);
							sl[u] = mulhi64(key_hash(c[u]), bx.fallback.cap);
						}
						while (__ballot(act[0] || act[1]) != 0) {
							ARKS_MD(8, 1);
							ARKS_MD(12, __popcll(__ballot(act[0])) + __popcll(__ballot(act[1])));
							u64 w[2][kSlotWords];
#pragma unroll
							for (int u = 0; u < 2; ++u)
								if (act[u]) {
									const u64* slot = bx.fallback.slots + sl[u] * kSlotWords;
#pragma unroll
									for (int x = 0; x < kSlotWords; ++x)
										w[u][x] = slot[x];
								}
							asm volatile("" : "+v"(w[0][0]), "+v"(w[1][0])); // both slots requested before either is looked at
#pragma unroll
							for (int u = 0; u < 2; ++u)
								if (act[u]) {
									const u32 st = (u32)w[u][3];
									bool eq = true;
#pragma unroll
									for (int x = 0; x < KW; ++x)
										eq = eq && w[u][x] == c[u].w[x];
									if (st == kEmpty)
										act[u] = false;
									else if (eq) {
										val[u] = (int)(st - 1u);
										act[u] = false;
									} else
										sl[u] = (sl[u] + 1 == bx.fallback.cap) ? 0 : sl[u] + 1;
								}
						}
#pragma unroll
						for (int u = 0; u < 2; ++u)
							if (ov[u]) { // (the window's place in the batch's packed arrays: its read's, not the tile's)
								const Key<KW> f = tile_window_key<KW>(S.cw, wi[u], g);
								const Key<KW> r = key_revcomp(f, g);
								const int jr = S.wread[wi[u] >> 5];
								val[u] = bindex_lookup_serial<KW, MM>(bx, g, codes, S.rbase[jr] * 32ull + (u64)(wi[u] - S.rstart[jr]), f, r);
							}
#pragma unroll
						for (int u = 0; u < 2; ++u)
							if (base + 64 * u + lane < nlist)
								rec[wi[u]] = val[u];
					}
				};
				// Without counters and with the vote's result as the only output, the windows need not ALL be looked up:
				// a read FAILS the vote (output 0, Arcs.cpp:1006-1010) as soon as the windows that can still vote are too
				// few -- no contig end can collect more than C + P votes (C = windows with a contig end so far, P = open
				// windows), and F = the largest count with F / windows <= j is known.  So only C + P - F of a read's
				// open windows (+ a margin of kSettleMargin) are probed first; if enough of them come back absent or
				// ambiguous the rest is never looked up, else it is, in a second round.  The reads this pays for are
				// the ones every seed of which is heavy (young copies of a repeat family, in the index or not): no
				// diagonal, no proof of absence, ~50 open windows of ~80, of which 7-10 settle the vote -- two of three
				// exact-key probes of the human-like draft.  With counters (-v) every window is looked up as before.
				constexpr bool kSettle = FULL && !STATS && !RAW;
				int nlist = 0;
				u32 deferred = 0; // reads with open windows left for the second round
				if (!kSettle) {
					for (int base = 0; base < n; base += 64) {
						const int i = base + lane;
						const int rv = i < n ? rec[i] : -3;
						const bool need = rv == kRecFallback || rv == kRecOverflow;
						const u64 nb = __ballot(need);
						ARKS_MD(7, __popcll(nb));
						if (need)
							flist[nlist + (int)mask_below(nb)] = (unsigned short)((u32)i | (rv == kRecOverflow ? 0x8000u : 0u));
						nlist += __popcll(nb);
					}
				} else {
					for (int j = 0; j < nr; ++j) {
						const int nwin = S.rlen[j] - k + 1, p0 = S.rstart[j];
						if (nwin <= 0)
							continue;
						int C = 0, P = 0;
						for (int base = 0; base < nwin; base += 64) {
							const int p = base + lane;
							const int v = p < nwin ? rec[p0 + p] : -3;
							C += __popcll(__ballot(v > 0));
							P += __popcll(__ballot(v == kRecFallback || v == kRecOverflow));
						}
						if (P == 0)
							continue;
						const int F = fail_count(nwin);
						const int s = ((S.redo >> j) & 1u) ? 0 : C + P - F; // (a read for the slow queue is decided there)
						int take = s <= 0 ? 0 : s + kSettleMargin;
						take = take > P ? P : take;
						if (take > 0 && take < P) {
							deferred |= 1u << j;
							S.vfail[j] = F;
						}
						ARKS_MD(7, take);
						if (s <= 0) {
							failmask |= 1u << j; // settled: whatever the open windows hold, the read fails (T7 writes its 0)
							ARKS_MD(13, 1);
							continue;
						}
						ARKS_MD(14, take < P ? 1 : 0);
						int seen = 0;
						for (int base = 0; base < nwin && seen < take; base += 64) { // the first `take` open windows
							const int p = base + lane;
							const int rv = p < nwin ? rec[p0 + p] : -3;
							const bool need = rv == kRecFallback || rv == kRecOverflow;
							const u64 nb = __ballot(need);
							const int rank = seen + (int)mask_below(nb);
							if (need && rank < take)
								flist[nlist + rank] = (unsigned short)((u32)(p0 + p) | (rv == kRecOverflow ? 0x8000u : 0u));
							seen += __popcll(nb);
						}
						nlist += take;
					}
				}
				ARKS_WAVE_SYNC();
				ARKS_SEC(11);
				probe_list(nlist);
				if (kSettle && deferred) {
					ARKS_WAVE_SYNC();
					nlist = 0;
					for (int j = 0; j < nr; ++j) {
						if (!((deferred >> j) & 1u))
							continue;
						const int nwin = S.rlen[j] - k + 1, p0 = S.rstart[j];
						int C = 0, P = 0;
						for (int base = 0; base < nwin; base += 64) {
							const int p = base + lane;
							const int v = p < nwin ? rec[p0 + p] : -3;
							C += __popcll(__ballot(v > 0));
							P += __popcll(__ballot(v == kRecFallback || v == kRecOverflow));
						}
						const bool settled = C + P <= S.vfail[j];
						ARKS_MD(10, settled ? 1 : 0);
						ARKS_MD(7, settled ? 0 : P);
						if (settled) {
							failmask |= 1u << j;
							continue;
						}
						for (int base = 0; base < nwin; base += 64) {
							const int p = base + lane;
							const int rv = p < nwin ? rec[p0 + p] : -3;
							const bool need = rv == kRecFallback || rv == kRecOverflow;
							const u64 nb = __ballot(need);
							if (need)
								flist[nlist + (int)mask_below(nb)] = (unsigned short)((u32)(p0 + p) | (rv == kRecOverflow ? 0x8000u : 0u));
							nlist += __popcll(nb);
						}
					}
					ARKS_WAVE_SYNC();
					probe_list(nlist);
				}
			}
			ARKS_WAVE_SYNC();
			}
			ARKS_SEC(7);
			// ---- T7: per read: counters, vote, output ---------------------------------------------------
			const u32 redo_mask = S.redo;
			const u32 redo2_mask = FULL ? 0u : S.redo2;
			if (!FULL) {
				// hot path: lane j = read j; popcounts over its window words.  At most two distinct
				// positive values can occur (one per diagonal): the vote of Arcs.cpp:998-1004 is a compare.
				u32 st_a = 0, st_b = 0, st_c = 0; // this read's share of the counters (hot-finished reads only)
				if (lane < nr) {
					const int j = lane;
					const long r = c0 + cur + j;
					const int L = S.rlen[j];
					if (L < 0) {
						put_none<RAW>(out_conreci, r);
					} else if ((redo_mask >> j) & 1u) {
						queue[atomicAdd(queue_count, 1u)] = (u32)r;
					} else {
						bool medium = (redo2_mask >> j) & 1u;
						const u32 ca = rcnt[j * 2], cb = rcnt[j * 2 + 1];
						const int rec_a = (int)(ca & 1023u), amb_a = (int)((ca >> 10) & 1023u);
						const int rec_b = (int)(cb & 1023u), amb_b = (int)((cb >> 10) & 1023u);
						const int nvalid = (int)(ca >> 20);
						const u32 own_a = rec_a ? rmin[j * 2] : 0u, own_b = rec_b ? rmin[j * 2 + 1] : 0u;
						// matches on one diagonal that belong to different contig ends: general path
						medium = medium || (rec_a && rmax[j * 2] != own_a) || (rec_b && rmax[j * 2 + 1] != own_b);
						if (medium) {
							mqueue[atomicAdd(queue_count + 2, 1u)] = (u32)r;
						} else {
							int best = 0, best_cnt = 0;
							if (rec_a > 0 && rec_b > 0 && own_a == own_b) {
								best = (int)own_a;
								best_cnt = rec_a + rec_b;
							} else if (rec_a > rec_b || (rec_a == rec_b && own_a < own_b)) {
								best = (int)own_a; // strict `>` of the reference walk: the smaller value keeps a tie
								best_cnt = rec_a;
							} else {
								best = (int)own_b;
								best_cnt = rec_b;
							}
							const int nwin = L - k + 1;
							const int total = nwin > 0 ? nwin : 0;
							const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
							const bool pass = maxj > j_index;
							put_result<RAW>(out_conreci, r, best, best_cnt, pass);
							if (STATS) { // <= 16 reads x <= 512 windows: three words of 16-bit (8-bit) fields
								st_a = (u32)nvalid | ((u32)total << 16);
								st_b = (u32)(rec_a + amb_a + rec_b + amb_b) | ((u32)(rec_a + rec_b) << 16);
								st_c = (u32)(amb_a + amb_b) | (pass ? 1u << 16 : 1u << 24);
							}
						}
					}
				}
				if (STATS) {
					// sum over the (<= 16) read lanes with four row-shift adds, one lane updates the wave's
					// counters: no atomics (an atomic on one address from many lanes costs a full reduction)
#define ARKS_ROW_ADD(v, ctrl) v += (u32)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true)
					ARKS_ROW_ADD(st_a, 0x111); ARKS_ROW_ADD(st_b, 0x111); ARKS_ROW_ADD(st_c, 0x111); // row_shr:1
					ARKS_ROW_ADD(st_a, 0x112); ARKS_ROW_ADD(st_b, 0x112); ARKS_ROW_ADD(st_c, 0x112); // row_shr:2
					ARKS_ROW_ADD(st_a, 0x114); ARKS_ROW_ADD(st_b, 0x114); ARKS_ROW_ADD(st_c, 0x114); // row_shr:4
					ARKS_ROW_ADD(st_a, 0x118); ARKS_ROW_ADD(st_b, 0x118); ARKS_ROW_ADD(st_c, 0x118); // row_shr:8
#undef ARKS_ROW_ADD
					if (lane == 15) {
						S.wstats[0] += st_a & 0xFFFFu;          // valid windows
						S.wstats[7] += st_a >> 16;              // all windows (bad = all - valid)
						S.wstats[2] += st_b & 0xFFFFu;          // found
						S.wstats[3] += st_b >> 16;              // recorded
						S.wstats[4] += st_c & 0xFFFFu;          // duplicate (value 0)
						S.wstats[5] += (st_c >> 16) & 0xFFu;    // reads passing
						S.wstats[6] += st_c >> 24;              // reads failing
					}
				}
			}
			for (int j = 0; FULL && j < nr; ++j) {
				const long r = (long)S.rid[j];
				const int L = S.rlen[j];
				if (L < 0) {
					if (lane == 0)
						put_none<RAW>(out_conreci, r);
					continue;
				}
				if ((redo_mask >> j) & 1u) {
					if (lane == 0)
						queue[atomicAdd(queue_count, 1u)] = (u32)r;
					continue;
				}
				if ((redo2_mask >> j) & 1u) {
					if (lane == 0)
						mqueue[atomicAdd(queue_count + 2, 1u)] = (u32)r;
					continue;
				}
				if ((failmask >> j) & 1u) { // (only without counters)
					if (lane == 0)
						put_none<RAW>(out_conreci, r);
					continue;
				}
				const int nwin = L - k + 1;
				const int p0 = S.rstart[j];
				int best = 0, best_cnt = 0;
				// first sweep: counters + is there more than one distinct positive value?
				int first = 0, first_cnt = 0;
				bool multi = false;
				for (int base = 0; base < nwin; base += 64) {
					const int p = base + lane;
					const int v = p < nwin ? rec[p0 + p] : -3;
					if (STATS) {
						ws.bad += __popcll(__ballot(v == -2));
						ws.valid += __popcll(__ballot(v >= -1));
						ws.found += __popcll(__ballot(v >= 0));
						ws.rec += __popcll(__ballot(v > 0));
						ws.dup += __popcll(__ballot(v == 0));
					}
					const u64 pos = __ballot(v > 0);
					if (pos) {
						if (first == 0)
							first = __shfl(v, __ffsll((long long)pos) - 1);
						const u64 eq = __ballot(v == first);
						first_cnt += __popcll(eq);
						multi = multi || (pos & ~eq) != 0;
					}
				}
				if (!multi) {
					best = first;
					best_cnt = first_cnt;
				} else {
					// ascending walk over the distinct values (the std::map order of Arcs.cpp:998)
					int prev = 0;
					for (;;) {
						int m = 0x7FFFFFFF;
						for (int base = 0; base < nwin; base += 64) {
							const int p = base + lane;
							const int v = p < nwin ? rec[p0 + p] : -3;
							m = (v > prev && v < m) ? v : m;
						}
						m = wave_min_i32(m);
						if (m == 0x7FFFFFFF)
							break;
						int cnt = 0;
						for (int base = 0; base < nwin; base += 64) {
							const int p = base + lane;
							cnt += __popcll(__ballot(p < nwin && rec[p0 + p] == m));
						}
						if (cnt > best_cnt) { // strict: the smallest value keeps a tie
							best_cnt = cnt;
							best = m;
						}
						prev = m;
					}
				}
				const int total = nwin > 0 ? nwin : 0;
				const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
				const bool pass = maxj > j_index;
				if (lane == 0)
					put_result<RAW>(out_conreci, r, best, best_cnt, pass);
				if (STATS) {
					ws.pass += pass;
					ws.fail += !pass;
					ws.win += (u64)total;
				}
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(8);
			cur = nxt;
		}
	}
#ifdef ARKS_PROFILE_SECTIONS
#ifdef ARKS_PROFILE_MEDIUM // (the medium kernel's sections alone: -DARKS_PROFILE_SECTIONS -DARKS_PROFILE_MEDIUM)
	if (FULL)
#endif
	if (lane_id == 0)
		for (int x = 0; x < 12; ++x)
			atomicAdd(&g_sec_cycles[x], sec_acc[x]);
#endif
#ifdef ARKS_MEDIUM_DIAG
	if (FULL)
		for (int x = 0; x < 16; ++x) {
			const bool per_lane = x == 6;
			if ((per_lane || lane_id == 0) && md[x])
				atomicAdd(&g_med_diag[x], md[x]);
		}
#endif
	if (STATS && !FULL) {
		ARKS_WAVE_SYNC();
		ws.valid += S.wstats[0];
		ws.bad += S.wstats[7] - S.wstats[0];
		ws.found += S.wstats[2];
		ws.rec += S.wstats[3];
		ws.dup += S.wstats[4];
		ws.pass += S.wstats[5];
		ws.fail += S.wstats[6];
		ws.win += S.wstats[7];
	}
	if (STATS && lane_id == 0) {
		if (ws.valid) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 0, ws.valid);
		if (ws.bad) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 1, ws.bad);
		if (ws.found) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 2, ws.found);
		if (ws.rec) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 3, ws.rec);
		if (ws.dup) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 4, ws.dup);
		if (ws.pass) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 5, ws.pass);
		if (ws.fail) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 6, ws.fail);
		if (ws.win) atomicAdd(stats + kStatRow * (blockIdx.x & (kStatRows - 1)) + 7, ws.win);
	}
}

// ------------------------------------------------------------------------------------------------
// K3s: the hot kernel of the SEED index (BIndexView::dense): bestContig for a batch, one wave per tile.
//
// With every m-mer position of the text in the table a read needs no minimizers: its windows are cut into
// groups of w = k - MM + 1 consecutive ones and the m-mer at the start of a group's last window -- which
// lies inside every window of the group -- is the group's seed.  What is left of the minimizer kernel is
// its back half, and that is reorganised here around twice the tile: up to 64 packed words (14 10x reads)
// per tile, lanes = words for ONE diagonal at a time (the second diagonal of a read -- a duplicated
// segment, a chance seed hit -- is rare: its pass over the same lanes only runs for a tile that has one),
// the seeds' entry lists stay in registers (lanes = seeds from the probe to the election of the
// diagonals), and everything a tile needs per read is written by the lane that holds the read's chunk
// metadata.  Per tile: one round trip for the read words, one for the seed probes, one for the text along
// the diagonals; 8 waves per SIMD.  Reads the hot path cannot finish (a third diagonal, a heavy seed, more
// than two entries, a match that spans two contig ends) go to the medium queue (map_reads_b_kernel<FULL,
// DENSE>), palindromes next to quirk images and reads beyond a medium tile to the slow one.
//   S0 tile = the next reads of the chunk that fit 64 words / 16 reads / 64 seeds (one ballot);
//   S1 words -> LDS; per-read metadata, per-word metadata and the seeds by the reads' own lanes;
//   S2 lanes = seeds: canonical m-mer, table probe (<= 2 entries), proposals of diagonals;
//   S3 election of <= 2 diagonals per read (LDS atomic minima keyed by seed index), flags;
//   S4 lanes = staging slots: text / visited / ambiguous / owner words of diagonal A (then B);
//   S5 lanes = words: read XOR text -> mismatch bit per base; valid windows, k clear mismatch bits,
//      visited, ambiguous -> popcounts into per-read counters;
//   S6 lane = read: the vote of Arcs.cpp:996-1013.
// ------------------------------------------------------------------------------------------------
// A seed's answer from its owner (sharded seed table): a0 = first entry or 0, a1 = second or 0; entries have
// bit 63 set, so 1 = "more than two entries" and 2 = "heavy seed" cannot be entries
constexpr u64 kAnsOverflow = 1ull, kAnsHeavy = 2ull;
__device__ __forceinline__ u32
seed_answer_count(u64 a0, u64 a1)
{
	if (a0 == kAnsOverflow)
		return kHnOverflow;
	if (a0 == kAnsHeavy)
		return kHnHeavy;
	return a0 == 0 ? 0u : (a1 == 0 ? 1u : 2u);
}

constexpr int sTW = 64;               // tile capacity in packed words (= lanes of S5)
constexpr int sTR = 16;               // reads per tile
constexpr int sNH = 64;               // seeds per tile (= lanes of S2)
constexpr int sChunk = 56;            // reads per grab of the work counter: four tiles of 10x reads
constexpr int sSlots = sTW + sTR + 2; // staging slots: one more than its words per read

#ifndef ARKS_PARK_FIRST
#define ARKS_PARK_FIRST 1
#endif
struct SeedTileLds
{
	u64 cw[sTW + 4];
	u32 nm[sTW + 4];
	u32 wmeta[sTW + 4]; // per word: read of the tile << 16 | local end position of that read (0 = not evaluated)
	// text words along ONE diagonal per read (read j: slots from (rstart[j] >> 5) + j), mismatch bits per base
	u64 tcodes[sSlots];
	u32 tvis[sSlots];
	u32 tamb[sSlots];
	u32 town[sSlots];
	u32 mm32[sTW + 8];
	unsigned short heads[sNH]; // seeds: [11:1] tile position, [15:12] read of the tile
	unsigned char sread[sSlots + 2];
	int rstart[sTR + 1];
	int rlen[sTR];
	u64 pdiag[sTR][2];  // the read's two diagonals: [39:0] D, [40] same strand, [41] valid
	u32 tfirst[sTR][2]; // first text word staged for read j on diagonal d
	u32 rcnt[sTR][2];   // matched unambiguous windows | ambiguous ones << 10 | (d = 0: valid windows) << 20
	u32 rmin[sTR][2];   // smallest / largest contig end among the former (they differ iff more than one)
	u32 rmax[sTR][2];
	u32 redo;  // reads for the slow queue
	u32 redo2; // reads for the medium queue
	u32 rempty[sTR]; // bit gi: seed gi of the read has no entry -- every window that holds it is absent (S6, flagged reads)
#if ARKS_PARK_FIRST
	// the answer the chunk's pre-pass (S-1) got for a read's FIRST seed, kept for the tile when it is the usual one -- exactly
	// one entry: its text position, and in ppark bit 0 "there is one", bit 1 its strand.  The tile then does not probe that
	// seed again (one probe in three of a kept 151-base read, one in two of a 128-base one).  280 of the 320 bytes that 32
	// waves per CU leave beside the tile.
	u32 ppos[sChunk];
	unsigned char ppark[sChunk];
#endif
	u64 wstats[8];
	// REMOTE without counters: the reads none of whose seeds has an entry are settled before the tiles are made, the
	// others move up into their places
	u64 wbase[sTR];            // tile word x of read j comes from codes[wbase[j] + x]
	int sbase[sTR];            // tile seed h of read j is seed sbase[j] + h of the chunk
	unsigned char rorig[sTR];  // read j of the tile is read c0 + rorig[j] of the batch
	unsigned char perm[64];
	// reads for the medium queue, collected per wave and handed over 47 or more at a time with ONE global atomic (round 5).
	// Until then every flagged read's lane did `mqueue[atomicAdd(queue_count + 2, 1)] = r`: a dependent round trip per
	// tile on a counter that every wave of the launch adds to -- nothing on the headline's draft (15 k of 40 M reads), the
	// hot kernel's whole difference on a repeat-rich one: 16.0 -> 8.4 ms per 20 M pairs on the human-like draft
	// (profiles/r09i_retry_kernels.txt found it by accident: a variant that pushed once per chunk instead of per tile)
	u32 mqb[64];
	u32 mqn;
};

#ifndef ARKS_SEED_WAVES
#define ARKS_SEED_WAVES 8
#endif

// REMOTE = true: the seed table is sharded over the ranks of a node (arks_index_build_seed_shard): the probes
//                of S2 were answered by the seeds' owners before the launch (arks_seeds_fill_device -> all-to-all
//                -> arks_seeds_probe_device -> all-to-all); seed_off[r] = index of read r's first seed,
//                ans[2 s], ans[2 s + 1] = the answer to seed s (seed_answer).
template <int KW, bool STATS, int MM, bool RAW, bool REMOTE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ARKS_SEED_WAVES)))
map_reads_s_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ nmask,
    const u64* __restrict__ word_off,
    const u32* __restrict__ lens,
    const uint8_t* __restrict__ eval, // may be NULL
    long n_reads,
    double j_index,
    KeyGeom g,
    BIndexView bx,
    int* __restrict__ out_conreci,
    u64* __restrict__ stats,
    u32* __restrict__ queue,       // slow queue
    u32* __restrict__ mqueue,      // medium queue
    u32* __restrict__ queue_count, // [0] slow length, [2] medium length; work counters behind (kWorkCtrOffset)
    const long* __restrict__ seed_off = nullptr, // REMOTE only
    const u64* __restrict__ ans = nullptr,       // REMOTE only
    const u32* __restrict__ seed_slot = nullptr, // REMOTE only, may be NULL: the answer of seed s is ans[2 seed_slot[s]]
                                                 // (arks_exchange: answers lie in send-buffer order) instead of ans[2 s]
    const u32* __restrict__ chunk_off = nullptr, // REMOTE only, instead of seed_off: first seed of chunk c0 / sChunk
    // the pair gate of chromiumRead (Arcs.cpp:1264-1268) worked out here instead of read from `eval` (round 5: the
    // gate launch -- a pass over three arrays of the batch -- was 1.6 % of a step): gate_class != NULL: reads 2p, 2p + 1
    // are mates, read r is evaluated iff gate_ok[r / 2] (NULL: yes) and both mates' class has bit 0; class bit 1 = ACGT only
    const uint8_t* __restrict__ gate_class = nullptr,
    const uint8_t* __restrict__ gate_ok = nullptr)
{
	typedef typename Mmer<MM>::type mm_t;
	__shared__ SeedTileLds S;
	constexpr u64 kDiagMask = (1ull << 42) - 1ull; // [39:0] D, [40] same strand, [41] valid
	const int lane_id = threadIdx.x;
	const int k = g.k, w = bx.w;
	const u32 wrecip = (65536u + (u32)w - 1u) / (u32)w; // x / w == (x * wrecip) >> 16 for x < 65536 / w
	const float jf = (float)j_index;
	// No counters, no raw votes: a read none of whose seeds has an entry -- half of a uniform read set lies outside the
	// contig ends -- has no window in the index: its result is 0, and it is settled per chunk, before the tiles are made;
	// the tiles then hold the other reads only (half as many tiles).  Not when the index holds quirk images: a
	// palindromic window's key is not its sequence's, the slow kernel decides those and finds them through the staged
	// words (below).
	// REMOTE: the seeds' answers are there before the launch.  Otherwise the kernel probes, per chunk, every read's FIRST
	// seeds -- as many as it takes for "none of them has an entry" to settle the vote: a seed without entries means that
	// every window of its group is absent from the index (the property the whole kernel rests on), so the best contig
	// can hold at most the windows of the groups behind, and when (those) / (all windows) cannot exceed j_index the
	// read's result is 0 whatever the other seeds say (Arcs.cpp:1006-1010): one of two seeds for a 128-base read at
	// k = 60 and j = 0.55, two of three for a 151-base one -- made from the batch's words; a read all of whose first
	// seeds are absent is settled, the others go into tiles, where every seed is probed as before (the first ones a
	// second time: cache hits).  An absent read costs its first probes and no tile; the dependent round trip is paid
	// once per chunk, not per tile (round 4: -4.7 % as a build option, profiles/r04s_ab_skip_dead_fused.txt; the
	// default since round 5).  Tried and dropped: the same two rounds PER TILE (21 % fewer probes, +1.5 % time,
	// profiles/r04c_ab_two_round.txt), and -- round 5 -- every seed of a chunk probed once from words staged in LDS with
	// the answers parked there, tiles of one round trip (text) each: 8-12 KB of LDS per wave leave 3-5 waves per SIMD
	// and the launch takes 5.1-6.0 ms against 3.6 (profiles/r09a_ab_chunk_probe.txt: the kernel's time is round trips
	// x waves, and LDS is what buys waves).
#if defined(ARKS_CAL_NO_PROBE) || defined(ARKS_CAL_NO_TREC)
#ifndef ARKS_CALIBRATION_BUILD
#error "ARKS_CAL_NO_PROBE / ARKS_CAL_NO_TREC are calibration builds (results wrong by design): pass -DARKS_CALIBRATION_BUILD too"
#endif
#endif
#ifdef ARKS_CAL_NO_PROBE
	// (calibration build: no probe anywhere, hence no diagonal and no text record -- every read goes through the tiles and
	// the kernel fetches its read stream and nothing else, a byte count that is known exactly; profiles/tools/fetch_calib.py)
	constexpr bool kSkipDead = REMOTE && !STATS && !RAW;
#else
	constexpr bool kSkipDead = !STATS && !RAW;
#endif
	const bool skip_dead = kSkipDead && !(bx.has_img && !(k & 1));
	if (STATS && lane_id < 8)
		S.wstats[lane_id] = 0;
	if (lane_id == 0)
		S.mqn = 0u;
	// the collected reads -> the medium queue: one atomic for all of them, their numbers by the lanes
	auto flush_mq = [&]() {
		ARKS_WAVE_SYNC();
		const u32 nq = S.mqn;
		if (nq != 0u) {
			int cl = lane_id; // (opaque: see the chunk's metadata below)
			asm volatile("" : "+v"(cl));
			u32 qb = 0;
			if (cl == 0)
				qb = atomicAdd(queue_count + 2, nq);
			qb = (u32)__builtin_amdgcn_readfirstlane((int)qb);
			if ((u32)cl < nq)
				mqueue[qb + (u32)cl] = S.mqb[cl];
			ARKS_WAVE_SYNC();
			if (cl == 0) {
				u32 z;
				asm volatile("v_mov_b32 %0, 0" : "=v"(z));
				S.mqn = z;
			}
			ARKS_WAVE_SYNC();
		}
	};
	u32* const work_ctr = reinterpret_cast<u32*>(reinterpret_cast<char*>(queue_count) + kWorkCtrOffset);
	u32 ctr = blockIdx.x & (u32)(kCounters - 1), misses = 0;
	bool first_grab = true;
#ifdef ARKS_PROFILE_SECTIONS
	unsigned long long sec_acc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long sec_t0 = __builtin_amdgcn_s_memtime();
#endif
	for (;;) {
		// ---- the next chunk: the first one is the wave's own (its block index), later ones come from the
		//      interleaved counters (see map_reads_b_kernel).  (Asking for the NEXT chunk's number while this one is
		//      worked on -- the atomic's round trip out of the chain -- was tried in round 4: no gain,
		//      profiles/r06g_ab_grab_prefetch.txt) -------------------------------------------
		long c0;
		if (first_grab) {
			c0 = (long)blockIdx.x * sChunk;
			first_grab = false;
			if (c0 >= n_reads)
				break;
		} else {
			u32 cnt = 0;
			if (lane_id == 0)
				cnt = atomicAdd(work_ctr + ctr * kCounterStride, 1u);
			cnt = (u32)__builtin_amdgcn_readfirstlane((int)cnt);
			c0 = ((long)gridDim.x + (long)ctr + (long)cnt * kCounters) * sChunk;
			if (c0 >= n_reads) {
				ctr = (ctr + 1u) & (u32)(kCounters - 1);
				if (++misses == (u32)kCounters)
					break;
				continue;
			}
			misses = 0;
		}
		const int nchunk = (int)((c0 + sChunk < n_reads ? c0 + sChunk : n_reads) - c0);
		// lane l holds read c0 + l: first word, length (-1 = not evaluated), seeds; lane nchunk the end offset
		u64 wo = 0;
		int rl = 0;
		bool may_n = false; // (lanes beyond the chunk: no read)
		{
			// the lane number as an opaque value: addresses built from it are formed here, per chunk, instead of
			// being hoisted out of the chunk loop as loop invariants -- which is what put two registers into
			// scratch memory at 64 VGPRs (the only scratch use of the kernel)
			int cl = lane_id;
			asm volatile("" : "+v"(cl));
			if (cl <= nchunk)
				wo = word_off[c0 + cl];
			if (cl < nchunk) {
				rl = (int)lens[c0 + cl];
				may_n = true; // (without an eval array nothing is known: the masks are fetched)
				if (gate_class) { // (arks_pair_gate_device's rule; n_reads is even)
					const long r = c0 + cl;
					const uint8_t c_me = gate_class[r];
					const uint8_t p_ok = gate_ok ? gate_ok[r >> 1] : (uint8_t)1;
					// (the mate's class sits in the neighbouring lane: chunks start at even reads and hold whole pairs)
					static_assert(sChunk % 2 == 0, "a chunk holds whole pairs");
					const int c_mate = __builtin_amdgcn_update_dpp(0, (int)c_me, 0xB1, 0xF, 0xF, true); // quad_perm [1, 0, 3, 2]
					if (!(p_ok && (c_me & 1) && (c_mate & 1)))
						rl = -1;
					may_n = !(c_me & 2);
				} else if (eval) {
					const uint8_t ev = eval[c0 + cl];
					if (!ev)
						rl = -1;
					// exactly ARKS_EVAL_ACGT_ONLY (3): evaluate, and the read is known to hold ACGT only -- what
					// arks_pair_gate_device writes from the read class; any other nonzero value (1, 2, 0xFF ...) is a plain
					// "evaluate" and the masks are fetched (arks_hip.h)
					may_n = ev != 3;
				}
			}
		}
		// reads of the chunk that may hold an invalid base: only a tile with one of them fetches its N masks (a
		// third of the read stream, and zero for > 98 % of the reads)
		ARKS_SEC(0);
		// index of the chunk's first seed in `ans` / `seed_slot`
		const long soff0 = REMOTE ? (chunk_off ? (long)chunk_off[c0 / sChunk] : seed_off[c0]) : 0;
		int nwin_l = rl - k + 1;
		int G = nwin_l > 0 ? (int)(((u32)(nwin_l + w - 1) * wrecip) >> 16) : 0;
		int gex = wave_incl_scan_i32(G) - G; // seeds of the chunk's reads before this one
		int wcnt = (int)(wave_next_u64(wo) - wo); // words of the read (garbage beyond the chunk: not looked at)
		// pos: where the read starts in the word space the tiles are cut from -- the batch's packed words (wo), or, when
		// reads are left out, the words of the reads that stay, closed up; wo stays the address of the read's first word
		u64 pos = wo;
		int gexo = gex;      // the read's first seed among the chunk's (answers are numbered with every read in place)
		int rorig = lane_id; // the read this lane holds is read c0 + rorig
		u32 park_pos = 0, park_code = 0; // (ARKS_PARK_FIRST) the first seed's one entry, if that is what the pre-pass found
		int nchunk_c = nchunk;
		if (kSkipDead && skip_dead) {
			bool keep = lane_id < nchunk && rl >= 0 && G > 0;
			if (!REMOTE) {
				if (keep) {
					// g1 = the read's first seeds: with all of them absent the windows behind them cannot reach j_index (the
					// float test errs to the safe side: a margin of 1e-4 against a rounding error of 1e-7)
					int g1 = 1;
					while (g1 < G && !((float)(nwin_l - g1 * w) * 1.0001f < jf * (float)nwin_l))
						++g1;
					if (g1 <= 2) { // (more: the read is kept unseen)
						// both seeds' words in flight together, then both probes (an aligned group of four entries each):
						// two round trips per chunk whatever the read's length
						bool act[2] = { false, false };
						u64 at[2] = { 0, 0 };
						u64 w0[2], w1[2], nm2[2] = { 0, 0 };
#pragma unroll
						for (int gi = 0; gi < 2; ++gi) {
							int q = (gi + 1) * w - 1;
							q = q < nwin_l - 1 ? q : nwin_l - 1;
							act[gi] = gi < g1;
							at[gi] = wo * 32ull + (u64)q;
							const u64* src = codes + (at[gi] >> 5);
							w0[gi] = act[gi] ? src[0] : 0ull;
							w1[gi] = act[gi] ? src[1] : 0ull;
							if (act[gi] && may_n) { // (a seed that holds an invalid base has no entries)
								const u32* nm = nmask + (at[gi] >> 5);
								nm2[gi] = ((u64)nm[0] << 32) | (u64)nm[1];
							}
						}
						mm_t cm[2];
						u64 slot[2];
						ulonglong2 h[2][2];
#pragma unroll
						for (int gi = 0; gi < 2; ++gi) {
							const mm_t mf = (mm_t)(funnel_l(w0[gi], w1[gi], (int)(at[gi] & 31) * 2) >> (64 - 2 * MM)), mr = mmer_rc<MM>(mf);
							cm[gi] = mf < mr ? mf : mr;
							act[gi] = act[gi] && ((nm2[gi] << (at[gi] & 31)) >> (64 - MM)) == 0;
							slot[gi] = mtab_home<MM>(cm[gi], bx.mtab_cap);
							h[gi][0] = make_ulonglong2(0ull, 0ull), h[gi][1] = make_ulonglong2(0ull, 0ull);
							if (act[gi]) {
								h[gi][0] = *reinterpret_cast<const ulonglong2*>(bx.mtab + slot[gi]);
								h[gi][1] = *reinterpret_cast<const ulonglong2*>(bx.mtab + slot[gi] + 2);
							}
						}
						bool any = false;
#pragma unroll
						for (int gi = 0; gi < 2; ++gi) {
							if (!act[gi])
								continue;
							const u32 fp = mmer_fp<MM>(cm[gi]);
							const u64 e[4] = { h[gi][0].x, h[gi][0].y, h[gi][1].x, h[gi][1].y };
							bool ended = false, hit = false;
#if ARKS_PARK_FIRST
							// (the whole group is looked at, as probe_minimizer_table does: how many entries, not just whether)
							u32 n_ent = 0;
							u64 first = 0;
#pragma unroll
							for (int x = 0; x < 4; ++x) {
								if (ended)
									continue;
								if (!(e[x] >> 63))
									ended = true; // an empty slot ends the probe sequence
								else if (((u32)(e[x] >> 32) & kFpMask) == fp) {
									hit = true;
									if ((u32)e[x] == kHeavyPos)
										ended = true, n_ent = 3; // the "heavy" mark
									else {
										if (n_ent == 0)
											first = e[x];
										++n_ent;
									}
								}
							}
							if (gi == 0 && ended && n_ent == 1) {
								park_pos = (u32)first;
								park_code = 1u | (((u32)(first >> 62) & 1u) << 1);
							}
#else
#pragma unroll
							for (int x = 0; x < 4; ++x) {
								if (ended)
									continue;
								if (!(e[x] >> 63))
									ended = true; // an empty slot ends the probe sequence
								else if (((u32)(e[x] >> 32) & kFpMask) == fp)
									hit = ended = true; // an entry, or the "heavy" mark
							}
#endif
							if (!ended && !hit) { // (a full group without the m-mer: rare -- the whole sequence)
								u64 e2[2];
								hit = probe_minimizer_table<MM>(bx, cm[gi], e2) != 0;
							}
							any = any || hit;
						}
						keep = any;
					}
				}
			} else if (keep && G <= 3) {
				// (three seeds at most: a 10x read has two or three; a read with more is kept unseen)
				u64 a[3] = { 0, 0, 0 };
				long si[3];
#pragma unroll
				for (int gi = 0; gi < 3; ++gi) {
					si[gi] = -1;
					if (gi < G) {
						si[gi] = soff0 + (long)(gex + gi);
						if (seed_slot) {
							const u32 sl = seed_slot[si[gi]];
							si[gi] = sl == ~0u ? -1 : (long)sl;
						}
					}
				}
#pragma unroll
				for (int gi = 0; gi < 3; ++gi)
					if (si[gi] >= 0)
						a[gi] = ans[2 * si[gi]];
				keep = (a[0] | a[1] | a[2]) != 0; // (an entry, or the "heavy" / "more than two" marks)
			}
			if (lane_id < nchunk && !keep) {
				int cl = lane_id; // (opaque: an address made of the lane number is otherwise formed once per kernel and kept)
				asm volatile("" : "+v"(cl));
				put_none<RAW>(out_conreci, c0 + cl);
			}
			const u64 kept = __ballot(keep);
			nchunk_c = __popcll((long long)kept);
			if (keep)
				S.perm[__builtin_amdgcn_mbcnt_hi((u32)(kept >> 32), __builtin_amdgcn_mbcnt_lo((u32)kept, 0u))] = (unsigned char)lane_id;
			ARKS_WAVE_SYNC();
			const bool have = lane_id < nchunk_c;
			rorig = have ? (int)S.perm[lane_id] : 0;
			const int sel = rorig << 2;
			wo = ((u64)(u32)__builtin_amdgcn_ds_bpermute(sel, (int)(u32)(wo >> 32)) << 32) |
			     (u64)(u32)__builtin_amdgcn_ds_bpermute(sel, (int)(u32)wo);
			// (every lane takes part in every permute: a lane that is switched off hands out zeros, and the lanes behind the
			// reads that stay are exactly the sources of the reads that move up)
			const int p_wcnt = __builtin_amdgcn_ds_bpermute(sel, wcnt);
			const int p_rl = __builtin_amdgcn_ds_bpermute(sel, rl);
			const int p_mn = __builtin_amdgcn_ds_bpermute(sel, (int)may_n | (REMOTE ? 0 : (int)(park_code << 1)));
#if ARKS_PARK_FIRST
			if (!REMOTE) {
				const int p_pp = __builtin_amdgcn_ds_bpermute(sel, (int)park_pos);
				if (have) {
					S.ppos[lane_id] = (u32)p_pp;
					S.ppark[lane_id] = (unsigned char)((u32)p_mn >> 1);
				}
			}
#endif
			gexo = __builtin_amdgcn_ds_bpermute(sel, gex);
			wcnt = have ? p_wcnt : 0;
			rl = have ? p_rl : 0;
			may_n = have && (p_mn & 1) != 0;
			nwin_l = rl - k + 1;
			G = have && nwin_l > 0 ? (int)(((u32)(nwin_l + w - 1) * wrecip) >> 16) : 0;
			gex = wave_incl_scan_i32(G) - G;
			pos = (u64)(u32)(wave_incl_scan_i32(wcnt) - wcnt); // (lane nchunk_c: the words of all of them)
			ARKS_WAVE_SYNC(); // (S.perm is rewritten by the next chunk)
		}
		// reads of the chunk that may hold an invalid base: only a tile with one of them fetches its N masks (a
		// third of the read stream, and zero for > 98 % of the reads)
		ARKS_SEC(5);
		const u64 nreads_mask = __ballot(may_n);
		// reads beyond kSW words (512 bases) do not enter a tile: the per-read counters hold 10-bit fields
		const u64 longmask = __ballot(lane_id < nchunk_c && wcnt > kSW);
		int cur = 0;
		while (cur < nchunk_c) {
			int lane = lane_id;
			asm volatile("" : "+v"(lane));
			if (S.mqn > 46u) // (a tile adds 16 at most, a read beyond a tile one: never more than 64)
				flush_mq();
			// ---- S0: tile = reads [cur, nxt) ---------------------------------------------------------
			const u64 base_w = lane_value_u64(pos, cur);
			const int gbase = __builtin_amdgcn_readlane(gex, cur);
			const u64 lm = longmask & (~0ull << cur);
			const int limit = lm ? __ffsll((long long)lm) - 1 : 64; // the first long read from cur on
			const u64 fit = __ballot(
			    lane > cur && lane <= nchunk_c && lane <= limit && pos - base_w <= (u64)sTW && lane - cur <= sTR &&
			    gex - gbase <= sNH);
			if (fit == 0) { // a single read beyond the tile (or with more seeds than lanes): general kernels
				if (lane == cur) {
					// (lane == cur: without reads left out the read's number is c0 + cur, a scalar -- written with the lane
					// number it would be an address the compiler forms once per kernel and keeps, or spills)
					const long r = c0 + (kSkipDead && skip_dead ? rorig : cur);
					if (rl < 0)
						put_none<RAW>(out_conreci, r);
					else if (wcnt <= kSW)
						S.mqb[atomicAdd(&S.mqn, 1u)] = (u32)r;
					else
						queue[atomicAdd(queue_count, 1u)] = (u32)r;
				}
				ARKS_WAVE_SYNC();
				cur++;
				continue;
			}
			const int nxt = 63 - __clzll((long long)fit);
			const int nr = nxt - cur;
			const int tw = (int)(lane_value_u64(pos, nxt) - base_w);
			const int nh = __builtin_amdgcn_readlane(gex, nxt) - gbase;
			ARKS_SEC(6);
			// ---- S1: words, metadata, seeds ----------------------------------------------------------
			// (wave-uniform) does a read of the tile hold an invalid base?  nr <= 16 reads from cur on
			const bool want_nm = ((nreads_mask >> cur) & ((1ull << nr) - 1ull)) != 0;
			// REMOTE with slot numbers: lane = seed asks for its slot here, in the round trip of the tile's words --
			// S2 then has one dependent load (the answer) where the fused kernel has its probe, not two
			u32 slot_pref = ~0u;
			{
				u64 c_in = 0, c_pad = 0;
				u32 m_in = 0, m_pad = 0;
				// (reads left out: the tile's words are not one stretch of the batch's -- the loads wait for the reads' lanes
				// to say where each word comes from, below)
				const bool gathered = kSkipDead && skip_dead;
				if (!gathered) {
					if (REMOTE && seed_slot && lane < nh)
						slot_pref = seed_slot[soff0 + (long)(gbase + lane)];
					if (lane < tw) {
						c_in = codes[base_w + (u64)lane];
						if (want_nm)
							m_in = nmask[base_w + (u64)lane];
					}
					if (lane < 4) { // the windows of the last words read past the tile
						c_pad = codes[base_w + (u64)(tw + lane)];
						if (want_nm)
							m_pad = nmask[base_w + (u64)(tw + lane)];
					}
				}
				if (lane >= cur && lane < nxt) {
					const int j = lane - cur;
					const int w0 = (int)(pos - base_w), w1 = w0 + wcnt;
					if (kSkipDead) {
						S.wbase[j] = wo - (u64)w0;
						S.sbase[j] = gexo - (gex - gbase);
						S.rorig[j] = (unsigned char)rorig;
					}
					const int rs = w0 * 32;
					const int rend = rl > 0 ? rs + rl : 0;
					S.rstart[j] = rs;
					S.rlen[j] = rl;
					if (lane == nxt - 1)
						S.rstart[j + 1] = w1 * 32;
					// (the three loops are kept rolled and scalar: unrolled or vectorised, their trip-count arithmetic -- per-chunk invariants of the
					// tile loop -- is what went to scratch memory at 64 VGPRs)
#pragma clang loop vectorize(disable) unroll(disable)
					for (int x = w0; x < w1; ++x)
						S.wmeta[x] = ((u32)j << 16) | (u32)rend;
#pragma clang loop vectorize(disable) unroll(disable)
					for (int x = w0; x <= w1; ++x) // one staging slot more than the read has words
						S.sread[x + j] = (unsigned char)j;
					const int hb = gex - gbase;
#pragma clang loop vectorize(disable) unroll(disable)
					for (int gi = 0; gi < G; ++gi) {
						int q = (gi + 1) * w - 1;
						q = q < nwin_l - 1 ? q : nwin_l - 1;
#if ARKS_PARK_FIRST
						// bit 0: this is the read's first seed and the pre-pass parked its answer (lane = the read's place in the chunk)
						const u32 pk = (kSkipDead && skip_dead && !REMOTE && gi == 0) ? (u32)(S.ppark[lane] & 1u) : 0u;
						S.heads[hb + gi] = (unsigned short)(((u32)(rs + q) << 1) | ((u32)j << 12) | pk);
#else
						S.heads[hb + gi] = (unsigned short)(((u32)(rs + q) << 1) | ((u32)j << 12));
#endif
					}
					S.pdiag[j][0] = ~0ull;
					S.pdiag[j][1] = ~0ull;
					S.rcnt[j][0] = 0u, S.rcnt[j][1] = 0u;
					S.rmin[j][0] = 0xFFFFFFFFu, S.rmin[j][1] = 0xFFFFFFFFu;
					S.rmax[j][0] = 0u, S.rmax[j][1] = 0u;
					S.rempty[j] = 0u;
				}
				if (lane == 0) {
					// (a zero made here: as a loop-invariant constant pair it was kept in -- and spilled from -- registers)
					u32 z;
					asm volatile("v_mov_b32 %0, 0" : "=v"(z));
					S.redo = z;
					S.redo2 = z;
				}
				if (gathered) {
					ARKS_WAVE_SYNC();
					if (REMOTE && seed_slot && lane < nh)
						slot_pref = seed_slot[soff0 + (long)(S.sbase[S.heads[lane] >> 12] + lane)];
					if (lane < tw) {
						const u64 at = S.wbase[S.wmeta[lane] >> 16] + (u64)lane;
						c_in = codes[at];
						if (want_nm)
							m_in = nmask[at];
					}
					if (lane < 4) { // (the words behind the tile's last read)
						const u64 at = S.wbase[nr - 1] + (u64)(tw + lane);
						c_pad = codes[at];
						if (want_nm)
							m_pad = nmask[at];
					}
				}
				asm volatile("" : "+v"(c_in), "+v"(m_in), "+v"(c_pad), "+v"(m_pad), "+v"(slot_pref)); // all loads in flight
				if (lane < tw) {
					S.cw[lane] = c_in;
					if (want_nm)
						S.nm[lane] = m_in;
				}
				if (lane < 4) {
					S.cw[tw + lane] = c_pad;
					if (want_nm)
						S.nm[tw + lane] = m_pad;
				}
			}
			ARKS_WAVE_SYNC();
			// (S.nm is only read under has_n)
			const bool has_n = want_nm && __ballot(lane < tw && S.nm[lane] != 0) != 0;
			if (bx.has_img && !(k & 1)) {
				// a reverse-complement palindrome carries its seed twice, mirrored about its centre: necessary
				// condition for a palindromic window (the slow kernel decides exactly); only looked for when the
				// index holds quirk images
				const int n = tw * 32;
				for (int base = 0; base < n; base += 64) {
					const int i = base + lane;
					const u32 wm = S.wmeta[i >> 5];
					const int j = (int)(wm >> 16);
					if ((int)(wm & 0xFFFFu) - i >= k) {
						const int p = i - S.rstart[j];
						const int nw = S.rlen[j] - k + 1;
						const int gi = (int)(((u32)p * wrecip) >> 16);
						int qr = (gi + 1) * w - 1;
						qr = qr < nw - 1 ? qr : nw - 1;
						const int q = S.rstart[j] + qr;
						const int qm = 2 * i + (k - MM) - q;
						if (tile_canonical_mmer<MM>(S.cw, qm) == tile_canonical_mmer<MM>(S.cw, q))
							atomicOr(&S.redo, 1u << j);
					}
				}
			}
			ARKS_SEC(7);
			// ---- S2: lanes = seeds: probe, proposals (0 = none) of up to two diagonals -------------------
			int jh = 0;
			u64 dk0 = 0, dk1 = 0;
			bool off = false;
			{
				int q = 0;
				bool parked = false;
				if (lane < nh) {
					const u32 hv = S.heads[lane];
					q = (int)((hv >> 1) & 2047u);
					jh = (int)(hv >> 12);
					parked = ARKS_PARK_FIRST && !REMOTE && (hv & 1u) != 0;
				}
				u64 ent[2];
				u32 cnt = 0;
				u32 rstrand = 0;
				// a seed that holds an invalid base has no entries: every window of its group holds that base too
				if (lane < nh && !(has_n && tile_span_has_n(S.nm, q, MM))) {
					const mm_t mf = tile_mmer<MM>(S.cw, q), mr = mmer_rc<MM>(mf);
					rstrand = mf < mr ? 1u : 0u;
					if (REMOTE) {
						long si = soff0 + (long)((kSkipDead && skip_dead ? S.sbase[jh] : gbase) + lane);
						if (seed_slot)
							si = slot_pref == ~0u ? -1 : (long)slot_pref;
						if (si >= 0) {
							const u64* a = ans + 2 * si;
							ent[0] = a[0], ent[1] = a[1];
							cnt = seed_answer_count(ent[0], ent[1]);
						}
#if ARKS_PARK_FIRST
					} else if (parked) {
						// (the pre-pass of the chunk probed this seed and found its one entry: read j of the tile is read cur + j of the chunk)
						ent[0] = mtab_entry(0u, (u32)(S.ppark[cur + jh] >> 1) & 1u, S.ppos[cur + jh]);
						cnt = 1;
#endif
					} else {
#ifndef ARKS_CAL_NO_PROBE
						cnt = probe_minimizer_table<MM>(bx, mf < mr ? mf : mr, ent);
#endif
					}
				}
				off = cnt == kHnHeavy || cnt == kHnOverflow;
				if (!STATS && !RAW && cnt == 0 && lane < nh) { // seed gi of its read (the last one sits at the last window, not at a multiple of w)
					// (a read of more than 32 seeds -- w = 4 at k = 20 and 24, tile reads reach 512 bases -- keeps the first 32
					// only: a shift by 32 or more is undefined, and on gfx950 wraps onto another seed's bit)
					const u32 gi = ((u32)(q - S.rstart[jh]) * wrecip) >> 16;
					if (gi < 32u)
						atomicOr(&S.rempty[jh], 1u << gi);
				}
				if (cnt >= 1 && cnt <= 2) {
					const int o = q - S.rstart[jh]; // offset of the seed in the read
					// same strand: read base x <-> text D + x ; opposite: read base x <-> text D - x
					const bool s0 = ((u32)(ent[0] >> 62) & 1u) == rstrand;
					dk0 = (s0 ? (u64)(u32)ent[0] - (u64)o : (u64)(u32)ent[0] + (u64)(MM - 1 + o)) | ((u64)s0 << 40) |
					      (1ull << 41);
					if (cnt == 2) {
						const bool s1 = ((u32)(ent[1] >> 62) & 1u) == rstrand;
						dk1 = (s1 ? (u64)(u32)ent[1] - (u64)o : (u64)(u32)ent[1] + (u64)(MM - 1 + o)) |
						      ((u64)s1 << 40) | (1ull << 41);
					}
				}
			}
			// ---- S3: A = entry 0 of the read's first seed that has entries, B = the first entry (in seed order)
			//      on another diagonal; a seed that proposes a third one sends its read to the medium queue ------
			if (dk0)
				atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][0]),
				          (unsigned long long)(((u64)lane << 42) | dk0));
			ARKS_WAVE_SYNC();
			if (dk0) {
				const u64 dA = S.pdiag[jh][0] & kDiagMask;
				if (dk0 != dA)
					atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][1]),
					          (unsigned long long)(((u64)lane << 43) | dk0));
				else if (dk1 && dk1 != dA)
					atomicMin(reinterpret_cast<unsigned long long*>(&S.pdiag[jh][1]),
					          (unsigned long long)(((u64)lane << 43) | (1ull << 42) | dk1));
			}
			ARKS_WAVE_SYNC();
			if (dk0) {
				const u64 dA = S.pdiag[jh][0] & kDiagMask;
				const u64 rb = S.pdiag[jh][1];
				const u64 dB = rb == ~0ull ? 0ull : (rb & kDiagMask);
				off = off || (dk0 != dA && dk0 != dB) || (dk1 && dk1 != dA && dk1 != dB);
			}
			if (off)
				atomicOr(&S.redo2, 1u << jh);
			ARKS_WAVE_SYNC();
			bool has_b = false;
			if (lane < nr) {
				// strip the seed index; first text word of the span [lo, lo + L) the read covers
#pragma unroll
				for (int d = 0; d < 2; ++d) {
					const u64 raw = S.pdiag[lane][d];
					const u64 pdv = raw == ~0ull ? 0ull : (raw & kDiagMask);
					S.pdiag[lane][d] = pdv;
					const u64 Dv = pdv & 0xFFFFFFFFFFull;
					const u64 lo = ((pdv >> 40) & 1ull) ? Dv : Dv - (u64)(S.rlen[lane] - 1);
					S.tfirst[lane][d] = (u32)(lo >> 5);
					has_b = d == 1 && pdv != 0;
				}
			}
			const bool any_b = __ballot(has_b) != 0;
			ARKS_WAVE_SYNC();
			ARKS_SEC(8);
			// ---- S4 / S5 per diagonal: A for every tile, B only for a tile that has one ---------------------
			u32 ok_a = 0; // the word's windows matched on diagonal A (a window matched on both counts once, on A)
			int jw = 0;
			u32 valid = 0;
			if (lane < tw) {
				const u32 wm = S.wmeta[lane];
				jw = (int)(wm >> 16);
				// windows of the word that exist and hold no invalid base
				valid = word_valid_windows(S.nm, lane, S.rlen[jw] - k + 1 - (lane * 32 - S.rstart[jw]), k, has_n);
			}
			for (int d = 0; d < (any_b ? 2 : 1); ++d) {
				// S4: lanes = staging slots (one round trip: text, visited, ambiguous, owner words)
				const int ns = tw + nr;
				// A tile of 10x reads has ~63 words + 14 reads = 77 staging slots: more than lanes.  As a loop (until round 4)
				// the second batch of slots was loaded only after the first was stored -- a dependent round trip more per
				// tile and diagonal, and the kernel's time is the sum of those: both batches are requested
				// before either is used (-4.3 %, profiles/r05f_ab_s4.txt).  codes | visited, ambig: one 16-byte record; the
				// owner from the table of 32-word blocks (cache resident), from word_owner only for a block that holds a
				// border of two ends.  (ARKS_CAL_NO_TREC: a calibration build, results wrong by design -- no text record is
				// fetched: FETCH_SIZE of the normal build minus this one's is what the records cost;
				// profiles/tools/traffic_classes.py)
				{
					static_assert(sSlots <= 128, "two slots per lane");
					ulonglong2 rec[2];
					u32 own[2];
					u32 twi[2]; // (text words: 2^32 positions / 32)
					bool act[2];
#pragma unroll
					for (int u = 0; u < 2; ++u) {
						const int sl = lane + 64 * u;
						act[u] = sl < ns;
						rec[u] = make_ulonglong2(0ull, 0ull);
						own[u] = 0;
						twi[u] = 0;
						if (act[u]) {
							const int j = S.sread[sl];
							act[u] = (S.pdiag[j][d] >> 41) != 0;
							twi[u] = S.tfirst[j][d] + (u32)(sl - ((S.rstart[j] >> 5) + j));
						}
						if (act[u]) {
#ifndef ARKS_CAL_NO_TREC
							rec[u] = *reinterpret_cast<const ulonglong2*>(bx.trec + 2 * (u64)twi[u]);
#endif
							own[u] = bx.owner_blk[twi[u] >> 5];
						}
					}
					asm volatile("" : "+v"(rec[0].x), "+v"(rec[1].x), "+v"(own[0]), "+v"(own[1])); // all four loads in flight
#pragma unroll
					for (int u = 0; u < 2; ++u)
						if (act[u]) {
							if (own[u] == 0xFFFFFFFFu)
								own[u] = bx.word_owner[twi[u]];
							const int sl = lane + 64 * u;
							S.tcodes[sl] = rec[u].x;
							S.tvis[sl] = (u32)rec[u].y;
							S.tamb[sl] = (u32)(rec[u].y >> 32);
							S.town[sl] = own[u];
						}
				}
				ARKS_SEC(9);
				if (lane < 8) // the spans of the last words read past the tile: no mismatch there
					S.mm32[tw + lane] = 0u;
				ARKS_WAVE_SYNC();
				// S5a: lanes = words: XOR with the 32 text bases the word faces -> one mismatch bit per base
				u64 pdv = 0;
				if (lane < tw) {
					pdv = S.pdiag[jw][d];
					S.mm32[lane] = (pdv >> 41) ? word_mismatch_bits(S.cw[lane], pdv, S.tcodes, lane * 32 - S.rstart[jw],
					                                                (S.rstart[jw] >> 5) + jw, S.tfirst[jw][d])
					                           : 0u;
				}
				ARKS_WAVE_SYNC();
				// S5b: lanes = words, one bit per window: k clear mismatch bits, visited, ambiguous
				u32 ok = 0, amb = 0, own = 0;
				if (lane < tw && (pdv >> 41) && valid)
					word_match(pdv, S.mm32, S.tvis, S.tamb, S.town, lane, lane * 32 - S.rstart[jw], (S.rstart[jw] >> 5) + jw,
					           S.tfirst[jw][d], k, valid, ok, amb, own);
				if (d) {
					ok &= ~ok_a;
					amb &= ok;
				} else
					ok_a = ok;
				const u32 recm = ok & ~amb;
				const u32 pack = (u32)__popc(recm) | ((u32)__popc(amb) << 10) | (d ? 0u : ((u32)__popc(valid) << 20));
				if (pack)
					atomicAdd(&S.rcnt[jw][d], pack);
				if (recm) {
					atomicMin(&S.rmin[jw][d], own == 0xFFFFFFFFu ? 1u : own); // mixed word: min != max
					atomicMax(&S.rmax[jw][d], own);
				}
				ARKS_WAVE_SYNC(); // the staging storage is rewritten by the next diagonal
			}
			ARKS_SEC(10);
			// ---- S6: lane = read: at most two distinct positive values (one per diagonal): the vote of
			//      Arcs.cpp:998-1004 is a compare ------------------------------------------------------------
			const u32 redo_mask = S.redo, redo2_mask = S.redo2;
			u32 st_a = 0, st_b = 0, st_c = 0;
			if (lane < nr) {
				const int j = lane;
				const long r = c0 + (kSkipDead && skip_dead ? (int)S.rorig[j] : cur + j);
				const int L = S.rlen[j];
				if (L < 0) {
					put_none<RAW>(out_conreci, r);
				} else if ((redo_mask >> j) & 1u) {
					queue[atomicAdd(queue_count, 1u)] = (u32)r;
				} else {
					bool medium = (redo2_mask >> j) & 1u;
					const u32 ca = S.rcnt[j][0], cb = S.rcnt[j][1];
					const int rec_a = (int)(ca & 1023u), amb_a = (int)((ca >> 10) & 1023u);
					const int rec_b = (int)(cb & 1023u), amb_b = (int)((cb >> 10) & 1023u);
					const int nvalid = (int)(ca >> 20);
					const u32 own_a = rec_a ? S.rmin[j][0] : 0u, own_b = rec_b ? S.rmin[j][1] : 0u;
					// EVERY valid window of the read was found on the staged diagonals: nothing that a heavy seed, a longer
					// entry list or a third diagonal could add -- a window's value is its KEY's (whichever text position shows
					// it), and every window has its value.  (Reads inside a repeat copy: the seed that is not heavy finds the
					// locus, the error-free ones are finished here instead of in the medium kernel.)
					if (medium && nvalid > 0 && rec_a + amb_a + rec_b + amb_b == nvalid)
						medium = false;
					if (medium && !STATS && !RAW) {
						// ... or the vote is settled by the windows that WERE found.  u windows are open; each could still turn
						// out to belong to any contig end.  If the leader among the found ones passes j_index on its own and
						// leads the other by more than u, it is the answer whatever the open windows hold (Arcs.cpp:998-1010:
						// the count only grows, nobody can catch up or tie); if even leader + u does not pass, the answer is 0.
						int u = nvalid - (rec_a + amb_a + rec_b + amb_b);
						// ... less the windows that hold a seed WITHOUT an entry: the seed table has every m-mer position of the
						// visited windows, so such a window is in no visited window whatever a heavy seed beside it says (round 5;
						// a read drawn from a repeat copy outside the indexed contig ends: one seed heavy, the others empty --
						// one flagged read in seven of the human-like draft ends here instead of in the medium kernel).  Seed gi
						// < G - 1 answers for the w windows of its group, the last one for the w windows that end at the last.
						if (const u32 em = S.rempty[j]) {
							const int nw = L - k + 1;
							const int G = (int)(((u32)(nw + w - 1) * wrecip) >> 16);
							// (G > 32: bits 0..31 are seeds that are not the last one, each answers for its w windows; the seeds
							// beyond have no bit and count as present)
							int gone = G > 32 ? __popc(em) * w : __popc(em & ((1u << (G - 1)) - 1u)) * w;
							if (G <= 32 && ((em >> (G - 1)) & 1u))
								gone += (G >= 2 && ((em >> (G - 2)) & 1u)) ? nw - (G - 1) * w : (nw < w ? nw : w);
							gone -= nw - nvalid; // (windows with an invalid base are not among the nvalid: all of them may lie there)
							u -= gone > 0 ? gone : 0;
						}
						int lead, other;
						if (rec_a > 0 && rec_b > 0 && own_a == own_b)
							lead = rec_a + rec_b, other = 0;
						else
							lead = rec_a > rec_b ? rec_a : rec_b, other = rec_a > rec_b ? rec_b : rec_a;
						const double tot = (double)(L - k + 1);
						if ((lead > other + u && (double)lead / tot > j_index) || !((double)(lead + u) / tot > j_index))
							medium = false;
					}
					// matches on one diagonal that belong to different contig ends: general path
					medium = medium || (rec_a && S.rmax[j][0] != own_a) || (rec_b && S.rmax[j][1] != own_b);
					if (medium) {
						S.mqb[atomicAdd(&S.mqn, 1u)] = (u32)r;
					} else {
						int best = 0, best_cnt = 0;
						if (rec_a > 0 && rec_b > 0 && own_a == own_b) {
							best = (int)own_a;
							best_cnt = rec_a + rec_b;
						} else if (rec_a > rec_b || (rec_a == rec_b && own_a < own_b)) {
							best = (int)own_a; // strict `>` of the reference walk: the smaller value keeps a tie
							best_cnt = rec_a;
						} else {
							best = (int)own_b;
							best_cnt = rec_b;
						}
						const int nwin = L - k + 1;
						const int total = nwin > 0 ? nwin : 0;
						const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
						const bool pass = maxj > j_index;
						put_result<RAW>(out_conreci, r, best, best_cnt, pass);
						if (STATS) { // <= 16 reads: three words of 16-bit (8-bit) fields
							st_a = (u32)nvalid | ((u32)total << 16);
							st_b = (u32)(rec_a + amb_a + rec_b + amb_b) | ((u32)(rec_a + rec_b) << 16);
							st_c = (u32)(amb_a + amb_b) | (pass ? 1u << 16 : 1u << 24);
						}
					}
				}
			}
			if (STATS) {
#define ARKS_ROW_ADD(v, ctrl) v += (u32)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true)
				ARKS_ROW_ADD(st_a, 0x111); ARKS_ROW_ADD(st_b, 0x111); ARKS_ROW_ADD(st_c, 0x111); // row_shr:1
				ARKS_ROW_ADD(st_a, 0x112); ARKS_ROW_ADD(st_b, 0x112); ARKS_ROW_ADD(st_c, 0x112); // row_shr:2
				ARKS_ROW_ADD(st_a, 0x114); ARKS_ROW_ADD(st_b, 0x114); ARKS_ROW_ADD(st_c, 0x114); // row_shr:4
				ARKS_ROW_ADD(st_a, 0x118); ARKS_ROW_ADD(st_b, 0x118); ARKS_ROW_ADD(st_c, 0x118); // row_shr:8
#undef ARKS_ROW_ADD
				if (lane == 15) {
					S.wstats[0] += st_a & 0xFFFFu;       // valid windows
					S.wstats[7] += st_a >> 16;           // all windows (bad = all - valid)
					S.wstats[2] += st_b & 0xFFFFu;       // found
					S.wstats[3] += st_b >> 16;           // recorded
					S.wstats[4] += st_c & 0xFFFFu;       // duplicate (value 0)
					S.wstats[5] += (st_c >> 16) & 0xFFu; // reads passing
					S.wstats[6] += st_c >> 24;           // reads failing
				}
			}
			ARKS_WAVE_SYNC();
			ARKS_SEC(11);
			cur = nxt;
		}
	}
	flush_mq();
#if defined(ARKS_PROFILE_SECTIONS) && !defined(ARKS_PROFILE_MEDIUM)
	if (!STATS && !RAW && lane_id == 0)
		for (int x = 0; x < 12; ++x)
			atomicAdd(&g_sec_cycles[x], sec_acc[x]);
#endif
	if (STATS) {
		ARKS_WAVE_SYNC();
		if (lane_id == 0) {
			u64* row = stats + kStatRow * (blockIdx.x & (kStatRows - 1));
			const u64 valid = S.wstats[0], all = S.wstats[7];
			if (valid) atomicAdd(row + 0, valid);
			if (all - valid) atomicAdd(row + 1, all - valid);
			if (S.wstats[2]) atomicAdd(row + 2, S.wstats[2]);
			if (S.wstats[3]) atomicAdd(row + 3, S.wstats[3]);
			if (S.wstats[4]) atomicAdd(row + 4, S.wstats[4]);
			if (S.wstats[5]) atomicAdd(row + 5, S.wstats[5]);
			if (S.wstats[6]) atomicAdd(row + 6, S.wstats[6]);
			if (all) atomicAdd(row + 7, all);
		}
	}
}

// ---- the sharded seed table: what a read asks, and what an owner answers -------------------------------
// seeds of read r (the G of map_reads_s_kernel): ceil(windows / w), 0 for a read that is not evaluated
__global__ void
seed_counts_kernel(
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k, int w, int* __restrict__ out)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	const int nwin = (eval && !eval[r]) ? 0 : (int)lens[r] - k + 1;
	out[r] = nwin > 0 ? (nwin + w - 1) / w : 0;
}

// canonical m-mer and owner rank of every seed, read-major (seed_off = exclusive prefix of the counts);
// ~0 = the seed holds an invalid base (no entries anywhere; owner 0)
template <int MM>
__global__ void
seeds_fill_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, const u64* __restrict__ word_off,
    const u32* __restrict__ lens, const uint8_t* __restrict__ eval, long n_reads, int k, int w, u32 n_owners,
    const long* __restrict__ seed_off, u64* __restrict__ out_cm, int* __restrict__ out_owner)
{
	typedef typename Mmer<MM>::type mm_t;
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	const int nwin = (eval && !eval[r]) ? 0 : (int)lens[r] - k + 1;
	const int G = nwin > 0 ? (nwin + w - 1) / w : 0;
	const u64 wb = word_off[r];
	const long s0 = seed_off[r];
	for (int gi = 0; gi < G; ++gi) {
		int q = (gi + 1) * w - 1;
		q = q < nwin - 1 ? q : nwin - 1;
		const u64 pos = wb * 32ull + (u64)q;
		const u32* nm = nmask + (pos >> 5);
		const u64 two = ((u64)nm[0] << 32) | (u64)nm[1];
		u64 cm = ~0ull;
		u32 own = 0;
		if (((two << (pos & 31)) >> (64 - MM)) == 0) {
			const mm_t mf = mmer_fw<MM>(codes, pos), mr = mmer_rc<MM>(mf);
			cm = (u64)(mf < mr ? mf : mr);
			own = seed_owner<MM>((mm_t)cm, n_owners);
		}
		out_cm[s0 + gi] = cm;
		out_owner[s0 + gi] = (int)own;
	}
}

// owner side: the entries of every asked m-mer in this rank's shard of the seed table
template <int MM>
__global__ void
seeds_probe_kernel(BIndexView bx, const u64* __restrict__ cm, long n, u64* __restrict__ ans)
{
	typedef typename Mmer<MM>::type mm_t;
	const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	u64 ent[2] = { 0, 0 };
	u64 a0 = 0, a1 = 0;
	if (cm[i] != ~0ull) {
		const u32 cnt = probe_minimizer_table<MM>(bx, (mm_t)cm[i], ent);
		if (cnt == kHnOverflow)
			a0 = kAnsOverflow;
		else if (cnt == kHnHeavy)
			a0 = kAnsHeavy;
		else {
			a0 = cnt >= 1 ? ent[0] : 0;
			a1 = cnt >= 2 ? ent[1] : 0;
		}
	}
	ans[2 * i] = a0;
	ans[2 * i + 1] = a1;
}

// the same for the seeds of several askers in one launch (arks_exchange): segment s = sg.src[s][0 .. n_s) -> sg.dst[s].
// (Round 5 tried it as a persistent grid of 8 / 16 / 32 waves per CU with four seeds per thread in flight, so that the
// askers' latency-bound map kernels keep their wave slots beside it: 163-165 ms per pass against 155 -- the step's time
// is the SUM of its kernels' times alone, whoever shares the device with whom; profiles/r09f_sharded_isolated.txt.)
template <int MM>
__global__ void __launch_bounds__(256)
seeds_probe_segs_kernel(BIndexView bx, ProbeSegs sg)
{
	typedef typename Mmer<MM>::type mm_t;
	const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= sg.end[sg.n_segs - 1])
		return;
	int s = 0;
	while (i >= sg.end[s]) // (a handful of segments: the askers of one owner)
		++s;
	const u64 j = i - (s ? sg.end[s - 1] : 0ull);
	const u64 cm = sg.src[s][j];
	u64 ent[2] = { 0, 0 };
	u64 a0 = 0, a1 = 0;
	if (cm != ~0ull) {
		const u32 cnt = probe_minimizer_table<MM, true>(bx, (mm_t)cm, ent);
		if (cnt == kHnOverflow)
			a0 = kAnsOverflow;
		else if (cnt == kHnHeavy)
			a0 = kAnsHeavy;
		else {
			a0 = cnt >= 1 ? ent[0] : 0;
			a1 = cnt >= 2 ? ent[1] : 0;
		}
	}
	*reinterpret_cast<ulonglong2*>(sg.dst[s] + 2 * j) = make_ulonglong2(a0, a1);
}

// ------------------------------------------------------------------------------------------------
// K4: pair gate and pair rule of chromiumRead.
// ------------------------------------------------------------------------------------------------
__global__ void
pair_gate_kernel(
    const uint8_t* __restrict__ pair_ok,
    const uint8_t* __restrict__ read_class,
    long n_pairs,
    uint8_t* __restrict__ eval)
{
	const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pairs)
		return;
	// class: bit 0 = checkReadSequence accepts the read, bit 1 = it is known to hold ACGT only; eval: 0 = the pair is
	// gated, else 1 | (class & 2) -- the map kernels skip the N masks of a tile all of whose reads have bit 1 (a caller's
	// own eval array of 0 / 1 says nothing about the bases: the masks are fetched)
	const uint8_t c0 = read_class[2 * p], c1 = read_class[2 * p + 1];
	const bool e = (pair_ok ? pair_ok[p] : 1) && (c0 & 1) && (c1 & 1);
	eval[2 * p] = e ? (uint8_t)(1 | (c0 & 2)) : (uint8_t)0;
	eval[2 * p + 1] = e ? (uint8_t)(1 | (c1 & 2)) : (uint8_t)0;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from arks_capi.cpp through arks_kernels.hpp)
// ------------------------------------------------------------------------------------------------
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

__global__ void
fold_stats_kernel(const u64* __restrict__ rows, u64* __restrict__ stats)
{
	const int c = threadIdx.x;
	if (c >= kStatRow)
		return;
	u64 v = 0;
	for (int r = 0; r < kStatRows; ++r)
		v += rows[r * kStatRow + c];
	if (v)
		atomicAdd(stats + c, v);
}

// arks_debug_set_medium_blocks(n) (include/arks_hip_debug.h): the medium kernel on n waves only, so that even a test's
// short queue gives every wave several reads per grab (tiles of several gathered reads: the path a long queue takes)
static std::atomic<unsigned> g_medium_blocks_cap{ 0 };
void
set_medium_blocks_cap(unsigned n)
{
	g_medium_blocks_cap.store(n, std::memory_order_relaxed);
}
static unsigned
medium_blocks(unsigned bb)
{
	const unsigned v = g_medium_blocks_cap.load(std::memory_order_relaxed);
	return v >= 1 && v < bb ? v : bb;
}

hipError_t
launch_map_reads(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens,
    const uint8_t* eval, long n_reads, double j_index, const KeyGeom& g, TableView t,
    const BIndexView& bx, int* out, u64* stats, u32* queue, u32* queue_count, int n_cu, hipStream_t st,
    bool raw, const uint8_t* gate_class, const uint8_t* gate_ok)
{
	if (n_reads <= 0)
		return hipSuccess;
	if (gate_class && (!bx.dense || raw || (n_reads & 1)))
		return hipErrorInvalidValue; // (the caller falls back to the gate launch: only the seed tile kernel works it out)
	if (raw && stats)
		return hipErrorInvalidValue; // the window counters of a shard are not the reference's
	// queue_count: 4 counters, then (64 bytes in) the partial rows of the statistics
	u64* const user_stats = stats;
	if (stats)
		stats = reinterpret_cast<u64*>(reinterpret_cast<char*>(queue_count) + 64);
	hipError_t e = hipMemsetAsync(queue_count, 0, kMapScratchBytes, st);
	if (e != hipSuccess)
		return e;
	// one wave per read at a time; enough resident waves to cover the memory latency
	const u64 want = ((u64)n_reads + 3) / 4;
	const u64 cap = (u64)(n_cu > 0 ? n_cu : 256) * 8ull;
	const unsigned b = (unsigned)(want < cap ? want : cap);
	const unsigned bs = (unsigned)(want < 256 ? want : 256); // slow path: the queue is short
#ifdef ARKS_DEBUG_KNOBS
	static const bool dbg_sync = std::getenv("ARKS_DEBUG_SYNC") != nullptr;
#else
	constexpr bool dbg_sync = false;
#endif
#define ARKS_DEBUG_STAGE(name)                                                                     \
	do {                                                                                           \
		if (dbg_sync) {                                                                            \
			std::fprintf(stderr, "[arks] map stage %s launched (dense %d stats %d)\n", name, bx.dense, stats != nullptr); \
			std::fflush(stderr);                                                                   \
			hipError_t de_ = hipStreamSynchronize(st);                                             \
			std::fprintf(stderr, "[arks] map stage %s: %s\n", name, hipGetErrorString(de_));       \
			std::fflush(stderr);                                                                   \
		}                                                                                          \
	} while (0)
#define ARKS_MAP_HASH(KWV, ST, RAWV)                                                               \
	do {                                                                                           \
		map_reads_kernel<KWV, ST, true, false, kMShort, RAWV><<<b, 256, 0, st>>>(                  \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
		map_reads_kernel<KWV, ST, false, false, kMShort, RAWV><<<bs, 256, 0, st>>>(                \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
	} while (0)
#define ARKS_MAP_B(KWV, ST, MMV, RAWV, DN)                                                         \
	do {                                                                                           \
		int per_cu = 0;                                                                            \
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(                                          \
		        &per_cu, map_reads_b_kernel<KWV, ST, false, MMV, RAWV, false>, 64, 0) != hipSuccess ||\
		    per_cu <= 0)                                                                           \
			per_cu = 8;                                                                            \
		const u64 res = (u64)(n_cu > 0 ? n_cu : 256) * (u64)per_cu;                                \
		const u64 wantw = ((u64)n_reads + 3) / 4;                                                  \
		const unsigned bb = (unsigned)(wantw < res ? wantw : res);                                 \
		if (DN) {                                                                                  \
			const u64 wants = ((u64)n_reads + sChunk - 1) / sChunk;                                \
			const u64 ress = (u64)(n_cu > 0 ? n_cu : 256) * 4ull * ARKS_SEED_WAVES;                \
			map_reads_s_kernel<KWV, ST, MMV, RAWV><<<(unsigned)(wants < ress ? wants : ress), 64, 0, st>>>( \
			    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bx, out, stats, queue,    \
			    queue + n_reads, queue_count, nullptr, nullptr, nullptr, nullptr, gate_class, gate_ok); \
		} else                                                                                     \
			map_reads_b_kernel<KWV, ST, false, MMV, RAWV, false><<<bb, 64, 0, st>>>(               \
			    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bx, out, stats, queue,    \
			    queue + n_reads, queue_count);                                                     \
		ARKS_DEBUG_STAGE("hot");                                                                   \
		map_reads_b_kernel<KWV, ST, true, MMV, RAWV, DN><<<medium_blocks(bb), 64, 0, st>>>(        \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bx, out, stats, queue,        \
		    queue + n_reads, queue_count);                                                         \
		ARKS_DEBUG_STAGE("medium");                                                                \
		map_reads_kernel<KWV, ST, false, true, MMV, RAWV><<<bs, 256, 0, st>>>(                     \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
		ARKS_DEBUG_STAGE("slow");                                                                  \
	} while (0)
#define ARKS_MAP_B_DN(KWV, ST, MMV, RAWV)                                                          \
	do {                                                                                           \
		if (bx.dense) ARKS_MAP_B(KWV, ST, MMV, RAWV, true);                                        \
		else ARKS_MAP_B(KWV, ST, MMV, RAWV, false);                                                \
	} while (0)
#define ARKS_MAP_B_ST(KWV, MMV)                                                                    \
	do {                                                                                           \
		if (raw) ARKS_MAP_B_DN(KWV, false, MMV, true);                                             \
		else if (stats) ARKS_MAP_B_DN(KWV, true, MMV, false);                                      \
		else ARKS_MAP_B_DN(KWV, false, MMV, false);                                                \
	} while (0)
#define ARKS_MAP_HASH_ST(KWV)                                                                      \
	do {                                                                                           \
		if (raw) ARKS_MAP_HASH(KWV, false, true);                                                  \
		else if (stats) ARKS_MAP_HASH(KWV, true, false);                                           \
		else ARKS_MAP_HASH(KWV, false, false);                                                     \
	} while (0)
	if (bx.enabled) {
		if (kw == 2 && bx.m == kMShort) ARKS_MAP_B_ST(2, kMShort);
		else if (kw == 2) ARKS_MAP_B_ST(2, kMLong);
		else if (bx.m == kMShort) ARKS_MAP_B_ST(3, kMShort);
		else ARKS_MAP_B_ST(3, kMLong);
	} else {
		if (kw == 2) ARKS_MAP_HASH_ST(2);
		else ARKS_MAP_HASH_ST(3);
	}
#undef ARKS_MAP_HASH_ST
#undef ARKS_MAP_HASH
#undef ARKS_MAP_B
#undef ARKS_MAP_B_DN
#undef ARKS_MAP_B_ST
	if (user_stats)
		fold_stats_kernel<<<1, 64, 0, st>>>(stats, user_stats);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

// The tail of bestContig (Arcs.cpp:996-1013) over votes that were gathered from index shards
// (put_result<true>; the caller has taken the maximum over the shards): total = every window of the
// read, NULL ones included (:962); accepted iff count / total > j_index in double (:1006).
__global__ void
resolve_votes_kernel(
    const u64* __restrict__ votes, const u32* __restrict__ lens, long n_reads, int k, double j_index,
    int* __restrict__ out)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_reads)
		return;
	const u64 v = votes[r];
	const int best_cnt = (int)(v >> 32);
	const int best = best_cnt > 0 ? (int)~(u32)v : 0;
	const int nwin = (int)lens[r] - k + 1;
	const int total = nwin > 0 ? nwin : 0;
	const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
	out[r] = maxj > j_index ? best : 0;
}

// acc[r] = max(acc[r], in[r]): folds the votes of another shard in (single-process drivers that copy
// votes between GPUs themselves; with one process per GPU the all-reduce(MAX) of RCCL does this)
__global__ void
max_votes_kernel(u64* __restrict__ acc, const u64* __restrict__ in, long n)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n) {
		const u64 a = acc[r], b = in[r];
		if (b > a)
			acc[r] = b;
	}
}

hipError_t
launch_seed_counts(const u32* lens, const uint8_t* eval, long n_reads, int k, int w, int* out, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	seed_counts_kernel<<<blocks_for((u64)n_reads, 256), 256, 0, st>>>(lens, eval, n_reads, k, w, out);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_seeds_fill(
    int mm, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval, long n_reads,
    int k, int w, u32 n_owners, const long* seed_off, u64* out_cm, int* out_owner, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	const unsigned b = blocks_for((u64)n_reads, 256);
	if (mm == kMShort)
		seeds_fill_kernel<kMShort><<<b, 256, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, seed_off, out_cm, out_owner);
	else
		seeds_fill_kernel<kMLong><<<b, 256, 0, st>>>(codes, nmask, word_off, lens, eval, n_reads, k, w, n_owners, seed_off, out_cm, out_owner);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_seeds_probe(int mm, const BIndexView& bx, const u64* cm, long n, u64* ans, hipStream_t st)
{
	if (n <= 0)
		return hipSuccess;
	const unsigned b = blocks_for((u64)n, 256);
	if (mm == kMShort)
		seeds_probe_kernel<kMShort><<<b, 256, 0, st>>>(bx, cm, n, ans);
	else
		seeds_probe_kernel<kMLong><<<b, 256, 0, st>>>(bx, cm, n, ans);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

// bestContig for a batch whose seed probes were answered beforehand (sharded seed table): the hot kernel with
// REMOTE answers, then the general kernels over the replicated minimizer table (bxg)
hipError_t
launch_seeds_probe_segs(int mm, const BIndexView& bx, const ProbeSegs& sg, hipStream_t st, int n_cu)
{
	if (sg.n_segs <= 0 || sg.n_segs > 65)
		return sg.n_segs == 0 ? hipSuccess : hipErrorInvalidValue;
	const u64 n = sg.end[sg.n_segs - 1];
	if (n == 0)
		return hipSuccess;
	(void)n_cu;
	const unsigned b = (unsigned)((n + 255) / 256);
	if (mm == kMShort)
		seeds_probe_segs_kernel<kMShort><<<b, 256, 0, st>>>(bx, sg);
	else
		seeds_probe_segs_kernel<kMLong><<<b, 256, 0, st>>>(bx, sg);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_map_reads_seeded(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens, const uint8_t* eval,
    long n_reads, double j_index, const KeyGeom& g, const BIndexView& bx, const BIndexView& bxg, const long* seed_off,
    const u64* ans, int* out, u64* stats, u32* queue, u32* queue_count, int n_cu, hipStream_t st, const u32* seed_slot,
    const u32* chunk_off, bool scratch_zeroed)
{
	static_assert(sChunk == 56, "arks_shard.hip (kBkChunk) numbers the seeds by chunks of the seed tile kernel");
	if (n_reads <= 0)
		return hipSuccess;
	u64* const user_stats = stats;
	if (stats)
		stats = reinterpret_cast<u64*>(reinterpret_cast<char*>(queue_count) + 64);
	// (scratch_zeroed: the caller's previous kernel on this stream zeroed the block -- arks_exchange's bucket kernel)
	hipError_t e = scratch_zeroed ? hipSuccess : hipMemsetAsync(queue_count, 0, kMapScratchBytes, st);
	if (e != hipSuccess)
		return e;
	const u64 cus = (u64)(n_cu > 0 ? n_cu : 256);
	const u64 wants = ((u64)n_reads + sChunk - 1) / sChunk, ress = cus * 4ull * ARKS_SEED_WAVES;
	const unsigned bh = (unsigned)(wants < ress ? wants : ress);
	const u64 wantw = ((u64)n_reads + 3) / 4;
	const unsigned bb = (unsigned)(wantw < cus * 16 ? wantw : cus * 16);
	const u64 want = ((u64)n_reads + 3) / 4;
	const unsigned bs = (unsigned)(want < 256 ? want : 256);
	const TableView none{ nullptr, 0 };
#define ARKS_SEEDED(KWV, ST, MMV)                                                                  \
	do {                                                                                           \
		map_reads_s_kernel<KWV, ST, MMV, false, true><<<bh, 64, 0, st>>>(                          \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bx, out, stats, queue,        \
		    queue + n_reads, queue_count, seed_off, ans, seed_slot, chunk_off);                    \
		map_reads_b_kernel<KWV, ST, true, MMV, false, false><<<medium_blocks(bb), 64, 0, st>>>(    \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bxg, out, stats, queue,       \
		    queue + n_reads, queue_count);                                                         \
		map_reads_kernel<KWV, ST, false, true, MMV, false><<<bs, 256, 0, st>>>(                    \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, none, bxg, out, stats, queue, \
		    queue_count);                                                                          \
	} while (0)
#define ARKS_SEEDED_ST(KWV, MMV)                                                                   \
	do {                                                                                           \
		if (stats) ARKS_SEEDED(KWV, true, MMV);                                                    \
		else ARKS_SEEDED(KWV, false, MMV);                                                         \
	} while (0)
	if (kw == 2 && bx.m == kMShort) ARKS_SEEDED_ST(2, kMShort);
	else if (kw == 2) ARKS_SEEDED_ST(2, kMLong);
	else if (bx.m == kMShort) ARKS_SEEDED_ST(3, kMShort);
	else ARKS_SEEDED_ST(3, kMLong);
#undef ARKS_SEEDED
#undef ARKS_SEEDED_ST
	if (user_stats)
		fold_stats_kernel<<<1, 64, 0, st>>>(stats, user_stats);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_max_votes(u64* acc, const u64* in, long n, hipStream_t st)
{
	if (n <= 0)
		return hipSuccess;
	max_votes_kernel<<<blocks_for((u64)n, 256), 256, 0, st>>>(acc, in, n);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_resolve_votes(
    const u64* votes, const u32* lens, long n_reads, int k, double j_index, int* out, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	resolve_votes_kernel<<<blocks_for((u64)n_reads, 256), 256, 0, st>>>(votes, lens, n_reads, k, j_index, out);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_pair_gate(const uint8_t* pair_ok, const uint8_t* read_class, long n_pairs, uint8_t* eval, hipStream_t st)
{
	if (n_pairs <= 0)
		return hipSuccess;
	pair_gate_kernel<<<blocks_for((u64)n_pairs, 256), 256, 0, st>>>(pair_ok, read_class, n_pairs, eval);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

#ifdef ARKS_MEDIUM_DIAG
void
read_medium_diag(unsigned long long* out16)
{
	(void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_med_diag), sizeof(unsigned long long) * 16);
	unsigned long long z[16] = { 0 };
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_med_diag), z, sizeof z);
}
#endif

#ifdef ARKS_PROFILE_SECTIONS
void
read_section_cycles(unsigned long long* out16)
{
	(void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sec_cycles), sizeof(unsigned long long) * 16);
	unsigned long long z[16] = { 0 };
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_sec_cycles), z, sizeof z);
}
#endif

} // namespace arks
