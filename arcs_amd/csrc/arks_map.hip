// arks_map.hip -- read-mapping kernels of libarks_hip (gfx950 / CDNA4, wave64): bestContig
// (Arcs/Arcs.cpp:939-1014) for a batch, the pair rule of chromiumRead (:1264-1292) and the
// (barcode, contig end) accumulation.  Reference behaviour restated, never its code.
#include "arks_kernels.hpp"

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

// per-wave counters of arks_map_stats (uniform across the lanes of a wave)
struct WaveStats
{
	u64 valid, bad, found, rec, dup, pass, fail, win;
};

// value of window p of a read: -2 = NULL k-mer, -1 = absent, 0 = ambiguous, > 0 = contig end;
// with QUIRK = false a palindromic window is not resolved but reported as -4.
// BMODE = true: the locality index (serial lookup) instead of the plain hash table.
template <int KW, bool QUIRK, bool BMODE>
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
		return bindex_lookup_serial<KW>(bx, g, codes, wbase * 32ull + (u64)p, f, r);
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
template <int KW, bool STATS, bool BMODE>
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
			const int v = p < nwin ? window_value<KW, true, BMODE>(codes, nmask, wbase, p, g, t, bx) : -3;
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

// FAST = true : the hot kernel.  Grid-stride over all reads; a read that needs one of the rare
//               paths (a reverse-complement palindrome window, whose key takes the reference's
//               damaged branch; or more than 64 * kMaxPass windows) is appended to `queue`
//               untouched, which keeps those paths' registers out of this kernel.
// FAST = false: the same algorithm with every path, over the reads listed in `queue`.
template <int KW, bool STATS, bool FAST, bool BMODE>
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
				out_conreci[r] = 0;
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
				vote_long_read<KW, STATS, BMODE>(codes, nmask, wbase, nwin, g, t, bx, lane, rs, best, best_cnt);
		} else {
			int vals[kMaxPass];
#pragma unroll
			for (int ps = 0; ps < kMaxPass; ++ps) {
				int v = -3;
				if (ps * 64 < nwin) { // wave-uniform
					const int p = ps * 64 + lane;
					if (p < nwin)
						v = window_value<KW, !FAST, BMODE>(codes, nmask, wbase, p, g, t, bx);
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
			out_conreci[r] = pass ? best : 0;
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
		if (ws.valid) atomicAdd(stats + 0, ws.valid);
		if (ws.bad) atomicAdd(stats + 1, ws.bad);
		if (ws.found) atomicAdd(stats + 2, ws.found);
		if (ws.rec) atomicAdd(stats + 3, ws.rec);
		if (ws.dup) atomicAdd(stats + 4, ws.dup);
		if (ws.pass) atomicAdd(stats + 5, ws.pass);
		if (ws.fail) atomicAdd(stats + 6, ws.fail);
		if (ws.win) atomicAdd(stats + 7, ws.win);
	}
}

// ------------------------------------------------------------------------------------------------
// K3b: read mapping over the locality index -- the hot kernel.
//
// One wave per read at a time.  Instead of one table probe per window the wave
//   1. hashes every 15-mer of the read once (lanes = positions) and takes the sliding-window
//      minimum by doubling through LDS  -> the minimizer position of every window;
//   2. compacts the run heads (first window of each run of windows that share a minimizer; a
//      151-bp read has ~5) and lets one lane per run walk the minimizer table;
//   3. for every distinct diagonal (text position of read base 0, strand) the runs propose, XORs
//      the read with the text once (lanes = 32-base words) and lets every window test its own
//      2k-bit span of that mismatch stream -- exact, so the answer is still key equality
//      (Arcs/Arcs.h:153-156); the value comes from the position's visited / ambiguous bits and owner;
//   4. sends windows under a heavy minimizer to the exact fallback table, and reads that may hold a
//      reverse-complement palindrome (detected through the mirrored minimizer) to the slow kernel;
//   5. votes exactly as bestContig does (Arcs.cpp:996-1013).
// HBM traffic per read: ~5 random 8-B minimizer entries + ~50 B of text and bit words, instead of
// ~100 random 64-B lines.
// ------------------------------------------------------------------------------------------------
constexpr int kFastMaxLen = 288;               // reads up to this length take the cooperative path
constexpr int kFastPasses = (kFastMaxLen + 63) / 64;
constexpr int kFastWords = kFastMaxLen / 32;   // 9

struct WaveLds
{
	u32 ord[kFastMaxLen + 96]; // ordering values; after the sliding minimum: minimizer position per window
	u32 mm[kFastMaxLen];       // [29:0] canonical 15-mer, [30] strand, [31] heavy
	int vals[kFastMaxLen];     // window values
	u64 diff[kFastWords + 3];  // read XOR text along the diagonal under test
	u64 rc[kFastWords + 3];    // reverse complement of the read
	unsigned short heads[kFastMaxLen];
};

// lanes of one wave communicate through LDS: order the compiler's view of it
#define ARKS_WAVE_SYNC()                                                                           \
	do {                                                                                           \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                     \
		__builtin_amdgcn_wave_barrier();                                                           \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                     \
	} while (0)

template <int KW, bool STATS>
__global__ void __launch_bounds__(256)
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
    u32* __restrict__ queue,
    u32* __restrict__ queue_count)
{
	__shared__ WaveLds lds_all[4];
	WaveLds& S = lds_all[threadIdx.x >> 6];
	const int lane = threadIdx.x & 63;
	const u64 lane_lt = (1ull << lane) - 1ull;
	const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const long n_waves = ((long)gridDim.x * blockDim.x) >> 6;
	WaveStats ws = { 0, 0, 0, 0, 0, 0, 0, 0 };
	const int k = g.k, w = bx.w;
	(void)wave;
	(void)n_waves;

	// reads are handed out in chunks through a device counter: hits, misses and rare paths cost
	// very different amounts, a static split leaves a long tail
	constexpr long kChunk = 16;
	for (;;) {
	long chunk0 = 0;
	if (lane == 0)
		chunk0 = (long)atomicAdd(queue_count + 1, (u32)kChunk);
	chunk0 = __shfl(chunk0, 0);
	if (chunk0 >= n_reads)
		break;
	const long chunk1 = chunk0 + kChunk < n_reads ? chunk0 + kChunk : n_reads;
	for (long r = chunk0; r < chunk1; ++r) {
		if (eval && !eval[r]) {
			if (lane == 0)
				out_conreci[r] = 0;
			continue;
		}
		const int L = (int)lens[r];
		const int nwin = L - k + 1;
		if (L > kFastMaxLen) {
			if (lane == 0)
				queue[atomicAdd(queue_count, 1u)] = (u32)r;
			continue;
		}
		int best = 0, best_cnt = 0;
		WaveStats rs = { 0, 0, 0, 0, 0, 0, 0, 0 };
		bool redo = false;
		if (nwin > 0) {
			const u64 wbase = word_off[r];
			const u64 rbase = wbase * 32ull;
			const int npos = L - kM + 1;
			// ---- 1. ordering value of every 15-mer ------------------------------------------------
			for (int i = lane; i < npos + 96; i += 64) {
				u32 o = 0xFFFFFFFFu;
				if (i < npos) {
					const u32 mf = mmer_fw(codes, rbase + (u64)i);
					const u32* nm = nmask + wbase + (u64)(i >> 5);
					const int t = i & 31;
					const u32 bits = (nm[0] << t) | ((nm[1] >> 1) >> (31 - t));
					const u32 mr = mmer_rc(mf);
					const u32 cm = mf < mr ? mf : mr;
					S.mm[i] = cm | ((mf < mr ? 1u : 0u) << 30);
					if ((bits >> (32 - kM)) == 0)
						o = (mmer_order(cm) << 9) | (u32)i;
				}
				S.ord[i] = o;
			}
			ARKS_WAVE_SYNC();
			// ---- 2. sliding minimum over w positions, doubling in place ---------------------------
			int span = 1;
			for (;;) {
				const int step = (2 * span <= w) ? span : (w - span);
				if (step <= 0)
					break;
				u32 v[kFastPasses];
#pragma unroll
				for (int t = 0; t < kFastPasses; ++t) {
					const int i = t * 64 + lane;
					v[t] = 0xFFFFFFFFu;
					if (i < npos) { // the padding beyond npos stays 0xFFFFFFFF
						const u32 a = S.ord[i], b = S.ord[i + step];
						v[t] = a < b ? a : b;
					}
				}
				ARKS_WAVE_SYNC();
#pragma unroll
				for (int t = 0; t < kFastPasses; ++t)
					if (t * 64 + lane < npos)
						S.ord[t * 64 + lane] = v[t];
				ARKS_WAVE_SYNC();
				if (2 * span > w)
					break;
				span *= 2;
			}
			// ---- 3. windows: validity, minimizer position, run heads ------------------------------
			int nheads = 0;
			u32 carry = 0xFFFFu;
			for (int base = 0; base < nwin; base += 64) {
				const int p = base + lane;
				const bool in = p < nwin;
				bool invalid = false;
				u32 q = 0xFFFFu;
				if (in) {
					invalid = window_has_invalid<KW>(nmask, wbase, p, k);
					if (!invalid)
						q = S.ord[p] & 511u;
				}
				u32 qprev = __shfl_up(q, 1);
				if (lane == 0)
					qprev = carry;
				carry = __shfl(q, 63);
				const bool head = in && !invalid && q != qprev;
				if (in) {
					S.vals[p] = invalid ? -2 : -1;
					if (!invalid && !(k & 1)) {
						// a reverse-complement palindrome has its minimizer twice, mirrored about its
						// centre: necessary condition, checked exactly by the slow kernel
						const int qm = 2 * p + (k - kM) - (int)q;
						if (mmer_order(S.mm[qm] & kMmerMask) == mmer_order(S.mm[q] & kMmerMask))
							redo = true;
					}
				}
				const u64 hb = __ballot(head);
				if (head)
					S.heads[nheads + __popcll(hb & lane_lt)] = (unsigned short)p;
				nheads += __popcll(hb);
			}
			redo = __ballot(redo) != 0;
			ARKS_WAVE_SYNC();
			// (window p's minimizer position stays in ord[p])
			// ---- 4. run heads walk the minimizer table; distinct diagonals get verified -----------
			bool rc_ready = false;
			u64 last_d = ~0ull;
			bool last_same = false;
			for (int hb0 = 0; hb0 < nheads && !redo; hb0 += 64) {
				const int h = hb0 + lane;
				bool active = h < nheads;
				u32 q = 0, cm = 0, rstrand = 0;
				u64 slot = 0;
				if (active) {
					const int ph = S.heads[h];
					q = S.ord[ph] & 511u;
					const u32 m = S.mm[q];
					cm = m & kMmerMask;
					rstrand = (m >> 30) & 1u;
					slot = mtab_home(cm, bx.mtab_cap);
				}
				bool have = false;
				u64 d = 0;
				bool same = false;
				for (;;) {
					if (active && !have) {
						for (;;) {
							const u64 e = bx.mtab[slot];
							if (!(e >> 63)) {
								active = false;
								break;
							}
							slot = (slot + 1 == bx.mtab_cap) ? 0 : slot + 1;
							if (((u32)(e >> 32) & kMmerMask) != cm)
								continue;
							const u32 tpos = (u32)e;
							if (tpos == kHeavyPos) {
								atomicOr(&S.mm[q], 0x80000000u);
								active = false;
							} else {
								same = ((u32)(e >> 62) & 1u) == rstrand;
								// text position of base 0 of the read (same strand) / of its
								// reverse complement (opposite strand)
								d = same ? (u64)tpos - (u64)q : (u64)tpos + (u64)(kM + (int)q) - (u64)L;
								have = true;
							}
							break;
						}
					}
					const u64 hm = __ballot(have);
					if (!hm)
						break;
					const int src = __ffsll((long long)hm) - 1;
					const u64 d0 = __shfl(d, src);
					const bool same0 = __shfl((int)same, src) != 0;
					if (have && d == d0 && same == same0)
						have = false; // consumed
					if (d0 == last_d && same0 == last_same)
						continue;
					last_d = d0;
					last_same = same0;
					// ---- verify diagonal (d0, same0) -------------------------------------------
					const int nw = (L + 31) >> 5;
					if (!same0 && !rc_ready) {
						// reverse complement of the read, packed like the read
						if (lane <= nw) {
							const int sh = 2 * (32 * nw - L); // < 64
							const int a = nw - 1 - lane, b = nw - 2 - lane;
							const u64 ra = a >= 0 ? ~rev_groups(codes[wbase + (u64)a]) : 0ull;
							const u64 rb = b >= 0 ? ~rev_groups(codes[wbase + (u64)b]) : 0ull;
							S.rc[lane] = lane < nw ? funnel_l(ra, rb, sh) : 0ull;
						}
						rc_ready = true;
						ARKS_WAVE_SYNC();
					}
					if (lane < nw + 2) {
						u64 x = 0;
						if (lane < nw) {
							const u64 rw = same0 ? codes[wbase + (u64)lane] : S.rc[lane];
							const u64 tp = d0 + 32ull * (u64)lane;
							const u64* tsrc = bx.codes + (tp >> 5);
							const u64 tw = funnel_l(tsrc[0], tsrc[1], (int)(tp & 31) * 2);
							x = rw ^ tw;
							const int rem = L - 32 * lane; // bases of the read in this word
							if (rem < 32)
								x &= ~(~0ull >> (2 * rem));
						}
						S.diff[lane] = x;
					}
					ARKS_WAVE_SYNC();
					for (int base = 0; base < nwin; base += 64) {
						const int p = base + lane;
						if (p < nwin && S.vals[p] == -1) {
							const int pp = same0 ? p : (L - k - p);
							const int wi = pp >> 5, sft = (pp & 31) * 2;
							u64 any = 0;
#pragma unroll
							for (int j = 0; j < KW; ++j)
								any |= funnel_l(S.diff[wi + j], S.diff[wi + j + 1], sft) & g.mask[j];
							if (any == 0) {
								const u64 t = d0 + (u64)pp;
								if (bit_at(bx.visited, t))
									S.vals[p] = bit_at(bx.ambig, t) ? 0 : (int)bx.word_owner[t >> 5];
							}
						}
					}
					ARKS_WAVE_SYNC();
				}
			}
			// ---- 5. windows under a heavy minimizer: exact fallback table ---------------------------
			if (!redo) {
				for (int base = 0; base < nwin; base += 64) {
					const int p = base + lane;
					if (p < nwin && S.vals[p] == -1 && (S.mm[S.ord[p] & 511u] >> 31)) {
						const Key<KW> f = window_key<KW>(codes, wbase, p, g);
						const Key<KW> rk = key_revcomp(f, g);
						Key<KW> c;
						const bool lt = key_less(f, rk);
#pragma unroll
						for (int j = 0; j < KW; ++j)
							c.w[j] = lt ? f.w[j] : rk.w[j];
						S.vals[p] = fallback_lookup<KW>(bx, c);
					}
				}
				ARKS_WAVE_SYNC();
			}
			// ---- 6. counters and vote ----------------------------------------------------------------
			if (!redo) {
				int vals[kFastPasses];
#pragma unroll
				for (int ps = 0; ps < kFastPasses; ++ps) {
					int v = -3;
					if (ps * 64 < nwin) {
						const int p = ps * 64 + lane;
						if (p < nwin)
							v = S.vals[p];
						if (STATS) {
							rs.bad += __popcll(__ballot(v == -2));
							rs.valid += __popcll(__ballot(v >= -1));
							rs.found += __popcll(__ballot(v >= 0));
							rs.rec += __popcll(__ballot(v > 0));
							rs.dup += __popcll(__ballot(v == 0));
						}
					}
					vals[ps] = v > 0 ? v : 0;
				}
				for (;;) {
					int m = 0x7FFFFFFF;
#pragma unroll
					for (int ps = 0; ps < kFastPasses; ++ps)
						m = (vals[ps] != 0 && vals[ps] < m) ? vals[ps] : m;
					m = wave_min_i32(m);
					if (m == 0x7FFFFFFF)
						break;
					int cnt = 0;
#pragma unroll
					for (int ps = 0; ps < kFastPasses; ++ps) {
						const bool is = vals[ps] == m;
						cnt += __popcll(__ballot(is));
						vals[ps] = is ? 0 : vals[ps];
					}
					if (cnt > best_cnt) {
						best_cnt = cnt;
						best = m;
					}
				}
			}
			ARKS_WAVE_SYNC();
		}
		if (redo) {
			if (lane == 0)
				queue[atomicAdd(queue_count, 1u)] = (u32)r;
			continue;
		}
		const int total = nwin > 0 ? nwin : 0;
		const double maxj = best_cnt > 0 ? (double)best_cnt / (double)total : 0.0;
		const bool pass = maxj > j_index;
		if (lane == 0)
			out_conreci[r] = pass ? best : 0;
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
	}
	if (STATS && lane == 0) {
		if (ws.valid) atomicAdd(stats + 0, ws.valid);
		if (ws.bad) atomicAdd(stats + 1, ws.bad);
		if (ws.found) atomicAdd(stats + 2, ws.found);
		if (ws.rec) atomicAdd(stats + 3, ws.rec);
		if (ws.dup) atomicAdd(stats + 4, ws.dup);
		if (ws.pass) atomicAdd(stats + 5, ws.pass);
		if (ws.fail) atomicAdd(stats + 6, ws.fail);
		if (ws.win) atomicAdd(stats + 7, ws.win);
	}
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
	const uint8_t e = ((pair_ok ? pair_ok[p] : 1) && read_class[2 * p] && read_class[2 * p + 1]) ? 1 : 0;
	eval[2 * p] = e;
	eval[2 * p + 1] = e;
}

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

// imap[(barcode, conreci)] += n   (keys are never 0 because conreci >= 1)
__device__ inline bool
imap_add(u64* keys, u32* counts, u64 cap, u64 key, u32 n)
{
	u64 s = mulhi64(mix64(key), cap);
	for (u64 probes = 0; probes < cap; ++probes) {
		u64 cur = __hip_atomic_load(keys + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			u64 expect = 0;
			if (__hip_atomic_compare_exchange_strong(
			        keys + s, &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
				cur = key;
			else
				cur = expect;
		}
		if (cur == key) {
			atomicAdd(counts + s, n);
			return true;
		}
		s = (s + 1 == cap) ? 0 : s + 1;
	}
	return false;
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
    u64* __restrict__ imap_keys,
    u32* __restrict__ imap_counts,
    u64 imap_cap,
    u32* __restrict__ imap_overflow,
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
	if (imap_keys == nullptr)
		return;
	const u64 key = ok ? (((u64)barcode_id[p] << 32) | (u32)agreed) : 0ull;
	const u64 prev = __shfl_up(key, 1);
	const bool head = lane == 0 || key != prev;
	const u64 heads = __ballot(head);
	if (head && key != 0) {
		const u64 later = lane == 63 ? 0ull : (heads >> (lane + 1));
		const int run = later ? (__ffsll((long long)later)) : (64 - lane);
		if (!imap_add(imap_keys, imap_counts, imap_cap, key, (u32)run))
			atomicOr(imap_overflow, 1u);
	}
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

hipError_t
launch_map_reads(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens,
    const uint8_t* eval, long n_reads, double j_index, const KeyGeom& g, TableView t,
    const BIndexView& bx, int* out, u64* stats, u32* queue, u32* queue_count, int n_cu, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	hipError_t e = hipMemsetAsync(queue_count, 0, 2 * sizeof(u32), st);
	if (e != hipSuccess)
		return e;
	// one wave per read at a time; enough resident waves to cover the memory latency
	const u64 want = ((u64)n_reads + 3) / 4;
	const u64 cap = (u64)(n_cu > 0 ? n_cu : 256) * 8ull;
	const unsigned b = (unsigned)(want < cap ? want : cap);
	const unsigned bs = (unsigned)(want < 256 ? want : 256); // slow path: the queue is short
#define ARKS_MAP_HASH(KWV, ST)                                                                     \
	do {                                                                                           \
		map_reads_kernel<KWV, ST, true, false><<<b, 256, 0, st>>>(                                 \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
		map_reads_kernel<KWV, ST, false, false><<<bs, 256, 0, st>>>(                               \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
	} while (0)
#define ARKS_MAP_B(KWV, ST)                                                                        \
	do {                                                                                           \
		int per_cu = 0;                                                                            \
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(                                          \
		        &per_cu, map_reads_b_kernel<KWV, ST>, 256, 0) != hipSuccess || per_cu <= 0)        \
			per_cu = 4;                                                                            \
		const u64 res = (u64)(n_cu > 0 ? n_cu : 256) * (u64)per_cu;                                \
		const unsigned bb = (unsigned)(want < res ? want : res);                                   \
		map_reads_b_kernel<KWV, ST><<<bb, 256, 0, st>>>(                                           \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, bx, out, stats, queue,        \
		    queue_count);                                                                          \
		map_reads_kernel<KWV, ST, false, true><<<bs, 256, 0, st>>>(                                \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, bx, out, stats, queue,     \
		    queue_count);                                                                          \
	} while (0)
	if (bx.enabled) {
		if (kw == 2) {
			if (stats) ARKS_MAP_B(2, true); else ARKS_MAP_B(2, false);
		} else {
			if (stats) ARKS_MAP_B(3, true); else ARKS_MAP_B(3, false);
		}
	} else {
		if (kw == 2) {
			if (stats) ARKS_MAP_HASH(2, true); else ARKS_MAP_HASH(2, false);
		} else {
			if (stats) ARKS_MAP_HASH(3, true); else ARKS_MAP_HASH(3, false);
		}
	}
#undef ARKS_MAP_HASH
#undef ARKS_MAP_B
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

hipError_t
launch_pairs(
    const int* conreci, const uint8_t* pair_ok, const u32* barcode_id, long n_pairs, int* out_pair,
    u64* imap_keys, u32* imap_counts, u64 imap_cap, u32* imap_overflow, u64* stored, hipStream_t st)
{
	if (n_pairs <= 0)
		return hipSuccess;
	pairs_kernel<<<blocks_for((u64)n_pairs, 256), 256, 0, st>>>(
	    conreci, pair_ok, barcode_id, n_pairs, out_pair, imap_keys, imap_counts, imap_cap,
	    imap_overflow, stored);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

} // namespace arks
