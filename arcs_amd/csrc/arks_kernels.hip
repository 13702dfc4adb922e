// arks_kernels.hip -- the HIP kernels of libarks_hip (gfx950 / CDNA4, wave64).
//
//   K0  pack_kernel           ASCII -> 2-bit codes + N-mask (+ checkReadSequence class)
//   K1  visit_kernel          the reference's index-build visit rule (i += k on a NULL k-mer)
//   K2  insert_kernel         owner-or-0 insertion of every visited contig-end k-mer
//   K2s build_stats_kernel    second pass for the "removed" counter
//   K3  map_reads_kernel      per read: window keys -> table probe -> vote   (the hot kernel)
//   K4  pair_gate_kernel / pairs_kernel   pair rule + (barcode, contig end) accumulation
//
// Reference behaviour restated (never its code): Arcs/Arcs.cpp:869-929 (mapKmers), :939-1014
// (bestContig), :1264-1292 (pair rule), :366-389 (checkReadSequence),
// Common/ReadsProcessor.cpp:376-535 (prepSeq).
#include "arks_kernels.hpp"

namespace arks {

// ------------------------------------------------------------------------------------------------
// K0: packing.  One thread per 32-base word.
// ------------------------------------------------------------------------------------------------
// class of an input byte: 0..3 = A C G T (either case), 4 = N/n, 5 = anything else
// (prepSeq's LUTs accept exactly ACGTacgt, Common/ReadsProcessor.cpp:39-317; checkReadSequence
// upper-cases first, Arcs/Arcs.cpp:373)
__device__ __forceinline__ u32
base_class(u32 ch)
{
	const u32 c = (ch >= 'a' && ch <= 'z') ? ch - 32u : ch;
	u32 r = 5;
	r = (c == 'A') ? 0u : r;
	r = (c == 'C') ? 1u : r;
	r = (c == 'G') ? 2u : r;
	r = (c == 'T') ? 3u : r;
	r = (c == 'N') ? 4u : r;
	return r;
}

__global__ void
pack_kernel(
    const uint8_t* __restrict__ ascii,
    const u64* __restrict__ offsets,
    const u32* __restrict__ lens,
    const u64* __restrict__ word_off,
    long n_seqs,
    u64 total_words,
    u64* __restrict__ codes,
    u32* __restrict__ nmask,
    u32* __restrict__ seq_n_count,  // per sequence: number of N/n (may be NULL)
    u32* __restrict__ seq_other)    // per sequence: != 0 if any byte outside ACGTN (may be NULL)
{
	const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= total_words)
		return;
	// sequence that owns word w: last r with word_off[r] <= w
	long lo = 0, hi = n_seqs - 1;
	while (lo < hi) {
		const long mid = (lo + hi + 1) >> 1;
		if (word_off[mid] <= w)
			lo = mid;
		else
			hi = mid - 1;
	}
	const long r = lo;
	const u64 first = (w - word_off[r]) * 32ull;
	const u32 len = lens[r];
	if (first >= len) { // zero-length sequence sharing an offset, or padding
		codes[w] = 0;
		nmask[w] = 0;
		return;
	}
	const uint8_t* src = ascii + offsets[r] + first;
	const u32 n = (len - first) < 32u ? (u32)(len - first) : 32u;
	u64 c = 0;
	u32 m = 0, nn = 0, other = 0;
	for (u32 i = 0; i < n; ++i) {
		const u32 cls = base_class(src[i]);
		c |= (u64)(cls < 4u ? cls : 0u) << (62 - 2 * i);
		m |= (cls >= 4u ? 1u : 0u) << (31 - i);
		nn += cls == 4u;
		other |= cls == 5u;
	}
	codes[w] = c;
	nmask[w] = m;
	if (seq_n_count && nn)
		atomicAdd(seq_n_count + r, nn);
	if (seq_other && other)
		atomicOr(seq_other + r, 1u);
}

// checkReadSequence, Arcs/Arcs.cpp:366-389: only ACGTN, and (double)N / (double)len <= 0.02
__global__ void
read_class_kernel(
    const u32* __restrict__ lens,
    const u32* __restrict__ seq_n_count,
    const u32* __restrict__ seq_other,
    long n,
    uint8_t* __restrict__ out)
{
	const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n)
		return;
	const double ar = (double)seq_n_count[r] / (double)lens[r]; // 0/0 = NaN -> "> 0.02" false
	out[r] = (seq_other[r] == 0 && !(ar > 0.02)) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// K1: visit rule of mapKmers (Arcs/Arcs.cpp:887-926).  One thread per contig end: walks the
// end's N-mask; a valid window is visited and i advances by 1, a NULL window makes i jump by k
// (so valid windows inside the jumped span are NOT indexed).  Emits a bit per visited start.
// ------------------------------------------------------------------------------------------------
// first position >= from (relative to the end's first base) whose N-mask bit is set, or len
__device__ inline int
next_invalid(const u32* __restrict__ nm, int from, int len)
{
	int w = from >> 5;
	const int nw = (len + 31) >> 5;
	if (w >= nw)
		return len;
	u32 bits = nm[w] & (0xFFFFFFFFu >> (from & 31));
	while (bits == 0) {
		if (++w >= nw)
			return len;
		bits = nm[w];
	}
	const int pos = (w << 5) + __clz((int)bits);
	return pos < len ? pos : len;
}

__global__ void
visit_kernel(
    const u32* __restrict__ nmask,
    const u64* __restrict__ word_off,
    const u32* __restrict__ lens,
    long n_ends,
    int k,
    u32* __restrict__ visited, // zero-initialised, same indexing as nmask
    u64* __restrict__ counters) // [0] += null k-mers, [1] += ends shorter than k
{
	const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n_ends)
		return;
	const int len = (int)lens[e];
	if (len < k) { // Arcs.cpp:877-882
		atomicAdd(counters + 1, 1ull);
		return;
	}
	const u32* nm = nmask + word_off[e];
	u32* vis = visited + word_off[e];
	const int last = len - k; // last window start
	u64 nulls = 0;
	int i = 0;
	int nb = next_invalid(nm, 0, len);
	while (i <= last) {
		if (nb >= i + k) {
			// windows i .. min(nb - k, last) are all valid: visited consecutively
			int e2 = nb - k;
			e2 = e2 > last ? last : e2;
			for (int p = i; p <= e2;) { // set bits [p, e2] a word at a time
				const int w = p >> 5, b = p & 31;
				int n = 32 - b;
				n = (e2 - p + 1) < n ? (e2 - p + 1) : n;
				const u32 m = (n == 32) ? 0xFFFFFFFFu : (((1u << n) - 1u) << (32 - b - n));
				vis[w] |= m; // this thread owns the end's words
				p += n;
			}
			i = e2 + 1;
		} else {
			nulls++;
			i += k; // Arcs.cpp:923
			if (nb < i)
				nb = next_invalid(nm, i, len);
		}
	}
	if (nulls)
		atomicAdd(counters + 0, nulls);
}

__global__ void
popcount_kernel(const u32* __restrict__ words, u64 n, u64* __restrict__ out)
{
	u64 acc = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		acc += (u64)__popc(words[i]);
	for (int off = 32; off > 0; off >>= 1)
		acc += __shfl_down(acc, off);
	if ((threadIdx.x & 63) == 0 && acc)
		atomicAdd(out, acc);
}

// ------------------------------------------------------------------------------------------------
// K2: insertion.  One thread per visited window.  State word protocol (agent scope):
//   EMPTY -CAS-> LOCKED (winner writes the key) -release-> value+1 ; equal key from another end
//   turns value+1 into 0+1 (once, counted).  The owner-or-0 rule is commutative, so the result is
//   independent of execution order, unlike the serial loop of Arcs.cpp:903-920 it replaces.
// ------------------------------------------------------------------------------------------------
template <int KW>
__global__ void
insert_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ visited,
    const u64* __restrict__ word_off, // n_ends + 1 entries
    long n_ends,
    u64 total_words,
    KeyGeom g,
    TableView t,
    u64* __restrict__ counters) // [2] += new keys, [3] += keys that lost their unique owner
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x; // global base position
	const u64 w = pos >> 5;
	bool active = w < total_words;
	if (active)
		active = (visited[w] >> (31 - (pos & 31))) & 1u;
	u32 n_new = 0, n_lost = 0;
	if (active) {
		long lo = 0, hi = n_ends - 1; // end that owns word w
		while (lo < hi) {
			const long mid = (lo + hi + 1) >> 1;
			if (word_off[mid] <= w)
				lo = mid;
			else
				hi = mid - 1;
		}
		const u32 owner = (u32)lo + 1u; // conreci
		const u64 wbase = word_off[lo];
		const int p = (int)(pos - wbase * 32ull);
		const Key<KW> c = reference_key(window_key<KW>(codes, wbase, p, g), g);
		u64 s = mulhi64(key_hash(c), t.cap);
		bool done = false;
		while (!done) {
			u64* slot = t.slots + s * kSlotWords;
			u32* state = reinterpret_cast<u32*>(slot + 3);
			u32* minown = state + 1;
			u32 st = __hip_atomic_load(state, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
			if (st == kEmpty) {
				u32 expect = kEmpty;
				if (__hip_atomic_compare_exchange_strong(
				        state, &expect, kLocked, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
				        __HIP_MEMORY_SCOPE_AGENT)) {
#pragma unroll
					for (int j = 0; j < KW; ++j)
						__hip_atomic_store(slot + j, c.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(minown, owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(state, owner + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
					n_new = 1;
					done = true;
				}
				// lost the race: look at the same slot again
			} else if (st != kLocked) {
				Key<KW> sk;
#pragma unroll
				for (int j = 0; j < KW; ++j)
					sk.w[j] = __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (key_eq(sk, c)) {
					__hip_atomic_fetch_min(minown, owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					while (st != owner + 1u && st != 1u) { // seen from a different end -> 0
						u32 expect = st;
						if (__hip_atomic_compare_exchange_strong(
						        state, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
						        __HIP_MEMORY_SCOPE_AGENT)) {
							n_lost = 1;
							break;
						}
						st = expect;
					}
					done = true;
				} else
					s = (s + 1 == t.cap) ? 0 : s + 1;
			}
			// st == kLocked: the writer is mid-flight (possibly a lane of this wave, which has
			// already finished its store sequence above) -- poll the same slot again
		}
	}
	// wave-level reduction of the two counters
	const u64 b_new = __ballot(n_new), b_lost = __ballot(n_lost);
	if ((threadIdx.x & 63) == 0) {
		if (b_new)
			atomicAdd(counters + 2, (u64)__popcll(b_new));
		if (b_lost)
			atomicAdd(counters + 3, (u64)__popcll(b_lost));
	}
}

// K2s: counts the visits whose end is the smallest end that visited the key; the reference's
// "removed" counter (Arcs.cpp:909, order dependent in the serial loop, ends in ascending order) is
// total visits minus that count.
template <int KW>
__global__ void
build_stats_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ visited,
    const u64* __restrict__ word_off,
    long n_ends,
    u64 total_words,
    KeyGeom g,
    TableView t,
    u64* __restrict__ counters) // [4] += visits by the key's smallest end
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 w = pos >> 5;
	bool active = w < total_words;
	if (active)
		active = (visited[w] >> (31 - (pos & 31))) & 1u;
	u32 hit = 0;
	if (active) {
		long lo = 0, hi = n_ends - 1;
		while (lo < hi) {
			const long mid = (lo + hi + 1) >> 1;
			if (word_off[mid] <= w)
				lo = mid;
			else
				hi = mid - 1;
		}
		const u32 owner = (u32)lo + 1u;
		const u64 wbase = word_off[lo];
		const int p = (int)(pos - wbase * 32ull);
		const Key<KW> c = reference_key(window_key<KW>(codes, wbase, p, g), g);
		u64 s = mulhi64(key_hash(c), t.cap);
		for (;;) {
			const u64* slot = t.slots + s * kSlotWords;
			const u64 meta = slot[3];
			if ((u32)meta == kEmpty)
				break; // cannot happen for a visited window
			if (key_eq(slot_key<KW>(slot), c)) {
				hit = (u32)(meta >> 32) == owner;
				break;
			}
			s = (s + 1 == t.cap) ? 0 : s + 1;
		}
	}
	const u64 b = __ballot(hit);
	if ((threadIdx.x & 63) == 0 && b)
		atomicAdd(counters + 4, (u64)__popcll(b));
}

// number of slots whose value is a real contig end (state >= 2): "unique kmers"
__global__ void
count_unique_kernel(TableView t, u64* __restrict__ out)
{
	u64 acc = 0;
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < t.cap; s += (u64)gridDim.x * blockDim.x)
		acc += (u32)t.slots[s * kSlotWords + 3] >= 2u;
	for (int off = 32; off > 0; off >>= 1)
		acc += __shfl_down(acc, off);
	if ((threadIdx.x & 63) == 0 && acc)
		atomicAdd(out, acc);
}

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
// with QUIRK = false a palindromic window is not resolved but reported as -4
template <int KW, bool QUIRK>
__device__ __forceinline__ int
window_value(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wbase, int p,
    const KeyGeom& g, const TableView& t)
{
	if (window_has_invalid<KW>(nmask, wbase, p, g.k))
		return -2;
	const Key<KW> f = window_key<KW>(codes, wbase, p, g);
	const Key<KW> r = key_revcomp(f, g);
	Key<KW> c;
	const bool lt = key_less(f, r);
#pragma unroll
	for (int j = 0; j < KW; ++j)
		c.w[j] = lt ? f.w[j] : r.w[j];
	if (key_eq(f, r)) {
		if (!QUIRK)
			return -4;
		c = key_palindrome_quirk(f, g);
	}
	return table_lookup<KW>(t, c);
}

// Slow path only.  Reads with more than 64 * kMaxPass windows (not produced by the linked-read
// pipelines, but bestContig accepts any length): instead of holding the window values, re-scan the
// read once per distinct value in ascending order.
template <int KW, bool STATS>
__device__ __forceinline__ void
vote_long_read(
    const u64* __restrict__ codes, const u32* __restrict__ nmask, u64 wbase, int nwin,
    const KeyGeom& g, const TableView& t, int lane, WaveStats& ws, int& best, int& best_cnt)
{
	best = 0;
	best_cnt = 0;
	int prev = 0;
	bool first = true;
	for (;;) {
		int m = 0x7FFFFFFF, cnt = 0;
		for (int base = 0; base < nwin; base += 64) {
			const int p = base + lane;
			const int v = p < nwin ? window_value<KW, true>(codes, nmask, wbase, p, g, t) : -3;
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
template <int KW, bool STATS, bool FAST>
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
				vote_long_read<KW, STATS>(codes, nmask, wbase, nwin, g, t, lane, rs, best, best_cnt);
		} else {
			int vals[kMaxPass];
#pragma unroll
			for (int ps = 0; ps < kMaxPass; ++ps) {
				int v = -3;
				if (ps * 64 < nwin) { // wave-uniform
					const int p = ps * 64 + lane;
					if (p < nwin)
						v = window_value<KW, !FAST>(codes, nmask, wbase, p, g, t);
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
launch_pack(
    const uint8_t* ascii, const u64* offsets, const u32* lens, const u64* word_off, long n_seqs,
    u64 total_words, u64* codes, u32* nmask, u32* n_count, u32* other, hipStream_t st)
{
	if (n_seqs <= 0 || total_words == 0)
		return hipSuccess;
	pack_kernel<<<blocks_for(total_words, 256), 256, 0, st>>>(
	    ascii, offsets, lens, word_off, n_seqs, total_words, codes, nmask, n_count, other);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_read_class(const u32* lens, const u32* n_count, const u32* other, long n, uint8_t* out, hipStream_t st)
{
	if (n <= 0)
		return hipSuccess;
	read_class_kernel<<<blocks_for((u64)n, 256), 256, 0, st>>>(lens, n_count, other, n, out);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_visit(
    const u32* nmask, const u64* word_off, const u32* lens, long n_ends, int k, u32* visited,
    u64* counters, hipStream_t st)
{
	if (n_ends <= 0)
		return hipSuccess;
	visit_kernel<<<blocks_for((u64)n_ends, 64), 64, 0, st>>>(nmask, word_off, lens, n_ends, k, visited, counters);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_popcount(const u32* words, u64 n, u64* out, hipStream_t st)
{
	if (n == 0)
		return hipSuccess;
	unsigned b = blocks_for(n, 256);
	b = b > 4096 ? 4096 : b;
	popcount_kernel<<<b, 256, 0, st>>>(words, n, out);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_insert(
    int kw, const u64* codes, const u32* visited, const u64* word_off, long n_ends, u64 total_words,
    const KeyGeom& g, TableView t, u64* counters, hipStream_t st)
{
	if (total_words == 0 || n_ends <= 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (kw == 2)
		insert_kernel<2><<<b, 256, 0, st>>>(codes, visited, word_off, n_ends, total_words, g, t, counters);
	else
		insert_kernel<3><<<b, 256, 0, st>>>(codes, visited, word_off, n_ends, total_words, g, t, counters);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_build_stats(
    int kw, const u64* codes, const u32* visited, const u64* word_off, long n_ends, u64 total_words,
    const KeyGeom& g, TableView t, u64* counters, hipStream_t st)
{
	if (total_words == 0 || n_ends <= 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (kw == 2)
		build_stats_kernel<2><<<b, 256, 0, st>>>(codes, visited, word_off, n_ends, total_words, g, t, counters);
	else
		build_stats_kernel<3><<<b, 256, 0, st>>>(codes, visited, word_off, n_ends, total_words, g, t, counters);
	ARKS_LAUNCH_CHECK();
	unsigned bu = blocks_for(t.cap, 256);
	bu = bu > 4096 ? 4096 : bu;
	count_unique_kernel<<<bu, 256, 0, st>>>(t, counters + 5);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_map_reads(
    int kw, const u64* codes, const u32* nmask, const u64* word_off, const u32* lens,
    const uint8_t* eval, long n_reads, double j_index, const KeyGeom& g, TableView t, int* out,
    u64* stats, u32* queue, u32* queue_count, int n_cu, hipStream_t st)
{
	if (n_reads <= 0)
		return hipSuccess;
	hipError_t e = hipMemsetAsync(queue_count, 0, sizeof(u32), st);
	if (e != hipSuccess)
		return e;
	// one wave per read at a time; enough resident waves to cover the table-probe latency
	const u64 want = ((u64)n_reads + 3) / 4;
	const u64 cap = (u64)(n_cu > 0 ? n_cu : 256) * 8ull;
	const unsigned b = (unsigned)(want < cap ? want : cap);
	const unsigned bs = (unsigned)(want < 256 ? want : 256); // slow path: the queue is short
#define ARKS_MAP(KWV, ST)                                                                          \
	do {                                                                                           \
		map_reads_kernel<KWV, ST, true><<<b, 256, 0, st>>>(                                        \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, out, stats, queue,         \
		    queue_count);                                                                          \
		map_reads_kernel<KWV, ST, false><<<bs, 256, 0, st>>>(                                      \
		    codes, nmask, word_off, lens, eval, n_reads, j_index, g, t, out, stats, queue,         \
		    queue_count);                                                                          \
	} while (0)
	if (kw == 2) {
		if (stats) ARKS_MAP(2, true); else ARKS_MAP(2, false);
	} else {
		if (stats) ARKS_MAP(3, true); else ARKS_MAP(3, false);
	}
#undef ARKS_MAP
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
