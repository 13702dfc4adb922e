// arks_build.hip -- index-build kernels of libarks_hip (gfx950 / CDNA4, wave64).
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
#include "arks_shard_stats.hpp"

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
	// bit 0: accepted; bit 1: ACGT only (the map kernels then skip the read's N masks)
	const uint8_t ok = (seq_other[r] == 0 && !(ar > 0.02)) ? 1 : 0;
	out[r] = (uint8_t)(ok | ((ok && seq_n_count[r] == 0) ? 2 : 0));
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
    const u32* __restrict__ word_owner, // contig end (conreci) of every text word
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
	u32 n_new = 0, n_lost = 0, owner = 0;
	Key<KW> c;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		c.w[j] = 0;
	u64 s = 0;
	if (active) {
		owner = word_owner[w]; // conreci (one load; a per-thread search over the ends costs 16 dependent ones)
		c = reference_key(window_key_at<KW>(codes, pos, g), g);
		s = mulhi64(key_hash(c), t.cap);
	}
	// The loop's trip count is wave-uniform (ballot): a lane that has finished idles inside the
	// loop until its whole wave has.  With a per-lane exit the compiler is free to sink the
	// winner's key/state stores to the loop exit, which a SIMT machine reaches only after every
	// lane has left the loop -- while the wave-mates polling the locked slot never would.
	//
	// Memory ordering: every access to a slot is an agent-scope ATOMIC (sc1: stores write through,
	// loads and read-modify-writes are served at the coherence point), so no cache holds a private
	// copy of slot data and the agent-scope acquire / release forms -- which on gfx950 invalidate /
	// write back the whole L2 of the XCD per operation -- are not needed; what is needed is program
	// order between the key stores and the publishing state store (and between the state load and
	// the key loads), which a workgroup-scope fence provides (s_waitcnt).
	bool done = !active;
	while (__ballot(!done) != 0) {
		if (!done) {
			u64* slot = t.slots + s * kSlotWords;
			u32* state = reinterpret_cast<u32*>(slot + 3);
			u32* minown = state + 1;
			u32 st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (st == kEmpty) {
				u32 expect = kEmpty;
				if (__hip_atomic_compare_exchange_strong(
				        state, &expect, kLocked, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
				        __HIP_MEMORY_SCOPE_AGENT)) {
#pragma unroll
					for (int j = 0; j < KW; ++j)
						__hip_atomic_store(slot + j, c.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__hip_atomic_store(minown, owner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the key is written before it is published
					__hip_atomic_store(state, owner + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					n_new = 1;
					done = true;
				}
				// lost the race: look at the same slot again
			} else if (st != kLocked) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
			// st == kLocked: the writer is mid-flight -- poll the same slot again
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

// K2p: a shard of the index (arks_index_build_shard) holds the keys of ITS contig ends; a key that also
// occurs in an end of another shard must read 0 here as it does in the whole map (Arcs.cpp:905-910).
// One thread per visited window of the FOREIGN ends: a key found with an owner loses it; nothing is
// inserted.  Runs after K2 has finished (no LOCKED states, keys and states are final), so plain
// relaxed accesses suffice and the store is idempotent.
template <int KW, bool MIN>
__global__ void
poison_kernel(
    const u64* __restrict__ codes,
    const u32* __restrict__ visited,
    u64 total_words,
    KeyGeom g,
    TableView t,
    const u32* __restrict__ word_end, // MIN: 1-based index into conreci of the end that owns a text word
    const u32* __restrict__ conreci,  // MIN: the foreign ends' numbers in the whole list
    u64* __restrict__ counter) // += keys that lost their owner to another shard
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 w = pos >> 5;
	bool active = w < total_words;
	if (active)
		active = (visited[w] >> (31 - (pos & 31))) & 1u;
	u32 n_lost = 0;
	if (active) {
		const Key<KW> c = reference_key(window_key_at<KW>(codes, pos, g), g);
		u64 s = mulhi64(key_hash(c), t.cap);
		for (;;) {
			u64* slot = t.slots + s * kSlotWords;
			u32* state = reinterpret_cast<u32*>(slot + 3);
			const u32 st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (st == kEmpty)
				break; // not a key of this shard
			Key<KW> sk;
#pragma unroll
			for (int j = 0; j < KW; ++j)
				sk.w[j] = __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (key_eq(sk, c)) {
				if (st != 1u)
					n_lost = __hip_atomic_exchange(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u;
				if (MIN) // the key's first holder is the shard of the smallest end over ALL shards (build_stats_kernel)
					(void)__hip_atomic_fetch_min(state + 1, conreci[word_end[w] - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
			s = (s + 1 == t.cap) ? 0 : s + 1;
		}
	}
	const u64 b_lost = __ballot(n_lost);
	if ((threadIdx.x & 63) == 0 && b_lost)
		atomicAdd(counter, (u64)__popcll(b_lost));
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
// Locality index ("B", arks_device.hpp) -- build kernels.  All run once per index, one thread per
// text position unless noted; bitmaps share the indexing of `visited`.
// ------------------------------------------------------------------------------------------------

// contig-end index of every 32-base word (0 in the front / back padding)
__global__ void
word_owner_kernel(const u64* __restrict__ word_off, long n_ends, u64 total_words, u32* __restrict__ owner)
{
	const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= total_words)
		return;
	u32 o = 0;
	if (n_ends > 0 && w >= word_off[0] && w < word_off[n_ends]) {
		long lo = 0, hi = n_ends - 1;
		while (lo < hi) {
			const long mid = (lo + hi + 1) >> 1;
			if (word_off[mid] <= w)
				lo = mid;
			else
				hi = mid - 1;
		}
		o = (u32)lo + 1u;
	}
	owner[w] = o;
}

// Is the regular (non-palindromic) reference key c the quirk image of the palindrome that its own
// first half spells?  (Only such keys can be hit by a palindromic query, see DESIGN.md.)
template <int KW>
__device__ __forceinline__ bool
is_quirk_image(const Key<KW>& c, const KeyGeom& g)
{
	if (g.k & 1)
		return false;
	// byte `half` of a quirk key is 0 (ReadsProcessor.cpp:505: the byte is skipped)
	u64 wsel = 0;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		wsel = ((g.half >> 3) == j) ? c.w[j] : wsel;
	if (((wsel >> (56 - 8 * (g.half & 7))) & 0xFFull) != 0)
		return false;
	// Q = first k/2 bases of c followed by their reverse complement
	const int hb = g.k; // bits of the first half: 2 * (k/2)
	Key<KW> h, lowmask;
#pragma unroll
	for (int j = 0; j < KW; ++j) {
		const int bits = hb - 64 * j;
		const u64 m = bits <= 0 ? 0ull : (bits >= 64 ? ~0ull : ~(~0ull >> bits));
		h.w[j] = c.w[j] & m;
		lowmask.w[j] = g.mask[j] & ~m;
	}
	const Key<KW> rh = key_revcomp(h, g); // T..T followed by revcomp(first half)
	Key<KW> q;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		q.w[j] = h.w[j] | (rh.w[j] & lowmask.w[j]);
	return key_eq(key_palindrome_quirk(q, g), c);
}

// per visited window: ambiguity bit (value 0 in the full table), palindrome / quirk-image bits,
// and the positions of its minimizer (all ties)
template <int KW, int MM>
__global__ void
bmark_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ visited, u64 total_words, KeyGeom g,
    TableView full, int w, int dense, u32* __restrict__ ambig, u32* __restrict__ is_min,
    u32* __restrict__ is_pal, u32* __restrict__ is_img)
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	if (word >= total_words)
		return;
	const u32 bit = 1u << (31 - (u32)(pos & 31));
	if (!(visited[word] & bit))
		return;
	const Key<KW> f = window_key_at<KW>(codes, pos, g);
	const Key<KW> r = key_revcomp(f, g);
	const bool pal = key_eq(f, r);
	Key<KW> c;
	const bool lt = key_less(f, r);
#pragma unroll
	for (int j = 0; j < KW; ++j)
		c.w[j] = lt ? f.w[j] : r.w[j];
	if (pal)
		c = key_palindrome_quirk(f, g);
	if (table_lookup<KW>(full, c) == 0)
		atomicOr(ambig + word, bit);
	if (pal)
		atomicOr(is_pal + word, bit);
	else if (is_quirk_image(c, g))
		atomicOr(is_img + word, bit);
	if (dense)
		return; // seed index: every m-mer position of a visited window is registered (bdilate_kernel)
	typedef typename Mmer<MM>::type mm_t;
	u32 min_h;
	int off;
	window_minimizer<MM>(codes, pos, w, min_h, off);
	for (int o = off; o < w; ++o) {
		const mm_t mf = mmer_fw<MM>(codes, pos + (u64)o);
		const mm_t mr = mmer_rc<MM>(mf);
		if (mmer_order<MM>(mf < mr ? mf : mr) == min_h) {
			const u64 q = pos + (u64)o;
			atomicOr(is_min + (q >> 5), 1u << (31 - (u32)(q & 31)));
		}
	}
}

// out bit q = OR of the in bits q - w + 1 .. q (bit 31 of a word = its first position): the m-mer start
// positions inside a visited window (seed index), one thread per 32-position word.  The text is
// front-padded (kFrontPadWords), so the words before the first one read as zeros.
__global__ void
bdilate_kernel(const u32* __restrict__ in, u64 total_words, int w, u32* __restrict__ out)
{
	const u64 word = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (word >= total_words)
		return;
	// the 32 + w - 1 <= 127 positions that end with this word, as a 128-bit stream (position word*32 - 96 first)
	u32 x[4];
#pragma unroll
	for (int j = 0; j < 4; ++j)
		x[j] = word + (u64)j >= 3 ? in[word + (u64)j - 3] : 0u;
	u32 acc = 0;
	for (int s = 0; s < w; ++s) {
		// 32 bits from position word*32 - s on
		const int first = 96 - s, wi = first >> 5, sh = first & 31;
		acc |= sh ? ((x[wi] << sh) | (x[wi + 1] >> (32 - sh))) : x[wi];
	}
	out[word] = acc;
}

// text records of the seed index: what the hot kernel stages per text word, side by side: 16 bytes = the word's codes,
// its visited and its ambiguous bits.  The owner (contig end) of a word is not in the record: it changes only at the
// borders of the ends, so a table of one u32 per 32 words (1024 bases; 5.5 MB at 3 Gbp: L2 / Infinity Cache
// resident) answers for every block that lies inside one end, ~0 sends the few blocks with a border (or padding) in
// them to word_owner.  (24-byte records with the owner inside cost the hot kernel a third load per staging slot and
// every second read a 64-byte line more: 4.13 -> 3.89 ms per 20 M pairs by the loads alone, profiles/r03g_ab.txt.)
__global__ void
btextrec_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ visited, const u32* __restrict__ ambig,
    const u32* __restrict__ word_owner, u64 alloc_words, u64* __restrict__ trec, u32* __restrict__ owner_blk)
{
	const u64 word = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (word >= alloc_words)
		return;
	trec[2 * word + 0] = codes[word];
	trec[2 * word + 1] = (u64)visited[word] | ((u64)ambig[word] << 32);
	if ((word & 31) == 0) { // this thread's block of 32 words
		const u32 o = word_owner[word];
		bool same = true;
		for (u64 x = 1; x < 32; ++x)
			same = same && (word + x < alloc_words ? word_owner[word + x] : 0u) == o;
		owner_blk[word >> 5] = same ? o : 0xFFFFFFFFu;
	}
}

// registered m-mer positions per owner rank of a sharded seed table
template <int MM>
__global__ void
bowners_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ is_min, u64 total_words, u32 n_own, u64* __restrict__ per_owner)
{
	typedef typename Mmer<MM>::type mm_t;
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	const bool reg = word < total_words && ((is_min[word] >> (31 - (u32)(pos & 31))) & 1u);
	u32 own = 0xFFFFFFFFu;
	if (reg) {
		const mm_t mf = mmer_fw<MM>(codes, pos);
		const mm_t mr = mmer_rc<MM>(mf);
		own = seed_owner<MM>(mf < mr ? mf : mr, n_own);
	}
	for (u32 o = 0; o < n_own; ++o) { // few owners: one ballot each
		const u64 b = __ballot(own == o);
		if ((threadIdx.x & 63) == 0 && b)
			atomicAdd(per_owner + o, (u64)__popcll(b));
	}
}

// bit of window pos = some position of [pos, pos + w) is set in `bits`
__device__ __forceinline__ bool
any_bit_in_span(const u32* __restrict__ bits, u64 pos, int w)
{
	u64 q = pos;
	int left = w;
	while (left > 0) {
		const u32 sh = (u32)(q & 31);
		const int take = left < 32 - (int)sh ? left : 32 - (int)sh;
		const u32 m = (0xFFFFFFFFu >> sh) & (take + (int)sh == 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (sh + (u32)take)));
		if (bits[q >> 5] & m)
			return true;
		q += (u64)take;
		left -= take;
	}
	return false;
}

// ---- minimizer occurrence counts: open-addressed u64 key (canonical m-mer | 1 << 63) -> u32 counter
constexpr u32 kCntForced = 0x80000000u; // heavy by decree (quirk-image minimizers)
constexpr u32 kCntMarker = 0x40000000u; // the HEAVY marker entry has been written to mtab
constexpr u32 kCntMask = 0x3FFFFFFFu;

template <int MM>
__device__ inline u32*
ctab_slot(u64* __restrict__ keys, u32* __restrict__ cnts, u64 cap, typename Mmer<MM>::type cm, bool insert)
{
	const u64 key = (u64)cm | (1ull << 63);
	u64 s = mtab_home<MM>(cm, cap);
	for (;;) {
		u64 cur = __hip_atomic_load(keys + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (cur == 0) {
			if (!insert)
				return nullptr;
			u64 expect = 0;
			if (__hip_atomic_compare_exchange_strong(
			        keys + s, &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
				cur = key;
			else
				cur = expect;
		}
		if (cur == key)
			return cnts + s;
		s = (s + 1 == cap) ? 0 : s + 1;
	}
}

template <int MM>
__global__ void
bcount_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ is_min, u64 total_words,
    u64* __restrict__ ckeys, u32* __restrict__ ccnts, u64 ccap, u32 own, u32 n_own)
{
	typedef typename Mmer<MM>::type mm_t;
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	if (word >= total_words || !((is_min[word] >> (31 - (u32)(pos & 31))) & 1u))
		return;
	const mm_t mf = mmer_fw<MM>(codes, pos);
	const mm_t mr = mmer_rc<MM>(mf);
	const mm_t cm = mf < mr ? mf : mr;
	if (n_own > 1 && seed_owner<MM>(cm, n_own) != own) // a sharded seed table: one owner's m-mers per pass
		return;
	atomicAdd(ctab_slot<MM>(ckeys, ccnts, ccap, cm, true), 1u);
}

// m-mer at offset o of a key (rare paths only)
template <int KW, int MM>
__device__ inline typename Mmer<MM>::type
key_mmer(const Key<KW>& x, int o)
{
	typename Mmer<MM>::type m = 0;
	for (int i = 0; i < MM; ++i)
		m = (m << 2) | key_base(x, o + i);
	return m;
}

// For every indexed palindrome Q: the sequence X' that spells its quirk key K'(Q).  A regular query
// whose canonical k-mer is X' must find K' -- which is not in the text -- so the minimizer of X' is
// decreed heavy: such queries then consult the fallback table, where K' lives.
// PHASE 0: decree (count table); PHASE 1: write the HEAVY marker into mtab (once per minimizer).
template <int KW, int MM, int PHASE>
__global__ void
bforce_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ is_pal, u64 total_words, KeyGeom g, int w, int dense,
    u64* __restrict__ ckeys, u32* __restrict__ ccnts, u64 ccap, u64* __restrict__ mtab, u64 mcap, u32 own, u32 n_own)
{
	typedef typename Mmer<MM>::type mm_t;
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	if (word >= total_words || !((is_pal[word] >> (31 - (u32)(pos & 31))) & 1u))
		return;
	const Key<KW> x = key_palindrome_quirk(window_key_at<KW>(codes, pos, g), g);
	const Key<KW> rx = key_revcomp(x, g);
	if (!key_less(x, rx))
		return; // X' is not a canonical non-palindromic k-mer: no regular query has key K'
	u32 min_h = 0xFFFFFFFFu;
	for (int o = 0; o < w; ++o) {
		const mm_t mf = key_mmer<KW, MM>(x, o), mr = mmer_rc<MM>(mf);
		const u32 h = mmer_order<MM>(mf < mr ? mf : mr);
		min_h = h < min_h ? h : min_h;
	}
	for (int o = 0; o < w; ++o) {
		const mm_t mf = key_mmer<KW, MM>(x, o), mr = mmer_rc<MM>(mf);
		const mm_t cm = mf < mr ? mf : mr;
		if (!dense && mmer_order<MM>(cm) != min_h) // seed index: a query may come through any m-mer of X'
			continue;
		if (n_own > 1 && seed_owner<MM>(cm, n_own) != own)
			continue;
		u32* cnt = ctab_slot<MM>(ckeys, ccnts, ccap, cm, true);
		if (PHASE == 0)
			atomicOr(cnt, kCntForced);
		else if (!(atomicOr(cnt, kCntMarker) & kCntMarker)) {
			u64 s = mtab_home<MM>(cm, mcap);
			const u64 e = mtab_entry(mmer_fp<MM>(cm), 0, kHeavyPos);
			for (;;) {
				u64 expect = 0;
				if (__hip_atomic_compare_exchange_strong(
				        mtab + s, &expect, e, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
					break;
				s = (s + 1 == mcap) ? 0 : s + 1;
			}
		}
	}
}

// every minimizer position -> one mtab entry, or (heavy minimizer) the heavy bit + one marker
template <int MM>
__global__ void
bfill_mtab_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ is_min, u64 total_words,
    u64* __restrict__ ckeys, u32* __restrict__ ccnts, u64 ccap, u64* __restrict__ mtab, u64 mcap,
    u32* __restrict__ heavy_min, u32 own, u32 n_own, int fill, u32 heavy_over)
{
	typedef typename Mmer<MM>::type mm_t;
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	if (word >= total_words || !((is_min[word] >> (31 - (u32)(pos & 31))) & 1u))
		return;
	const mm_t mf = mmer_fw<MM>(codes, pos);
	const mm_t mr = mmer_rc<MM>(mf);
	const mm_t cm = mf < mr ? mf : mr;
	if (n_own > 1 && seed_owner<MM>(cm, n_own) != own)
		return;
	u32* cnt = ctab_slot<MM>(ckeys, ccnts, ccap, cm, false);
	const u32 c = *cnt;
	u64 e;
	if ((c & kCntForced) || (c & kCntMask) > heavy_over) {
		atomicOr(heavy_min + word, 1u << (31 - (u32)(pos & 31)));
		if (!fill || (atomicOr(cnt, kCntMarker) & kCntMarker))
			return;
		e = mtab_entry(mmer_fp<MM>(cm), 0, kHeavyPos);
	} else {
		if (!fill) // another rank's shard: only the heavy bits are wanted (the fallback table is the same everywhere)
			return;
		e = mtab_entry(mmer_fp<MM>(cm), mf < mr ? 1u : 0u, (u32)pos);
	}
	u64 s = mtab_home<MM>(cm, mcap);
	for (;;) {
		u64 expect = 0;
		if (__hip_atomic_compare_exchange_strong(
		        mtab + s, &expect, e, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			break;
		s = (s + 1 == mcap) ? 0 : s + 1;
	}
}

// (key, value) -> table, first writer wins (every occurrence of a key carries the same value).
template <int KW>
__device__ inline void
table_put(const TableView& t, const Key<KW>& c, u32 value, bool active)
{
	u64 s = active ? mulhi64(key_hash(c), t.cap) : 0;
	bool done = !active;
	// wave-uniform trip count (see insert_kernel): every lane stays in the loop until the whole
	// wave is done, so a lane that polls a slot locked by a wave-mate cannot starve it
	while (__ballot(!done) != 0) {
		if (!done) {
			u64* slot = t.slots + s * kSlotWords;
			u32* state = reinterpret_cast<u32*>(slot + 3);
			// relaxed agent-scope atomics + program order, as in insert_kernel
			const u32 st = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (st == kEmpty) {
				u32 expect = kEmpty;
				if (__hip_atomic_compare_exchange_strong(
				        state, &expect, kLocked, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
				        __HIP_MEMORY_SCOPE_AGENT)) {
#pragma unroll
					for (int j = 0; j < KW; ++j)
						__hip_atomic_store(slot + j, c.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					__hip_atomic_store(state, value + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					done = true;
				}
			} else if (st != kLocked) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
				Key<KW> sk;
#pragma unroll
				for (int j = 0; j < KW; ++j)
					sk.w[j] = __hip_atomic_load(slot + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (key_eq(sk, c))
					done = true;
				else
					s = (s + 1 == t.cap) ? 0 : s + 1;
			}
		}
	}
}

// Windows the text path cannot (or must not) answer go to the fallback table: palindromes (their
// key is not their sequence), quirk images (a palindromic query may ask for them), and every
// window one of whose minimizer positions is heavy.  INSERT = false only counts them.
template <int KW, int MM, bool INSERT>
__global__ void
bfallback_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ visited, const u32* __restrict__ ambig,
    const u32* __restrict__ is_pal, const u32* __restrict__ is_img, const u32* __restrict__ heavy_min,
    const u32* __restrict__ word_owner, u64 total_words, KeyGeom g, int w, int dense, TableView fb,
    u64* __restrict__ counter)
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	bool take = false;
	const u32 sh = 31 - (u32)(pos & 31);
	if (word < total_words && ((visited[word] >> sh) & 1u)) {
		take = ((is_pal[word] | is_img[word]) >> sh) & 1u;
		if (!take && dense) {
			take = any_bit_in_span(heavy_min, pos, w); // a query may come through any m-mer of the window
		} else if (!take) {
			typedef typename Mmer<MM>::type mm_t;
			u32 min_h;
			int off;
			window_minimizer<MM>(codes, pos, w, min_h, off);
			for (int o = off; o < w && !take; ++o) {
				const mm_t mf = mmer_fw<MM>(codes, pos + (u64)o);
				const mm_t mr = mmer_rc<MM>(mf);
				if (mmer_order<MM>(mf < mr ? mf : mr) == min_h)
					take = bit_at(heavy_min, pos + (u64)o);
			}
		}
	}
	if (INSERT) {
		Key<KW> c;
		u32 value = 0;
#pragma unroll
		for (int j = 0; j < KW; ++j)
			c.w[j] = 0;
		if (take) {
			c = reference_key(window_key_at<KW>(codes, pos, g), g);
			value = ((ambig[word] >> sh) & 1u) ? 0u : word_owner[word];
		}
		table_put<KW>(fb, c, value, take);
	} else {
		const u64 b = __ballot(take);
		if ((threadIdx.x & 63) == 0 && b)
			atomicAdd(counter, (u64)__popcll(b));
	}
}

// export: one (key, value) record per visited window (the host removes duplicates)
template <int KW>
__global__ void
bexport_kernel(
    const u64* __restrict__ codes, const u32* __restrict__ visited, const u32* __restrict__ ambig,
    const u32* __restrict__ word_owner, u64 total_words, KeyGeom g, u64* __restrict__ out_keys,
    int* __restrict__ out_vals, u64* __restrict__ counter)
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 word = pos >> 5;
	const u32 sh = 31 - (u32)(pos & 31);
	if (word >= total_words || !((visited[word] >> sh) & 1u))
		return;
	const Key<KW> c = reference_key(window_key_at<KW>(codes, pos, g), g);
	const u64 i = atomicAdd(counter, 1ull);
#pragma unroll
	for (int j = 0; j < KW; ++j)
		out_keys[i * KW + j] = c.w[j];
	out_vals[i] = ((ambig[word] >> sh) & 1u) ? 0 : (int)word_owner[word];
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
    int kw, const u64* codes, const u32* visited, const u32* word_owner, long n_ends, u64 total_words,
    const KeyGeom& g, TableView t, u64* counters, hipStream_t st)
{
	if (total_words == 0 || n_ends <= 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (kw == 2)
		insert_kernel<2><<<b, 256, 0, st>>>(codes, visited, word_owner, total_words, g, t, counters);
	else
		insert_kernel<3><<<b, 256, 0, st>>>(codes, visited, word_owner, total_words, g, t, counters);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_poison(
    int kw, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView t,
    u64* counter, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const u64 n = total_words * 32;
	const unsigned blocks = (unsigned)((n + 255) / 256);
	if (kw == 2)
		poison_kernel<2, false><<<blocks, 256, 0, st>>>(codes, visited, total_words, g, t, nullptr, nullptr, counter);
	else
		poison_kernel<3, false><<<blocks, 256, 0, st>>>(codes, visited, total_words, g, t, nullptr, nullptr, counter);
	return hipGetLastError();
}

hipError_t
launch_poison_min(
    int kw, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView t,
    const u32* word_end, const u32* conreci, u64* counter, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned blocks = blocks_for(total_words * 32ull, 256);
	if (kw == 2)
		poison_kernel<2, true><<<blocks, 256, 0, st>>>(codes, visited, total_words, g, t, word_end, conreci, counter);
	else
		poison_kernel<3, true><<<blocks, 256, 0, st>>>(codes, visited, total_words, g, t, word_end, conreci, counter);
	return hipGetLastError();
}

// keys whose smallest end (over the shards: poison_kernel<MIN>) is an end of this shard
__global__ void
count_first_holder_kernel(TableView t, const u32* __restrict__ lens, u64* __restrict__ out)
{
	u64 acc = 0;
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < t.cap; s += (u64)gridDim.x * blockDim.x) {
		const u64 meta = t.slots[s * kSlotWords + 3];
		if ((u32)meta != kEmpty)
			acc += lens[(u32)(meta >> 32) - 1u] != 0u;
	}
	for (int off = 32; off > 0; off >>= 1)
		acc += __shfl_down(acc, off);
	if ((threadIdx.x & 63) == 0 && acc)
		atomicAdd(out, acc);
}

// K2q: a key that ends of several shards visit stays in ONE shard's index -- that of the smallest end of the list
// that visited it (the slot's smallest-end field after poison_kernel<MIN>).  The other shards take their visits of
// it back: the key reads 0 wherever it is (shared between ends), so no vote changes, and with it in one place only
// the read stage's found / duplicate counters of the shards add up to the one map's (Arcs.cpp:975-982).
template <int KW>
__global__ void
drop_later_holders_kernel(
    const u64* __restrict__ codes, u32* __restrict__ visited, const u32* __restrict__ lens, u64 total_words,
    KeyGeom g, TableView t, u64* __restrict__ counter) // += visits taken back
{
	const u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 w = pos >> 5;
	bool active = w < total_words;
	if (active)
		active = (visited[w] >> (31 - (pos & 31))) & 1u;
	u32 drop = 0;
	if (active) {
		const Key<KW> c = reference_key(window_key_at<KW>(codes, pos, g), g);
		u64 s = mulhi64(key_hash(c), t.cap);
		for (;;) {
			const u64* slot = t.slots + s * kSlotWords;
			const u64 meta = slot[3];
			if ((u32)meta == kEmpty)
				break; // cannot happen for a visited window
			if (key_eq(slot_key<KW>(slot), c)) {
				drop = lens[(u32)(meta >> 32) - 1u] == 0u;
				break;
			}
			s = (s + 1 == t.cap) ? 0 : s + 1;
		}
		if (drop) // (a bit is read and cleared by its own thread only)
			atomicAnd(visited + w, ~(1u << (31 - (pos & 31))));
	}
	const u64 b = __ballot(drop);
	if ((threadIdx.x & 63) == 0 && b)
		atomicAdd(counter, (u64)__popcll(b));
}

// ... and where the table itself is the index (no text: k < 20, ARKS_INDEX_KIND=hash), the first holders' slots are
// re-inserted into a table of their own (distinct keys: a free slot is claimed, nothing is compared)
template <int KW>
__global__ void
keep_first_holders_kernel(TableView from, const u32* __restrict__ lens, TableView to)
{
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < from.cap; s += (u64)gridDim.x * blockDim.x) {
		const u64* slot = from.slots + s * kSlotWords;
		const u64 meta = slot[3];
		if ((u32)meta == kEmpty || lens[(u32)(meta >> 32) - 1u] == 0u)
			continue;
		const Key<KW> c = slot_key<KW>(slot);
		u64 d = mulhi64(key_hash(c), to.cap);
		for (;;) {
			u64* dst = to.slots + d * kSlotWords;
			u32* state = reinterpret_cast<u32*>(dst + 3);
			if (atomicCAS(state, kEmpty, (u32)meta) == kEmpty) {
#pragma unroll
				for (int j = 0; j < KW; ++j)
					dst[j] = c.w[j];
				state[1] = (u32)(meta >> 32);
				break;
			}
			d = (d + 1 == to.cap) ? 0 : d + 1;
		}
	}
}

hipError_t
launch_drop_later_holders(
    int kw, const u64* codes, u32* visited, const u32* lens, u64 total_words, const KeyGeom& g, TableView t, u64* counter,
    hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (kw == 2)
		drop_later_holders_kernel<2><<<b, 256, 0, st>>>(codes, visited, lens, total_words, g, t, counter);
	else
		drop_later_holders_kernel<3><<<b, 256, 0, st>>>(codes, visited, lens, total_words, g, t, counter);
	return hipGetLastError();
}

hipError_t
launch_keep_first_holders(int kw, TableView from, const u32* lens, TableView to, hipStream_t st)
{
	unsigned b = blocks_for(from.cap, 256);
	b = b > 4096 ? 4096 : b;
	if (kw == 2)
		keep_first_holders_kernel<2><<<b, 256, 0, st>>>(from, lens, to);
	else
		keep_first_holders_kernel<3><<<b, 256, 0, st>>>(from, lens, to);
	return hipGetLastError();
}

hipError_t
launch_count_first_holder(TableView t, const u32* lens, u64* out, hipStream_t st)
{
	unsigned b = blocks_for(t.cap, 256);
	b = b > 4096 ? 4096 : b;
	count_first_holder_kernel<<<b, 256, 0, st>>>(t, lens, out);
	return hipGetLastError();
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
launch_word_owner(const u64* word_off, long n_ends, u64 total_words, u32* owner, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	word_owner_kernel<<<blocks_for(total_words, 256), 256, 0, st>>>(word_off, n_ends, total_words, owner);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

#define ARKS_KW_DISPATCH(kw, CALL2, CALL3)                                                         \
	do {                                                                                           \
		if ((kw) == 2) { CALL2; } else { CALL3; }                                                  \
	} while (0)

#define ARKS_KM_DISPATCH(kw, mm, CALL)                                                             \
	do {                                                                                           \
		if ((kw) == 2 && (mm) == kMShort) { CALL(2, kMShort); }                                    \
		else if ((kw) == 2) { CALL(2, kMLong); }                                                   \
		else if ((mm) == kMShort) { CALL(3, kMShort); }                                            \
		else { CALL(3, kMLong); }                                                                  \
	} while (0)

hipError_t
launch_bmark(
    int kw, int mm, const u64* codes, const u32* visited, u64 total_words, const KeyGeom& g, TableView full,
    int w, bool dense, u32* ambig, u32* is_min, u32* is_pal, u32* is_img, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
#define ARKS_CALL(KWV, MMV)                                                                        \
	bmark_kernel<KWV, MMV><<<b, 256, 0, st>>>(codes, visited, total_words, g, full, w, dense ? 1 : 0, ambig, is_min, is_pal, is_img)
	ARKS_KM_DISPATCH(kw, mm, ARKS_CALL);
#undef ARKS_CALL
	ARKS_LAUNCH_CHECK();
	if (dense) {
		bdilate_kernel<<<blocks_for(total_words, 256), 256, 0, st>>>(visited, total_words, w, is_min);
		ARKS_LAUNCH_CHECK();
	}
	return hipSuccess;
}

hipError_t
launch_bcount(
    int mm, const u64* codes, const u32* is_min, u64 total_words, u64* ckeys, u32* ccnts, u64 ccap, u32 own, u32 n_own,
    hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (mm == kMShort)
		bcount_kernel<kMShort><<<b, 256, 0, st>>>(codes, is_min, total_words, ckeys, ccnts, ccap, own, n_own);
	else
		bcount_kernel<kMLong><<<b, 256, 0, st>>>(codes, is_min, total_words, ckeys, ccnts, ccap, own, n_own);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bforce(
    int kw, int mm, int phase, const u64* codes, const u32* is_pal, u64 total_words, const KeyGeom& g, int w,
    bool dense, u64* ckeys, u32* ccnts, u64 ccap, u64* mtab, u64 mcap, u32 own, u32 n_own, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
#define ARKS_CALL0(KWV, MMV)                                                                       \
	bforce_kernel<KWV, MMV, 0><<<b, 256, 0, st>>>(codes, is_pal, total_words, g, w, dense ? 1 : 0, ckeys, ccnts, ccap, mtab, mcap, own, n_own)
#define ARKS_CALL1(KWV, MMV)                                                                       \
	bforce_kernel<KWV, MMV, 1><<<b, 256, 0, st>>>(codes, is_pal, total_words, g, w, dense ? 1 : 0, ckeys, ccnts, ccap, mtab, mcap, own, n_own)
	if (phase == 0)
		ARKS_KM_DISPATCH(kw, mm, ARKS_CALL0);
	else
		ARKS_KM_DISPATCH(kw, mm, ARKS_CALL1);
#undef ARKS_CALL0
#undef ARKS_CALL1
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bfill_mtab(
    int mm, const u64* codes, const u32* is_min, u64 total_words, u64* ckeys, u32* ccnts, u64 ccap, u64* mtab,
    u64 mcap, u32* heavy_min, u32 own, u32 n_own, bool fill, u32 heavy_over, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (mm == kMShort)
		bfill_mtab_kernel<kMShort><<<b, 256, 0, st>>>(
		    codes, is_min, total_words, ckeys, ccnts, ccap, mtab, mcap, heavy_min, own, n_own, fill ? 1 : 0, heavy_over);
	else
		bfill_mtab_kernel<kMLong><<<b, 256, 0, st>>>(
		    codes, is_min, total_words, ckeys, ccnts, ccap, mtab, mcap, heavy_min, own, n_own, fill ? 1 : 0, heavy_over);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bfallback(
    int kw, int mm, bool insert, const u64* codes, const u32* visited, const u32* ambig, const u32* is_pal,
    const u32* is_img, const u32* heavy_min, const u32* word_owner, u64 total_words, const KeyGeom& g,
    int w, bool dense, TableView fb, u64* counter, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
#define ARKS_FB_T(KWV, MMV)                                                                        \
	bfallback_kernel<KWV, MMV, true><<<b, 256, 0, st>>>(                                           \
	    codes, visited, ambig, is_pal, is_img, heavy_min, word_owner, total_words, g, w, dense ? 1 : 0, fb, counter)
#define ARKS_FB_F(KWV, MMV)                                                                        \
	bfallback_kernel<KWV, MMV, false><<<b, 256, 0, st>>>(                                          \
	    codes, visited, ambig, is_pal, is_img, heavy_min, word_owner, total_words, g, w, dense ? 1 : 0, fb, counter)
	if (insert)
		ARKS_KM_DISPATCH(kw, mm, ARKS_FB_T);
	else
		ARKS_KM_DISPATCH(kw, mm, ARKS_FB_F);
#undef ARKS_FB_T
#undef ARKS_FB_F
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bdilate(const u32* visited, u64 total_words, int w, u32* is_min, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	bdilate_kernel<<<blocks_for(total_words, 256), 256, 0, st>>>(visited, total_words, w, is_min);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bowners(int mm, const u64* codes, const u32* is_min, u64 total_words, u32 n_own, u64* per_owner, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	if (mm == kMShort)
		bowners_kernel<kMShort><<<b, 256, 0, st>>>(codes, is_min, total_words, n_own, per_owner);
	else
		bowners_kernel<kMLong><<<b, 256, 0, st>>>(codes, is_min, total_words, n_own, per_owner);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_btextrec(
    const u64* codes, const u32* visited, const u32* ambig, const u32* word_owner, u64 alloc_words, u64* trec,
    u32* owner_blk, hipStream_t st)
{
	if (alloc_words == 0)
		return hipSuccess;
	btextrec_kernel<<<blocks_for(alloc_words, 256), 256, 0, st>>>(codes, visited, ambig, word_owner, alloc_words, trec, owner_blk);
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

hipError_t
launch_bexport(
    int kw, const u64* codes, const u32* visited, const u32* ambig, const u32* word_owner,
    u64 total_words, const KeyGeom& g, u64* out_keys, int* out_vals, u64* counter, hipStream_t st)
{
	if (total_words == 0)
		return hipSuccess;
	const unsigned b = blocks_for(total_words * 32ull, 256);
	ARKS_KW_DISPATCH(kw,
	    (bexport_kernel<2><<<b, 256, 0, st>>>(codes, visited, ambig, word_owner, total_words, g, out_keys, out_vals, counter)),
	    (bexport_kernel<3><<<b, 256, 0, st>>>(codes, visited, ambig, word_owner, total_words, g, out_keys, out_vals, counter)));
	ARKS_LAUNCH_CHECK();
	return hipSuccess;
}

} // namespace arks
