// arks_device.hpp -- device-side k-mer primitives shared by every kernel of libarks_hip (gfx950).
//
// A k-mer key is the reference's packed key (Common/ReadsProcessor.cpp:376-535: 4 bases per byte,
// first base in bits 7:6, zero padded) viewed as KW big-endian 64-bit words: w[0] holds bases
// 0..31 with base 0 in bits 63:62.  Lexicographic byte order == unsigned word order, so
// "the smaller of forward / reverse complement" (ReadsProcessor.cpp:427,464) is a word compare.
// Reads and contig ends are stored in exactly this bit order (include/arks_hip.h, "Packed read
// layout"), which makes a window's forward key a funnel-shifted bit-field of the stream -- no
// per-base work, no rolling state, any lane can produce any window.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arks {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kMaxKW = 3; // 64-bit words per key: k <= 32 * KW

template <int KW>
struct Key
{
	u64 w[KW];
};

// Byte geometry of a k-mer, Common/ReadsProcessor.cpp:20-37, plus the word masks the kernels use.
struct KeyGeom
{
	int k;
	int full;     // k / 4
	int hang;     // k % 4
	int half;     // ceil(k / 8)
	int nbytes;   // full + (hang != 0)
	int rc_shift; // 64*KW - 2k: left shift that re-aligns a reversed key
	u64 mask[kMaxKW]; // valid bits of word j of a left-aligned 2k-bit key
};

inline KeyGeom
make_geom(int k, int kw)
{
	KeyGeom g;
	g.k = k;
	g.full = k / 4;
	g.hang = k % 4;
	g.half = k / 8 + ((k % 8) ? 1 : 0);
	g.nbytes = g.full + (g.hang ? 1 : 0);
	g.rc_shift = 64 * kw - 2 * k;
	for (int j = 0; j < kMaxKW; ++j) {
		int bits = 2 * k - 64 * j;
		if (bits <= 0)
			g.mask[j] = 0;
		else if (bits >= 64)
			g.mask[j] = ~0ull;
		else
			g.mask[j] = ~(~0ull >> bits);
	}
	return g;
}

inline int
key_words_for_k(int k)
{
	return k <= 64 ? 2 : 3; // KW = 2 serves every k <= 64 (w[1] == 0 when k <= 32)
}

// (a:b) << s, upper 64 bits; s in [0, 63]
__device__ __forceinline__ u64
funnel_l(u64 a, u64 b, int s)
{
	return (a << s) | ((b >> 1) >> (63 - s));
}

// (b:a) >> s, lower 64 bits (a = low word); s in [0, 63]
__device__ __forceinline__ u64
funnel_r(u64 a, u64 b, int s)
{
	return (a >> s) | ((b << 1) << (63 - s));
}

// reverse the order of the 32 two-bit groups of x
__device__ __forceinline__ u64
rev_groups(u64 x)
{
	u64 t = __brevll(x);
	return ((t >> 1) & 0x5555555555555555ull) | ((t & 0x5555555555555555ull) << 1);
}

template <int KW>
__device__ __forceinline__ bool
key_less(const Key<KW>& a, const Key<KW>& b)
{
	bool lt = false, decided = false;
#pragma unroll
	for (int j = 0; j < KW; ++j) {
		bool ne = a.w[j] != b.w[j];
		lt = (!decided && ne) ? (a.w[j] < b.w[j]) : lt;
		decided = decided || ne;
	}
	return lt;
}

template <int KW>
__device__ __forceinline__ bool
key_eq(const Key<KW>& a, const Key<KW>& b)
{
	bool eq = true;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		eq = eq && (a.w[j] == b.w[j]);
	return eq;
}

// Forward key of the window that starts `p` bases after word `wbase` of a packed stream.
template <int KW>
__device__ __forceinline__ Key<KW>
window_key(const u64* __restrict__ codes, u64 wbase, int p, const KeyGeom& g)
{
	const u64* src = codes + wbase + (u64)(p >> 5);
	const int s = (p & 31) * 2;
	u64 w[KW + 1];
#pragma unroll
	for (int j = 0; j <= KW; ++j)
		w[j] = src[j];
	Key<KW> f;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		f.w[j] = funnel_l(w[j], w[j + 1], s) & g.mask[j];
	return f;
}

// same, for an absolute 64-bit base position of the stream (text side)
template <int KW>
__device__ __forceinline__ Key<KW>
window_key_at(const u64* __restrict__ codes, u64 pos, const KeyGeom& g)
{
	const u64* src = codes + (pos >> 5);
	const int s = (int)(pos & 31) * 2;
	u64 w[KW + 1];
#pragma unroll
	for (int j = 0; j <= KW; ++j)
		w[j] = src[j];
	Key<KW> f;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		f.w[j] = funnel_l(w[j], w[j + 1], s) & g.mask[j];
	return f;
}

// True iff any of the k bases of the window is flagged in the N-mask (=> NULL k-mer,
// ReadsProcessor.cpp:397-421; the reference validates every base of the window on every branch).
template <int KW>
__device__ __forceinline__ bool
window_has_invalid(const u32* __restrict__ nmask, u64 wbase, int p, int k)
{
	const u32* src = nmask + wbase + (u64)(p >> 5);
	const int t = p & 31, e = t + k; // bit range [t, e) relative to src[0]
	u32 any = 0;
#pragma unroll
	for (int j = 0; j <= KW; ++j) {
		int lo = t - 32 * j, hi = e - 32 * j;
		lo = lo < 0 ? 0 : lo;
		hi = hi > 32 ? 32 : hi;
		if (lo < hi) {
			u32 m = (0xFFFFFFFFu >> lo) & ~(hi == 32 ? 0u : (0xFFFFFFFFu >> hi));
			any |= src[j] & m;
		}
	}
	return any != 0;
}

// Reverse complement of a left-aligned 2k-bit key.
template <int KW>
__device__ __forceinline__ Key<KW>
key_revcomp(const Key<KW>& f, const KeyGeom& g)
{
	u64 r[KW + 1];
#pragma unroll
	for (int j = 0; j < KW; ++j)
		r[j] = rev_groups(f.w[KW - 1 - j]); // reversed sequence, right-aligned in 64*KW bits
	r[KW] = 0;
	const int ws = g.rc_shift >> 6, bs = g.rc_shift & 63;
	Key<KW> out;
#pragma unroll
	for (int j = 0; j < KW; ++j) {
		u64 a = 0, b = 0;
#pragma unroll
		for (int t = 0; t <= KW; ++t) { // static indices only: r[] stays in registers
			a = (t == j + ws) ? r[t] : a;
			b = (t == j + ws + 1) ? r[t] : b;
		}
		out.w[j] = ~funnel_l(a, b, bs) & g.mask[j];
	}
	return out;
}

template <int KW>
__device__ __forceinline__ u32
key_base(const Key<KW>& f, int i)
{
	u64 w = 0;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		w = ((i >> 5) == j) ? f.w[j] : w;
	return (u32)(w >> (62 - 2 * (i & 31))) & 3u;
}

template <int KW>
__device__ __forceinline__ void
key_or_byte(Key<KW>& o, int b, u32 v)
{
	const u64 x = (u64)(v & 0xFFu) << (56 - 8 * (b & 7));
#pragma unroll
	for (int j = 0; j < KW; ++j)
		o.w[j] |= ((b >> 3) == j) ? x : 0ull;
}

// The key the reference produces for a reverse-complement palindrome (forward == revcomp), its
// damaged branch Common/ReadsProcessor.cpp:503-534: the first `half` bytes are the forward bytes,
// byte `half` stays 0, later full bytes are refilled from a cursor that advances 3 bases per byte,
// and a hanging last byte keeps only one base.  Rare path (4^-(k/2) of random windows).
template <int KW>
__device__ __forceinline__ Key<KW>
key_palindrome_quirk(const Key<KW>& f, const KeyGeom& g)
{
	Key<KW> o;
#pragma unroll
	for (int j = 0; j < KW; ++j)
		o.w[j] = 0;
	for (int b = 0; b < g.half; ++b) {
		u32 v = (key_base(f, 4 * b) << 6) | (key_base(f, 4 * b + 1) << 4) |
		        (key_base(f, 4 * b + 2) << 2) | key_base(f, 4 * b + 3);
		key_or_byte(o, b, v);
	}
	int idx = 4 * g.half;
	for (int b = g.half + 1; b < g.full; ++b) {
		u32 v = (key_base(f, idx) << 6) | (key_base(f, idx + 1) << 4) | (key_base(f, idx + 2) << 2) |
		        key_base(f, idx + 3);
		key_or_byte(o, b, v);
		idx += 3;
	}
	if (g.hang) {
		int last = g.k - 1;
		u32 v = key_base(f, last) << 6;
		for (; idx < last; --last)
			v = ((v << 2) & 0xFFu) | (key_base(f, last) << 6);
		key_or_byte(o, g.full, v);
	}
	return o;
}

// The reference key (prepSeq + getStr) of a window whose forward key is f and that has no invalid
// base: min(forward, revcomp), or the palindrome quirk when they are equal.
template <int KW>
__device__ __forceinline__ Key<KW>
reference_key(const Key<KW>& f, const KeyGeom& g)
{
	const Key<KW> r = key_revcomp(f, g);
	Key<KW> c;
	const bool lt = key_less(f, r);
#pragma unroll
	for (int j = 0; j < KW; ++j)
		c.w[j] = lt ? f.w[j] : r.w[j];
	if (key_eq(f, r))
		c = key_palindrome_quirk(f, g);
	return c;
}

// 64-bit mix of a key; the table position is free to use any hash because the path's results
// depend only on exact key equality (Arcs/Arcs.h:153-156), not on CityHash64 (Arcs/Arcs.h:150).
template <int KW>
__device__ __forceinline__ u64
key_hash(const Key<KW>& c)
{
	u64 h = c.w[0] * 0x9E3779B97F4A7C15ull;
#pragma unroll
	for (int j = 1; j < KW; ++j) {
		h = (h << 31) | (h >> 33);
		h ^= c.w[j] * 0xC2B2AE3D27D4EB4Full;
	}
	h ^= h >> 32;
	h *= 0xD6E8FEB86659FD93ull;
	h ^= h >> 29;
	return h;
}

__device__ __forceinline__ u64
mulhi64(u64 a, u64 b)
{
	return __umul64hi(a, b);
}

// ---- the contig k-mer table ------------------------------------------------------------------
// Open addressing, linear probing.  One slot = 4 x u64 (32 B, two per 64-B line):
//   [0, KW)   key words
//   [3]       low 32 bits: state (0 = empty, 0xFFFFFFFF = being written, else value + 1)
//             high 32 bits: smallest contig-end index that visited the key (build statistics)
constexpr int kSlotWords = 4;
constexpr u32 kEmpty = 0u;
constexpr u32 kLocked = 0xFFFFFFFFu;

struct TableView
{
	u64* slots;
	u64 cap;
};

template <int KW>
__device__ __forceinline__ Key<KW>
slot_key(const u64* slot)
{
	Key<KW> s;
	if (KW == 2) {
		const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(slot);
		s.w[0] = v.x;
		s.w[KW - 1] = v.y;
	} else {
#pragma unroll
		for (int j = 0; j < KW; ++j)
			s.w[j] = slot[j];
	}
	return s;
}

// value of key c, or -1 when absent (read-only phase: no slot is locked any more)
template <int KW>
__device__ __forceinline__ int
table_lookup(const TableView& t, const Key<KW>& c)
{
	u64 s = mulhi64(key_hash(c), t.cap);
	for (;;) {
		const u64* slot = t.slots + s * kSlotWords;
		const u32 st = *reinterpret_cast<const u32*>(slot + 3);
		if (st == kEmpty)
			return -1;
		if (key_eq(slot_key<KW>(slot), c))
			return (int)(st - 1u);
		s = (s + 1 == t.cap) ? 0 : s + 1;
	}
}


// ================================================================================================
// The locality index ("B"): contig-end text + minimizer table.
//
// The hash table above costs one random 64-B line per window (4.4x the algorithmic bytes, see
// profiles/r01a_*).  Consecutive windows of a read overlap in k-1 bases, so the index below is
// organised around that overlap instead: the packed contig-end TEXT itself is the key store
// (2 bits per base, every k-mer of an end shares its bytes with its neighbours), a window is
// located through its MINIMIZER (the smallest-hashing canonical m-mer inside it, m = 15 or 21,
// shared by ~w/2 consecutive windows, w = k - m + 1), and membership is decided by comparing the read against the
// text along the implied diagonal -- exact, so the result is still "key equality" and nothing
// else (Arcs/Arcs.h:153-156).  Per position the text carries two bits: `visited` (the window
// starting here was inserted by mapKmers' visit rule, Arcs.cpp:887-926) and `ambig` (its key was
// seen from two different ends => value 0, Arcs.cpp:907-914); the value of an unambiguous window is
// the contig end that owns the position.  Keys that do not equal their own sequence (the
// reference's palindrome quirk, ReadsProcessor.cpp:503-534), k-mers under over-full minimizers and
// the few regular keys a palindromic query could collide with live in a small exact hash table
// (`fallback`), reached only when the query says so.
// ================================================================================================
constexpr int kMShort = 17; // minimizer length where k leaves no room for the long one (k < 24)
constexpr int kMLong = 21;  // the default: specific at any text size (a 15-mer has ~ text / 5.4e8 chance
                            // occurrences, and every chance occurrence is one more diagonal to rule out)
constexpr u32 kFpMask = (1u << 30) - 1u;
// An m-mer with more occurrences than this inside visited windows is "heavy": one marker in the table instead of its
// positions, and every window that holds it lives in the exact fallback table.  2 = what a probe reads (up to two
// entries): until round 5 it was 8, and a seed with 3-8 entries -- useless as a proposal of diagonals -- sent its
// windows through a walk over the entries and the text behind each (a chain of ~7 dependent reads per window: a tenth
// of the medium kernel on a human-like repeat spectrum) instead of one exact-key probe; the fallback table grows by 5 %
// there (327 -> 344 M keys), not at all on a draft without repeat families.  ARKS_HEAVY_OVER=8 in the environment
// builds an index the old way (tests: the walk stays reachable -- fingerprint collisions can still show a probe more
// than two entries).
constexpr int kHeavy = 2;
constexpr int kFrontPadWords = 16; // the text starts 512 bases into its arrays (diagonals may underrun)
constexpr u32 kHeavyPos = 0xFFFFFFFFu;

// Both lengths are odd: no minimizer is its own reverse complement, so its strand is defined.
template <int MM>
struct Mmer
{
	typedef u64 type;
};

// minimizer-table entry: [63] occupied, [62] strand (1 = the text m-mer is the canonical one),
// [61:32] 30-bit fingerprint (a hash) of the canonical m-mer,
// [31:0] text position (kHeavyPos = "heavy: ask the fallback table").  A fingerprint collision only
// proposes a diagonal that the exact verification then rejects.
__device__ __forceinline__ u64
mtab_entry(u32 fp, u32 strand, u32 pos)
{
	return (1ull << 63) | ((u64)(strand & 1u) << 62) | ((u64)(fp & kFpMask) << 32) | pos;
}

struct BIndexView
{
	const u64* codes;      // packed text, front-padded by kFrontPadWords
	const u32* visited;    // 1 bit per text position
	const u32* ambig;      // 1 bit per text position
	const u32* word_owner; // contig-end index (conreci) of every 32-base word, 0 in padding
	const u64* mtab;
	u64 mtab_cap;
	TableView fallback;
	int m;       // minimizer length (kMShort or kMLong)
	int w;       // minimizer window: k - m + 1
	int enabled; // 0 => the plain hash table `TableView` is the index (small k)
	int has_img; // the index holds regular keys that are quirk images (a palindromic query could hit them)
	const u64* trec; // seed index: codes, visited | ambig << 32 of every text word side by side (2 u64 = 16 B per word):
	                 // the hot kernel reads one record where the general kernels read three arrays
	const u32* owner_blk; // seed index: contig end of the 32 text words [32 b, 32 b + 32), ~0 where they differ
	                      // (a border of two ends, padding): word_owner then
	int dense;   // 1: the table holds EVERY m-mer position inside a visited window ("seed index"): a query
	             // window may be looked up through any m-mer it contains, so a read needs one fixed-position
	             // seed per w windows and no minimizers at all; 0: minimizer positions only
};

// the m-mer starting at base `pos` of a packed stream, right-aligned in 2*MM bits
template <int MM>
__device__ __forceinline__ typename Mmer<MM>::type
mmer_fw(const u64* __restrict__ codes, u64 pos)
{
	const u64* src = codes + (pos >> 5);
	return (typename Mmer<MM>::type)(funnel_l(src[0], src[1], (int)(pos & 31) * 2) >> (64 - 2 * MM));
}

__device__ __forceinline__ u32
mmer_rc_bits(u32 f, int mm)
{
	u32 t = __brev(f) >> (32 - 2 * mm);
	t = ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
	return ~t & ((1u << (2 * mm)) - 1u);
}

__device__ __forceinline__ u64
mmer_rc_bits(u64 f, int mm)
{
	u64 t = __brevll(f) >> (64 - 2 * mm);
	t = ((t >> 1) & 0x5555555555555555ull) | ((t & 0x5555555555555555ull) << 1);
	return ~t & ((1ull << (2 * mm)) - 1ull);
}

template <int MM>
__device__ __forceinline__ typename Mmer<MM>::type
mmer_rc(typename Mmer<MM>::type f)
{
	return mmer_rc_bits(f, MM);
}

__device__ __forceinline__ u32
mmer_fold(u32 cm)
{
	return cm;
}

__device__ __forceinline__ u32
mmer_fold(u64 cm)
{
	return (u32)cm ^ ((u32)(cm >> 32) * 0x85EBCA6Bu);
}

// 20-bit ordering hash of a canonical m-mer (decides WHICH m-mer of a window is its minimizer; equal
// values are ties and every tied position is registered on the text side).  Text side and query
// side must use the very same function: the map kernel packs it above an 11-bit position and a
// strand bit.
template <int MM>
__device__ __forceinline__ u32
mmer_order(typename Mmer<MM>::type cm)
{
	// two 24-bit multiply-adds (full rate on CDNA; a 32-bit v_mul_lo_u32 is quarter rate and this runs
	// for every base of every read): low 12 bases and the bases above them, mixed into the top bits
	const u32 lo = (u32)cm, up = (u32)(cm >> 24);
	return (__umul24(up, 0x85EBCBu) + __umul24(lo, 0x9E3779u) + 0x7F4A7C15u) >> 12;
}

// the 30 bits of a canonical m-mer that a table entry keeps
template <int MM>
__device__ __forceinline__ u32
mmer_fp(typename Mmer<MM>::type cm)
{
	return ((mmer_fold(cm) ^ 0x68E31DA4u) * 0xB5297A4Du) >> 2;
}

template <int MM>
__device__ __forceinline__ u64
mtab_home(typename Mmer<MM>::type cm, u64 cap)
{
	// home slots are multiples of 4 (cap is one too): the map kernel reads four entries per round trip,
	// and an aligned group is one 32-byte sector instead of a span that straddles two
	u64 h = (u64)cm * 0x9E3779B97F4A7C15ull;
	h ^= h >> 29;
	return mulhi64(h * 0xD6E8FEB86659FD93ull, cap) & ~3ull;
}

// Seed table sharded over the ranks of a node (BASELINE configs[3]): the rank that owns the canonical m-mer
// cm -- by the top bits of a hash that is independent of the slot hash above, so that a shard's entries
// still spread over all of its slots
template <int MM>
__device__ __forceinline__ u32
seed_owner(typename Mmer<MM>::type cm, u32 n_owners)
{
	u64 h = ((u64)cm ^ 0x5851F42D4C957F2Dull) * 0xC2B2AE3D27D4EB4Full;
	h ^= h >> 32;
	return (u32)mulhi64(h * 0x9FB21C651E98DF25ull, (u64)n_owners);
}

__device__ __forceinline__ u32
bit_at(const u32* __restrict__ bits, u64 pos)
{
	return (bits[pos >> 5] >> (31 - (u32)(pos & 31))) & 1u;
}

// minimizer of the window that starts at base `pos` of a packed stream (window free of invalid
// bases): smallest ordering value and the LEFTMOST offset that has it
template <int MM>
__device__ __forceinline__ void
window_minimizer(const u64* __restrict__ codes, u64 pos, int w, u32& min_h, int& min_off)
{
	min_h = 0xFFFFFFFFu;
	min_off = 0;
	for (int o = 0; o < w; ++o) {
		const typename Mmer<MM>::type f = mmer_fw<MM>(codes, pos + (u64)o);
		const typename Mmer<MM>::type r = mmer_rc<MM>(f);
		const u32 h = mmer_order<MM>(f < r ? f : r);
		if (h < min_h) {
			min_h = h;
			min_off = o;
		}
	}
}

// exact value of the key c in the fallback table, -1 when absent
template <int KW>
__device__ __forceinline__ int
fallback_lookup(const BIndexView& bx, const Key<KW>& c)
{
	return table_lookup<KW>(bx.fallback, c);
}

// Serial (one lane) exact lookup of one window in the locality index: value of the window whose
// forward / reverse-complement keys are f / r (equal for a palindrome) and that starts at base p of the packed
// stream (wbase).  -1 = absent, 0 = ambiguous, > 0 = contig end.  The slow path, and the in-kernel
// definition the cooperative fast path has to agree with.
template <int KW, int MM>
__device__ __forceinline__ int
bindex_lookup_serial(
    const BIndexView& bx, const KeyGeom& g, const u64* __restrict__ codes, u64 pos,
    const Key<KW>& f, const Key<KW>& r)
{
	typedef typename Mmer<MM>::type mm_t;
	u32 min_h;
	int off = 0; // seed index: any m-mer of the window will do, the first one is as good as any
	if (!bx.dense)
		window_minimizer<MM>(codes, pos, bx.w, min_h, off);
	const mm_t mf = mmer_fw<MM>(codes, pos + (u64)off);
	const mm_t mr = mmer_rc<MM>(mf);
	const mm_t cm = mf < mr ? mf : mr;
	const u32 fp = mmer_fp<MM>(cm);
	const u32 rstrand = mf < mr ? 1u : 0u;
	u64 s = mtab_home<MM>(cm, bx.mtab_cap);
	for (;;) {
		const u64 e = bx.mtab[s];
		if (!(e >> 63))
			return -1;
		s = (s + 1 == bx.mtab_cap) ? 0 : s + 1;
		if (((u32)(e >> 32) & kFpMask) != fp)
			continue;
		const u32 tpos = (u32)e;
		if (tpos == kHeavyPos) {
			const bool lt = key_less(f, r);
			Key<KW> c;
#pragma unroll
			for (int j = 0; j < KW; ++j)
				c.w[j] = lt ? f.w[j] : r.w[j];
			if (key_eq(f, r)) // a palindrome lives in the fallback table under its damaged key
				c = key_palindrome_quirk(f, g);
			return fallback_lookup<KW>(bx, c);
		}
		const bool same = ((u32)(e >> 62) & 1u) == rstrand;
		// start of the text window that would hold this k-mer
		const u64 t = same ? (u64)tpos - (u64)off : (u64)tpos - (u64)(g.k - MM - off);
		const Key<KW> tk = window_key_at<KW>(bx.codes, t, g);
		if (!key_eq(tk, same ? f : r))
			continue;
		if (!bit_at(bx.visited, t))
			continue;
		return bit_at(bx.ambig, t) ? 0 : (int)bx.word_owner[t >> 5];
	}
}

} // namespace arks
