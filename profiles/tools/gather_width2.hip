// Second round of profiles/tools/gather_width.hip (whose volatile loads turned out to be UNCACHED loads: every
// instruction its own fabric request -- not the question).  Here the cache-control bits of the gfx950 load are set
// explicitly (inline asm): the same random aligned 32-byte groups of a 32 GiB table, read as
//   2 x dwordx4            cached (what the compiler emits for the seed-table probe)
//   2 x dwordx4 nt         non-temporal
//   2 x dwordx4 sc1        / sc0 sc1: coherence scopes beyond the workgroup
//   1 x dwordx4 (16 B)     the same four ways: is it the line fill (128 B) or the request that costs?
// Rates by HIP events; fabric requests under rocprofv3 --pmc TCC_EA0_RDREQ_sum, FETCH_SIZE.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                                      \
	do {                                                                                           \
		hipError_t e_ = (x);                                                                       \
		if (e_ != hipSuccess) {                                                                    \
			printf("%s -> %s\n", #x, hipGetErrorString(e_));                                       \
			return 1;                                                                              \
		}                                                                                          \
	} while (0)

__device__ __forceinline__ u64
mix(u64 x)
{
	x ^= x >> 33;
	x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33;
	x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return x;
}

#define LOAD16(dst, ptr, off, bits) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off " " bits : "=v"(dst) : "v"(ptr) : "memory")

// BITS: 0 none, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0;  HALF: only the first 16 bytes
template <int BITS, bool HALF>
__global__ void
gather(const char* __restrict__ tab, u64 nslots32, u64 n_it, u64* out)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u32 acc = 0;
	for (u64 it = 0; it < n_it; ++it) {
		const u64 s = __umul64hi(mix(tid * 0x9E3779B97F4A7C15ull + it), nslots32);
		const char* p = tab + 32 * s;
		u32x4 a, b = { 0, 0, 0, 0 };
		if (BITS == 0) {
			LOAD16(a, p, 0, "");
			if (!HALF) LOAD16(b, p, 16, "");
		} else if (BITS == 1) {
			LOAD16(a, p, 0, "nt");
			if (!HALF) LOAD16(b, p, 16, "nt");
		} else if (BITS == 2) {
			LOAD16(a, p, 0, "sc1");
			if (!HALF) LOAD16(b, p, 16, "sc1");
		} else if (BITS == 3) {
			LOAD16(a, p, 0, "sc0 sc1");
			if (!HALF) LOAD16(b, p, 16, "sc0 sc1");
		} else {
			LOAD16(a, p, 0, "sc0");
			if (!HALF) LOAD16(b, p, 16, "sc0");
		}
		asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)::"memory");
		acc += a.x ^ b.w;
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

template <int BITS, bool HALF>
int
run(const char* tab, u64 bytes, u64* out, const char* name)
{
	const unsigned blocks = 256 * 32;
	const u64 n_it = 64, n = (u64)blocks * 256 * n_it;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	gather<BITS, HALF><<<blocks, 256>>>(tab, bytes / 32, 4, out);
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(a));
		gather<BITS, HALF><<<blocks, 256>>>(tab, bytes / 32, n_it, out);
		CK(hipEventRecord(b));
		CK(hipEventSynchronize(b));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, a, b));
		best = ms < best ? ms : best;
	}
	printf("%-26s %8.3f ms  %6.2fe10 probes/s\n", name, best, n / (best * 1e-3) / 1e10);
	return 0;
}

int
main()
{
	const u64 bytes = 32ull << 30;
	void* tab = nullptr;
	u64* out = nullptr;
	CK(hipMalloc(&tab, bytes));
	CK(hipMalloc(reinterpret_cast<void**>(&out), 64));
	CK(hipMemset(tab, 1, bytes));
	CK(hipDeviceSynchronize());
	const char* t = static_cast<const char*>(tab);
	if (run<0, false>(t, bytes, out, "32 B 2 x dwordx4")) return 1;
	if (run<1, false>(t, bytes, out, "32 B 2 x dwordx4 nt")) return 1;
	if (run<2, false>(t, bytes, out, "32 B 2 x dwordx4 sc1")) return 1;
	if (run<3, false>(t, bytes, out, "32 B 2 x dwordx4 sc0 sc1")) return 1;
	if (run<4, false>(t, bytes, out, "32 B 2 x dwordx4 sc0")) return 1;
	if (run<0, true>(t, bytes, out, "16 B 1 x dwordx4")) return 1;
	if (run<1, true>(t, bytes, out, "16 B 1 x dwordx4 nt")) return 1;
	if (run<2, true>(t, bytes, out, "16 B 1 x dwordx4 sc1")) return 1;
	if (run<3, true>(t, bytes, out, "16 B 1 x dwordx4 sc0 sc1")) return 1;
	if (run<4, true>(t, bytes, out, "16 B 1 x dwordx4 sc0")) return 1;
	return 0;
}
