// Does the WIDTH of the load instruction decide how much a random 32-byte probe drags out of HBM on gfx950?
// profiles/r04_fetch_calibration.json: a 4-byte-per-lane stream makes 64-byte read requests (two per 128-byte L2
// line, each a miss of its own), an 8-byte-per-lane stream makes 128-byte requests -- and the random 32-byte probes of
// gather32 (u64 / uint4 loads) are tallied as "128 B" requests too.  If a probe fetches the whole 128-byte line, the
// "random-access ceiling" of 3.8e10 probes/s is 4.9 TB/s of HBM traffic -- a BANDWIDTH ceiling -- and probing with
// narrower loads (64-byte requests) would halve the traffic.  Arms, same random aligned 32-byte groups over a 32 GiB
// table: the group read as 2 x 16 B, 4 x 8 B, 8 x 4 B (volatile: no merging); only the first 16 B (1 x 16, 4 x 4);
// rates by HIP events here, request counts under rocprofv3 --pmc TCC_EA0_RDREQ_sum.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_width gather_width.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned u32;
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

// MODE 0: 2 x uint4, 1: 4 x u64, 2: 8 x u32, 3: 1 x uint4 (first half), 4: 4 x u32 (first half), 5: 2 x u64 (first half)
template <int MODE>
__global__ void
gather(const char* __restrict__ tab, u64 nslots32, u64 n_it, u64* out)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u64 acc = 0;
	for (u64 it = 0; it < n_it; ++it) {
		const u64 s = __umul64hi(mix(tid * 0x9E3779B97F4A7C15ull + it), nslots32);
		const char* p = tab + 32 * s;
		if (MODE == 0) {
			const volatile uint4* q = reinterpret_cast<const volatile uint4*>(p);
			acc += q[0].x ^ q[1].w;
		} else if (MODE == 1) {
			const volatile u64* q = reinterpret_cast<const volatile u64*>(p);
			acc += q[0] ^ q[1] ^ q[2] ^ q[3];
		} else if (MODE == 2) {
			const volatile u32* q = reinterpret_cast<const volatile u32*>(p);
			acc += q[0] ^ q[1] ^ q[2] ^ q[3] ^ q[4] ^ q[5] ^ q[6] ^ q[7];
		} else if (MODE == 3) {
			const volatile uint4* q = reinterpret_cast<const volatile uint4*>(p);
			acc += q[0].x;
		} else if (MODE == 4) {
			const volatile u32* q = reinterpret_cast<const volatile u32*>(p);
			acc += q[0] ^ q[1] ^ q[2] ^ q[3];
		} else {
			const volatile u64* q = reinterpret_cast<const volatile u64*>(p);
			acc += q[0] ^ q[1];
		}
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

template <int MODE>
int
run(const char* tab, u64 bytes, u64* out, const char* name)
{
	const unsigned blocks = 256 * 32;
	const u64 n_it = 64, n = (u64)blocks * 256 * n_it;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	gather<MODE><<<blocks, 256>>>(tab, bytes / 32, 4, out); // warm
	CK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(a));
		gather<MODE><<<blocks, 256>>>(tab, bytes / 32, n_it, out);
		CK(hipEventRecord(b));
		CK(hipEventSynchronize(b));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, a, b));
		best = ms < best ? ms : best;
	}
	printf("%-22s %8.3f ms  %6.2fe10 probes/s  (%.0f probes; x64 B = %.2f TB/s, x128 B = %.2f TB/s)\n", name, best,
	       n / (best * 1e-3) / 1e10, (double)n, n * 64.0 / (best * 1e-3) / 1e12, n * 128.0 / (best * 1e-3) / 1e12);
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
	if (run<0>(t, bytes, out, "32 B as 2 x 16 B")) return 1;
	if (run<1>(t, bytes, out, "32 B as 4 x 8 B")) return 1;
	if (run<2>(t, bytes, out, "32 B as 8 x 4 B")) return 1;
	if (run<3>(t, bytes, out, "16 B as 1 x 16 B")) return 1;
	if (run<5>(t, bytes, out, "16 B as 2 x 8 B")) return 1;
	if (run<4>(t, bytes, out, "16 B as 4 x 4 B")) return 1;
	return 0;
}
