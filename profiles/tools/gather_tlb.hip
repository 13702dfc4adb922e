// Random 32-byte gathers over a big table: what bounds the rate once the table outgrows ~1 GiB -- HBM or the reach
// of address translation?  ("what's weak" 6.)  Arms:
//   size sweep   hipMalloc tables of 1..64 GiB: where does the rate fall?
//   alloc        the same 64 GiB table from (a) hipMalloc, (b) hipExtMallocWithFlags(hipDeviceMallocContiguous),
//                (c) the virtual memory API: one VA range aligned to G, physical chunks of G, G in {2 MiB, 1 GiB, 2 GiB}
//   slice        probes of a launch confined to a window of S GiB that moves over the table with the iteration
//                number (what binning a launch's seeds by table slice would give)
// build: hipcc --offload-arch=gfx950 -O3 -o gather_tlb gather_tlb.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x)                                                                                                          \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                           \
			return false;                                                                                              \
		}                                                                                                              \
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

// every thread does n_it probes of 32 bytes; probe `it` of all threads falls into the window
// [win_base(it), win_base(it) + win_slots) of the table (win_slots = nslots: no confinement)
__global__ void
gather32(const uint4* __restrict__ tab, u64 nslots, u64 win_slots, u64 n_it, u64* out)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 nwin = nslots / win_slots;
	u64 acc = 0;
	for (u64 it = 0; it < n_it; ++it) {
		const u64 h = mix(tid * 0x9E3779B97F4A7C15ull + it);
		const u64 wbase = ((it * nwin) / n_it) * win_slots;
		const u64 s = wbase + __umul64hi(h, win_slots);
		const uint4* p = tab + 2 * s;
		const uint4 a = p[0], b = p[1];
		acc += a.x ^ b.w;
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

__global__ void
fill(uint4* tab, u64 n16)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x)
		tab[i] = make_uint4((unsigned)i, 1, 2, 3);
}

static double
rate(const uint4* tab, u64 bytes, u64 win_bytes, u64* out)
{
	const u64 nslots = bytes / 32, win = win_bytes / 32;
	const int blocks = 256 * 8, threads = 256;
	const u64 n_it = 512;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	gather32<<<blocks, threads>>>(tab, nslots, win, n_it, out);
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < 3; ++r)
		gather32<<<blocks, threads>>>(tab, nslots, win, n_it, out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	ms /= 3;
	return (double)blocks * threads * n_it / ms / 1e6; // G probes / s
}

static bool
vmm_table(u64 bytes, u64 gran, uint4** out_ptr, std::vector<hipMemGenericAllocationHandle_t>& handles)
{
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t gmin = 0, grec = 0;
	CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
	CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
	printf("  vmm granularity: minimum %zu, recommended %zu\n", gmin, grec);
	void* va = nullptr;
	CK(hipMemAddressReserve(&va, bytes, gran, nullptr, 0));
	printf("  reserved VA %p (aligned to %llu MiB: %s)\n", va, gran >> 20, ((u64)va % gran) == 0 ? "yes" : "NO");
	for (u64 off = 0; off < bytes; off += gran) {
		hipMemGenericAllocationHandle_t h;
		CK(hipMemCreate(&h, gran, &prop, 0));
		handles.push_back(h);
		CK(hipMemMap((char*)va + off, gran, 0, h, 0));
	}
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	CK(hipMemSetAccess(va, bytes, &acc, 1));
	*out_ptr = (uint4*)va;
	return true;
}

int
main(int argc, char** argv)
{
	const u64 GiB = 1ull << 30;
	const u64 big = (argc > 1 ? strtoull(argv[1], 0, 10) : 64) * GiB;
	u64* out;
	hipMalloc(&out, 8);
	printf("== size sweep (hipMalloc), 32-byte probes, G probes/s\n");
	for (u64 g : {1ull, 2ull, 4ull, 8ull, 16ull, 32ull, 64ull}) {
		if (g * GiB > big)
			break;
		uint4* tab;
		if (hipMalloc(&tab, g * GiB) != hipSuccess) {
			printf("alloc fail %llu\n", g);
			return 1;
		}
		fill<<<4096, 256>>>(tab, g * GiB / 16);
		printf("table %3llu GiB: %.2f   VA %p\n", g, rate(tab, g * GiB, g * GiB, out), (void*)tab);
		fflush(stdout);
		hipFree(tab);
	}
	printf("== slices of a %llu GiB hipMalloc table (window moves with the iteration)\n", big / GiB);
	{
		uint4* tab;
		if (hipMalloc(&tab, big) != hipSuccess)
			return 1;
		fill<<<4096, 256>>>(tab, big / 16);
		for (u64 mib : {64ull, 256ull, 512ull, 1024ull, 2048ull, 4096ull, 8192ull})
			printf("window %5llu MiB: %.2f\n", mib, rate(tab, big, mib << 20, out));
		printf("window    whole: %.2f\n", rate(tab, big, big, out));
		fflush(stdout);
		hipFree(tab);
	}
	printf("== allocation arms, %llu GiB, whole-table probes\n", big / GiB);
	{
		uint4* tab = nullptr;
		hipError_t e = hipExtMallocWithFlags((void**)&tab, big, hipDeviceMallocContiguous);
		if (e == hipSuccess) {
			fill<<<4096, 256>>>(tab, big / 16);
			printf("hipExtMallocWithFlags(Contiguous): %.2f   VA %p\n", rate(tab, big, big, out), (void*)tab);
			hipFree(tab);
		} else
			printf("hipExtMallocWithFlags(Contiguous) %llu GiB: %s\n", big / GiB, hipGetErrorString(e));
		(void)hipGetLastError();
		// contiguous pieces of 1 / 2 / 4 GiB, each its own allocation: does contiguity inside 1 GiB change the rate there?
		for (u64 g : {1ull, 4ull}) {
			e = hipExtMallocWithFlags((void**)&tab, g * GiB, hipDeviceMallocContiguous);
			if (e == hipSuccess) {
				fill<<<4096, 256>>>(tab, g * GiB / 16);
				printf("Contiguous %llu GiB alone: %.2f   VA %p\n", g, rate(tab, g * GiB, g * GiB, out), (void*)tab);
				hipFree(tab);
			} else
				printf("Contiguous %llu GiB: %s\n", g, hipGetErrorString(e));
			(void)hipGetLastError();
		}
		fflush(stdout);
	}
	for (u64 gran : {2ull << 20, 1ull << 30, 2ull << 30}) {
		printf("vmm, granule %llu MiB:\n", gran >> 20);
		uint4* tab = nullptr;
		std::vector<hipMemGenericAllocationHandle_t> hs;
		if (vmm_table(big, gran, &tab, hs)) {
			fill<<<4096, 256>>>(tab, big / 16);
			if (hipDeviceSynchronize() != hipSuccess) {
				printf("  fill failed\n");
				return 1;
			}
			printf("  whole: %.2f   window 1 GiB: %.2f\n", rate(tab, big, big, out), rate(tab, big, GiB, out));
		}
		(void)hipGetLastError();
		if (tab) {
			hipMemUnmap(tab, big);
			hipMemAddressFree(tab, big);
		}
		for (auto h : hs)
			hipMemRelease(h);
		fflush(stdout);
	}
	return 0;
}
