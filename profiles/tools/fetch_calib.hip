// What does rocprofv3's FETCH_SIZE count on gfx950 for the access classes of map_reads_s_kernel?  ("next" 4:
// the guide -- MI355X_MICROARCH.md, HBM -- says 1/2 of the bytes for a wide coalesced stream, "uncalibrated" otherwise.)
// Kernels with KNOWN byte counts over 8 GiB tables (far beyond the 256 MiB Infinity Cache), one launch each:
//   stream16 / stream8 / stream4   coalesced streaming reads of 16 / 8 / 4 bytes per lane (the kernel streams its packed
//                                  words 8 B per lane, lengths and N masks 4 B per lane)
//   gather32                       random aligned 32-byte reads (a seed-table probe: four u64 entries)
//   gather16x5                     runs of five consecutive 16-byte records at a random 16-byte-aligned place (the text
//                                  records along a diagonal: a 128-base read faces five text words)
//   write4                         coalesced 4-byte stores (the conreci array), for WRITE_SIZE
// Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE, TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum in passes of their
// own); profiles/tools/fetch_calib.py divides the known bytes by what the counter says.
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned long long u64;
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

template <typename T>
__global__ void
stream_kernel(const T* __restrict__ p, u64 n, u64* out)
{
	u64 acc = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
		const T v = p[i];
		acc += *reinterpret_cast<const unsigned*>(&v);
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

__global__ void
gather32_kernel(const u64* __restrict__ tab, u64 nslots32, u64 n_it, u64* out)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u64 acc = 0;
	for (u64 it = 0; it < n_it; ++it) {
		const u64 s = __umul64hi(mix(tid * 0x9E3779B97F4A7C15ull + it), nslots32);
		const u64* p = tab + 4 * s;
		acc += p[0] ^ p[1] ^ p[2] ^ p[3]; // (as probe_minimizer_table reads a group of four entries)
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

// lanes in groups of 5: a group reads 5 consecutive 16-byte records (as S4's staging slots do)
__global__ void
gather16x5_kernel(const ulonglong2* __restrict__ tab, u64 nrec, u64 n_it, u64* out)
{
	const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	const u64 grp = tid / 5, sub = tid % 5;
	u64 acc = 0;
	for (u64 it = 0; it < n_it; ++it) {
		const u64 s = __umul64hi(mix(grp * 0x9E3779B97F4A7C15ull + it), nrec - 8);
		const ulonglong2 v = tab[s + sub];
		acc += v.x ^ v.y;
	}
	if (acc == 0x1234567)
		out[0] = acc;
}

__global__ void
write4_kernel(unsigned* __restrict__ p, u64 n)
{
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
		p[i] = (unsigned)i;
}

int
main()
{
	const u64 bytes = 8ull << 30;
	void* tab = nullptr;
	u64* out = nullptr;
	CK(hipMalloc(&tab, bytes));
	CK(hipMalloc(reinterpret_cast<void**>(&out), 64));
	CK(hipMemset(tab, 1, bytes));
	CK(hipDeviceSynchronize());
	const unsigned blocks = 256 * 32;
	// streams: the whole table once
	stream_kernel<uint4><<<blocks, 256>>>(static_cast<const uint4*>(tab), bytes / 16, out);
	CK(hipDeviceSynchronize());
	printf("kernel=stream_kernel<uint4> requested_bytes=%llu lines64_bytes=%llu\n", bytes, bytes);
	stream_kernel<u64><<<blocks, 256>>>(static_cast<const u64*>(tab), bytes / 8, out);
	CK(hipDeviceSynchronize());
	printf("kernel=stream_kernel<unsigned long long> requested_bytes=%llu lines64_bytes=%llu\n", bytes, bytes);
	stream_kernel<unsigned><<<blocks, 256>>>(static_cast<const unsigned*>(tab), bytes / 4 / 2, out);
	CK(hipDeviceSynchronize());
	printf("kernel=stream_kernel<unsigned int> requested_bytes=%llu lines64_bytes=%llu\n", bytes / 2, bytes / 2);
	// gathers: 2^27 probes (the 64-byte line of a probe is fetched for 32 bytes of it; repeats among 2^27 draws over
	// 2^27 lines of 8 GiB are ~37 %: distinct lines = (1 - 1/e) 2^27 -- but a repeat seconds later has left the caches)
	const u64 n_thr = (u64)blocks * 256, n_it = 64;
	gather32_kernel<<<blocks, 256>>>(static_cast<const u64*>(tab), bytes / 32, n_it, out);
	CK(hipDeviceSynchronize());
	printf("kernel=gather32_kernel requested_bytes=%llu lines64_bytes=%llu\n", n_thr * n_it * 32, n_thr * n_it * 64);
	gather16x5_kernel<<<blocks, 320>>>(static_cast<const ulonglong2*>(tab), bytes / 16, n_it, out);
	CK(hipDeviceSynchronize());
	// 80 bytes at a random 16-byte-aligned offset touch 2 lines of 64 bytes (offsets 0, 16, 32, 48: 80 bytes span 2, 2, 2, 2)
	printf("kernel=gather16x5_kernel requested_bytes=%llu lines64_bytes=%llu\n", (u64)blocks * 320 * n_it * 16,
	       (u64)blocks * 320 / 5 * n_it * 128);
	write4_kernel<<<blocks, 256>>>(static_cast<unsigned*>(tab), bytes / 4 / 2);
	CK(hipDeviceSynchronize());
	printf("kernel=write4_kernel requested_bytes=%llu lines64_bytes=%llu\n", bytes / 2, bytes / 2);
	return 0;
}
