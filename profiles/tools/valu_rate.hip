// valu_rate.hip -- how fast one SIMD of gfx950 issues plain 32-bit integer VALU instructions of a wave64
// (the ceiling the map kernel is held against in DESIGN.md section 4).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate profiles/tools/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void __launch_bounds__(64)
spin(unsigned* out, int iters)
{
	unsigned v[CHAINS];
#pragma unroll
	for (int c = 0; c < CHAINS; ++c)
		v[c] = threadIdx.x + c;
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int c = 0; c < CHAINS; ++c)
			asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(i)); // one VALU per chain, the chains independent
	}
	unsigned s = 0;
#pragma unroll
	for (int c = 0; c < CHAINS; ++c)
		s ^= v[c];
	if (s == 0x12345678u)
		out[0] = s;
}

int
main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	unsigned* d;
	hipMalloc(&d, 4);
	const int cus = p.multiProcessorCount, iters = 200000;
	for (int waves_per_simd : { 1, 2, 4, 8 }) {
		const int blocks = cus * 4 * waves_per_simd;
		hipEvent_t a, b;
		hipEventCreate(&a), hipEventCreate(&b);
		spin<16><<<blocks, 64>>>(d, 100);
		hipDeviceSynchronize();
		hipEventRecord(a);
		spin<16><<<blocks, 64>>>(d, iters);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		const double instr = (double)blocks * iters * 16; // wave instructions
		const double per_simd_per_s = instr / (cus * 4) / (ms * 1e-3);
		std::printf("%d waves/SIMD: %.3f ms, %.3g wave-VALU/s per SIMD = one per %.2f cycles at %.2f GHz (clock rate as reported: %d kHz)\n",
		            waves_per_simd, ms, per_simd_per_s, (p.clockRate * 1e3) / per_simd_per_s, p.clockRate / 1e6, p.clockRate);
	}
	return 0;
}
