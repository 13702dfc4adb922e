// lds_poisoner -- test tooling (nothing in the product links this): ANOTHER PROCESS on the same GPU that keeps filling
// the LDS of every CU with a pattern, the way a neighbour's kernels would.  LDS is not cleared between kernels, so a
// kernel of libarks_hip that reads LDS it has not written meets this pattern instead of what its own last workgroup
// left there -- the fault class that only shows when processes share a device (tests/test_zz_gpu_stress.py,
// `arcs --ranks` with --share-devices).
//   hipcc --offload-arch=gfx950 -O2 -o lds_poisoner tests/lds_poisoner.hip ; lds_poisoner <seconds> <hex word | "rand">
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

__global__ void
fill_lds(unsigned word, unsigned salt)
{
	extern __shared__ unsigned lds[];
	const unsigned n = 65536u / 4u; // 64 KB per workgroup: two or more of them cover a CU's 160 KB
	for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
		lds[i] = salt ? (word ^ (i * 2654435761u + salt * blockIdx.x)) : word;
	__syncthreads();
	if (lds[(threadIdx.x * 7u) % n] == 0x12345u && salt == 0xFFFFFFFFu) // (keeps the stores)
		__builtin_trap();
}

int
main(int argc, char** argv)
{
	const double secs = argc > 1 ? atof(argv[1]) : 30.0;
	const bool rnd = argc > 2 && !strcmp(argv[2], "rand");
	const unsigned word = argc > 2 && !rnd ? (unsigned)strtoul(argv[2], nullptr, 16) : 0xFFFFFFFFu;
	hipDeviceProp_t p;
	if (hipGetDeviceProperties(&p, 0) != hipSuccess)
		return 1;
	const auto t0 = std::chrono::steady_clock::now();
	unsigned long launches = 0;
	unsigned salt = 1;
	while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
		for (int i = 0; i < 16; ++i, ++launches)
			fill_lds<<<(unsigned)p.multiProcessorCount * 2u, 256, 65536>>>(word, rnd ? salt++ : 0u);
		if (hipDeviceSynchronize() != hipSuccess)
			return 2;
		std::this_thread::sleep_for(std::chrono::microseconds(200)); // leave the device to the others most of the time
	}
	printf("lds_poisoner: %lu launches of %d workgroups, word %08x%s\n", launches, p.multiProcessorCount * 2, word, rnd ? " (random)" : "");
	return 0;
}
