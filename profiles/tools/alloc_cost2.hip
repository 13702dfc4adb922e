// With the whole VRAM dirty (allocated, written, freed by this process): what does hipMalloc cost, what does the
// virtual-memory API cost, and does a block that is freed and taken again cost every time?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(what, expr) do { double t0 = now(); hipError_t e = (expr); hipDeviceSynchronize(); printf("%-52s %8.1f ms  %s\n", what, now() - t0, hipGetErrorString(e)); fflush(stdout); } while (0)
int main()
{
	const size_t G = 1ull << 30;
	void *a = nullptr, *b = nullptr;
	T("hipFree(0)", hipFree(0));
	T("hipMalloc 250 GB", hipMalloc(&a, 250 * G));
	T("hipMemset 250 GB", hipMemset(a, 0x5a, 250 * G));
	T("hipFree 250 GB", hipFree(a));
	T("hipMalloc 45 GB (VRAM dirty)", hipMalloc(&b, 45 * G));
	T("hipFree 45 GB", hipFree(b));
	T("hipMalloc 45 GB again", hipMalloc(&b, 45 * G));
	T("hipFree 45 GB", hipFree(b));
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	hipMemGenericAllocationHandle_t h1;
	void* va = nullptr;
	T("hipMemAddressReserve 45 GB", hipMemAddressReserve(&va, 45 * G, 0, nullptr, 0));
	T("hipMemCreate 45 GB (VRAM dirty)", hipMemCreate(&h1, 45 * G, &prop, 0));
	T("hipMemMap", hipMemMap(va, 45 * G, 0, h1, 0));
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	T("hipMemSetAccess", hipMemSetAccess(va, 45 * G, &acc, 1));
	unsigned* probe = nullptr;
	hipHostMalloc((void**)&probe, 4096, 0);
	T("copy first 4 KB out (what does fresh VMM memory hold?)", hipMemcpy(probe, va, 4096, hipMemcpyDeviceToHost));
	printf("first words: %08x %08x %08x %08x\n", probe[0], probe[1], probe[512], probe[1023]);
	T("copy 4 KB at +30 GB", hipMemcpy(probe, (char*)va + 30 * G, 4096, hipMemcpyDeviceToHost));
	printf("words at +30 GB: %08x %08x\n", probe[0], probe[1023]);
	T("hipMemset 45 GB", hipMemset(va, 0, 45 * G));
	T("hipMemUnmap", hipMemUnmap(va, 45 * G));
	T("hipMemRelease", hipMemRelease(h1));
	T("hipMemCreate 45 GB again", hipMemCreate(&h1, 45 * G, &prop, 0));
	T("hipMemMap", hipMemMap(va, 45 * G, 0, h1, 0));
	T("hipMemSetAccess", hipMemSetAccess(va, 45 * G, &acc, 1));
	T("hipMemset 45 GB", hipMemset(va, 0, 45 * G));
	T("hipMemUnmap", hipMemUnmap(va, 45 * G));
	T("hipMemRelease", hipMemRelease(h1));
	T("hipMalloc 90 GB (after all that)", hipMalloc(&a, 90 * G));
	T("hipFree", hipFree(a));
	return 0;
}
