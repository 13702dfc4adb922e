// What do big device allocations cost in a fresh process?  (round 6: the CLI's index build spends 5.5 of its 6.7 s in
// hipMalloc / hipFree of its 90 GB exact table and the 79 GB behind it.)   hipcc --offload-arch=gfx950 -O2 -o alloc_cost alloc_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(what, expr) do { double t0 = now(); hipError_t e = (expr); hipDeviceSynchronize(); printf("%-44s %8.1f ms  %s\n", what, now() - t0, hipGetErrorString(e)); } while (0)
int main()
{
	const size_t G = 1ull << 30;
	void *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr;
	T("hipFree(0) (runtime start)", hipFree(0));
	T("hipMalloc 90 GB", hipMalloc(&a, 90 * G));
	T("hipMemset 90 GB", hipMemset(a, 0, 90 * G));
	T("hipFree 90 GB", hipFree(a));
	T("hipMalloc 22 GB", hipMalloc(&b, 22 * G));
	T("hipMalloc 11 GB", hipMalloc(&c, 11 * G));
	T("hipMalloc 45 GB", hipMalloc(&d, 45 * G));
	T("hipFree 22", hipFree(b));
	T("hipFree 11", hipFree(c));
	T("hipFree 45", hipFree(d));
	T("hipMalloc 90 GB again", hipMalloc(&a, 90 * G));
	T("hipFree 90 GB again", hipFree(a));
	// the virtual-memory API: two physical blocks of 45 GB behind one address range
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t gran = 0;
	T("granularity", hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	printf("granularity %zu\n", gran);
	hipMemGenericAllocationHandle_t h1, h2;
	void* va = nullptr;
	T("hipMemAddressReserve 90 GB", hipMemAddressReserve(&va, 90 * G, 0, nullptr, 0));
	T("hipMemCreate 45 GB", hipMemCreate(&h1, 45 * G, &prop, 0));
	T("hipMemCreate 45 GB", hipMemCreate(&h2, 45 * G, &prop, 0));
	T("hipMemMap first", hipMemMap(va, 45 * G, 0, h1, 0));
	T("hipMemMap second", hipMemMap((char*)va + 45 * G, 45 * G, 0, h2, 0));
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	T("hipMemSetAccess 90 GB", hipMemSetAccess(va, 90 * G, &acc, 1));
	T("hipMemset 90 GB (mapped)", hipMemset(va, 0, 90 * G));
	T("hipMemUnmap second", hipMemUnmap((char*)va + 45 * G, 45 * G));
	T("hipMemRelease second", hipMemRelease(h2));
	T("hipMemset 45 GB (first half still mapped)", hipMemset(va, 1, 45 * G));
	T("hipMemUnmap first", hipMemUnmap(va, 45 * G));
	T("hipMemRelease first", hipMemRelease(h1));
	T("hipMemAddressFree", hipMemAddressFree(va, 90 * G));
	return 0;
}
