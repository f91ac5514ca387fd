// Experiment: allocate a buffer through the HIP virtual-memory API (one physical handle, or one per chunk) and return its
// address, to compare the relax sweep on operators placed this way with operators from hipMalloc.
// Built on the box: hipcc -shared -fPIC -o gpurun_out/libvmmprobe.so tools/vmm_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern "C" void *vmm_alloc(size_t bytes, size_t chunk_bytes, size_t va_align)
{
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	int dev = 0;
	hipGetDevice(&dev);
	prop.location.id = dev;
	size_t gran = 0;
	if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return nullptr;
	if (chunk_bytes == 0) chunk_bytes = bytes;
	chunk_bytes = (chunk_bytes + gran - 1) / gran * gran;
	const size_t total = (bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
	void *va = nullptr;
	if (hipMemAddressReserve(&va, total, va_align, nullptr, 0) != hipSuccess) { fprintf(stderr, "reserve failed\n"); return nullptr; }
	for (size_t off = 0; off < total; off += chunk_bytes) {
		hipMemGenericAllocationHandle_t h;
		if (hipMemCreate(&h, chunk_bytes, &prop, 0) != hipSuccess) { fprintf(stderr, "create failed at %zu\n", off); return nullptr; }
		if (hipMemMap((char *)va + off, chunk_bytes, 0, h, 0) != hipSuccess) { fprintf(stderr, "map failed\n"); return nullptr; }
		hipMemRelease(h);
	}
	hipMemAccessDesc acc = {};
	acc.location = prop.location;
	acc.flags = hipMemAccessFlagsProtReadWrite;
	if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { fprintf(stderr, "set access failed\n"); return nullptr; }
	return va;
}

extern "C" void vmm_free(void *va, size_t bytes, size_t chunk_bytes)
{
	if (chunk_bytes == 0) chunk_bytes = bytes;
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	size_t gran = 4096;
	hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
	chunk_bytes = (chunk_bytes + gran - 1) / gran * gran;
	const size_t total = (bytes + chunk_bytes - 1) / chunk_bytes * chunk_bytes;
	hipDeviceSynchronize();
	hipMemUnmap(va, total);
	hipMemAddressFree(va, total);
}
