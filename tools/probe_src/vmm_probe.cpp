// HIP virtual-memory-management probe behind the draw-stream cache of canvas_cbs (round 6): address range reserved once, physical memory mapped piece by piece, costs of every call,
// D2H of 2.5 KB out of the range.  hipcc -O2 --offload-arch=gfx950 tools/probe_src/vmm_probe.cpp -o /tmp/vmm_probe && /tmp/vmm_probe
// Measured on an MI355X box: reserve 64 GB 0.013 ms; first 256 MB piece create 0.015 / map 19.4 / access 0.02 ms, later pieces ~0.01 ms each; 1 GB filled across four pieces in 0.18 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(uint32_t* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    CK(hipSetDevice(0));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f GB total %.1f GB\n", fr / 1e9, tot / 1e9);
    int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0)); printf("VMM supported attr: %d\n", vmm);
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended)); printf("granularity %zu\n", gran);
    const size_t VA = (size_t)64 << 30, CH = (size_t)256 << 20;
    void* base = nullptr; double t0 = now(); CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0)); printf("reserve 64 GB VA: %.3f ms\n", (now() - t0) * 1e3);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (int i = 0; i < 4; i++) {
        hipMemGenericAllocationHandle_t h; t0 = now(); CK(hipMemCreate(&h, CH, &prop, 0)); double t1 = now();
        CK(hipMemMap((char*)base + i * CH, CH, 0, h, 0)); double t2 = now();
        hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess((char*)base + i * CH, CH, &ad, 1)); double t3 = now();
        printf("granule %d (256 MB): create %.3f ms, map %.3f ms, access %.3f ms\n", i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3); hs.push_back(h);
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    k_fill<<<2048, 256, 0, s>>>((uint32_t*)base, 4 * CH / 4); CK(hipStreamSynchronize(s));
    t0 = now(); k_fill<<<2048, 256, 0, s>>>((uint32_t*)base, 4 * CH / 4); CK(hipStreamSynchronize(s)); printf("fill 1 GB across 4 granules: %.3f ms\n", (now() - t0) * 1e3);
    uint32_t v[2]; CK(hipMemcpy(v, (char*)base + CH - 4, 8, hipMemcpyDeviceToHost)); printf("values across the granule boundary: %u %u (expect %zu %zu)\n", v[0], v[1], CH / 4 - 1, CH / 4);
    // mapping while a kernel runs on another stream
    hipStream_t s2; CK(hipStreamCreate(&s2));
    k_fill<<<2048, 256, 0, s>>>((uint32_t*)base, 4 * CH / 4);
    { hipMemGenericAllocationHandle_t h; t0 = now(); CK(hipMemCreate(&h, CH, &prop, 0)); CK(hipMemMap((char*)base + 4 * CH, CH, 0, h, 0));
      hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess((char*)base + 4 * CH, CH, &ad, 1));
      printf("create+map+access with a kernel in flight: %.3f ms\n", (now() - t0) * 1e3); hs.push_back(h); }
    CK(hipStreamSynchronize(s));
    // 1 GB granule
    { hipMemGenericAllocationHandle_t h; t0 = now(); CK(hipMemCreate(&h, (size_t)1 << 30, &prop, 0)); CK(hipMemMap((char*)base + ((size_t)2 << 30), (size_t)1 << 30, 0, h, 0));
      hipMemAccessDesc ad = {}; ad.location.type = hipMemLocationTypeDevice; ad.location.id = 0; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess((char*)base + ((size_t)2 << 30), (size_t)1 << 30, &ad, 1));
      printf("1 GB granule create+map+access: %.3f ms\n", (now() - t0) * 1e3); hs.push_back(h); }
    for (size_t gb : {1, 8, 32}) { void* p = nullptr; t0 = now(); CK(hipMalloc(&p, gb << 30)); double t1 = now(); CK(hipFree(p)); printf("hipMalloc %zu GB: %.3f ms, free %.3f ms\n", gb, (t1 - t0) * 1e3, (now() - t1) * 1e3); }
    // D2H of 2.5 KB from the VMM range into pinned memory
    void* pin; CK(hipHostMalloc(&pin, 4096, 0));
    for (int r = 0; r < 3; r++) { t0 = now(); CK(hipMemcpyAsync(pin, (char*)base + 12345 * 4, 2496, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); printf("2.5 KB D2H + sync: %.1f us\n", (now() - t0) * 1e6); }
    printf("OK\n");
    return 0;
}
