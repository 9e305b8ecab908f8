// micro-benchmark behind DESIGN §8 "bin tail": what does one sparse cache-line touch cost?  k_bin_resolve opens, per bin, one 64-byte slice of
// bases, one of hits and one mask word, the bins lying ~640 positions apart.  Variants: bytes per touch (8 / 64 / 128), number of arrays (1-3),
// plain vs non-temporal loads.  If 64 B and 128 B per touch cost the same, the L2 fills whole 128-byte lines and the kernel is HBM-bound on 3 lines per bin.
//   hipcc -O3 --offload-arch=gfx950 tools/line_probe.hip -o tools/line_probe && tools/line_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LANES, bool NT>
__global__ void __launch_bounds__(256) k_touch(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, const uint8_t* __restrict__ c, int narr,
                                               long long nbins, int stride, uint32_t* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = t / LANES; const int sub = (int)(t % LANES);
    if (i >= nbins) return;
    // pseudo-random jitter so that the touches are not perfectly periodic (as the bin boundaries are not)
    const uint32_t h = (uint32_t)i * 2654435761u;
    const long long off = ((i * stride + (h >> 26)) & (LANES > 4 ? ~127ll : ~63ll)) + 16 * sub;
    uint32_t acc = 0;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    auto ld = [&](const uint8_t* p) { const u4* q = reinterpret_cast<const u4*>(p + off); u4 v = NT ? __builtin_nontemporal_load(q) : *q; acc += v.x ^ v.y ^ v.z ^ v.w; };
    ld(a); if (narr > 1) ld(b); if (narr > 2) ld(c);
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_touch8(const uint64_t* __restrict__ a, long long nbins, int stride, uint32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nbins) return;
    const uint32_t h = (uint32_t)i * 2654435761u;
    const long long off = ((i * stride + (h >> 26)) & ~63ll) >> 3;
    const uint64_t v = a[off >> 3];      // the mask: one word per 64 positions, i.e. 1/8 of the byte offset
    if (v == 0x1234567812345678ull) out[0] = 1;
}
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ a, long long n16, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) { uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const long long L = 3088269832ll, nbins = 4786578; const int stride = 645;
    uint8_t *a, *b, *c; uint32_t* out;
    CK(hipMalloc(&a, L + 4096)); CK(hipMalloc(&b, L + 4096)); CK(hipMalloc(&c, L + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 1, L + 4096)); CK(hipMemset(b, 2, L + 4096)); CK(hipMemset(c, 3, L + 4096));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double bytes64, double bytes128) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4*)c, (long long)(600ll << 20) / 16, out);      // evict (600 MB > the 256 MB Infinity Cache)
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-58s %8.1f us   %6.2f TB/s if 64 B/line, %6.2f TB/s if 128 B/line\n", name, best * 1e3, bytes64 / best / 1e9, bytes128 / best / 1e9);
    };
    const double T = (double)nbins;
#define RUN(LANES, NT, NARR, NAME) timeit(NAME, [&]() { hipLaunchKernelGGL((k_touch<LANES, NT>), dim3((unsigned)((nbins * LANES + 255) / 256)), dim3(256), 0, 0, a, b, c, NARR, nbins, stride, out); }, \
                                          T * NARR * 64.0 * ((LANES + 3) / 4), T * NARR * 128.0 * (LANES > 4 ? 1 : 1))
    RUN(4, false, 1, "1 array, 64 B per bin (4 lanes x 16 B)");
    RUN(8, false, 1, "1 array, 128 B per bin (8 lanes x 16 B)");
    RUN(4, false, 2, "2 arrays, 64 B per bin each");
    RUN(4, false, 3, "3 arrays, 64 B per bin each (k_bin_resolve's pattern)");
    RUN(8, false, 3, "3 arrays, 128 B per bin each");
    RUN(4, true, 3, "3 arrays, 64 B each, non-temporal loads");
    RUN(1, false, 3, "3 arrays, 16 B per bin each (1 lane)");
    timeit("mask only: one 8-byte word per bin", [&]() { hipLaunchKernelGGL(k_touch8, dim3((unsigned)((nbins + 255) / 256)), dim3(256), 0, 0, (const uint64_t*)a, nbins, stride, out); }, T * 64.0, T * 128.0);
    timeit("streaming read of 3.09 GB", [&]() { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, (const uint4*)a, L / 16, out); }, (double)L, (double)L);
    // the same touch count at a 10x coarser stride (one touch per ~6.4 kB): DRAM page locality gone
    {
        const long long nb2 = nbins / 10; const int st2 = 6450;
        timeit("3 arrays, 64 B each, 10x fewer bins at 10x the stride", [&]() { hipLaunchKernelGGL((k_touch<4, false>), dim3((unsigned)((nb2 * 4 + 255) / 256)), dim3(256), 0, 0, a, b, c, 3, nb2, st2, out); }, nb2 * 3 * 64.0, nb2 * 3 * 128.0);
    }
    return 0;
}
