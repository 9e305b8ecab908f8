// micro-benchmark: dependent-issue latency of FP64 VALU operations on one wave (used to reason about the Wavelets chain kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k_chain(double* out, double a, double b, int iters, int mode) {
    double x = a + threadIdx.x, y = b;
    long long t0 = clock64();
    if (mode == 0) { for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = __builtin_fma(x, y, y); } }
    else if (mode == 1) { for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = x + y; } }
    else if (mode == 2) { for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) x = x * y; } }
    else { double z = x + 1.0; for (int i = 0; i < iters; i++) {       // two independent chains
#pragma unroll
        for (int k = 0; k < 64; k++) { x = __builtin_fma(x, y, y); z = __builtin_fma(z, y, y); } } x += z; }
    long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64 + mode] = (double)(t1 - t0) / (64.0 * iters);
}
int main() {
    double* d; hipMalloc(&d, 128 * 8);
    const char* names[4] = {"fma chain", "add chain", "mul chain", "2 independent fma chains (per pair)"};
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 1.0, 0.999999, 2000, mode);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 1.0, 0.999999, 20000, mode); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-40s %.2f clock64 ticks/op, %.2f ns/op\n", names[mode], h[64 + mode], ms * 1e6 / (64.0 * 20000));
    }
    return 0;
}
