// Dependent-issue latency of FP64 / FP32 VALU operations of one wave on gfx950, and the shader clock while nothing else runs (the Wavelets exact chain is one wave of dependent
// FP64 operations: what does a step cost at best?).  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_latency tools/dp_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(64) k_chain(double* out, long long* cyc, int n, double a, double b) {
    double x = a + threadIdx.x * 0.0; float xf = (float)a;
    double y = a * 0.5;
    const long long c0 = clock64(); const long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (MODE == 0) x = __builtin_fma(x, b, a);                                   // dependent v_fma_f64
            if (MODE == 1) xf = __builtin_fmaf(xf, (float)b, (float)a);                  // dependent v_fma_f32
            if (MODE == 2) { x = __builtin_fma(x, b, a); y = __builtin_fma(y, b, a); }   // two independent FP64 chains
            if (MODE == 3) x = x + b;                                                    // dependent v_add_f64
            if (MODE == 4) x = x * b;                                                    // dependent v_mul_f64
        }
    }
    const long long c1 = clock64(); const long long w1 = wall_clock64();
    if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
    out[threadIdx.x] = x + xf + y;
}
template <int MODE> void run(const char* name, int lanesActive) {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16);
    const int n = 1 << 20;
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_chain<MODE>, dim3(1), dim3(lanesActive), 0, 0, out, cyc, n, 1.0000001, 0.9999999); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("%-28s lanes %2d: %8.3f ms  %6.2f ns/op  s_memtime ticks/op %6.2f  wall(100MHz) %8.3f ms\n", name, lanesActive, ms, ms * 1e6 / n, (double)h[0] / n, h[1] / 1e5);
    }
}
int main() {
    run<0>("dependent v_fma_f64", 64); run<0>("dependent v_fma_f64", 16); run<0>("dependent v_fma_f64", 1);
    run<3>("dependent v_add_f64", 64); run<4>("dependent v_mul_f64", 64);
    run<1>("dependent v_fma_f32", 64); run<2>("two v_fma_f64 chains", 64);
    return 0;
}
