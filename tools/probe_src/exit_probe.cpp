// what leaving a process costs once it has used the GPU: canvas_create (+ optionally one device allocation of <MB> megabytes that is written once) and _exit, against the caller's
// clock.  build: g++ -O2 exit_probe.cpp -I include -L canvas_amd -lcanvas_hip -Wl,-rpath,canvas_amd ...  (tools/exit_probe.sh)
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
#include <sys/mman.h>
#include <cstdint>
#include <vector>
#include "canvas_hip.h"
static double real_now() { struct timespec t; clock_gettime(CLOCK_REALTIME, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
int main(int argc, char** argv) {
    const double t0 = real_now();
    const int mode = argc > 1 ? atoi(argv[1]) : 0;        // 0: no GPU at all; 1: context only; 2: context + device buffer; 3: as 2, freed and destroyed before leaving;
                                                          // 4: no GPU, <MB> of host memory in transparent huge pages, touched; 5: context + that host memory; 6: as 5, uploaded
                                                          // from (canvas_memcpy_h2d of every MB); 7: as 6, leaving through a forked child that inherits the pages
    const long mb = argc > 2 ? atol(argv[2]) : 2048;
    canvas_ctx* ctx = nullptr; void* d = nullptr;
    if (mode >= 1 && mode < 4) { ctx = canvas_create(0); if (!ctx) { fprintf(stderr, "no GPU\n"); return 1; } }
    if (mode >= 2 && mode < 4) { d = canvas_device_malloc(ctx, (int64_t)mb << 20); std::vector<char> h(1 << 20, 1); for (long i = 0; i < mb; i += 64) canvas_memcpy_h2d(ctx, (char*)d + (i << 20), h.data(), 1 << 20); canvas_synchronize(ctx); }
    char* host = nullptr;
    if (mode >= 4) {
        const size_t len = (size_t)mb << 20, H = (size_t)2 << 20;
        host = (char*)mmap(nullptr, len + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        host = (char*)(((uintptr_t)host + H - 1) & ~(uintptr_t)(H - 1));
        madvise(host, len, MADV_HUGEPAGE);
        for (size_t i = 0; i < len; i += 4096) host[i] = 1;
        if (mode >= 5) { ctx = canvas_create(0); if (!ctx) return 1; }
        if (mode >= 6) { d = canvas_device_malloc(ctx, 64 << 20); for (size_t i = 0; i + (64u << 20) <= len; i += (64u << 20)) canvas_memcpy_h2d(ctx, d, host + i, 64 << 20); canvas_synchronize(ctx); }
        if (mode >= 7) { if (fork() == 0) _exit(0); }
    }
    const double t1 = real_now();
    if (mode == 3) { canvas_device_free(ctx, d); canvas_destroy(ctx); }
    fprintf(stderr, "mode %d: main entered %.4f, work %.4f s, leaving at %.4f\n", mode, t0, t1 - t0, real_now());
    _exit(0);
}
