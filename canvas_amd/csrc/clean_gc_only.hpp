// CanvasClean -g alone (BASELINE configs[1]): RemoveBinsWithExtremeGC + NormalizeByGC (MedianByGC), CanvasClean.cs:163-237, in THREE launches.
//
// The general chain (clean_fast.hpp) serves every flag combination with seven launches: two compactions, grouped keys, counting selects.  With -g alone nothing is removed in
// front of the GC statistics, the counts of a .binned file are small integers, and the whole stage is
//   K1  k_go_count    one sweep over (chr, gc, count): a counter per (GC bucket, count value) of the autosomal bins (device-scope atomics: ~10 k hot addresses, they pipeline);
//   K2  k_go_decide   one workgroup per GC bucket: bucket size -> stripped or kept (the threshold is min(100, max(-w, n / 101)) = 100 on this path, so a kept bucket has its
//                     own median: no neighbour-weighted quantile), SortedList.Median() of the bucket from its counters; kept rows are added into the genome's row and the
//                     last workgroup to arrive takes the genome's median; every row is cleared by the workgroup that read it (the buffer is zero for the next call);
//   K3  k_go_apply    persistent: a workgroup loads its contiguous range of bins into registers, counts the bins it keeps, publishes the count, waits for the counts of the
//                     ranges in front of it (their owners loaded their ranges before they published, so compacting in place over them is safe), and writes
//                     count = (float)(globalMedian * (double)count / median) behind the prefix.
// Anything outside the assumptions (a count that is not an integer in [0, GO_VMAX), a GC value outside 0..100) raises a flag in K1; K3 then leaves the arrays untouched and the
// caller takes the general chain.  Results are the reference's bit for bit (tests/test_clean_gpu.py::test_clean_gc_only_*).
// MEASURED (30x genome, 2.53 M bins): k_go_count 140 us, k_go_decide 27 us, k_go_apply 138 us = 0.33 ms per call against 0.11 ms for the general chain, whose grouped keys let it
// count in LDS.  Kept as an opt-in (CANVAS_CLEAN_GC_ONLY_3K=1) and as the record of what three launches cost here; what would make it win is in DESIGN.md section 8.
#pragma once

#define GO_VMAX 4096          // count values with a counter of their own
#define GO_KMAX 16            // bins per thread of the apply kernel (registers): n <= grid * 256 * GO_KMAX
#define GO_SENT 0xFFFFFFFFu   // "not published yet"
struct GoDec {                // decisions of K2, read by K3 (and mirrored to the host by K3's last workgroup)
    double med[NGC]; double globalMedian;
    uint32_t keep[NGC];
    uint32_t bad, noop, pad0, pad1;      // bad: take the general chain; noop: no bucket survives the strip -> the bins stay as they are (CanvasClean.cs:501-503)
    unsigned long long nOut, nKeptAuto;
};
struct GoIsAuto { uint8_t v[256]; };

#define GO_REP 8              // one replica of the counters per XCD: a workgroup adds into the replica of the XCD it runs on with L2-resident atomics (workgroup scope: no
                              // sc1, the line stays in that XCD's L2, which every CU of the XCD shares); device-scope atomics on one table were 450 us for 2.5 M bins (≈ 5.6 G/s)
#define GO_PUB 16             // hand-off slots of the apply kernel: one 64-byte line each (polls of one line serialise at ≈ 12 ns)
__device__ __forceinline__ unsigned go_xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u; }      // HW_REG_XCC_ID[3:0]
__global__ void __launch_bounds__(256) k_go_count(const int32_t* __restrict__ chr, const int32_t* __restrict__ gc, const float* __restrict__ count, long long n, const GoIsAuto isAuto, int nchr,
                                                  uint32_t* __restrict__ cnt /* [GO_REP][NGC][GO_VMAX], zero */, GoDec* __restrict__ dec, unsigned long long* __restrict__ nCounted) {
    __shared__ unsigned long long sh16[16];
    uint32_t* __restrict__ mineRep = cnt + (size_t)go_xcc_id() * NGC * GO_VMAX;
    uint32_t bad = 0; unsigned long long counted = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int32_t c = chr[i], g = gc[i]; const float x = count[i];
        if ((uint32_t)g >= (uint32_t)NGC || (uint32_t)c >= (uint32_t)nchr) { bad = 1; continue; }
        if (!isAuto.v[c]) continue;                                            // GetCountsByGC / the bucket sizes: autosomes only (EnrichmentUtilities.cs:65-84, CanvasClean.cs:212-220)
        const int k = (int)x;
        if (!(x >= 0.0f && x < (float)GO_VMAX) || (float)k != x) { bad = 1; continue; }
        __hip_atomic_fetch_add(&mineRep[(size_t)g * GO_VMAX + k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        counted++;
    }
    if (bad) dec->bad = 1u;
    // how many bins went into the counters (one device-scope atomic per workgroup): k_go_decide compares it with what it finds — should the replicas ever not be per XCD
    // (a placement this code does not know), counts would be lost, the sums differ, and the call falls back to the general chain
    counted = cf_block_sum_u64(counted, sh16);
    if (threadIdx.x == 0 && counted) atomicAdd(nCounted, counted);
}

// SortedList<float>.Median() of a multiset given as counters c[k] of the values k = 0 .. GO_VMAX-1 (one row in LDS): (v[n/2 - 1] + v[n/2]) / 2 in float for an even n
__device__ __forceinline__ float go_median_of_row(const uint32_t* __restrict__ row /* LDS, GO_VMAX */, uint32_t total, uint32_t* sh4, int* sV /* [2] */) {
    // every thread owns GO_VMAX / 256 consecutive values
    constexpr int PER = GO_VMAX / 256;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) mine += row[threadIdx.x * PER + j];
    uint32_t tot; uint32_t ex = cf_excl_scan256(mine, sh4, &tot);
    const uint32_t r0 = (total - 1) / 2, r1 = total / 2;
    uint32_t cum = ex;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t c = row[threadIdx.x * PER + j];
        if (r0 >= cum && r0 < cum + c) sV[0] = threadIdx.x * PER + j;
        if (r1 >= cum && r1 < cum + c) sV[1] = threadIdx.x * PER + j;
        cum += c;
    }
    __syncthreads();
    const float a = (float)sV[0], b = (float)sV[1];
    return (total & 1u) ? b : (a + b) / 2.0f;
}
__global__ void __launch_bounds__(256) k_go_decide(uint32_t* __restrict__ cnt, uint32_t* __restrict__ all /* [GO_VMAX], zero */, GoDec* __restrict__ dec, uint32_t* __restrict__ tick,
                                                   uint32_t* __restrict__ pub, int npub, int threshold, uint32_t* __restrict__ tick3, unsigned long long* __restrict__ nCounted, unsigned long long* __restrict__ nFound) {
    __shared__ uint32_t sRow[GO_VMAX];
    __shared__ uint32_t sh4[4];
    __shared__ unsigned long long sh16[16];
    __shared__ int sV[2];
    __shared__ int sLast;
    constexpr int PER = GO_VMAX / 256;
    const int g = blockIdx.x;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < GO_REP; r++) { uint32_t* __restrict__ q = cnt + ((size_t)r * NGC + g) * GO_VMAX + threadIdx.x * PER + j; c += *q; *q = 0u; }      // (read and cleared: zero for the next call)
        sRow[threadIdx.x * PER + j] = c; mine += c;
    }
    const uint32_t total = (uint32_t)cf_block_sum_u64(mine, sh16);
    if (threadIdx.x == 0 && total) atomicAdd(nFound, (unsigned long long)total);
    const bool kept = total >= (uint32_t)threshold;                           // RemoveBinsWithExtremeGC: counts[gc] < threshold -> removed (CanvasClean.cs:226-235)
    float med = 0.0f;
    if (kept) {                                                               // (uniform over the workgroup)
        med = go_median_of_row(sRow, total, sh4, sV);
#pragma unroll
        for (int j = 0; j < PER; j++) { const uint32_t c = sRow[threadIdx.x * PER + j]; if (c) atomicAdd(&all[threadIdx.x * PER + j], c); }
    }
    if (threadIdx.x == 0) { dec->med[g] = (double)med; dec->keep[g] = kept ? 1u : 0u; }      // (read by the next launch)
    if (!cf_arrive_last(tick, NGC, &sLast)) return;
    // ---- last workgroup: the genome's median over the kept buckets, the sentinels of K3's hand-off, the tickets back to zero
    mine = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { const uint32_t c = all[threadIdx.x * PER + j]; sRow[threadIdx.x * PER + j] = c; mine += c; all[threadIdx.x * PER + j] = 0u; }
    const uint32_t tAll = (uint32_t)cf_block_sum_u64(mine, sh16);
    float gm = 0.0f;
    if (tAll > 0) gm = go_median_of_row(sRow, tAll, sh4, sV);
    for (int i = threadIdx.x; i < npub; i += 256) pub[(size_t)i * GO_PUB] = GO_SENT;
    if (threadIdx.x == 0) {
        if (*nCounted != *nFound) dec->bad = 1u;                               // (both complete: every workgroup's atomic is in front of its ticket)
        *nCounted = 0ull; *nFound = 0ull;
        dec->globalMedian = (double)gm; dec->noop = tAll == 0 ? 1u : 0u; *tick = 0u; *tick3 = 0u;
    }
}

// K3: in-place apply + strip.  Workgroup w owns the bins [w * 4096, (w + 1) * 4096), thread t of it the GO_KMAX consecutive bins behind a + t * GO_KMAX: four 16-byte loads
// per column, and a kept bin's slot is (kept bins of the ranges in front) + (kept bins of the threads in front) + (kept bins of the thread in front of it).
__global__ void __launch_bounds__(256) k_go_apply(int32_t* __restrict__ chr, int32_t* __restrict__ start, int32_t* __restrict__ stop, int32_t* __restrict__ gc, float* __restrict__ count, long long n,
                                                  const GoDec* __restrict__ dec, uint32_t* __restrict__ pub, uint32_t* __restrict__ tick3, GoDec* __restrict__ hostDec, unsigned* __restrict__ hostSeq, unsigned seq) {
    __shared__ double sMed[NGC];
    __shared__ uint32_t sKeep[NGC];
    __shared__ uint32_t sh4[4];
    __shared__ unsigned long long sh16[16];
    __shared__ int sLast;
    const uint32_t bad = dec->bad, noop = dec->noop;
    if (bad || noop) {                                                         // nothing is touched; the last workgroup reports (and clears the flag for the next call)
        if (cf_arrive_last(tick3, gridDim.x, &sLast) && threadIdx.x == 0) { hostDec->bad = bad; hostDec->noop = noop; hostDec->nOut = (unsigned long long)n; cvx_mail_publish(hostSeq, seq); const_cast<GoDec*>(dec)->bad = 0u; }
        return;
    }
    for (int i = threadIdx.x; i < NGC; i += 256) { sMed[i] = dec->med[i]; sKeep[i] = dec->keep[i]; }
    const double gm = dec->globalMedian;
    __syncthreads();
    const long long a = ((long long)blockIdx.x * 256 + threadIdx.x) * GO_KMAX;
    int32_t vc[GO_KMAX], vs[GO_KMAX], ve[GO_KMAX], vg[GO_KMAX]; uint32_t vx[GO_KMAX];
    if (a + GO_KMAX <= n) {
#pragma unroll
        for (int q = 0; q < GO_KMAX / 4; q++) {
            const uint4 c4 = gload_uint4(as_global(reinterpret_cast<const uint32_t*>(chr)) + a + 4 * q), s4 = gload_uint4(as_global(reinterpret_cast<const uint32_t*>(start)) + a + 4 * q),
                        e4 = gload_uint4(as_global(reinterpret_cast<const uint32_t*>(stop)) + a + 4 * q), g4 = gload_uint4(as_global(reinterpret_cast<const uint32_t*>(gc)) + a + 4 * q),
                        x4 = gload_uint4(as_global(reinterpret_cast<const uint32_t*>(count)) + a + 4 * q);
            vc[4 * q] = (int32_t)c4.x; vc[4 * q + 1] = (int32_t)c4.y; vc[4 * q + 2] = (int32_t)c4.z; vc[4 * q + 3] = (int32_t)c4.w;
            vs[4 * q] = (int32_t)s4.x; vs[4 * q + 1] = (int32_t)s4.y; vs[4 * q + 2] = (int32_t)s4.z; vs[4 * q + 3] = (int32_t)s4.w;
            ve[4 * q] = (int32_t)e4.x; ve[4 * q + 1] = (int32_t)e4.y; ve[4 * q + 2] = (int32_t)e4.z; ve[4 * q + 3] = (int32_t)e4.w;
            vg[4 * q] = (int32_t)g4.x; vg[4 * q + 1] = (int32_t)g4.y; vg[4 * q + 2] = (int32_t)g4.z; vg[4 * q + 3] = (int32_t)g4.w;
            vx[4 * q] = x4.x; vx[4 * q + 1] = x4.y; vx[4 * q + 2] = x4.z; vx[4 * q + 3] = x4.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < GO_KMAX; j++) { const bool in = a + j < n; vc[j] = in ? chr[a + j] : 0; vs[j] = in ? start[a + j] : 0; ve[j] = in ? stop[a + j] : 0; vg[j] = in ? gc[a + j] : 0; vx[j] = in ? __float_as_uint(count[a + j]) : 0u; }
    }
    uint32_t keepMask = 0;
#pragma unroll
    for (int j = 0; j < GO_KMAX; j++) if (a + j < n && sKeep[vg[j]]) keepMask |= 1u << j;
    uint32_t totalW; const uint32_t ex = cf_excl_scan256((uint32_t)__popc(keepMask), sh4, &totalW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // every load of the range has landed in registers
    __syncthreads();
    if (threadIdx.x == 0) cf_st(&pub[(size_t)blockIdx.x * GO_PUB], totalW);
    // the ranges in front: wait for their counts
    unsigned long long before = 0;
    for (int v = threadIdx.x; v < (int)blockIdx.x; v += 256) {
        uint32_t c;
        do { c = __hip_atomic_load(&pub[(size_t)v * GO_PUB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (c == GO_SENT) __builtin_amdgcn_s_sleep(2); } while (c == GO_SENT);
        before += c;
    }
    before = cf_block_sum_u64(before, sh16);
    long long o = (long long)before + ex;
#pragma unroll
    for (int j = 0; j < GO_KMAX; j++) {
        if ((keepMask >> j) & 1u) {
            const double m = sMed[vg[j]];
            float x = __uint_as_float(vx[j]);
            if (m > 0) x = (float)(gm * (double)x / m);                        // CanvasClean.cs:190-194
            chr[o] = vc[j]; start[o] = vs[j]; stop[o] = ve[j]; gc[o] = vg[j]; count[o] = x;
            o++;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { hostDec->bad = 0u; hostDec->noop = 0u; hostDec->nOut = before + totalW; hostDec->globalMedian = gm; cvx_mail_publish(hostSeq, seq); }
}

// Returns handled = false when the stage has to go through the general chain (nothing was modified).
static int32_t clean_gc_only(canvas_ctx* ctx, int64_t n, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, float* d_count, int32_t* d_gc, int32_t nchr, const uint8_t* h_chr_is_autosome,
                             int64_t* h_n_out, int32_t* h_info, bool* handled) {
    *handled = false;
    if (nchr > 256 || n <= 0) return CANVAS_OK;
    static const unsigned gridA = [&] { int per = 0, cus = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)k_go_apply, 256, 0) != hipSuccess || per <= 0) per = 2;
                                        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = 256; return (unsigned)(per * cus); }();
    if ((unsigned long long)n > (unsigned long long)gridA * 256ull * GO_KMAX) return CANVAS_OK;      // more bins than the resident workgroups of the apply kernel hold in registers: the general chain
    if (((uintptr_t)d_chr | (uintptr_t)d_start | (uintptr_t)d_stop | (uintptr_t)d_gc | (uintptr_t)d_count) & 15) return CANVAS_OK;      // (16-byte loads)
    // persistent state of the path: counters (zero between calls), the genome row, decisions, tickets, the hand-off slots
    const size_t bytes = ((size_t)GO_REP * NGC * GO_VMAX + GO_VMAX) * 4 + sizeof(GoDec) + 256 + (size_t)gridA * GO_PUB * 4 + 256;
    if (!ctx->go_buf) {
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->go_buf, bytes));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(ctx->go_buf, 0, bytes, ctx->stream));
    }
    uint32_t* cnt = (uint32_t*)ctx->go_buf; uint32_t* all = cnt + (size_t)GO_REP * NGC * GO_VMAX; GoDec* dec = (GoDec*)(all + GO_VMAX);
    uint32_t* tick = (uint32_t*)((char*)dec + ((sizeof(GoDec) + 63) & ~size_t(63))); uint32_t* tick3 = tick + 16; unsigned long long* nCounted = (unsigned long long*)(tick + 32); unsigned long long* nFound = nCounted + 2;
    uint32_t* pub = tick + 64;
    int32_t rc = canvas_pin_reserve(ctx, sizeof(GoDec) + 64); if (rc) return rc;
    GoDec* hostDec = (GoDec*)ctx->pin; unsigned* hostSeq = (unsigned*)((char*)ctx->pin + ((sizeof(GoDec) + 15) & ~size_t(15)));
    const unsigned goSeq = cvx_mail_arm(ctx, hostSeq);       // the decisions' mailbox stamp (common.hpp)
    GoIsAuto ia; memset(&ia, 0, sizeof ia); memcpy(ia.v, h_chr_is_autosome, (size_t)nchr);
    // the grid of the apply kernel: every workgroup resident, ranges of whole 256-bin rounds
    const unsigned gA = (unsigned)((n + 256 * GO_KMAX - 1) / (256 * GO_KMAX));
    ProfScope ps(ctx, "clean_total");
    hipLaunchKernelGGL(k_go_count, dim3((unsigned)std::min<long long>(2048, (n + 1023) / 1024)), dim3(256), 0, ctx->stream, d_chr, d_gc, d_count, (long long)n, ia, (int)nchr, cnt, dec, nCounted);
    hipLaunchKernelGGL(k_go_decide, dim3(NGC), dim3(256), 0, ctx->stream, cnt, all, dec, tick, pub, (int)gA, 100, tick3, nCounted, nFound);
    hipLaunchKernelGGL(k_go_apply, dim3(gA), dim3(256), 0, ctx->stream, d_chr, d_start, d_stop, d_gc, d_count, (long long)n, dec, pub, tick3, hostDec, hostSeq, goSeq);
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    CANVAS_HIP_TRY(ctx, hipGetLastError());
    { int32_t rcm = cvx_mail_await(ctx, hostSeq, goSeq, "CanvasClean -g: decisions"); if (rcm) return rcm; }
    if (hostDec->bad) return CANVAS_OK;                                        // nothing was modified
    *handled = true;
    *h_n_out = (int64_t)hostDec->nOut;
    if (h_info) { int32_t info[8] = {0}; info[0] = (int32_t)n; info[1] = (int32_t)n; info[2] = (int32_t)hostDec->nOut; info[3] = (int32_t)hostDec->nOut; info[4] = 0; info[5] = 1; info[6] = 1 /* the three-launch path */; memcpy(h_info, info, sizeof info); }
    return CANVAS_OK;
}
