// One sample, chromosomes sharded over the ranks of a node (north_star / SURVEY 8e; BASELINE configs[3], [4]): one process per GPU, rank r holds the per-base
// arrays of an LPT group of chromosomes only.  The reference runs CanvasBin and CanvasPartition as independent per-chromosome tasks (CanvasBin.cs:513-539,
// HiddenMarkovModelsRunner.cs:51-104); the genome-wide couplings are few and small, and each becomes ONE all-gather (RCCL over xGMI, cvx_allgather):
//
//   local   sweep of the owned chromosomes (k_tile_summary)                                   -> (#hit > 0, popcount(mask), positions before pos0) per chromosome
//   GATHER  the 3 x nchr rate table (576 B per rank)                                           -> same bin size on every rank (CanvasBin.cs:73-83), #bins of every chromosome
//   local   bins of the owned chromosomes closed from the summaries
//   GATHER  the owned bins, 16 B/bin packed to the largest rank's count (~2.4 MB per rank at 8 x) -> whole-genome SoA in file order on every rank
//   every   CanvasClean on the whole-genome SoA: redundant and deterministic (its order statistics are genome-wide; exchanging 16 radix histograms
//           per select would cost more than recomputing 1 ms), F2 hand-off, chromosome offsets
//   local   PerSampleHMM of the owned chromosomes, emission parameters from the genome-wide quartiles (HiddenMarkovModelsRunner.cs:36-50)
//   GATHER  the segment boundary records [n, (chr, startBin, endBin, state) ...] (KBs)         -> canvas_allgather_boundaries: THE collective north_star names
//   every   state per bin from the records, PostProcessSegments: running segment id in file order (SegmentationResultsProcessor.cs:57-62, Q17)
//
// Every rank ends with the same cleaned bins, states and segment ids, bit-identical to the single-rank canvas_sample_pipeline (tests/test_sharded_gpu.py).
#include "common.hpp"
#include <algorithm>
#include <vector>
#include <chrono>
#include <cstdio>

#define SH_BLK 2048

__global__ void __launch_bounds__(256) k_sh_unshard(const int32_t* __restrict__ recv, int64_t slotInts /* ints per rank: 4 columns of maxB + the status word */, int64_t maxB, const long long* __restrict__ binOff, const int32_t* __restrict__ owner,
                                                    const long long* __restrict__ rankOff, int nchr, int64_t total, int32_t* __restrict__ oChr, int32_t* __restrict__ oStart,
                                                    int32_t* __restrict__ oStop, int32_t* __restrict__ oGc, float* __restrict__ oCount) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = nchr - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (binOff[mid] <= g) lo = mid; else hi = mid - 1; }
    const int32_t* base = recv + (size_t)owner[lo] * (size_t)slotInts;
    const int64_t j = rankOff[lo] + (g - binOff[lo]);
    oChr[g] = lo; oStart[g] = base[j]; oStop[g] = base[maxB + j]; oGc[g] = base[2 * maxB + j]; oCount[g] = __int_as_float(base[3 * maxB + j]);
}
// a record starts at the first bin of a chromosome and wherever the state changes
__global__ void __launch_bounds__(256) k_sh_flags(const int32_t* __restrict__ state, const long long* __restrict__ loff, int nl, int64_t n, uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = nl - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (loff[mid] <= i) lo = mid; else hi = mid - 1; }
    flags[i] = (i == loff[lo] || state[i] != state[i - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_sh_count(const uint8_t* __restrict__ flags, int64_t n, uint32_t* __restrict__ blockCnt) {
    __shared__ uint32_t sh[4];
    const int64_t base = (int64_t)blockIdx.x * SH_BLK;
    uint32_t c = 0;
    for (int j = 0; j < SH_BLK / 256; j++) { const int64_t i = base + j * 256 + threadIdx.x; if (i < n) c += flags[i]; }
    c = wave_reduce_add_u32(c);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockCnt[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void __launch_bounds__(64) k_sh_scan(uint32_t* __restrict__ blockCnt, int nblocks, unsigned int* __restrict__ total) {
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 64) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? blockCnt[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(v);
        if (i < nblocks) blockCnt[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) *total = carry;
}
// records (chrGlobal, startBin, endBin, state); endBin is filled by k_sh_ends
__global__ void __launch_bounds__(256) k_sh_scatter(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blockOff, const int32_t* __restrict__ state, const long long* __restrict__ loff,
                                                    const int32_t* __restrict__ localToGlobal, int nl, int64_t n, int32_t cap, int32_t* __restrict__ rec) {
    __shared__ uint32_t sh[4];
    const int64_t base = (int64_t)blockIdx.x * SH_BLK;
    uint32_t running = blockOff[blockIdx.x];
    for (int j = 0; j < SH_BLK / 256; j++) {
        const int64_t i = base + j * 256 + threadIdx.x;
        const uint32_t f = i < n ? flags[i] : 0;
        const uint32_t inc = wave_inclusive_scan_u32(f);
        if (lane_id() == 63) sh[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int k = 0; k < 4; k++) { if (k < (int)(threadIdx.x >> 6)) woff += sh[k]; tot += sh[k]; }
        if (f) {
            const uint32_t d = running + woff + inc - 1;
            if ((int32_t)d < cap) {
                int lo = 0, hi = nl - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (loff[mid] <= i) lo = mid; else hi = mid - 1; }
                rec[4 * d] = localToGlobal[lo]; rec[4 * d + 1] = (int32_t)(i - loff[lo]); rec[4 * d + 2] = (int32_t)(loff[lo + 1] - loff[lo]) - 1; rec[4 * d + 3] = state[i];
            }
        }
        running += tot;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_sh_ends(int32_t* __restrict__ rec, const unsigned int* __restrict__ nrec, int32_t cap) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n = min((int)*nrec, cap);
    if (k + 1 < n && rec[4 * (k + 1)] == rec[4 * k]) rec[4 * k + 2] = rec[4 * (k + 1) + 1] - 1;     // otherwise: the chromosome's last bin, written by k_sh_scatter
}
// state of every bin of the genome from the gathered records of the rank that owns its chromosome
__global__ void __launch_bounds__(256) k_sh_fill_state(const int32_t* __restrict__ all, int recStride, const int32_t* __restrict__ owner, const long long* __restrict__ chrOff, int nchr,
                                                       int64_t N, int32_t* __restrict__ state, int* __restrict__ bad) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    int c = 0, hi = nchr - 1;
    while (c < hi) { const int mid = (c + hi + 1) >> 1; if (chrOff[mid] <= g) c = mid; else hi = mid - 1; }
    const int32_t* blk = all + (size_t)owner[c] * recStride;
    const int n = blk[0] / 4; const int32_t* rec = blk + 1;
    const int32_t b = (int32_t)(g - chrOff[c]);
    int lo = 0, hj = n - 1;                                                // last record with (chr, start) <= (c, b)
    while (lo < hj) { const int mid = (lo + hj + 1) >> 1; const int32_t rc = rec[4 * mid], rs = rec[4 * mid + 1]; if (rc < c || (rc == c && rs <= b)) lo = mid; else hj = mid - 1; }
    if (n <= 0 || rec[4 * lo] != c || rec[4 * lo + 1] > b || rec[4 * lo + 2] < b) { *bad = 1; state[g] = -1; return; }
    state[g] = rec[4 * lo + 3];
}

namespace {
struct ShardHook {
    canvas_ctx* ctx; int nchr; const int32_t* owner; const uint8_t* isAuto; int countsPerBin; int binSizeIn; const int* localToGlobal;
    long long* dBuf;                       // device: [1 + nranks][nchr * 3 + 1]
    std::vector<long long> obs, pop, popBefore;      // every chromosome, after the exchange
    int gcwDone = 0;                       // reductions of the GCContentWeighted pre-pass that have taken place on this rank (2 per call: fragment means, read-GC profile)
    bool exchanged = false;                // the rate exchange has taken place on this rank (a rank that fails before it still has to take part: see canvas_sample_pipeline_sharded)
    int failedRank = -1; long long failedCode = 0;   // a peer announced a failure in its status slot
};
// every rank's slice ends with a status word: 0, or the error code of a rank that failed before the exchange and only takes part so that nobody waits for it
int32_t shard_rates_exchange(ShardHook& H, int nl, const long long* obs, const long long* pop, const long long* popBefore, int32_t status = 0) {
    canvas_ctx* ctx = H.ctx;
    const int W = ctx->nranks, n3 = H.nchr * 3, slice = n3 + 1;
    std::vector<long long> mine(slice, 0), all((size_t)W * slice, 0);
    for (int i = 0; i < nl; i++) { const int c = H.localToGlobal[i]; mine[3 * c] = obs[i]; mine[3 * c + 1] = pop[i]; mine[3 * c + 2] = popBefore[i]; }
    mine[n3] = status;
    H.exchanged = true;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(H.dBuf, mine.data(), (size_t)slice * 8, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = cvx_allgather(ctx, H.dBuf, H.dBuf + slice, (size_t)slice * 8); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(all.data(), H.dBuf + slice, (size_t)W * slice * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    H.obs.assign(H.nchr, 0); H.pop.assign(H.nchr, 0); H.popBefore.assign(H.nchr, 0);
    for (int c = 0; c < H.nchr; c++) { const long long* e = &all[(size_t)H.owner[c] * slice + 3 * c]; H.obs[c] = e[0]; H.pop[c] = e[1]; H.popBefore[c] = e[2]; }
    for (int r = 0; r < W; r++) if (all[(size_t)r * slice + n3] != 0 && H.failedRank < 0) { H.failedRank = r; H.failedCode = all[(size_t)r * slice + n3]; }
    return CANVAS_OK;
}
// element-wise sum of n counters over the ranks (+ the status word, as above): the genome-wide statistics of the GCContentWeighted pre-pass (CanvasBin.cs:164-174, 372-391)
int32_t shard_reduce(ShardHook& H, unsigned long long* v, int n, int32_t status = 0) {
    canvas_ctx* ctx = H.ctx;
    const int W = ctx->nranks, slice = n + 1;
    std::vector<long long> mine(slice, 0), all((size_t)W * slice, 0);
    for (int i = 0; i < n; i++) mine[i] = v ? (long long)v[i] : 0;
    mine[n] = status;
    H.gcwDone++;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(H.dBuf, mine.data(), (size_t)slice * 8, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = cvx_allgather(ctx, H.dBuf, H.dBuf + slice, (size_t)slice * 8); if (rc) return rc;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(all.data(), H.dBuf + slice, (size_t)W * slice * 8, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; v && i < n; i++) { unsigned long long t = 0; for (int r = 0; r < W; r++) t += (unsigned long long)all[(size_t)r * slice + i]; v[i] = t; }
    for (int r = 0; r < W; r++) if (all[(size_t)r * slice + n] != 0 && H.failedRank < 0) { H.failedRank = r; H.failedCode = all[(size_t)r * slice + n]; }
    return CANVAS_OK;
}
int32_t shard_gcw_reduce_hook(void* user, unsigned long long* v, int n) {
    ShardHook& H = *(ShardHook*)user;
    int32_t rc = shard_reduce(H, v, n); if (rc) return rc;
    if (H.failedRank >= 0) CANVAS_FAIL(H.ctx, CANVAS_ERR_COMM, "canvas_bin_sample_sharded: rank " + std::to_string(H.failedRank) + " failed in the GCContentWeighted pre-pass (code " + std::to_string(H.failedCode) + ")");
    return CANVAS_OK;
}
int32_t shard_bin_size_hook(void* user, int nl, const long long* obs, const long long* pop, const long long* popBefore, int32_t* binSizeOut) {
    ShardHook& H = *(ShardHook*)user;
    int32_t rc = shard_rates_exchange(H, nl, obs, pop, popBefore); if (rc) return rc;
    if (H.failedRank >= 0) CANVAS_FAIL(H.ctx, CANVAS_ERR_COMM, "canvas_sample_pipeline_sharded: rank " + std::to_string(H.failedRank) + " failed before the rate exchange (code " + std::to_string(H.failedCode) + ")");
    if (H.binSizeIn > 0) { *binSizeOut = H.binSizeIn; return CANVAS_OK; }
    std::vector<double> rates;                                              // SampleHitArrays.GetRates / GetBinSize over the autosomes (CanvasBin.cs:30-83)
    for (int c = 0; c < H.nchr; c++) if (H.isAuto[c]) rates.push_back((int)H.obs[c] / (double)(int)H.pop[c]);
    if (rates.empty()) CANVAS_FAIL(H.ctx, CANVAS_ERR_INVALID, "no autosome to derive the bin size from");
    *binSizeOut = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), H.countsPerBin);
    return CANVAS_OK;
}
}  // namespace

// h_pos0 != NULL: d_bases / d_hits are the packed reference / hit planes of the owned chromosomes (canvas_bin_sample_packed), d_mask is not read
static int32_t pipeline_sharded_impl(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                     const uint8_t* const* d_hits, const int64_t* h_pos0, const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y,
                                     int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                     int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                     double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                     int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments,
                                     const int16_t* const* d_fraglen = nullptr, bool binsOnly = false /* canvas_bin_sample_sharded: return with the whole-genome bins */) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (nchr <= 0 || !h_chr_owner || !d_bases || !d_mask || !d_hits || !h_len || !h_chr_is_autosome || !d_chr || !d_start || !d_stop || !d_gc || !d_count ||
        (!binsOnly && (!d_cov || !d_state || !d_segment_id || !h_chr_offset)) || (bin_size_in <= 0 && counts_per_bin <= 0))
        CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline_sharded: bad arguments");
    const bool gcw = mode == CANVAS_MODE_GC_CONTENT_WEIGHTED;
    if (gcw && (!d_fraglen || h_pos0)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_bin_sample_sharded: GCContentWeighted needs the fragment lengths and the per-base arrays");
    if (mode != CANVAS_MODE_BINARY && mode != CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE && !gcw) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_sample_pipeline_sharded: modes 0, 3 and 5");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int W = ctx->nranks, me = ctx->rank;
    for (int c = 0; c < nchr; c++) if (h_chr_owner[c] < 0 || h_chr_owner[c] >= W || h_len[c] <= 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline_sharded: owner outside [0, nranks)");
    std::vector<int> mine;                                         // owned chromosomes, file order
    for (int c = 0; c < nchr; c++) if (h_chr_owner[c] == me) mine.push_back(c);
    const int nl = (int)mine.size();
    // upper bounds that every rank can compute: bins per rank for the smallest possible bin size
    const int64_t minBin = bin_size_in > 0 ? bin_size_in : std::max(1, counts_per_bin);
    std::vector<int64_t> capRank(W, 16);
    for (int c = 0; c < nchr; c++) capRank[h_chr_owner[c]] += h_len[c] / minBin + 1;
    const int64_t capLocal = capRank[me], capMax = *std::max_element(capRank.begin(), capRank.end());
    const int32_t maxRecInts = 4 * 16384;                            // boundary records per rank in the first attempt (grown if a rank has more)
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const int nbLocal = (int)((capLocal + SH_BLK - 1) / SH_BLK) + 1;
    size_t need = 0;
    const size_t oBins = need; need += 5 * al((size_t)capLocal * 4);
    const size_t oSend = need; need += al((size_t)capMax * 16 + 16);
    const size_t oRecv = need; need += al((size_t)W * ((size_t)capMax * 16 + 16));
    const size_t oRates = need; need += al((size_t)(W + 1) * (size_t)std::max(nchr * 3 + 1, 203) * 8);      // (the rate table; the 202 counters of the read-GC profile)
    const size_t oTab = need; need += 6 * al((size_t)(nchr + 2) * 8);
    const size_t oCovL = need; need += al((size_t)capLocal * 8);
    const size_t oStateL = need; need += al((size_t)capLocal * 4);
    const size_t oFlags = need; need += al((size_t)capLocal);
    const size_t oBlk = need; need += al((size_t)nbLocal * 4);
    const size_t oCnt = need; need += 256;
    if (need > ctx->shard_ws_bytes) {
        if (ctx->shard_ws) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->shard_ws)); ctx->shard_ws = nullptr; ctx->shard_ws_bytes = 0; }
        CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->shard_ws, need)); ctx->shard_ws_bytes = need;
    }
    char* S = (char*)ctx->shard_ws;
    int32_t* lChr = (int32_t*)(S + oBins); int32_t* lStart = (int32_t*)(S + oBins + al((size_t)capLocal * 4)); int32_t* lStop = (int32_t*)(S + oBins + 2 * al((size_t)capLocal * 4));
    int32_t* lGc = (int32_t*)(S + oBins + 3 * al((size_t)capLocal * 4)); float* lCount = (float*)(S + oBins + 4 * al((size_t)capLocal * 4));
    int32_t* dSend = (int32_t*)(S + oSend); int32_t* dRecv = (int32_t*)(S + oRecv);
    long long* dRates = (long long*)(S + oRates);
    const size_t tabStride = al((size_t)(nchr + 2) * 8);
    long long* dBinOff = (long long*)(S + oTab); int32_t* dOwner = (int32_t*)(S + oTab + tabStride); long long* dRankOff = (long long*)(S + oTab + 2 * tabStride);
    long long* dChrOff = (long long*)(S + oTab + 3 * tabStride); long long* dLoff = (long long*)(S + oTab + 4 * tabStride); int32_t* dL2G = (int32_t*)(S + oTab + 5 * tabStride);
    double* dCovL = (double*)(S + oCovL); int32_t* dStateL = (int32_t*)(S + oStateL); uint8_t* dFlags = (uint8_t*)(S + oFlags); uint32_t* dBlk = (uint32_t*)(S + oBlk);
    unsigned int* dNrec = (unsigned int*)(S + oCnt); int* dBad = (int*)(S + oCnt + 64);

    // A failure on ONE rank must not leave the others waiting in a collective (ncclAllGather / the host callback would block for ever): every exchange carries a status
    // word per rank — an extra slot of the rate table, 16 bytes behind the bin columns, a negative count in the boundary gather — a rank that has failed keeps taking part
    // in the remaining exchanges with an empty payload, and every rank returns an error once it has seen a non-zero status (its own error for the rank that failed,
    // CANVAS_ERR_COMM naming that rank for the others).
    int32_t localErr = CANVAS_OK; std::string localMsg;
    auto fail_local = [&](int32_t code) { if (!localErr) { localErr = code ? code : CANVAS_ERR_HIP; localMsg = ctx->err; } };
    auto peer_failed = [&](int r, long long code, const char* where) -> int32_t {
        if (localErr) { ctx->err = localMsg; return localErr; }
        CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string("canvas_sample_pipeline_sharded: rank ") + std::to_string(r) + " failed " + where + " (code " + std::to_string(code) + ")");
    };
    static const bool shTiming = cvx_hook("CANVAS_PIPELINE_TIMING") != nullptr;
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    auto mark = [&](const char* what) { if (shTiming) { (void)hipStreamSynchronize(ctx->stream); marks.push_back({what, std::chrono::steady_clock::now()}); } };
    mark("start");
    // ---- 1. local sweep, exchange of the rate table, one bin size, local bins
    ShardHook H{ctx, nchr, h_chr_owner, h_chr_is_autosome, counts_per_bin, bin_size_in, mine.data(), dRates, {}, {}, {}};
    int32_t binSize = 0; int64_t nbMine = 0;
    int32_t rc;
    if (nl > 0) {
        std::vector<const uint8_t*> lb(nl), lh(nl); std::vector<const uint64_t*> lm(nl); std::vector<int64_t> ll(nl), perChr(nl), lp0(nl, 0);
        bool haveArrays = true;
        for (int i = 0; i < nl; i++) { lb[i] = d_bases[mine[i]]; lm[i] = h_pos0 ? (const uint64_t*)d_bases[mine[i]] : d_mask[mine[i]]; lh[i] = d_hits[mine[i]]; ll[i] = h_len[mine[i]]; if (h_pos0) lp0[i] = h_pos0[mine[i]];
                                       if (!lb[i] || !lm[i] || !lh[i]) haveArrays = false; }
        if (!haveArrays) { ctx->err = "canvas_sample_pipeline_sharded: an owned chromosome has no arrays"; fail_local(CANVAS_ERR_INVALID); }
        else {
            std::vector<const int16_t*> lf(nl, nullptr);
            if (gcw) for (int i = 0; i < nl; i++) { lf[i] = d_fraglen[mine[i]]; if (!lf[i]) haveArrays = false; }
            if (!haveArrays) { ctx->err = "canvas_bin_sample_sharded: an owned chromosome has no fragment lengths"; fail_local(CANVAS_ERR_INVALID); }
            else {
                if (gcw) { ctx->gcw_reduce = shard_gcw_reduce_hook; ctx->gcw_reduce_user = &H; }
                rc = cvx_bin_sample_hooked(ctx, nl, lb.data(), lm.data(), lh.data(), ll.data(), mode, shard_bin_size_hook, &H, lChr, lStart, lStop, lGc, lCount, capLocal, perChr.data(), &nbMine, h_pos0 ? lp0.data() : nullptr,
                                           gcw ? lf.data() : nullptr);
                ctx->gcw_reduce = nullptr; ctx->gcw_reduce_user = nullptr;
                if (rc) fail_local(rc);
            }
        }
        // failed before a hook ran: the exchanges it would have made still take place, in their order (two reductions of the GCContentWeighted pre-pass, then the rate table)
        if (gcw) { static const int nRed[2] = {2, 202}; while (H.gcwDone < 2) { rc = shard_reduce(H, nullptr, nRed[H.gcwDone], localErr ? localErr : CANVAS_ERR_COMM); if (rc) return rc; } }
        if (localErr && !H.exchanged) { rc = shard_rates_exchange(H, 0, nullptr, nullptr, nullptr, localErr); if (rc) return rc; }
    } else {                                                         // more ranks than chromosomes: this rank only takes part in the exchanges
        if (gcw) { rc = shard_reduce(H, nullptr, 2); if (rc) return rc; rc = shard_reduce(H, nullptr, 202); if (rc) return rc; }
        rc = shard_rates_exchange(H, 0, nullptr, nullptr, nullptr); if (rc) return rc;
    }
    if (H.failedRank >= 0) return peer_failed(H.failedRank, H.failedCode, "before the rate exchange");
    if (!localErr) {   // the bin size of the hook, recomputed here so that a rank without chromosomes has it too
        if (bin_size_in > 0) binSize = bin_size_in;
        else { std::vector<double> rates; for (int c = 0; c < nchr; c++) if (h_chr_is_autosome[c]) rates.push_back((int)H.obs[c] / (double)(int)H.pop[c]);
               if (rates.empty()) { ctx->err = "no autosome to derive the bin size from"; fail_local(CANVAS_ERR_INVALID); }
               else binSize = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), counts_per_bin); }
        if (!localErr && binSize <= 0) { ctx->err = "derived bin size is not positive"; fail_local(CANVAS_ERR_INVALID); }
    }
    // (the rate table is the same everywhere, so the two failures above happen on every rank or on none; a rank whose sweep failed AFTER the exchange goes on to the
    //  next one with its status set)
    if (localErr && binSize <= 0) {
        // no bin size: nothing below can be sized.  Every rank has the same table, so every healthy rank computed a size; this rank cannot know maxB — it derives it from the
        // table with the size the others use only if that size can be recomputed; otherwise the failure is global (same table, same result) and everybody returns here
        if (bin_size_in > 0) binSize = bin_size_in;
        else { std::vector<double> rates; for (int c = 0; c < nchr; c++) if (h_chr_is_autosome[c]) rates.push_back((int)H.obs[c] / (double)(int)H.pop[c]);      // (the SAME expression as the healthy ranks': a different filter here would size the next exchange differently on this rank)
               if (!rates.empty()) binSize = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), counts_per_bin); }
        if (binSize <= 0) { ctx->err = localMsg; return localErr; }
    }
    if (h_bin_size) *h_bin_size = binSize;
    mark("sweep + rate exchange + local bins");
    // ---- 2. bins of every chromosome: counts are known everywhere, the columns travel in ONE all-gather
    std::vector<long long> binOff(nchr + 1, 0), rankOff(nchr, 0), nbRank(W, 0);
    for (int c = 0; c < nchr; c++) { const long long nb = (H.pop[c] - H.popBefore[c]) / binSize; binOff[c + 1] = binOff[c] + nb; rankOff[c] = nbRank[h_chr_owner[c]]; nbRank[h_chr_owner[c]] += nb; }
    const int64_t total = binOff[nchr];
    if (!localErr && nbRank[me] != nbMine) { ctx->err = "canvas_sample_pipeline_sharded: local bin count disagrees with the exchanged table"; fail_local(CANVAS_ERR_COMM); }
    if (h_nbins) *h_nbins = total;
    if (!localErr && total > cap) { ctx->err = "canvas_sample_pipeline_sharded: output capacity too small"; fail_local(CANVAS_ERR_CAPACITY); }
    const int64_t maxB = std::max<long long>(1, *std::max_element(nbRank.begin(), nbRank.end()));
    if ((size_t)maxB > (size_t)capMax) { ctx->err = "canvas_sample_pipeline_sharded: bin table exceeds the exchange buffers"; return localErr ? localErr : CANVAS_ERR_CAPACITY; }   // (global: same table everywhere)
    if (!localErr && nbMine > 0) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dSend, lStart, (size_t)nbMine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dSend + maxB, lStop, (size_t)nbMine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dSend + 2 * maxB, lGc, (size_t)nbMine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dSend + 3 * maxB, lCount, (size_t)nbMine * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    const size_t slotB = (size_t)maxB * 16 + 16;                     // the four columns + the status word of the rank
    {
        const int32_t st4[4] = {localErr, 0, 0, 0};
        rc = canvas_h2d_small(ctx, dSend + 4 * maxB, st4, sizeof st4); if (rc) return rc;
    }
    rc = cvx_allgather(ctx, dSend, dRecv, slotB); if (rc) return rc;
    {
        std::vector<int32_t> st((size_t)W, 0);
        CANVAS_HIP_TRY(ctx, hipMemcpy2DAsync(st.data(), 4, (const char*)dRecv + (size_t)maxB * 16, slotB, 4, (size_t)W, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int r = 0; r < W; r++) if (st[(size_t)r] != 0) return peer_failed(r, st[(size_t)r], "before the bin exchange");
    }
    if (total == 0) { if (h_nbins_clean) *h_nbins_clean = 0; if (h_nsegments) *h_nsegments = 0; if (h_chr_offset) for (int c = 0; c <= nchr; c++) h_chr_offset[c] = 0; if (h_local_sd) *h_local_sd = -1.0; return CANVAS_OK; }
    {
        rc = canvas_h2d_small(ctx, dBinOff, binOff.data(), (size_t)(nchr + 1) * 8); if (rc) fail_local(rc);
        if (!localErr) { rc = canvas_h2d_small(ctx, dOwner, h_chr_owner, (size_t)nchr * 4); if (rc) fail_local(rc); }
        if (!localErr) { rc = canvas_h2d_small(ctx, dRankOff, rankOff.data(), (size_t)nchr * 8); if (rc) fail_local(rc); }
    }
    if (!localErr) hipLaunchKernelGGL(k_sh_unshard, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dRecv, (int64_t)(slotB / 4), maxB, dBinOff, dOwner, dRankOff, nchr, total, d_chr, d_start, d_stop, d_gc, d_count);
    mark("bin all-gather + unshard");
    if (binsOnly) {
        if (localErr) { ctx->err = localMsg; return localErr; }
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CANVAS_OK;
    }
    // ---- 3. CanvasClean on the whole genome (every rank, deterministic), F2 hand-off, chromosome offsets of the cleaned bins
    std::vector<uint8_t> noY((size_t)nchr, 0);
    double lsd = -1.0; int64_t nClean = 0; int32_t info[8];
    if (!localErr) { rc = canvas_clean2(ctx, total, d_chr, d_start, d_stop, d_count, d_gc, nchr, h_chr_is_autosome, h_chr_is_y ? h_chr_is_y : noY.data(), clean_flags, min_bins_per_gc, &lsd, &nClean, info); if (rc) fail_local(rc); }
    if (h_nbins_clean) *h_nbins_clean = nClean;
    if (h_local_sd) *h_local_sd = lsd;
    const void* hCovQ = nullptr;                                   // the genome-wide quartile counters come out of the quantisation sweep and arrive with the offsets' synchronisation
    if (!localErr) { rc = cvx_quantize_f2_covq(ctx, d_count, nClean, d_cov, &hCovQ); if (rc) fail_local(rc); }
    if (!localErr) { rc = canvas_chromosome_offsets(ctx, d_chr, nClean, nchr, h_chr_offset); if (rc) fail_local(rc); }
    mark("clean + f2 + offsets");
    // ---- 4. PerSampleHMM of the owned chromosomes (compact copy of their coverage; quartiles over the whole sample)
    std::vector<int64_t> loff(nl + 1, 0);
    if (!localErr) for (int i = 0; i < nl; i++) loff[i + 1] = loff[i] + (h_chr_offset[mine[i] + 1] - h_chr_offset[mine[i]]);
    const int64_t nLocal = localErr ? 0 : loff[nl];
    int32_t maxPer = maxRecInts, nrecInts = 0;
    if (nLocal > 0) {
        for (int i = 0; i < nl && !localErr; i++) { const int64_t b0 = h_chr_offset[mine[i]], T = loff[i + 1] - loff[i];
            if (T > 0 && hipMemcpyAsync(dCovL + loff[i], d_cov + b0, (size_t)T * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { ctx->err = "canvas_sample_pipeline_sharded: copy of the owned coverage failed"; fail_local(CANVAS_ERR_HIP); } }
        if (!localErr) { rc = cvx_hmm_per_sample_subset(ctx, nl, dCovL, loff.data(), d_cov, nClean, dStateL, hCovQ); if (rc) fail_local(rc); }
        // boundary records of the owned chromosomes
        std::vector<long long> loff64(loff.begin(), loff.end()); std::vector<int32_t> l2g(mine.begin(), mine.end());
        if (!localErr) { rc = canvas_h2d_small(ctx, dLoff, loff64.data(), (size_t)(nl + 1) * 8); if (rc) fail_local(rc); }
        if (!localErr) { rc = canvas_h2d_small(ctx, dL2G, l2g.data(), (size_t)nl * 4); if (rc) fail_local(rc); }
        if (!localErr) {
            const int nb = (int)((nLocal + SH_BLK - 1) / SH_BLK);
            hipLaunchKernelGGL(k_sh_flags, dim3((unsigned)((nLocal + 255) / 256)), dim3(256), 0, ctx->stream, dStateL, dLoff, nl, nLocal, dFlags);
            hipLaunchKernelGGL(k_sh_count, dim3(nb), dim3(256), 0, ctx->stream, dFlags, nLocal, dBlk);
            hipLaunchKernelGGL(k_sh_scan, dim3(1), dim3(64), 0, ctx->stream, dBlk, nb, dNrec);
        }
    }
    mark("hmm of the owned chromosomes + record flags");
    // the records are built in the workspace of the context (nothing else runs between here and the gather); its size follows maxPer
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t recBytes = al((size_t)maxPer * 4) + al((size_t)W * (1 + (size_t)maxPer) * 4) + 4096;
        rc = canvas_ws_reserve(ctx, recBytes + (size_t)(1 + maxPer) * 4 + 4096); if (rc) return rc;      // (the same allocation on every rank: a failure here is an out-of-memory device, nothing to salvage)
        // canvas_allgather_boundaries packs into the FRONT of the workspace: records and the gathered table sit behind that area
        char* wsb = (char*)ctx->ws + al((size_t)(1 + maxPer) * 4 + 256);
        int32_t* dRec = (int32_t*)wsb; int32_t* dAll = (int32_t*)(wsb + al((size_t)maxPer * 4));
        unsigned int nrec = 0;
        if (nLocal > 0 && !localErr) {
            const int nb = (int)((nLocal + SH_BLK - 1) / SH_BLK);
            hipLaunchKernelGGL(k_sh_scatter, dim3(nb), dim3(256), 0, ctx->stream, dFlags, dBlk, dStateL, dLoff, dL2G, nl, nLocal, maxPer / 4, dRec);
            hipLaunchKernelGGL(k_sh_ends, dim3((unsigned)((maxPer / 4 + 255) / 256)), dim3(256), 0, ctx->stream, dRec, dNrec, maxPer / 4);
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&nrec, dNrec, 4, hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        nrecInts = (int32_t)std::min<long long>((long long)nrec * 4, maxPer);      // a rank with more records than fit announces the full count below
        std::vector<int32_t> counts(W, 0);
        // ---- 5. THE collective: segment boundaries of every rank (a rank that has failed sends its negative error code as its count)
        rc = cvx_allgather_boundaries_status(ctx, dRec, localErr ? (localErr < 0 ? localErr : -localErr) : nrecInts, maxPer, dAll, counts.data()); if (rc) return rc;
        for (int r = 0; r < W; r++) if (counts[r] < 0) return peer_failed(r, counts[r], "before the boundary exchange");
        // overflow protocol without a second collective type: a rank whose records did not fit sends count = maxPer and everyone retries with the bound every rank can derive
        bool overflow = false;
        for (int r = 0; r < W; r++) if (counts[r] >= maxPer) overflow = true;
        if ((long long)nrec * 4 >= maxPer) overflow = true;
        if (overflow && attempt == 0) { maxPer = (int32_t)std::min<long long>(4ll * (capMax + 16), 0x7FFFFFF0ll); continue; }
        if (overflow) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_sample_pipeline_sharded: boundary records do not fit");
        ctx->shard_stats[0] = W; ctx->shard_stats[1] = nl; ctx->shard_stats[2] = nbMine; ctx->shard_stats[3] = (long long)slotB; ctx->shard_stats[4] = nrec; ctx->shard_stats[5] = (long long)(1 + maxPer) * 4;
        // ---- 6. state of every bin from the records, then the running segment id in file order (no further exchange: a failure from here on is local)
        std::vector<long long> chrOff64(h_chr_offset, h_chr_offset + nchr + 1);
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dChrOff, chrOff64.data(), (size_t)(nchr + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemsetAsync(dBad, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_sh_fill_state, dim3((unsigned)((nClean + 255) / 256)), dim3(256), 0, ctx->stream, dAll, 1 + maxPer, dOwner, dChrOff, nchr, nClean, d_state, dBad);
        int bad = 0;
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&bad, dBad, 4, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        CANVAS_HIP_TRY(ctx, hipGetLastError());
        if (bad) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, "canvas_sample_pipeline_sharded: the gathered boundary records do not cover every bin");
        break;
    }
    mark("boundary records + all-gather + states");
    int64_t nseg = 0;
    rc = canvas_segment_ids(ctx, nchr, h_chr_offset, d_state, d_start, d_stop, max_inter_bin_dist, d_segment_id, &nseg); if (rc) return rc;
    if (h_nsegments) *h_nsegments = nseg;
    mark("segment ids");
    if (shTiming && ctx->rank == 0) { fprintf(stderr, "sharded pipeline us:"); for (size_t i = 1; i < marks.size(); i++) fprintf(stderr, " %s %.0f |", marks[i].first, std::chrono::duration<double, std::micro>(marks[i].second - marks[i - 1].second).count()); fprintf(stderr, "\n"); }
    return CANVAS_OK;
}

extern "C" int32_t canvas_sample_pipeline_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                                  const uint8_t* const* d_hits, const int64_t* h_len, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y,
                                                  int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                                  int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                                  double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                                  int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments) {
    return pipeline_sharded_impl(ctx, nchr, h_chr_owner, d_bases, d_mask, d_hits, nullptr, h_len, h_chr_is_autosome, h_chr_is_y, counts_per_bin, bin_size_in, mode, clean_flags, min_bins_per_gc,
                                 max_inter_bin_dist, d_chr, d_start, d_stop, d_gc, d_count, cap, d_cov, d_state, d_segment_id, h_bin_size, h_nbins, h_nbins_clean, h_local_sd, h_chr_offset, h_nsegments);
}
// the same over the packed planes (0.75 B/base): d_ref / d_hit_planes / h_pos0 as for canvas_sample_pipeline_packed, entries of chromosomes this rank does not own are ignored
extern "C" int32_t canvas_sample_pipeline_sharded_packed(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint64_t* const* d_ref, const uint64_t* const* d_hit_planes,
                                                         const int64_t* h_len, const int64_t* h_pos0, const uint8_t* h_chr_is_autosome, const uint8_t* h_chr_is_y,
                                                         int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, uint32_t clean_flags, int32_t min_bins_per_gc, int32_t max_inter_bin_dist,
                                                         int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count, int64_t cap,
                                                         double* d_cov, int32_t* d_state, int32_t* d_segment_id,
                                                         int32_t* h_bin_size, int64_t* h_nbins, int64_t* h_nbins_clean, double* h_local_sd, int64_t* h_chr_offset, int64_t* h_nsegments) {
    if (ctx && !h_pos0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_sample_pipeline_sharded_packed: pos0 missing");
    return pipeline_sharded_impl(ctx, nchr, h_chr_owner, (const uint8_t* const*)d_ref, d_ref, (const uint8_t* const*)d_hit_planes, h_pos0, h_len, h_chr_is_autosome, h_chr_is_y, counts_per_bin, bin_size_in, mode,
                                 clean_flags, min_bins_per_gc, max_inter_bin_dist, d_chr, d_start, d_stop, d_gc, d_count, cap, d_cov, d_state, d_segment_id, h_bin_size, h_nbins, h_nbins_clean, h_local_sd,
                                 h_chr_offset, h_nsegments);
}

// CanvasBin alone, chromosomes sharded over the ranks: every rank ends with the bins of the whole genome (modes 0, 3 and 5; mode 5 adds two small reductions — the
// per-chromosome fragment means and the 202 counters of the read-GC profile, CanvasBin.cs:164-174, 372-391 — in front of the rate table).  BASELINE configs[4]: the tumour's bins.
extern "C" int32_t canvas_bin_sample_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const uint8_t* const* d_bases, const uint64_t* const* d_mask,
                                             const uint8_t* const* d_hits, const int16_t* const* d_fraglen, const int64_t* h_len, const uint8_t* h_chr_is_autosome,
                                             int32_t counts_per_bin, int32_t bin_size_in, int32_t mode, int32_t* d_chr, int32_t* d_start, int32_t* d_stop, int32_t* d_gc, float* d_count,
                                             int64_t cap, int32_t* h_bin_size, int64_t* h_nbins) {
    return pipeline_sharded_impl(ctx, nchr, h_chr_owner, d_bases, d_mask, d_hits, nullptr, h_len, h_chr_is_autosome, nullptr, counts_per_bin, bin_size_in, mode, 0u, 0, 0,
                                 d_chr, d_start, d_stop, d_gc, d_count, cap, nullptr, nullptr, nullptr, h_bin_size, h_nbins, nullptr, nullptr, nullptr, nullptr, d_fraglen, true);
}

extern "C" int32_t canvas_sharded_stats(canvas_ctx* ctx, int64_t* h_out6) {
    if (!ctx || !h_out6) return CANVAS_ERR_INVALID;
    for (int i = 0; i < 6; i++) h_out6[i] = ctx->shard_stats[i];
    return CANVAS_OK;
}

// ================================================================================================ CanvasPartition -m CBS / -m Wavelets, chromosomes sharded over the ranks
// Both methods work chromosome by chromosome (CBSRunner.cs:62-89 one task per chromosome, WaveletsRunner.cs:115-135 Parallel.ForEach over chromosomes); what couples the
// chromosomes is computed from the whole coverage on every rank — the seeds drawn in file order and the trimmed SD of SDUndo (CBSRunner.cs:102-112), the coverage variability
// (Segmentation.cs:297-330) — and every rank HAS the whole cleaned coverage (canvas_sample_pipeline_sharded leaves it everywhere).  So a rank segments the chromosomes it
// owns and ONE exchange of variable-length integer lists ([chromosome, count, values ...] per owned chromosome) gives every rank the whole result.  The exchange is
// canvas_allgather_boundaries' (count slot + payload, padded to the largest rank; a count that does not fit announces itself and everybody retries with the bound all ranks
// can derive); a rank that failed locally still enters it and sends its negative error code as its count, so nobody waits for it.
static int32_t exchange_lists(canvas_ctx* ctx, const std::vector<int32_t>& mine, int32_t localErr, const std::string& localMsg, int64_t hardMax, const char* what, std::vector<std::vector<int32_t>>& all) {
    const int W = ctx->nranks;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    // three sizes: 8 192 words per rank (a WGS sample's segment lists are a few hundred words, its breakpoint lists a few thousand: 32 KB per rank through the collective and
    // W x 32 KB back to the host, where round 5 moved 256 KB per rank every time), then 2^18, then the hard bound (every rank takes the same step: the counts are the same everywhere)
    const int64_t first = cvx_hook("CANVAS_SHARDED_LIST_FIRST") ? std::max(4, atoi(cvx_hook("CANVAS_SHARDED_LIST_FIRST"))) : (1 << 13);      // (test hook: a first size the lists overflow)
    const int64_t sizes[3] = {std::min<int64_t>(hardMax, first), std::min<int64_t>(hardMax, std::max<int64_t>(first * 4, cvx_hook("CANVAS_SHARDED_LIST_FIRST") ? first * 4 : (1 << 18))), hardMax};
    int64_t maxPer = sizes[0];
    for (int attempt = 0; attempt < 3; attempt++) {
        maxPer = sizes[attempt];
        const size_t front = al((size_t)(1 + maxPer) * 4 + 256);
        int32_t rc = canvas_ws_reserve(ctx, front + al((size_t)maxPer * 4) + al((size_t)W * (1 + (size_t)maxPer) * 4) + 4096); if (rc) return rc;
        char* wsb = (char*)ctx->ws + front;
        int32_t* dRec = (int32_t*)wsb; int32_t* dAll = (int32_t*)(wsb + al((size_t)maxPer * 4));
        const bool fits = (int64_t)mine.size() < maxPer;
        if (!localErr && fits && !mine.empty()) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dRec, mine.data(), mine.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        std::vector<int32_t> counts((size_t)W, 0);
        const int32_t announce = localErr ? (localErr < 0 ? localErr : -localErr) : (fits ? (int32_t)mine.size() : (int32_t)maxPer);
        rc = cvx_allgather_boundaries_status(ctx, dRec, announce, (int32_t)maxPer, dAll, counts.data()); if (rc) return rc;
        for (int r = 0; r < W; r++) if (counts[(size_t)r] < 0) {
            if (localErr) { ctx->err = localMsg; return localErr; }
            CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": rank " + std::to_string(r) + " failed before the exchange (code " + std::to_string(counts[(size_t)r]) + ")");
        }
        bool overflow = false;
        for (int r = 0; r < W; r++) if (counts[(size_t)r] >= maxPer) overflow = true;
        if (overflow && attempt < 2 && maxPer < hardMax) continue;
        if (overflow) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, std::string(what) + ": the lists do not fit the exchange");
        std::vector<int32_t> flat((size_t)W * (1 + (size_t)maxPer));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(flat.data(), dAll, flat.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        all.assign((size_t)W, std::vector<int32_t>());
        for (int r = 0; r < W; r++) { const int32_t* p = flat.data() + (size_t)r * (1 + (size_t)maxPer); all[(size_t)r].assign(p + 1, p + 1 + counts[(size_t)r]); }
        return CANVAS_OK;
    }
    return CANVAS_OK;
}
// the lists of every rank -> per-chromosome lists; every chromosome must come from exactly its owner
static int32_t lists_by_chromosome(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const std::vector<std::vector<int32_t>>& all, const char* what, std::vector<std::vector<int32_t>>& perChr) {
    perChr.assign((size_t)nchr, std::vector<int32_t>());
    std::vector<char> seen((size_t)nchr, 0);
    for (size_t r = 0; r < all.size(); r++) {
        const std::vector<int32_t>& v = all[r];
        for (size_t i = 0; i < v.size();) {
            if (i + 2 > v.size()) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": truncated list");
            const int32_t c = v[i], n = v[i + 1];
            if (c < 0 || c >= nchr || n < 0 || i + 2 + (size_t)n > v.size() || h_chr_owner[c] != (int32_t)r || seen[(size_t)c]) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": malformed list");
            seen[(size_t)c] = 1; perChr[(size_t)c].assign(v.begin() + (long)i + 2, v.begin() + (long)i + 2 + n);
            i += 2 + (size_t)n;
        }
    }
    for (int c = 0; c < nchr; c++) if (!seen[(size_t)c] && h_chr_owner[c] >= 0 && h_chr_owner[c] < (int32_t)all.size()) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": a chromosome's list is missing");
    return CANVAS_OK;
}
static int32_t check_owner_table(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const char* what) {
    if (nchr <= 0 || !h_chr_owner) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, std::string(what) + ": bad arguments");
    for (int c = 0; c < nchr; c++) if (h_chr_owner[c] < 0 || h_chr_owner[c] >= ctx->nranks) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, std::string(what) + ": owner outside [0, ranks)");
    return CANVAS_OK;
}

extern "C" int32_t canvas_cbs_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, double alpha, uint32_t nperm,
                                      int32_t undo, double undo_sd, int32_t* d_seg_len, int32_t* h_nseg, int64_t* h_stats) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const char* what = "canvas_cbs_sharded";
    int32_t rc = check_owner_table(ctx, nchr, h_chr_owner, what); if (rc) return rc;          // (the same table on every rank: fails everywhere or nowhere)
    if (!h_chr_offset || !d_seg_len || !h_nseg) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_cbs_sharded: bad arguments");
    std::vector<uint8_t> mine((size_t)nchr);
    for (int c = 0; c < nchr; c++) mine[(size_t)c] = h_chr_owner[c] == ctx->rank;
    std::vector<std::vector<int>> segs;
    int32_t localErr = cvx_cbs_masked(ctx, nchr, d_cov, h_chr_offset, alpha, nperm, undo, undo_sd, mine.data(), segs, h_stats);
    const std::string localMsg = ctx->err;
    std::vector<int32_t> list;
    if (!localErr) for (int c = 0; c < nchr; c++) if (mine[(size_t)c]) { list.push_back(c); list.push_back((int32_t)segs[(size_t)c].size()); for (int v : segs[(size_t)c]) list.push_back(v); }
    const int64_t N = h_chr_offset[nchr];
    std::vector<std::vector<int32_t>> all, perChr;
    rc = exchange_lists(ctx, list, localErr, localMsg, N + 2 * (int64_t)nchr + 16, what, all); if (rc) return rc;
    rc = lists_by_chromosome(ctx, nchr, h_chr_owner, all, what, perChr); if (rc) return rc;
    std::vector<int32_t> flat((size_t)N + 1, 0);
    for (int c = 0; c < nchr; c++) {
        const int64_t L = h_chr_offset[c + 1] - h_chr_offset[c]; long long sum = 0;
        for (int32_t v : perChr[(size_t)c]) sum += v;
        if ((int64_t)perChr[(size_t)c].size() > L || (L > 0 && sum != L)) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, "canvas_cbs_sharded: a gathered chromosome's segment lengths do not add up to its bins");
        h_nseg[c] = (int32_t)perChr[(size_t)c].size();
        for (size_t i = 0; i < perChr[(size_t)c].size(); i++) flat[(size_t)h_chr_offset[c] + i] = perChr[(size_t)c][i];
    }
    if (N > 0) CANVAS_HIP_TRY(ctx, hipMemcpyAsync(d_seg_len, flat.data(), (size_t)N * 4, hipMemcpyHostToDevice, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}

static int32_t shard_ws_reserve(canvas_ctx* ctx, size_t need);
// PerSampleHMM with the chromosomes sharded over the ranks, on a coverage every rank holds (e.g. behind the bin intersection of a pedigree: the step that lies between
// CanvasClean and CanvasPartition there, so canvas_sample_pipeline_sharded cannot serve it).  The emission parameters come from the quartiles of the WHOLE coverage
// (HiddenMarkovModelsRunner.cs:36-50); a rank runs the Viterbi passes of its own chromosomes (cvx_hmm_per_sample_subset on a compact copy).  The result never leaves the device
// (round 5; before, the states went to the host, became run lists there, were gathered, expanded on the host and uploaded again): the state runs are recorded by the kernels
// of the sharded pipeline (k_sh_flags / k_sh_count / k_sh_scan / k_sh_scatter / k_sh_ends: (chromosome, first bin, last bin, state) per run), gathered by
// canvas_allgather_boundaries' collective (ncclAllGather on an RCCL communicator) and expanded by k_sh_fill_state into every rank's d_state.  The host sees one count.
extern "C" int32_t canvas_hmm_per_sample_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, int32_t* d_state) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const char* what = "canvas_hmm_per_sample_sharded";
    int32_t rc = check_owner_table(ctx, nchr, h_chr_owner, what); if (rc) return rc;
    if (!h_chr_offset || !d_cov || !d_state || h_chr_offset[0] != 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_per_sample_sharded: bad arguments");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int W = ctx->nranks;
    const int64_t N = h_chr_offset[nchr];
    if (N >= 0x7FFFFFF0ll / 4) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_hmm_per_sample_sharded: too many bins");
    std::vector<int> mine; std::vector<long long> loff{0};
    for (int c = 0; c < nchr; c++) if (h_chr_owner[c] == ctx->rank) { mine.push_back(c); loff.push_back(loff.back() + (h_chr_offset[c + 1] - h_chr_offset[c])); }
    const int nl = (int)mine.size(); const int64_t nLocal = loff.back();
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    // persistent buffers of this rank (one allocation: growing it frees the old one): compact coverage and states, record flags, block counts, tables
    const int nb = (int)((nLocal + SH_BLK - 1) / SH_BLK);
    const size_t oCov = 0, oState = oCov + al((size_t)std::max<int64_t>(nLocal, 1) * 8), oFlags = oState + al((size_t)std::max<int64_t>(nLocal, 1) * 4), oBlk = oFlags + al((size_t)nLocal + 16),
                 oNrec = oBlk + al((size_t)(nb + 2) * 4), oLoff = oNrec + 256, oL2G = oLoff + al((size_t)(nl + 1) * 8), oOwner = oL2G + al((size_t)(nl + 1) * 4), oChrOff = oOwner + al((size_t)nchr * 4),
                 oBad = oChrOff + al((size_t)(nchr + 1) * 8), totalB = oBad + 256;
    int32_t localErr = shard_ws_reserve(ctx, totalB); std::string localMsg;
    if (localErr) localMsg = ctx->err;
    char* S = (char*)ctx->shard_ws;
    double* dCovL = (double*)(S + oCov); int32_t* dStateL = (int32_t*)(S + oState); uint8_t* dFlags = (uint8_t*)(S + oFlags); uint32_t* dBlk = (uint32_t*)(S + oBlk); unsigned int* dNrec = (unsigned int*)(S + oNrec);
    long long* dLoff = (long long*)(S + oLoff); int32_t* dL2G = (int32_t*)(S + oL2G); int32_t* dOwner = (int32_t*)(S + oOwner); long long* dChrOff = (long long*)(S + oChrOff); int* dBad = (int*)(S + oBad);
    auto fail_local = [&](int32_t code) { if (!localErr) { localErr = code; localMsg = ctx->err; } };
    if (!localErr && nLocal > 0) {
        for (int i = 0; i < nl && !localErr; i++) { const int64_t T = loff[(size_t)i + 1] - loff[(size_t)i];
            if (T > 0 && hipMemcpyAsync(dCovL + loff[(size_t)i], d_cov + h_chr_offset[mine[(size_t)i]], (size_t)T * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { ctx->err = std::string(what) + ": copy of the owned coverage failed"; fail_local(CANVAS_ERR_HIP); } }
        if (!localErr) { std::vector<int64_t> loff64(loff.begin(), loff.end()); int32_t rch = cvx_hmm_per_sample_subset(ctx, nl, dCovL, loff64.data(), d_cov, N, dStateL, nullptr); if (rch) fail_local(rch); }
        std::vector<int32_t> l2g(mine.begin(), mine.end());
        if (!localErr) { int32_t r2 = canvas_h2d_small(ctx, dLoff, loff.data(), (size_t)(nl + 1) * 8); if (r2) fail_local(r2); }
        if (!localErr) { int32_t r2 = canvas_h2d_small(ctx, dL2G, l2g.data(), (size_t)nl * 4); if (r2) fail_local(r2); }
        if (!localErr) {
            hipLaunchKernelGGL(k_sh_flags, dim3((unsigned)((nLocal + 255) / 256)), dim3(256), 0, ctx->stream, dStateL, dLoff, nl, nLocal, dFlags);
            hipLaunchKernelGGL(k_sh_count, dim3(nb), dim3(256), 0, ctx->stream, dFlags, nLocal, dBlk);
            hipLaunchKernelGGL(k_sh_scan, dim3(1), dim3(64), 0, ctx->stream, dBlk, nb, dNrec);
        }
    }
    if (!localErr) { int32_t r2 = canvas_h2d_small(ctx, dOwner, h_chr_owner, (size_t)nchr * 4); if (r2) fail_local(r2); }
    // every rank derives the same first bound (a run per 256 bins and two per chromosome: a WGS path has a few hundred runs) and the same hard one (a run per bin)
    int32_t maxPer = (int32_t)std::min<long long>(4ll * (N / 256 + 2ll * nchr + 64), 0x7FFFFFF0ll);
    const long long hardMax = std::min<long long>(4ll * (N + nchr + 16), 0x7FFFFFF0ll);
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t recBytes = al((size_t)maxPer * 4) + al((size_t)W * (1 + (size_t)maxPer) * 4) + 4096;
        // (the same allocation on every rank.)  A rank whose reservation fails — on the second attempt it is W x 4 (N + nchr) words — must still ENTER the collective, or its
        // peers wait in ncclAllGather for ever: it announces the failure from a pinned host area of the same size (device-accessible, so the collective accepts it) and every
        // rank returns an error.  Only when that allocation fails too is there nothing left to announce with.
        rc = canvas_ws_reserve(ctx, recBytes + (size_t)(1 + maxPer) * 4 + 4096);
        { const char* inj = cvx_hook("CANVAS_HMM_SHARDED_FAIL_RESERVE");      // test hook "rank:attempt": that rank's reservation of that attempt fails
          if (inj && !rc) { int ir = -1, ia = -1; if (sscanf(inj, "%d:%d", &ir, &ia) == 2 && ir == ctx->rank && ia == attempt) { rc = CANVAS_ERR_HIP; ctx->err = std::string(what) + ": workspace reservation failed (injected by CANVAS_HMM_SHARDED_FAIL_RESERVE)"; } } }
        if (rc) {
            fail_local(rc);
            const size_t slot = (size_t)(1 + maxPer) * 4;
            void* area = nullptr;
            if (hipHostMalloc(&area, slot * (size_t)(W + 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->err = localMsg; return localErr; }
            int32_t* hs = (int32_t*)area; memset(hs, 0, slot); hs[0] = localErr < 0 ? localErr : -localErr;
            const int32_t rcg = cvx_allgather(ctx, hs, (char*)area + slot, slot);
            (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(area);
            if (rcg) return rcg;
            ctx->err = localMsg; return localErr;
        }
        char* wsb = (char*)ctx->ws + al((size_t)(1 + maxPer) * 4 + 256);      // canvas_allgather_boundaries packs into the FRONT of the workspace
        int32_t* dRec = (int32_t*)wsb; int32_t* dAll = (int32_t*)(wsb + al((size_t)maxPer * 4));
        unsigned int nrec = 0;
        if (nLocal > 0 && !localErr) {
            hipLaunchKernelGGL(k_sh_scatter, dim3(nb), dim3(256), 0, ctx->stream, dFlags, dBlk, dStateL, dLoff, dL2G, nl, nLocal, maxPer / 4, dRec);
            hipLaunchKernelGGL(k_sh_ends, dim3((unsigned)((maxPer / 4 + 255) / 256)), dim3(256), 0, ctx->stream, dRec, dNrec, maxPer / 4);
            if (hipMemcpyAsync(&nrec, dNrec, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = std::string(what) + ": the record count did not come back"; fail_local(CANVAS_ERR_HIP); }
        }
        const int32_t nrecInts = (int32_t)std::min<long long>((long long)nrec * 4, maxPer);
        std::vector<int32_t> counts((size_t)W, 0);
        rc = cvx_allgather_boundaries_status(ctx, dRec, localErr ? (localErr < 0 ? localErr : -localErr) : nrecInts, maxPer, dAll, counts.data()); if (rc) return rc;
        for (int r = 0; r < W; r++) if (counts[(size_t)r] < 0) {
            if (localErr) { ctx->err = localMsg; return localErr; }
            CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": rank " + std::to_string(r) + " failed before the exchange (code " + std::to_string(counts[(size_t)r]) + ")");
        }
        bool overflow = (long long)nrec * 4 >= maxPer;
        for (int r = 0; r < W; r++) if (counts[(size_t)r] >= maxPer) overflow = true;
        if (overflow && attempt == 0 && maxPer < hardMax) { maxPer = (int32_t)hardMax; continue; }
        if (overflow) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, std::string(what) + ": the state runs do not fit the exchange");
        if (N > 0) {
            std::vector<long long> chrOff64(h_chr_offset, h_chr_offset + nchr + 1);
            // (from here on a failure is local: the function's only collective is behind every rank)
            rc = canvas_h2d_small(ctx, dChrOff, chrOff64.data(), (size_t)(nchr + 1) * 8); if (rc) return rc;
            CANVAS_HIP_TRY(ctx, hipMemsetAsync(dBad, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_sh_fill_state, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, dAll, 1 + maxPer, dOwner, dChrOff, nchr, N, d_state, dBad);
            int bad = 0;
            CANVAS_HIP_TRY(ctx, hipMemcpyAsync(&bad, dBad, 4, hipMemcpyDeviceToHost, ctx->stream));
            CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            CANVAS_HIP_TRY(ctx, hipGetLastError());
            if (bad) CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string(what) + ": the gathered state runs do not cover every bin");
        }
        ctx->shard_stats[0] = W; ctx->shard_stats[1] = nl; ctx->shard_stats[4] = nrec; ctx->shard_stats[5] = (long long)(1 + maxPer) * 4;
        return CANVAS_OK;
    }
    return CANVAS_OK;
}

extern "C" int32_t canvas_wavelets_sharded(canvas_ctx* ctx, int32_t nchr, const int32_t* h_chr_owner, const double* d_cov, const int64_t* h_chr_offset, int32_t is_germline,
                                           double threshold_lower, double threshold_upper, double mad_factor, int32_t variability_window, int32_t min_size,
                                           int32_t* h_breakpoints, int64_t cap, int64_t* h_bp_offset) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const char* what = "canvas_wavelets_sharded";
    int32_t rc = check_owner_table(ctx, nchr, h_chr_owner, what); if (rc) return rc;
    if (!h_chr_offset || !h_breakpoints || !h_bp_offset || cap < 0) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_wavelets_sharded: bad arguments");
    std::vector<uint8_t> mine((size_t)nchr);
    for (int c = 0; c < nchr; c++) mine[(size_t)c] = h_chr_owner[c] == ctx->rank;
    const int64_t N = h_chr_offset[nchr] - h_chr_offset[0];
    std::vector<int32_t> bp((size_t)std::max<int64_t>(N, 1)); std::vector<int64_t> bo((size_t)nchr + 1, 0);
    int32_t localErr = cvx_wavelets_masked(ctx, nchr, d_cov, h_chr_offset, is_germline, threshold_lower, threshold_upper, mad_factor, variability_window, min_size, mine.data(), bp.data(), (int64_t)bp.size(), bo.data());
    const std::string localMsg = ctx->err;
    std::vector<int32_t> list;
    if (!localErr) for (int c = 0; c < nchr; c++) if (mine[(size_t)c]) { list.push_back(c); list.push_back((int32_t)(bo[(size_t)c + 1] - bo[(size_t)c])); list.insert(list.end(), bp.begin() + bo[(size_t)c], bp.begin() + bo[(size_t)c + 1]); }
    std::vector<std::vector<int32_t>> all, perChr;
    rc = exchange_lists(ctx, list, localErr, localMsg, N + 2 * (int64_t)nchr + 16, what, all); if (rc) return rc;
    rc = lists_by_chromosome(ctx, nchr, h_chr_owner, all, what, perChr); if (rc) return rc;
    int64_t total = 0;
    for (int c = 0; c < nchr; c++) {
        h_bp_offset[c] = total;
        if (total + (int64_t)perChr[(size_t)c].size() > cap) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_wavelets_sharded: breakpoint capacity too small");
        for (int32_t v : perChr[(size_t)c]) h_breakpoints[total++] = v;
    }
    h_bp_offset[nchr] = total;
    return CANVAS_OK;
}

// ================================================================================================ the sample axis: one sample of a pedigree per rank
// CanvasRunner runs CanvasBin and CanvasClean once per sample of a pedigree (CanvasRunner.cs:905-927, 1010-1070) — independent until two points: the multi-sample bin
// size (the median rate over ALL samples' autosomes, CanvasBin.cs:86-110) and the bin intersection between CanvasClean and CanvasPartition
// (MergeMultiSampleCleanedBedFile, Utilities.cs:834-920).  With one sample per rank those are two exchanges:
//   canvas_allgather_host          a few host bytes per rank (the per-chromosome rates): every rank then derives the same bin size with canvas_bin_size_from_rates;
//   canvas_merge_cleaned_sharded   every rank contributes the (chr, start, stop) columns of its cleaned bins, padded to the largest sample (12 B per bin), receives
//                                  everybody's and runs the same intersection as canvas_merge_cleaned with the samples in rank order: the merged bin list (first sample's
//                                  order, stop from the last sample) comes out identical on every rank, together with THIS rank's counts of the surviving bins.
// The count exchange in front of the columns carries the status word: a rank that failed sends a negative count and every rank returns an error.
static int32_t shard_ws_reserve(canvas_ctx* ctx, size_t need) {
    if (need <= ctx->shard_ws_bytes) return CANVAS_OK;
    if (ctx->shard_ws) { CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); CANVAS_HIP_TRY(ctx, hipFree(ctx->shard_ws)); ctx->shard_ws = nullptr; ctx->shard_ws_bytes = 0; }
    CANVAS_HIP_TRY(ctx, hipMalloc(&ctx->shard_ws, need)); ctx->shard_ws_bytes = need;
    return CANVAS_OK;
}
extern "C" int32_t canvas_allgather_host(canvas_ctx* ctx, const void* h_send, int64_t bytes_per_rank, void* h_recv) {
    if (!ctx) return CANVAS_ERR_INVALID;
    if (!h_send || !h_recv || bytes_per_rank <= 0 || bytes_per_rank > (1 << 24)) CANVAS_FAIL(ctx, CANVAS_ERR_INVALID, "canvas_allgather_host: 1 .. 16 MiB per rank");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t b = ((size_t)bytes_per_rank + 255) & ~size_t(255);
    int32_t rc = shard_ws_reserve(ctx, b * (size_t)(ctx->nranks + 1) + 256); if (rc) return rc;
    char* dS = (char*)ctx->shard_ws; char* dR = dS + b;
    CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dS, h_send, (size_t)bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
    rc = cvx_allgather(ctx, dS, dR, b); if (rc) return rc;
    for (int r = 0; r < ctx->nranks; r++) CANVAS_HIP_TRY(ctx, hipMemcpyAsync((char*)h_recv + (size_t)r * (size_t)bytes_per_rank, dR + (size_t)r * b, (size_t)bytes_per_rank, hipMemcpyDeviceToHost, ctx->stream));
    CANVAS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANVAS_OK;
}
extern "C" int32_t canvas_merge_cleaned_sharded(canvas_ctx* ctx, int64_t n_mine, const int32_t* d_chr, const int32_t* d_start, const int32_t* d_stop, const float* d_count,
                                                int32_t* d_out_chr, int32_t* d_out_start, int32_t* d_out_stop, float* d_out_count, int64_t cap, int64_t* h_n_out) {
    if (!ctx) return CANVAS_ERR_INVALID;
    const int W = ctx->nranks, me = ctx->rank;
    if (W > 16) CANVAS_FAIL(ctx, CANVAS_ERR_UNSUPPORTED, "canvas_merge_cleaned_sharded: at most 16 samples (ranks)");
    CANVAS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t localErr = CANVAS_OK; std::string localMsg;
    if (n_mine < 0 || n_mine > 0x7FFFFFF0ll || !h_n_out || (n_mine > 0 && (!d_chr || !d_start || !d_stop || !d_count)) || !d_out_chr || !d_out_start || !d_out_stop || !d_out_count) {
        localErr = CANVAS_ERR_INVALID; localMsg = "canvas_merge_cleaned_sharded: bad arguments";
    }
    // ---- 1. how many bins every sample has (negative: that rank has failed)
    std::vector<int64_t> ns((size_t)W, 0);
    { const int64_t mine = localErr ? (int64_t)(localErr < 0 ? localErr : -localErr) : n_mine;
      int32_t rc = canvas_allgather_host(ctx, &mine, 8, ns.data()); if (rc) return rc; }
    for (int r = 0; r < W; r++) if (ns[(size_t)r] < 0) {
        if (localErr) { ctx->err = localMsg; return localErr; }
        CANVAS_FAIL(ctx, CANVAS_ERR_COMM, std::string("canvas_merge_cleaned_sharded: rank ") + std::to_string(r) + " failed before the exchange (code " + std::to_string((long long)ns[(size_t)r]) + ")");
    }
    const int64_t nmax = std::max<int64_t>(1, *std::max_element(ns.begin(), ns.end()));
    // ---- 2. the key columns of every sample
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t col = al((size_t)nmax * 4), slot = 3 * col;
    int32_t rc = shard_ws_reserve(ctx, slot * (size_t)(W + 1) + al((size_t)ns[0] * 4 + 4) + 4096); if (rc) return rc;       // (the same size on every rank)
    char* dS = (char*)ctx->shard_ws; char* dR = dS + slot; float* dScratch = (float*)(dR + slot * (size_t)W);
    if (n_mine > 0) {
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dS, d_chr, (size_t)n_mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dS + col, d_start, (size_t)n_mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
        CANVAS_HIP_TRY(ctx, hipMemcpyAsync(dS + 2 * col, d_stop, (size_t)n_mine * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    rc = cvx_allgather(ctx, dS, dR, slot); if (rc) return rc;
    // ---- 3. the intersection, samples in rank order (no further exchange: a failure from here on is local)
    if (cap < ns[0]) CANVAS_FAIL(ctx, CANVAS_ERR_CAPACITY, "canvas_merge_cleaned_sharded: the output arrays must hold the first sample's bins");
    std::vector<const int32_t*> pc((size_t)W), ps((size_t)W), pe((size_t)W); std::vector<const float*> pv((size_t)W); std::vector<float*> po((size_t)W);
    for (int r = 0; r < W; r++) {
        const char* base = dR + (size_t)r * slot;
        pc[(size_t)r] = (const int32_t*)base; ps[(size_t)r] = (const int32_t*)(base + col); pe[(size_t)r] = (const int32_t*)(base + 2 * col);
        pv[(size_t)r] = r == me ? d_count : (const float*)(base + col);          // the other samples' counts are not here and not needed: any readable array of their length
        po[(size_t)r] = r == me ? d_out_count : dScratch;
    }
    return canvas_merge_cleaned(ctx, W, ns.data(), pc.data(), ps.data(), pe.data(), pv.data(), d_out_chr, d_out_start, d_out_stop, po.data(), h_n_out);
}
