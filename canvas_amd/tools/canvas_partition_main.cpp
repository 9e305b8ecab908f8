// Drop-in CanvasPartition executable on top of the C ABI: CLI and file formats of CanvasPartition.Main (CanvasPartition/CanvasPartition.cs:24-190).
//   CanvasPartition -i S.cleaned [-i ...] -o S.partitioned [-o ...] -r refDir [-m Wavelets|PerSampleHMM|HMM|CBS] [-g] [-v S.vaf] [-b filter.bed] [-s None|Prune|SDUndo] [--config params.json]
// Built methods: Wavelets (the default, one sample), PerSampleHMM, HMM (joint) and CBS.
//   -p ploidy.vcf            reference ploidy (CanvasRunner.InvokeCanvasPartition ALWAYS passes it, CanvasRunner.cs:950): a segment also starts where the reference
//                            ploidy changes between two bins (SegmentationResultsProcessor.cs:117-128, PloidyInfo.cs:78-165)
//   --evenness-metric-file F "#evenness\t<score>" (Somatic-WGS, CanvasRunner.cs:958-960): canvas_evenness_score; like the reference only the Wavelets method
//                            computes it (WaveletsRunner.cs:58-67)
//   -c commonCNVs.bed        accepted and, exactly as in the reference, unused: CanvasPartition.cs:121 hands the path to WaveletsRunnerParams.CommonCNVs
//                            (WaveletsRunner.cs:20,34), which nothing reads; the other methods never see it (CanvasRunner.cs:916-917 passes it with PerSampleHMM)
// Wavelets and -v: WaveletsRunner.Run only derives segments for the chromosomes of SegmentationInput.VafByChr (WaveletsRunner.cs:75), which LoadVAFInput
// fills for every chromosome of the coverage file when -v is given and leaves empty otherwise (Segmentation.cs:78-79, 158-168).  The allele frequencies themselves
// never reach the Wavelets method (AdjustBreakpoints gets null, WaveletsRunner.cs:71), so this tool only checks that the -v file exists.
#include "tool_common.hpp"
#include <atomic>
#include <algorithm>
#include <set>
using namespace tool;

struct Sample {
    std::vector<std::string> chromNames; std::vector<int64_t> off;           // CoverageInfo in file order
    std::vector<uint32_t> start, end; std::vector<double> cov;
};

// GenomicBinFilter.SkipBin (CanvasCommon/GenomicBinFilter.cs:29-58)
struct BinFilter {
    const std::map<std::string, std::vector<std::pair<int, int>>>* excl; std::string prevChrom; bool havePrev = false; uint32_t prevStart = 0; const std::vector<std::pair<int, int>>* iv = nullptr; size_t idx = 0;
    bool skip(const std::string& chrom, uint32_t start, uint32_t stop) {
        static const std::vector<std::pair<int, int>> none;
        if (!havePrev || chrom != prevChrom) { prevChrom = chrom; havePrev = true; auto it = excl->find(chrom); iv = it == excl->end() ? &none : &it->second; idx = 0; }
        else if (start < prevStart) idx = 0;
        prevStart = start;
        for (; idx < iv->size(); idx++) { if ((uint32_t)(*iv)[idx].second <= start) continue; if ((uint32_t)(*iv)[idx].first >= stop) return false; return true; }
        return false;
    }
};

// PloidyInterval (PloidyInfo.cs:182-198): one-based Start = POS, End = INFO/END, Ploidy = the sample's CN field ("." = 2)
struct PloidyIv { int start, end, ploidy; };
// the single-sample ploidy VCF as Isas' VcfReader exposes it to PloidyInfo.LoadPloidyFromVcfFile (PloidyInfo.cs:112-165); plain or gzip text
static bool load_ploidy_vcf(const std::string& path, std::map<std::string, std::vector<PloidyIv>>& out, std::string& err) {
    GzReader rd(path); if (!rd.ok()) { err = "cannot open ploidy VCF '" + path + "'"; return false; }
    std::string row; int samples = -1;
    while (rd.line(row)) {
        if (row.empty()) continue;
        if (row[0] == '#') { if (row.rfind("#CHROM", 0) == 0) { auto h = split_tab(row); samples = (int)h.size() > 9 ? (int)h.size() - 9 : 0; } continue; }
        if (samples < 0) { err = "File '" + path + "' has no #CHROM header line"; return false; }
        if (samples == 0) { err = "File '" + path + "' does not contain any genotype column"; return false; }
        if (samples > 1) { err = "File '" + path + "' cannot have more than one genotype columns when no sample ID provided"; return false; }
        auto f = split_tab(row);
        if (f.size() < 10) { err = "malformed ploidy VCF record: " + row; return false; }
        PloidyIv iv; iv.start = atoi(f[1].c_str()); iv.end = -1; iv.ploidy = 2;
        bool haveEnd = false;
        for (size_t a0 = 0; a0 <= f[7].size();) { size_t b = f[7].find(';', a0); std::string kv = f[7].substr(a0, b == std::string::npos ? b : b - a0);
            if (kv.rfind("END=", 0) == 0) { iv.end = atoi(kv.c_str() + 4); haveEnd = true; } if (b == std::string::npos) break; a0 = b + 1; }
        if (!haveEnd) { err = "ploidy VCF record without INFO/END: " + row; return false; }            // InfoFields["END"] throws KeyNotFoundException
        std::vector<std::string> keys, vals;
        for (int which = 0; which < 2; which++) { const std::string& src = f[which == 0 ? 8 : 9]; auto& dst = which == 0 ? keys : vals;
            for (size_t a0 = 0;;) { size_t b = src.find(':', a0); dst.push_back(src.substr(a0, b == std::string::npos ? b : b - a0)); if (b == std::string::npos) break; a0 = b + 1; } }
        bool haveCn = false;
        for (size_t k = 0; k < keys.size() && k < vals.size(); k++) if (keys[k] == "CN") { haveCn = true; iv.ploidy = vals[k] == "." ? 2 : atoi(vals[k].c_str()); }
        if (!haveCn) { err = "File '" + path + "' must contain one genotype CN column!"; return false; }
        out[f[0]].push_back(iv);
    }
    if (samples < 0) { err = "File '" + path + "' has no #CHROM header line"; return false; }
    if (samples == 0) { err = "File '" + path + "' does not contain any genotype column"; return false; }
    if (samples > 1) { err = "File '" + path + "' cannot have more than one genotype columns when no sample ID provided"; return false; }
    return true;
}
// PloidyInfo.IsUniformReferencePloidy over the one-based interval [qs, qe] (PloidyInfo.cs:78-110); -1: a ploidy outside 0..4 indexes past baseCounts (the reference throws)
static int is_uniform_reference_ploidy(const std::vector<PloidyIv>& ivs, int qs, int qe) {
    int baseCounts[5] = {0, 0, qe - qs + 1, 0, 0};
    for (auto& iv : ivs) {
        if (iv.ploidy == 2) continue;
        const int overlapStart = std::max(qs - 1, iv.start - 1);
        if (overlapStart > iv.end) continue;
        const int overlapBases = std::min(qe, iv.end) - overlapStart;
        if (overlapBases <= 0) continue;
        if (iv.ploidy < 0 || iv.ploidy > 4) return -1;
        baseCounts[2] -= overlapBases; baseCounts[iv.ploidy] += overlapBases;
    }
    int nonZero = 0; for (int v : baseCounts) if (v > 0) nonZero++;
    return nonZero < 2 ? 1 : 0;
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);       // (read by the HIP runtime at its first call: -m CBS keeps a dozen kernels in flight, the default maps all streams onto 4 hardware queues)
    printf(">>>Command-line arguments:\n"); for (int i = 1; i < argc; i++) printf("%s ", argv[i]); printf("\n");
    std::vector<Opt> opts = {{"i", "infile", true}, {"v", "vaffile", true}, {"o", "outfile", true}, {"m", "method", true}, {"r", "reference", true}, {"s", "split", true},
                             {"b", "bedfile", true}, {"c", "commoncnvs", true}, {"g", "germline", false}, {"", "evenness-metric-file", true}, {"p", "ploidyVcfFile", true},
                             {"", "config", true}, {"h", "help", false}};
    Parsed a = parse(argc, argv, opts);
    if (!a.extra.empty()) { fprintf(stderr, "Unknown arguments: %s\n", a.extra[0].c_str()); return 2; }
    auto help = []() { printf("Usage: CanvasPartition.exe [OPTIONS]+\nDivide bins into consistent intervals based on their counts\n\nOptions:\n  -i, --infile=VALUE (repeatable)  -o, --outfile=VALUE (repeatable)  -m, --method=VALUE  -r, --reference=VALUE\n"
                              "  -s, --split=VALUE  -b, --bedfile=VALUE  -c, --commoncnvs=VALUE  -g, --germline  --evenness-metric-file=VALUE  -p, --ploidyVcfFile=VALUE  --config=VALUE  -h, --help\n"); };
    auto inFiles = a.all("infile"), outFiles = a.all("outfile");
    if (a.has("help") || inFiles.empty() || outFiles.empty() || !a.has("reference")) { help(); return 0; }     // CanvasPartition.cs:66-76
    for (auto& f : inFiles) if (!file_exists(f)) { printf("CanvasPartition.exe: File %s does not exist! Exiting.\n", f.c_str()); return 1; }
    const std::string bed = a.get("bedfile");
    if (!bed.empty() && !file_exists(bed)) { printf("CanvasPartition.exe: File %s does not exist! Exiting.\n", bed.c_str()); return 1; }
    std::string method = a.get("method", "Wavelets");
    if (method != "PerSampleHMM" && method != "CBS" && method != "HMM" && method != "Wavelets") { fprintf(stderr, "CanvasPartition (MI355X): unknown method %s (Wavelets, PerSampleHMM, HMM, CBS)\n", method.c_str()); return 1; }
    const std::string ploidyVcf = a.get("ploidyVcfFile");
    if (!ploidyVcf.empty() && !file_exists(ploidyVcf)) { printf("CanvasPartition.exe: File %s does not exist! Exiting.\n", ploidyVcf.c_str()); return 1; }   // CanvasPartition.cs:96-100
    // PloidyInfo.LoadPloidyFromVcfFileNoSampleId (PloidyInfo.cs:112-165).  `-p ""` (CanvasRunner.cs:950 with no ploidy VCF configured) reaches the loader in the
    // reference as well (ploidyVcfPath != null, CanvasPartition.cs:114) and throws there: same outcome here, exit code 1.
    std::map<std::string, std::vector<PloidyIv>> ploidyByChrom; const bool havePloidy = a.has("ploidyVcfFile");
    if (havePloidy) { std::string err; if (!load_ploidy_vcf(ploidyVcf, ploidyByChrom, err)) { fprintf(stderr, "CanvasPartition: %s\n", err.c_str()); return 1; } }
    const std::string evennessFile = a.get("evenness-metric-file");
    for (auto& f : a.all("vaffile")) if (!file_exists(f)) { printf("CanvasPartition.exe: File %s does not exist! Exiting.\n", f.c_str()); return 1; }
    if (method == "Wavelets" && inFiles.size() != 1) { fprintf(stderr, "CanvasPartition: -m Wavelets takes exactly one sample (segmentationInputs.Single(), CanvasPartition.cs:125)\n"); return 1; }
    if (inFiles.size() != outFiles.size()) { fprintf(stderr, "CanvasPartition: the number of -o must match the number of -i\n"); return 1; }
    std::string split = a.get("split", "None");
    int undo = split == "None" ? 0 : (split == "SDUndo" ? 2 : (split == "Prune" ? 1 : -1));
    if (undo < 0) { fprintf(stderr, "Invalid split method '%s'\n", split.c_str()); return 2; }
    // CanvasPartitionParameters.json (CanvasPartitionParameters.cs:11-16): only the two values this path uses
    int maxInterBinDist = 1000000; double cbsAlpha = 0.01, madFactor = 5.0, thresholdLowerMaf = 0.05; int evennessWindow = 100000;
    if (a.has("config")) { FILE* f = fopen(a.get("config").c_str(), "rb"); if (!f) { printf("CanvasPedigreeCaller.exe: File %s does not exist! Exiting.\n", a.get("config").c_str()); return 1; }
        std::string js; char buf[4096]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) js.append(buf, k); fclose(f);
        auto num = [&](const char* key, double d) { size_t p = js.find(key); if (p == std::string::npos) return d; p = js.find(':', p); return p == std::string::npos ? d : strtod(js.c_str() + p + 1, nullptr); };
        maxInterBinDist = (int)num("\"MaxInterBinDistInSegment\"", maxInterBinDist); cbsAlpha = num("\"CBSalpha\"", cbsAlpha);
        madFactor = num("\"MadFactor\"", madFactor); thresholdLowerMaf = num("\"ThresholdLowerMaf\"", thresholdLowerMaf); evennessWindow = (int)num("\"EvennessScoreWindow\"", evennessWindow); }
    std::map<std::string, std::vector<std::pair<int, int>>> excluded;
    if (!bed.empty()) load_bed(bed, excluded);

    Phases ph("CanvasPartition");
    // the context comes up while the files are read; for -m CBS the helper thread also runs the method once on a toy sample (three chromosomes of 4 000 bins with a step)
    auto warmCbs = [cbsAlpha, undo](canvas_ctx* c) {
        // the chromosomes' draw streams are constants of the method (MersenneTwister(seed_k), seed_k from MersenneTwister(0) in file order, CBSRunner.cs:107-112): the library's
        // generator starts on them NOW, on its own thread and stream, while this process is still parsing its input (25 streams: a human reference's chromosomes; a file with more
        // gets the others when canvas_cbs asks for them)
        (void)canvas_cbs_prefetch(c, 25, int64_t(16) << 20);
        const int nchr = 3; const int64_t per = 4000, N = nchr * per;
        std::vector<double> cov((size_t)N); std::vector<int64_t> off{0, per, 2 * per, 3 * per};
        uint32_t x = 12345u;
        for (int64_t i = 0; i < N; i++) { x = x * 1664525u + 1013904223u; cov[(size_t)i] = 100.0 + (double)((x >> 16) % 21) - 10.0 + ((i % per) > per / 2 && (i % per) < per / 2 + 300 ? 4.0 : 0.0); }
        void* dCov = canvas_device_malloc(c, N * 8); void* dLen = canvas_device_malloc(c, (N + 1) * 4);
        std::vector<int32_t> nseg(nchr); int64_t stats[8];
        if (dCov && dLen && canvas_memcpy_h2d(c, dCov, cov.data(), N * 8) == 0) (void)canvas_cbs_undo(c, nchr, (const double*)dCov, off.data(), cbsAlpha, 10000, undo, 3.0, (int32_t*)dLen, nseg.data(), stats);
        if (dCov) canvas_device_free(c, dCov);
        if (dLen) canvas_device_free(c, dLen);
    };
    AsyncCtx actx(method == "CBS" ? std::function<void(canvas_ctx*)>(warmCbs) : nullptr);
    // CanvasSegment.ReadBedInput (CanvasCommon/CanvasSegment.cs:1117-1163)
    std::vector<Sample> samples(inFiles.size());
    for (size_t s = 0; s < inFiles.size(); s++) {
        Sample& S = samples[s]; BinFilter filt; filt.excl = &excluded;
        std::map<std::string, int> index; std::vector<std::vector<uint32_t>> st, en; std::vector<std::vector<double>> cv;
        TextRows rows;
        if (!read_text_rows(inFiles[s], 4, rows)) { fprintf(stderr, "CanvasPartition: cannot read %s\n", inFiles[s].c_str()); return 1; }      // (rows with fewer than four fields are skipped)
        std::vector<int> ciOf(rows.chromNames.size(), -1);     // chromosome of the file -> chromosome of the sample (a chromosome whose bins are all filtered never gets one)
        for (size_t i = 0; i < rows.chr.size(); i++) {
            const std::string& name = rows.chromNames[(size_t)rows.chr[i]];
            const uint32_t b = rows.start[i], e = rows.stop[i];
            if (filt.skip(name, b, e)) continue;
            int& ci = ciOf[(size_t)rows.chr[i]];
            if (ci < 0) { ci = (int)S.chromNames.size(); index[name] = ci; S.chromNames.push_back(name); st.emplace_back(); en.emplace_back(); cv.emplace_back(); }
            st[ci].push_back(b); en[ci].push_back(e); cv[ci].push_back(rows.value[i]); }
        S.off.push_back(0);
        for (size_t c = 0; c < S.chromNames.size(); c++) { S.start.insert(S.start.end(), st[c].begin(), st[c].end()); S.end.insert(S.end.end(), en[c].begin(), en[c].end()); S.cov.insert(S.cov.end(), cv[c].begin(), cv[c].end()); S.off.push_back((int64_t)S.start.size()); }
    }
    ph.mark("read");
    canvas_ctx* ctx = actx.get();
    if (!ctx) { fprintf(stderr, "CanvasPartition (MI355X): no usable GPU (this build has no CPU fallback)\n"); return 1; }
    // per sample: segments per chromosome as (start, end) genomic pairs
    typedef std::vector<std::pair<uint32_t, uint32_t>> Segs;
    std::vector<std::map<std::string, Segs>> segBySample(samples.size());
    // SegmentationInput.DeriveSegments (Segmentation.cs:83-125) from a state path
    auto derive = [](const Sample& S, const std::vector<int32_t>& state, std::map<std::string, Segs>& out) {
        for (size_t c = 0; c < S.chromNames.size(); c++) {
            int64_t b0 = S.off[c], T = S.off[c + 1] - b0;
            if (!(T > 10)) continue;                                         // chromosome skipped: no entry in segmentByChr (HiddenMarkovModelsRunner.cs:69)
            std::vector<int> bp = {0};
            for (int64_t i = 1; i < T; i++) if (state[b0 + i] != state[b0 + i - 1]) bp.push_back((int)i);
            Segs sg;
            if (bp.size() >= 2) { for (size_t k = 0; k < bp.size(); k++) { int64_t a0 = bp[k], a1 = (k + 1 < bp.size() ? bp[k + 1] : T) - 1; sg.push_back({S.start[b0 + a0], S.end[b0 + a1]}); } }
            else sg.push_back({S.start[b0], S.end[b0 + T - 1]});
            out[S.chromNames[c]] = sg;
        }
    };
    std::map<std::string, Segs> jointSegs;
    if (method == "HMM") {
        // one Viterbi path for all samples over the first sample's bins (HiddenMarkovModelsRunner.cs:51-63; CanvasPartition.cs:146-158)
        printf("Running HMM Partitioning\n");
        const Sample& S0 = samples[0]; const int nchr = (int)S0.chromNames.size(); const int64_t N = S0.off.back();
        for (auto& S : samples) if (S.off != S0.off) { fprintf(stderr, "CanvasPartition: -m HMM needs the same bins in every input\n"); return 1; }
        if (N > 0) {
            std::vector<std::unique_ptr<Dev>> dCovs;
            std::vector<const double*> ptrs;
            for (auto& S : samples) { dCovs.push_back(std::make_unique<Dev>(ctx, N * 8)); TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dCovs.back()->p, S.cov.data(), N * 8)); }
            for (auto& d : dCovs) ptrs.push_back(d->as<double>());
            Dev dState(ctx, N * 4);
            TOOL_TRY(ctx, canvas_hmm_joint(ctx, (int32_t)samples.size(), nchr, ptrs.data(), S0.off.data(), dState.as<int32_t>()));
            std::vector<int32_t> state(N); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, state.data(), dState.p, N * 4));
            derive(S0, state, jointSegs);
        }
    }
    for (size_t s = 0; s < samples.size() && method != "HMM"; s++) {
        Sample& S = samples[s]; const int nchr = (int)S.chromNames.size(); const int64_t N = S.off.back();
        if (N == 0) continue;
        Dev dCov(ctx, N * 8); TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dCov.p, S.cov.data(), N * 8));
        if (method == "Wavelets") {
            printf("Running Wavelet Partitioning\n");
            if (!evennessFile.empty()) {      // WaveletsRunner.cs:58-67: score first, "#evenness\t<double>" (IO.cs:88-98); no file when the reference's Quartiles / Median throw
                double score = 0; int32_t valid = 0;
                TOOL_TRY(ctx, canvas_evenness_score(ctx, nchr, dCov.as<double>(), S.off.data(), evennessWindow, &score, &valid));
                if (valid) { FILE* ef = fopen(evennessFile.c_str(), "wb"); if (!ef) { fprintf(stderr, "cannot write %s\n", evennessFile.c_str()); return 1; }
                    fprintf(ef, "#evenness\t%s\n", format_g(score, 15).c_str()); fclose(ef); }
                else fprintf(stderr, "Unable to calculate an evenness score, using coverage for segmentation\n");
            }
            std::vector<int32_t> bps((size_t)N + nchr + 1); std::vector<int64_t> bo(nchr + 1);
            TOOL_TRY(ctx, canvas_wavelets(ctx, nchr, dCov.as<double>(), S.off.data(), a.has("germline") ? 1 : 0, thresholdLowerMaf, 80.0, madFactor, evennessWindow, 10,
                                          bps.data(), (int64_t)bps.size(), bo.data()));
            if (!a.has("vaffile")) fprintf(stderr, "CanvasPartition: no -v file: like the reference, Wavelets then derives no segment for any chromosome (WaveletsRunner.cs:75)\n");
            for (int c = 0; c < nchr && a.has("vaffile"); c++) {            // SegmentationInput.DeriveSegments (Segmentation.cs:83-125)
                const int64_t b0 = S.off[c], T = S.off[c + 1] - b0;
                std::vector<int32_t> bp(bps.begin() + bo[c], bps.begin() + bo[c + 1]);
                Segs sg;
                if (bp.size() >= 2 && T > 10) {
                    if (bp[0] != 0) bp.insert(bp.begin(), 0);
                    for (size_t k = 0; k < bp.size(); k++) { const int64_t a0 = bp[k], a1 = (k + 1 < bp.size() ? bp[k + 1] : T) - 1; sg.push_back({S.start[b0 + a0], S.end[b0 + a1]}); }
                } else sg.push_back({S.start[b0], S.end[b0 + T - 1]});
                segBySample[s][S.chromNames[c]] = sg;
            }
        } else if (method == "PerSampleHMM") {
            printf("Running Per-sample HMM Partitioning\n");
            Dev dState(ctx, N * 4);
            TOOL_TRY(ctx, canvas_hmm_per_sample(ctx, nchr, dCov.as<double>(), S.off.data(), dState.as<int32_t>()));
            std::vector<int32_t> state(N); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, state.data(), dState.p, N * 4));
            derive(S, state, segBySample[s]);
        } else {
            printf("Running CBS Partitioning\n");
            Dev dLen(ctx, (N + 1) * 4); std::vector<int32_t> nseg(nchr); int64_t stats[8];
            TOOL_TRY(ctx, canvas_cbs_undo(ctx, nchr, dCov.as<double>(), S.off.data(), cbsAlpha, 10000, undo, 3.0, dLen.as<int32_t>(), nseg.data(), stats));
            std::vector<int32_t> lens(N + 1); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, lens.data(), dLen.p, (N + 1) * 4));
            for (int c = 0; c < nchr; c++) {                                  // CBSRunner.cs:127-137
                int64_t b0 = S.off[c]; Segs sg; int64_t cs1 = 0, cs2 = -1;
                for (int k = 0; k < nseg[c]; k++) { cs2 += lens[b0 + k]; sg.push_back({S.start[b0 + cs1], S.end[b0 + cs2]}); cs1 += lens[b0 + k]; }
                segBySample[s][S.chromNames[c]] = sg;
            }
        }
    }
    if (getenv("CANVAS_TOOL_FULL_TEARDOWN")) canvas_destroy(ctx);
    ph.mark("device");
    // GenomeSegmentationResults.SplitOverlappingSegments (GenomeSegmentationResults.cs:18-55)
    std::map<std::string, Segs> merged;
    if (method == "HMM") merged = jointSegs;
    else if (samples.size() == 1) merged = segBySample[0];
    else for (auto& kv : segBySample[0]) {
        const std::string& chrom = kv.first;
        std::vector<std::vector<uint32_t>> st(samples.size()), en(samples.size()); std::vector<const uint32_t*> ps, pe; std::vector<int32_t> ns;
        for (size_t s = 0; s < samples.size(); s++) { for (auto& sg : segBySample[s][chrom]) { st[s].push_back(sg.first); en[s].push_back(sg.second); } }
        for (size_t s = 0; s < samples.size(); s++) { ps.push_back(st[s].data()); pe.push_back(en[s].data()); ns.push_back((int32_t)st[s].size()); }
        int cap = 2; for (auto v : ns) cap += 2 * v;
        std::vector<uint32_t> os(cap), oe(cap); int32_t nout = 0;
        if (canvas_split_overlapping((int32_t)samples.size(), ps.data(), pe.data(), ns.data(), os.data(), oe.data(), cap, &nout) != 0) { fprintf(stderr, "canvas_split_overlapping failed\n"); return 1; }
        Segs sg; for (int k = 0; k < nout; k++) sg.push_back({os[k], oe[k]});
        merged[chrom] = sg;
    }
    // SegmentationResultsProcessor.PostProcessSegments (SegmentationResultsProcessor.cs:17-129) + WriteCanvasPartitionResults (Segmentation.cs:235-252)
    for (size_t s = 0; s < samples.size(); s++) {
        Sample& S = samples[s];
        struct OutRow { uint32_t s, e; double cov; int id; int chrom; }; std::vector<OutRow> outRows;
        // The segment id is a running counter over the whole file (SegmentationResultsProcessor.cs:17-129), but WHERE it advances is decided chromosome by chromosome: every
        // chromosome is walked on its own host thread with a local counter, the counters are offset in file order afterwards (the walk of 4.7 M bins on one thread was half of
        // the tool's "write" phase).
        struct Row { uint32_t s, e; double cov; int id; };
        const size_t nC = S.chromNames.size();
        std::vector<std::vector<Row>> rowsC(nC); std::vector<int> incC(nC, 0); std::atomic<int> badChrom(-1);
        parallel_for((int64_t)nC, [&](int64_t ci) {
            const size_t c = (size_t)ci;
            const std::string& chrom = S.chromNames[c];
            std::vector<uint32_t> starts; auto mit = merged.find(chrom); if (mit != merged.end()) for (auto& sg : mit->second) starts.push_back(sg.first);
            std::sort(starts.begin(), starts.end());
            const std::vector<std::pair<int, int>>* ex = nullptr; auto eit = excluded.find(chrom); if (eit != excluded.end()) ex = &eit->second;
            const std::vector<PloidyIv>* pl = nullptr; if (havePloidy) { auto pit = ploidyByChrom.find(chrom); if (pit != ploidyByChrom.end()) pl = &pit->second; }
            size_t exIdx = 0; uint32_t prevEnd = 0; int local = 0;                      // local: advances of the counter inside this chromosome so far
            std::vector<Row>& rows = rowsC[c]; rows.reserve((size_t)(S.off[c + 1] - S.off[c]));
            for (int64_t b = S.off[c]; b < S.off[c + 1]; b++) {
                uint32_t st = S.start[b], en = S.end[b];
                bool newSeg = std::binary_search(starts.begin(), starts.end(), st);
                if (ex) { while (exIdx < ex->size() && (int64_t)(*ex)[exIdx].second < (int64_t)prevEnd) exIdx++;
                    if (exIdx < ex->size()) { int mid = ((*ex)[exIdx].first + (*ex)[exIdx].second) / 2; if ((int64_t)prevEnd < mid && (int64_t)en >= mid) newSeg = true; } }
                if (prevEnd > 0 && maxInterBinDist >= 0 && (int64_t)prevEnd + maxInterBinDist < (int64_t)st && !newSeg) newSeg = true;
                if (!newSeg && pl) {                                          // SegmentationResultsProcessor.cs:117-128
                    const int u = is_uniform_reference_ploidy(*pl, prevEnd > 0 ? (int)prevEnd : 1, (int)en);
                    if (u < 0) { int expect = -1; badChrom.compare_exchange_strong(expect, (int)c); return; }
                    if (!u) newSeg = true;
                }
                if (newSeg) local++;
                rows.push_back({st, en, S.cov[b], local});
                prevEnd = en;
            }
            incC[c] = local;
            // bins of a segment are written ordered by start (SegmentWithBins.Bins, Models/SegmentWithBins.cs:11-14); OrderBy is stable
            for (size_t g0 = 0; g0 < rows.size();) { size_t g1 = g0; while (g1 < rows.size() && rows[g1].id == rows[g0].id) g1++;
                if (!std::is_sorted(rows.begin() + g0, rows.begin() + g1, [](const Row& x, const Row& y) { return x.s < y.s; }))
                    std::stable_sort(rows.begin() + g0, rows.begin() + g1, [](const Row& x, const Row& y) { return x.s < y.s; });
                g0 = g1; }
        });
        if (badChrom.load() >= 0) { fprintf(stderr, "CanvasPartition: reference ploidy outside 0..4 on %s (the reference throws IndexOutOfRangeException)\n", S.chromNames[(size_t)badChrom.load()].c_str()); return 1; }
        {
            std::vector<size_t> at(nC + 1, 0); std::vector<int> base(nC, -1); int segmentNum = -1;
            for (size_t c = 0; c < nC; c++) { at[c + 1] = at[c] + rowsC[c].size(); base[c] = segmentNum; segmentNum += incC[c]; }
            outRows.resize(at[nC]);
            parallel_for((int64_t)nC, [&](int64_t ci) { const size_t c = (size_t)ci; size_t o = at[c]; for (auto& r : rowsC[c]) outRows[o++] = OutRow{r.s, r.e, r.cov, base[c] + r.id, (int)c}; std::vector<Row>().swap(rowsC[c]); });
        }
        if (!write_gz_rows(outFiles[s], (int64_t)outRows.size(), [&](int64_t i, std::string& o) { const OutRow& r = outRows[(size_t)i];
                o += S.chromNames[(size_t)r.chrom]; o.push_back('\t'); append_uint(o, r.s); o.push_back('\t'); append_uint(o, r.e); o.push_back('\t'); o += format_g(r.cov, 15); o.push_back('\t'); append_int(o, r.id); }))
            { fprintf(stderr, "cannot write %s\n", outFiles[s].c_str()); return 1; }
    }
    printf("CanvasPartition results written out\n");
    ph.mark("write");
    return finish(ph, 0);
}
