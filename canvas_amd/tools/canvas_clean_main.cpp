// Drop-in CanvasClean executable on top of the C ABI: same CLI and file formats as CanvasClean.Main (CanvasClean/CanvasClean.cs:415-533).
//   CanvasClean -i S.binned -o S.cleaned [-g] [-s] [-r] [--local-sd-metric-file F] [-w N] [-m MedianByGC]
// Exit codes follow the reference: help / missing -i,-o -> 0 (:455-465); missing input file -> 1 (:468-472); unknown arguments throw.
#include "tool_common.hpp"
using namespace tool;

int main(int argc, char** argv) {
    printf(">>>Command-line arguments:\n"); for (int i = 1; i < argc; i++) printf("%s ", argv[i]); printf("\n");   // Utilities.LogCommandLine
    std::vector<Opt> opts = {{"i", "infile", true}, {"o", "outfile", true}, {"g", "gcnorm", false}, {"s", "filtsize", false}, {"r", "outliers", false},
                             {"", "local-sd-metric-file", true}, {"t", "manifest", true}, {"w", "weightedmedian", true}, {"m", "mode", true}, {"h", "help", false}};
    Parsed a = parse(argc, argv, opts);
    if (!a.extra.empty()) { fprintf(stderr, "Unknown arguments: %s\n", a.extra[0].c_str()); return 2; }
    auto help = []() { printf("Usage: CanvasClean.exe [OPTIONS]+\nCorrect bin counts based on genomic parameters\n\nOptions:\n  -i, --infile=VALUE  -o, --outfile=VALUE  -g, --gcnorm  -s, --filtsize  -r, --outliers\n"
                              "      --local-sd-metric-file=VALUE  -t, --manifest=VALUE  -w, --weightedmedian=VALUE  -m, --mode=VALUE  -h, --help\n"); };
    if (a.has("help") || !a.has("infile") || !a.has("outfile")) { help(); return 0; }
    const std::string inFile = a.get("infile"), outFile = a.get("outfile");
    if (!file_exists(inFile)) { printf("CanvasClean.exe: File %s does not exist! Exiting.\n", inFile.c_str()); return 1; }
    std::string mode = a.get("mode", "medianbygc"); for (auto& c : mode) c = (char)tolower(c);
    uint32_t flags = (a.has("gcnorm") ? CANVAS_CLEAN_GCNORM : 0) | (a.has("filtsize") ? CANVAS_CLEAN_FILTSIZE : 0) | (a.has("outliers") ? CANVAS_CLEAN_OUTLIERS : 0) |
                     (a.has("local-sd-metric-file") ? CANVAS_CLEAN_LOCALSD : 0);
    if (mode == "loess") flags |= CANVAS_CLEAN_LOESS; else if (mode != "medianbygc") { fprintf(stderr, "Invalid CanvasClean mode '%s'\n", mode.c_str()); return 2; }
    if (a.has("manifest")) { fprintf(stderr, "CanvasClean (MI355X): -t/--manifest is not supported by this build\n"); return 1; }
    int minBins = a.has("weightedmedian") ? atoi(a.get("weightedmedian").c_str()) : 100;

    Phases ph("CanvasClean");
    AsyncCtx actx;                                              // the context comes up while the file is read
    // CanvasIO.ReadFromTextFile (CanvasCommon/IO.cs:26-52)
    std::vector<std::string> chromNames;
    std::vector<int32_t> chr, start, stop, gc; std::vector<float> count;
    {   // rows with fewer than five fields are skipped; chromosome indices in order of first appearance (parsed on several threads: fast_io.hpp)
        TextRows rows;
        if (!read_text_rows(inFile, 5, rows)) { printf("CanvasClean.exe: cannot read %s\n", inFile.c_str()); return 1; }
        const size_t m = rows.chr.size();
        chromNames = rows.chromNames; chr = std::move(rows.chr); start.resize(m); stop.resize(m); gc = std::move(rows.gc); count.resize(m);
        parallel_for((int64_t)((m + 65535) / 65536), [&](int64_t blk) { const size_t a = (size_t)blk * 65536, b = std::min(m, a + 65536);
            for (size_t i = a; i < b; i++) { start[i] = (int32_t)rows.start[i]; stop[i] = (int32_t)rows.stop[i]; count[i] = (float)rows.value[i]; } });
    }
    const int64_t n = (int64_t)chr.size(); const int nchr = (int)chromNames.size();
    std::vector<uint8_t> isAuto(nchr > 0 ? nchr : 1, 0), isY(nchr > 0 ? nchr : 1, 0);
    for (int c = 0; c < nchr; c++) { isAuto[c] = is_autosome(chromNames[c]); std::string lo = chromNames[c]; for (auto& ch : lo) ch = (char)tolower(ch); isY[c] = (lo == "chry" || lo == "y"); }   // LoessGCNormalizer.cs:49-50
    // chromosome indices must be non-decreasing for the library (bins grouped by chromosome in file order): first-appearance indexing gives that
    ph.mark("read");
    int64_t nOut = n; double localSd = -1.0;
    if (n > 0) {
        canvas_ctx* ctx = actx.get();
        if (!ctx) { fprintf(stderr, "CanvasClean (MI355X): no usable GPU (this build has no CPU fallback)\n"); return 1; }
        { Dev dChr(ctx, n * 4), dStart(ctx, n * 4), dStop(ctx, n * 4), dCount(ctx, n * 4), dGc(ctx, n * 4);
          TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dChr.p, chr.data(), n * 4)); TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dStart.p, start.data(), n * 4));
          TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dStop.p, stop.data(), n * 4)); TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dCount.p, count.data(), n * 4));
          TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dGc.p, gc.data(), n * 4));
          int32_t info[8];
          TOOL_TRY(ctx, canvas_clean2(ctx, n, dChr.as<int32_t>(), dStart.as<int32_t>(), dStop.as<int32_t>(), dCount.as<float>(), dGc.as<int32_t>(), nchr, isAuto.data(), isY.data(), flags, minBins,
                                    &localSd, &nOut, info));
          TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, chr.data(), dChr.p, nOut * 4)); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, start.data(), dStart.p, nOut * 4));
          TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, stop.data(), dStop.p, nOut * 4)); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, count.data(), dCount.p, nOut * 4));
          TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, gc.data(), dGc.p, nOut * 4)); }
        if (getenv("CANVAS_TOOL_FULL_TEARDOWN")) canvas_destroy(ctx);
    }
    ph.mark("device");
    // CanvasIO.WriteLocalSdMetricToTextFile (IO.cs:83-98) — only when the metric was computed (>= 50000 bins, CanvasClean.cs:483-494)
    if (a.has("local-sd-metric-file") && localSd >= 0) { FILE* f = fopen(a.get("local-sd-metric-file").c_str(), "wb"); if (f) { fprintf(f, "#localSD\t%s\n", format_g(localSd, 15).c_str()); fclose(f); } }
    // CanvasIO.WriteToTextFile (IO.cs:15-24)
    if (!write_gz_rows(outFile, nOut, [&](int64_t i, std::string& o) {
            o += chromNames[chr[i]]; o.push_back('\t'); append_int(o, start[i]); o.push_back('\t'); append_int(o, stop[i]); o.push_back('\t'); o += format_f2(count[i]); o.push_back('\t'); append_int(o, gc[i]); }))
        { fprintf(stderr, "cannot write %s\n", outFile.c_str()); return 1; }
    ph.mark("write");
    return finish(ph, 0);
}
