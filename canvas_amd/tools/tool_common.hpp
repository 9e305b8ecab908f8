// Shared host code of the drop-in tool drivers (CanvasClean / CanvasPartition): option parsing in the reference's NDesk OptionSet
// style, gzip text I/O of the intermediate files (CanvasCommon/IO.cs), .NET Core 2.x number formatting, and the
// IsAutosome assumption (Isas.SequencingFiles is not in /root/reference: "optional chr prefix + integer").
// The drivers use ONLY the C ABI of include/canvas_hip.h (what the C# hosts would P/Invoke).
#pragma once
#include <zlib.h>
#include <time.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <atomic>
#include <set>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#include <sys/mman.h>
#include <mutex>
#include <new>
#include "../../include/canvas_hip.h"
#include "fast_io.hpp"

#ifndef CANVAS_SRC_HASH
#define CANVAS_SRC_HASH "unhashed-build-0000000000000000"
#endif
__attribute__((used)) static const char tool_src_hash_marker[] = "CANVAS_SRC_HASH=" CANVAS_SRC_HASH;

// ---- big host buffers (bases, hits, fragment lengths: gigabytes per run) come from 2 MB-aligned anonymous mappings advised to transparent huge pages: the box runs THP
// in "madvise" mode, and 4 KB pages cost a fault per page while a file is read plus 0.5 s of page freeing when the process leaves.  Everything below 8 MB is malloc's.
// (each tool is one translation unit: the replaced global operators live here.)  CANVAS_TOOL_NO_HUGEPAGES=1 keeps malloc for everything.
// This header DEFINES the global operators: it must be part of exactly one translation unit per executable.  A second inclusion is a duplicate-symbol error at link time
// (the definition below has external linkage on purpose) instead of an ODR violation nobody sees.  Memory released by a library that carries its own allocator never
// reaches these operators (libcanvas_hip.so hands out no ownership of host memory); blocks of 8 MB and more that such a library allocates for itself are its own business.
extern "C" { int canvas_tool_common_hpp_is_in_one_translation_unit = 1; }
namespace tool { struct BigAllocs { std::mutex m; std::map<void*, size_t> len; }; static inline BigAllocs& big_allocs() { static BigAllocs* b = new BigAllocs; return *b; } }
void* operator new(size_t n) {
    static const bool huge = !getenv("CANVAS_TOOL_NO_HUGEPAGES");
    const size_t H = (size_t)2 << 20;
    if (huge && n >= ((size_t)8 << 20)) {
        const size_t len = (n + H - 1) & ~(H - 1);
        char* p = (char*)mmap(nullptr, len + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p != (char*)MAP_FAILED) {
            char* q = (char*)(((uintptr_t)p + H - 1) & ~(uintptr_t)(H - 1));
            if (q > p) munmap(p, (size_t)(q - p));
            if (q + len < p + len + H) munmap(q + len, (size_t)((p + len + H) - (q + len)));
            madvise(q, len, MADV_HUGEPAGE);
            { auto& B = tool::big_allocs(); std::lock_guard<std::mutex> g(B.m); B.len[q] = len; }
            return q;
        }
    }
    void* r = malloc(n ? n : 1); if (!r) throw std::bad_alloc(); return r;
}
void operator delete(void* p) noexcept {
    if (!p) return;
    if (((uintptr_t)p & (((uintptr_t)2 << 20) - 1)) == 0) {
        auto& B = tool::big_allocs(); size_t len = 0;
        { std::lock_guard<std::mutex> g(B.m); auto it = B.len.find(p); if (it != B.len.end()) { len = it->second; B.len.erase(it); } }
        if (len) { munmap(p, len); return; }
    }
    free(p);
}
void operator delete(void* p, size_t) noexcept { operator delete(p); }

namespace tool {

// ---- NDesk.OptionSet-like parsing: -x value, -x=value, --long value, --long=value, /x value; flags without value
struct Opt { std::string shortName, longName; bool takesValue; };
struct Parsed { std::multimap<std::string, std::string> values; std::vector<std::string> extra; bool has(const std::string& k) const { return values.count(k) > 0; }
    std::string get(const std::string& k, const std::string& d = "") const { auto it = values.find(k); return it == values.end() ? d : it->second; }
    std::vector<std::string> all(const std::string& k) const { std::vector<std::string> r; auto rg = values.equal_range(k); for (auto it = rg.first; it != rg.second; ++it) r.push_back(it->second); return r; } };
static Parsed parse(int argc, char** argv, const std::vector<Opt>& opts) {
    Parsed p;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], name, val; bool hasVal = false;
        if (a.rfind("--", 0) == 0) name = a.substr(2); else if (a.size() > 1 && (a[0] == '-' || a[0] == '/')) name = a.substr(1); else { p.extra.push_back(a); continue; }
        size_t eq = name.find_first_of("=:");
        if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); hasVal = true; }
        const Opt* o = nullptr;
        for (auto& c : opts) if (c.shortName == name || c.longName == name) o = &c;
        if (!o) { p.extra.push_back(a); continue; }
        std::string key = o->longName.empty() ? o->shortName : o->longName;
        if (o->takesValue) { if (!hasVal) { if (i + 1 < argc) val = argv[++i]; else { p.extra.push_back(a); continue; } } p.values.insert({key, val}); }
        else p.values.insert({key, "1"});
    }
    return p;
}

// ---- gzip text
struct GzReader { gzFile f; explicit GzReader(const std::string& path) { f = gzopen(path.c_str(), "rb"); } ~GzReader() { if (f) gzclose(f); }
    bool ok() const { return f != nullptr; }
    bool line(std::string& out) { out.clear(); char buf[1 << 16]; bool any = false;
        while (gzgets(f, buf, sizeof buf)) { any = true; out += buf; if (!out.empty() && out.back() == '\n') break; }
        while (!out.empty() && (out.back() == '\n' || out.back() == '\r')) out.pop_back();
        return any; } };
static std::atomic<int> g_open_writers{0};      // output objects that still hold buffered data: finish() leaves without unwinding only when there is none
struct GzWriter { gzFile f; explicit GzWriter(const std::string& path) { f = gzopen(path.c_str(), "wb"); if (f) g_open_writers++; } ~GzWriter() { close(); }
    bool ok() const { return f != nullptr; } void line(const std::string& s) { gzwrite(f, s.data(), (unsigned)s.size()); gzputc(f, '\n'); }
    void close() { if (f) { gzclose(f); f = nullptr; g_open_writers--; } } };
static std::vector<std::string> split_tab(const std::string& s) { std::vector<std::string> r; size_t a = 0; for (;;) { size_t b = s.find('\t', a); r.push_back(s.substr(a, b == std::string::npos ? b : b - a)); if (b == std::string::npos) break; a = b + 1; } return r; }
static bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) { fclose(f); return true; } return false; }

// ---- .NET Core 2.x formatting: float "F2" (7 significant digits, then half-up at 2 decimals) and double "G15"
static void sig_digits(double v, int prec, std::string& digits, int& scale) {
    char buf[64]; snprintf(buf, sizeof buf, "%.*e", prec - 1, std::fabs(v));
    digits.clear(); const char* p = buf;
    for (; *p && *p != 'e'; p++) if (*p >= '0' && *p <= '9') digits.push_back(*p);
    scale = atoi(p + 1) + 1;
    while (!digits.empty() && digits.back() == '0') digits.pop_back();
    if (digits.empty()) scale = 0;
}
static std::string format_f2(float v) {
    if (std::isnan(v)) return "NaN"; if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string d; int scale; sig_digits((double)v, 7, d, scale);
    int pos = scale + 2;
    if (pos < 0) d.clear();
    else if (pos < (int)d.size()) { bool up = d[pos] >= '5'; d.resize(pos); if (up) { int i = pos - 1; while (i >= 0 && d[i] == '9') { d[i] = '0'; i--; } if (i >= 0) d[i]++; else { d.insert(d.begin(), '1'); scale++; } } }
    std::string ip, fp;
    for (int i = 0; i < scale; i++) ip.push_back(i < (int)d.size() ? d[i] : '0');
    if (ip.empty()) ip = "0";
    for (int i = 0; i < 2; i++) { int k = scale + i; fp.push_back(k >= 0 && k < (int)d.size() ? d[k] : '0'); }
    bool zero = true; for (char c : ip + fp) if (c != '0') zero = false;
    return std::string((std::signbit(v) && !zero) ? "-" : "") + ip + "." + fp;
}
static std::string format_g(double v, int prec) {
    if (std::isnan(v)) return "NaN"; if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string d; int scale; sig_digits(v, prec, d, scale);
    if (d.empty()) return "0";
    std::string out = std::signbit(v) ? "-" : ""; int e10 = scale - 1;
    if (e10 >= prec || e10 < -5) { out.push_back(d[0]); if (d.size() > 1) { out.push_back('.'); out += d.substr(1); } char eb[16]; snprintf(eb, sizeof eb, "E%c%02d", e10 < 0 ? '-' : '+', std::abs(e10)); return out + eb; }
    if (scale <= 0) return out + "0." + std::string(-scale, '0') + d;
    for (int i = 0; i < scale; i++) out.push_back(i < (int)d.size() ? d[i] : '0');
    if ((int)d.size() > scale) { out.push_back('.'); out += d.substr(scale); }
    return out;
}

static bool is_autosome(std::string name) {
    if (name.rfind("chr", 0) == 0) name = name.substr(3);
    if (name.empty()) return false;
    for (char c : name) if (c < '0' || c > '9') return false;
    return true;
}

// excluded intervals of a BED file (Utilities.LoadBedFile, CanvasCommon/Utilities.cs:793-829): chr -> (start, stop) in file order
static bool load_bed(const std::string& path, std::map<std::string, std::vector<std::pair<int, int>>>& out) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    char buf[1 << 14];
    while (fgets(buf, sizeof buf, f)) { std::string s(buf); while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back(); auto t = split_tab(s); if (t.size() < 3) continue; out[t[0]].push_back({atoi(t[1].c_str()), atoi(t[2].c_str())}); }
    fclose(f); return true;
}

struct Dev {      // tiny RAII around the ABI's device memory
    canvas_ctx* ctx; void* p = nullptr;
    Dev(canvas_ctx* c, int64_t bytes) : ctx(c) { p = canvas_device_malloc(c, bytes > 0 ? bytes : 1); }
    ~Dev() { if (p) canvas_device_free(ctx, p); }
    Dev(const Dev&) = delete; Dev& operator=(const Dev&) = delete;
    template <class T> T* as() { return (T*)p; }
};
// wall-clock phases of a tool run, printed as one JSON line on stderr when CANVAS_TOOL_TIMING is set (bench.py's `executables` leg: what part of a run is file I/O)
struct Phases {
    const char* tool; std::vector<std::pair<std::string, double>> v; double t0;
    static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    explicit Phases(const char* name) : tool(name), t0(now()) {}
    void mark(const char* phase) { v.push_back({phase, now()}); }
    bool reported = false;
    ~Phases() { report(); }
    void report() {
        if (reported || !getenv("CANVAS_TOOL_TIMING")) return;
        reported = true;
        std::string js = std::string("{\"tool\": \"") + tool + "\", \"phases\": {"; double prev = t0;
        for (size_t i = 0; i < v.size(); i++) { char b[96]; snprintf(b, sizeof b, "%s\"%s\": %.4f", i ? ", " : "", v[i].first.c_str(), v[i].second - prev); js += b; prev = v[i].second; }
        // (wall-clock stamps of the first and the last statement of main: what lies outside them — the loader in front, the kernel's teardown of the process behind — is the
        // difference to the caller's own clock around the process, tools/exe_probe.sh)
        struct timespec tr; clock_gettime(CLOCK_REALTIME, &tr); const double real = (double)tr.tv_sec + 1e-9 * (double)tr.tv_nsec;
        char b[160]; snprintf(b, sizeof b, "}, \"total\": %.4f, \"main_entered_unix\": %.4f, \"leaving_unix\": %.4f}", now() - t0, real - (now() - t0), real); js += b;
        fprintf(stderr, "%s\n", js.c_str());
        if (FILE* f = fopen("/proc/self/smaps_rollup", "r")) {      // how much of the process is resident, and how much of that in transparent huge pages (what leaving the process has to give back)
            char line[256]; std::string rss, thp;
            while (fgets(line, sizeof line, f)) { if (!strncmp(line, "Rss:", 4)) rss = line + 4; if (!strncmp(line, "AnonHugePages:", 14)) thp = line + 14; }
            fclose(f);
            auto trim = [](std::string v) { while (!v.empty() && (v.back() == '\n' || v.back() == ' ')) v.pop_back(); size_t a = v.find_first_not_of(' '); return a == std::string::npos ? std::string() : v.substr(a); };
            fprintf(stderr, "[memory] resident %s, of it in transparent huge pages %s\n", trim(rss).c_str(), trim(thp).c_str());
        }
    }
};
// The GPU context is created on a helper thread while the main thread reads and parses the input files (HIP initialisation + the context's pinned buffers and streams
// take 0.15-0.3 s on a cold process; no file of a tool depends on it).  get() joins.
struct AsyncCtx {
    std::thread th; canvas_ctx* ctx = nullptr;
    // warm (optional): run on the helper thread behind canvas_create — e.g. a CBS call on a toy sample, so that the code objects are loaded and the launcher threads, streams
    // and engine buffers of the method exist by the time the real coverage has been read (a cold process paid 0.45 s in the device phase of -m CBS for 0.1 s of work)
    explicit AsyncCtx(std::function<void(canvas_ctx*)> warm = nullptr) { th = std::thread([this, warm] { ctx = canvas_create(0); if (ctx && !getenv("CANVAS_TOOL_KEEP_PINNING")) (void)canvas_set_one_shot(ctx, 1);      // (one OS process per sample: nothing amortises pinned staging)
                                                                                              if (ctx && warm && !getenv("CANVAS_TOOL_NO_WARMUP")) warm(ctx); }); }
    canvas_ctx* get() { if (th.joinable()) th.join(); return ctx; }
    ~AsyncCtx() { if (th.joinable()) th.join(); }
};
// End of a successful run: every output file is closed by now; the process leaves without unwinding (freeing gigabytes of host vectors, the context's device buffers and
// the HIP runtime's own teardown cost 0.1-0.6 s of wall time and change nothing on disk).  CANVAS_TOOL_FULL_TEARDOWN=1 returns through main instead.
static inline int finish(Phases& ph, int rc) {
    ph.report(); fflush(stdout); fflush(stderr);
    if (g_open_writers.load() != 0) return rc;      // (a writer that is still open flushes in its destructor: leave through main)
    // Leaving costs CanvasBin as much as its longest phase, and nothing done here changes it (measured, tools/exe_probe.sh + tools/exit_probe.sh: stamps of main's last
    // statement against the caller's clock): 0.43 s pass between _exit and the caller's wait returning, of which 0.26 s is the kernel giving back 7.6 GB of resident host
    // pages — 37 ms per GB, with or without a GPU in the process — and 0.07-0.09 s the driver's release of a process that has used the device.  Freeing the device buffers and
    // destroying the context first (4 ms) leaves the figure where it is, handing the pages to a forked child makes it worse (0.47 s), and unwinding main frees the same pages in
    // user space for the same price.  What would help is holding less anonymous memory: binning straight from the mapped .dat / FASTA files.
    if (!getenv("CANVAS_TOOL_FULL_TEARDOWN")) _exit(rc);
    return rc;
}
// (CANVAS_TOOL_FULL_TEARDOWN + CANVAS_TOOL_TIMING: where the unwinding of main spends its time — declared BEFORE the object whose destruction it reports)
struct ExitStamp { const char* what; explicit ExitStamp(const char* w) : what(w) {}
    ~ExitStamp() { if (getenv("CANVAS_TOOL_TIMING") && getenv("CANVAS_TOOL_FULL_TEARDOWN")) fprintf(stderr, "[teardown] %.4f s  %s\n", Phases::now(), what); } };
#define TOOL_TRY(ctx, expr) do { int32_t rc_ = (expr); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, canvas_last_error(ctx)); return 1; } } while (0)

}  // namespace tool
