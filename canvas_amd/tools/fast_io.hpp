// Host-side file I/O of the drop-in tools on several threads (included by tool_common.hpp).  The device work of a WGS sample is a few milliseconds; with one thread
// formatting and compressing 4.8 M rows (5 s per file), gzgets + strtod line by line (1.1 s) and fgets + string appends over a 3 GB FASTA (2.5 s) a tool run was
// 6-12 s of file handling.  Formats are unchanged byte for byte where a reader can tell: the gzip files are ONE member each (blocks compressed independently and
// joined at sync-flush boundaries, the way pigz -i does), so any GzipReader that reads the reference's files reads these.
#pragma once
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace tool {

static inline int io_threads() { const char* e = getenv("CANVAS_TOOL_THREADS"); if (e && atoi(e) > 0) return atoi(e); const unsigned hw = std::thread::hardware_concurrency(); return (int)std::max(1u, std::min(hw ? hw : 1u, 32u)); }
template <class Fn>
static void parallel_for(int64_t n, const Fn& fn, int threads = 0) {
    const int nt = (int)std::min<int64_t>(n, threads > 0 ? threads : io_threads());
    if (nt <= 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&]() { for (int64_t i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& t : th) t.join();
}

// ---- a file mapped read-only (the chunk reader below)
struct MappedFileRO {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY); if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size; if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) { p = nullptr; return false; }
        p = (const char*)m; return true;
    }
    ~MappedFileRO() { if (p && n) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};
// ---- writing: rows [0, nrows) formatted by fmt(i, out) (appends the row WITHOUT the newline), chunks of rows compressed on several threads, one gzip member
// The header carries an index of the chunks in a gzip "extra" subfield (RFC 1952 FEXTRA, id "CV": chunk count, then {compressed bytes, text bytes} per chunk — what BGZF
// does per block): every reader of gzip skips it, and read_gz_all below inflates the chunks of a file written here on several threads.
static bool write_gz_rows(const std::string& path, int64_t nrows, const std::function<void(int64_t, std::string&)>& fmt, int64_t chunkRows = 8192) {
    // deflate level 2 by default: 4x the speed of level 6 on bin rows for files 11 % larger (CANVAS_TOOL_GZIP_LEVEL=6 gives zlib's default; any level inflates to the same rows)
    static const int level = [] { const char* e = getenv("CANVAS_TOOL_GZIP_LEVEL"); const int v = e ? atoi(e) : 2; return v >= 0 && v <= 9 ? v : 2; }();
    chunkRows = std::max<int64_t>(chunkRows, nrows / 8000 + 1);             // the index has to fit one subfield (64 KB)
    const int64_t nchunks = std::max<int64_t>(1, (nrows + chunkRows - 1) / chunkRows);
    struct Chunk { std::vector<unsigned char> z; uLong crc = 0; uint64_t len = 0; bool ok = true; };
    std::vector<Chunk> chunks((size_t)nchunks);
    parallel_for(nchunks, [&](int64_t c) {
        Chunk& C = chunks[(size_t)c];
        std::string text; text.reserve((size_t)chunkRows * 40);
        const int64_t a = c * chunkRows, b = std::min(nrows, a + chunkRows);
        for (int64_t i = a; i < b; i++) { fmt(i, text); text.push_back('\n'); }
        C.len = text.size(); C.crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef*)text.data(), (uInt)text.size());
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { C.ok = false; return; }
        // deflateBound covers Z_FINISH only: room for the stored-block overhead of an incompressible chunk and the flush marker on top, and the flush is complete only
        // when it returns with output space LEFT (zlib: Z_OK with avail_out == 0 means more is pending) — otherwise the buffer grows and the call is repeated
        C.z.resize(deflateBound(&zs, (uLong)text.size()) + text.size() / 1000 + 512);
        zs.next_in = (Bytef*)text.data(); zs.avail_in = (uInt)text.size(); zs.next_out = C.z.data(); zs.avail_out = (uInt)C.z.size();
        const int last = c == nchunks - 1;
        for (;;) {
            const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);      // a sync flush ends on a byte boundary without the final-block bit: the next chunk's blocks follow
            if (last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0)) break;
            if ((rc == Z_OK || rc == Z_BUF_ERROR) && zs.avail_out == 0) {     // output full: more room, again
                const size_t used = C.z.size(); C.z.resize(used * 2 + 4096); zs.next_out = C.z.data() + used; zs.avail_out = (uInt)(C.z.size() - used);
                continue;
            }
            C.ok = false; break;
        }
        C.z.resize(C.z.size() - zs.avail_out);
        deflateEnd(&zs);
    });
    FILE* f = fopen(path.c_str(), "wb"); if (!f) return false;
    bool indexed = nchunks <= 8000;
    for (auto& C : chunks) if (C.z.size() > 0xFFFFFFFFull || C.len > 0xFFFFFFFFull) indexed = false;
    const unsigned char hdr[10] = {0x1f, 0x8b, 8, (unsigned char)(indexed ? 4 : 0), 0, 0, 0, 0, 0, 0xff};
    bool ok = fwrite(hdr, 1, 10, f) == 10;
    if (indexed) {
        std::vector<unsigned char> ex; auto put16 = [&](unsigned v) { ex.push_back((unsigned char)(v & 0xFF)); ex.push_back((unsigned char)(v >> 8)); };
        auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) ex.push_back((unsigned char)((v >> (8 * i)) & 0xFF)); };
        const unsigned dlen = (unsigned)(4 + 8 * nchunks);
        put16(dlen + 4); ex.push_back('C'); ex.push_back('V'); put16(dlen); put32((uint32_t)nchunks);
        for (auto& C : chunks) { put32((uint32_t)C.z.size()); put32((uint32_t)C.len); }
        ok = ok && fwrite(ex.data(), 1, ex.size(), f) == ex.size();
    }
    uLong crc = crc32(0L, Z_NULL, 0); uint64_t total = 0;
    for (auto& C : chunks) { ok = ok && C.ok && fwrite(C.z.data(), 1, C.z.size(), f) == C.z.size(); crc = crc32_combine(crc, C.crc, (z_off_t)C.len); total += C.len; }
    unsigned char tr[8]; for (int i = 0; i < 4; i++) { tr[i] = (unsigned char)((crc >> (8 * i)) & 0xFF); tr[4 + i] = (unsigned char)((total >> (8 * i)) & 0xFF); }
    ok = ok && fwrite(tr, 1, 8, f) == 8;
    return fclose(f) == 0 && ok;
}
static inline void append_uint(std::string& out, unsigned long long v) { char b[24]; int n = 0; do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) out.push_back(b[--n]); }
static inline void append_int(std::string& out, long long v) { if (v < 0) { out.push_back('-'); append_uint(out, (unsigned long long)(-(v + 1)) + 1ull); } else append_uint(out, (unsigned long long)v); }

// ---- reading gzip text.  A file that carries the chunk index of write_gz_rows is inflated chunk by chunk on several threads (every chunk is a raw deflate stream of its
// own: the writer compressed them independently) and its CRC checked from the chunks' CRCs; any other gzip file is inflated by one thread (a deflate stream is
// sequential).  The lines are then parsed on several threads.
static bool read_gz_indexed(const std::string& path, std::string& data) {
    MappedFileRO mf; if (!mf.open(path) || mf.n < 18 + 10) return false;
    const unsigned char* p = (const unsigned char*)mf.p;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return false;                 // exactly the header the writer produces: FEXTRA and nothing else
    const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (12 + xlen + 8 > mf.n || xlen < 8 || p[12] != 'C' || p[13] != 'V') return false;
    const size_t dlen = (size_t)p[14] | ((size_t)p[15] << 8);
    if (dlen + 4 != xlen) return false;
    auto get32 = [&](size_t at) { return (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16) | ((uint32_t)p[at + 3] << 24); };
    const uint32_t nchunks = get32(16);
    if ((size_t)nchunks * 8 + 4 != dlen || nchunks == 0) return false;
    std::vector<size_t> zoff((size_t)nchunks + 1), toff((size_t)nchunks + 1);
    zoff[0] = 12 + xlen; toff[0] = 0;
    for (uint32_t c = 0; c < nchunks; c++) { zoff[c + 1] = zoff[c] + get32(20 + 8 * (size_t)c); toff[c + 1] = toff[c] + get32(24 + 8 * (size_t)c); }
    if (zoff[nchunks] + 8 != mf.n) return false;
    if ((uint32_t)(toff[nchunks] & 0xFFFFFFFFull) != get32(mf.n - 4)) return false;
    data.resize(toff[nchunks]);
    std::vector<uLong> crcs((size_t)nchunks); std::atomic<bool> ok(true);
    parallel_for((int64_t)nchunks, [&](int64_t c) {
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { ok = false; return; }
        zs.next_in = (Bytef*)(p + zoff[(size_t)c]); zs.avail_in = (uInt)(zoff[(size_t)c + 1] - zoff[(size_t)c]);
        zs.next_out = (Bytef*)&data[toff[(size_t)c]]; zs.avail_out = (uInt)(toff[(size_t)c + 1] - toff[(size_t)c]);
        const int rc = inflate(&zs, Z_FINISH);               // (every chunk but the last ends at a sync flush, not at a final block: Z_BUF_ERROR / Z_OK with everything consumed is its normal end)
        if (!(rc == Z_STREAM_END || rc == Z_OK || rc == Z_BUF_ERROR) || zs.avail_in != 0 || zs.avail_out != 0) ok = false;
        inflateEnd(&zs);
        crcs[(size_t)c] = crc32(crc32(0L, Z_NULL, 0), (const Bytef*)&data[toff[(size_t)c]], (uInt)(toff[(size_t)c + 1] - toff[(size_t)c]));
    });
    if (!ok) return false;
    uLong crc = crc32(0L, Z_NULL, 0);
    for (uint32_t c = 0; c < nchunks; c++) crc = crc32_combine(crc, crcs[c], (z_off_t)(toff[c + 1] - toff[c]));
    return (uint32_t)crc == get32(mf.n - 8);
}
static bool read_gz_all(const std::string& path, std::string& data) {
    if (!getenv("CANVAS_TOOL_SERIAL_GUNZIP") && read_gz_indexed(path, data)) return true;
    gzFile f = gzopen(path.c_str(), "rb"); if (!f) return false;
    gzbuffer(f, 1 << 20);
    data.clear();
    std::vector<char> buf(16 << 20);
    for (;;) { const int k = gzread(f, buf.data(), (unsigned)buf.size()); if (k < 0) { gzclose(f); return false; } if (k == 0) break; data.append(buf.data(), (size_t)k); }
    gzclose(f); return true;
}
// a tab-separated row of the intermediate files: chromosome, start, stop, value, [gc]
struct TextRow { int32_t chrLocal; uint32_t start, stop; double value; int32_t gc; int32_t nfields; };
struct TextRows { std::vector<std::string> chromNames; std::vector<int32_t> chr; std::vector<uint32_t> start, stop; std::vector<double> value; std::vector<int32_t> gc, nfields; };
// every line with at least minFields fields, in file order; chromosome indices in order of first appearance; value = strtod of field 3, gc = atoi of field 4 (-1 if absent)
static bool read_text_rows(const std::string& path, int minFields, TextRows& out) {
    std::string data; if (!read_gz_all(path, data)) return false;
    const int nt = io_threads();
    const size_t n = data.size();
    std::vector<size_t> cut((size_t)nt + 1, n); cut[0] = 0;
    for (int t = 1; t < nt; t++) { size_t p = n / nt * t; if (p < cut[t - 1]) p = cut[t - 1]; while (p < n && data[p] != '\n') p++; cut[t] = p < n ? p + 1 : n; }
    struct Part { std::vector<std::string> names; std::vector<TextRow> rows; };
    std::vector<Part> parts((size_t)nt);
    parallel_for(nt, [&](int64_t t) {
        Part& P = parts[(size_t)t];
        const char* p = data.data() + cut[t]; const char* end = data.data() + cut[t + 1];
        std::string lastName; int lastIdx = -1;
        while (p < end) {
            const char* e = (const char*)memchr(p, '\n', (size_t)(end - p)); if (!e) e = end;
            const char* le = e; while (le > p && (le[-1] == '\r')) le--;
            const char* fld[6]; int nf = 0; const char* q = p; fld[nf++] = p;
            while (q < le && nf < 6) { if (*q == '\t') fld[nf++] = q + 1; q++; }
            int total = nf; for (; q < le; q++) if (*q == '\t') total++;
            if (total >= minFields && total >= 4) {
                const size_t nameLen = (size_t)(fld[1] - 1 - fld[0]);
                if (lastIdx < 0 || lastName.size() != nameLen || memcmp(lastName.data(), fld[0], nameLen) != 0) {
                    lastName.assign(fld[0], nameLen); lastIdx = -1;
                    for (size_t k = 0; k < P.names.size(); k++) if (P.names[k] == lastName) lastIdx = (int)k;
                    if (lastIdx < 0) { lastIdx = (int)P.names.size(); P.names.push_back(lastName); }
                }
                TextRow r; r.chrLocal = lastIdx; r.nfields = total;
                char tmp[64];
                auto field = [&](int k) -> const char* {      // field k as a C string of its own (the number parsers skip white space: they must not run into the next field or line)
                    const char* fe = (k + 1 < nf) ? fld[k + 1] - 1 : le;
                    if (k + 1 >= nf && total > nf) { fe = fld[k]; while (fe < le && *fe != '\t') fe++; }
                    const size_t vl = std::min<size_t>(63, (size_t)(fe - fld[k])); memcpy(tmp, fld[k], vl); tmp[vl] = 0; return tmp; };
                r.start = (uint32_t)strtoul(field(1), nullptr, 10); r.stop = (uint32_t)strtoul(field(2), nullptr, 10);
                r.value = strtod(field(3), nullptr);
                r.gc = total > 4 ? atoi(field(4)) : -1;
                P.rows.push_back(r);
            }
            p = e < end ? e + 1 : end;
        }
    });
    out = TextRows();
    size_t totalRows = 0; for (auto& P : parts) totalRows += P.rows.size();
    out.chr.resize(totalRows); out.start.resize(totalRows); out.stop.resize(totalRows); out.value.resize(totalRows); out.gc.resize(totalRows); out.nfields.resize(totalRows);
    std::vector<std::vector<int>> remap((size_t)nt); std::vector<size_t> base((size_t)nt + 1, 0);
    for (int t = 0; t < nt; t++) {
        for (auto& nm : parts[(size_t)t].names) { int g = -1; for (size_t k = 0; k < out.chromNames.size(); k++) if (out.chromNames[k] == nm) g = (int)k; if (g < 0) { g = (int)out.chromNames.size(); out.chromNames.push_back(nm); } remap[(size_t)t].push_back(g); }
        base[(size_t)t + 1] = base[(size_t)t] + parts[(size_t)t].rows.size();
    }
    parallel_for(nt, [&](int64_t t) {
        const Part& P = parts[(size_t)t]; size_t at = base[(size_t)t];
        for (const TextRow& r : P.rows) { out.chr[at] = remap[(size_t)t][(size_t)r.chrLocal]; out.start[at] = r.start; out.stop[at] = r.stop; out.value[at] = r.value; out.gc[at] = r.gc; out.nfields[at] = r.nfields; at++; }
    });
    return true;
}

// ---- a file mapped read-only
struct MappedFile {
    const char* p = nullptr; size_t n = 0; int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY); if (fd < 0) return false;
        struct stat st; if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size; if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0); if (m == MAP_FAILED) return false;
        p = (const char*)m; madvise(m, n, MADV_SEQUENTIAL); return true;
    }
    ~MappedFile() { if (p && n) munmap((void*)p, n); if (fd >= 0) ::close(fd); }
};

}  // namespace tool
