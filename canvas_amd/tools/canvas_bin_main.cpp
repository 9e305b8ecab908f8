// Drop-in CanvasBin executable on top of the C ABI: CLI of CanvasBin (CanvasBin/Program.cs:12-200) and both of its phases
// (CanvasBin.Run, CanvasBin.cs:955-972):
//   CanvasBin -b S.bam -r kmer.fa -c chr1 -o chr1.dat -d 100 [-f filter.bed] [-p] [-m mode]            BAM -> per-chromosome intermediate
//   CanvasBin -b S.bam -r kmer.fa -i chr1.dat -i chr2.dat ... -o S.binned -d 100 [-z size] [-y] [-m]   intermediates -> S.binned
// Phase 1 keeps the reference's host work (FASTA and BAM parsing; BGZF inflate through zlib) and runs the per-base array
// preparation on the GPU (possible mask from the FASTA case, BED exclusion, hit screening).  Phase 2 is canvas_bin_sample /
// canvas_bin_sample_gcweighted.  The intermediate file is the reference's: protobuf-net's encoding of CanvasBin.IntermediateData (CanvasBin.cs:1037-1148,
// protobuf_dat.hpp) including the bit-order quirk of its writer / reader pair, so the C# CanvasBin -c and this CanvasBin -i (or the other way round) can be mixed.
// BAM flag semantics follow the SAM specification; Isas.SequencingFiles.BamReader is not part of /root/reference (parity unpinned):
// IsMainAlignment := neither secondary (0x100) nor supplementary (0x800).
//   CanvasBin -b S.bam -r kmer.fa -i ... -n bins.bed -o S.binned -d 100 [-m 0|3|5]                     predefined bins (canvas_bin_predefined)
//   CanvasBin -b S.bam -r genome.fa -n bins.bed -o S.binned -m Fragment -p                              FragmentBinner.Bin (FragmentBinner.cs:26-80): host code
// Fragment mode is a sequential dictionary algorithm over the BAM stream keyed by read name (the mate confirms or undoes what the first read of the pair did): its cost is
// BGZF inflation and string hashing, there is no data-parallel part, so it runs on the host exactly as in the reference.
// Not built (exit code 1 with a message): -t manifest (the Nextera manifest parser lives in Isas.Manifests, outside /root/reference).  -j behaves as in this version of the
// reference: the json file must exist and is never read, nothing is binned or written (CanvasBin.cs:936-944), exit code 0; with -i it is the reference's ArgumentException.
#include "tool_common.hpp"
#include "protobuf_dat.hpp"
#include <algorithm>
#include <memory>
#include <set>
using namespace tool;

// ---------------------------------------------------------------- FASTA (kmer.fa: upper case = start of a unique k-mer)
// An entry whose sequence is ONE line (FastaWriter-style kmer.fa files and the samples of bench.py) is a VIEW into the mapped file: nothing is copied and the process holds no
// anonymous copy of the reference (3.1 GB of a human genome: 0.1 s to copy on the host threads and another 0.11 s for the kernel to take back when the process leaves).  An entry
// folded into lines is copied with its line ends dropped, as before.
struct FastaEntry {
    std::string name, owned; const char* view = nullptr; size_t viewLen = 0; std::shared_ptr<MappedFile> keep;
    const char* data() const { return view ? view : owned.data(); }
    size_t size() const { return view ? viewLen : owned.size(); }
};
// the file is mapped, the entry headers are located in one scan and the entries are copied (line ends dropped) on several threads
static bool read_fasta(const std::string& path, const std::string* only, std::vector<FastaEntry>& out) {
    std::shared_ptr<MappedFile> mfp = std::make_shared<MappedFile>(); MappedFile& mf = *mfp; if (!mf.open(path)) return false;
    const char* p = mf.p; const size_t n = mf.n;
    struct Ent { size_t hdr, seq, end; std::string name; };
    std::vector<Ent> ents;
    {   // '>' at the start of a line (scanned in slices on several threads, then put in order)
        const int nt = io_threads();
        std::vector<std::vector<size_t>> found((size_t)nt);
        parallel_for(nt, [&](int64_t t) {
            size_t a = n / (size_t)nt * (size_t)t, b = t == nt - 1 ? n : n / (size_t)nt * (size_t)(t + 1);
            for (const char* q = p + a; q < p + b;) { q = (const char*)memchr(q, '>', (size_t)(p + b - q)); if (!q) break; if (q == p || q[-1] == '\n') found[(size_t)t].push_back((size_t)(q - p)); q++; }
        });
        for (auto& v : found) for (size_t h : v) { Ent e; e.hdr = h; e.seq = e.end = n; ents.push_back(e); }
    }
    for (size_t i = 0; i < ents.size(); i++) {
        const char* le = (const char*)memchr(p + ents[i].hdr, '\n', n - ents[i].hdr);
        const size_t lineEnd = le ? (size_t)(le - p) : n;
        std::string name(p + ents[i].hdr + 1, lineEnd - ents[i].hdr - 1);
        while (!name.empty() && (name.back() == '\r')) name.pop_back();
        const size_t sp = name.find_first_of(" \t"); if (sp != std::string::npos) name = name.substr(0, sp);
        ents[i].name = name; ents[i].seq = std::min(n, lineEnd + 1); ents[i].end = i + 1 < ents.size() ? ents[i + 1].hdr : n;
    }
    std::vector<size_t> keep;
    for (size_t i = 0; i < ents.size(); i++) if (!only || ents[i].name == *only) keep.push_back(i);
    const size_t base = out.size();
    out.resize(base + keep.size());
    parallel_for((int64_t)keep.size(), [&](int64_t k) {
        const Ent& e = ents[keep[(size_t)k]]; FastaEntry& fe = out[base + (size_t)k];
        fe.name = e.name; fe.owned.clear(); fe.view = nullptr; fe.viewLen = 0;
        {   // one line?  (nothing but line ends behind the first line end)
            const char* le = (const char*)memchr(p + e.seq, '\n', e.end - e.seq); const char* stop = le ? le : p + e.end;
            bool single = true; for (const char* q = stop; q < p + e.end; q++) if (*q != '\n' && *q != '\r') { single = false; break; }
            if (single && !getenv("CANVAS_TOOL_COPY_FASTA")) {
                const char* te = stop; while (te > p + e.seq && te[-1] == '\r') te--;
                fe.view = p + e.seq; fe.viewLen = (size_t)(te - (p + e.seq)); fe.keep = mfp;
                return;
            }
        }
        fe.owned.reserve(e.end - e.seq);
        for (const char* q = p + e.seq; q < p + e.end;) {
            const char* le = (const char*)memchr(q, '\n', (size_t)(p + e.end - q)); const char* stop = le ? le : p + e.end;
            const char* te = stop; while (te > q && te[-1] == '\r') te--;
            fe.owned.append(q, (size_t)(te - q));
            q = le ? le + 1 : p + e.end;
        }
    });
    return true;
}

// ---------------------------------------------------------------- BGZF / BAM / BAI
struct Bgzf {
    FILE* f = nullptr; std::vector<uint8_t> block; size_t pos = 0; int64_t blockAddr = 0; bool eof = false;
    bool open(const std::string& p) { f = fopen(p.c_str(), "rb"); return f != nullptr; }
    ~Bgzf() { if (f) fclose(f); }
    bool next_block() {
        blockAddr = ftello(f);
        uint8_t h[18];
        if (fread(h, 1, 18, f) != 18) { eof = true; return false; }
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        const int xlen = h[10] | (h[11] << 8);
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), h + 12, std::min(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != (size_t)(xlen - 6)) return false;
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) { int slen = extra[i + 2] | (extra[i + 3] << 8); if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = extra[i + 4] | (extra[i + 5] << 8); i += 4 + slen; }
        if (bsize < 0) return false;
        const int clen = bsize - xlen - 19;
        std::vector<uint8_t> comp(clen + 8);
        if (fread(comp.data(), 1, clen + 8, f) != (size_t)(clen + 8)) return false;
        const uint32_t isize = comp[clen + 4] | (comp[clen + 5] << 8) | (comp[clen + 6] << 16) | ((uint32_t)comp[clen + 7] << 24);
        block.resize(isize); pos = 0;
        if (isize == 0) return true;
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = comp.data(); zs.avail_in = clen; zs.next_out = block.data(); zs.avail_out = isize;
        int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
        return rc == Z_STREAM_END;
    }
    bool read(void* dst, size_t n) {
        uint8_t* d = (uint8_t*)dst;
        while (n) {
            if (pos >= block.size()) { do { if (!next_block()) return false; } while (block.empty()); }
            size_t k = std::min(n, block.size() - pos); memcpy(d, block.data() + pos, k); pos += k; d += k; n -= k;
        }
        return true;
    }
    bool seek_virtual(uint64_t voff) { if (fseeko(f, (off_t)(voff >> 16), SEEK_SET) != 0) return false; if (!next_block()) return false; pos = voff & 0xFFFF; return pos <= block.size(); }
};
// smallest virtual offset of a chunk of reference `ref` in the .bai (BamReader.Jump(ref, 0)); 0 = the reference has no reads
static bool bai_first_offset(const std::string& path, int ref, uint64_t& voff, bool& any) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
    auto rd = [&](void* p, size_t n) { return fread(p, 1, n, f) == n; };
    char magic[4]; int32_t nref;
    if (!rd(magic, 4) || memcmp(magic, "BAI\1", 4) != 0 || !rd(&nref, 4)) { fclose(f); return false; }
    any = false; voff = ~0ull;
    for (int r = 0; r < nref; r++) {
        int32_t nbin; if (!rd(&nbin, 4)) break;
        for (int b = 0; b < nbin; b++) {
            uint32_t bin; int32_t nchunk; if (!rd(&bin, 4) || !rd(&nchunk, 4)) { fclose(f); return false; }
            for (int c = 0; c < nchunk; c++) { uint64_t cb, ce; if (!rd(&cb, 8) || !rd(&ce, 8)) { fclose(f); return false; } if (r == ref && bin != 37450) { any = true; voff = std::min(voff, cb); } }
        }
        int32_t nintv; if (!rd(&nintv, 4)) break;
        if (fseeko(f, (off_t)nintv * 8, SEEK_CUR) != 0) break;
        if (r == ref) break;
    }
    fclose(f); return true;
}

// LoadObservedAlignmentsBAM (CanvasBin.cs:207-275)
static int load_bam(const std::string& bam, bool pairedEnd, const std::string& chrom, int mode, std::vector<uint8_t>& hits, std::vector<int16_t>& frag) {
    if (!file_exists(bam + ".bai")) { fprintf(stderr, "Fatal error: Bam index not found at %s.bai\n", bam.c_str()); return 1; }
    Bgzf z; if (!z.open(bam)) return 1;
    char magic[4]; int32_t ltext, nref;
    if (!z.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !z.read(&ltext, 4)) { fprintf(stderr, "CanvasBin: %s is not a BAM file\n", bam.c_str()); return 1; }
    { std::vector<char> t(ltext); if (ltext && !z.read(t.data(), ltext)) return 1; }
    if (!z.read(&nref, 4)) return 1;
    int desired = -1;
    for (int r = 0; r < nref; r++) { int32_t ln; if (!z.read(&ln, 4)) return 1; std::vector<char> nm(ln); int32_t lref; if (!z.read(nm.data(), ln) || !z.read(&lref, 4)) return 1; if (chrom == nm.data()) desired = r; }
    if (desired < 0) { fprintf(stderr, "Unable to retrieve the reference sequence index for %s in %s.\n", chrom.c_str(), bam.c_str()); return 1; }
    uint64_t voff; bool any;
    if (!bai_first_offset(bam + ".bai", desired, voff, any)) { fprintf(stderr, "CanvasBin: cannot read %s.bai\n", bam.c_str()); return 1; }
    if (!any) return 0;                                   // no reads for this chromosome: not an error (:231-235)
    if (!z.seek_virtual(voff)) return 1;
    long readCount = 0, kept = 0;
    std::vector<uint8_t> rec;
    for (;;) {
        int32_t bs; if (!z.read(&bs, 4)) break;
        rec.resize(bs); if (!z.read(rec.data(), bs)) break;
        readCount++;
        int32_t refID, pos, lseq, tlen; uint8_t lname; uint16_t ncig, flag;
        memcpy(&refID, &rec[0], 4); memcpy(&pos, &rec[4], 4); lname = rec[8]; memcpy(&ncig, &rec[12], 2); memcpy(&flag, &rec[14], 2); memcpy(&lseq, &rec[16], 4); memcpy(&tlen, &rec[28], 4);
        (void)lseq;
        if (flag & 0x4) continue;                          // !IsMapped
        if (flag & 0x200) continue;                        // IsFailedQC
        if (flag & 0x400) continue;                        // IsDuplicate
        if (flag & 0x10) continue;                         // IsReverseStrand
        if (flag & 0x900) continue;                        // !IsMainAlignment
        if (ncig == 0) continue;
        uint32_t c0; memcpy(&c0, &rec[32 + lname], 4);
        if ((c0 & 0xF) != 0 || (c0 >> 4) < 35) continue;   // must start with 35 bases of 'M'
        if (pairedEnd && !(flag & 0x2)) continue;          // IsProperPair
        if (refID != desired) break;
        if (refID == -1) continue;
        kept++;
        if (pos < 0 || (size_t)pos >= hits.size()) continue;   // the reference would throw IndexOutOfRange
        if (mode == CANVAS_MODE_BINARY) hits[pos] = 1; else hits[pos] = hits[pos] == 255 ? 255 : (uint8_t)(hits[pos] + 1);
        if (mode == CANVAS_MODE_GC_CONTENT_WEIGHTED) frag[pos] = (int16_t)std::max(std::min(32767, tlen), 0);
    }
    printf("Kept %ld of %ld total reads\n", kept, readCount);
    return 0;
}

// ---------------------------------------------------------------- intermediate file: see protobuf_dat.hpp
struct Inter { std::string name; int64_t len = 0; std::vector<uint64_t> maskWords; std::vector<uint8_t> hits; std::vector<int16_t> frag;
               const uint8_t* hitsView = nullptr; size_t hitsViewLen = 0; std::shared_ptr<pbdat::Mapped> keep;      // -i: the observed alignments inside the mapped intermediate file
               const uint8_t* hits_data() const { return hitsView ? hitsView : hits.data(); }
               size_t hits_size() const { return hitsView ? hitsViewLen : hits.size(); } };

// ---------------------------------------------------------------- predefined bins: Utilities.LoadBedFile(path, gcIndex: 3) (CanvasCommon/Utilities.cs:793-829)
struct PreBin { int start, stop, gc; float count; };
static bool load_predefined_bins(const std::string& path, std::map<std::string, std::vector<PreBin>>& out, std::vector<std::string>& chromOrder, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) { err = "cannot open " + path; return false; }
    char buf[1 << 14];
    while (fgets(buf, sizeof buf, f)) {
        std::string s(buf); while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        auto t = split_tab(s);
        if (t.size() < 3) { fclose(f); err = "malformed BED line: " + s; return false; }
        PreBin b{atoi(t[1].c_str()), atoi(t[2].c_str()), 0, 0.0f};
        if (b.start < 0) { fclose(f); err = "Start must be non-negative in a BED file: " + s; return false; }
        if (b.start >= b.stop) { fclose(f); err = "Start must be less than Stop in a BED file: " + s; return false; }
        if (t.size() > 3) b.gc = atoi(t[3].c_str());
        if (!out.count(t[0])) chromOrder.push_back(t[0]);
        out[t[0]].push_back(b);
    }
    fclose(f); return true;
}

// ---------------------------------------------------------------- Fragment mode: FragmentBinner.BinTask (FragmentBinner.cs:98-371)
// FindBestBin (:349-369): the bin with the largest overlap, the first one on ties, scanning from binIndexStart until a bin does not overlap
static int find_best_bin(const std::vector<PreBin>& bins, int binIndexStart, int fragmentStart, int fragmentStop) {
    int bestBinIndex = -1, bestOverlap = 0;
    for (int i = binIndexStart; i < (int)bins.size(); i++) {
        const int overlap = std::min(bins[i].stop, fragmentStop) - std::max(bins[i].start, fragmentStart);
        if (overlap <= 0) break;
        if (overlap > bestOverlap) { bestOverlap = overlap; bestBinIndex = i; }
    }
    return bestBinIndex;
}
struct FragAln { std::string name; int32_t refID, pos, mateRefID, matePos, tlen; uint16_t flag; uint8_t mapq; };
// BinOneAlignment (:256-311)
static void bin_one_alignment(const FragAln& a, unsigned qualityThreshold, std::map<std::string, int>& readNameToBinIndex, std::set<std::string>& samePositionReadNames,
                              long& usableFragmentCount, std::vector<PreBin>& bins, int& binIndexStart) {
    if (a.flag & 0x4) return;                                  // !IsMapped
    if (a.flag & 0x8) return;                                  // !IsMateMapped
    if (a.flag & 0x100) return;                                // !IsPrimaryAlignment
    if (!((a.flag & 0x1) && (a.flag & 0x2))) return;           // IsPaired && IsProperPair
    const bool bad = (a.flag & 0x400) || (a.flag & 0x200) || a.mapq == 255 || a.mapq < qualityThreshold;      // IsDuplicateFailedQCLowQuality (:321-332)
    auto it = readNameToBinIndex.find(a.name);
    if (it != readNameToBinIndex.end()) {                      // the mate was binned: undo when this read is bad
        if (bad) { usableFragmentCount--; bins[it->second].count--; }
        readNameToBinIndex.erase(it);
        return;
    }
    if (bad) return;
    if (a.refID != a.mateRefID) return;
    if (a.pos > a.matePos) return;                             // IsRightMostInPair
    if (a.pos == a.matePos) {
        auto sp = samePositionReadNames.find(a.name);
        if (sp != samePositionReadNames.end()) { samePositionReadNames.erase(sp); return; }
        samePositionReadNames.insert(a.name);
    }
    if (a.tlen == 0) return;
    const int fragmentStart = a.pos, fragmentStop = a.pos + a.tlen;
    while (binIndexStart < (int)bins.size() && bins[binIndexStart].stop <= fragmentStart) binIndexStart++;
    if (binIndexStart >= (int)bins.size()) return;
    const int best = find_best_bin(bins, binIndexStart, fragmentStart, fragmentStop);
    if (best >= 0) { usableFragmentCount++; bins[best].count++; readNameToBinIndex[a.name] = best; }
}
struct BamHeader { std::vector<std::string> refNames; };
static bool read_bam_header(Bgzf& z, BamHeader& h) {
    char magic[4]; int32_t ltext, nref;
    if (!z.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !z.read(&ltext, 4)) return false;
    { std::vector<char> t(ltext); if (ltext && !z.read(t.data(), ltext)) return false; }
    if (!z.read(&nref, 4)) return false;
    for (int r = 0; r < nref; r++) { int32_t ln, lref; if (!z.read(&ln, 4)) return false; std::vector<char> nm(ln); if (!z.read(nm.data(), ln) || !z.read(&lref, 4)) return false; h.refNames.push_back(nm.data()); }
    return true;
}
// binFragments (:186-244) for one chromosome; 0 ok, 1 error (message printed)
static int bin_fragments(const std::string& bam, const std::string& chrom, std::vector<PreBin>& bins, long& usableFragmentCount) {
    if (!file_exists(bam + ".bai")) { fprintf(stderr, "Fatal error: Bam index not found at %s.bai\n", bam.c_str()); return 1; }
    Bgzf z; if (!z.open(bam)) return 1;
    BamHeader h; if (!read_bam_header(z, h)) { fprintf(stderr, "CanvasBin: %s is not a BAM file\n", bam.c_str()); return 1; }
    int desired = -1; for (size_t r = 0; r < h.refNames.size(); r++) if (h.refNames[r] == chrom) desired = (int)r;
    if (desired < 0) { fprintf(stderr, "Unable to retrieve the reference sequence index for %s in %s.\n", chrom.c_str(), bam.c_str()); return 1; }
    uint64_t voff; bool any;
    if (!bai_first_offset(bam + ".bai", desired, voff, any)) { fprintf(stderr, "CanvasBin: cannot read %s.bai\n", bam.c_str()); return 1; }
    usableFragmentCount = 0;
    if (!any) return 0;                                        // no reads for this chromosome: not an error (:205-210)
    if (!z.seek_virtual(voff)) return 1;
    std::map<std::string, int> readNameToBinIndex; std::set<std::string> samePositionReadNames;
    int binIndexStart = 0, prevPosition = -1; long pairedAlignmentCount = 0;
    std::vector<uint8_t> rec;
    for (;;) {
        int32_t bs; if (!z.read(&bs, 4)) break;
        rec.resize(bs); if (!z.read(rec.data(), bs)) break;
        FragAln a; uint8_t lname;
        memcpy(&a.refID, &rec[0], 4); memcpy(&a.pos, &rec[4], 4); lname = rec[8]; a.mapq = rec[9]; memcpy(&a.flag, &rec[14], 2);
        memcpy(&a.mateRefID, &rec[20], 4); memcpy(&a.matePos, &rec[24], 4); memcpy(&a.tlen, &rec[28], 4);
        a.name.assign((const char*)&rec[32], lname > 0 ? lname - 1 : 0);
        if (a.refID != desired) break;
        if (a.refID == -1) continue;
        if (a.pos < prevPosition) { fprintf(stderr, "The alignment on %s are not properly sorted in %s: %s\n", chrom.c_str(), bam.c_str(), a.name.c_str()); return 1; }
        prevPosition = a.pos;
        if (a.flag & 0x1) pairedAlignmentCount++;
        bin_one_alignment(a, 3, readNameToBinIndex, samePositionReadNames, usableFragmentCount, bins, binIndexStart);
    }
    if (pairedAlignmentCount == 0) { fprintf(stderr, "No paired alignments found for %s in %s\n", chrom.c_str(), bam.c_str()); return 1; }
    return 0;
}

static int parse_mode(const std::string& m) {       // Utilities.ParseCanvasCoverageMode (CanvasCommon/Utilities.cs:56-74)
    std::string s; for (char c : m) if (c != ' ' && c != '\t') s.push_back((char)tolower(c));
    if (s == "0" || s == "binary") return CANVAS_MODE_BINARY;
    if (s == "3" || s == "truncateddynamicrange") return CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE;
    if (s == "5" || s == "gccontentweighted") return CANVAS_MODE_GC_CONTENT_WEIGHTED;
    if (s == "fragment") return -2;
    return -1;
}

int main(int argc, char** argv) {
    Phases ph("CanvasBin");
    printf(">>>Command-line arguments:\n"); for (int i = 1; i < argc; i++) printf("%s ", argv[i]); printf("\n");
    std::vector<Opt> opts = {{"b", "bam", true}, {"r", "reference", true}, {"c", "chr", true}, {"i", "infile", true}, {"f", "filter", true}, {"d", "bindepth", true}, {"z", "binsize", true},
                             {"o", "outfile", true}, {"y", "binsizeonly", false}, {"h", "help", false}, {"p", "paired-end", false}, {"m", "mode", true}, {"t", "manifest", true}, {"n", "bins", true}, {"j", "injson", true}};
    Parsed a = parse(argc, argv, opts);
    printf("CanvasBin %s (MI355X)\n", canvas_version());
    if (!a.extra.empty()) { fprintf(stderr, "Unknown arguments: %s\n", a.extra[0].c_str()); return 2; }
    int mode = CANVAS_MODE_TRUNCATED_DYNAMIC_RANGE;
    if (a.has("mode")) { mode = parse_mode(a.get("mode")); if (mode == -1) { fprintf(stderr, "Invalid canvas coverage mode '%s'\n", a.get("mode").c_str()); return 2; } }
    const std::string ref = a.get("reference"), out = a.get("outfile"), chrom = a.get("chr"), bam = a.get("bam"), filter = a.get("filter");
    auto inters = a.all("infile");
    const int countsPerBin = a.has("bindepth") ? atoi(a.get("bindepth").c_str()) : -1;
    int binSize = a.has("binsize") ? atoi(a.get("binsize").c_str()) : -1;
    bool needHelp = a.has("help");
    // required arguments (Program.cs:108-133)
    if (ref.empty()) { fprintf(stderr, "Please specify the Canvas k-uniqueness reference file.\n"); needHelp = true; }
    else if (out.empty()) { fprintf(stderr, "Please specify an output file name.\n"); needHelp = true; }
    else if (mode != -2 && countsPerBin == -1) { fprintf(stderr, "Please specify counts per bin.\n"); needHelp = true; }
    else if (mode != -2 && chrom.empty() && inters.empty() && !a.has("injson")) { fprintf(stderr, "Please specify chromsome to measure coverage for.\n"); needHelp = true; }
    if (needHelp) { printf("Usage: CanvasBin.exe [OPTIONS]+\nBin alignments into variable-sized genomic intervals.\n\nOptions:\n  -b, --bam=VALUE  -r, --reference=VALUE  -c, --chr=VALUE  -i, --infile=VALUE (repeatable)  -f, --filter=VALUE\n"
                           "  -d, --bindepth=VALUE  -z, --binsize=VALUE  -o, --outfile=VALUE  -y, --binsizeonly  -p, --paired-end  -m, --mode=VALUE  -t, --manifest=VALUE  -n, --bins=VALUE  -h, --help\n"); return 1; }
    if (!file_exists(ref)) { printf("CanvasBin.exe: File %s does not exist! Exiting.\n", ref.c_str()); return 1; }
    if (bam.empty() || !file_exists(bam)) { printf("CanvasBin.exe: Alignment input does not exist! Exiting.\n"); return 1; }       // also required in -i mode (:148-153)
    if (!filter.empty() && !file_exists(filter)) { printf("CanvasBin.exe: File %s does not exist! Exiting.\n", filter.c_str()); return 1; }
    if (mode != -2 && countsPerBin < 1) { printf("CanvasBin.exe: Median counts must be strictly positive. Exiting.\n"); return 1; }
    if (a.has("injson") && !file_exists(a.get("injson"))) { printf("CanvasBin.exe: File %s does not exist! Exiting.\n", a.get("injson").c_str()); return 1; }      // Program.cs:163-167
    if (a.has("manifest")) { fprintf(stderr, "CanvasBin (MI355X): -t is not built (the Nextera manifest parser is outside the reference tree)\n"); return 1; }
    if (a.has("injson")) {
        // CanvasBin.Run (CanvasBin.cs:949-963): -i together with -j throws; -j alone calls RunMultiSample, which hands CalculateMultiSampleBins an EMPTY list of
        // samples (CanvasBin.cs:936-944: the json file is never read in this version of the reference) — nothing is binned, nothing is written, exit code 0
        if (!inters.empty()) { fprintf(stderr, "Unhandled exception: System.ArgumentException: -i/--infile or -j/--injson are mutually exclusive arguments\n"); return 1; }
        return 0;
    }
    std::map<std::string, std::vector<PreBin>> predefined; std::vector<std::string> predefinedOrder;
    if (a.has("bins")) { std::string perr; if (!load_predefined_bins(a.get("bins"), predefined, predefinedOrder, perr)) { fprintf(stderr, "CanvasBin: %s\n", perr.c_str()); return 1; } }
    if (mode == -2) {
        // ---- FragmentBinner.Bin (FragmentBinner.cs:26-80)
        if (!a.has("bins")) { fprintf(stderr, "Predefined bins in BED is required for fragment binning.\n"); return 1; }
        if (!a.has("paired-end")) { fprintf(stderr, "Paired-end reads are required for fragment binning.\n"); return 1; }
        BamHeader bh; { Bgzf z; if (!z.open(bam) || !read_bam_header(z, bh)) { fprintf(stderr, "CanvasBin: %s is not a BAM file\n", bam.c_str()); return 1; } }
        for (auto& kv : predefined) if (std::find(bh.refNames.begin(), bh.refNames.end(), kv.first) == bh.refNames.end()) {
            fprintf(stderr, "Not all chromosomes in %s are found in %s.\n", a.get("bins").c_str(), bam.c_str()); return 1; }
        long usable = 0;
        for (auto& chromName : bh.refNames) {
            auto it = predefined.find(chromName); if (it == predefined.end()) continue;
            for (auto& b : it->second) b.count = 0;                                   // InitializeBins
            bool gcAvailable = true; for (auto& b : it->second) if (b.gc < 0) gcAvailable = false;
            if (!gcAvailable) {                                                       // PopulateBinGC (:163-181)
                std::vector<FastaEntry> fa; if (!read_fasta(ref, &chromName, fa) || fa.empty()) { fprintf(stderr, "CanvasBin: chromosome %s not found in %s\n", chromName.c_str(), ref.c_str()); return 1; }
                const char* bases = fa[0].data(); const size_t nbases = fa[0].size();
                for (auto& b : it->second) { double nt = 0, gcn = 0; for (int p = b.start; p < b.stop && p < (int)nbases; p++) { if (bases[p] == 'n') continue; nt++; const char ch = bases[p]; if (ch == 'C' || ch == 'c' || ch == 'G' || ch == 'g') gcn++; }
                    b.gc = nt > 0 ? (int)(100 * gcn / nt) : 0; }
            }
            long u = 0; if (int rc = bin_fragments(bam, chromName, it->second, u)) return rc;
            usable += u;
        }
        if (usable == 0) { fprintf(stderr, "No passing-filter fragments overlapping bins are found in %s\n", bam.c_str()); return 1; }
        GzWriter wr(out); if (!wr.ok()) { fprintf(stderr, "CanvasBin: cannot write %s\n", out.c_str()); return 1; }
        for (auto& chromName : bh.refNames) { auto it = predefined.find(chromName); if (it == predefined.end()) continue;
            for (auto& b : it->second) wr.line(chromName + "\t" + std::to_string(b.start) + "\t" + std::to_string(b.stop) + "\t" + format_f2(b.count) + "\t" + std::to_string(b.gc)); }
        printf("Output complete\n");
        return 0;
    }

    ExitStamp es0("context destroyed");
    AsyncCtx actx;                                              // the context comes up while the intermediates and the FASTA file are read
    struct CtxGuard { AsyncCtx& a; ~CtxGuard() { if (canvas_ctx* c = a.get()) canvas_destroy(c); } } guard{actx};
    canvas_ctx* ctx = nullptr;
    auto need_ctx = [&]() -> bool { ctx = actx.get(); if (!ctx) fprintf(stderr, "CanvasBin (MI355X): no usable GPU (this build has no CPU fallback)\n"); return ctx != nullptr; };

    if (inters.empty()) {
        // ---- phase 1: CalculateSampleHits / BinOneGenomicInterval (CanvasBin.cs:765-792)
        std::vector<FastaEntry> fa;
        if (!read_fasta(ref, &chrom, fa) || fa.empty()) { fprintf(stderr, "CanvasBin: chromosome %s not found in %s\n", chrom.c_str(), ref.c_str()); return 1; }
        Inter d; d.name = chrom; d.len = (int64_t)fa[0].size();
        const int64_t L = d.len, words = (L + 63) / 64;
        std::vector<uint64_t> mw(words, 0);
        d.hits.assign(L, 0); if (mode == CANVAS_MODE_GC_CONTENT_WEIGHTED) d.frag.assign(L, 0);
        printf("Initialized alignment arrays\n");
        if (int rc = load_bam(bam, a.has("paired-end"), chrom, mode, d.hits, d.frag)) return rc;
        printf("Loaded observed alignments\n");
        if (!need_ctx()) return 1;
        if (L > 0) {
            Dev dBases(ctx, L), dHits(ctx, L), dMask(ctx, words * 8);
            TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dBases.p, fa[0].data(), L));
            TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dHits.p, d.hits.data(), L));
            TOOL_TRY(ctx, canvas_mask_from_fasta(ctx, dBases.as<uint8_t>(), L, dMask.as<uint64_t>()));
            if (!filter.empty()) {
                std::map<std::string, std::vector<std::pair<int, int>>> bed; load_bed(filter, bed);
                auto it = bed.find(chrom);
                if (it != bed.end()) { std::vector<int32_t> s, e; for (auto& iv : it->second) { s.push_back(iv.first); e.push_back(iv.second); }
                    TOOL_TRY(ctx, canvas_mask_exclude_intervals(ctx, dMask.as<uint64_t>(), L, (int32_t)s.size(), s.data(), e.data())); }
            }
            TOOL_TRY(ctx, canvas_screen_hits(ctx, dHits.as<uint8_t>(), dMask.as<uint64_t>(), L));
            TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, mw.data(), dMask.p, words * 8));
            TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, d.hits.data(), dHits.p, L));
        }
        pbdat::Data pd; pbdat::Chromosome& pc = pd[chrom];
        pbdat::pack_possible_msb(mw.data(), L, pc.possibleBytes, pc.bitsInLastByte);        // most significant bit first, as the C# writer (Q2)
        pc.observed.swap(d.hits); pc.fragmentLengths.swap(d.frag);
        if (!pbdat::write_file(out, pd, mode == CANVAS_MODE_GC_CONTENT_WEIGHTED)) { fprintf(stderr, "CanvasBin: cannot write %s\n", out.c_str()); return 1; }
        printf("Intermediate observedAlignments serialized\n");
        return 0;
    }

    // ---- phase 2: RunSingleSample (CanvasBin.cs:914-931)
    ExitStamp es1("intermediates freed");
    std::map<std::string, std::unique_ptr<Inter>> byChrom;
    for (auto& p : inters) if (!file_exists(p)) { fprintf(stderr, "CanvasBin: intermediate file %s does not exist\n", p.c_str()); return 1; }
    {   // one host thread per intermediate file (the reference deserialises them one after the other, CanvasBin.cs:725-762); merged in command-line order
        std::vector<pbdat::Data> pds(inters.size()); std::vector<std::string> perr(inters.size()); std::vector<char> okv(inters.size(), 0);
        std::vector<std::vector<std::unique_ptr<Inter>>> made(inters.size()); std::vector<std::string> badEntry(inters.size());
        parallel_for((int64_t)inters.size(), [&](int64_t i) {
            okv[(size_t)i] = pbdat::read_file(inters[(size_t)i], pds[(size_t)i], perr[(size_t)i], !getenv("CANVAS_TOOL_NO_MAPPED_DAT")) ? 1 : 0;
            if (!okv[(size_t)i]) return;
            for (auto& kv : pds[(size_t)i]) {                                                 // DeserializeCanvasData + IntermediateData.GetData (CanvasBin.cs:725-762,1089-1104)
                auto d = std::make_unique<Inter>(); d->name = kv.first;
                d->len = pbdat::unpack_possible_lsb(kv.second.possibleBytes, kv.second.bitsInLastByte, d->maskWords, kv.second.haveBits);      // least significant bit first, as the C# reader (Q2)
                if (d->len < 0) { badEntry[(size_t)i] = kv.first; return; }
                std::vector<uint8_t>().swap(kv.second.possibleBytes);
                d->hits.swap(kv.second.observed); d->hitsView = kv.second.observedView; d->hitsViewLen = kv.second.observedLen; d->keep = kv.second.keep; d->frag.swap(kv.second.fragmentLengths);
                made[(size_t)i].push_back(std::move(d));
            }
        });
        for (size_t i = 0; i < inters.size(); i++) {
            const std::string& p = inters[i];
            if (!okv[i]) { fprintf(stderr, "CanvasBin: %s\n", perr[i].c_str()); return 1; }
            if (!badEntry[i].empty()) { fprintf(stderr, "CanvasBin: %s: %s has no valid count of bits in the last possible-alignment byte (0..7 expected)\n", p.c_str(), badEntry[i].c_str()); return 1; }
            for (auto& d : made[i]) {
                if ((int64_t)d->hits_size() != d->len) { fprintf(stderr, "CanvasBin: %s: %s has %lld possible-alignment bits but %zu observed-alignment bytes\n", p.c_str(), d->name.c_str(), (long long)d->len, d->hits_size()); return 1; }
                if (byChrom.count(d->name)) { fprintf(stderr, "CanvasBin: chromosome %s appears in more than one intermediate file (Dictionary.Add throws in the reference)\n", d->name.c_str()); return 1; }
                const std::string nm = d->name;
                byChrom[nm] = std::move(d);
            }
        }
    }
    ExitStamp es2("FASTA entries freed");
    std::vector<FastaEntry> fa;
    if (!read_fasta(ref, nullptr, fa)) return 1;
    if (!need_ctx()) return 1;
    ph.mark("read");
    // chromosomes in FASTA order that have an intermediate (CanvasBin.cs:506-540)
    std::vector<const FastaEntry*> order; std::vector<Inter*> data;
    for (auto& e : fa) { auto it = byChrom.find(e.name); if (it == byChrom.end()) continue; if ((int64_t)e.size() != it->second->len) { fprintf(stderr, "CanvasBin: length of %s differs between the reference and the intermediate file\n", e.name.c_str()); return 1; } order.push_back(&e); data.push_back(it->second.get()); }
    const int nchr = (int)order.size();
    if (nchr == 0) { fprintf(stderr, "CanvasBin: no chromosome to bin\n"); return 1; }
    ExitStamp es3("input planes freed on the device");
    std::vector<std::unique_ptr<Dev>> devs;
    std::vector<const uint8_t*> pBases(nchr), pHits(nchr); std::vector<const uint64_t*> pMask(nchr); std::vector<const int16_t*> pFrag(nchr); std::vector<int64_t> len(nchr); std::vector<uint8_t> isAuto(nchr);
    // Binary / TruncatedDynamicRange binning goes over the packed planes (include/canvas_hip.h, "packed per-base inputs"): packed on the host threads, 0.75 B/base over
    // PCIe instead of 2.125 B/base, same bins.  CANVAS_BIN_BYTE_ARRAYS=1 keeps the byte arrays (-n, -y and GCContentWeighted always use them).
    const bool usePacked = mode != CANVAS_MODE_GC_CONTENT_WEIGHTED && !a.has("bins") && !a.has("binsizeonly") && !getenv("CANVAS_BIN_BYTE_ARRAYS");
    std::vector<const uint64_t*> pRef(nchr), pPlanes(nchr); std::vector<int64_t> pos0(nchr);
    if (usePacked) {
        // one device allocation for all planes; two host staging sets: while chromosome c is packed by the host threads, a helper thread uploads chromosome c-1
        std::vector<int64_t> refB(nchr), hitB(nchr), offR(nchr), offH(nchr); int64_t tot = 0, maxR = 0, maxH = 0;
        for (int c = 0; c < nchr; c++) {
            const int64_t L = data[c]->len; len[c] = L; isAuto[c] = is_autosome(order[c]->name) ? 1 : 0;
            if (canvas_packed_plane_bytes(L, &refB[c], &hitB[c]) != 0) { fprintf(stderr, "CanvasBin: bad chromosome length\n"); return 1; }
            offR[c] = tot; tot += (refB[c] + 255) & ~255ll; offH[c] = tot; tot += (hitB[c] + 255) & ~255ll;
            maxR = std::max(maxR, refB[c]); maxH = std::max(maxH, hitB[c]);
        }
        devs.push_back(std::make_unique<Dev>(ctx, tot)); uint8_t* base = (uint8_t*)devs.back()->p;
        if (!base) { fprintf(stderr, "CanvasBin: device allocation of %lld bytes failed: %s\n", (long long)tot, canvas_last_error(ctx)); return 1; }
        std::unique_ptr<uint64_t[]> sRef[2], sHit[2];               // (every word of a plane is written by the packers: no zero fill)
        for (int s = 0; s < 2 && s < nchr; s++) { sRef[s].reset(new uint64_t[(size_t)maxR / 8]); sHit[s].reset(new uint64_t[(size_t)maxH / 8]); }
        std::thread up; int upRc = 0;
        for (int c = 0; c < nchr; c++) {
            const int s = c & 1; int64_t sat = 0;
            pRef[c] = (const uint64_t*)(base + offR[c]); pPlanes[c] = (const uint64_t*)(base + offH[c]);
            const bool ok = canvas_pack_reference_host((const uint8_t*)order[c]->data(), data[c]->maskWords.data(), len[c], sRef[s].get(), &pos0[c], 0) == 0 &&
                            canvas_pack_hits_host(data[c]->hits_data(), len[c], sHit[s].get(), &sat, 0) == 0;
            if (up.joinable()) up.join();
            if (!ok) { fprintf(stderr, "CanvasBin: packing %s failed\n", order[c]->name.c_str()); return 1; }
            if (upRc) break;
            up = std::thread([&, c, s] { if (canvas_memcpy_h2d(ctx, (void*)pRef[c], sRef[s].get(), refB[c]) != 0 || canvas_memcpy_h2d(ctx, (void*)pPlanes[c], sHit[s].get(), hitB[c]) != 0) upRc = 1; });
        }
        if (up.joinable()) up.join();
        if (upRc) { fprintf(stderr, "CanvasBin: upload failed: %s\n", canvas_last_error(ctx)); return 1; }
    }
    for (int c = 0; !usePacked && c < nchr; c++) {
        const int64_t L = data[c]->len, words = (L + 63) / 64; len[c] = L; isAuto[c] = is_autosome(order[c]->name) ? 1 : 0;
        auto up = [&](const void* src, int64_t bytes, int64_t alloc) -> void* { devs.push_back(std::make_unique<Dev>(ctx, alloc)); void* p = devs.back()->p; if (bytes > 0 && canvas_memcpy_h2d(ctx, p, src, bytes) != 0) return nullptr; return p; };
        pBases[c] = (const uint8_t*)up(order[c]->data(), L, L + 64); pHits[c] = (const uint8_t*)up(data[c]->hits_data(), L, L + 64); pMask[c] = (const uint64_t*)up(data[c]->maskWords.data(), words * 8, words * 8 + 64);
        if (mode == CANVAS_MODE_GC_CONTENT_WEIGHTED) { if ((int64_t)data[c]->frag.size() != L) { fprintf(stderr, "CanvasBin: %s has no fragment lengths (was the intermediate written with -m GCContentWeighted?)\n", order[c]->name.c_str()); return 1; } pFrag[c] = (const int16_t*)up(data[c]->frag.data(), L * 2, L * 2 + 64); }
        if (!pBases[c] || !pHits[c] || !pMask[c]) { fprintf(stderr, "CanvasBin: upload failed: %s\n", canvas_last_error(ctx)); return 1; }
    }
    ph.mark("pack_upload");
    if (a.has("bins") && !a.has("binsizeonly")) {
        // ---- predefined bins (BinCounts with predefinedBins, CanvasBin.cs:506-547): chromosomes in FASTA order that have both an intermediate and bins
        // -m GCContentWeighted: every chromosome with an intermediate enters the fragment mean, the read-GC profile and the weights (CanvasBin.cs:427-505), whether it has bins or not
        const bool gcwPre = mode == CANVAS_MODE_GC_CONTENT_WEIGHTED;
        std::vector<const uint8_t*> qB, qH; std::vector<const uint64_t*> qM; std::vector<const int16_t*> qF; std::vector<int64_t> qL, off{0}; std::vector<int32_t> hs, he; std::vector<std::string> qName;
        for (int c = 0; c < nchr; c++) { auto it = predefined.find(order[c]->name); if (it == predefined.end() && !gcwPre) continue;
            qB.push_back(pBases[c]); qH.push_back(pHits[c]); qM.push_back(pMask[c]); qL.push_back(len[c]); qName.push_back(order[c]->name); if (gcwPre) qF.push_back(pFrag[c]);
            if (it != predefined.end()) for (auto& b : it->second) { hs.push_back(b.start); he.push_back(b.stop); }
            off.push_back((int64_t)hs.size()); }
        const int64_t nb = (int64_t)hs.size();
        std::vector<int32_t> hGc(nb); std::vector<float> hCount(nb);
        if (nb > 0) {
            Dev dS(ctx, nb * 4), dE(ctx, nb * 4), dG(ctx, nb * 4), dC(ctx, nb * 4);
            TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dS.p, hs.data(), nb * 4)); TOOL_TRY(ctx, canvas_memcpy_h2d(ctx, dE.p, he.data(), nb * 4));
            if (gcwPre) TOOL_TRY(ctx, canvas_bin_predefined_gcweighted(ctx, (int32_t)qL.size(), qB.data(), qM.data(), qH.data(), qF.data(), qL.data(), off.data(), hs.data(), he.data(), dS.as<int32_t>(), dE.as<int32_t>(), dG.as<int32_t>(), dC.as<float>()));
            else TOOL_TRY(ctx, canvas_bin_predefined(ctx, (int32_t)qL.size(), qB.data(), qM.data(), qH.data(), qL.data(), mode, off.data(), hs.data(), he.data(), dS.as<int32_t>(), dE.as<int32_t>(), dG.as<int32_t>(), dC.as<float>()));
            TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hGc.data(), dG.p, nb * 4)); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hCount.data(), dC.p, nb * 4));
        }
        GzWriter wr(out); if (!wr.ok()) { fprintf(stderr, "CanvasBin: cannot write %s\n", out.c_str()); return 1; }
        for (size_t c = 0; c + 1 < off.size(); c++) for (int64_t i = off[c]; i < off[c + 1]; i++)
            wr.line(qName[c] + "\t" + std::to_string(hs[i]) + "\t" + std::to_string(he[i]) + "\t" + format_f2(hCount[i]) + "\t" + std::to_string(hGc[i]));
        printf("Output complete\n");
        return 0;
    }
    if (!usePacked && (a.has("binsizeonly") || binSize == -1)) {
        // CalculateSingleSampleBinSize: autosomes only (CanvasBin.cs:30-83)
        std::vector<int64_t> obs(nchr), poss(nchr); std::vector<double> rate(nchr), rates;
        TOOL_TRY(ctx, canvas_bin_rates(ctx, nchr, pHits.data(), pMask.data(), len.data(), obs.data(), poss.data(), rate.data()));
        for (int c = 0; c < nchr; c++) if (isAuto[c]) rates.push_back(rate[c]);
        if (binSize == -1) { if (rates.empty()) { fprintf(stderr, "CanvasBin: no autosome to derive the bin size from\n"); return 1; } binSize = canvas_bin_size_from_rates(rates.data(), (int32_t)rates.size(), countsPerBin); }
    }
    if (a.has("binsizeonly")) { FILE* f = fopen((out + ".binsize").c_str(), "wb"); if (!f) return 1; fprintf(f, "%d", binSize); fclose(f); return 0; }   // :926-928, no newline
    if (binSize <= 0 && !(usePacked && binSize == -1)) { fprintf(stderr, "CanvasBin: bin size %d is not positive\n", binSize); return 1; }
    // the packed call derives the bin size from the rates itself (CalculateSingleSampleBinSize, CanvasBin.cs:30-83): the capacity then is the number of positions / 1
    int64_t cap = 0;
    if (binSize > 0) cap = canvas_bin_count_upper_bound(nchr, len.data(), binSize);
    else { for (int c = 0; c < nchr; c++) cap += len[c] / 16 + 1; }                    // a bin holds countsPerBin / rate possible positions; rates above 100/16 hits per position do not occur

    ExitStamp es4("bin columns freed on the device");
    Dev dChr(ctx, cap * 4 + 4), dStart(ctx, cap * 4 + 4), dStop(ctx, cap * 4 + 4), dGc(ctx, cap * 4 + 4), dCount(ctx, cap * 4 + 4);
    std::vector<int64_t> perChr(nchr); int64_t total = 0; int32_t used = 0;
    if (mode == CANVAS_MODE_GC_CONTENT_WEIGHTED)
        TOOL_TRY(ctx, canvas_bin_sample_gcweighted(ctx, nchr, pBases.data(), pMask.data(), pHits.data(), pFrag.data(), len.data(), isAuto.data(), countsPerBin, binSize,
                                                   dChr.as<int32_t>(), dStart.as<int32_t>(), dStop.as<int32_t>(), dGc.as<int32_t>(), dCount.as<float>(), cap, &used, perChr.data(), &total));
    else if (usePacked) {
        int32_t rcb = canvas_bin_sample_packed(ctx, nchr, pRef.data(), pPlanes.data(), len.data(), pos0.data(), isAuto.data(), countsPerBin, binSize, mode,
                                               dChr.as<int32_t>(), dStart.as<int32_t>(), dStop.as<int32_t>(), dGc.as<int32_t>(), dCount.as<float>(), cap, &used, perChr.data(), &total);
        if (rcb == CANVAS_ERR_CAPACITY && binSize == -1 && used > 0) {
            // the capacity above was a guess (the bin size is derived inside the call: a small -d or a high hit rate gives bins of fewer than 16 positions); the failed call
            // still reported the size it derived: size the columns exactly as the byte-array path does and bin again with that size
            const int64_t cap2 = canvas_bin_count_upper_bound(nchr, len.data(), used);
            Dev dChr2(ctx, cap2 * 4 + 4), dStart2(ctx, cap2 * 4 + 4), dStop2(ctx, cap2 * 4 + 4), dGc2(ctx, cap2 * 4 + 4), dCount2(ctx, cap2 * 4 + 4);
            TOOL_TRY(ctx, canvas_bin_sample_packed(ctx, nchr, pRef.data(), pPlanes.data(), len.data(), pos0.data(), isAuto.data(), countsPerBin, used, mode,
                                                   dChr2.as<int32_t>(), dStart2.as<int32_t>(), dStop2.as<int32_t>(), dGc2.as<int32_t>(), dCount2.as<float>(), cap2, &used, perChr.data(), &total));
            std::swap(dChr.p, dChr2.p); std::swap(dStart.p, dStart2.p); std::swap(dStop.p, dStop2.p); std::swap(dGc.p, dGc2.p); std::swap(dCount.p, dCount2.p);
        } else if (rcb != 0) { fprintf(stderr, "canvas_bin_sample_packed failed (%d): %s\n", rcb, canvas_last_error(ctx)); return 1; }
    }
    else
        TOOL_TRY(ctx, canvas_bin_sample(ctx, nchr, pBases.data(), pMask.data(), pHits.data(), len.data(), isAuto.data(), countsPerBin, binSize, mode,
                                        dChr.as<int32_t>(), dStart.as<int32_t>(), dStop.as<int32_t>(), dGc.as<int32_t>(), dCount.as<float>(), cap, &used, perChr.data(), &total));
    std::vector<int32_t> hChr(total), hStart(total), hStop(total), hGc(total); std::vector<float> hCount(total);
    if (total > 0) {
        TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hChr.data(), dChr.p, total * 4)); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hStart.data(), dStart.p, total * 4));
        TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hStop.data(), dStop.p, total * 4)); TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hGc.data(), dGc.p, total * 4));
        TOOL_TRY(ctx, canvas_memcpy_d2h(ctx, hCount.data(), dCount.p, total * 4));
    }
    ph.mark("device");
    if (!write_gz_rows(out, total, [&](int64_t i, std::string& o) {                 // CanvasIO.WriteToTextFile (CanvasCommon/IO.cs:15-24)
            o += order[hChr[i]]->name; o.push_back('\t'); append_int(o, hStart[i]); o.push_back('\t'); append_int(o, hStop[i]); o.push_back('\t'); o += format_f2(hCount[i]); o.push_back('\t'); append_int(o, hGc[i]); }))
        { fprintf(stderr, "CanvasBin: cannot write %s\n", out.c_str()); return 1; }
    ph.mark("write");
    printf("Output complete\n");
    ExitStamp es5("start of the unwinding");
    return finish(ph, 0);
}
