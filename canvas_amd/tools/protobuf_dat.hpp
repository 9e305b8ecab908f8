// The CanvasBin intermediate file (*.dat): CanvasBin.IntermediateData (CanvasBin/CanvasBin.cs:1037-1148) as protobuf-net 2.3.7 writes it with
// Serializer.Serialize (no length prefix).  protobuf-net is not part of /root/reference; the encoding below is the protobuf wire format of the contract:
//   [ProtoMember(1)] Dictionary<string, byte[]>  PossibleAlignments               repeated field 1, each a nested message { 1: key (string), 2: value (bytes) }
//   [ProtoMember(2)] Dictionary<string, byte[]>  ObservedAlignments               repeated field 2, same entry shape
//   [ProtoMember(3)] Dictionary<string, int>     BitsInLastBytePossibleAlignments repeated field 3, entry { 1: key, 2: value (varint, two's complement) }
//   [ProtoMember(4)] Dictionary<string, Int16[]> FragmentLengths                  repeated field 4, entry { 1: key, 2: repeated value (varint; not packed) }
// (a dictionary is a repeated key/value message on the wire, whether protobuf-net treats it as a proto3 map or as a list of KeyValuePair).  The reader also accepts
// a packed field 2 in the fragment-length entries and entries whose zero-valued int was omitted.
//
// Q2: the C# writer packs the possible-alignment bits most-significant-bit first ("bytes[byteIndex] *= 2; if (bit) bytes[byteIndex]++", :1060-1068) and its
// reader unpacks them with new BitArray(bytes), least-significant-bit first (:1118-1138): across the .dat round trip the pipeline always takes, the order of the bits
// inside every byte is reversed.  pack_possible_msb / unpack_possible_lsb reproduce exactly that, so a .dat of this tool and one of the C# tool are interchangeable
// and the bins that come out are the ones the reference's pipeline produces.
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <string>
#include <vector>

namespace pbdat {

// a file mapped read-only: read_file (map = true) leaves the observed-alignment bytes of a chromosome — 1 byte per base, contiguous in the file — where they are
struct Mapped { const uint8_t* p = nullptr; size_t len = 0; Mapped() = default; Mapped(const Mapped&) = delete; Mapped& operator=(const Mapped&) = delete;
                ~Mapped() { if (p && len) munmap((void*)p, len); } };
struct Chromosome { std::vector<uint8_t> possibleBytes, observed; int bitsInLastByte = 0; bool haveBits = false; std::vector<int16_t> fragmentLengths;
                    const uint8_t* observedView = nullptr; size_t observedLen = 0; std::shared_ptr<Mapped> keep;      // (map = true) the observed alignments inside the mapped file instead of `observed`
                    const uint8_t* observed_data() const { return observedView ? observedView : observed.data(); }
                    size_t observed_size() const { return observedView ? observedLen : observed.size(); } };
typedef std::map<std::string, Chromosome> Data;     // keyed by chromosome name (each file of the pipeline holds one)

// ---- writer
static void put_varint(std::vector<uint8_t>& o, uint64_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
static size_t varint_size(uint64_t v) { size_t n = 1; while (v >= 0x80) { v >>= 7; n++; } return n; }
static bool write_all(FILE* f, const void* p, size_t n) { return n == 0 || fwrite(p, 1, n, f) == n; }
static bool write_entry_bytes(FILE* f, int field, const std::string& key, const uint8_t* val, size_t nval) {
    std::vector<uint8_t> h;
    const size_t entry = 1 + varint_size(key.size()) + key.size() + 1 + varint_size(nval) + nval;
    put_varint(h, (uint64_t)(field << 3 | 2)); put_varint(h, entry);
    h.push_back(0x0A); put_varint(h, key.size()); h.insert(h.end(), key.begin(), key.end());
    h.push_back(0x12); put_varint(h, nval);
    return write_all(f, h.data(), h.size()) && write_all(f, val, nval);
}
static bool write_file(const std::string& path, const Data& d, bool withFragments) {
    FILE* f = fopen(path.c_str(), "wb"); if (!f) return false;
    bool ok = true;
    for (auto& kv : d) ok = ok && write_entry_bytes(f, 1, kv.first, kv.second.possibleBytes.data(), kv.second.possibleBytes.size());
    for (auto& kv : d) ok = ok && write_entry_bytes(f, 2, kv.first, kv.second.observed.data(), kv.second.observed.size());
    for (auto& kv : d) {
        std::vector<uint8_t> e; e.push_back(0x0A); put_varint(e, kv.first.size()); e.insert(e.end(), kv.first.begin(), kv.first.end());
        e.push_back(0x10); put_varint(e, (uint64_t)(int64_t)kv.second.bitsInLastByte);
        std::vector<uint8_t> h; h.push_back(0x1A); put_varint(h, e.size());
        ok = ok && write_all(f, h.data(), h.size()) && write_all(f, e.data(), e.size());
    }
    if (withFragments) for (auto& kv : d) {
        std::vector<uint8_t> e; e.reserve(kv.second.fragmentLengths.size() * 3 + 64);
        e.push_back(0x0A); put_varint(e, kv.first.size()); e.insert(e.end(), kv.first.begin(), kv.first.end());
        for (int16_t v : kv.second.fragmentLengths) { e.push_back(0x10); put_varint(e, (uint64_t)(int64_t)v); }     // Int16 widened, negative values as 10-byte two's complement
        std::vector<uint8_t> h; h.push_back(0x22); put_varint(h, e.size());
        ok = ok && write_all(f, h.data(), h.size()) && write_all(f, e.data(), e.size());
    }
    ok = ok && !ferror(f);
    fclose(f); return ok;
}

// ---- reader
struct Cursor { const uint8_t* p; const uint8_t* end; bool ok = true;
    uint64_t varint() { uint64_t v = 0; int sh = 0; while (p < end) { const uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; if (sh > 63) break; } ok = false; return 0; }
    bool skip(int wire) { if (wire == 0) { varint(); } else if (wire == 1) { p += 8; } else if (wire == 2) { const uint64_t n = varint(); if (!ok || n > (uint64_t)(end - p)) { ok = false; return false; } p += n; } else if (wire == 5) { p += 4; } else ok = false; if (p > end) ok = false; return ok; } };
// map = true: the file is mapped instead of read, and the observed alignments are left in the mapping (Chromosome::observedView) — 3 GB per genome that are neither
// read into a buffer nor copied out of it (anonymous pages cost a fault when they are first touched and 37 ms per GB when the process gives them back, tools/exit_probe.sh)
static bool read_file(const std::string& path, Data& d, std::string& err, bool map = false) {
    std::vector<uint8_t> buf; std::shared_ptr<Mapped> mp; const uint8_t* base = nullptr; size_t total = 0;
    if (map) {
        const int fd = open(path.c_str(), O_RDONLY); if (fd < 0) { err = "cannot open " + path; return false; }
        struct stat sb; if (fstat(fd, &sb) != 0) { close(fd); err = "cannot stat " + path; return false; }
        total = (size_t)sb.st_size;
        if (total > 0) {
            void* m = mmap(nullptr, total, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); err = "cannot map " + path; return false; }
            (void)madvise(m, total, MADV_WILLNEED);
            mp = std::make_shared<Mapped>(); mp->p = (const uint8_t*)m; mp->len = total; base = mp->p;
        }
        close(fd);
    } else {
        FILE* f = fopen(path.c_str(), "rb"); if (!f) { err = "cannot open " + path; return false; }
        fseeko(f, 0, SEEK_END); const int64_t size = ftello(f); fseeko(f, 0, SEEK_SET);
        buf.resize((size_t)size);
        if (size > 0 && fread(buf.data(), 1, (size_t)size, f) != (size_t)size) { fclose(f); err = "short read on " + path; return false; }
        fclose(f);
        base = buf.data(); total = buf.size();
    }
    Cursor c{base, base + total};
    while (c.ok && c.p < c.end) {
        const uint64_t tag = c.varint(); const int field = (int)(tag >> 3), wire = (int)(tag & 7);
        if (!c.ok) break;
        if (field < 1 || field > 4 || wire != 2) { if (!c.skip(wire)) break; continue; }
        const uint64_t n = c.varint(); if (!c.ok || n > (uint64_t)(c.end - c.p)) { c.ok = false; break; }
        Cursor e{c.p, c.p + n}; c.p += n;
        std::string key; const uint8_t* val = nullptr; size_t nval = 0; int64_t ival = 0; std::vector<int16_t> arr;
        while (e.ok && e.p < e.end) {
            const uint64_t t2 = e.varint(); const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
            if (!e.ok) break;
            if (f2 == 1 && w2 == 2) { const uint64_t k = e.varint(); if (!e.ok || k > (uint64_t)(e.end - e.p)) { e.ok = false; break; } key.assign((const char*)e.p, (size_t)k); e.p += k; }
            else if (f2 == 2 && w2 == 2 && field <= 2) { const uint64_t k = e.varint(); if (!e.ok || k > (uint64_t)(e.end - e.p)) { e.ok = false; break; } val = e.p; nval = (size_t)k; e.p += k; }
            else if (f2 == 2 && w2 == 0 && field == 3) ival = (int64_t)e.varint();
            else if (f2 == 2 && w2 == 0 && field == 4) arr.push_back((int16_t)(int64_t)e.varint());
            else if (f2 == 2 && w2 == 2 && field == 4) { const uint64_t k = e.varint(); if (!e.ok || k > (uint64_t)(e.end - e.p)) { e.ok = false; break; } Cursor pk{e.p, e.p + k}; e.p += k;
                while (pk.ok && pk.p < pk.end) arr.push_back((int16_t)(int64_t)pk.varint()); if (!pk.ok) e.ok = false; }
            else if (!e.skip(w2)) break;
        }
        if (!e.ok) { c.ok = false; break; }
        Chromosome& ch = d[key];
        if (field == 1) ch.possibleBytes.assign(val, val + nval);
        else if (field == 2) { if (mp) { ch.observedView = val; ch.observedLen = nval; ch.keep = mp; std::vector<uint8_t>().swap(ch.observed); } else { ch.observed.assign(val, val + nval); ch.observedView = nullptr; ch.observedLen = 0; } }
        else if (field == 3) { ch.bitsInLastByte = (int)ival; ch.haveBits = true; }
        else ch.fragmentLengths.swap(arr);
    }
    if (!c.ok) { err = path + " is not a CanvasBin intermediate file (protobuf parse error)"; return false; }
    return true;
}

// ---- the possible-alignment bits
// IntermediateData constructor (CanvasBin.cs:1052-1072): bits in array order, most significant bit of each byte first; the last byte holds length % 8 bits in its LOW bits
static void pack_possible_msb(const uint64_t* lsbWords, int64_t length, std::vector<uint8_t>& bytes, int& bitsInLastByte) {
    bitsInLastByte = (int)(length % 8);
    bytes.assign((size_t)(length / 8 + (bitsInLastByte == 0 ? 0 : 1)), 0);
    for (int64_t i = 0; i < length; i++) {
        const int bit = (int)((lsbWords[i >> 6] >> (i & 63)) & 1ull);
        bytes[(size_t)(i >> 3)] = (uint8_t)(bytes[(size_t)(i >> 3)] * 2 + bit);
    }
}
// IntermediateData.Convert (CanvasBin.cs:1106-1135): new BitArray(bytes) is least significant bit first; of the last byte only bitsInLastByte bits are taken (when it is not 0).
// Returns the number of positions and the bits in the library's mask layout (bit i of the chromosome = bit (i & 63) of word i >> 6).
// -1: the entry is malformed (no bit count, a count outside 0..7, or a partial last byte without any byte) — the reference's writer only emits length % 8
static int64_t unpack_possible_lsb(const std::vector<uint8_t>& bytes, int bitsInLastByte, std::vector<uint64_t>& words, bool haveBits = true) {
    if (!haveBits || bitsInLastByte < 0 || bitsInLastByte > 7 || (bitsInLastByte > 0 && bytes.empty())) return -1;
    const int64_t length = bitsInLastByte > 0 ? 8 * ((int64_t)bytes.size() - 1) + bitsInLastByte : 8 * (int64_t)bytes.size();
    words.assign((size_t)((length + 63) / 64), 0);
    // bit i of the chromosome = bit (i & 7) of byte i >> 3 = bit (i & 63) of the little-endian word i >> 6: the bytes ARE the words
    if (!bytes.empty()) memcpy(words.data(), bytes.data(), std::min(bytes.size(), words.size() * 8));
    if (length & 63) words.back() &= (~0ull) >> (64 - (length & 63));         // bits of the last byte beyond bitsInLastByte are not part of the chromosome
    return length;
}

}  // namespace pbdat
