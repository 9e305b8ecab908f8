"""The drop-in CanvasBin executable (canvas_amd/bin/CanvasBin): BAM + kmer.fa -> per-chromosome intermediates -> S.binned, compared
row for row with the oracle run on hit arrays built by a numpy restatement of the reference's read filters (CanvasBin.cs:239-270)."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "canvas_amd", "bin", "CanvasBin")


@pytest.fixture(autouse=True, params=["packed_planes", "byte_arrays"])
def input_format(request, monkeypatch):
    """The executable bins Binary / TruncatedDynamicRange over the packed planes by default; CANVAS_BIN_BYTE_ARRAYS=1 keeps the byte arrays.  Both must write the same files."""
    if request.param == "byte_arrays":
        monkeypatch.setenv("CANVAS_BIN_BYTE_ARRAYS", "1")
    else:
        monkeypatch.delenv("CANVAS_BIN_BYTE_ARRAYS", raising=False)
    return request.param


def _bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    c = co.compress(data) + co.flush()
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(c) + 25)
    return hdr + c + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def _write_bam(path, refs, reads):
    """refs: [(name, length)]; reads: list of dicts sorted by (ref, pos).  Every reference starts a new BGZF block (its virtual offset goes
    into the .bai); inside a reference the stream is cut every 3000 bytes, so records span blocks."""
    text = b"@HD\tVN:1.0\tSO:coordinate\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for name, ln in refs:
        nm = name.encode() + b"\x00"
        hdr += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    out = bytearray(_bgzf_block(hdr))
    first = {}
    by_ref = {}
    for r in reads:
        by_ref.setdefault(r["ref"], []).append(r)
    for ref in sorted(by_ref, key=lambda x: (x < 0, x)):
        data = bytearray()
        for r in by_ref[ref]:
            name = r.get("name", "r").encode() + b"\x00"
            cig = b"".join(struct.pack("<I", (ln << 4) | "MIDNSHP=X".index(op)) for ln, op in r["cigar"])
            lseq = 36
            body = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), r.get("mapq", 30), 4680, len(r["cigar"]), r["flag"], lseq, r.get("mate_ref", -1), r.get("mate_pos", -1), r.get("tlen", 0))
            body += name + cig + bytes((lseq + 1) // 2) + bytes([30] * lseq)
            data += struct.pack("<i", len(body)) + body
        first[ref] = len(out) << 16
        for i in range(0, len(data), 3000):
            out += _bgzf_block(bytes(data[i:i + 3000]))
    out += _bgzf_block(b"")
    open(path, "wb").write(out)
    bai = b"BAI\x01" + struct.pack("<i", len(refs))
    for i in range(len(refs)):
        if i in first:
            bai += struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", first[i], len(out) << 16) + struct.pack("<i", 0)
        else:
            bai += struct.pack("<i", 0) + struct.pack("<i", 0)
    open(path + ".bai", "wb").write(bai)


# ---- protobuf wire format, written from the specification (varints, tags = field << 3 | wire type, length-delimited fields): the fixtures below do not share
# code with canvas_amd/tools/protobuf_dat.hpp
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80); v >>= 7
    out.append(v)
    return bytes(out)


def _ld(field, payload):
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _encode_dat(name, possible_bytes, observed, bits_in_last_byte, frag=None, packed=False, omit_zero=False):
    """CanvasBin.IntermediateData of one chromosome as protobuf-net lays it out: every dictionary = repeated { 1: key, 2: value } entries under its member number"""
    key = _ld(1, name.encode())
    msg = _ld(1, key + _ld(2, bytes(possible_bytes))) + _ld(2, key + _ld(2, bytes(observed)))
    msg += _ld(3, key + (b"" if (omit_zero and bits_in_last_byte == 0) else _varint(2 << 3 | 0) + _varint(bits_in_last_byte)))
    if frag is not None:
        if packed:
            msg += _ld(4, key + _ld(2, b"".join(_varint(int(v)) for v in frag)))
        else:
            msg += _ld(4, key + b"".join(_varint(2 << 3 | 0) + _varint(int(v)) for v in frag))
    return msg


def _decode_dat(blob):
    """{member number: {key: value}} of a .dat (bytes values as bytes, member 3 as int, member 4 as list of ints)"""
    def rd(buf, i):
        v = 0; sh = 0
        while True:
            b = buf[i]; i += 1; v |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80: return v, i
    out = {}
    i = 0
    while i < len(blob):
        tag, i = rd(blob, i); field, wire = tag >> 3, tag & 7
        assert wire == 2
        n, i = rd(blob, i); entry = blob[i:i + n]; i += n
        j = 0; key = None; val = [] if field == 4 else None
        while j < len(entry):
            t, j = rd(entry, j); f2, w2 = t >> 3, t & 7
            if w2 == 2:
                k, j = rd(entry, j); payload = entry[j:j + k]; j += k
                if f2 == 1: key = payload.decode()
                else: val = payload
            else:
                v, j = rd(entry, j)
                if v >= 1 << 63: v -= 1 << 64
                if field == 4: val.append(v)
                else: val = v
        out.setdefault(field, {})[key] = val
    return out


def _msb_pack(bits):
    """IntermediateData's writer (CanvasBin.cs:1052-1072): most significant bit of every byte first; a last partial byte keeps its bits in the low positions"""
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i >> 3] = (out[i >> 3] * 2 + int(b)) & 0xFF
    return bytes(out)


def _lsb_unpack(data, bits_in_last_byte):
    """IntermediateData.Convert (CanvasBin.cs:1106-1135): new BitArray(bytes) is least significant bit first"""
    n = 8 * (len(data) - 1) + bits_in_last_byte if bits_in_last_byte > 0 else 8 * len(data)
    return np.array([(data[i >> 3] >> (i & 7)) & 1 for i in range(n)], bool)


def _kept(r, paired):
    f = r["flag"]
    if f & 0x4 or f & 0x200 or f & 0x400 or f & 0x10 or f & 0x900: return False
    ln, op = r["cigar"][0]
    if op != "M" or ln < 35: return False
    if paired and not f & 0x2: return False
    return True


def test_canvasbin_bam_to_binned(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from canvas_amd import build
    build.build(); build.build_tools()
    rng = np.random.RandomState(20260927)
    refs = [("chr1", 260_000), ("chr2", 180_000), ("chrEmpty", 5_000), ("chrX", 120_000)]
    # kmer.fa: upper case = unique k-mer start; lower-case runs and 'n' gaps
    fa = str(tmp_path / "kmer.fa"); seqs = {}
    with open(fa, "w") as f:
        for name, ln in refs:
            b = rng.choice(np.frombuffer(b"ACGT", np.uint8), ln, p=[0.3, 0.2, 0.2, 0.3])
            low = np.zeros(ln, bool)
            p = 0
            while p < ln:
                run = rng.geometric(1 / 3000.0); p += run
                gap = rng.geometric(1 / 600.0); low[p:p + gap] = True; p += gap
            b = np.where(low, b | 0x20, b).astype(np.uint8)
            b[:2000] = ord("n"); b[ln // 2:ln // 2 + 1500] = ord("n")
            seqs[name] = b
            f.write(f">{name} some description\n")
            s = b.tobytes().decode()
            for i in range(0, ln, 70): f.write(s[i:i + 70] + "\n")
    # reads: ~0.25 per base, a mix of flags / cigars that the filters must drop
    reads = []
    for ri, (name, ln) in enumerate(refs):
        if name == "chrEmpty": continue
        n = int(ln * 0.25)
        pos = np.sort(rng.randint(0, ln - 40, n))
        pos[: n // 50] = pos[n // 50]                      # a pile-up that saturates the byte counter at 255
        pos = np.sort(pos)
        for p in pos:
            u = rng.rand()
            flag = 0x1 | 0x2 | 0x40
            cigar = [(36, "M")]
            if u < 0.05: flag |= 0x10
            elif u < 0.07: flag |= 0x400
            elif u < 0.08: flag |= 0x200
            elif u < 0.09: flag |= 0x100
            elif u < 0.10: flag |= 0x800
            elif u < 0.12: cigar = [(5, "S"), (31, "M")]
            elif u < 0.14: cigar = [(20, "M"), (2, "I"), (14, "M")]
            elif u < 0.16: flag &= ~0x2
            reads.append(dict(ref=ri, pos=int(p), flag=flag, cigar=cigar, tlen=int(rng.randint(200, 500))))
    reads.append(dict(ref=-1, pos=-1, flag=0x4 | 0x1, cigar=[], tlen=0))
    bam = str(tmp_path / "S.bam")
    _write_bam(bam, refs, reads)
    bed = str(tmp_path / "filter.bed")
    excl = {"chr1": [(50_000, 52_000), (199_990, 200_500)], "chrX": [(10, 3_000)]}
    with open(bed, "w") as f:
        for c, ivs in excl.items():
            for a, b in ivs: f.write(f"{c}\t{a}\t{b}\n")

    true_masks = {}

    def expected_arrays(paired, mode):
        masks, hits = {}, {}
        for ri, (name, ln) in enumerate(refs):
            m = (seqs[name] >= ord("A")) & (seqs[name] <= ord("Z"))
            for a, b in excl.get(name, []): m[a:b] = False
            h = np.zeros(ln, np.int64)
            ps = np.array([r["pos"] for r in reads if r["ref"] == ri and _kept(r, paired)], np.int64)
            np.add.at(h, ps, 1)
            h = np.minimum(h, 1 if mode == 0 else 255)
            h[~m] = 0                                     # ScreenObservedTags runs BEFORE the .dat round trip, with the mask as computed (CanvasBin.cs:780)
            # ... and the bins are made from what CanvasBin -i reads back: the writer packs MSB-first, the reader unpacks LSB-first
            m_read = _lsb_unpack(_msb_pack(m), ln % 8)
            assert len(m_read) == ln
            true_masks[name] = m
            masks[name] = np.packbits(m_read, bitorder="little"); hits[name] = h.astype(np.uint8)
        return masks, hits

    for paired, mode, mflag in ((True, 3, "TruncatedDynamicRange"), (False, 0, "0")):
        dats = []
        for name, ln in refs:
            dat = str(tmp_path / f"{name}.{mode}.dat")
            cmd = [BIN, "-b", bam, "-r", fa, "-c", name, "-o", dat, "-d", "100", "-f", bed, "-m", mflag] + (["-p"] if paired else [])
            r = subprocess.run(cmd, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            dats += ["-i", dat]
        binned = str(tmp_path / f"S.{mode}.binned")
        r = subprocess.run([BIN, "-b", bam, "-r", fa, "-o", binned, "-d", "100", "-m", mflag] + dats, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        masks, hits = expected_arrays(paired, mode)
        names = [n for n, _ in refs]
        rates = [O.bin_rate(hits[n], masks[n]) for n in ("chr1", "chr2")]                 # autosomes only; chrEmpty / chrX are not
        bs = O.bin_size(rates, 100)
        res = O.bin_genome([seqs[n] for n in names], [masks[n] for n in names], [hits[n] for n in names], bs, mode=mode, threads=2)
        exp = []
        for c, n in enumerate(names):
            for s, e, g, k in zip(res[0][c], res[1][c], res[2][c], res[3][c]):
                exp.append(f"{n}\t{s}\t{e}\t{O.format_f2(float(k))}\t{g}")
        with gzip.open(binned, "rt") as f:
            got = f.read().splitlines()
        assert len(got) > 50 and got == exp
        # the intermediate files are the reference's protobuf-net encoding of IntermediateData: decoded here by an independent reader
        for name, ln in refs:
            d = _decode_dat(open(str(tmp_path / f"{name}.{mode}.dat"), "rb").read())
            assert set(d) == {1, 2, 3} and list(d[1]) == [name] and list(d[2]) == [name]
            assert d[3][name] == ln % 8 and d[1][name] == _msb_pack(true_masks[name]) and d[2][name] == hits[name].tobytes()
        # -y: bin size only, written without a newline (CanvasBin.cs:926-928)
        r = subprocess.run([BIN, "-b", bam, "-r", fa, "-o", binned, "-d", "100", "-y", "-m", mflag] + dats, capture_output=True, text=True)
        assert r.returncode == 0 and open(binned + ".binsize").read() == str(bs)
    # ---- a .dat that this tool did not write: hand-encoded per the wire specification (what a C# CanvasBin -c leaves behind), incl. the variants a protobuf
    # writer may choose (packed fragment lengths, an omitted zero) and -m GCContentWeighted with its fragment-length member
    rng2 = np.random.RandomState(5)
    fa2 = str(tmp_path / "two.fa"); lens2 = {"chrA": 40_003, "chrB": 24_000}
    seq2 = {n: rng2.choice(np.frombuffer(b"ACGTacgt", np.uint8), L, p=[0.2, 0.2, 0.2, 0.2, 0.05, 0.05, 0.05, 0.05]) for n, L in lens2.items()}
    with open(fa2, "w") as f:
        for n, b in seq2.items():
            f.write(f">{n}\n" + b.tobytes().decode() + "\n")
    for mode, mflag in ((3, "3"), (5, "GCContentWeighted")):
        args = []; mk, hk, fk = [], [], []
        for k, (n, L) in enumerate(lens2.items()):
            stored = rng2.randint(0, 256, (L + 7) // 8).astype(np.uint8)                   # the bytes as they sit in the file
            m_read = _lsb_unpack(stored.tobytes(), L % 8)
            h = (rng2.poisson(0.3, L) * m_read).astype(np.uint8)
            fl = np.where(h > 0, rng2.randint(150, 500, L), 0).astype(np.int16)
            dat = str(tmp_path / f"hand.{n}.{mode}.dat")
            open(dat, "wb").write(_encode_dat(n, stored.tobytes(), h.tobytes(), L % 8, frag=fl if mode == 5 else None, packed=(k == 1), omit_zero=True))
            args += ["-i", dat]; mk.append(np.packbits(m_read, bitorder="little")); hk.append(h); fk.append(fl)
        binned = str(tmp_path / f"hand.{mode}.binned")
        r = subprocess.run([BIN, "-b", bam, "-r", fa2, "-o", binned, "-d", "100", "-z", "150", "-m", mflag] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        names2 = list(lens2)
        if mode == 3:
            args_mode3, mk3, hk3 = list(args), mk, hk
            res = O.bin_genome([seq2[n] for n in names2], mk, hk, 150, mode=3, threads=2)
            exp = [f"{n}\t{s}\t{e}\t{O.format_f2(float(k))}\t{g}" for c, n in enumerate(names2) for s, e, g, k in zip(res[0][c], res[1][c], res[2][c], res[3][c])]
        else:
            resw, _, _, _ = O.bin_gc_weighted([seq2[n] for n in names2], mk, hk, fk, 150)
            exp = [f"{n}\t{s}\t{e}\t{O.format_f2(float(k))}\t{g}" for c, n in enumerate(names2) for s, e, g, k in zip(resw[c][0], resw[c][1], resw[c][2], resw[c][3])]
        with gzip.open(binned, "rt") as f:
            assert f.read().splitlines() == exp
    # ---- predefined bins (-n): counts and GC of given intervals from the same intermediates (BinCountsForChromosome with usePredefinedBins, CanvasBin.cs:575-655)
    pre = str(tmp_path / "predefined.bed")
    pre_bins = {"chrA": [(0, 700), (700, 1500), (1400, 2000), (30_000, 40_003)], "chrB": [(5, 6), (100, 24_000)]}          # touching, overlapping, to the last base; chrB's first bin is 1 base
    with open(pre, "w") as f:
        for n, bl in pre_bins.items():
            for a_, b_ in bl: f.write(f"{n}\t{a_}\t{b_}\n")
    binned = str(tmp_path / "pre.binned")
    r = subprocess.run([BIN, "-b", bam, "-r", fa2, "-o", binned, "-d", "100", "-m", "3", "-n", pre] + args_mode3, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = []
    for c, n in enumerate(names2):
        k, g, cnt = O.bin_predefined(seq2[n], mk3[c], hk3[c], [b[0] for b in pre_bins[n]], [b[1] for b in pre_bins[n]], 3)
        assert k == len(pre_bins[n])
        exp += [f"{n}\t{a_}\t{b_}\t{O.format_f2(float(cc))}\t{gg}" for (a_, b_), gg, cc in zip(pre_bins[n], g, cnt)]
    with gzip.open(binned, "rt") as f:
        assert f.read().splitlines() == exp
    # -n with -m GCContentWeighted (CanvasBin.cs:617-636): bins on chrA only — chrB still enters the fragment mean, the read-GC profile and the weights (:427-505)
    pre5 = str(tmp_path / "predefined5.bed")
    with open(pre5, "w") as f:
        for a_, b_ in pre_bins["chrA"]: f.write(f"chrA\t{a_}\t{b_}\n")
    binned = str(tmp_path / "pre5.binned")
    r = subprocess.run([BIN, "-b", bam, "-r", fa2, "-o", binned, "-d", "100", "-m", "GCContentWeighted", "-n", pre5] + args, capture_output=True, text=True)      # (args, mk, hk, fk: the mode-5 intermediates)
    assert r.returncode == 0, r.stdout + r.stderr
    res5 = O.bin_predefined_gc_weighted([seq2[n] for n in names2], mk, hk, fk, [[b[0] for b in pre_bins["chrA"]], []], [[b[1] for b in pre_bins["chrA"]], []])
    assert res5[0][0] == len(pre_bins["chrA"])
    exp = [f"chrA\t{a_}\t{b_}\t{O.format_f2(float(cc))}\t{gg}" for (a_, b_), gg, cc in zip(pre_bins["chrA"], res5[0][1], res5[0][2])]
    with gzip.open(binned, "rt") as f:
        assert f.read().splitlines() == exp
    # a truncated file is refused
    bad = str(tmp_path / "bad.dat"); open(bad, "wb").write(_encode_dat("chrA", b"\xff" * 10, b"\x01" * 80, 0)[:-7])
    assert subprocess.run([BIN, "-b", bam, "-r", fa2, "-o", str(tmp_path / "x.binned"), "-d", "100", "-i", bad], capture_output=True).returncode == 1
    # error conventions (Program.cs:108-170)
    assert subprocess.run([BIN], capture_output=True).returncode == 1
    # -j: the reference's RunMultiSample works on an empty sample list (CanvasBin.cs:936-944): exit 0, nothing written; missing file / -i together with -j: exit 1
    js = str(tmp_path / "samples.json"); open(js, "w").write("{}")
    r = subprocess.run([BIN, "-b", bam, "-r", fa, "-o", str(tmp_path / "multi.binned"), "-d", "100", "-j", js], capture_output=True)
    assert r.returncode == 0 and not os.path.exists(str(tmp_path / "multi.binned"))
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-o", str(tmp_path / "multi.binned"), "-d", "100", "-j", str(tmp_path / "absent.json")], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-o", str(tmp_path / "multi.binned"), "-d", "100", "-j", js, "-i", bad], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", str(tmp_path / "none.bam"), "-r", fa, "-c", "chr1", "-o", str(tmp_path / "x.dat"), "-d", "100"], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-c", "chr1", "-o", str(tmp_path / "x.dat"), "-d", "0"], capture_output=True).returncode == 1


# ---- Fragment mode (-m Fragment): FragmentBinner.BinTask.BinOneAlignment restated in Python (FragmentBinner.cs:256-369)
def _py_fragment_counts(bins, reads, quality_threshold=3):
    count = [0] * len(bins); name_to_bin = {}; same_pos = set(); usable = 0; start_idx = 0
    for r in reads:
        f = r["flag"]
        if f & 0x4 or f & 0x8 or f & 0x100 or not (f & 0x1 and f & 0x2):
            continue
        bad = bool(f & 0x400 or f & 0x200 or r["mapq"] == 255 or r["mapq"] < quality_threshold)
        if r["name"] in name_to_bin:
            if bad:
                usable -= 1; count[name_to_bin[r["name"]]] -= 1
            del name_to_bin[r["name"]]
            continue
        if bad or r["ref"] != r["mate_ref"] or r["pos"] > r["mate_pos"]:
            continue
        if r["pos"] == r["mate_pos"]:
            if r["name"] in same_pos:
                same_pos.remove(r["name"]); continue
            same_pos.add(r["name"])
        if r["tlen"] == 0:
            continue
        fs, fe = r["pos"], r["pos"] + r["tlen"]
        while start_idx < len(bins) and bins[start_idx][1] <= fs:
            start_idx += 1
        if start_idx >= len(bins):
            continue
        best, best_ov = -1, 0
        for i in range(start_idx, len(bins)):
            ov = min(bins[i][1], fe) - max(bins[i][0], fs)
            if ov <= 0: break
            if ov > best_ov: best_ov, best = ov, i
        if best >= 0:
            usable += 1; count[best] += 1; name_to_bin[r["name"]] = best
    return count, usable


def test_canvasbin_fragment_mode(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    refs = [("chr1", 50_000), ("chr2", 30_000), ("chr3", 10_000)]
    fa = str(tmp_path / "genome.fa")
    rng = np.random.RandomState(3)
    seqs = {n: rng.choice(np.frombuffer(b"ACGTn", np.uint8), L, p=[0.27, 0.2, 0.2, 0.27, 0.06]) for n, L in refs}
    with open(fa, "w") as f:
        for n, _ in refs: f.write(f">{n}\n" + seqs[n].tobytes().decode() + "\n")
    # ---- the reference's own unit test (TestCanvasBin.cs:14-78): one bin chr1:100-200, one pair at (pos1, pos2), four quality combinations -> count 1, 0, 0, 0
    bed1 = str(tmp_path / "one.bed"); open(bed1, "w").write("chr1\t100\t200\t50\n")
    for pos1, pos2 in ((100, 120), (100, 100)):
        for q1, q2, expect in ((10, 10, 1), (10, 2, 0), (2, 10, 0), (2, 2, 0)):
            reads = [dict(ref=0, pos=pos1, flag=0x1 | 0x2, cigar=[(36, "M")], name="ReadName", mate_ref=0, mate_pos=pos2, tlen=100, mapq=q1),
                     dict(ref=0, pos=pos2, flag=0x1 | 0x2, cigar=[(36, "M")], name="ReadName", mate_ref=0, mate_pos=pos1, tlen=-100, mapq=q2)]
            bam = str(tmp_path / f"ut_{pos1}_{pos2}_{q1}_{q2}.bam"); _write_bam(bam, refs, reads)
            out = str(tmp_path / "ut.binned")
            r = subprocess.run([BIN, "-b", bam, "-r", fa, "-n", bed1, "-o", out, "-m", "Fragment", "-p"], capture_output=True, text=True)
            if expect == 0:
                assert r.returncode == 1 and "No passing-filter fragments" in r.stderr            # usableFragmentCount == 0 (FragmentBinner.cs:63-67)
            else:
                assert r.returncode == 0, r.stdout + r.stderr
                assert gzip.open(out, "rt").read().splitlines() == ["chr1\t100\t200\t1.00\t50"]
    # ---- a randomised sample: pairs with duplicates / QC failures / low mapping quality on either mate, same-position mates, improper pairs, fragments between and across bins
    bins = {"chr1": [], "chr2": []}
    for n, L in (("chr1", 50_000), ("chr2", 30_000)):
        p = 500
        while p + 900 < L:
            size = int(rng.randint(120, 800)); bins[n].append((p, p + size)); p += size + int(rng.choice([0, 0, 37, 400]))
    bed = str(tmp_path / "bins.bed")
    with open(bed, "w") as f:
        for n in ("chr2", "chr1"):                                   # BED order differs from the BAM's: the output follows the BAM header (FragmentBinner.cs:69-75)
            for k, (a_, b_) in enumerate(bins[n]):
                f.write(f"{n}\t{a_}\t{b_}" + ("\t-1" if n == "chr2" else f"\t{40 + k % 20}") + "\n")      # chr2: GC missing (-1) -> PopulateBinGC
    reads = []
    for ri, (n, L) in enumerate(refs[:2]):
        for k in range(6000):
            pos = int(rng.randint(0, L - 700)); tlen = int(rng.choice([0, 150, 300, 450, 600], p=[0.02, 0.2, 0.4, 0.28, 0.1])); mpos = pos + max(0, tlen - 36) if rng.rand() > 0.03 else pos
            name = f"q{ri}_{k}"
            f1 = f2 = 0x1 | 0x2
            u = rng.rand()
            if u < 0.05: f1 |= 0x400
            elif u < 0.10: f2 |= 0x400
            elif u < 0.13: f1 |= 0x200
            elif u < 0.16: f2 &= ~0x2
            elif u < 0.18: f1 |= 0x100
            elif u < 0.20: f2 |= 0x8
            q1, q2 = (int(rng.choice([0, 2, 3, 30, 60, 255], p=[0.03, 0.03, 0.04, 0.45, 0.42, 0.03])) for _ in range(2))
            reads.append(dict(ref=ri, pos=pos, flag=f1 | 0x40, cigar=[(36, "M")], name=name, mate_ref=ri, mate_pos=mpos, tlen=tlen, mapq=q1))
            reads.append(dict(ref=ri, pos=mpos, flag=f2 | 0x80 | 0x10, cigar=[(36, "M")], name=name, mate_ref=ri, mate_pos=pos, tlen=-tlen, mapq=q2))
    reads.sort(key=lambda r: (r["ref"], r["pos"]))
    bam = str(tmp_path / "pairs.bam"); _write_bam(bam, refs, reads)
    out = str(tmp_path / "pairs.binned")
    r = subprocess.run([BIN, "-b", bam, "-r", fa, "-n", bed, "-o", out, "-m", "Fragment", "-p"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = []; total_usable = 0
    for ri, n in enumerate(("chr1", "chr2")):
        cnt, usable = _py_fragment_counts(bins[n], [x for x in reads if x["ref"] == ri]); total_usable += usable
        for k, ((a_, b_), c_) in enumerate(zip(bins[n], cnt)):
            if n == "chr1": gc = 40 + k % 20
            else:
                seg = seqs[n][a_:b_]; nt = int((seg != ord("n")).sum()); g = int(np.isin(seg, np.frombuffer(b"CGcg", np.uint8)).sum()); gc = int(100 * g / nt) if nt else 0
            exp.append(f"{n}\t{a_}\t{b_}\t{O.format_f2(float(c_))}\t{gc}")
    assert total_usable > 1000
    assert gzip.open(out, "rt").read().splitlines() == exp
    # error conventions: no -n, no -p, a chromosome of the BED file that the BAM does not have
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-o", out, "-m", "Fragment", "-p"], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-n", bed, "-o", out, "-m", "Fragment"], capture_output=True).returncode == 1
    bedU = str(tmp_path / "u.bed"); open(bedU, "w").write("chrU\t1\t100\n")
    r = subprocess.run([BIN, "-b", bam, "-r", fa, "-n", bedU, "-o", out, "-m", "Fragment", "-p"], capture_output=True, text=True)
    assert r.returncode == 1 and "Not all chromosomes" in r.stderr


def _read_bam(path):
    """independent reader of a BAM file (BGZF = concatenated gzip members; SAM specification section 4): ([(name, length)], [read dicts as _kept() wants them])"""
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4); p = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, p); p += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, p); p += 4
        name = raw[p:p + l_name - 1].decode(); p += l_name
        l_ref, = struct.unpack_from("<i", raw, p); p += 4
        refs.append((name, l_ref))
    reads = []
    while p < len(raw):
        block_size, = struct.unpack_from("<i", raw, p); p += 4
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHiiii", raw, p)
        q = p + 32 + l_read_name
        cigar = []
        for k in range(n_cigar):
            v, = struct.unpack_from("<I", raw, q + 4 * k); cigar.append((v >> 4, "MIDNSHP=X"[v & 15]))
        reads.append(dict(ref=ref_id, pos=pos, flag=flag, mapq=mapq, cigar=cigar, tlen=tlen))
        p += block_size
    return refs, reads


def test_canvasbin_on_the_references_own_bam(tmp_path):
    """CanvasBin -c chrM on CanvasTest/Data/single-end.bam (the reference's fixture, tests/golden/ref_data): the hit array of the intermediate file must hold one hit per
    read that passes the reference's filters (CanvasBin.cs:239-270), as counted by an independent BAM reader"""
    bam = os.path.join(ROOT, "tests", "golden", "ref_data", "single-end.bam")
    refs, reads = _read_bam(bam)
    names = [n for n, _ in refs]
    assert "chrM" in names
    cid = names.index("chrM"); L = refs[cid][1]
    rng = np.random.RandomState(4)
    seq = rng.choice(np.frombuffer(b"ACGTacgt", np.uint8), L)
    fa = str(tmp_path / "kmer.fa")
    with open(fa, "wb") as f:
        f.write(b">chrM\n")
        for i in range(0, L, 70): f.write(seq[i:i + 70].tobytes() + b"\n")
    dat = str(tmp_path / "chrM.dat")
    r = subprocess.run([BIN, "-b", bam, "-r", fa, "-c", "chrM", "-o", dat, "-d", "100", "-m", "TruncatedDynamicRange"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _decode_dat(open(dat, "rb").read())
    exp = np.zeros(L, np.int64)
    kept = [x for x in reads if x["ref"] == cid and x["cigar"] and _kept(x, False)]
    for x in kept: exp[x["pos"]] += 1
    mask = (seq >= ord("A")) & (seq <= ord("Z"))
    exp = np.where(mask, np.minimum(exp, 255), 0)                       # ScreenObservedTags (CanvasBin.cs:694-716): hits outside the possible positions are dropped
    got = np.frombuffer(d[2]["chrM"], np.uint8)
    assert len(got) == L and (got == exp).all()
    # the fixture holds nine chrM reads and every one of them starts with a soft clip: the reference's "first CIGAR operation is a match of 35 and more" rule keeps none
    assert len(reads) == 9 and len(kept) == 0 and int(got.sum()) == 0
    assert "Kept 0 of 9 total reads" in r.stdout
