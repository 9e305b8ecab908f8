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
            name = b"r\x00"
            cig = b"".join(struct.pack("<I", (ln << 4) | "MIDNSHP=X".index(op)) for ln, op in r["cigar"])
            lseq = 36
            body = struct.pack("<iiBBHHHiiii", r["ref"], r["pos"], len(name), 30, 4680, len(r["cigar"]), r["flag"], lseq, -1, -1, r.get("tlen", 0))
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

    def expected_arrays(paired, mode):
        masks, hits = {}, {}
        for ri, (name, ln) in enumerate(refs):
            m = (seqs[name] >= ord("A")) & (seqs[name] <= ord("Z"))
            for a, b in excl.get(name, []): m[a:b] = False
            h = np.zeros(ln, np.int64)
            ps = np.array([r["pos"] for r in reads if r["ref"] == ri and _kept(r, paired)], np.int64)
            np.add.at(h, ps, 1)
            h = np.minimum(h, 1 if mode == 0 else 255)
            h[~m] = 0
            masks[name] = np.packbits(m, bitorder="little"); hits[name] = h.astype(np.uint8)
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
        # -y: bin size only, written without a newline (CanvasBin.cs:926-928)
        r = subprocess.run([BIN, "-b", bam, "-r", fa, "-o", binned, "-d", "100", "-y", "-m", mflag] + dats, capture_output=True, text=True)
        assert r.returncode == 0 and open(binned + ".binsize").read() == str(bs)
    # error conventions (Program.cs:108-170)
    assert subprocess.run([BIN], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", str(tmp_path / "none.bam"), "-r", fa, "-c", "chr1", "-o", str(tmp_path / "x.dat"), "-d", "100"], capture_output=True).returncode == 1
    assert subprocess.run([BIN, "-b", bam, "-r", fa, "-c", "chr1", "-o", str(tmp_path / "x.dat"), "-d", "0"], capture_output=True).returncode == 1
