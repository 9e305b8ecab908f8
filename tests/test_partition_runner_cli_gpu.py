"""CanvasPartition under the command lines the UNMODIFIED orchestrator issues (Canvas/CanvasRunner.cs:939-971 InvokeCanvasPartition, :904-937
InvokeCanvasPartitionMultisample): -p is always there, Somatic-WGS adds --evenness-metric-file, pedigree runs add -c.  The strings below are built
exactly as the C# builds them (same order, same quoting, same double spaces) and split the way a process launcher splits them."""
import gzip
import os
import shlex
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from canvas_amd import synth
from gpu_common import get_canvas, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "canvas_amd", "bin", "CanvasPartition")
NAMES = synth.CHROM_NAMES


def _sample(seed, n, nchr=24, exact_text=True):
    bins = synth.generate_bins(seed, n, nchr=nchr)
    if exact_text:      # the double CanvasPartition parses from the F2 text of the float count
        cov = np.array([float(O.format_f2(float(v))) for v in bins["count"]])
    else:
        cov = np.round(bins["count"].astype(np.float64), 2)
    return bins, cov


def _write_cleaned(path, bins, cov):
    with gzip.open(path, "wt") as f:
        for c, s, e, v, g in zip(bins["chr"], bins["start"], bins["stop"], cov, bins["gc"]):
            f.write(f"{NAMES[c]}\t{s}\t{e}\t{O.format_f2(float(np.float32(v)))}\t{g}\n")


def _read(path):
    with gzip.open(path, "rt") as f:
        return f.read().splitlines()


def _ploidy_vcf(path, bins, nchr, male=True):
    """what Canvas' ploidy VCF looks like for a male sample: chrX and chrY haploid outside the pseudo-autosomal stretches.  The records are placed so
    that some boundaries fall inside a bin, some between two bins and one exactly on a bin end."""
    x, y = nchr - 2, nchr - 1
    recs = {}
    for c in (x, y):
        idx = np.nonzero(bins["chr"] == c)[0]
        s, e = bins["start"][idx], bins["stop"][idx]
        k1, k2, k3 = len(idx) // 10, len(idx) // 2, 9 * len(idx) // 10
        # PAR1 diploid [1, mid of bin k1]; haploid up to the end of bin k2 EXACTLY; diploid island until the gap after bin k3; haploid to the end
        a = int((s[k1] + e[k1]) // 2)
        recs[c] = [(a + 1, int(e[k2]), 1), (int(e[k2]) + 1, int(e[k3]) + 3, "."), (int(e[k3]) + 4, int(e[-1]) + 1000, 1 if c == x else 0)]
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.1\n##INFO=<ID=END,Number=1,Type=Integer,Description=\"End\">\n##FORMAT=<ID=CN,Number=1,Type=Integer,Description=\"CN\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE1\n")
        for c in (x, y):
            for a, b, cn in recs[c]:
                f.write(f"{NAMES[c]}\t{a}\t.\tN\t<CNV>\t.\tPASS\tEND={b}\tCN\t{cn}\n")
    ploidy = [None] * nchr
    for c in (x, y):
        ploidy[c] = (np.array([r[0] for r in recs[c]], np.int32), np.array([r[1] for r in recs[c]], np.int32), np.array([2 if r[2] == "." else r[2] for r in recs[c]], np.int32))
    return ploidy


def _filter_bed(path, bins, nchr, seed):
    rng = np.random.RandomState(seed)
    excl = {}
    with open(path, "w") as f:
        for c in (0, 3, nchr - 2):
            idx = np.nonzero(bins["chr"] == c)[0]
            pick = np.sort(rng.choice(len(idx) - 2, 6, replace=False))
            a = bins["stop"][idx[pick]] + 1; b = a + 50
            excl[c] = (a.astype(np.int32), b.astype(np.int32))
            for x_, y_ in zip(a, b):
                f.write(f"{NAMES[c]}\t{x_}\t{y_}\n")
    return excl


def _after_filter(bins, cov, excl, nchr):
    chrs, st, en = bins["chr"], bins["start"].astype(np.uint32), bins["stop"].astype(np.uint32)
    keep = np.ones(len(chrs), bool)
    for c, (a, b) in excl.items():
        for x_, y_ in zip(a, b):
            keep &= ~((chrs == c) & (st < y_) & (en > x_))
    chrs, st, en, cov = chrs[keep], st[keep], en[keep], cov[keep]
    off = np.concatenate([[0], np.cumsum(np.bincount(chrs, minlength=nchr))]).astype(np.int64)
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    bs = [np.ascontiguousarray(st[off[c]:off[c + 1]]) for c in range(nchr)]; be = [np.ascontiguousarray(en[off[c]:off[c + 1]]) for c in range(nchr)]
    ex_list = [excl.get(c, (np.zeros(0, np.int32), np.zeros(0, np.int32))) for c in range(nchr)]
    return per, bs, be, ex_list, off


def _rows(per, bs, be, ids, nchr):
    return [f"{NAMES[c]}\t{s_}\t{e_}\t{O.format_g15(float(v))}\t{i}" for c in range(nchr) for s_, e_, v, i in zip(bs[c], be[c], per[c], ids[c])]


def _run(cmdline):
    r = subprocess.run([EXE] + shlex.split(cmdline), capture_output=True, text=True)
    return r


def test_germline_and_somatic_wgs_command_lines(tmp_path):
    get_canvas()
    nchr = 24
    bins, cov = _sample(20260927 + 50, 300_000, nchr)
    cleaned = tmp_path / "S.cleaned"; part = tmp_path / "S.partitioned"; snv = tmp_path / "VFResultsS.txt.gz"; bed = tmp_path / "filter.bed"; vcf = tmp_path / "ploidy.vcf"
    ref = tmp_path / "WholeGenomeFasta"; ref.mkdir()
    _write_cleaned(str(cleaned), bins, cov); open(snv, "w").write("")
    excl = _filter_bed(str(bed), bins, nchr, 4)
    ploidy = _ploidy_vcf(str(vcf), bins, nchr)
    per, bs, be, ex_list, off = _after_filter(bins, cov, excl, nchr)

    def expected(germline):
        bps = O.wavelets_genome(per, is_germline=germline)
        wstarts = [bs[c][bps[c]].astype(np.uint32) if (len(bps[c]) >= 2 and len(bs[c]) > 10) else bs[c][:1].astype(np.uint32) for c in range(nchr)]
        ids, last = O.postprocess_ploidy(bs, be, wstarts, ex_list, ploidy)
        ids0, last0 = O.postprocess(bs, be, wstarts, ex_list)
        assert last > last0                                   # the ploidy boundaries really split segments
        return _rows(per, bs, be, ids, nchr)

    # ---- Germline-WGS (CanvasRunner.cs:943-953): " -v {snv} -i "{cleaned}" -b "{bed}" -o "{out}"  -r "{ref}"  -p "{ploidy}"  -g"
    cmd = f" -v {snv} " + f"-i \"{cleaned}\" " + f"-b \"{bed}\" " + f"-o \"{part}\" " + f" -r \"{ref}\" " + f" -p \"{vcf}\" " + " -g"
    r = _run(cmd)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _read(str(part)) == expected(True)
    # ---- Somatic-WGS (:954-961): no -g, plus --evenness-metric-file "{SampleOutputFolder}/EvennessMetric.txt"
    ev = tmp_path / "EvennessMetric.txt"
    cmd = f" -v {snv} " + f"-i \"{cleaned}\" " + f"-b \"{bed}\" " + f"-o \"{part}\" " + f" -r \"{ref}\" " + f" -p \"{vcf}\" " + f"--evenness-metric-file \"{ev}\" "
    r = _run(cmd)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _read(str(part)) == expected(False)
    score = O.evenness_score(per, 100000)
    assert score is None and not ev.exists()                 # 300 k bins: no chromosome reaches a 100000-bin window -> Median throws -> no file (WaveletsRunner.cs:58-67)
    cfg = tmp_path / "CanvasPartitionParameters.json"; open(cfg, "w").write('{"EvennessScoreWindow": 1500}')
    r = _run(cmd + f"--config {cfg}")
    assert r.returncode == 0, r.stdout + r.stderr
    score = O.evenness_score(per, 1500)
    assert score is not None and open(ev).read() == "#evenness\t" + O.format_g15(score) + "\n"
    # ---- -p pointing nowhere (CanvasPartition.cs:96-100) and a VCF with two sample columns (PloidyInfo.cs:117-125)
    assert _run(cmd.replace(str(vcf), str(tmp_path / "missing.vcf"))).returncode == 1
    two = tmp_path / "two.vcf"; open(two, "w").write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\nchrX\t1\t.\tN\t<CNV>\t.\tPASS\tEND=10\tCN\t1\t1\n")
    assert _run(cmd.replace(str(vcf), str(two))).returncode == 1


def test_small_pedigree_command_line(tmp_path):
    """CanvasRunner.cs:904-937: -i per sample, -b, -c commonCnvs.bed, -o per sample, -r, -m PerSampleHMM (no -p: the multisample ploidy VCF goes to the caller)"""
    get_canvas()
    nchr = 24
    bins, cov1 = _sample(20260927 + 51, 50_000, nchr)
    rng = np.random.RandomState(5)
    cov2 = np.array([float(O.format_f2(float(np.float32(max(0.0, v * 0.9 + rng.normal(0, 3)))))) for v in cov1])
    c1 = tmp_path / "A.cleaned"; c2 = tmp_path / "B.cleaned"; p1 = tmp_path / "A.partitioned"; p2 = tmp_path / "B.partitioned"
    bed = tmp_path / "filter.bed"; common = tmp_path / "commonCnvs.bed"; ref = tmp_path / "WholeGenomeFasta"; ref.mkdir()
    _write_cleaned(str(c1), bins, cov1); _write_cleaned(str(c2), bins, cov2)
    open(common, "w").write("chr1\t1000\t5000\n")
    excl = _filter_bed(str(bed), bins, nchr, 6)
    cmd = f"-i \"{c1}\" " + f"-i \"{c2}\" " + f"-b \"{bed}\" " + f"-c \"{common}\" " + f"-o \"{p1}\" " + f"-o \"{p2}\" " + f"-r \"{ref}\" " + "-m PerSampleHMM"
    r = _run(cmd)
    assert r.returncode == 0, r.stdout + r.stderr
    perA, bs, be, ex_list, off = _after_filter(bins, cov1, excl, nchr)
    perB = _after_filter(bins, cov2, excl, nchr)[0]
    seg = []
    for per in (perA, perB):
        paths, ran = O.hmm_genome_per_sample(per, threads=8)
        seg.append([O.segments_from_path(paths[c], ran[c], bs[c], be[c]) for c in range(nchr)])
    merged = []
    for c in range(nchr):
        if len(seg[0][c][0]) == 0: merged.append(np.zeros(0, np.uint32)); continue      # chromosome skipped by sample 0: no key in its SegmentByChr
        ms, me = O.split_overlapping([seg[0][c][0], seg[1][c][0]], [seg[0][c][1], seg[1][c][1]])
        merged.append(ms)
    ids, _ = O.postprocess(bs, be, merged, ex_list)
    assert _read(str(p1)) == _rows(perA, bs, be, ids, nchr)
    assert _read(str(p2)) == _rows(perB, bs, be, ids, nchr)


def test_segment_ids_with_reference_ploidy_on_device():
    """canvas_segment_ids_ploidy (the in-memory form of the -p branch) vs the oracle's PostProcessSegments"""
    cv = get_canvas()
    nchr = 24
    bins, cov = _sample(20260927 + 52, 60_000, nchr)
    off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=nchr))]).astype(np.int64)
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    bs = [bins["start"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]; be = [bins["stop"][off[c]:off[c + 1]].astype(np.uint32) for c in range(nchr)]
    paths, ran = O.hmm_genome_per_sample(per, threads=8)
    segstarts = [O.segments_from_path(paths[c], ran[c], bs[c], be[c])[0] for c in range(nchr)]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        ploidy = _ploidy_vcf(os.path.join(d, "p.vcf"), bins, nchr)
    # add overlapping / odd records on an autosome: CN 3 island, a CN 0 record overlapping it, a zero-length and a reversed record
    s5, e5 = bs[5], be[5]
    ploidy[5] = (np.array([int(s5[100]) + 7, int(e5[150]), int(s5[300]), int(e5[400]) + 5], np.int32), np.array([int(e5[200]), int(e5[180]) + 1, int(s5[300]) - 1, int(s5[390])], np.int32),
                 np.array([3, 0, 1, 4], np.int32))
    z = (np.zeros(0, np.int32),) * 3
    rng = np.random.RandomState(11)
    excl = []
    for c in range(nchr):
        pick = np.sort(rng.choice(len(bs[c]) - 2, 5, replace=False))
        st = be[c][pick].astype(np.int64) - rng.randint(0, 400, 5); en = st + rng.randint(10, 900, 5)
        order = np.argsort(en, kind="stable")
        excl.append((st[order].astype(np.int32), en[order].astype(np.int32)))
    state = to_dev(np.concatenate(paths), cv.device)
    ds, de = to_dev(bins["start"], cv.device), to_dev(bins["stop"], cv.device)
    for ex in (None, excl):
        ids, last = O.postprocess_ploidy(bs, be, segstarts, ex, ploidy)
        seg, nseg = cv.segment_ids(off, state, ds, de, 1000000, excluded=ex, ploidy=[p if p is not None else z for p in ploidy])
        assert (seg.cpu().numpy() == np.concatenate(ids)).all()
        assert nseg == last + 1
        ids0, last0 = O.postprocess(bs, be, segstarts, ex)
        assert last > last0
    # a ploidy of 5 is an IndexOutOfRangeException in the reference: refused
    from canvas_amd.lib import CanvasError
    bad = [z] * nchr; bad[3] = (np.array([1], np.int32), np.array([10 ** 9], np.int32), np.array([5], np.int32))
    with pytest.raises(CanvasError):
        cv.segment_ids(off, state, ds, de, 1000000, ploidy=bad)


@pytest.mark.parametrize("n,window,kind", [(1_500_000, 100000, "plain"), (400_000, 3000, "plain"), (400_000, 1234, "cnv"), (400_000, 2000, "zeros"), (400_000, 2000, "negative"), (30_000, 100000, "plain")])
def test_evenness_score_on_device(n, window, kind):
    """canvas_evenness_score vs the oracle: bit-identical double (the window sums are sequential in both)"""
    cv = get_canvas()
    nchr = 24
    bins, cov = _sample(20260927 + 53, n, nchr, exact_text=False)
    off = np.concatenate([[0], np.cumsum(np.bincount(bins["chr"], minlength=nchr))]).astype(np.int64)
    cov = cov.copy()
    if kind == "cnv": cov[off[2]:off[2] + 20_000] *= 3.0; cov[off[7]:off[8]] *= 0.5
    if kind == "zeros": cov[off[1]:off[1] + 25_000] = 0.0                       # windows with Sum() == 0: count / 0 = Infinity, dropped (Segmentation.cs:291)
    if kind == "negative": cov[off[4]:off[4] + 12_000] = -3.5; cov[off[9] + 5: off[9] + 900] = float("nan")
    per = [np.ascontiguousarray(cov[off[c]:off[c + 1]]) for c in range(nchr)]
    exp = O.evenness_score(per, window)
    got = cv.evenness_score(to_dev(cov, cv.device), off, window)
    if exp is None: assert got is None
    else: assert got is not None and np.float64(got).tobytes() == np.float64(exp).tobytes(), (got, exp)
    if n >= 400_000: assert exp is not None
    if n == 30_000: assert exp is None
