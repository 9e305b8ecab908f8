"""The two tests of the reference that run on its own data files (CanvasTest/TestCanvasBin.cs:80-125, files under CanvasTest/Data, copied as fixtures to
tests/golden/ref_data): Fragment-mode binning of a single-end BAM must stop with "No paired alignments found", and predefined bins on a chromosome the BAM does not
have must stop with "Not all chromosomes in <bed> are found in <bam>.".  Through the drop-in executable; Fragment mode is host code, no GPU involved."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "canvas_amd", "bin", "CanvasBin")
DATA = os.path.join(ROOT, "tests", "golden", "ref_data")


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(BIN):
        import sys
        sys.path.insert(0, ROOT)
        from canvas_amd import build
        build.build_tools()
    assert os.path.exists(BIN)


def _run(bed, tmp_path):
    ref = tmp_path / "dummy.fa"; ref.write_text(">chrM\nACGT\n")           # FragmentBinner never opens the reference; the option parser wants the file to exist
    bam = os.path.join(DATA, "single-end.bam")
    return subprocess.run([BIN, "-b", bam, "-r", str(ref), "-n", os.path.join(DATA, bed), "-o", str(tmp_path / "out.binned"), "-m", "Fragment", "-p"], capture_output=True, text=True), bam


def test_bin_single_end_bam(tmp_path):
    """TestBinSingleEndBam (TestCanvasBin.cs:80-101)"""
    r, bam = _run("bins_chrM.bed", tmp_path)
    assert r.returncode != 0
    assert "No paired alignments found" in r.stderr + r.stdout
    assert not os.path.exists(tmp_path / "out.binned")


def test_all_chroms_in_bed_are_in_bam(tmp_path):
    """TestAllChromsInBedAreInBam (TestCanvasBin.cs:103-125)"""
    r, bam = _run("bins_chrU.bed", tmp_path)
    assert r.returncode != 0
    assert "Not all chromosomes in %s are found in %s." % (os.path.join(DATA, "bins_chrU.bed"), bam) in r.stderr + r.stdout
