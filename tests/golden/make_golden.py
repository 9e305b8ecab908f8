#!/usr/bin/env python3
"""Extracts the known-answer DATA (inputs / expected outputs) that the reference's own xUnit tests hold for the
read-depth hot path into small JSON fixtures.  Run in the build container only (reads /root/reference);
the JSON files it writes are committed and are what the tests use.  No reference source text is stored.

  LOESS vectors            <- CanvasTest/TestLoessInterpolator.cs:13-81
  split-overlapping cases  <- CanvasTest/CanvasPartition/GenomeSegmentationResultsTests.cs:14-248
  wavelets known answer    <- CanvasTest/CanvasPartition/WaveletTests.cs:10-92 (530 coverage values, 12 expected breakpoints)
"""
import json, os, re, sys

REF = "/root/reference/Src/Canvas/CanvasTest"
OUT = os.path.dirname(os.path.abspath(__file__))


def arrays(text, name):
    m = re.search(r"double\[\]\s+%s\s*=\s*new double\[\]\s*\{([^}]*)\}" % name, text)
    return [float(v) for v in m.group(1).split(",") if v.strip()]


def loess():
    t = open(f"{REF}/TestLoessInterpolator.cs", encoding="utf-8-sig").read()
    d = {k: arrays(t, k) for k in ("x", "y", "fittedR", "weightedFittedR")}
    d["bandwidth"] = 0.3
    d["tolerance_sum_abs"] = 0.31
    json.dump(d, open(f"{OUT}/loess_r_vectors.json", "w"))
    print("loess:", {k: len(v) if isinstance(v, list) else v for k, v in d.items()})


def split_overlapping():
    t = open(f"{REF}/CanvasPartition/GenomeSegmentationResultsTests.cs", encoding="utf-8-sig").read()
    cases = []
    for m in re.finditer(r"\[Fact\]\s*public void (\w+)\(\)\s*\{(.*?)\n        \}", t, re.S):
        name, body = m.group(1), m.group(2)
        var = {}
        for v in re.finditer(r"var (\w+)\s*=\s*new GenomeSegmentationResults\(new Dictionary<string, Segment\[\]>\s*\{(.*?)\}\);", body, re.S):
            chroms = {}
            parts = re.split(r'\["(\w+)"\]\s*=', v.group(2))
            for name_, seg_text in zip(parts[1::2], parts[2::2]):
                chroms[name_] = [[int(a), int(b)] for a, b in re.findall(r"start\s*=\s*(\d+)\s*,\s*end\s*=\s*(\d+)", seg_text)]
            var[v.group(1)] = chroms
        call = re.search(r"SplitOverlappingSegments\((.*?)\);", body, re.S).group(1)
        used = [n for n in re.findall(r"\w+", call) if n in var]
        exp = re.findall(r"AssertEqualSegmentations\((\w+),\s*segmentationResults\)", body)
        cases.append({"name": name, "samples": [var[u] for u in used], "expected": var[exp[-1]]})
    json.dump(cases, open(f"{OUT}/split_overlapping_cases.json", "w"), indent=1)
    print("split cases:", [(c["name"], len(c["samples"])) for c in cases])


def wavelets():
    t = open(f"{REF}/CanvasPartition/WaveletTests.cs", encoding="utf-8-sig").read()
    body = re.search(r"var coverage = new double\[\]\s*\{(.*?)\};", t, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body)
    cov = [float(v) for v in body.replace("\n", " ").split(",") if v.strip()]
    call = re.search(r"GetCoverageVariability\((\d+),", t)
    hw = re.search(r"HaarWavelets\(coverage,\s*([\d.]+),\s*([\d.]+),\s*breakpoints,\s*(\w+),\s*([\d.]+),", t, re.S)
    count = int(re.search(r"Assert.Equal\((\d+), breakpoints.Count\)", t).group(1))
    exp = [int(v) for v, i in sorted(re.findall(r"Assert.Equal\((\d+), breakpoints\[(\d+)\]\)", t), key=lambda p: int(p[1]))]
    assert len(exp) == count
    d = {"coverage": cov, "variability_window": int(call.group(1)), "threshold_lower": float(hw.group(1)), "threshold_upper": float(hw.group(2)),
         "is_germline": hw.group(3) == "true", "mad_factor": float(hw.group(4)), "expected_breakpoints": exp}
    json.dump(d, open(f"{OUT}/wavelets_minimal.json", "w"))
    print("wavelets:", len(cov), "values ->", exp)


if __name__ == "__main__":
    wavelets()
    loess()
    split_overlapping()
