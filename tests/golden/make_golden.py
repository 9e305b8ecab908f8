#!/usr/bin/env python3
"""Extracts the known-answer DATA (inputs / expected outputs) that the reference's own xUnit tests hold for the
read-depth hot path into small JSON fixtures.  Run in the build container only (reads /root/reference);
the JSON files it writes are committed and are what the tests use.  No reference source text is stored.

  LOESS vectors            <- CanvasTest/TestLoessInterpolator.cs:13-81
  split-overlapping cases  <- CanvasTest/CanvasPartition/GenomeSegmentationResultsTests.cs:14-248
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


if __name__ == "__main__":
    loess()
    split_overlapping()
