#!/usr/bin/env python3
"""Summarises the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected in separate runs as the MI355X guide prescribes)
into profiles/.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
16 B/lane streaming read -> doubled for the streaming kernels below; WRITE_SIZE is taken as reported (KB)."""
import collections
import csv
import json
import sys

fetch_csv, write_csv, out_txt, out_json = sys.argv[1:5]
kernel = sys.argv[5] if len(sys.argv) > 5 else "k_tile_summary"


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


F, W = load(fetch_csv), load(write_csv)
stream16 = {"k_bin_pass", "k_tile_stats", "k_tile_summary", "k_tile_summary_packed"}     # kernels whose reads are 16 B/lane coalesced streams
rows = []
for k in sorted(set(F) | set(W), key=lambda k: -(sum(F.get(k, [0])) + sum(W.get(k, [0])))):
    f = sum(F.get(k, [0])) / max(1, len(F.get(k, [1]))) * 1024.0
    w = sum(W.get(k, [0])) / max(1, len(W.get(k, [1]))) * 1024.0
    corr = 2.0 if k in stream16 else 1.0
    rows.append((k, len(F.get(k, [])), f, corr, w, f * corr + w))
with open(out_txt, "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes per launch; bench.py --steps 1 --warmup 0, 60x genome\n")
    o.write("%-44s %6s %16s %5s %16s %16s\n" % ("kernel", "calls", "FETCH_SIZE_B", "corr", "WRITE_SIZE_B", "HBM_bytes"))
    for r in rows[:40]:
        o.write("%-44s %6d %16.0f %5.1f %16.0f %16.0f\n" % (r[0][:44], r[1], r[2], r[3], r[4], r[5]))
bp = [r for r in rows if r[0] == kernel][0]
json.dump({"kernel": kernel, "fetch_size_bytes_reported": bp[2], "fetch_correction": bp[3], "write_size_bytes": bp[4], "hbm_bytes_per_launch": bp[5],
           "workload": {"scale": 1.0, "rate": 0.21}}, open(out_json, "w"), indent=1)
print(open(out_txt).read()[:2500])
