#!/usr/bin/env python3
"""Sums the FETCH_SIZE / WRITE_SIZE counters (separate rocprofv3 --pmc passes over `tools/clean_batch_probe.py 8 1`) per CanvasClean kernel.
usage: tools/pmc_clean_batch.py fetch_counter_collection.csv write_counter_collection.csv bins_per_sample samples [out.json] > profiles/rNN_pmc_clean_batch.txt
(out.json: the per-bin totals for bench.py's clean_frac_counter_bytes)"""
import collections, csv, sys

fetch_csv, write_csv, bins, samples = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith(("k_cf_", "k_cq_")): agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return agg


F, W = load(fetch_csv), load(write_csv)
print(f"# canvas_clean_batch, {samples} WGS samples ({bins} bins each), one call: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes summed over the launches of the call, as reported (no correction)")
print("%-26s %6s %16s %16s" % ("kernel", "calls", "FETCH_SIZE_B", "WRITE_SIZE_B"))
tf = tw = 0.0
for k in sorted(set(F) | set(W), key=lambda k: -(sum(F.get(k, [0])) + sum(W.get(k, [0])))):
    f, w = sum(F.get(k, [0])), sum(W.get(k, [0])); tf += f; tw += w
    print("%-26s %6d %16.0f %16.0f" % (k, len(F.get(k, W.get(k, []))), f, w))
print("%-26s %6s %16.0f %16.0f" % ("total", "", tf, tw))
print(f"# per bin: {tf / bins / samples:.1f} B fetched + {tw / bins / samples:.1f} B written (SURVEY 8(d) stage-sum figure: 232 B/bin)")
if len(sys.argv) > 5:
    import json
    json.dump({"fetched_bytes_per_bin": round(tf / bins / samples, 2), "written_bytes_per_bin": round(tw / bins / samples, 2), "bytes_per_bin": round((tf + tw) / bins / samples, 2),
               "samples_in_call": samples, "bins_per_sample": bins, "flags": "-g -s -r --local-sd-metric-file", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/clean_batch_probe.py 8 1"},
              open(sys.argv[5], "w"), indent=1)
