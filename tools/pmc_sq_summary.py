#!/usr/bin/env python3
"""Per-kernel summary of one `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES` pass (the SQ counters count quad-cycles,
MI355X_MICROARCH.md): cycles from the first to the last instruction of a wave, the share of them parked in s_waitcnt / barriers, the share issuing, and the VALU time
per SIMD — which tells a latency-bound kernel (high wait share, long waves) from a VALU-bound one (VALU time per SIMD ~ kernel duration).
usage: tools/pmc_sq_summary.py counter_collection.csv > profiles/rNN_pmc_sq_pass.txt"""
import collections, csv, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES -- python tools/packed_probe.py 3   (averages per launch; 1024 SIMDs, 2.4 GHz)")
print("%-30s %6s %9s %12s %7s %7s %14s" % ("kernel", "calls", "waves", "cycles/wave", "wait%", "issue%", "VALU_us/SIMD"))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]["SQ_WAVE_CYCLES"]) / len(kv[1]["SQ_WAVE_CYCLES"])):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if m["SQ_WAVE_CYCLES"] < 2e5 or k.startswith(("k_synth", "k_pack_", "__amd")): continue
    print("%-30s %6d %9d %12.0f %7.1f %7.1f %14.1f" % (k[:30], len(d["SQ_WAVES"]), m["SQ_WAVES"], 4 * m["SQ_WAVE_CYCLES"] / max(1, m["SQ_WAVES"]), 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                                                      100 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4 * m["SQ_ACTIVE_INST_VALU"] / 1024 / 2400))
