#!/usr/bin/env python3
"""Per-pass kernel timeline of bench.py from a rocprofv3 (rocpd sqlite) kernel trace: which kernels ran in the last timed pass, their
durations and the idle gaps between them.  usage: tools/timeline.py <results.db> [min_gap_us]"""
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in rows]
# a pass starts at k_find_pos0 / k_tile_stats (first kernel of canvas_bin_sample)
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_tile_stats") or r[0].startswith("k_tile_summary<") or r[0] in ("k_tile_summary", "k_tile_summary_packed")]
if len(starts) < 2:
    sys.exit("no passes found")
# which pass: TIMELINE_PASS = index of the pass among the passes of the run (bench.py: warmup + steps - 1 = the last timed one); default: the one before the last
pi = int(os.environ.get("TIMELINE_PASS", "-2"))
a, b = starts[pi], starts[pi + 1]
seg = rows[a:b]
t0 = seg[0][1]
busy = sum(e - s for _, s, e in seg)
span = seg[-1][2] - t0
print("pass: %d kernels, span %.1f us, busy %.1f us, idle %.1f us" % (len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3))
import os, json
if os.environ.get("TIMELINE_JSON"):      # the pass summary for bench.py (roofline.idle_us_per_pass): TIMELINE_JSON=<file> TIMELINE_SCALE=1.0 TIMELINE_RATE=0.21
    json.dump({"span_us": round(span / 1e3, 1), "busy_us": round(busy / 1e3, 1), "idle_us": round((span - busy) / 1e3, 1), "kernels": len(seg),
               "scale": float(os.environ.get("TIMELINE_SCALE", "1.0")), "rate": float(os.environ.get("TIMELINE_RATE", "0.21"))}, open(os.environ["TIMELINE_JSON"], "w"))
prev_end = t0
for n, s, e in seg:
    gap = (s - prev_end) / 1e3
    mark = "   <-- gap %.1f us" % gap if gap >= min_gap else ""
    print("%9.1f  %-40s %8.1f%s" % ((s - t0) / 1e3, n[:40], (e - s) / 1e3, mark))
    prev_end = e
