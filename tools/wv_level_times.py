#!/usr/bin/env python3
"""Durations of the k_wv_level launches of the last canvas_wavelets call in a rocprofv3 (rocpd sqlite) trace, in launch order, ten per line (us).
usage: tools/wv_level_times.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in rows]
pre = [i for i, r in enumerate(rows) if r[0].startswith("k_wv_prefix_tiles")]
lv = [(e - s) / 1e3 for n, s, e in rows[pre[-1]:] if n.startswith("k_wv_level")]
print("level launches:", len(lv), "busy %.1f us" % sum(lv))
for a in range(0, len(lv), 10): print("%4d: " % a + " ".join("%6.1f" % v for v in lv[a:a + 10]))
