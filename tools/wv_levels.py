#!/usr/bin/env python3
"""Per-launch durations of the wavelet chain kernel from a rocprofv3 (rocpd sqlite) trace.  usage: tools/wv_levels.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
ch = [(e - s) / 1e3 for n, s, e, g in rows if "k_wv_chain_long" in n]
gx = [g for n, s, e, g in rows if "k_wv_chain_long" in n]
print("launches", len(ch), "total ms %.1f" % (sum(ch) / 1e3))
print("first 40 launches (us, waves):", [(round(a), g // 64) for a, g in zip(ch[:40], gx[:40])])
for name in ("k_wv_coeff", "k_wv_verify", "k_wv_short"):
    d = [(e - s) / 1e3 for n, s, e, g in rows if name in n]
    print(name, "launches", len(d), "total ms %.1f" % (sum(d) / 1e3))
