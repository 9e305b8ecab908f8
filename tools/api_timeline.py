#!/usr/bin/env python3
"""Host API calls and kernels of one bench pass on one time axis, from a rocprofv3 --hip-runtime-trace --kernel-trace run (rocpd sqlite).
usage: TIMELINE_PASS=<i> tools/api_timeline.py <results.db>"""
import os
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kern = db.execute("select name, start, end from kernels order by start").fetchall()
kern = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in kern]
starts = [i for i, r in enumerate(kern) if r[0] in ("k_tile_summary", "k_tile_summary_packed")]
pi = int(os.environ.get("TIMELINE_PASS", "-2"))
t0, t1 = kern[starts[pi]][1], kern[starts[pi + 1]][1]
api = None
for cand in ("regions", "region", "api", "hip_api"):
    if cand in tabs:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % cand)]
        if "start" in cols and "end" in cols and "name" in cols:
            api = cand
            break
if api is None:
    print("tables:", tabs)
    sys.exit("no API table found")
rows = db.execute("select name, start, end from %s where end >= ? and start <= ? order by start" % api, (t0 - 200000, t1 + 50000)).fetchall()
ev = [("K", n, s, e) for n, s, e in kern if s >= t0 - 200000 and s <= t1 + 50000] + [("A", n, s, e) for n, s, e in rows]
ev.sort(key=lambda r: r[2])
for kind, n, s, e in ev:
    if kind == "A" and (e - s) < 1500 and not n.startswith("hipStreamSync") and not n.startswith("hipEventSync"):
        continue          # short API calls: noise
    print("%s %10.1f %9.1f  %s" % (kind, (s - t0) / 1e3, (e - s) / 1e3, n[:60]))
