#!/usr/bin/env python3
"""Kernel timeline of the last canvas_wavelets call in a rocprofv3 (rocpd sqlite) kernel trace: the level kernels (busy time, gaps), the chain / subtree kernels and how they
overlap.  usage: tools/wv_timeline.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
rows = [(r[0].split("(")[0].replace("void ", ""), r[1], r[2], r[3] if qcol else 0) for r in rows]
pre = [i for i, r in enumerate(rows) if r[0].startswith("k_wv_prefix")]
if not pre: sys.exit("no wavelets call found")
a = pre[-1]; seg = [r for r in rows[a:] if r[0].startswith("k_wv")]
t0 = seg[0][1]
lv = [r for r in seg if r[0].startswith("k_wv_level")]
busy = sum(e - s for _, s, e, _ in lv); span = lv[-1][2] - lv[0][1]
gaps = sorted(((lv[i + 1][1] - lv[i][2]) / 1e3, i) for i in range(len(lv) - 1))
print("levels: %d launches, first at %.1f us, span %.1f us, busy %.1f us, gaps %.1f us; largest gaps: %s" % (len(lv), (lv[0][1] - t0) / 1e3, span / 1e3, busy / 1e3, (span - busy) / 1e3, ", ".join("%.0f us after #%d" % g for g in gaps[-8:])))
nonempty = [r for r in lv if r[2] - r[1] > 1500]
print("level launches longer than 1.5 us: %d, their busy time %.1f us, median %.1f us" % (len(nonempty), sum(e - s for _, s, e, _ in nonempty) / 1e3, sorted((e - s) / 1e3 for _, s, e, _ in nonempty)[len(nonempty) // 2] if nonempty else 0))
for n, s, e, q in seg:
    if n.startswith("k_wv_level") or n.startswith("k_wv_list_init"): continue
    if (e - s) < 20000 and not n.startswith("k_wv_chain"): continue
    print("%10.1f .. %10.1f  %-28s %9.1f us  queue %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n[:28], (e - s) / 1e3, q))
print("end of the last wavelets kernel: %.1f us" % ((max(e for _, _, e, _ in seg) - t0) / 1e3))
