"""kernels of the last CanvasClean call in a rocprofv3 kernel trace (start order, durations, gaps): python tools/cf_timeline.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = [(n.split("(")[0].replace("void ", ""), s, e) for n, s, e in rows]
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_cf_init") or r[0].startswith("k_cg_count")]
a = idx[-1]
t0 = rows[a][1]; prev = t0
for n, s, e in rows[a:a + 14]:
    print("%8.1f %-30s %7.1f  gap %5.1f" % ((s - t0) / 1e3, n[:30], (e - s) / 1e3, (s - prev) / 1e3)); prev = e
    if "copyBuffer" in n: break
