#!/usr/bin/env python3
"""How busy the device is during the CBS call of the tumour / normal flow, from a rocprofv3 kernel trace (rocpd sqlite): inside the window from the first to the last
permutation kernel of the LAST flow, the share of the time with 0, 1, 2, ... kernels running and the time-weighted sum of (workgroups x LDS bytes) as a share of the device's
256 x 160 KB.  usage: tools/gpu_busy.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
q = "select name, start, end, %s, %s, %s from kernels order by start" % ("grid_x" if "grid_x" in cols else "0", "workgroup_x" if "workgroup_x" in cols else "1", "lds_size" if "lds_size" in cols else ("lds_block_size" if "lds_block_size" in cols else "0"))
rows = [(n.split("(")[0].replace("void ", ""), s, e, gx, wx, lds) for n, s, e, gx, wx, lds in db.execute(q).fetchall()]
perm = [r for r in rows if r[0].startswith("k_perm_rp") or r[0].startswith("k_perm_fy")]
if not perm: sys.exit("no permutation kernels")
# flows are separated by long gaps without permutation kernels: take the last cluster
clusters = [[perm[0]]]
for r in perm[1:]:
    if r[1] - clusters[-1][-1][2] > 50e6: clusters.append([r])
    else: clusters[-1].append(r)
w0, w1 = clusters[-1][0][1], max(r[2] for r in clusters[-1])
ev = []
for n, s, e, gx, wx, lds in rows:
    s2, e2 = max(s, w0), min(e, w1)
    if e2 > s2:
        wgs = max(1, (gx or 1) // max(1, wx or 1)); ev.append((s2, 1, n, wgs, lds or 0)); ev.append((e2, -1, n, wgs, lds or 0))
ev.sort()
conc = 0; last = w0; hist = {}; byname = {}; running = {}
for t, d, n, wgs, lds in ev:
    hist[conc] = hist.get(conc, 0) + (t - last)
    for k in running: byname[k] = byname.get(k, 0) + (t - last) * (1.0 / max(1, conc))      # the window's time shared among the kernels that run
    last = t; conc += d
    if d > 0: running[n] = running.get(n, 0) + 1
    else:
        running[n] -= 1
        if running[n] == 0: del running[n]
span = w1 - w0
print("window %.1f ms (the last flow's permutation kernels); kernels running at once -> share of the window:" % (span / 1e6), {k: "%.1f %%" % (100.0 * v / span) for k, v in sorted(hist.items())})
print("the window's time, shared equally among the kernels running at each moment:", {k: "%.1f %%" % (100.0 * v / span) for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:10]})
