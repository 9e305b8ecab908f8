#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary committed under profiles/.
usage: tools/rocprof_summary.py <results.db> <out.txt> [bench-log]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
                  "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n")
    if len(sys.argv) > 3:
        for line in open(sys.argv[3]):
            if line.startswith('{"metric"'):
                f.write("# bench line under the profiler: " + line.strip()[:900] + "\n")
    f.write("%-52s %6s %12s %11s %10s %10s %6s %5s %5s %7s %9s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid_x"))
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")[:52]
        f.write("%-52s %6d %12.1f %11.2f %10.2f %10.2f %6.2f %5d %5d %7d %9d\n" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))
print(open(sys.argv[2]).read()[:3000])
