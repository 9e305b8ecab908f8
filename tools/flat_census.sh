#!/bin/bash
# Which kernels still hold flat_load / flat_store instructions (a pointer read from a table in memory is generic to the compiler unless it is typed gptr<T>, common.hpp)?
# Compiles every csrc/*.hip to gfx950 assembly (no GPU needed) and prints the kernels / device functions with flat instructions next to their LDS instruction counts.
R=$(cd "$(dirname "$0")/.." && pwd); T=${TMPDIR:-/tmp}/flat_census; mkdir -p $T
for f in $R/canvas_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -I$R/canvas_amd/csrc --cuda-device-only -S -o $T/$b.s $f 2>/dev/null &
done; wait
python3 - $T <<'PY'
import re, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.s")):
    cur = None; cnt = {}
    for l in open(f):
        m = re.match(r"^(_Z\w+|\w+):\s+; @", l)
        if m: cur = m.group(1); cnt[cur] = [0, 0, 0, 0]; continue
        if cur is None: continue
        t = l.strip()
        if t.startswith("flat_load"): cnt[cur][0] += 1
        elif t.startswith(("flat_store", "flat_atomic")): cnt[cur][1] += 1
        elif t.startswith("global_"): cnt[cur][2] += 1
        elif t.startswith("ds_"): cnt[cur][3] += 1
    for k, v in cnt.items():
        if v[0] + v[1]: print("%-14s %-60s flat loads %3d  flat stores %3d  global %3d  lds %3d" % (f.split("/")[-1], k[:60], *v))
PY
