#!/bin/bash
# the permutation loops of the tumour / normal flow (BASELINE configs[4]): one line per loop (segment length, permutations, seconds).  usage: tools/somatic_loops.sh <tag>
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
tag=${1:-loops}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
CANVAS_CBS_TIMING=2 timeout 600 python tools/somatic_probe.py > $O/probe.log 2> $O/loops.err; echo "rc $?"
grep -c "cbs loop" $O/loops.err
python - <<PY
import re
rows=[]
for l in open("$O/loops.err"):
    m=re.search(r"cbs loop: n (\d+) nrejc (-?\d+) stop-if-no-rejection (\d+) outcome (\d+) seconds ([\d.]+) perms (\d+) batches (\d+) computed (\d+)",l)
    if m: rows.append(tuple(float(x) for x in m.groups()))
rows=rows[len(rows)//2:]   # the second flow
tot_el=sum(r[0]*r[5] for r in rows); print("loops",len(rows),"elements looked at %.4g"%tot_el,"computed %.4g"%sum(r[0]*r[7] for r in rows),"perms",sum(r[5] for r in rows),"computed",sum(r[7] for r in rows),"batches",sum(r[6] for r in rows))
bins=[0,1024,4096,16384,32768,65536,131072,262144,1<<30]
for a,b in zip(bins,bins[1:]):
    rr=[r for r in rows if a<=r[0]<b]
    if rr: print(f"n in [{a},{b}): loops {len(rr)} perms {sum(r[5] for r in rr):.0f} batches {sum(r[6] for r in rr):.0f} elements {sum(r[0]*r[5] for r in rr):.3g} ({100*sum(r[0]*r[5] for r in rr)/tot_el:.1f}%) loop-seconds {sum(r[4] for r in rr):.3f}")
print("largest loops (n, nrejc, stop, outcome, seconds, perms, batches):")
for r in sorted(rows,key=lambda r:-r[0]*r[5])[:25]: print(r, "%.2f ns/element"%(1e9*r[4]/(r[0]*r[5])))
PY
grep -v "cbs loop" $O/loops.err | tail -30
tail -5 $O/probe.log
