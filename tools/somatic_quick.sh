#!/bin/bash
# the tumour / normal flow of bench.py alone, with its kernel summary.  usage: tools/somatic_quick.sh <tag>
tag=${1:-som}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o som -- python $R/bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --steps 1 --warmup 0 > /tmp/som.log 2>&1; echo "somatic profile rc $?"
db=$(find /tmp/prof2 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/somatic_kernel_stats.txt /tmp/som.log > /dev/null)
head -30 $O/somatic_kernel_stats.txt
(cd $R; python tools/gpu_busy.py $db | tee $O/gpu_busy.txt)
tail -c 3000 /tmp/som.log | python -c "
import sys, json
for l in sys.stdin.read().splitlines()[::-1]:
    try:
        d = json.loads(l); print(json.dumps(d['somatic_flow'])[:1500]); break
    except Exception: pass
"
