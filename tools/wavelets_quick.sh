#!/bin/bash
# the Wavelets leg of bench.py alone, with the call's own phase timing, its kernel summary and the timeline of the last call.  usage: tools/wavelets_quick.sh <tag> [ENV=VAL ...]
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
tag=${1:-wv}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-cbs --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --steps 1 --warmup 0"
env "$@" CANVAS_WV_TIMING=1 python $R/bench.py $FLAGS > /tmp/wv0.log 2>&1; echo "plain rc $?"
grep "canvas_wavelets" /tmp/wv0.log | tail -6 > $O/wavelets_phases.txt; cat $O/wavelets_phases.txt
rm -rf /tmp/prof3; env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o wv -- python $R/bench.py $FLAGS > /tmp/wv.log 2>&1; echo "profile rc $?"
db=$(find /tmp/prof3 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/wavelets_kernel_stats.txt /tmp/wv.log > /dev/null; python tools/wv_timeline.py $db > $O/wavelets_timeline.txt)
grep -i "wv\|Name" $O/wavelets_kernel_stats.txt | head -14; cat $O/wavelets_timeline.txt
tail -c 4000 /tmp/wv0.log | python -c "
import sys, json
for l in sys.stdin.read().splitlines()[::-1]:
    try:
        d = json.loads(l)['wavelets_path']; print({k: d[k] for k in ('seconds','seconds_of_each_call','first_call_seconds','chain_kernel_seconds')}); break
    except Exception: pass
"
