#!/bin/bash
# CanvasClean -g alone (BASELINE configs[1]) under rocprofv3: kernel summary and the timeline of the last call.  usage: tools/gc_only_probe.sh <tag>
export CANVAS_TEST_HOOKS=1
tag=${1:-gc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/clean_probe.py 8 0.105 g > $O/probe.txt 2>&1; tail -4 $O/probe.txt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profg -o g -- python $R/tools/clean_probe.py 8 0.105 g > /tmp/g.log 2>&1; echo "profile rc $?"
db=$(find /tmp/profg -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/kernel_stats.txt > /dev/null; grep -E "k_cg_|kernel " $O/kernel_stats.txt; python tools/cf_timeline.py $db | tee $O/timeline.txt)
