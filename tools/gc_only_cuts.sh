#!/bin/bash
# timing hook of clean_gc_only.hpp: the three kernels stopped at successive points (CANVAS_CG_CUT=1..8; results are void).  usage: tools/gc_only_cuts.sh <tag>
export CANVAS_TEST_HOOKS=1
tag=${1:-gccut}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cut in 1 2 3 7 8 0; do
  rm -rf /tmp/profg
  CANVAS_CG_CUT=$cut timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/profg -o g -- python $R/tools/clean_probe.py 6 0.105 g > /tmp/g.log 2>&1
  db=$(find /tmp/profg -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db /tmp/ks.txt > /dev/null; echo "cut $cut: $(grep -E 'k_cg_' /tmp/ks.txt | awk '{printf "%s %s us | ", $1, $4}')") | tee -a $O/cuts.txt
done
