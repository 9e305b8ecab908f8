#!/bin/bash
# quick iteration on a GPU box: selected tests, then the kernel timeline of one bench pass.  usage: tools/quick_pass.sh <tag> "<pytest -k expression or empty>" [extra bench flags]
tag=${1:-q}; sel=${2:-}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ -n "$sel" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$sel" 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt; fi
cd /tmp; export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree"
timeout 300 python $R/bench.py $FLAGS --steps 20 --warmup 5 "$@" > $O/bench_plain.log 2>&1; echo "bench rc $?"; tail -c 1500 $O/bench_plain.log | python -c "
import sys,json
for l in sys.stdin.read().splitlines()[::-1]:
    try: d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline'].get('frac'), d['roofline'].get('avg_ms')); break
    except Exception: pass
"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o pass -- python $R/bench.py $FLAGS --steps 5 --warmup 2 "$@" > /tmp/bench_prof.log 2>&1; echo "pass profile rc $?"
db=$(find /tmp/prof1 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/kernel_stats.txt /tmp/bench_prof.log > /dev/null; TIMELINE_PASS=6 TIMELINE_JSON=$O/pass_timeline.json python tools/timeline.py $db 6 > $O/pass_timeline.txt 2>&1)
cat $O/pass_timeline.txt
