#!/bin/bash
# Collects the per-round evidence under gpurun_out/<tag>/ on a GPU box (copy what is to be judged into profiles/):
#   kernel summary + timeline of the bench pass, kernel summaries of the somatic flow and of the CBS probe (rocprofv3 --kernel-trace --stats), the five parity soaks.
# usage: tools/round_profiles.sh <tag> [soak minutes]      (run from the repo root or via gpurun; every step is bounded by `timeout`)
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
tag=${1:-rXX}; mins=${2:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o pass -- python $R/bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --steps 5 --warmup 2 > /tmp/bench_prof.log 2>&1; echo "pass profile rc $?"
db=$(find /tmp/prof1 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/kernel_stats.txt /tmp/bench_prof.log > /dev/null; python tools/timeline.py $db > $O/pass_timeline.txt 2>&1)
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o som -- python $R/bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --steps 1 --warmup 0 > /tmp/som.log 2>&1; echo "somatic profile rc $?"
db=$(find /tmp/prof2 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/somatic_kernel_stats.txt /tmp/som.log > /dev/null)
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o cbs -- python $R/tools/cbs_time.py 4700000 2 > /tmp/cbs.log 2>&1; echo "cbs profile rc $?"
db=$(find /tmp/prof3 -name "*.db" | head -1); (cd $R; python tools/rocprof_summary.py $db $O/cbs_kernel_stats.txt > /dev/null; echo "# $(tail -1 /tmp/cbs.log)" >> $O/cbs_kernel_stats.txt)
cd $R; lim=$((mins * 60 + 90))
(timeout $lim python tools/soak.py $mins 31337 2>&1 | tail -2 > $O/soak.txt)
(timeout $lim python tools/soak_bin.py $mins 31337 2>&1 | tail -2 >> $O/soak.txt)
(timeout $lim python tools/soak_wavelets.py $mins 31337 2>&1 | tail -2 >> $O/soak.txt)
(timeout $lim python tools/soak_cbs.py $mins 31337 2>&1 | tail -2 >> $O/soak.txt)
(timeout $lim python tools/soak_gcw.py $mins 31337 2>&1 | tail -1 >> $O/soak.txt)
(CANVAS_CBS_FY_MIN_N=1024 timeout 330 python tools/soak_cbs.py 4 4242 2>&1 | tail -2 | sed -e 's/^/[k_perm_fy on every device segment] /' >> $O/soak.txt)
cat $O/soak.txt
