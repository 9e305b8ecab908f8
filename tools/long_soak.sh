#!/bin/bash
# the six randomised parity soaks for <minutes> each with a seed of their own (a longer companion of tools/round_profiles.sh): gpurun_out/long_soak.txt
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
mins=${1:-8}; seed=${2:-20260929}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/long_soak.txt; : > $O; cd $R; lim=$((mins * 60 + 120))
for t in soak soak_bin soak_wavelets soak_cbs soak_gcw; do (timeout $lim python tools/$t.py $mins $seed 2>&1 | grep -v amdgpu.ids | tail -n 2 >> $O); done
(CANVAS_CBS_FY_MIN_N=1024 timeout $lim python tools/soak_cbs.py $mins $((seed + 1)) 2>&1 | grep -v amdgpu.ids | tail -n 1 | sed -e 's/^/[k_perm_fy on every device segment] /' >> $O)
cat $O
