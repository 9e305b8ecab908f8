#!/bin/bash
# experiments on the one-shot executables: bench.py's executables leg with the sample's files kept, then CanvasBin timed under a few environment settings
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
export CANVAS_EXE_KEEP=/tmp/exe_root.txt
python bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-gc-only --no-pedigree --steps 2 --warmup 1 > gpurun_out/exe.log 2>&1
python - <<PY
import json
for l in open("gpurun_out/exe.log"):
    if l.startswith("{"):
        d=json.loads(l)["executables"]
        for k,v in d.items():
            if isinstance(v,dict) and "wall_seconds" in v: print(k, v["wall_seconds"], v.get("phases"))
        print("total", d.get("wall_seconds_bin_clean_partition(PerSampleHMM)"))
PY
root=$(cat /tmp/exe_root.txt); cmd=$(cat $root/bin_cmd.txt)
export CANVAS_TOOL_TIMING=1
TIMEFORMAT="wall %R"
t() { time "$@" > /dev/null; }
stamp() { echo "caller clock: before the spawn $1, after the wait $(date +%s.%N)"; }
echo "== default"; b=$(date +%s.%N); t $cmd; stamp $b
echo "== default, again"; b=$(date +%s.%N); t $cmd; stamp $b
echo "== full teardown"; export CANVAS_TOOL_FULL_TEARDOWN=1; t $cmd; date +%s.%N; unset CANVAS_TOOL_FULL_TEARDOWN
echo "== help"; t canvas_amd/bin/CanvasBin -h
grep -i "thp\|AnonHugePages" /proc/meminfo; cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; nproc; free -g | head -2
rm -rf $root
