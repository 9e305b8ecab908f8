#!/bin/bash
# where the "device" phase of the CanvasPartition -m CBS executable goes: bench.py's executables leg with the files kept, then the tool again with the CBS timing report
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
export CANVAS_EXE_KEEP=/tmp/exe_root.txt
python bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-gc-only --no-pedigree --steps 2 --warmup 1 > gpurun_out/exe.log 2>&1
root=$(cat /tmp/exe_root.txt)
mkdir -p $root/WholeGenomeFasta
for i in 1 2; do CANVAS_TOOL_TIMING=1 CANVAS_CBS_TIMING=1 canvas_amd/bin/CanvasPartition -i $root/S.cleaned -o $root/S.cbs.partitioned -r $root/WholeGenomeFasta -m CBS 2>&1 | grep -v "helpers\|arc searches" | cut -c1-400; done
rm -rf $root
