#!/bin/bash
# where the time BEHIND main goes in CanvasPartition -m CBS (the "exit" phase of bench.py's executables leg): the tool under a few settings of the draw-stream cache,
# wall time by the caller's clock against the tool's own total.  usage: tools/cbs_exit_probe.sh
export CANVAS_TEST_HOOKS=1
export CANVAS_EXE_KEEP=/tmp/exe_root.txt
python bench.py --no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-gc-only --no-pedigree --steps 2 --warmup 1 > gpurun_out/exe.log 2>&1
root=$(cat /tmp/exe_root.txt); mkdir -p $root/WholeGenomeFasta
run() {
  for i in 1 2 3; do
    b=$(date +%s.%N)
    env "$@" CANVAS_TOOL_TIMING=1 canvas_amd/bin/CanvasPartition -i $root/S.cleaned -o $root/S.cbs.partitioned -r $root/WholeGenomeFasta -m CBS 2>&1 | grep '"tool"' | python -c "
import sys, json, time
j = json.loads(sys.stdin.readline()); print('  total in main %.3f  phases %s  leaving_unix %.3f' % (j['total'], j['phases'], j['leaving_unix']))" 
    e=$(date +%s.%N); echo "  wall $(python -c "print('%.3f' % ($e - $b))")"
  done
}
echo "== default"; run A=1
echo "== pinned staging kept (CANVAS_TOOL_KEEP_PINNING=1: without canvas_set_one_shot)"; run CANVAS_TOOL_KEEP_PINNING=1
echo "== cache off (CANVAS_CBS_CACHE_GB=0)"; run CANVAS_CBS_CACHE_GB=0
echo "== fixed allotments instead of address ranges (CANVAS_CBS_CACHE_NO_VMM=1)"; run CANVAS_CBS_CACHE_NO_VMM=1
echo "== full teardown through main"; run CANVAS_TOOL_FULL_TEARDOWN=1
echo "== -m PerSampleHMM for comparison"
for i in 1 2; do b=$(date +%s.%N); CANVAS_TOOL_TIMING=1 canvas_amd/bin/CanvasPartition -i $root/S.cleaned -o $root/S.hmm.partitioned -r $root/WholeGenomeFasta -m PerSampleHMM 2>&1 | grep '"tool"' | cut -c1-160; e=$(date +%s.%N); echo "  wall $(python -c "print('%.3f' % ($e - $b))")"; done
rm -rf $root
