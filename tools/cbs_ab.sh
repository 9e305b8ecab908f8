#!/bin/bash
# A/B of canvas_cbs wall time: this tree's cbs.hip against another version of it ($1), both on the same box, with the box's CPU budget printed
# usage: tools/cbs_ab.sh other_cbs.hip  -> gpurun_out/cbs_ab.txt
mkdir -p gpurun_out; out=gpurun_out/cbs_ab.txt; : > $out
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; uptime; } >> $out
python -c "import __graft_entry__ as g; g.build()" >> $out 2>&1
for r in 1 2; do uptime >> $out; python tools/cbs_time.py 4700000 5 >> $out 2>&1; done
if [ -n "$1" ]; then
  cp canvas_amd/csrc/cbs.hip /tmp/cbs_new.hip; cp "$1" canvas_amd/csrc/cbs.hip
  python -c "import __graft_entry__ as g; g.build()" >> $out 2>&1
  echo "--- other version" >> $out
  for r in 1 2; do uptime >> $out; python tools/cbs_time.py 4700000 5 >> $out 2>&1; done
  cp /tmp/cbs_new.hip canvas_amd/csrc/cbs.hip
fi
echo "--- threads pinned to few cores? taskset: $(taskset -p $$)" >> $out
