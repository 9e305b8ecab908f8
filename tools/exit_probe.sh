#!/bin/bash
# what a process pays AFTER its last statement once it has used the GPU (tools/probe_src/exit_probe.cpp), against the caller's clock
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TL=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
g++ -O2 -std=c++17 tools/probe_src/exit_probe.cpp -Iinclude -Lcanvas_amd -lcanvas_hip -Wl,-rpath,$R/canvas_amd -L$TL -Wl,-rpath,$TL -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -lrccl -o /tmp/exit_probe || exit 1
for m in 1 2 3 4 5 6 7 4 6; do mbs=2048; [ $m -ge 4 ] && mbs=7000; b=$(date +%s.%N); /tmp/exit_probe $m $mbs; e=$(date +%s.%N); echo "   caller: spawn $b, wait returned $e  -> total $(python -c "print(round($e-$b,3))")"; done
