python bench.py --no-cpu-baseline --steps 30 > gpurun_out/b30.log 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -3; sleep 0.4; done
wait $PID
tail -1 gpurun_out/b30.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['other_kernels'])"
