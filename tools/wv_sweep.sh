export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
python -m pytest tests/test_wavelets_gpu.py -x -q -m gpu > gpurun_out/wvt.log 2>&1; tail -2 gpurun_out/wvt.log
run() { echo "== $*"; env "$@" CANVAS_WV_TIMING=1 python bench.py --no-cbs --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "canvas_wavelets\|^{" | tail -7 | python -c "
import sys, json
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d = json.loads(l)['wavelets_path']; print({k: d[k] for k in ('seconds','seconds_of_each_call','first_call_seconds','chain_kernel_seconds')})
    elif 'levels' in l or 'decomposition' in l: print(l)
"; }
run CANVAS_WV_LONG=64 CANVAS_WV_CHAIN_CUS=0
run CANVAS_WV_LONG=64 CANVAS_WV_CHAIN_CUS=32
run CANVAS_WV_LONG=128 CANVAS_WV_CHAIN_CUS=0
run CANVAS_WV_LONG=128 CANVAS_WV_CHAIN_CUS=32
run CANVAS_WV_LONG=256 CANVAS_WV_CHAIN_CUS=0
run CANVAS_WV_LONG=64 CANVAS_WV_CHAIN_CUS=0 CANVAS_WV_NO_TABLE=1
