export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
for fb in 16 8 4; do echo "first batch $fb"; CANVAS_WV_FIRST_BATCH=$fb CANVAS_WV_TIMING=1 python bench.py --no-cbs --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "variability\|^{" | tail -2 | python -c "
import sys, json
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d = json.loads(l)['wavelets_path']; print({k: d.get(k) for k in ('seconds','seconds_of_each_call','first_call_seconds')})
    else: print(l)
"; done
