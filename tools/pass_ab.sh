#!/bin/bash
# A/B of the bench pass on ONE box.  usage: tools/pass_ab.sh "VAR=1" ...   ("" = defaults)
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
FLAGS="--no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree"
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py $FLAGS --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; o=r['other_kernels']
print('rep $rep [${cfg:-defaults}] ms_per_step', d['ms_per_step'], 'sweep', r['avg_ms'], 'tail', r.get('bin_tail_ms'), 'clean', r.get('clean_ms'), 'viterbi', r.get('viterbi_ms'), 'retries', o['viterbi(speculate+backbone+verify)']['second_attempts'], 'fallbacks', o['viterbi(speculate+backbone+verify)']['sequential_fallbacks'], 'span', r.get('pass_span_ms'), 'hand-over us', r.get('hand_over_us_per_pass'))"
done
done
