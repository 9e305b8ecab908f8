#!/bin/bash
# Regenerates every counter / timeline file bench.py quotes, on the code of this tree (stamped with the git head passed in, gpurun snapshots carry no .git):
#   profiles/pmc_tile_summary.json, pmc_tile_summary_packed.json (FETCH_SIZE / WRITE_SIZE in SEPARATE rocprofv3 --pmc passes, as the MI355X guide prescribes),
#   profiles/pmc_clean_batch.json (+ the per-kernel table), profiles/pass_timeline.json.  Outputs land in gpurun_out/<tag>/ — copy them into profiles/.
# usage: tools/pmc_round.sh <tag> <git head>
export CANVAS_TEST_HOOKS=1
tag=${1:-pmc}; head=${2:-unknown}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-executables --no-gc-only --no-pedigree"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py $OFF --steps 1 --warmup 0 > /tmp/pmc_$c.log 2>&1; echo "pass $c rc $?"
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
(cd $R; python tools/pmc_summary.py $f $w $O/pmc_hbm_bytes.txt $O/pmc_tile_summary.json k_tile_summary; python tools/pmc_summary.py $f $w $O/pmc_hbm_bytes_packed.txt $O/pmc_tile_summary_packed.json k_tile_summary_packed)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcc_$c; timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcc_$c -o p -- python $R/tools/clean_batch_probe.py 8 1 > /tmp/pmcc_$c.log 2>&1; echo "clean pass $c rc $?"
done
f=$(find /tmp/pmcc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmcc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
bins=$(grep -o "bins per sample [0-9]*" /tmp/pmcc_FETCH_SIZE.log | head -1 | awk '{print $4}')
(cd $R; python tools/pmc_clean_batch.py $f $w ${bins:-4786578} 8 $O/pmc_clean_batch.json > $O/pmc_clean_batch.txt; tail -2 $O/pmc_clean_batch.txt)
# the pass timeline of the default bench command
rm -rf /tmp/prof_tl; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -o pass -- python $R/bench.py $OFF --no-packed --steps 5 --warmup 2 > /tmp/tl.log 2>&1; echo "timeline rc $?"
db=$(find /tmp/prof_tl -name "*.db" | head -1); (cd $R; TIMELINE_PASS=6 TIMELINE_JSON=$O/pass_timeline.json python tools/timeline.py $db 6 > $O/pass_timeline.txt 2>&1; head -1 $O/pass_timeline.txt)
(cd $R; python - "$head" $O <<'PY'
import json, sys, os, time
head, O = sys.argv[1], sys.argv[2]
for f in ("pmc_tile_summary.json", "pmc_tile_summary_packed.json", "pmc_clean_batch.json", "pass_timeline.json"):
    p = os.path.join(O, f)
    if os.path.exists(p):
        d = json.load(open(p)); d["git_head"] = head; d["collected"] = time.strftime("%Y-%m-%d"); json.dump(d, open(p, "w"), indent=1); print(f, {k: d[k] for k in list(d)[:6]})
PY
)
