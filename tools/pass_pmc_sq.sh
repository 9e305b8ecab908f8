#!/bin/bash
# SQ counters of the kernels of one bench pass (two rocprofv3 --pmc runs, counters only): gpurun_out/<tag>/pmc_sq_pass.txt, pmc_lds_pass.txt.  usage: tools/pass_pmc_sq.sh <tag>
tag=${1:-sq}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OFF="--no-cpu-baseline --no-cbs --no-wavelets --no-somatic --no-h2d --no-packed --no-executables --no-gc-only --no-pedigree --steps 3 --warmup 1"
rm -rf /tmp/pq1 /tmp/pq2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pq1 -o g -- python $R/bench.py $OFF > /tmp/pq1.log 2>&1; echo "sq rc $?"
python $R/tools/pmc_sq_summary.py $(find /tmp/pq1 -name "*counter_collection.csv" | head -n 1) > $O/pmc_sq_pass.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pq2 -o g -- python $R/bench.py $OFF > /tmp/pq2.log 2>&1; echo "lds rc $?"
python - $(find /tmp/pq2 -name "*counter_collection.csv" | head -n 1) > $O/pmc_lds_pass.txt 2>&1 <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if k.startswith(("k_synth", "k_pack", "__amd")): continue
    print("%-32s" % k[:32], {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "calls", len(next(iter(d.values()))))
PY
cat $O/pmc_sq_pass.txt; cat $O/pmc_lds_pass.txt
