#!/bin/bash
# SQ counters + kernel summary of the GCContentWeighted chain (tools/gcw_probe.py): gpurun_out/gcw_kernel_stats.txt, gpurun_out/gcw_pmc_sq.txt, gpurun_out/gcw_pmc_lds.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g /tmp/pmc_g1 /tmp/pmc_g2
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o gcw -- python $R/tools/gcw_probe.py 1.0 > /tmp/gcw_prof.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/prof_g -name "*.db" | head -n 1) $O/gcw_kernel_stats.txt > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc_g1 -o g -- python $R/tools/gcw_probe.py 1.0 > /tmp/gcw_pmc1.log 2>&1
python $R/tools/pmc_sq_summary.py $(find /tmp/pmc_g1 -name "*counter_collection.csv" | head -n 1) > $O/gcw_pmc_sq.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_g2 -o g -- python $R/tools/gcw_probe.py 1.0 > /tmp/gcw_pmc2.log 2>&1
python - $(find /tmp/pmc_g2 -name "*counter_collection.csv" | head -n 1) > $O/gcw_pmc_lds.txt 2>&1 <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if not k.startswith(("k_read_gc", "k_nonzero", "k_gcw", "k_bin_weighted", "k_tile_summary")): continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "calls", len(next(iter(d.values()))))
PY
head -n 16 $O/gcw_kernel_stats.txt | cut -c1-160; cat $O/gcw_pmc_sq.txt $O/gcw_pmc_lds.txt
