#!/bin/bash
# A/B of the tumour / normal flow's CBS time on ONE box under different switches.  usage: tools/somatic_ab.sh "VAR1=1" "VAR2=1 VAR3=1" ...   (an empty string = the defaults)
export CANVAS_TEST_HOOKS=1      # (the library reads its CANVAS_* switches only with this set)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
for cfg in "$@"; do
  line=$(env $cfg timeout 300 python tools/somatic_probe.py 2>/dev/null | tail -1 | sed -e "s/.*'cbs': \([0-9.]*\)}.*/\1/")
  echo "rep $rep [${cfg:-defaults}] cbs $line s"
done
done
