#!/bin/bash
# N fresh processes (default 32, P at a time, default 4), each a small Bin -> Clean -> {HMM, CBS, Wavelets} flow against the oracle (tools/start_child.py).
# Prints one line per start and a summary: mismatches must be 0; "waited" counts the looks at a pinned result that came before it had arrived (the library then polled).
N=${1:-32}; P=${2:-4}; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tmp=$(mktemp -d); fail=0
for ((s = 0; s < N; s += P)); do
  for ((k = s; k < s + P && k < N; k++)); do ( timeout 300 python $R/tools/start_child.py $k > $tmp/$k.out 2> $tmp/$k.err; echo $? > $tmp/$k.rc ) & done
  wait
done
python - "$tmp" "$N" <<'PY'
import json, sys, os
d, n = sys.argv[1], int(sys.argv[2]); bad = 0; aw = 0; wt = 0
for k in range(n):
    rc = open(os.path.join(d, f"{k}.rc")).read().strip(); line = ""
    for l in open(os.path.join(d, f"{k}.out")):
        if l.startswith("{"): line = l.strip()
    try: j = json.loads(line)
    except Exception: j = {"ok": False, "what": "no result line (rc %s): %s" % (rc, open(os.path.join(d, f"{k}.err")).read()[-400:]), "awaits": 0, "waited": 0}
    aw += j["awaits"]; wt += j["waited"]; bad += 0 if j["ok"] and rc == "0" else 1
    print(f"start {k}: {'ok' if j['ok'] else 'MISMATCH: ' + j['what']} (pinned results looked at {j['awaits']}, came early {j['waited']})")
print(f"start stress: {n} process starts, {bad} mismatches, {aw} pinned results looked at, {wt} looks before arrival (polled)")
sys.exit(1 if bad else 0)
PY
rc=$?; rm -rf $tmp; exit $rc
