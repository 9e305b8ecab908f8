#!/usr/bin/env python3
"""Repeats the scenario of test_closed_form_decisions_agree_with_the_chains (a chromosome with events, a plain one, a flat one: undecided nodes) and reports every call whose
breakpoints differ from the first call's.  usage: tools/wv_stress.py [n]"""
import os as _os; _os.environ.setdefault("CANVAS_TEST_HOOKS", "1")      # (the library reads its CANVAS_* switches only with this set)
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_wavelets_gpu import _coverage, _run, get_canvas
cv = get_canvas()
rng = np.random.RandomState(21)
per = [_coverage(rng, 90_000, wave=0.05), _coverage(rng, 20_000), np.full(3000, 77.0)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per3 = [per[0] * 1.0000001, per[1]]
scen = [("closed form", per, None), ("chain only", per, "1"), ("not two-decimal", per3, None)]
ref = [None] * 3; bad = [0] * 3
for i in range(n):
    for k, (name, data, env) in enumerate(scen):
        if env: os.environ["CANVAS_WV_CHAIN_ONLY"] = env
        got = [g.tolist() for g in _run(cv, data, window=5000)]
        os.environ.pop("CANVAS_WV_CHAIN_ONLY", None)
        dec = cv.wavelets_decisions()
        if ref[k] is None: ref[k] = got; print("reference (%s):" % name, [len(g) for g in got], dec)
        elif got != ref[k]:
            bad[k] += 1; print("call %d (%s) differs:" % (i, name), [len(g) for g in got], dec, cv.wavelets_stats(), flush=True)
print("calls that differ:", dict(zip([s[0] for s in scen], bad)), "of", n)
