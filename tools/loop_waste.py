#!/usr/bin/env python3
"""Permutations computed against permutations looked at by the sequential stopping rule, per loop of a CBS call, from the stderr of a run with
CANVAS_TEST_HOOKS=1 CANVAS_CBS_TIMING=2 (lines "cbs loop: n .. perms .. batches .. computed ..").  usage: tools/loop_waste.py <stderr file> [label]"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'cbs loop: n (\d+) nrejc (\d+) stop-if-no-rejection (\d+) outcome (\d+) seconds ([\d.]+) perms (\d+) batches (\d+) computed (\d+)', l)
    if m:
        rows.append(tuple(float(v) if i == 4 else int(v) for i, v in enumerate(m.groups())))
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = rows[len(rows) - len(rows) // calls:]            # the last call of the run
looked = sum(r[0] * r[5] for r in rows); comp = sum(r[0] * r[7] for r in rows)
print("%s: %d loops (%d end 'significant'), %d batches; permuted elements looked at %.4g, computed %.4g: %.1f %% never looked at" %
      (sys.argv[2] if len(sys.argv) > 2 else "last call", len(rows), sum(1 for r in rows if r[3] == 1), sum(r[6] for r in rows), looked, comp, 100.0 * (comp - looked) / max(comp, 1)))
for r in sorted(rows, key=lambda r: -(r[7] - r[5]) * r[0])[:8]:
    print("  n %6d nrejc %3d stop-if-no-rejection %5d outcome %d: looked at %5d of %5d computed in %d batches" % (r[0], r[1], r[2], r[3], r[5], r[7], r[6]))
