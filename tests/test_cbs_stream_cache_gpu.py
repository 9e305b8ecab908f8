"""The draw-stream cache of canvas_cbs (cbs.hip: MtStreamCache; VERDICT r05 Next 1): the k-th chromosome's generator is MersenneTwister(seed_k), seed_k from MersenneTwister(0)
in file order (CBSRunner.cs:107-112), consumed in sequence by XPerm / TPermP (ChangePoint.cs:407-421, CBSTStatistic.cs:1009) — its words are constants, kept in HBM.  The
cached words must BE that generator's outputs (the oracle's MT19937, itself pinned to numpy's in test_oracle_golden.py) at any offset — inside the first extension, across
the seams between extensions (a seam is where the strided generator continues from the words in front of it), across the 64 MB pieces the address range is backed with — and
CBS must give the oracle's segments and RNG consumption whichever way a batch gets its draws: from the cache, from its own generator (cache off), or first one then the other
(the cache's bound reached in the middle of a call)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from gpu_common import get_canvas

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIRST = 8 << 20            # MTS_FIRST_WORDS: the first extension of a stream
HISTORY = 19937 * 128      # MT_HISTORY
GRANULE = 16 << 20         # words per 64 MB piece of backing memory
JOB = 48 << 20             # MTS_JOB_MAX_WORDS: the longest extension
CHUNK = 4 << 20            # MT_CHUNK: a stream is generated chunk by chunk, every chunk from a state obtained by jump-ahead (k_mt_jump) — its seams are where a wrong jump would show


def _expected(seed, pos, n):
    return O.mt_u32(int(np.uint32(seed)), pos + n)[pos:]


def test_cached_stream_is_the_generators_output_at_random_offsets_and_across_extensions():
    cv = get_canvas()
    seeds = O.cbs_seeds(24)
    rng = np.random.RandomState(5)
    # chromosome 3: the start, the bootstrap's seams (19937 x 2^k), the end of the sequential history, the seam behind the first extension, a piece boundary, random offsets
    c = 3
    full = O.mt_u32(int(np.uint32(seeds[c])), FIRST + JOB + 3 * GRANULE + 5000)
    spots = [0, 19937 - 50, 2 * 19937 - 7, 64 * 19937 - 300, HISTORY - 400, FIRST - 777, GRANULE - 123, FIRST + JOB - 999, FIRST + JOB + GRANULE - 5]
    spots += [k * CHUNK - 1000 for k in (1, 2, 3, 5, 9, 14, 17)] + [CHUNK - 1, CHUNK, 19 * CHUNK - 1999]      # straddling chunk seams (2 000 words are read from every spot)
    spots += [int(x) for x in rng.randint(0, FIRST + JOB + 2 * GRANULE, 12)]
    for pos in spots:
        n = 2000
        got = cv.cbs_stream_read(c, pos, n)
        assert (got == full[pos:pos + n]).all(), (c, pos, np.nonzero(got != full[pos:pos + n])[0][:5])
    # another chromosome's stream starts from ITS seed; a long read over several extensions and pieces in one go
    c = 17
    n = FIRST + 2 * GRANULE + 4321
    got = cv.cbs_stream_read(c, 0, n)
    assert (got == O.mt_u32(int(np.uint32(seeds[c])), n)).all()
    st = cv.cbs_cache_stats()
    assert st[4] >= 4 * n and st[5] >= n                   # bytes mapped, words held


CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch
import oracle_lib as O
from canvas_amd import Canvas
from canvas_amd.lib import CanvasError
mode = sys.argv[2]
cv = Canvas(0)
rng = np.random.RandomState(91)
per = []
for c in range(3):
    n = 30000 + 9000 * c
    x = rng.normal(50.0, 7.0, n)
    for k in range(6): a = rng.randint(0, n - 3000); x[a:a + rng.randint(300, 3000)] += rng.choice([-1.5, 1.2, 2.0])      # weak aberrations: long permutation loops
    per.append(np.round(x, 2))
per.append(np.round(rng.normal(50.0, 7.0, 180), 2))                                                                            # a non-hybrid chromosome (k_perm_small)
off = np.concatenate([[0], np.cumsum([len(p) for p in per])]).astype(np.int64)
cov = torch.from_numpy(np.concatenate(per)).to(cv.device)
exp, est = O.cbs_genome(per, 0.01, 10000, threads=4)
for call in range(2):
    seg_len, nseg, st = cv.cbs(cov, off, 0.01, 10000)
    got = seg_len.cpu().numpy()
    ok = all(int(nseg[c]) == len(exp[c]) and (got[off[c]:off[c] + nseg[c]] == exp[c]).all() for c in range(len(per))) and int(st[0]) == int(est[0]) and int(st[2]) == int(est[2]) and int(st[4]) == int(est[4])
    cs = cv.cbs_cache_stats(); dv = cv.cbs_device_stats()
    print("call", call, "ok", ok, "served", int(cs[0]), "own", int(cs[1]), "produced", int(cs[2]), "mapped", int(cs[4]), "permutations", int(st[2]), "verified", int(dv[4]), "violations", int(dv[5]), flush=True)
    if not ok: sys.exit(1)
    if int(dv[5]) != 0: sys.exit(2)
    if mode == "off" and (int(cs[0]) != 0 or int(cs[4]) != 0): sys.exit(3)
    if mode == "on" and (int(cs[0]) == 0 or int(cs[1]) != 0): sys.exit(4)
    if mode == "on" and call == 1 and int(cs[2]) != 0: sys.exit(5)            # the second call finds every word it reads
    if mode == "tiny" and (int(cs[0]) == 0 or int(cs[1]) == 0): sys.exit(6)   # both ways in one call: the cache's bound lies inside the streams
print("DONE")
'''


@pytest.mark.parametrize("mode,env", [("on", {}), ("on", {"CANVAS_CBS_CACHE_NO_VMM": "1"}), ("on", {"CANVAS_CBS_CACHE_JUMP_GENERATOR": "1"}), ("off", {"CANVAS_CBS_CACHE_GB": "0"}), ("tiny", {"CANVAS_CBS_CACHE_GB": "0.27"}),
                                      ("on", {"CANVAS_CBS_TEST_VERIFY": "1"}), ("tiny", {"CANVAS_CBS_CACHE_GB": "0.27", "CANVAS_CBS_TEST_VERIFY": "1"})])
def test_cbs_equals_the_oracle_however_a_batch_gets_its_draws(mode, env):
    """cache on (address-range form, fixed-allotment form, and extended by jump-ahead + plain recurrence instead of the strided generator), cache off, and a bound of four 64 MB pieces for four streams (every stream gets one piece: 16 M draws, less than
    the loops read) — each twice in one process; with CANVAS_CBS_TEST_VERIFY every interval the device returns is checked against the statistic computed in the reference's
    order from a host generator at the batch's position, and the batch's first cached words against that generator"""
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, mode], env=e, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "DONE" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])


def test_cached_stream_from_the_jump_ahead_generator_is_the_generators_output():
    """the same content check with the streams extended by k_mt_jump + k_mt_chunk (CANVAS_CBS_CACHE_JUMP_GENERATOR=1): chunk seams every 4 M words — where a wrong jump would show"""
    child = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle_lib as O
from canvas_amd import Canvas
cv = Canvas(0)
seeds = O.cbs_seeds(24)
CH = 4 << 20
c = 5
full = O.mt_u32(int(np.uint32(seeds[c])), 21 * CH + 5000)
spots = [0, 623, 624, 19937, CH - 1, CH, CH + 1] + [k * CH - 1000 for k in (1, 2, 3, 4, 7, 12, 13, 20)] + [int(x) for x in np.random.RandomState(8).randint(0, 20 * CH, 10)]
for pos in spots:
    got = cv.cbs_stream_read(c, pos, 2000)
    assert (got == full[pos:pos + 2000]).all(), (pos, np.nonzero(got != full[pos:pos + 2000])[0][:5])
got = cv.cbs_stream_read(11, 0, 9 * CH + 77)
assert (got == O.mt_u32(int(np.uint32(seeds[11])), 9 * CH + 77)).all()
print("DONE")
'''
    p = subprocess.run([sys.executable, "-c", child, ROOT], env=dict(os.environ, CANVAS_CBS_CACHE_JUMP_GENERATOR="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "DONE" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


def test_reading_beyond_the_bound_is_refused_not_wrong():
    child = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from canvas_amd import Canvas
from canvas_amd.lib import CanvasError
cv = Canvas(0)
cv.cbs_stream_read(0, 0, 1000)
try:
    cv.cbs_stream_read(0, 40 << 20, 1000)
    print("READ")
except CanvasError as e:
    print("REFUSED", e)
'''
    p = subprocess.run([sys.executable, "-c", child, ROOT], env=dict(os.environ, CANVAS_CBS_CACHE_GB="0.1"), capture_output=True, text=True, timeout=600)
    assert "REFUSED" in p.stdout and "READ" not in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
