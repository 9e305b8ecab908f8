"""canvas_amd/csrc/cbs_mt_jump.hpp (tools/gen_mt_jump.py): the compiled-in jump-ahead polynomial of MT19937 — the list of non-zero terms of x^(2^22) mod phi — must satisfy
u[k + 2^22] = XOR_i u[k + idx_i] on the untempered output stream of numpy's MT19937 (the generator the oracle's is pinned to, test_oracle_golden.py) for seeds the generator
script did not use, at the stream's very first outputs and far inside it.  k_mt_jump (cbs.hip) computes the generator states of the draw-stream cache's chunks with this table;
the states themselves are checked on the device by tests/test_cbs_stream_cache_gpu.py (cached words against the oracle's generator across chunk seams).  CPU only."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table():
    src = open(os.path.join(ROOT, "canvas_amd", "csrc", "cbs_mt_jump.hpp")).read()
    log2 = int(re.search(r"#define MT_JUMP_LOG2 (\d+)", src).group(1)); nterms = int(re.search(r"#define MT_JUMP_NTERMS (\d+)", src).group(1))
    body = src[src.index("{", src.index("MT_JUMP_IDX")) + 1:src.rindex("}")]
    idx = np.array([int(v) for v in body.replace("\n", " ").split(",") if v.strip()], np.int64)
    return log2, nterms, idx


def _untemper(y):
    y = y.copy(); y ^= y >> 18; y ^= (y << 15) & np.uint32(0xefc60000)
    t = y.copy()
    for _ in range(4):
        t = y ^ ((t << 7) & np.uint32(0x9d2c5680))
    y = t; t = y.copy()
    for _ in range(2):
        t = y ^ (t >> 11)
    return t


def test_table_shape():
    log2, nterms, idx = _table()
    assert log2 == 22 and len(idx) == nterms and (np.diff(idx) > 0).all() and idx[0] >= 0 and idx[-1] < 19937


def test_jump_identity_on_numpys_generator():
    log2, _, idx = _table()
    J = 1 << log2
    for seed in (7, 20260930):
        raw = np.random.RandomState(seed).randint(0, 2**32, size=2 * J + 19937 + 700, dtype=np.uint64).astype(np.uint32)
        u = _untemper(raw)
        for k in (0, 1, 623, 624, 4242, J - 1, J, J + 12345):
            assert np.bitwise_xor.reduce(u[k + idx]) == u[k + J], (seed, k)
