"""Synthetic inputs for tests and bench.  The per-base generator is a counter-based integer
hash so that canvas_amd/csrc/synth.hip (on the GPU) and this numpy mirror produce identical bytes."""
import numpy as np

# GRCh38 primary assembly lengths chr1..22, X, Y
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415]
CHROM_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
IS_AUTOSOME = np.array([1] * 22 + [0, 0], np.uint8)

_M = np.uint32


def mix32(x):
    x = x.astype(np.uint32, copy=True)
    x ^= x >> _M(16); x *= _M(0x7feb352d); x ^= x >> _M(15); x *= _M(0x846ca68b); x ^= x >> _M(16)
    return x


def H(seed, chrom, stream, p):
    with np.errstate(over="ignore"):
        a = mix32(p.astype(np.uint32) + _M((0x9e3779b9 * (stream + 1)) & 0xFFFFFFFF))
        k = _M((seed * 0x85ebca6b + chrom * 0xc2b2ae35 + stream) & 0xFFFFFFFF)
        return mix32(a ^ k)


def poisson_thresholds(rate, purity=1.0, flat=False):
    """[5 CN][16 gc levels][8] uint32 cumulative thresholds: hits = #(u >= thr).  purity < 1: a tumour sample whose copy-number segments are diluted by
    normal cells (effective CN = purity * CN + (1 - purity) * 2); flat: a matched normal (every segment diploid)."""
    import math
    thr = np.zeros((5, 16, 8), np.uint32)
    for cn in range(5):
        cnf = 0.02 if cn == 0 else cn / 2.0
        cnf = 1.0 if flat else purity * cnf + (1.0 - purity) * 1.0
        for lvl in range(16):
            g = 26 + 2 * lvl
            lam = rate * cnf * (1 + 0.004 * (g - 41) - 0.0003 * (g - 41) ** 2)
            cdf = 0.0
            for k in range(8):
                cdf += math.exp(-lam) * lam ** k / math.factorial(k)
                thr[cn, lvl, k] = min(0xFFFFFFFF, int(cdf * 4294967296.0))
    return thr


def chrom_params(chrom, length):
    gap0 = min(10000, length // 8)
    g1s = int(length * 0.4) & ~63
    g1e = g1s + (int(length * 0.02) & ~63)
    base_cn = 1 if chrom == 23 else 2
    return gap0, g1s, g1e, base_cn


def generate_chromosome(seed, chrom, length, rate, thr=None, hit_seed=None, with_fraglen=False):
    """numpy mirror of k_synth: returns (bases u8[L], hits u8[L], mask u8[ceil(L/64)*8]) [+ Int16 fragment lengths with with_fraglen].
    hit_seed (default: seed) drives the copy-number segments and the hit draws: samples that share `seed` share the reference."""
    if thr is None:
        thr = poisson_thresholds(rate)
    if hit_seed is None:
        hit_seed = seed
    CH = 1 << 22                      # every value is a function of the position alone: long chromosomes are generated in pieces (bounded memory)
    if length > CH:
        parts = [_generate_range(seed, chrom, length, thr, hit_seed, with_fraglen, a, min(length, a + CH)) for a in range(0, length, CH)]
        b = np.concatenate([q[0] for q in parts]); h = np.concatenate([q[1] for q in parts]); m = np.concatenate([q[2] for q in parts])
        out = (b, h, _pack_mask_bits(m, length))
        return out + (np.concatenate([q[3] for q in parts]),) if with_fraglen else out
    b, h, m, fl = _generate_range(seed, chrom, length, thr, hit_seed, with_fraglen, 0, length)
    return (b, h, _pack_mask_bits(m, length), fl) if with_fraglen else (b, h, _pack_mask_bits(m, length))


def _pack_mask_bits(m, length):
    words = (length + 63) // 64
    bits = np.zeros(words * 64, np.uint8); bits[:length] = m
    return np.packbits(bits, bitorder="little")


def _generate_range(seed, chrom, total_length, thr, hit_seed, with_fraglen, lo, hi):
    gap0, g1s, g1e, base_cn = chrom_params(chrom, total_length)
    p = np.arange(lo, hi, dtype=np.uint32)
    length = hi - lo
    gap = (p < gap0) | ((p >= g1s) & (p < g1e))
    cell1k = p >> _M(10); off = p & _M(1023)
    hc = H(seed, chrom, 1, cell1k)
    m = np.ones(length, bool)
    nonu = (hc % _M(100)) < 22
    a = (hc >> _M(8)) & _M(511); ln = _M(300) + ((hc >> _M(17)) % _M(724))
    m &= ~(nonu & (off >= a) & (off < a + ln))
    hp = H(seed, chrom, 4, p)
    m &= ~((hp % _M(100)) < 3)
    lvl = ((H(seed, chrom, 2, p >> _M(16)) & _M(7)) + (H(seed, chrom, 5, p >> _M(12)) & _M(7)) + _M(1)) & _M(15)
    gcfrac = _M(26) + _M(2) * lvl
    ub = H(seed, chrom, 3, p)
    isgc = ((ub & _M(0xFFFF)) * _M(100)) < (gcfrac << _M(16))
    which = ((ub >> _M(16)) & _M(1)).astype(bool)
    b = np.where(isgc, np.where(which, ord('G'), ord('C')), np.where(which, ord('A'), ord('T'))).astype(np.uint8)
    b = np.where(m, b, b | 0x20).astype(np.uint8)
    hcn = H(hit_seed, chrom, 6, p >> _M(20)) % _M(1000)
    cn = np.full(length, base_cn, np.int64)
    cn = np.where(hcn < 15, base_cn - 1, np.where(hcn < 30, base_cn + 1, np.where(hcn < 33, 0, np.where(hcn < 36, base_cn + 2, cn))))
    cn = np.minimum(cn, 4)
    u = H(hit_seed, chrom, 7, p)
    t = thr.reshape(-1, 8)[cn * 16 + lvl.astype(np.int64)]
    h = (u[:, None] >= t).sum(1).astype(np.uint8)
    m &= ~gap
    b[gap] = ord('n')
    h[~m] = 0
    fl = None
    if with_fraglen:
        uf = H(hit_seed, chrom, 8, p)
        ssum = (uf & _M(255)) + ((uf >> _M(8)) & _M(255)) + ((uf >> _M(16)) & _M(255)) + (uf >> _M(24))
        fl = np.where(h > 0, _M(143) + ssum * _M(60) // _M(148), 0).astype(np.int16)
    return b, h, m, fl


def generate_bins(seed, n, nchr=24, lengths=None):
    """Per-bin SoA resembling CanvasBin output at 30x: returns dict of numpy arrays."""
    rng = np.random.RandomState(seed)
    lengths = np.array(GRCH38[:nchr] if lengths is None else lengths, np.float64)
    per = np.maximum(12, (n * lengths / lengths.sum()).astype(np.int64))
    chr_id = np.repeat(np.arange(nchr, dtype=np.int32), per)
    N = len(chr_id)
    size = np.exp(rng.normal(np.log(1050), 0.25, N))
    tail = rng.rand(N) < 0.02
    size[tail] *= 10 ** rng.uniform(1, 3, tail.sum())
    size = np.maximum(100, size).astype(np.int64)
    start = np.zeros(N, np.int64)
    for c in range(nchr):
        s = chr_id == c
        start[s] = 10000 + np.concatenate([[0], np.cumsum(size[s])[:-1]])
    stop = start + size
    gc = np.clip(np.round(rng.normal(41, 6, N)), 0, 100).astype(np.int32)
    # CN segments
    cn = np.full(N, 2.0)
    i = 0
    while i < N:
        ln = rng.geometric(1 / 800.0)
        r = rng.rand()
        v = 2.0
        if r < 0.015: v = 1.0
        elif r < 0.03: v = 3.0
        elif r < 0.0315: v = 0.05
        elif r < 0.033: v = 4.0
        cn[i:i + ln] = v
        i += ln
    mean = 100 * cn / 2 * (1 + 0.004 * (gc - 41) - 0.0003 * (gc - 41.0) ** 2)
    mean = np.maximum(mean, 0.5)
    r = 60.0
    count = rng.negative_binomial(r, r / (r + mean)).astype(np.float32)
    return dict(chr=chr_id, start=start.astype(np.int32), stop=np.minimum(stop, 2**31 - 1).astype(np.int32), gc=gc, count=count)
