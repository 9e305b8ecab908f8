"""The host-side packers of the packed per-base planes (include/canvas_hip.h, "packed per-base inputs") against a numpy restatement of the format.
Plain CPU code of the library: no GPU involved."""
import numpy as np
import pytest

from canvas_amd.lib import pack_reference_host, pack_hits_host, packed_plane_words


def _bits_to_words(bits):
    """bool[64 k] -> u64[k], bit i of word w = bits[64 w + i]"""
    return np.packbits(bits.astype(np.uint8), bitorder="little").view(np.uint64)


def ref_planes_numpy(bases, mask_words, length):
    W = packed_plane_words(length)
    pos = np.zeros(W * 64, bool); gc = np.zeros(W * 64, bool)
    mbits = np.unpackbits(mask_words.view(np.uint8), bitorder="little")[:length].astype(bool)
    pos[:length] = mbits
    lb = bases[:length] | 0x20
    gc[:length] = (lb == ord("c")) | (lb == ord("g"))
    out = np.empty(2 * W, np.uint64)
    out[0::2] = _bits_to_words(pos); out[1::2] = _bits_to_words(gc)
    non_n = np.flatnonzero(bases[:length] != ord("n"))
    return out, (int(non_n[0]) if len(non_n) else length)


def hit_planes_numpy(hits, length):
    W = packed_plane_words(length)
    h = np.zeros(W * 64, np.uint8); h[:length] = np.minimum(hits[:length], 15)
    out = np.empty(4 * W, np.uint64)
    for k in range(4):
        out[k::4] = _bits_to_words((h >> k) & 1)
    return out, int((hits[:length] > 15).sum())


@pytest.mark.parametrize("length", [1, 63, 64, 65, 4095, 4096, 4097, 100_003, 1_000_000])
@pytest.mark.parametrize("threads", [1, 5])
def test_host_packers_match_the_format(length, threads):
    rng = np.random.RandomState(length % 9973)
    bases = rng.choice(np.frombuffer(b"ACGTacgtnN", np.uint8), length)
    lead = int(rng.randint(0, min(length, 300)))
    bases[:lead] = ord("n")
    mask = _bits_to_words(np.concatenate([rng.rand(length) < 0.8, np.ones((-length) % 64, bool)]))   # garbage bits beyond len must be dropped
    hits = rng.poisson(0.3, length).astype(np.uint8)
    spikes = rng.randint(0, length, max(1, length // 50))
    hits[spikes] = rng.randint(9, 256, len(spikes)).astype(np.uint8)
    ref, p0 = pack_reference_host(bases, mask, length, threads=threads)
    eref, ep0 = ref_planes_numpy(bases, mask, length)
    assert p0 == ep0
    assert (ref == eref).all()
    planes, sat = pack_hits_host(hits, length, threads=threads)
    eplanes, esat = hit_planes_numpy(hits, length)
    assert sat == esat
    assert (planes == eplanes).all()


def test_all_n_chromosome_has_pos0_at_len():
    L = 5000
    bases = np.full(L, ord("n"), np.uint8)
    mask = np.zeros((L + 63) // 64, np.uint64)
    ref, p0 = pack_reference_host(bases, mask, L)
    assert p0 == L and not ref.any()
