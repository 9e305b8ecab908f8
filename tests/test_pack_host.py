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


def hits2_numpy(hits, length):
    """the two-bit wire form from the four planes: lo = {b0, b1} per word, hdr = {xmask, xoff} per tile, extras = {b2, b3} of the words that have one"""
    planes, sat = hit_planes_numpy(hits, length)
    W = packed_plane_words(length)
    b = planes.reshape(W, 4)
    lo = np.ascontiguousarray(b[:, :2]).reshape(-1)
    has = (b[:, 2] | b[:, 3]) != 0
    extras = np.ascontiguousarray(b[has][:, 2:]).reshape(-1)
    hdr = np.zeros((W // 64, 2), np.uint64)
    hb = has.reshape(W // 64, 64)
    hdr[:, 0] = np.packbits(hb.astype(np.uint8), axis=1, bitorder="little").view(np.uint64).reshape(-1)
    hdr[:, 1] = np.concatenate([[0], np.cumsum(hb.sum(axis=1))[:-1]]).astype(np.uint64)
    return lo, hdr.reshape(-1), extras, int(has.sum()), sat


@pytest.mark.parametrize("length", [63, 4097, 100_003, 1_000_000])
@pytest.mark.parametrize("threads", [1, 7])
def test_two_bit_wire_form_of_the_hit_planes(length, threads):
    from canvas_amd.lib import pack_hits2_host
    rng = np.random.RandomState(length % 7919 + threads)
    hits = rng.poisson(0.3, length).astype(np.uint8)
    spikes = rng.randint(0, length, max(1, length // 40))
    hits[spikes] = rng.randint(3, 256, len(spikes)).astype(np.uint8)
    lo, hdr, extras, nx, sat = pack_hits2_host(hits, length, threads=threads)
    elo, ehdr, eextras, enx, esat = hits2_numpy(hits, length)
    assert (nx, sat) == (enx, esat)
    assert (lo == elo).all() and (hdr == ehdr).all() and (extras[:2 * nx] == eextras).all()
    # a buffer that is too small reports how much is needed
    from canvas_amd.lib import load_library, _host_addr
    import ctypes as C
    if nx > 1:
        small = np.zeros(2, np.uint64); need = C.c_int64(0)
        rc = load_library().canvas_pack_hits2_host(_host_addr(hits), C.c_int64(length), _host_addr(lo), _host_addr(hdr), _host_addr(small), C.c_int64(1), C.byref(need), None, threads)
        assert rc == -4 and need.value == nx
