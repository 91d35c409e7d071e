"""Shared helpers for the tests: bit extraction from a uint32 MSB-first stream, table kinds,
named synthetic distributions."""
import numpy as np

import hvlc


def extract_bits(words_u32, bit_start, nbits):
    """Bits [bit_start, bit_start+nbits) of an MSB-first uint32 stream, re-packed from bit 0
    into uint32 words (last word zero padded)."""
    w = np.ascontiguousarray(words_u32, dtype=np.uint32)
    w0, sh = bit_start // 32, bit_start % 32
    nw = (nbits + 31) // 32
    seg = w[w0: w0 + nw + 1].astype(np.uint64)
    if seg.size < nw + 1:
        seg = np.concatenate([seg, np.zeros(nw + 1 - seg.size, dtype=np.uint64)])
    out = ((seg[:-1] << np.uint64(sh)) | (seg[1:] >> np.uint64(32 - sh))) & np.uint64(0xFFFFFFFF) if sh else seg[:-1].copy()
    out = out[:nw].astype(np.uint32)
    rem = nbits % 32
    if rem and nw:
        out[-1] &= np.uint32((0xFFFFFFFF << (32 - rem)) & 0xFFFFFFFF)
    return out


def mask_codes(cw, cl):
    return (cw.astype(np.uint64) & ((np.uint64(1) << cl.astype(np.uint64)) - np.uint64(1))).astype(np.uint32)


def random_table(rng, kind):
    if kind == "any":
        cl = rng.integers(0, 32, 256)
    elif kind == "short":
        cl = rng.integers(1, 5, 256)
    elif kind == "long":
        cl = rng.integers(24, 32, 256)
    elif kind == "all31":
        cl = np.full(256, 31)
    elif kind == "alt1_31":
        cl = np.where(np.arange(256) % 2 == 0, 1, 31)
    elif kind == "zeros":
        cl = np.where(rng.random(256) < 0.7, 0, rng.integers(1, 32, 256))
    elif kind == "allzero":
        cl = np.zeros(256, dtype=np.int64)
    elif kind == "len4":
        cl = np.full(256, 4)
    elif kind == "len16":
        cl = np.full(256, 16)
    elif kind == "len17":
        cl = np.full(256, 17)
    else:
        raise ValueError(kind)
    cl = cl.astype(np.uint32)
    cw = mask_codes(rng.integers(0, 1 << 32, 256, dtype=np.uint64).astype(np.uint32), cl)
    return cw, cl


G1_LIVE = {12: 21, 16: 20, 19: 19, 25: 8, 37: 4, 47: 21, 55: 17, 74: 20, 86: 20, 99: 5, 101: 17, 104: 20,
           111: 11, 112: 14, 116: 1, 129: 2, 131: 7, 138: 6, 142: 20, 156: 20, 164: 3, 165: 20, 168: 19,
           171: 9, 176: 10, 210: 13, 216: 20, 223: 20, 229: 20, 231: 15, 241: 12, 255: 20}


def dist_weights(name, period=None):
    """Symbol weights of the named benchmark distributions."""
    w = np.zeros(256, dtype=np.float64)
    if name == "H2.2":       # empirical distribution of the shipped test file
        assert period is not None
        w = np.bincount(period, minlength=256).astype(np.float64)
    elif name == "H4.0":     # 16 symbols (byte values 16j+3) uniform
        w[np.arange(16) * 16 + 3] = 1.0
    elif name == "H1.0":     # p(0)=0.83, p(i)=0.17*2^-i for i=1..24, remainder on 24
        w[0] = 0.83
        for i in range(1, 25):
            w[i] = 0.17 * 2.0 ** -i
        w[24] += 1.0 - w.sum()
    elif name == "H8.0":     # uniform over all 256 symbols (incompressible; general path)
        w[:] = 1.0
    else:
        raise ValueError(name)
    return w


SEEDS = {"H2.2": 0x5EED0002, "H4.0": 0x5EED0003, "H1.0": 0x5EED0004, "H8.0": 0x5EED0008}


def synth_on_device(enc, name, n_bytes, period=None, first_draw=0, seed=None):
    """Generate the named synthetic stream on the GPU, histogram it on the GPU, build the
    Huffman table on the host.  Returns (d_in uint8 tensor, cw, cl, hist)."""
    cdf = hvlc.cdf32_from_counts(dist_weights(name, period))
    d_in = enc.gen_symbols(n_bytes, SEEDS[name] if seed is None else seed, cdf, first_draw=first_draw)
    hist = enc.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist)
    return d_in, cw, cl, hist
