"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when it has been
built in this container, the compiled reference (oracle/_ref/libref_cpuencode.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (huffman-gpu_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libref_cpuencode.so")
REF_HUFFTABLE = os.path.join(ORACLE_DIR, "_ref", "ref_hufftable")

_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


def build_oracle():
    """(Re)build oracle/liboracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _load():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    lib.oracle_vlc_encode.argtypes = [_u32p, C.c_uint32, _u32p, _u32p, _u32p, _u32p]
    lib.oracle_vlc_encode.restype = None
    lib.oracle_vlc_encode64.argtypes = [_u32p, C.c_uint64, _u32p, _u32p, _u32p]
    lib.oracle_vlc_encode64.restype = C.c_uint64
    lib.oracle_segment_offsets.argtypes = [_u32p, C.c_uint64, C.c_uint32, _u32p, _u32p, _u64p]
    lib.oracle_segment_offsets.restype = C.c_uint64
    lib.oracle_generate_codewords.argtypes = [_u32p, _u32p, C.c_uint32]
    lib.oracle_generate_codewords.restype = None
    lib.oracle_gen_symbols.argtypes = [_u8p, C.c_uint64, C.c_uint64, C.c_uint64, _u32p]
    lib.oracle_gen_symbols.restype = None
    lib.oracle_histogram.argtypes = [_u8p, C.c_uint64, _u64p]
    lib.oracle_histogram.restype = None
    lib.oracle_bit_append.argtypes = [_u32p, C.c_uint64, _u32p, C.c_uint64]
    lib.oracle_bit_append.restype = C.c_uint64
    lib.oracle_vlc_decode.argtypes = [_u32p, C.c_uint64, _u32p, C.c_uint64, _u32p, _u32p]
    lib.oracle_vlc_decode.restype = C.c_uint64
    lib.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
    lib.oracle_fnv1a64.restype = C.c_uint64
    lib.oracle_vlc_encode_mt.argtypes = [_u32p, C.c_uint64, _u32p, C.c_uint64, _u32p, _u32p,
                                         C.c_int, C.POINTER(C.c_double)]
    lib.oracle_vlc_encode_mt.restype = C.c_uint64
    lib.oracle_compare_stream_mt.argtypes = [_u32p, C.c_uint64, _u32p, _u32p, _u32p, C.c_uint64, _u64p, C.c_uint64,
                                             C.c_uint64, C.c_int, _u64p]
    lib.oracle_compare_stream_mt.restype = C.c_uint64
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _tables(codewords, codewordlens):
    cw = np.ascontiguousarray(codewords, dtype=np.uint32)
    cl = np.ascontiguousarray(codewordlens, dtype=np.uint32)
    assert cw.shape == (256,) and cl.shape == (256,)
    return cw, cl


def out_capacity_words(n_words, codewordlens):
    """Words the caller must provide for outdata: floor(bits/32)+1 at worst."""
    mx = int(np.max(codewordlens)) if len(codewordlens) else 0
    return (int(n_words) * 4 * mx) // 32 + 2


def encode(indata, codewords, codewordlens):
    """oracle_vlc_encode64: returns (words[floor(bits/32)+1], total_bits)."""
    x = np.ascontiguousarray(indata, dtype=np.uint32)
    cw, cl = _tables(codewords, codewordlens)
    out = np.empty(out_capacity_words(x.size, cl), dtype=np.uint32)
    bits = lib().oracle_vlc_encode64(_ptr(x, _u32p), x.size, _ptr(out, _u32p),
                                     _ptr(cw, _u32p), _ptr(cl, _u32p))
    return out[: (bits >> 5) + 1].copy(), int(bits)


def encode32(indata, codewords, codewordlens):
    """oracle_vlc_encode (reference signature): returns (words, outsize_bytes)."""
    x = np.ascontiguousarray(indata, dtype=np.uint32)
    cw, cl = _tables(codewords, codewordlens)
    out = np.empty(out_capacity_words(x.size, cl), dtype=np.uint32)
    osz = C.c_uint32(0)
    lib().oracle_vlc_encode(_ptr(x, _u32p), x.size, _ptr(out, _u32p), C.byref(osz),
                            _ptr(cw, _u32p), _ptr(cl, _u32p))
    return out, int(osz.value)


def segment_offsets(indata, codewordlens, seg_words=256):
    """(counts u32[S], exclusive offsets u64[S], total_bits)."""
    x = np.ascontiguousarray(indata, dtype=np.uint32)
    cl = np.ascontiguousarray(codewordlens, dtype=np.uint32)
    S = (x.size + seg_words - 1) // seg_words
    counts = np.zeros(S, dtype=np.uint32)
    offs = np.zeros(S, dtype=np.uint64)
    total = lib().oracle_segment_offsets(_ptr(x, _u32p), x.size, seg_words, _ptr(cl, _u32p),
                                         _ptr(counts, _u32p), _ptr(offs, _u64p))
    return counts, offs, int(total)


def generate_codewords(num_symbols=256):
    cw = np.zeros(num_symbols, dtype=np.uint32)
    cl = np.zeros(num_symbols, dtype=np.uint32)
    lib().oracle_generate_codewords(_ptr(cw, _u32p), _ptr(cl, _u32p), num_symbols)
    return cw, cl


def gen_symbols(n_bytes, seed, cdf32, first_draw=0):
    assert n_bytes % 2 == 0
    out = np.empty(n_bytes, dtype=np.uint8)
    cdf = np.ascontiguousarray(cdf32, dtype=np.uint32)
    assert cdf.shape == (256,)
    lib().oracle_gen_symbols(_ptr(out, _u8p), n_bytes, seed, first_draw, _ptr(cdf, _u32p))
    return out


def histogram(data_u8):
    d = np.ascontiguousarray(data_u8).view(np.uint8)
    h = np.zeros(256, dtype=np.uint64)
    lib().oracle_histogram(_ptr(d, _u8p), d.size, _ptr(h, _u64p))
    return h


def bit_append(dst, dst_bits, src, nbits):
    """Append the first nbits bits of src to the stream in dst (in place).  Returns the new bit count."""
    assert dst.dtype == np.uint32 and src.dtype == np.uint32
    return int(lib().oracle_bit_append(_ptr(dst, _u32p), int(dst_bits), _ptr(np.ascontiguousarray(src), _u32p), int(nbits)))


def decode(bits, num_words, codewords, codewordlens, start_bit=0):
    """CPU decoder (prefix-free table).  Returns (words, bits consumed)."""
    cw, cl = _tables(codewords, codewordlens)
    out = np.zeros(int(num_words), dtype=np.uint32)
    used = int(lib().oracle_vlc_decode(_ptr(np.ascontiguousarray(bits), _u32p), int(start_bit), _ptr(out, _u32p),
                                       int(num_words), _ptr(cw, _u32p), _ptr(cl, _u32p)))
    return out, used


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().oracle_fnv1a64(a.ctypes.data, a.nbytes))


def encode_mt(indata, codewords, codewordlens, threads):
    """Timed multi-threaded CPU baseline.  Returns (total_bits, seconds)."""
    x = np.ascontiguousarray(indata, dtype=np.uint32)
    cw, cl = _tables(codewords, codewordlens)
    per = (x.size + threads - 1) // threads + 256
    stride = out_capacity_words(per, cl) + 2
    scratch = np.empty(stride * threads, dtype=np.uint32)
    sec = C.c_double(0.0)
    bits = lib().oracle_vlc_encode_mt(_ptr(x, _u32p), x.size, _ptr(scratch, _u32p), scratch.size,
                                      _ptr(cw, _u32p), _ptr(cl, _u32p), threads, C.byref(sec))
    return int(bits), float(sec.value)


def compare_stream_mt(indata, codewords, codewordlens, stream, seg_offsets, total_bits, chunk_segments=1024, threads=None):
    """Whole-stream check in chunks on all host threads: (number of differing chunks, first bad segment)."""
    x = np.ascontiguousarray(indata, dtype=np.uint32)
    st = np.ascontiguousarray(stream, dtype=np.uint32)
    offs = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
    cw, cl = _tables(codewords, codewordlens)
    assert offs.size == (x.size + 255) // 256
    first = C.c_uint64(0)
    bad = lib().oracle_compare_stream_mt(_ptr(x, _u32p), x.size, _ptr(cw, _u32p), _ptr(cl, _u32p), _ptr(st, _u32p), st.size,
                                         _ptr(offs, _u64p), int(total_bits), int(chunk_segments),
                                         int(threads or os.cpu_count() or 1), C.byref(first))
    return int(bad), int(first.value)


# ---------------------------------------------------------------------------
# the compiled reference (only in the build container; never on the GPU box's
# critical path -- tests that need it skip when it is absent)
# ---------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref_encode32(indata, codewords, codewordlens):
    """The REFERENCE's cpu_vlc_encode (cpuencode.cpp:12-46) via oracle/_ref."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.cpu_vlc_encode.argtypes = [_u32p, C.c_uint32, _u32p, _u32p, _u32p, _u32p]
        _ref.cpu_vlc_encode.restype = None
    x = np.ascontiguousarray(indata, dtype=np.uint32).copy()
    cw, cl = _tables(codewords, codewordlens)
    cw, cl = cw.copy(), cl.copy()
    out = np.empty(out_capacity_words(x.size, cl), dtype=np.uint32)
    osz = C.c_uint32(0)
    _ref.cpu_vlc_encode(_ptr(x, _u32p), x.size, _ptr(out, _u32p), C.byref(osz),
                        _ptr(cw, _u32p), _ptr(cl, _u32p))
    return out, int(osz.value)
