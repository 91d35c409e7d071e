"""-m gpu: the file codec (huffman-gpu_amd/host/hvlc_file, plain C over the C ABI): compress a file into the container
(table + per-segment bit counts + stream), decompress it again (hvlc_exclusive_scan -> offsets, hvlc_decode), compare
the bytes; and read the container from Python to check that its stream IS the oracle's encoding of the file with the
container's table and that its segment bit counts are the oracle's."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "huffman-gpu_amd", "host", "hvlc_file")


def _roundtrip(raw, tmp_path, name):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-s", "-C", os.path.dirname(BIN)], check=True)
    src, box, back = tmp_path / f"{name}.bin", tmp_path / f"{name}.hvlc", tmp_path / f"{name}.out"
    raw.tofile(src)
    r = subprocess.run([BIN, "c", str(src), str(box)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    r = subprocess.run([BIN, "d", str(box), str(back)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.fromfile(back, dtype=np.uint8), raw), name
    # the container, read independently
    c = open(box, "rb").read()
    magic, nbytes, nwords, bits, nseg = struct.unpack_from("<8sQQQQ", c, 0)
    assert magic == b"HVLC\x01\x00\x00\x00" and nbytes == raw.size and nwords == raw.size // 4 and nseg == (nwords + 255) // 256
    assert info["total_bits"] == bits and info["container_bytes"] == len(c)
    tab = np.frombuffer(c, dtype=np.uint32, count=512, offset=40)
    seg = np.frombuffer(c, dtype=np.uint16, count=nseg, offset=40 + 2048)
    off_stream = 40 + 2048 + (2 * nseg + 7) // 8 * 8
    stream = np.frombuffer(c, dtype=np.uint32, count=bits // 32 + 1, offset=off_stream)
    x = raw[: nwords * 4].view(np.uint32)
    ref, rbits = O.encode(x, tab[:256], tab[256:])
    assert rbits == bits and np.array_equal(stream, ref), name
    counts, _, _ = O.segment_offsets(x, tab[256:])
    assert np.array_equal(seg.astype(np.uint32), counts), name
    return info


def test_file_codec_round_trips(golden, tmp_path):
    rng = np.random.default_rng(5)
    info = _roundtrip(np.tile(golden["period"], 4), tmp_path, "g1")            # the reference's test file
    assert info["ratio"] < 0.29
    _roundtrip(np.tile(golden["period"], 3)[:700003], tmp_path, "ragged")       # not a multiple of 4, partial segment and tile
    _roundtrip(rng.integers(0, 256, 300_001, dtype=np.uint8), tmp_path, "uniform")      # incompressible: 8-bit codes
    skew = rng.choice(256, size=2_000_000, p=np.r_[0.9, np.full(255, 0.1 / 255)]).astype(np.uint8)
    _roundtrip(skew, tmp_path, "skew")                                          # long codes: the decoder's second level
    _roundtrip(np.full(5000, 7, dtype=np.uint8), tmp_path, "one_symbol")
    _roundtrip(np.array([1, 2, 3], dtype=np.uint8), tmp_path, "three_bytes")    # no whole word at all
    _roundtrip(np.zeros(0, dtype=np.uint8), tmp_path, "empty")


def test_file_codec_rejects_a_damaged_container(golden, tmp_path):
    raw = np.tile(golden["period"], 1)[:100000]
    _roundtrip(raw, tmp_path, "ok")
    c = bytearray(open(tmp_path / "ok.hvlc", "rb").read())
    c[16] ^= 1                                                                  # num_words no longer matches num_bytes
    bad = tmp_path / "bad.hvlc"
    open(bad, "wb").write(c)
    r = subprocess.run([BIN, "d", str(bad), str(tmp_path / "bad.out")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not a valid hvlc container" in r.stderr
