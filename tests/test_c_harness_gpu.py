"""-m gpu: the C host (huffman-gpu_amd/host/hvlc_bench, plain C over the C ABI, system HIP runtime, no torch)
run on the shipped fixture; its dumped stream / offsets are compared with the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "huffman-gpu_amd", "host", "hvlc_bench")


def test_c_host_end_to_end(golden, tmp_path):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-s", "-C", os.path.dirname(BIN)], check=True)
    raw = np.tile(golden["period"], 4)[: (1 << 20) - 4096 - 3]      # ragged: not a multiple of 4 or of a tile
    inp = tmp_path / "in.bin"
    raw.tofile(inp)
    tab = os.path.join(ROOT, "tests", "golden", "g1_table.bin")
    r = subprocess.run([BIN, str(inp), tab, "--iters", "3", "--dump", str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    x = raw[: raw.size // 4 * 4].view(np.uint32)                   # tail bytes dropped like load_data.h:19
    ref, bits = O.encode(x, golden["cw"], golden["cl"])
    _, offs, _ = O.segment_offsets(x, golden["cl"])
    assert info["total_bits"] == bits and info["outsize"] == (bits + 7) // 8
    got = np.fromfile(str(tmp_path / "out.stream.bin"), dtype=np.uint32)
    assert np.array_equal(got, ref)
    assert np.array_equal(np.fromfile(str(tmp_path / "out.offsets.bin"), dtype=np.uint64), offs)


def test_c_host_builds_its_table_verifies_and_logs(golden, tmp_path):
    """The file front end: table built from the file (device histogram + hvlc_build_table), entropy
    line as the reference prints it, device round trip (--verify), stats series file in the stats_logger format;
    the dumped stream is the oracle's encoding of the file with the dumped table."""
    subprocess.run(["make", "-s", "-C", os.path.dirname(BIN)], check=True)
    raw = np.tile(golden["period"], 3)[: 700001]                     # not a multiple of 4
    inp = tmp_path / "in.bin"
    raw.tofile(inp)
    for _ in range(2):
        r = subprocess.run([BIN, str(inp), "-", "--iters", "2", "--dump", str(tmp_path / "o"), "--verify",
                            "--stats", str(tmp_path)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    assert f"{inp}, 700000 bytes, entropy " in r.stdout
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["verified"] == 1 and info["tail_bytes_dropped"] == 1
    tab = np.fromfile(str(tmp_path / "o.table.bin"), dtype=np.uint32)
    x = raw[:700000].view(np.uint32)
    ref, bits = O.encode(x, tab[:256], tab[256:])
    assert info["total_bits"] == bits
    assert np.array_equal(np.fromfile(str(tmp_path / "o.stream.bin"), dtype=np.uint32), ref)
    hist = np.bincount(raw[:700000], minlength=256)
    assert int((hist * tab[256:]).sum()) == bits                    # the table is a Huffman table of this file
    series = open(tmp_path / "encoding__0_hvlc.txt").read().splitlines()
    assert series[0] == "SERIES_NAME" and series[1] == "hvlc" and "__DATA__" in series
    assert len(series) - series.index("__DATA__") - 1 == 2          # one point per run
