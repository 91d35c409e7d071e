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
