"""-m gpu: the C multi-GPU host (huffman-gpu_amd/host/hvlc_mgpu: one process, hipSetDevice per GPU, ncclCommInitAll,
per-device hvlc_encode, ncclAllGather of one uint64 per rank, hvlc_rebase_offsets_dev, peer-mapped hvlc_merge_shard) --
the role of the reference's driver runVLCTest (main_test_cu.cu:52-180) for the sharded job.  Its dumps (per-device streams
and offsets, the job's table, the merged stream) are compared with the oracle's single-stream encoding of the whole job.
N = 1 always runs (RCCL with one rank); N = 2 runs where two devices are present."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import hvlc
import oracle_lib as O
from util import dist_weights, extract_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "huffman-gpu_amd", "host", "hvlc_mgpu")
PERIOD = os.path.join(ROOT, "tests", "golden", "g1_period.bin")


def _run(gpus, mib, dist, tmp_path, golden):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-s", "-C", os.path.dirname(BIN)], check=True)
    pre = str(tmp_path / f"job{gpus}")
    r = subprocess.run([BIN, "--gpus", str(gpus), "--mib", str(mib), "--iters", "3", "--dist", dist, "--period", PERIOD,
                        "--merge", "--dump", pre], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["n_gpus"] == gpus and info["stitch_ok"] is True and info["merged"] is True
    # the job stream: ONE generation from draw 0 (the devices generated their shards from draw g * shard_bytes / 2)
    shard = mib << 20
    cdf = hvlc.cdf32_from_counts(dist_weights(dist, golden["period"]))
    whole = O.gen_symbols(gpus * shard, 0x5EED0005, cdf)
    tab = np.fromfile(pre + ".table.bin", dtype=np.uint32)
    cw, cl = tab[:256], tab[256:]
    ecw, ecl = hvlc.build_table(O.histogram(whole))                 # the table of the JOB's histogram
    assert np.array_equal(cw, ecw) and np.array_equal(cl, ecl)
    x = whole.view(np.uint32)
    ref, rbits = O.encode(x, cw, cl)
    _, roffs, _ = O.segment_offsets(x, cl)
    assert info["job_bits"] == rbits
    offs = np.concatenate([np.fromfile(f"{pre}.dev{g}.offsets.bin", dtype=np.uint64) for g in range(gpus)])
    assert np.array_equal(offs, roffs)                              # stitched offsets == single-stream offsets
    S = shard // 1024
    for g in range(gpus):                                           # every shard stream == its bits of the single stream
        st = np.fromfile(f"{pre}.dev{g}.stream.bin", dtype=np.uint32)
        b0 = int(roffs[g * S]); b1 = int(roffs[(g + 1) * S]) if g + 1 < gpus else rbits
        assert np.array_equal(extract_bits(st, 0, b1 - b0), extract_bits(ref, b0, b1 - b0))
    merged = np.fromfile(pre + ".merged.bin", dtype=np.uint32)      # the seam-merged stream == cpu_vlc_encode of the whole job
    assert np.array_equal(merged, ref)
    return info


def test_c_mgpu_host_one_gpu(golden, tmp_path):
    _run(1, 48, "H2.2", tmp_path, golden)


def test_c_mgpu_host_two_gpus(golden, tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); the same code path runs at N = 1 above")
    _run(2, 48, "H2.2", tmp_path, golden)
