#!/usr/bin/env python3
"""Developer tool: the PCIe-inclusive rate of the host-buffer entry point gpu_vlc_encode (cpuencode.h:4-7's signature:
pageable host buffers in, host stream out) next to the CPU reference on the same buffers.  Never bench.py's `value`.
    python tools/bench_host_boundary.py [mib] [repeats]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import synth_on_device
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = mib << 20
enc = hvlc.Encoder(0, n // 4)
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
d_in, cw, cl, hist = synth_on_device(enc, "H2.2", n, period)
x = d_in.cpu().numpy().view(np.uint8).view(np.uint32).copy()
bits_expected = int((hist * cl.astype(np.uint64)).sum())
del d_in; enc.close(); torch.cuda.empty_cache()
ts = []
for _ in range(reps + 1):
    t0 = time.perf_counter()
    out, osz = hvlc.gpu_vlc_encode(x, cw, cl)
    ts.append(time.perf_counter() - t0)
    assert osz == (bits_expected >> 5) * 4 + ((bits_expected & 31) + 7) // 8, (osz, bits_expected)
ts = sorted(ts[1:])
print(json.dumps({"entry": "gpu_vlc_encode (pageable host buffers in and out, device buffers allocated per call)",
                  "mib": mib, "repeats": reps, "seconds_median": round(ts[len(ts) // 2], 4), "seconds_best": round(ts[0], 4),
                  "GBps_median_input": round(n / ts[len(ts) // 2] / 1e9, 2), "GBps_best_input": round(n / ts[0] / 1e9, 2),
                  "out_bytes": int(osz)}))
