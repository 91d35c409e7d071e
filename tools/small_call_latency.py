#!/usr/bin/env python3
"""Developer tool: time per hvlc_encode call, back to back on one stream, for small inputs (one launch per call)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import synth_on_device
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
e = hvlc.Encoder(0, (64 << 20) // 4)
d_all, cw, cl, hist = synth_on_device(e, "H2.2", 64 << 20, period)
d_tab = e.table_to_device(cw, cl)
out = torch.empty(hvlc.out_capacity_words((64 << 20) // 4, int(cl.max())), dtype=torch.int32, device="cuda")
offs = torch.empty(hvlc.num_segments((64 << 20) // 4), dtype=torch.int64, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
for kib in (1, 28, 256, 1024, 4096, 16384, 65536):
    n = kib << 10
    call = lambda: e.encode_ptr(d_all.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), tot.data_ptr())
    for _ in range(20): call()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{kib:6d} KiB  {dt * 1e6:8.1f} us per call  {n / dt / 1e9:8.1f} GB/s", flush=True)
