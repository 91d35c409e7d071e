#!/usr/bin/env python3
"""Developer tool: what the memory system gives at different write shares (hvlc_probe_bandwidth): copy (1:1), 1:2, the
encoder's 0.28:1, read only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd"))
import torch, hvlc
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
e = hvlc.Encoder(0, n // 4)
d_in = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
out = torch.empty(n // 4, dtype=torch.int32, device="cuda")
for wpt, shape in ((1024, 2), (1024, 1), (512, 2), (288, 2), (288, 4), (288, 1), (288, 3), (0, 0)):
    best, med = e.probe_bandwidth(d_in, out if shape else None, wpt, shape)
    tot = n * (1 + wpt / 1024.0)
    print(f"read + {wpt*4} B written per 4 KiB (store shape {shape}): total {tot/best/1e6:.0f} GB/s, read part {n/best/1e6:.0f} GB/s")
