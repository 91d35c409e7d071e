#!/usr/bin/env python3
"""Developer tool: throughput of hvlc_histogram on the synthetic workloads."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import dist_weights, SEEDS
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
n = 4 << 30
e = hvlc.Encoder(0, n // 4)
for name in ("H2.2", "H4.0", "H1.0"):
    d = e.gen_symbols(n, SEEDS[name], hvlc.cdf32_from_counts(dist_weights(name, period)))
    e.histogram(d); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): h = e.histogram(d)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"workload": name, "ms": round(ms, 3), "GBs": round(n / ms / 1e6, 1), "sum_ok": int(h.sum().item()) == n}))
    del d
