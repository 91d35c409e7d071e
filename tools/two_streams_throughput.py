#!/usr/bin/env python3
"""Developer tool: aggregate throughput of TWO contexts encoding 2 GiB each on two streams at the same time (each grid
is sized for the whole device, so neither is co-resident while both run) against one context encoding 4 GiB."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import synth_on_device
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
n = 2 << 30
encs = [hvlc.Encoder(0, n // 4), hvlc.Encoder(0, n // 4)]
d_in, cw, cl, hist = synth_on_device(encs[0], "H2.2", 2 * n, period)
halves = [d_in[:n], d_in[n:]]
tabs = [e.table_to_device(cw, cl) for e in encs]
cap = hvlc.out_capacity_words(n // 4, int(cl.max()))
outs = [torch.empty(cap, dtype=torch.int32, device="cuda") for _ in range(2)]
tots = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def go(i):
    encs[i].encode_ptr(halves[i].data_ptr(), n // 4, tabs[i].data_ptr(), tabs[i].data_ptr() + 1024, outs[i].data_ptr(), cap,
                       0, tots[i].data_ptr(), stream=streams[i].cuda_stream)
K = 20
for label, who in (("one stream, 2 GiB per call", (0,)), ("two streams, 2 GiB per call each", (0, 1)), ("one stream, 2 GiB per call", (1,))):
    for _ in range(3):
        for i in who: go(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for i in who: go(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:34s} {len(who) * K * n / dt / 1e9:8.1f} GB/s aggregate", flush=True)
bits = [int(t.item()) for t in tots]
half_hist = [encs[0].histogram(h).cpu().numpy().astype(np.uint64) for h in halves]
print("bit totals ok:", all(bits[i] == int((half_hist[i] * cl.astype(np.uint64)).sum()) for i in range(2)))
