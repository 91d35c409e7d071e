#!/usr/bin/env python3
"""Developer tool (GPU box): where the encoder's two workgroup geometries cross over.  Streams of 2^-k distributions with
2.5 ... 4.5 bits per symbol, each encoded by a context fixed to ten data waves and by one fixed to eight (HVLC_GEOMETRY),
round robin in one process.    python tools/geometry_crossover.py [MiB]"""
import os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = mib << 20
os.environ["HVLC_GEOMETRY"] = "10"; e10 = hvlc.Encoder(0, n // 4)
os.environ["HVLC_GEOMETRY"] = "8";  e8 = hvlc.Encoder(0, n // 4)
del os.environ["HVLC_GEOMETRY"]

def weights(lens):                      # symbol i with probability 2^-lens[i] (a complete code: the Huffman lengths are these)
    w = np.zeros(256); w[: len(lens)] = [2.0 ** -l for l in lens]; assert abs(w.sum() - 1.0) < 1e-9; return w

CASES = {                               # expected bits per symbol = sum 2^-l * l
    "2.50": [1, 3, 3, 4, 4, 4, 4],
    "3.00": [3] * 8,
    "3.25": [3] * 6 + [4] * 4,
    "3.50": [3] * 4 + [4] * 8,
    "3.75": [3] * 2 + [4] * 12,
    "4.00": [4] * 16,
    "4.50": [4] * 8 + [5] * 16,
}
for name, lens in CASES.items():
    w = weights(lens)
    bps = float((w[w > 0] * -np.log2(w[w > 0])).sum())
    d_in = e10.gen_symbols(n, 0x5EED1000 + len(lens), hvlc.cdf32_from_counts(w))
    hist = e10.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    res = {}
    tabs = {10: e10.table_to_device(cw, cl), 8: e8.table_to_device(cw, cl)}
    encs = {10: e10, 8: e8}
    times = {10: [], 8: []}
    for r in range(8):
        for g in ((10, 8) if r & 1 else (8, 10)):
            e = encs[g]
            call = lambda: e.encode_ptr(d_in.data_ptr(), n // 4, tabs[g].data_ptr(), tabs[g].data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
            call()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(6): call()
            b.record(); b.synchronize()
            times[g].append(a.elapsed_time(b) / 6)
            bits, rc = e.finish(check=False)
            assert rc == 0 and e.launch_info()["data_waves"] == g
    t10, t8 = statistics.median(times[10]), statistics.median(times[8])
    print(f"{bps:.2f} bits per symbol: ten data waves {n / t10 / 1e6:7.1f} GB/s, eight {n / t8 / 1e6:7.1f} GB/s, eight / ten = {t10 / t8:.3f}", flush=True)
    del d_in, out, offs
