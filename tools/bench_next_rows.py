#!/usr/bin/env python3
"""Developer tool: throughput of the SURVEY 8(f) rows built so far (seam merge, decoder) on the bench workload."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
from util import dist_weights, SEEDS
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
n = mib << 20
e = hvlc.Encoder(0, n // 4)
d_in = e.gen_symbols(n, SEEDS["H2.2"], hvlc.cdf32_from_counts(dist_weights("H2.2", period)))
hist = e.histogram(d_in).cpu().numpy().astype(np.uint64)
cw, cl = hvlc.huffman_table(hist); d_tab = e.table_to_device(cw, cl)
o, offs, bits = e.encode(d_in, d_tab, max_len=int(cl.max()))
dst = torch.zeros(bits // 32 + 1024, dtype=torch.int32, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms_merge = timed(lambda: e.merge_shard(o, bits, dst, 12345))          # odd bit offset: every word is funnel-shifted
ms_dec = timed(lambda: e.decode(o, bits, offs, d_tab, n // 4), reps=3)
stream_bytes = bits / 8
# the reference pipeline's stage entry points kept as separate calls (include/hvlc.h): per-segment bit counts (A1's outidx)
# and the stand-alone exclusive scan (A3) of 64 Mi counts
ms_seg = timed(lambda: e.segment_bits(d_in, d_tab))
cnt = torch.randint(0, 30000, (64 << 20,), dtype=torch.int32, device="cuda")
ms_scan = timed(lambda: e.exclusive_scan(cnt))
print(json.dumps({"workload": f"H2.2 {mib} MiB", "stream_bytes": int(stream_bytes),
                  "merge_ms": round(ms_merge, 3), "merge_GBs_stream_read_plus_write": round(2 * stream_bytes / ms_merge / 1e6, 1),
                  "decode_ms": round(ms_dec, 3), "decode_GBs_output_bytes": round(n / ms_dec / 1e6, 1),
                  "segment_bits_ms": round(ms_seg, 3), "segment_bits_GBs_input": round(n / ms_seg / 1e6, 1),
                  "exclusive_scan_64Mi_ms": round(ms_scan, 3), "exclusive_scan_GBs_read_plus_written": round((64 << 20) * 12 / ms_scan / 1e6, 1)}))
