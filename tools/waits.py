#!/usr/bin/env python3
"""Developer tool: how often the data waves had to wait for a staging slot (counters-only experiment build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HVLC_LIB"] = os.path.join(ROOT, "huffman-gpu_amd", "libhvlc_exp.so")
import torch, hvlc
from util import dist_weights, SEEDS
L = hvlc.lib(); L.hvlc_debug_set_buffer.argtypes = [C.c_void_p]
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
n = 1024 << 20
for name in sys.argv[1:] or ["H2.2"]:
    enc = hvlc.Encoder(0, n // 4)
    d_in = enc.gen_symbols(n, SEEDS[name], hvlc.cdf32_from_counts(dist_weights(name, period)))
    hist = enc.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist); d_tab = enc.table_to_device(cw, cl)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    buf = torch.zeros(4096 * 2 * 16, dtype=torch.int64, device="cuda")
    L.hvlc_debug_set_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(3):
        enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
    bits, rc = enc.finish(check=False)
    g = enc.launch_info()["grid_blocks"]
    d = buf.cpu().numpy().reshape(-1, 2, 16)[:g, 0, :].astype(np.float64)
    print(name, "rc", rc, "grid", g, "tiles/WG", d[:, 8].mean(), "slot-wait polls/WG", d[:, 11].mean(),
          "p10/p50/p90", np.percentile(d[:, 11], [10, 50, 90]))
    enc.close()
