#!/usr/bin/env python3
"""Developer tool: the life of a SMALL call (-DHVLC_EXPERIMENT build): 100 MHz stamps of tile 0 and of the last tile
relative to workgroup 0's entry, and the time per call between HIP events.
    python tools/one_tile_timeline.py [KiB ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HVLC_LIB"] = os.path.join(ROOT, "huffman-gpu_amd", "libhvlc_exp.so")
import torch, hvlc
from util import synth_on_device
L = hvlc.lib(); L.hvlc_debug_set_timeline.argtypes = [C.c_void_p]
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
MAXB = 512 << 20
e = hvlc.Encoder(0, MAXB // 4)
d_all, cw, cl, hist = synth_on_device(e, "H2.2", MAXB, period)
d_tab = e.table_to_device(cw, cl)
out = torch.empty(hvlc.out_capacity_words(MAXB // 4, int(cl.max())), dtype=torch.int32, device="cuda")
offs = torch.empty(hvlc.num_segments(MAXB // 4), dtype=torch.int64, device="cuda")
NAMES = ["bits published", "scout polls record", "leader: group bits complete", "leader: records published", "record seen valid",
         "store begins", "stored", "wg0 enters"]
for kib in [int(x) for x in sys.argv[1:]] or [1, 40, 80, 2560, 20480, 262144]:
    n = kib << 10
    ntiles = (n + 40959) // 40960
    call = lambda: e.encode_ptr(d_all.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
    tl = torch.zeros((ntiles + 1) * 8, dtype=torch.int64, device="cuda")
    L.hvlc_debug_set_timeline(C.c_void_p(tl.data_ptr()))
    for _ in range(5): call()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); e1.synchronize()
    tx = tl.cpu().numpy().reshape(ntiles + 1, 8).astype(np.float64); t, x = tx[:ntiles], tx[ntiles]
    t0 = t[0, 7]
    print(f"{kib} KiB: {ntiles} tiles, {e0.elapsed_time(e1) / 50 * 1000:.1f} us per call (events, 50 calls back to back)")
    for name, row in (("tile 0", t[0]), ("last tile", t[-1])) if ntiles > 1 else (("tile 0", t[0]),):
        print("   " + name + ": " + "  ".join(f"{NAMES[k]} {(row[k] - t0) / 100:.2f}" for k in (0, 1, 2, 3, 4, 5, 6) if row[k] > 0))
    XN = ["table loaded", "barrier 1", "LDS set up", "wave 0: first tile number, loads issued", "first words here", "first part encoded"]
    print("   workgroup 0: " + "  ".join(f"{XN[k]} {(x[k] - t0) / 100:.2f}" for k in range(6) if x[k] > 0))
    if ntiles > 1:
        print(f"   latest stamp of any tile: {(t[:, :7].max() - t0) / 100:.2f} us after wg0's entry; first 'bits published' of any tile {(t[:, 0][t[:, 0] > 0].min() - t0) / 100:.2f}")
    L.hvlc_debug_set_timeline(None)
