#!/usr/bin/env python3
"""Developer tool: encode with a forced (over-subscribed) grid and report where the stream differs from the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hvlc
import oracle_lib as O
factor = int(sys.argv[1]) if len(sys.argv) > 1 else 2
period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
tab = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_table.bin"), dtype=np.uint32)
cw, cl = tab[:256].copy(), tab[256:].copy()
e = hvlc.Encoder(0, 1 << 25)
info = e.launch_info()
if factor: e.test_hook(hvlc.HOOK_FORCE_GRID, factor * info["compute_units"] * info["blocks_per_cu"])
x = np.tile(period, 512).view(np.uint32)
d_tab = e.table_to_device(cw, cl)
ref, rbits = O.encode(x, cw, cl)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    o, offs, bits = e.encode(torch.from_numpy(x.view(np.int32)).cuda(), d_tab, max_len=21)
    got = o[: bits // 32 + 1].cpu().numpy().view(np.uint32)
    nbad = int((got != ref[: got.size]).sum()) if bits == rbits else -1
    msg = f"iter {it}: bits {bits} ref {rbits} bad words {nbad}"
    if nbad > 0:
        idx = np.nonzero(got != ref[: got.size])[0]
        offs_np = offs.cpu().numpy().view(np.uint64)
        seg = np.searchsorted(offs_np, idx[0] * 32, side="right") - 1
        msg += f" first bad word {idx[0]} (segment {seg}, tile {seg // 28}) last bad {idx[-1]} runs {np.count_nonzero(np.diff(idx) > 1) + 1}"
        msg += f" got {got[idx[0]]:08x} ref {ref[idx[0]]:08x}"
    print(msg, e.launch_info())
