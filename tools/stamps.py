#!/usr/bin/env python3
"""Developer tool: per-phase cycle breakdown of k_encode from the -DHVLC_EXPERIMENT build."""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HVLC_LIB"] = os.path.join(ROOT, "huffman-gpu_amd", "libhvlc_exp.so")

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024); ap.add_argument("--bpc", default="1,3")
    ap.add_argument("--workload", default="H2.2")
    a = ap.parse_args()
    import torch, hvlc
    from util import dist_weights, SEEDS
    L = hvlc.lib(); L.hvlc_debug_set_buffer.argtypes = [C.c_void_p]
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    n = a.mib << 20
    base = hvlc.Encoder(0, n // 4)
    d_in = base.gen_symbols(n, SEEDS[a.workload], hvlc.cdf32_from_counts(dist_weights(a.workload, period)))
    hist = base.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist); d_tab = base.table_to_device(cw, cl)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    names_d = ["top", "finish(store)", "encode(+wait raw)", "prefetch issue", "barrier1", "pack", "zero", "barrier2"]
    names_s = ["start", "steps", "handover+idle", "-", "-", "-", "-"]
    for bpc in [int(b) for b in a.bpc.split(",")]:
        os.environ["HVLC_BLOCKS_PER_CU"] = str(bpc)
        enc = hvlc.Encoder(0, n // 4)
        buf = torch.zeros(4096 * 2 * 16, dtype=torch.int64, device="cuda")
        L.hvlc_debug_set_buffer(C.c_void_p(buf.data_ptr()))
        for _ in range(3):
            enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
        bits, rc = enc.finish(check=False)
        info = enc.launch_info(); g = info["grid_blocks"]
        b = buf.cpu().numpy().reshape(-1, 2, 16)[:g].astype(np.float64)
        d, s = b[:, 0, :], b[:, 1, :]
        iters = d[:, 8].mean()
        print(f"== bpc {bpc} grid {g} rc {rc} iters/WG {iters:.1f} pack {d[:,9].mean():.1f} store {d[:,10].mean():.1f} idle(must_block) {d[:,11].mean():.1f}")
        tot = d[:, :8].sum(axis=1).mean()
        print("  data wave0 cycles/iter: " + ", ".join(f"{nm}={d[:,i].mean()/iters:.0f}" for i, nm in enumerate(names_d)) + f"  total={tot/iters:.0f}")
        tots = s[:, :7].sum(axis=1).mean()
        print(f"  scout rounds/WG {s[:,9].mean():.1f} handovers/WG {s[:,8].mean():.1f} steps/WG {s[:,10].mean():.1f} "
              f"incomplete: tile_bits {s[:,11].mean():.1f} group {s[:,12].mean():.1f} tail {s[:,13].mean():.1f} leader-group-only {s[:,14].mean():.1f}")
        print("  scout      cycles/iter: " + ", ".join(f"{nm}={s[:,i].mean()/iters:.0f}" for i, nm in enumerate(names_s)) + f"  total={tots/iters:.0f}")
        idle = d[:, 11] / d[:, 8]
        busy = (d[:, 1] + d[:, 2] + d[:, 3] + d[:, 5] + d[:, 6]) / d[:, 9]      # cycles of real work per packed tile
        print(f"  idle fraction per WG: min {idle.min():.2f} p10 {np.percentile(idle,10):.2f} p50 {np.percentile(idle,50):.2f} p90 {np.percentile(idle,90):.2f} max {idle.max():.2f}")
        print(f"  work cycles per tile per WG: min {busy.min():.0f} p50 {np.percentile(busy,50):.0f} p90 {np.percentile(busy,90):.0f} max {busy.max():.0f}")
        by_xcd = [idle[x::8].mean() for x in range(8)]
        print("  idle fraction by blockIdx%8:", np.round(by_xcd, 2))
        enc.close()
main()
