#!/usr/bin/env python3
"""Developer tool: per-tile timeline (pack -> request -> resolved -> store) from the -DHVLC_EXPERIMENT build."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["HVLC_LIB"] = os.path.join(ROOT, "huffman-gpu_amd", "libhvlc_exp.so")

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=4096); ap.add_argument("--bpc", type=int, default=0)
    ap.add_argument("--workload", default="H2.2")
    a = ap.parse_args()
    import torch, hvlc
    from util import dist_weights, SEEDS
    L = hvlc.lib(); L.hvlc_debug_set_timeline.argtypes = [C.c_void_p]
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    n = a.mib << 20
    if a.bpc: os.environ["HVLC_BLOCKS_PER_CU"] = str(a.bpc)
    enc = hvlc.Encoder(0, n // 4)
    d_in = enc.gen_symbols(n, SEEDS[a.workload], hvlc.cdf32_from_counts(dist_weights(a.workload, period)))
    hist = enc.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist); d_tab = enc.table_to_device(cw, cl)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    TILE = 40960                                     # bytes per tile (kTileWords * 4)
    ntiles = n // TILE + (1 if n % TILE else 0)
    for it in range(3):
        tl = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
        L.hvlc_debug_set_timeline(C.c_void_p(tl.data_ptr()))
        enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
        bits, rc = enc.finish(check=False)
    t = tl.cpu().numpy().reshape(ntiles, 8).astype(np.float64)
    G = enc.launch_info()["grid_blocks"]
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    # 0 bits published (packed), 1 scout starts polling the record, 2 (leader tile) group bits complete,
    # 3 (leader tile) group prefix + records published, 4 record seen valid, 5 store begins
    pack, start, lbits, lpub, res, store = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3]), us(t[:, 4]), us(t[:, 5])
    print(f"grid {G} tiles {ntiles} rc {rc} span {us(t[:,5].max()):.1f} us")
    def st(name, x):
        print(f"  {name:44s} mean {x.mean():7.2f}  p50 {np.percentile(x,50):7.2f}  p90 {np.percentile(x,90):7.2f}  max {x.max():7.2f}")
    mid = slice(ntiles // 4, 3 * ntiles // 4)
    st("pack -> scout starts polling (us)", (start - pack)[mid])
    st("pack -> resolved", (res - pack)[mid])
    st("resolved -> store begins", (store - res)[mid])
    st("store begins -> stored", (us(t[:, 6]) - store)[mid])
    st("pack -> store", (store - pack)[mid])
    ng = ntiles // 64
    gsl = slice(ng // 4, 3 * ng // 4)
    pk = pack[: ng * 64].reshape(ng, 64)
    lead = np.arange(ng) * 64 + 63
    st("group: last pack - first pack", (pk.max(axis=1) - pk.min(axis=1))[gsl])
    st("group: leader pack - last pack of group", (pack[lead] - pk.max(axis=1))[gsl])
    st("group: bits complete - last pack", (lbits[lead] - pk.max(axis=1))[gsl])
    st("group: records published - bits complete", (lpub[lead] - lbits[lead])[gsl])
    st("group: records published - last pack", (lpub[lead] - pk.max(axis=1))[gsl])
    rs = res[: ng * 64].reshape(ng, 64)
    st("tile: record seen - records published", (rs - lpub[lead][:, None])[gsl].ravel())
    st("group: publish time - previous group's", np.diff(lpub[lead])[gsl])
    gen = pack[: ntiles // G * G].reshape(-1, G)
    print("  generation pack time (mean over WGs) first 12:", np.round(gen.mean(axis=1)[:12], 1))
    print("  generation pack spread (max-min)   first 12:", np.round((gen.max(axis=1) - gen.min(axis=1))[:12], 1))
    print("  generation pack spread (p95-p5)    first 12:", np.round((np.percentile(gen, 95, axis=1) - np.percentile(gen, 5, axis=1))[:12], 1))
    # how far out of tile order are tiles packed?  lateness = pack time minus the running max of earlier tiles' pack times
    runmax = np.maximum.accumulate(pack)
    st("tile: (latest earlier pack) - own pack", (np.concatenate([[0], runmax[:-1]]) - pack)[mid])
    # per-class progress (tile % 8 classes interleave): pack time of every 8th tile
    for cidx in range(0, 8, 7):
        x = pack[cidx::8]
        print(f"  class-slot {cidx}: pack time at tile 25/50/75 %:", np.round([x[len(x)//4], x[len(x)//2], x[3*len(x)//4]], 1))
main()
