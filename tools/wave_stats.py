#!/usr/bin/env python3
"""Developer tool: where a data wave of k_encode spends its time (-DHVLC_EXPERIMENT build): shader clocks per phase and
per cause of waiting, summed per wave, printed as shares of the wave's lifetime.
    python tools/wave_stats.py [--mib 4096] [--workload H2.2]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HVLC_LIB", os.path.join(ROOT, "huffman-gpu_amd", "libhvlc_exp.so"))
NAMES = ["encode(+load wait)", "prefetch issue | STORE", "arrive", "zero+pack", "wait ticket | RESOLVE", "load wait (HVLC_WS_LOADWAIT)", "wait slot", "wait ring",
         "slow path", "wait tail", "total", "tiles", "nwaits"]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=4096); ap.add_argument("--workload", default="H2.2")
    a = ap.parse_args()
    import torch, hvlc
    from util import dist_weights, SEEDS
    L = hvlc.lib(); L.hvlc_debug_set_wave_stats.argtypes = [C.c_void_p]
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    n = a.mib << 20
    enc = hvlc.Encoder(0, n // 4)
    d_in = enc.gen_symbols(n, SEEDS[a.workload], hvlc.cdf32_from_counts(dist_weights(a.workload, period)))
    hist = enc.histogram(d_in).cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.huffman_table(hist); d_tab = enc.table_to_device(cw, cl)
    out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
    offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
    G = 768
    for it in range(3):
        ws = torch.zeros(4096 * 16 * 16, dtype=torch.int64, device="cuda")
        L.hvlc_debug_set_wave_stats(C.c_void_p(ws.data_ptr()))
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(), out.numel(), offs.data_ptr(), 0)
        e1.record()
        bits, rc = enc.finish(check=False)
    G = enc.launch_info()["grid_blocks"]
    allw = ws.cpu().numpy().reshape(-1, 16, 16)[:G].astype(np.float64)
    NW = 10
    sw = allw[:, NW + 1, :]                                                      # the store wave
    print(f"store wave: lifetime (Mclk) {sw[:, 10].mean()/1e6:.3f}  tiles {sw[:, 11].mean():.1f}  storing {100*(sw[:, 1]/sw[:, 10]).mean():.1f} %"
          f" = {sw[:, 1].sum()/sw[:, 11].sum():.0f} clocks per tile, waiting for a resolved tile {100*(sw[:, 4]/sw[:, 10]).mean():.1f} %")
    t = allw[:, :NW, :]                                                          # data waves only
    tot = t[:, :, 10]
    print(f"{a.workload} {a.mib} MiB rc {rc} kernel {e0.elapsed_time(e1):.3f} ms  grid {G}  wave lifetime (Mclk) mean {tot.mean()/1e6:.3f} min {tot.min()/1e6:.3f} max {tot.max()/1e6:.3f}"
          f"  tiles per wave {t[:, :, 11].mean():.1f}")
    for i in range(10):
        sh = t[:, :, i] / tot
        print(f"  {NAMES[i]:20s} {100*sh.mean():6.2f} %   (p10 {100*np.percentile(sh,10):5.2f}  p90 {100*np.percentile(sh,90):5.2f})   clocks per tile {t[:, :, i].sum() / t[:, :, 11].sum():8.0f}")
    print(f"  clocks per tile total {tot.sum() / t[:, :, 11].sum():8.0f}")
    bywave = [(t[:, w, 0] / tot[:, w]).mean() for w in range(NW)]
    print("  encode share by wave:", " ".join(f"{100*x:.1f}" for x in bywave))

if __name__ == "__main__":
    main()
