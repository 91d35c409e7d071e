#!/usr/bin/env python3
"""Developer tool (GPU box): time several builds of libhvlc.so against each other IN ONE PROCESS, round robin on the same
buffers (runs on different boxes, and different processes on one box, differ by more than most changes are worth).
    python tools/ab.py [--mib 4096] [--workloads H2.2,H4.0,H1.0] [--rounds 12] [--steps 8] name[=path] ...
`name` alone means huffman-gpu_amd/variants/libhvlc_<name>.so; "product" = huffman-gpu_amd/libhvlc.so.
Per variant and workload: median and minimum over the rounds of (HIP events around `steps` back-to-back encodes) / steps,
and the total bit count against sum(hist * len)."""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load(path):
    L = C.CDLL(path)
    vp, u64 = C.c_void_p, C.c_uint64
    L.hvlc_create.argtypes = [C.POINTER(vp), C.c_int, u64]
    L.hvlc_encode.argtypes = [vp, vp, u64, vp, vp, vp, u64, vp, vp, vp]
    L.hvlc_finish.argtypes = [vp, vp, C.POINTER(u64)]
    L.hvlc_destroy.argtypes = [vp]
    try:
        L.hvlc_get_clock_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    except AttributeError:
        pass
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=4096)
    ap.add_argument("--workloads", default="H2.2")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    import torch
    import hvlc
    from util import dist_weights, SEEDS
    libs = []
    for v in a.variants:
        name, _, path = v.partition("=")
        if not path:
            path = os.path.join(ROOT, "huffman-gpu_amd", "libhvlc.so" if name == "product" else f"variants/libhvlc_{name}.so")
        libs.append((name, load(path)))
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    n = a.mib << 20
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for wl in a.workloads.split(","):
        base = hvlc.Encoder(0, n // 4)
        d_in = base.gen_symbols(n, SEEDS[wl], hvlc.cdf32_from_counts(dist_weights(wl, period)))
        hist = base.histogram(d_in).cpu().numpy().astype(np.uint64)
        cw, cl = hvlc.huffman_table(hist)
        d_tab = base.table_to_device(cw, cl)
        out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
        offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
        expect = int((hist * cl).sum())
        ctxs = []
        for name, L in libs:
            ctx = C.c_void_p()
            assert L.hvlc_create(C.byref(ctx), 0, n // 4) == 0
            ctxs.append(ctx)
        def enc(L, ctx):
            rc = L.hvlc_encode(ctx, C.c_void_p(d_in.data_ptr()), n // 4, C.c_void_p(d_tab.data_ptr()), C.c_void_p(d_tab.data_ptr() + 1024),
                               C.c_void_p(out.data_ptr()), out.numel(), C.c_void_p(offs.data_ptr()), None, stream)
            assert rc == 0, rc
        times = {name: [] for name, _ in libs}
        clocks = {}
        ok = {}
        for (name, L), ctx in zip(libs, ctxs):            # warm-up, and the bit count
            for _ in range(3):
                enc(L, ctx)
            bits = C.c_uint64(0)
            rc = L.hvlc_finish(ctx, stream, C.byref(bits))
            ok[name] = (rc == 0 and bits.value == expect)
        for r in range(a.rounds):
            order = list(zip(libs, ctxs))
            if r & 1:
                order.reverse()
            for (name, L), ctx in order:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                enc(L, ctx)                                   # (one untimed call: the clocks are up, the L2 holds this variant's code)
                e0.record()
                for _ in range(a.steps):
                    enc(L, ctx)
                e1.record()
                e1.synchronize()
                times[name].append(e0.elapsed_time(e1) / a.steps)
                if hasattr(L, "hvlc_get_clock_info"):
                    bits = C.c_uint64(0); L.hvlc_finish(ctx, stream, C.byref(bits))
                    cc, rr = C.c_uint64(0), C.c_uint64(0); L.hvlc_get_clock_info(ctx, C.byref(cc), C.byref(rr))
                    if rr.value: clocks.setdefault(name, []).append(cc.value / rr.value / 10.0)
        ref = statistics.median(times[libs[0][0]])
        for name, _ in libs:
            t = times[name]
            med = statistics.median(t)
            ck = f"  clock {statistics.median(clocks[name]):.2f} GHz" if name in clocks else ""
            print(f"{wl:5s} {name:14s} median {med:.4f} ms  min {min(t):.4f}  GB/s {n / med / 1e6:7.1f}  vs {libs[0][0]} {ref / med:.3f}  bits_ok {ok[name]}{ck}", flush=True)
        for (name, L), ctx in zip(libs, ctxs):
            L.hvlc_destroy(ctx)
        del d_in, out, offs


if __name__ == "__main__":
    main()
