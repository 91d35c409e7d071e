#!/usr/bin/env python3
"""Developer tool (not part of the product or the judged bench): time hvlc_encode on the named
synthetic workloads for several blocks-per-CU settings in one process, HIP-event timed.
    python tools/perf_sweep.py [--mib 1024] [--bpc 1,2,3] [--workloads H2.2,H4.0,H1.0] [--iters 10]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--bpc", default="0")
    ap.add_argument("--workloads", default="H2.2,H4.0,H1.0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch
    import hvlc
    from util import dist_weights, SEEDS
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)
    n = args.mib << 20
    for name in args.workloads.split(","):
        base = hvlc.Encoder(0, n // 4)
        cdf = hvlc.cdf32_from_counts(dist_weights(name, period))
        d_in = base.gen_symbols(n, SEEDS[name], cdf)
        hist = base.histogram(d_in).cpu().numpy().astype(np.uint64)
        cw, cl = hvlc.huffman_table(hist)
        d_tab = base.table_to_device(cw, cl)
        out = torch.empty(hvlc.out_capacity_words(n // 4, int(cl.max())), dtype=torch.int32, device="cuda")
        offs = torch.empty(hvlc.num_segments(n // 4), dtype=torch.int64, device="cuda")
        expect = int((hist * cl).sum())
        for bpc in [int(b) for b in args.bpc.split(",")]:
            if bpc:
                os.environ["HVLC_BLOCKS_PER_CU"] = str(bpc)
            else:
                os.environ.pop("HVLC_BLOCKS_PER_CU", None)
            enc = hvlc.Encoder(0, n // 4)
            for _ in range(2):
                enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(),
                               out.numel(), offs.data_ptr(), 0)
            bits, rc = enc.finish(check=False)
            enc.profile_begin(args.iters)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                enc.encode_ptr(d_in.data_ptr(), n // 4, d_tab.data_ptr(), d_tab.data_ptr() + 1024, out.data_ptr(),
                               out.numel(), offs.data_ptr(), 0)
            e1.record()
            torch.cuda.synchronize()
            kms = enc.profile_collect(args.iters)
            bits2, rc2 = enc.finish(check=False)
            ms = e0.elapsed_time(e1) / args.iters
            alg = n + bits // 8 + 8 * offs.numel()
            print(json.dumps({"workload": name, "mib": args.mib, "info": enc.launch_info(), "rc": rc2,
                              "bits_ok": bits2 == expect, "ms_call": round(ms, 4),
                              "ms_kernel": round(float(np.mean(kms)), 4), "ms_kernel_min": round(float(np.min(kms)), 4),
                              "in_GBs": round(n / np.mean(kms) / 1e6, 1), "alg_GBs": round(alg / np.mean(kms) / 1e6, 1),
                              "bits_per_sym": round(bits2 / n, 3)}), flush=True)
            enc.close()
        base.close()


if __name__ == "__main__":
    main()
