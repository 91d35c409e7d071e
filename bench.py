#!/usr/bin/env python3
"""bench.py -- headline benchmark of the encode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--size-mib M]

N=1 : one process.  N>1: launched by `python -m torch.distributed.run --nproc-per-node N`,
one rank per GPU, backend nccl (= RCCL over xGMI).

Workload (config.workload): the north-star / BASELINE config 5 shard -- a 4 GiB synthetic
H~2.2 byte stream PER GPU (weak scaling; at N ranks the job is the 4N GiB stream of
SURVEY.md section 8(d), shard g drawing from index g*2^31, seed 0x5EED0005), encoded with
the Huffman table of the job's global histogram.  A "step" is one whole pass of the hot
path over the rank's shard: hvlc_encode (k_prepare + the fused k_encode) and, for N>1,
the stitch: ONE RCCL all-gather of the per-shard bit totals + the device-side rebase of the
segment offsets.  Inputs are resident in HBM before the timed region; nothing is skipped.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel k_encode, timed live
with HIP events on its launch stream (hvlc_profile_begin/_collect); `cpu_baseline` times
the compiled reference encoder (oracle/_ref, kind "reference") or, if absent, the repo's C
restatement (kind "port") on a bounded sample of the same stream, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "huffman-gpu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
WORKLOADS = {                    # name -> (distribution, default MiB per GPU, seed)
    "H2.2-4GiB":   ("H2.2", 4096, 0x5EED0005),
    "H2.2-256MiB": ("H2.2", 256, 0x5EED0002),
    "H4.0-4GiB":   ("H4.0", 4096, 0x5EED0003),
    "H1.0-4GiB":   ("H1.0", 4096, 0x5EED0004),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="H2.2-4GiB", choices=sorted(WORKLOADS))
    ap.add_argument("--size-mib", type=int, default=0, help="override the per-GPU shard size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hvlc
    from util import dist_weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    dname, mib, seed = WORKLOADS[args.workload]
    if args.size_mib:
        mib = args.size_mib
    n_bytes = mib << 20
    n_words = n_bytes // 4
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)

    enc = hvlc.Encoder(local_rank, n_words)
    # ---- synthetic shard, resident in HBM -----------------------------------------------------
    cdf = hvlc.cdf32_from_counts(dist_weights(dname, period))
    d_in = enc.gen_symbols(n_bytes, seed, cdf, first_draw=rank * (n_bytes // 2))
    hist = enc.histogram(d_in)
    if world > 1:
        dist.all_reduce(hist)                              # table of the WHOLE job's histogram
    hist_np = hist.cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.build_table(hist_np)                     # libhvlc's host-side table builder
    max_len = int(cl.max())
    d_table = enc.table_to_device(cw, cl)
    S = hvlc.num_segments(n_words)
    out = torch.empty(hvlc.out_capacity_words(n_words, max_len), dtype=torch.int32, device=dev)
    offsets = torch.empty(S, dtype=torch.int64, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    gathered = torch.zeros(world, dtype=torch.int64, device=dev)

    def step():
        enc.encode_ptr(d_in.data_ptr(), n_words, d_table.data_ptr(), d_table.data_ptr() + 1024,
                       out.data_ptr(), out.numel(), offsets.data_ptr(), total.data_ptr())
        if world > 1:                                      # the stitch: one 8-byte-per-rank exchange over xGMI
            dist.all_gather_into_tensor(gathered, total)
            enc.rebase_offsets_dev(offsets, gathered, rank)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    bits, rc = enc.finish()
    assert rc == 0, hvlc.status_string(rc)
    local_hist = enc.histogram(d_in).cpu().numpy().astype(np.uint64)
    assert bits == int((local_hist * cl).sum()), "bit total does not match histogram x lengths"

    enc.profile_begin(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = enc.profile_collect(args.steps)
    bits2, rc = enc.finish()
    assert rc == 0 and bits2 == bits

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_bytes / (dt / args.steps) / 1e9

    # ---- roofline of the dominant kernel (rank-local) -----------------------------------------
    # algorithmic bytes per launch: read input once + write the stream once
    # + one uint64 offset per 1024-byte segment
    alg_bytes = n_bytes + (bits + 7) // 8 + 8 * S
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes per launch, if profiled
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            traffic = tj.get(args.workload if not args.size_mib else "", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "k_encode", "kernel_ms": round(k_ms, 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "algorithmic_bytes": int(alg_bytes),
                "input_only_frac": round(n_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline on the host cores, same stream, bounded sample ---------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        smp = min(args.cpu_sample_mib << 20, n_bytes)
        x = d_in[:smp].cpu().numpy().view(np.uint32)
        if O.have_ref():
            kind = "reference"
            t1 = time.perf_counter()
            _, osz = O.ref_encode32(x, cw, cl)
            sec = time.perf_counter() - t1
        else:
            kind = "port"
            t1 = time.perf_counter()
            _, osz = O.encode32(x, cw, cl)
            sec = time.perf_counter() - t1
        ncores = os.cpu_count() or 1
        _, sec_mt = O.encode_mt(x, cw, cl, ncores)
        cpu = {"value": round(smp / sec / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
               "sample": f"first {smp >> 20} MiB of the same stream, cpu_vlc_encode, 1 thread, {sec:.2f} s",
               "all_cores": {"value": round(smp / sec_mt / 1e9, 3), "cores": ncores, "kind": "port",
                             "note": "independent contiguous chunks, one per host thread"}}

    if rank == 0:
        info = enc.launch_info()
        line = {
            "metric": "encode GB/s (input bytes)", "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{mib} MiB/GPU synthetic {dname} byte stream (splitmix64 seed {seed:#x}), "
                                   f"Huffman table of the job histogram (max len {max_len}), bit-exact vs cpu_vlc_encode",
                       "bytes_per_gpu": n_bytes, "segments_per_gpu": S, "out_bits_per_gpu": bits,
                       "bits_per_symbol": round(bits / n_bytes, 4),
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "grid_blocks": info["grid_blocks"], "blocks_per_cu": info["blocks_per_cu"]},
            "hbm_frac_of_peak_input_bytes": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
