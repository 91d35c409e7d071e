#!/usr/bin/env python3
"""bench.py -- headline benchmark of the encode path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--size-mib M]

N=1 : one process.  N>1: launched by `python -m torch.distributed.run --nproc-per-node N`,
one rank per GPU, backend nccl (= RCCL over xGMI).

Workload (config.workload): the north-star / BASELINE config 5 shard -- a 4 GiB synthetic
H~2.2 byte stream PER GPU (weak scaling; at N ranks the job is the 4N GiB stream of
SURVEY.md section 8(d), shard g drawing from index g*2^31, seed 0x5EED0005), encoded with
the Huffman table of the job's global histogram.  A "step" is one whole pass of the hot
path over the rank's shard: hvlc_encode (ONE launch of the fused k_encode) and, for N>1,
the stitch (hvlc.stitch): ONE RCCL all-gather of the per-shard bit totals + the device-side
rebase of the segment offsets.  Inputs are resident in HBM before the timed region; nothing
is skipped.

Prints ONE JSON line (rank 0):
  value / ms_per_step   K steps timed as a whole (barrier + synchronize on both sides, max over ranks), nothing
                        on the stream between the steps; step_ms_median / step_ms_best from an instrumented pass
                        of the same K steps right behind the timed region (per-step events; not part of `value`)
  roofline              dominant kernel k_encode: HIP events on its launch stream around the timed region / K
                        (-> achieved; N > 1: per-launch events of the instrumented pass); median and best from the
                        instrumented pass (hvlc_profile_begin/_collect);
                        measured_read_ceiling / measured_mix_ceiling: in the SAME process, a kernel that only
                        streams the same input (and, for the mix, writes this workload's output volume):
                        what the memory system gives this traffic mix with no encoding work at all
  bit_exact             after the timed region the stream of the LAST step is compared word for word with the CPU
                        encoder's (main_test_cu.cu:170-171): at N = 1 a prefix (1 GiB) against the reference's
                        cpu_vlc_encode (oracle/_ref; else the repo's C restatement), at N > 1 every rank's WHOLE shard
                        and every segment offset against the C restatement on all host threads
  cpu_baseline          the same CPU run, timed (rank 0, N=1 only): kind "reference" / "port", 1 thread, plus
                        the port on all host threads
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


def host_cpu():
    """(model name, physical cores, hardware threads) of the host, from /proc/cpuinfo."""
    model, cores, phys, core = "unknown", set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    threads = os.cpu_count() or 1
    return model, (len(cores) or threads), threads


def compare_prefix(gpu_words, ref_words, nbits):
    """Word-for-word comparison of the first nbits bits of two MSB-first uint32 streams."""
    full = nbits // 32
    if not np.array_equal(gpu_words[:full], ref_words[:full]):
        return False
    rem = nbits % 32
    if rem:
        mask = np.uint32((0xFFFFFFFF << (32 - rem)) & 0xFFFFFFFF)
        return bool((gpu_words[full] & mask) == (ref_words[full] & mask))
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="H2.2-4GiB", choices=sorted(WORKLOADS))
    ap.add_argument("--size-mib", type=int, default=0, help="override the per-GPU shard size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=1024)
    ap.add_argument("--no-probe", action="store_true", help="skip the measured read / mix ceilings")
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
    # REHEARSAL of the N > 1 code on a one-GPU box (developer use, never the driver's run): HVLC_BENCH_REHEARSAL=1 puts
    # every rank on cuda:0 and carries the collectives over gloo (staged through host memory); everything else is the
    # code the 8-GPU run executes.  The JSON line says so ("rehearsal": true) and such a number is not a measurement.
    rehearsal = os.environ.get("HVLC_BENCH_REHEARSAL", "") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if rehearsal:
            h = t.cpu(); dist.all_reduce(h, op=op); t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def all_gather_into_tensor(out_t, in_t):
        if rehearsal:
            ho = out_t.cpu(); dist.all_gather_into_tensor(ho, in_t.cpu()); out_t.copy_(ho)
        else:
            dist.all_gather_into_tensor(out_t, in_t)

    dname, mib, seed = WORKLOADS[args.workload]
    if args.size_mib:
        mib = args.size_mib
    n_bytes = mib << 20
    n_words = n_bytes // 4
    period = np.fromfile(os.path.join(ROOT, "tests", "golden", "g1_period.bin"), dtype=np.uint8)

    enc = hvlc.Encoder(local_rank, n_words)
    # ---- synthetic shard, resident in HBM -----------------------------------------------------
    cdf = hvlc.cdf32_from_counts(dist_weights(dname, period))
    d_in = enc.gen_symbols(n_bytes, seed, cdf, first_draw=hvlc.shard_first_draw(rank, n_bytes))
    hist = enc.histogram(d_in)
    local_hist = hist.cpu().numpy().astype(np.uint64)
    if world > 1:
        all_reduce(hist)                                   # table of the WHOLE job's histogram
    hist_np = hist.cpu().numpy().astype(np.uint64)
    cw, cl = hvlc.build_table(hist_np)                     # libhvlc's host-side table builder
    max_len = int(cl.max())
    d_table = enc.table_to_device(cw, cl)
    S = hvlc.num_segments(n_words)
    out = torch.empty(hvlc.out_capacity_words(n_words, max_len), dtype=torch.int32, device=dev)
    offsets = torch.empty(S, dtype=torch.int64, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    gathered = torch.zeros(world, dtype=torch.int64, device=dev)

    def rebase(g, r):
        enc.rebase_offsets_dev(offsets, g, r)

    def step():
        enc.encode_ptr(d_in.data_ptr(), n_words, d_table.data_ptr(), d_table.data_ptr() + 1024,
                       out.data_ptr(), out.numel(), offsets.data_ptr(), total.data_ptr())
        # the stitch: one 8-byte-per-rank exchange over xGMI + the rebase of this rank's offsets
        hvlc.stitch(total, gathered, rank, world, all_gather_into_tensor, rebase)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step()                                   # (set-up, not a warm-up step: its bit total sizes the probes and is checked here)
    fence()
    bits, rc = enc.finish()
    assert rc == 0, hvlc.status_string(rc)
    assert bits == int((local_hist * cl).sum()), "bit total does not match histogram x lengths"

    # ---- what the memory system gives this traffic mix with no encoding work ("in the same run").  Here,
    # in FRONT of the warm-up steps: the GPU then goes through the warm-up into the timed region busy instead of idle
    # (after the set-up's host work its clocks take ~30 ms of work to come back: the first timed steps of an idle start ran
    # 20 % slower than the last, profiles/r02_kernel_trace.csv), and nothing but the W warm-up steps and one fence lies
    # between the probes and the timed steps (round 3's first arrangement freed the probes' buffer there: a pause of
    # milliseconds, and a mean step 1.3-2.5 % above the median)
    probe = None
    pbuf = None
    if not args.no_probe:
        wpt = max(4, (int(round(1024.0 * bits / n_bytes / 32.0 * 4.0)) + 3) // 4 * 4)   # stream words per 4 KiB of input
        rd_best, _ = enc.probe_bandwidth(d_in, None, 0, 0)
        # every store shape the probe knows (dword / dwordx4 per step, plain / non-temporal; non-temporal bursts of 10 and 32
        # steps; readers and a writer wave per workgroup): the ceiling is the BEST of them -- round 2 took the per-step shapes only and the encoder beat its "ceiling"
        shapes = {1: "dword", 2: "dwordx4", 3: "dword nt", 4: "dwordx4 nt", 5: "dwordx4 nt, bursts of 10 steps", 6: "dwordx4 nt, bursts of 32 steps",
                  8: "dwordx4 nt, 15 reader waves (one step in flight) + a writer wave per CU",
                  9: "dwordx4 nt, 15 reader waves (two steps in flight) + a writer wave per CU",
                  10: "dwordx4 nt, 15 reader waves (three steps in flight) + a writer wave per CU",
                  11: "dwordx4 nt, 2 x (10 reader waves, one step in flight, + a writer wave) per CU"}
        mix = {}
        # (the burst and role shapes write a region per wave / workgroup and burst: up to twice the stream plus slack, more
        # than a tight output buffer holds -- the probes get a buffer of their own, freed behind the timed region)
        pbuf = torch.empty(max(out.numel(), (n_bytes // 4096) * wpt * 2 + (64 << 20), n_words), dtype=torch.int32, device=dev)
        for shape in shapes:
            try:
                mix[shape] = enc.probe_bandwidth(d_in, pbuf, wpt, shape)[0]
            except hvlc.HvlcError:
                pass
        copy_ms = enc.probe_bandwidth(d_in, pbuf, 0, 7)[0]

    # ---- the W warm-up steps, untimed, right in front of the timed region
    for _ in range(args.warmup):
        step()

    # ---- the timed region: EXACTLY args.steps steps between two fences, nothing else on the stream but ONE HIP event in
    # front and one behind (per-launch events cost 4-13 us per step: 1 % of a 4 GiB step, 12 % of a 256 MiB one,
    # tools/launch_gap.py)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import gc
    gc.disable()                             # (no collector pause between the fences)
    fence()
    t0 = time.perf_counter()
    ev_a.record()
    for i in range(args.steps):
        step()
    ev_b.record()
    while not ev_b.query():                  # the host spins until the last step is done: a blocking wait's wake-up costs
        pass                                 # milliseconds on a busy host (one run of round 3: 2.7 ms on a 41 ms region)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    region_ms = ev_a.elapsed_time(ev_b)
    bits2, rc = enc.finish()
    assert rc == 0 and bits2 == bits
    if not args.no_probe:
        # the three best mix shapes, the copy and the read-only stream once more BEHIND the timed region (the GPU's state
        # drifts over a run: whichever of the two measurements is the better one counts -- a ceiling is a maximum)
        for shape in sorted(mix, key=mix.get)[:3]:
            mix[shape] = min(mix[shape], enc.probe_bandwidth(d_in, pbuf, wpt, shape)[0])
        copy_ms = min(copy_ms, enc.probe_bandwidth(d_in, pbuf, 0, 7)[0])
        rd_best = min(rd_best, enc.probe_bandwidth(d_in, None, 0, 0)[0])
        best_shape = min(mix, key=mix.get)
        probe = (rd_best, mix[best_shape], wpt, shapes[best_shape], {shapes[k]: round(n_bytes * (1.0 + wpt / 1024.0) / v / 1e6, 1) for k, v in mix.items()}, copy_ms)
    del pbuf                                 # (the probes' buffer: kept until here so that nothing pauses the GPU in front of the timed region)
    # ---- an instrumented pass of the same steps right behind it, NOT part of `value`: HIP events around every k_encode
    # launch (hvlc_profile_*) and between the steps, for the median / best figures of SURVEY.md 8(d)
    enc.profile_begin(args.steps)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    fence()
    kernel_ms = enc.profile_collect(args.steps)
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    bits2, rc = enc.finish()
    assert rc == 0 and bits2 == bits

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * n_bytes / (dt / args.steps) / 1e9

    # ---- bit-exactness of what was just timed, and the CPU baseline from the same CPU run -------
    import oracle_lib as O
    clock_ghz = enc.clock_ghz()
    base_bits = int(gathered[:rank].sum().item()) if world > 1 else 0
    if world == 1:
        # N = 1: a prefix of the stream against the REFERENCE's cpu_vlc_encode (oracle/_ref), one thread -- the same run is
        # the CPU baseline
        smp = min((args.cpu_sample_mib if not args.no_cpu_baseline else 64) << 20, n_bytes)
        smp -= smp % 1024                                       # whole segments
        x = d_in[:smp].cpu().numpy().view(np.uint32)
        use_ref = O.have_ref() and smp * max_len // 8 < (1 << 32)
        t1 = time.perf_counter()
        ref_words, osz = (O.ref_encode32 if use_ref else O.encode32)(x, cw, cl)
        sec = time.perf_counter() - t1
        seg_end = smp // 1024
        pre_bits = int(offsets[seg_end].item()) if seg_end < S else bits
        got = out[: pre_bits // 32 + 1].cpu().numpy().view(np.uint32)
        ok = (osz == (pre_bits + 7) // 8) and compare_prefix(got, ref_words, pre_bits)
        checked_here, checker = (pre_bits + 7) // 8, f"cpu_vlc_encode ({'reference' if use_ref else 'port'}) on a prefix"
    else:
        # N > 1: the WHOLE shard, every word and every segment offset, against the CPU encoder in chunks on this rank's share
        # of the host threads (oracle_compare_stream_mt: a fraction of a second per 4 GiB)
        use_ref = False
        x = d_in.cpu().numpy().view(np.uint32)
        loc = (offsets.cpu().numpy().view(np.uint64) - np.uint64(base_bits))
        got = out[: bits // 32 + 1].cpu().numpy().view(np.uint32)
        nbad, _ = O.compare_stream_mt(x, cw, cl, got, loc, bits, threads=max(1, (os.cpu_count() or 1) // world))
        ok = nbad == 0
        checked_here, checker = (bits + 7) // 8, "the whole shard against the CPU encoder (port), chunked, all host threads"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    checked = torch.tensor([checked_here], dtype=torch.int64, device=dev)
    kms = torch.zeros(world, dtype=torch.float64, device=dev)
    kms[rank] = float(np.mean(kernel_ms))
    if world > 1:
        all_reduce(flag, op=dist.ReduceOp.MIN)
        all_reduce(checked)
        all_reduce(kms)
    bit_exact = bool(flag.item())
    checked_bytes = int(checked.item())
    kms = [float(v) for v in kms.cpu().numpy()]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import hashlib
        model, pcores, threads = host_cpu()
        mt = sorted(O.encode_mt(x, cw, cl, threads)[1] for _ in range(3))     # (varies with the box's other tenants: median of three)
        ref_sha = hashlib.sha256(open(O.REF_SO, "rb").read()).hexdigest() if use_ref else None
        cpu = {"value": round(smp / sec / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "reference" if use_ref else "port",
               "sample": f"first {smp >> 20} MiB of the same stream, cpu_vlc_encode, 1 thread, {sec:.2f} s",
               "binary": ("oracle/_ref/libref_cpuencode.so (the reference's cpuencode.cpp compiled in the build container by oracle/Makefile; "
                          "this box cannot rebuild it), sha256 " + ref_sha) if use_ref else "oracle/liboracle.so (the repo's C restatement)",
               "cpu_model": model, "physical_cores": pcores, "hardware_threads": threads,
               "all_threads": {"value": round(smp / mt[1] / 1e9, 3), "min": round(smp / mt[2] / 1e9, 3), "max": round(smp / mt[0] / 1e9, 3),
                               "threads": threads, "kind": "port",
                               "note": "independent contiguous chunks, one per hardware thread; median, min and max of three runs (noisy: shared host)"}}

    # ---- roofline of the dominant kernel (rank-local) -----------------------------------------
    # algorithmic bytes per launch: read input once + write the stream once
    # + one uint64 offset per 1024-byte segment
    alg_bytes = n_bytes + (bits + 7) // 8 + 8 * S
    # the kernel's average launch duration: HIP events over the timed region / steps at N = 1 (one launch per step: the
    # figure includes the gaps between launches, it is an upper bound); with the stitch in the step (N > 1) the
    # per-launch events of the instrumented pass
    if world == 1:
        k_ms, k_src = region_ms / args.steps, "HIP events around the timed region / steps (includes the gaps between launches)"
    else:
        k_ms, k_src = float(np.mean(kernel_ms)), "per-launch HIP events of the instrumented pass behind the timed region"
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_src, valu_insts = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")      # PMC-derived HBM bytes per launch, if profiled
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
            ent = tj.get(args.workload if not args.size_mib else "", {})
            traffic = ent.get("hbm_bytes_per_launch")
            valu_insts = ent.get("valu_wave_insts_per_launch")
            traffic_src = "from profile: profiles/traffic.json (rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command, not measured in this run)"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_encode", "kernel_ms": round(k_ms, 4), "kernel_ms_source": k_src,
                "kernel_ms_median": round(float(np.median(kernel_ms)), 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "kernel_ms_mean_instrumented": round(float(np.mean(kernel_ms)), 4),
                "median_min_source": "per-launch HIP events of an instrumented pass of the same steps behind the timed region",
                "algorithmic_bytes": int(alg_bytes),
                "input_only_frac": round(n_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "shader_clock_ghz": round(clock_ghz, 3)}
    if world > 1:
        # every rank's kernel time (instrumented pass) and fraction: what a SCALE_rNN line needs per GPU
        roofline["ranks"] = {"kernel_ms": [round(v, 4) for v in kms],
                             "frac": [round(alg_bytes / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for v in kms],
                             "note": "rank r's own k_encode launches (per-launch HIP events); algorithmic bytes of rank 0's shard"}
    if valu_insts:
        # SURVEY.md 8(d)'s secondary ceiling, VALU issue: wave64 instructions per launch (SQ_INSTS_VALU of the profiled
        # command) against what the chip issues -- one integer VALU instruction per 4 clocks per SIMD, measured 0.575 per
        # ns per SIMD (profiles/r02_ubench.txt), 4 SIMDs per CU
        cus = enc.launch_info()["compute_units"]
        peak = 0.575e9 * 4 * cus
        roofline["valu_issue"] = {"wave_insts_per_launch": int(valu_insts), "peak_wave_insts_per_s": round(peak, 0),
                                  "frac": round(valu_insts / (k_ms * 1e-3) / peak, 4),
                                  "source": "from profile: SQ_INSTS_VALU in profiles/traffic.json (not measured in this run)"}
    if probe is not None:
        rd_best, mix_best, wpt, mix_shape, mix_all, copy_ms = probe
        mix_total = n_bytes * (1.0 + wpt / 1024.0)
        roofline.update({
            "measured_read_ceiling": round(n_bytes / rd_best / 1e6, 1),
            "measured_mix_ceiling": round(mix_total / mix_best / 1e6, 1),
            "measured_mix_ceiling_input": round(n_bytes / mix_best / 1e6, 1),
            "measured_mix_ceiling_shape": mix_shape,
            "measured_mix_by_shape": mix_all,
            "measured_copy": round(2.0 * n_bytes / copy_ms / 1e6, 1) if copy_ms else None,
            "measured_ceiling_note": f"same process, GB/s of bytes read + written, no encoding work, measured in front of the warm-up and, the best shapes once more, behind the timed "
                                     f"steps: read-only dwordx4 nt stream of the input; the same + {wpt * 4} B written per 4 KiB read, the MAXIMUM "
                                     f"over ten shapes (measured_mix_by_shape); a plain 1:1 dwordx4 nt copy of the input (measured_copy; the "
                                     f"guide's figure for a float4 copy is 6290)",
            "frac_of_measured_mix": round(achieved / (mix_total / mix_best / 1e6), 4)})

    if rank == 0:
        info = enc.launch_info()
        line = {
            "metric": "encode GB/s (input bytes)", "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "step_ms_median": round(float(np.median(step_ms)), 4),
            "step_ms_best": round(float(np.min(step_ms)), 4),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "bit_exact": bit_exact, "checked_bytes": checked_bytes,
            **({"rehearsal": True} if rehearsal else {}),
            "config": {"workload": f"{mib} MiB/GPU synthetic {dname} byte stream (splitmix64 seed {seed:#x}), "
                                   f"Huffman table of the job histogram (max len {max_len}); after the timed region the stream is "
                                   f"compared word for word with {checker}",
                       "bytes_per_gpu": n_bytes, "segments_per_gpu": S, "out_bits_per_gpu": bits,
                       "bits_per_symbol": round(bits / n_bytes, 4),
                       "parallelism": f"shard{world}" if world > 1 else "single",
                       "grid_blocks": info["grid_blocks"], "blocks_per_cu": info["blocks_per_cu"], "data_waves_per_workgroup": info["data_waves"]},
            "hbm_frac_of_peak_input_bytes": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not bit_exact:
        sys.exit("bench.py: the GPU stream differs from the CPU encoder's stream")


if __name__ == "__main__":
    main()
