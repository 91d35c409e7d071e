/*
 * hvlc_mgpu.c -- C multi-GPU host of libhvlc: ONE process drives 1..8 MI355X of a node (plain C, HIP runtime,
 * RCCL over xGMI).  The native counterpart of the driver role of runVLCTest (main_test_cu.cu:52-180) for the sharded
 * job of SURVEY.md section 8(e): the reference has one device and no communication layer; here
 *
 *   per device g    hipSetDevice(g), its own hvlc_ctx, stream and buffers; shard g of the synthetic job stream is
 *                   generated in place (hvlc_gen_symbols from draw index g * shard_bytes / 2: the shards of all
 *                   devices, concatenated, are ONE generation from draw 0)
 *   table           per-device byte histograms (hvlc_histogram) summed with ncclAllReduce, Huffman table of the JOB's
 *                   histogram built on the host (hvlc_build_table), uploaded to every device
 *   one step        hvlc_encode on every device (each from local bit 0), then the stitch: ONE ncclAllGather of one
 *                   uint64 per rank (the per-shard bit totals) and hvlc_rebase_offsets_dev (base = exclusive prefix
 *                   of the gathered totals, computed on the device) -- all enqueued on the devices' streams
 *   --merge         the contiguous stream of the whole job on device 0: every device copies its shard to its global
 *                   bit position with hvlc_merge_shard, writing device 0's memory over xGMI (peer access); the words
 *                   at the shard seams are merged with 4-byte atomic ORs, as pack2's first/last word are
 *                   (pack_kernels.cu:32-34,45-51)
 *
 *   hvlc_mgpu [--gpus N] [--mib M] [--iters K] [--dist H2.2|H4.0|H1.0] [--period FILE] [--seed S] [--merge] [--dump PREFIX]
 *
 * --dist H2.2 needs --period FILE (the 256 KiB period of the reference's test file: its byte histogram is the
 * distribution).  Prints one JSON line.  No CPU encoder here: tests/test_c_mgpu_gpu.py compares the dumps with the oracle.
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/hvlc.h"

#define MAXG 8
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HV(x) do { int r_ = (x); if (r_ != HVLC_OK) { \
    fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hvlc_status_string(r_)); return 3; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { \
    fprintf(stderr, "%s:%d: RCCL: %s\n", __FILE__, __LINE__, ncclGetErrorString(r_)); return 4; } } while (0)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* 256-entry 32-bit cumulative table of the synthetic generator (the same arithmetic as hvlc.cdf32_from_counts) */
static void cdf32_from_weights(const double* w, uint32_t* cdf)
{
    double sum = 0.0, run = 0.0;
    int lastlive = 0;
    for (int s = 0; s < 256; ++s) { sum += w[s]; if (w[s] > 0.0) lastlive = s; }
    for (int s = 0; s < 256; ++s) {
        run += w[s];
        double v = floor(run / sum * 4294967296.0);
        if (v > 4294967295.0) v = 4294967295.0;
        cdf[s] = (uint32_t)v;
    }
    for (int s = lastlive; s < 256; ++s) cdf[s] = 0xFFFFFFFFu;
}

static int weights_for(const char* dist, const char* period_path, double* w)
{
    memset(w, 0, 256 * sizeof(double));
    if (!strcmp(dist, "H4.0")) { for (int j = 0; j < 16; ++j) w[16 * j + 3] = 1.0; return 0; }
    if (!strcmp(dist, "H1.0")) {
        double s = 0.83; w[0] = 0.83;
        for (int i = 1; i <= 24; ++i) { w[i] = 0.17 * ldexp(1.0, -i); }
        for (int i = 1; i <= 24; ++i) s += w[i];
        w[24] += 1.0 - s;
        return 0;
    }
    if (!strcmp(dist, "H2.2")) {
        FILE* f = period_path ? fopen(period_path, "rb") : NULL;
        if (!f) { fprintf(stderr, "--dist H2.2 needs --period FILE\n"); return 1; }
        int c;
        while ((c = fgetc(f)) != EOF) w[c] += 1.0;
        fclose(f);
        return 0;
    }
    fprintf(stderr, "unknown --dist %s\n", dist);
    return 1;
}

int main(int argc, char** argv)
{
    int gpus = 1, iters = 10, merge = 0;
    uint64_t mib = 256, seed = 0x5EED0005ull;
    const char *dist = "H4.0", *period = NULL, *dump = NULL;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--mib") && i + 1 < argc) mib = strtoull(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dist") && i + 1 < argc) dist = argv[++i];
        else if (!strcmp(argv[i], "--period") && i + 1 < argc) period = argv[++i];
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = strtoull(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--merge")) merge = 1;
        else { fprintf(stderr, "usage: %s [--gpus N] [--mib M] [--iters K] [--dist H2.2|H4.0|H1.0] [--period FILE] [--seed S] [--merge] [--dump PREFIX]\n", argv[0]); return 1; }
    }
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    if (gpus < 1 || gpus > MAXG || gpus > ndev) { fprintf(stderr, "--gpus %d but %d device(s) present (max %d)\n", gpus, ndev, MAXG); return 1; }
    const uint64_t shard_bytes = mib << 20, W = shard_bytes / 4, S = hvlc_num_segments(W);
    double w[256];
    uint32_t cdf[256];
    if (weights_for(dist, period, w)) return 1;
    cdf32_from_weights(w, cdf);

    /* ---- per-device state ------------------------------------------------------------------------------------- */
    hvlc_ctx* hv[MAXG]; hipStream_t st[MAXG]; ncclComm_t comm[MAXG]; int devs[MAXG];
    uint8_t* d_in[MAXG]; uint32_t *d_cdf[MAXG], *d_tab[MAXG], *d_out[MAXG];
    uint64_t *d_hist[MAXG], *d_offs[MAXG], *d_total[MAXG], *d_gath[MAXG];
    for (int g = 0; g < gpus; ++g) devs[g] = g;
    CHECK_NCCL(ncclCommInitAll(comm, gpus, devs));
    for (int g = 0; g < gpus; ++g) {
        CHECK_HIP(hipSetDevice(g));
        CHECK_HIP(hipStreamCreate(&st[g]));
        CHECK_HV(hvlc_create(&hv[g], g, W));
        CHECK_HIP(hipMalloc((void**)&d_in[g], shard_bytes));
        CHECK_HIP(hipMalloc((void**)&d_cdf[g], 1024));
        CHECK_HIP(hipMalloc((void**)&d_tab[g], 2048));
        CHECK_HIP(hipMalloc((void**)&d_hist[g], 256 * 8));
        CHECK_HIP(hipMalloc((void**)&d_offs[g], (S ? S : 1) * 8));
        CHECK_HIP(hipMalloc((void**)&d_total[g], 8));
        CHECK_HIP(hipMalloc((void**)&d_gath[g], MAXG * 8));
        CHECK_HIP(hipMemcpyAsync(d_cdf[g], cdf, 1024, hipMemcpyHostToDevice, st[g]));
        /* shard g of the job stream: two symbols per draw -> shards start shard_bytes / 2 draws apart */
        CHECK_HV(hvlc_gen_symbols(hv[g], d_in[g], shard_bytes, seed, (uint64_t)g * (shard_bytes / 2), d_cdf[g], st[g]));
        CHECK_HV(hvlc_histogram(hv[g], d_in[g], shard_bytes, d_hist[g], st[g]));
    }
    /* ---- the job's histogram (RCCL all-reduce) and its table --------------------------------------------------- */
    CHECK_NCCL(ncclGroupStart());
    for (int g = 0; g < gpus; ++g) CHECK_NCCL(ncclAllReduce(d_hist[g], d_hist[g], 256, ncclUint64, ncclSum, comm[g], st[g]));
    CHECK_NCCL(ncclGroupEnd());
    uint64_t hist[256];
    uint32_t tab[512];
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMemcpyAsync(hist, d_hist[0], 256 * 8, hipMemcpyDeviceToHost, st[0]));
    CHECK_HIP(hipStreamSynchronize(st[0]));
    CHECK_HV(hvlc_build_table(hist, 31, tab, tab + 256));
    uint32_t max_len = 0;
    for (int s = 0; s < 256; ++s) if (tab[256 + s] > max_len) max_len = tab[256 + s];
    const uint64_t cap = hvlc_out_capacity_words(W, max_len);
    for (int g = 0; g < gpus; ++g) {
        CHECK_HIP(hipSetDevice(g));
        CHECK_HIP(hipMalloc((void**)&d_out[g], cap * 4));
        CHECK_HIP(hipMemcpyAsync(d_tab[g], tab, 2048, hipMemcpyHostToDevice, st[g]));
    }

    /* ---- the step: encode every shard, stitch the offsets with one all-gather of 8 bytes per rank -------------- */
    double dt = 0.0;
    for (int it = -2; it < iters; ++it) {                 /* two warm-up steps */
        if (it == 0) {
            for (int g = 0; g < gpus; ++g) { CHECK_HIP(hipSetDevice(g)); CHECK_HIP(hipStreamSynchronize(st[g])); }
            dt = now_s();
        }
        for (int g = 0; g < gpus; ++g) {
            CHECK_HIP(hipSetDevice(g));
            CHECK_HV(hvlc_encode(hv[g], (const uint32_t*)d_in[g], W, d_tab[g], d_tab[g] + 256, d_out[g], cap, d_offs[g], d_total[g], st[g]));
        }
        CHECK_NCCL(ncclGroupStart());
        for (int g = 0; g < gpus; ++g) CHECK_NCCL(ncclAllGather(d_total[g], d_gath[g], 1, ncclUint64, comm[g], st[g]));
        CHECK_NCCL(ncclGroupEnd());
        for (int g = 0; g < gpus; ++g) {
            CHECK_HIP(hipSetDevice(g));
            CHECK_HV(hvlc_rebase_offsets_dev(hv[g], d_offs[g], S, d_gath[g], g, st[g]));
        }
    }
    uint64_t totals[MAXG], job_bits = 0;
    for (int g = 0; g < gpus; ++g) {
        CHECK_HIP(hipSetDevice(g));
        CHECK_HV(hvlc_finish(hv[g], st[g], &totals[g]));   /* synchronises the stream, reports device-side errors */
    }
    dt = (now_s() - dt) / (double)(iters > 0 ? iters : 1);
    uint64_t gathered0[MAXG];
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMemcpy(gathered0, d_gath[0], (size_t)gpus * 8, hipMemcpyDeviceToHost));
    int stitch_ok = 1;
    for (int g = 0; g < gpus; ++g) { stitch_ok &= gathered0[g] == totals[g]; job_bits += totals[g]; }

    /* ---- optional: the contiguous stream of the whole job on device 0, shards written over xGMI ---------------- */
    uint32_t* d_merged = NULL;
    double merge_ms = 0.0;
    int peer_ok = 1;
    const uint64_t merged_words = job_bits / 32 + 1;
    if (merge) {
        CHECK_HIP(hipSetDevice(0));
        CHECK_HIP(hipMalloc((void**)&d_merged, merged_words * 4));
        CHECK_HIP(hipMemset(d_merged, 0, merged_words * 4));        /* the seam words are OR-merged: they must start as zero */
        CHECK_HIP(hipDeviceSynchronize());                          /* ... on device 0, before any peer writes into it */
        for (int g = 1; g < gpus; ++g) {
            int can = 0;
            CHECK_HIP(hipDeviceCanAccessPeer(&can, g, 0));
            if (!can) { peer_ok = 0; continue; }
            CHECK_HIP(hipSetDevice(g));
            hipError_t e = hipDeviceEnablePeerAccess(0, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { fprintf(stderr, "peer access %d->0: %s\n", g, hipGetErrorString(e)); return 2; }
            (void)hipGetLastError();
        }
        if (!peer_ok) { fprintf(stderr, "no peer access to device 0: --merge skipped\n"); }
        else {
            const double t0 = now_s();
            uint64_t base = 0;
            for (int g = 0; g < gpus; ++g) {               /* every device writes its own shard: all seams in flight at once */
                CHECK_HIP(hipSetDevice(g));
                CHECK_HV(hvlc_merge_shard(hv[g], d_out[g], totals[g], d_merged, merged_words, base, st[g]));
                base += totals[g];
            }
            for (int g = 0; g < gpus; ++g) { CHECK_HIP(hipSetDevice(g)); CHECK_HIP(hipStreamSynchronize(st[g])); }
            merge_ms = (now_s() - t0) * 1e3;
        }
    }

    if (dump) {
        char path[1024];
        FILE* f;
        snprintf(path, sizeof path, "%s.table.bin", dump);
        f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
        fwrite(tab, 4, 512, f); fclose(f);
        for (int g = 0; g < gpus; ++g) {
            const uint64_t nw = totals[g] / 32 + 1;
            uint32_t* out = (uint32_t*)malloc(nw * 4);
            uint64_t* offs = (uint64_t*)malloc((S ? S : 1) * 8);
            CHECK_HIP(hipSetDevice(g));
            CHECK_HIP(hipMemcpy(out, d_out[g], nw * 4, hipMemcpyDeviceToHost));
            CHECK_HIP(hipMemcpy(offs, d_offs[g], S * 8, hipMemcpyDeviceToHost));
            snprintf(path, sizeof path, "%s.dev%d.stream.bin", dump, g);
            f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
            fwrite(out, 4, nw, f); fclose(f);
            snprintf(path, sizeof path, "%s.dev%d.offsets.bin", dump, g);
            f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
            fwrite(offs, 8, S, f); fclose(f);
            free(out); free(offs);
        }
        if (d_merged && peer_ok) {
            uint32_t* m = (uint32_t*)malloc(merged_words * 4);
            CHECK_HIP(hipSetDevice(0));
            CHECK_HIP(hipMemcpy(m, d_merged, merged_words * 4, hipMemcpyDeviceToHost));
            snprintf(path, sizeof path, "%s.merged.bin", dump);
            f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
            fwrite(m, 4, merged_words, f); fclose(f);
            free(m);
        }
    }

    printf("{\"metric\": \"encode GB/s (input bytes)\", \"value\": %.2f, \"unit\": \"GB/s\", \"n_gpus\": %d, \"steps\": %d, "
           "\"ms_per_step\": %.4f, \"scaling\": \"weak\", \"host\": \"C, one process, RCCL\", \"workload\": \"%llu MiB/GPU synthetic %s\", "
           "\"job_bits\": %llu, \"max_len\": %u, \"stitch_ok\": %s, \"merged\": %s, \"merge_ms\": %.3f, \"version\": \"%s\"}\n",
           (double)gpus * (double)shard_bytes / dt / 1e9, gpus, iters, dt * 1e3, (unsigned long long)mib, dist,
           (unsigned long long)job_bits, max_len, stitch_ok ? "true" : "false", (d_merged && peer_ok) ? "true" : "false", merge_ms,
           hvlc_version());

    for (int g = 0; g < gpus; ++g) {
        CHECK_HIP(hipSetDevice(g));
        hvlc_destroy(hv[g]);
        hipFree(d_in[g]); hipFree(d_cdf[g]); hipFree(d_tab[g]); hipFree(d_hist[g]); hipFree(d_offs[g]); hipFree(d_total[g]);
        hipFree(d_gath[g]); hipFree(d_out[g]);
        ncclCommDestroy(comm[g]);
        hipStreamDestroy(st[g]);
    }
    if (d_merged) { CHECK_HIP(hipSetDevice(0)); hipFree(d_merged); }
    return stitch_ok ? 0 : 5;
}
