/*
 * hvlc_bench.c -- C host for libhvlc (plain C, HIP runtime for buffers and events).
 *
 * The native counterpart of the reference's driver runVLCTest (main_test_cu.cu:52-180), re-created, not
 * ported: read a file (or take a table + stream produced elsewhere), upload, run the GPU encode path
 * through the C ABI, time it with HIP events, optionally dump the stream / offsets for an external
 * bit-exact comparison, print one JSON line.  It deliberately contains no CPU encoder: verification
 * against the oracle is done by the tests (tests/test_c_harness_gpu.py), which run this binary and compare
 * its dump with oracle/.
 *
 *   hvlc_bench <input file> <table file: codewords[256] | lens[256], uint32 LE> [--iters N] [--dump PREFIX]
 *
 * Tail bytes beyond a multiple of 4 are dropped, as load_data.h:19 does (num_elements = mem_size / 4).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hvlc.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HV(x) do { int r_ = (x); if (r_ != HVLC_OK) { \
    fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hvlc_status_string(r_)); return 3; } } while (0)

static void* read_file(const char* path, size_t* size)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n : 1);
    if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *size = (size_t)n;
    return p;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <input> <table.bin> [--iters N] [--dump PREFIX]\n", argv[0]); return 1; }
    int iters = 10;
    const char* dump = NULL;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
    }
    size_t in_bytes = 0, tab_bytes = 0;
    uint32_t* in = (uint32_t*)read_file(argv[1], &in_bytes);
    uint32_t* tab = (uint32_t*)read_file(argv[2], &tab_bytes);
    if (!in || !tab || tab_bytes != 2048) { fprintf(stderr, "bad input or table (need 2048-byte table)\n"); return 1; }
    const uint64_t W = in_bytes / 4;
    uint32_t max_len = 0;
    for (int s = 0; s < 256; ++s) if (tab[256 + s] > max_len) max_len = tab[256 + s];
    const uint64_t cap = hvlc_out_capacity_words(W, max_len);
    const uint64_t S = hvlc_num_segments(W);

    uint32_t *d_in, *d_tab, *d_out;
    uint64_t *d_offs;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMalloc((void**)&d_in, W ? W * 4 : 4));
    CHECK_HIP(hipMalloc((void**)&d_tab, 2048));
    CHECK_HIP(hipMalloc((void**)&d_out, cap * 4));
    CHECK_HIP(hipMalloc((void**)&d_offs, (S ? S : 1) * 8));
    CHECK_HIP(hipMemcpy(d_in, in, W * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tab, tab, 2048, hipMemcpyHostToDevice));

    hvlc_ctx* hv = NULL;
    CHECK_HV(hvlc_create(&hv, 0, W));
    uint64_t bits = 0;
    CHECK_HV(hvlc_encode(hv, d_in, W, d_tab, d_tab + 256, d_out, cap, d_offs, NULL, NULL));   /* warm-up */
    CHECK_HV(hvlc_finish(hv, NULL, &bits));

    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, NULL));
    for (int i = 0; i < iters; ++i)
        CHECK_HV(hvlc_encode(hv, d_in, W, d_tab, d_tab + 256, d_out, cap, d_offs, NULL, NULL));
    CHECK_HIP(hipEventRecord(e1, NULL));
    CHECK_HV(hvlc_finish(hv, NULL, &bits));
    float ms = 0.f;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)(iters > 0 ? iters : 1);

    if (dump) {
        const uint64_t nw = bits / 32 + 1;
        uint32_t* out = (uint32_t*)malloc(nw * 4);
        uint64_t* offs = (uint64_t*)malloc((S ? S : 1) * 8);
        CHECK_HIP(hipMemcpy(out, d_out, nw * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(offs, d_offs, S * 8, hipMemcpyDeviceToHost));
        char path[1024];
        snprintf(path, sizeof path, "%s.stream.bin", dump);
        FILE* f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
        fwrite(out, 4, nw, f); fclose(f);
        snprintf(path, sizeof path, "%s.offsets.bin", dump);
        f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
        fwrite(offs, 8, S, f); fclose(f);
        free(out); free(offs);
    }
    int grid = 0, bpc = 0, cus = 0;
    hvlc_get_launch_info(hv, &grid, &bpc, &cus);
    /* the reference prints refbytesize the same way: whole words * 4 + remaining bits rounded up (cpuencode.cpp:44-45) */
    const uint64_t outsize = (bits >> 5) * 4 + ((bits & 31) + 7) / 8;
    printf("{\"input_bytes\": %llu, \"total_bits\": %llu, \"outsize\": %llu, \"ms\": %.4f, \"GBps_in\": %.2f, "
           "\"grid\": %d, \"blocks_per_cu\": %d, \"cus\": %d, \"version\": \"%s\"}\n",
           (unsigned long long)(W * 4), (unsigned long long)bits, (unsigned long long)outsize, ms,
           ms > 0 ? (double)W * 4 / ms / 1e6 : 0.0, grid, bpc, cus, hvlc_version());
    hvlc_destroy(hv);
    hipFree(d_in); hipFree(d_tab); hipFree(d_out); hipFree(d_offs);
    free(in); free(tab);
    return 0;
}
