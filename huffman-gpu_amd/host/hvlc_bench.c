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
 *   hvlc_bench <input file> <table file: codewords[256] | lens[256], uint32 LE, or "-"> [--iters N]
 *              [--dump PREFIX] [--verify] [--stats DIR]
 *
 * Table "-": the table is built from the file itself as loadData does (load_data.h:25-58): byte histogram on the
 * device (hvlc_histogram), Huffman table on the host (hvlc_build_table), entropy line printed in the reference's
 * format.  --verify decodes the stream on the device (hvlc_decode) and compares it with the input.
 * --stats DIR appends one point (input MiB, encode GB/s) to a series file in the reference's stats_logger format
 * (stats_logger.cpp:26-42).  Tail bytes beyond a multiple of 4 are dropped, as load_data.h:19 does
 * (num_elements = mem_size / 4), but -- unlike the reference, whose grid needs multiples of 4096 words
 * (load_data.h:21) -- every whole word is encoded.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

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
    if (argc < 3) { fprintf(stderr, "usage: %s <input> <table.bin|-> [--iters N] [--dump PREFIX] [--verify] [--stats DIR]\n", argv[0]); return 1; }
    int iters = 10, verify = 0;
    const char* dump = NULL;
    const char* stats = NULL;
    for (int i = 3; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--stats") && i + 1 < argc) stats = argv[++i];
        else if (!strcmp(argv[i], "--verify")) verify = 1;
    }
    const int build = !strcmp(argv[2], "-");
    size_t in_bytes = 0, tab_bytes = 0;
    uint32_t* in = (uint32_t*)read_file(argv[1], &in_bytes);
    uint32_t* tab = build ? (uint32_t*)calloc(512, 4) : (uint32_t*)read_file(argv[2], &tab_bytes);
    if (!in || !tab || (!build && tab_bytes != 2048)) { fprintf(stderr, "bad input or table (need 2048-byte table)\n"); return 1; }
    const uint64_t W = in_bytes / 4;
    if (build) {                                      /* loadData (load_data.h:25-58): histogram -> table -> entropy */
        hvlc_ctx* h0 = NULL;
        uint8_t* d_b; uint64_t* d_h; uint64_t hist[256];
        CHECK_HIP(hipSetDevice(0));
        CHECK_HV(hvlc_create(&h0, 0, 0));
        CHECK_HIP(hipMalloc((void**)&d_b, W ? W * 4 : 4));
        CHECK_HIP(hipMalloc((void**)&d_h, 256 * 8));
        CHECK_HIP(hipMemcpy(d_b, in, W * 4, hipMemcpyHostToDevice));
        CHECK_HV(hvlc_histogram(h0, d_b, W * 4, d_h, NULL));
        CHECK_HIP(hipMemcpy(hist, d_h, 256 * 8, hipMemcpyDeviceToHost));
        CHECK_HV(hvlc_build_table(hist, 31, tab, tab + 256));
        double H = 0.0;
        for (int s = 0; s < 256; ++s)
            if (hist[s]) { const double pr = (double)hist[s] / (double)(W * 4); H -= pr * log(pr) / log(2.0); }
        printf("\n%s, %llu bytes, entropy %f\n\n", argv[1], (unsigned long long)(W * 4), H);      /* load_data.h:56 */
        hipFree(d_b); hipFree(d_h); hvlc_destroy(h0);
    }
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
    int verified = -1;
    if (verify) {                                     /* round trip on the device: decode and compare with the input */
        uint32_t* d_back;
        CHECK_HIP(hipMalloc((void**)&d_back, W ? W * 4 : 4));
        CHECK_HV(hvlc_decode(hv, d_out, bits, d_offs, d_tab, d_tab + 256, d_back, W, NULL));
        uint32_t* back = (uint32_t*)malloc(W ? W * 4 : 4);
        CHECK_HIP(hipMemcpy(back, d_back, W * 4, hipMemcpyDeviceToHost));
        verified = memcmp(back, in, W * 4) == 0;
        free(back); hipFree(d_back);
        if (!verified) fprintf(stderr, "VERIFY FAILED: decode(encode(input)) != input\n");
    }
    if (stats) {                                      /* one point of the series, stats_logger.cpp:26-42 format */
        char path[1024];
        snprintf(path, sizeof path, "%s/encoding__0_hvlc.txt", stats);
        FILE* probe = fopen(path, "rt");
        FILE* f = fopen(path, "at");
        if (!f) { perror(path); return 1; }
        if (!probe)
            fprintf(f, "SERIES_NAME\nhvlc\nX_AXIS_QUANTITY\ninput size\nY_AXIS_QUANTITY\nencode throughput\nX_AXIS_UNIT\nMiB\n"
                       "Y_AXIS_UNIT\nGB/s\nX_AXIS_SCALE_TYPE\nlog\nY_AXIS_SCALE_TYPE\nlin\nDESCRIPTION\nlibhvlc %s\n__DATA__\n", hvlc_version());
        else fclose(probe);
        fprintf(f, "%f %f\n", (double)W * 4 / 1048576.0, ms > 0 ? (double)W * 4 / ms / 1e6 : 0.0);
        fclose(f);
    }
    int grid = 0, bpc = 0, cus = 0;
    hvlc_get_launch_info(hv, &grid, &bpc, &cus);
    /* the reference prints refbytesize the same way: whole words * 4 + remaining bits rounded up (cpuencode.cpp:44-45) */
    const uint64_t outsize = (bits >> 5) * 4 + ((bits & 31) + 7) / 8;
    printf("{\"input_bytes\": %llu, \"total_bits\": %llu, \"outsize\": %llu, \"ms\": %.4f, \"GBps_in\": %.2f, "
           "\"grid\": %d, \"blocks_per_cu\": %d, \"cus\": %d, \"verified\": %d, \"tail_bytes_dropped\": %d, \"version\": \"%s\"}\n",
           (unsigned long long)(W * 4), (unsigned long long)bits, (unsigned long long)outsize, ms,
           ms > 0 ? (double)W * 4 / ms / 1e6 : 0.0, grid, bpc, cus, verified, (int)(in_bytes & 3), hvlc_version());
    if (dump && build) {                              /* the table that was built, in the table-file format */
        char path[1024];
        snprintf(path, sizeof path, "%s.table.bin", dump);
        FILE* f = fopen(path, "wb"); if (!f) { perror(path); return 1; }
        fwrite(tab, 4, 512, f); fclose(f);
    }
    hvlc_destroy(hv);
    hipFree(d_in); hipFree(d_tab); hipFree(d_out); hipFree(d_offs);
    free(in); free(tab);
    return verified == 0 ? 4 : 0;
}
