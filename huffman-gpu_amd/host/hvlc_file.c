/*
 * hvlc_file.c -- file codec over the C ABI of libhvlc (plain C, HIP runtime for buffers): SURVEY.md section 8(f) #3,
 * "decoder + container".  The reference has neither: its driver (runVLCTest, main_test_cu.cu:52-180) encodes a file
 * and throws the stream away after comparing it.  This host keeps it:
 *
 *   hvlc_file c <input> <output.hvlc>     compress:   byte histogram on the device (hvlc_histogram), Huffman table on
 *                                         the host (hvlc_build_table), hvlc_encode, container written
 *   hvlc_file d <input.hvlc> <output>     decompress: per-segment bit counts -> offsets on the device
 *                                         (hvlc_exclusive_scan), hvlc_decode, original bytes written
 *
 * Container (little endian; all sections 8-byte aligned):
 *   char     magic[8]   "HVLC\1\0\0\0"
 *   uint64   num_bytes          original file size
 *   uint64   num_words          = num_bytes / 4: the words that are encoded (load_data.h:19)
 *   uint64   total_bits         length of the stream in bits
 *   uint64   num_segments       = ceil(num_words / 256)
 *   uint32   codewords[256], codewordlens[256]     the table, in the format hvlc_encode takes (load_data.h:40-47)
 *   uint16   seg_bits[num_segments] (+ padding)    bits of every 1024-byte segment (what the reference calls cindex,
 *                                                  vlc_kernel_sm64huff.cu:119-122; < 2^15 each); their exclusive scan
 *                                                  = the offsets hvlc_encode returned
 *   uint32   stream[total_bits / 32 + 1]           cpu_vlc_encode's stream, word for word
 *   uint8    tail[num_bytes % 4] (+ padding)       the bytes behind the last whole word, stored as they are
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

static const char kMagic[8] = { 'H', 'V', 'L', 'C', 1, 0, 0, 0 };
typedef struct { char magic[8]; uint64_t num_bytes, num_words, total_bits, num_segments; } header_t;

static void* read_file(const char* path, size_t* size)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc(n > 0 ? (size_t)n + 16 : 16);
    if (p && n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *size = (size_t)n;
    return p;
}
static size_t pad8(size_t n) { return (n + 7u) & ~(size_t)7u; }

static int compress(const char* in_path, const char* out_path)
{
    size_t n = 0;
    uint8_t* in = (uint8_t*)read_file(in_path, &n);
    if (!in) return 1;
    const uint64_t W = n / 4, S = hvlc_num_segments(W);
    hvlc_ctx* hv = NULL;
    uint8_t* d_in; uint64_t *d_hist, *d_offs; uint32_t *d_tab, *d_out;
    uint64_t hist[256]; uint32_t tab[512];
    CHECK_HIP(hipSetDevice(0));
    CHECK_HV(hvlc_create(&hv, 0, W));
    CHECK_HIP(hipMalloc((void**)&d_in, W ? W * 4 : 16));
    CHECK_HIP(hipMalloc((void**)&d_hist, 256 * 8));
    CHECK_HIP(hipMalloc((void**)&d_tab, 2048));
    CHECK_HIP(hipMalloc((void**)&d_offs, (S ? S : 1) * 8));
    CHECK_HIP(hipMemcpy(d_in, in, W * 4, hipMemcpyHostToDevice));
    CHECK_HV(hvlc_histogram(hv, d_in, W * 4, d_hist, NULL));          /* the whole words only: what gets encoded */
    CHECK_HIP(hipMemcpy(hist, d_hist, 256 * 8, hipMemcpyDeviceToHost));
    CHECK_HV(hvlc_build_table(hist, 31, tab, tab + 256));
    uint32_t max_len = 0;
    for (int s = 0; s < 256; ++s) if (tab[256 + s] > max_len) max_len = tab[256 + s];
    const uint64_t cap = hvlc_out_capacity_words(W, max_len);
    CHECK_HIP(hipMalloc((void**)&d_out, cap * 4));
    CHECK_HIP(hipMemcpy(d_tab, tab, 2048, hipMemcpyHostToDevice));
    uint64_t bits = 0;
    CHECK_HV(hvlc_encode(hv, (const uint32_t*)d_in, W, d_tab, d_tab + 256, d_out, cap, d_offs, NULL, NULL));
    CHECK_HV(hvlc_finish(hv, NULL, &bits));

    const uint64_t nw = bits / 32 + 1;
    uint32_t* stream = (uint32_t*)malloc(nw * 4);
    uint64_t* offs = (uint64_t*)malloc((S ? S : 1) * 8);
    uint16_t* segb = (uint16_t*)calloc(pad8(S * 2) / 2 + 4, 2);
    CHECK_HIP(hipMemcpy(stream, d_out, nw * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(offs, d_offs, S * 8, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < S; ++i) segb[i] = (uint16_t)((i + 1 < S ? offs[i + 1] : bits) - offs[i]);   /* < 1024 * 31 */

    header_t h;
    memcpy(h.magic, kMagic, 8);
    h.num_bytes = n; h.num_words = W; h.total_bits = bits; h.num_segments = S;
    FILE* f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 1; }
    uint8_t tail[8] = {0};
    memcpy(tail, in + W * 4, n & 3u);
    int ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(tab, 4, 512, f) == 512 &&
             fwrite(segb, 1, pad8(S * 2), f) == pad8(S * 2) && fwrite(stream, 4, nw, f) == nw &&
             ((nw & 1u) == 0 || fwrite(tail + 4, 4, 1, f) == 1) &&      /* keep the tail section 8-byte aligned */
             fwrite(tail, 1, 8, f) == 8;
    const long out_size = ftell(f);
    fclose(f);
    if (!ok) { fprintf(stderr, "short write to %s\n", out_path); return 1; }
    printf("{\"op\": \"compress\", \"input_bytes\": %llu, \"container_bytes\": %ld, \"total_bits\": %llu, \"segments\": %llu, "
           "\"max_len\": %u, \"ratio\": %.4f}\n", (unsigned long long)n, out_size, (unsigned long long)bits,
           (unsigned long long)S, max_len, n ? (double)out_size / (double)n : 0.0);
    hvlc_destroy(hv);
    hipFree(d_in); hipFree(d_hist); hipFree(d_tab); hipFree(d_offs); hipFree(d_out);
    free(in); free(stream); free(offs); free(segb);
    return 0;
}

static int decompress(const char* in_path, const char* out_path)
{
    size_t n = 0;
    uint8_t* c = (uint8_t*)read_file(in_path, &n);
    if (!c) return 1;
    header_t h;
    if (n < sizeof h + 2048) { fprintf(stderr, "%s: too short for a container\n", in_path); return 1; }
    memcpy(&h, c, sizeof h);
    const uint64_t W = h.num_words, S = h.num_segments, nw = h.total_bits / 32 + 1;
    const size_t off_tab = sizeof h, off_seg = off_tab + 2048, off_stream = off_seg + pad8(S * 2),
                 off_tail = off_stream + pad8(nw * 4);
    if (memcmp(h.magic, kMagic, 8) || W != h.num_bytes / 4 || S != hvlc_num_segments(W) || off_tail + 8 != n ||
        h.total_bits > W * 4u * 31u) {
        fprintf(stderr, "%s: not a valid hvlc container\n", in_path); return 1;
    }
    const uint32_t* tab = (const uint32_t*)(c + off_tab);
    const uint16_t* segb = (const uint16_t*)(c + off_seg);
    uint32_t* counts = (uint32_t*)malloc((S ? S : 1) * 4);
    uint64_t sum = 0;
    for (uint64_t i = 0; i < S; ++i) { counts[i] = segb[i]; sum += segb[i]; }
    if (sum != h.total_bits) { fprintf(stderr, "%s: segment bit counts do not add up to the stream length\n", in_path); return 1; }

    hvlc_ctx* hv = NULL;
    uint32_t *d_bits, *d_tab, *d_counts, *d_out; uint64_t* d_offs;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HV(hvlc_create(&hv, 0, W));
    CHECK_HIP(hipMalloc((void**)&d_bits, nw * 4));
    CHECK_HIP(hipMalloc((void**)&d_tab, 2048));
    CHECK_HIP(hipMalloc((void**)&d_counts, (S ? S : 1) * 4));
    CHECK_HIP(hipMalloc((void**)&d_offs, (S ? S : 1) * 8));
    CHECK_HIP(hipMalloc((void**)&d_out, W ? W * 4 : 16));
    CHECK_HIP(hipMemcpy(d_bits, c + off_stream, nw * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_tab, tab, 2048, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_counts, counts, S * 4, hipMemcpyHostToDevice));
    uint64_t total = 0;
    CHECK_HV(hvlc_exclusive_scan(hv, d_counts, S, d_offs, NULL, NULL));    /* bit counts -> segment offsets (A3) */
    CHECK_HV(hvlc_finish(hv, NULL, &total));
    if (S && total != h.total_bits) { fprintf(stderr, "scan total %llu != %llu\n", (unsigned long long)total, (unsigned long long)h.total_bits); return 1; }
    CHECK_HV(hvlc_decode(hv, d_bits, h.total_bits, d_offs, d_tab, d_tab + 256, d_out, W, NULL));
    uint8_t* out = (uint8_t*)malloc(h.num_bytes ? h.num_bytes : 1);
    CHECK_HIP(hipMemcpy(out, d_out, W * 4, hipMemcpyDeviceToHost));
    memcpy(out + W * 4, c + off_tail, h.num_bytes & 3u);
    FILE* f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 1; }
    const int ok = fwrite(out, 1, h.num_bytes, f) == h.num_bytes;
    fclose(f);
    if (!ok) { fprintf(stderr, "short write to %s\n", out_path); return 1; }
    printf("{\"op\": \"decompress\", \"container_bytes\": %llu, \"output_bytes\": %llu, \"total_bits\": %llu}\n",
           (unsigned long long)n, (unsigned long long)h.num_bytes, (unsigned long long)h.total_bits);
    hvlc_destroy(hv);
    hipFree(d_bits); hipFree(d_tab); hipFree(d_counts); hipFree(d_offs); hipFree(d_out);
    free(c); free(counts); free(out);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 4 && !strcmp(argv[1], "c")) return compress(argv[2], argv[3]);
    if (argc == 4 && !strcmp(argv[1], "d")) return decompress(argv[2], argv[3]);
    fprintf(stderr, "usage: %s c <input> <output.hvlc> | d <input.hvlc> <output>\n", argv[0]);
    return 1;
}
