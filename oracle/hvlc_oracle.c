/*
 * hvlc_oracle.c -- CPU ORACLE (test infrastructure only; see hvlc_oracle.h).
 *
 * A plain-C restatement of the reference's sequential encoder and of the
 * per-segment bookkeeping its GPU stages produce.  Written from the behaviour
 * of the reference (file:line cited per function), not from its text; pinned
 * against the reference compiled in place (oracle/_ref) and tests/golden/.
 */
#define _GNU_SOURCE
#include "hvlc_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* One sequential bit writer.  `cur` is the word under construction, `fill` the
 * number of bits already in it (0..31), `w` the index it will be stored at. */
typedef struct {
    uint32_t* out;  /* may be NULL: count only */
    uint64_t  w;
    uint32_t  cur;
    uint32_t  fill;
} bitw_t;

/* Append the low n bits of cw (n in 0..31), most significant first.
 * Follows cpuencode.cpp:32-40 including its one quirk: when the codeword
 * straddles a word boundary the leading part is `cw >> spill` WITHOUT masking
 * (cpuencode.cpp:35), so a table entry with bits set above its length leaks
 * them into the current word; the trailing part is masked (cpuencode.cpp:34). */
static inline void bitw_put(bitw_t* b, uint32_t cw, uint32_t n)
{
    if (n == 0) return;                       /* cpuencode.cpp:32: loop not entered */
    uint32_t room = 32u - b->fill;
    if (n < room) {                           /* fits, word stays open */
        b->cur |= (cw & ((1u << n) - 1u)) << (room - n);
        b->fill += n;
        return;
    }
    if (n == room) {                          /* fits exactly, word completes */
        b->cur |= cw & ((1u << n) - 1u);
    } else {                                  /* straddles: leading `room` bits now */
        b->cur |= cw >> (n - room);
    }
    if (b->out) b->out[b->w] = b->cur;
    b->w++;
    uint32_t rest = n - room;                 /* 0..30 bits go to the fresh word */
    b->cur  = rest ? (cw & ((1u << rest) - 1u)) << (32u - rest) : 0u;
    b->fill = rest;
}

/* Store the open word.  The reference zero-initialises every word it reaches,
 * including the one just past a stream that ends on a word boundary
 * (cpuencode.cpp:17,39), so out[w] is always written. */
static inline uint64_t bitw_close(bitw_t* b)
{
    if (b->out) b->out[b->w] = b->cur;
    return b->w * 32u + b->fill;
}

static inline void put_word(bitw_t* b, uint32_t v, const uint32_t* cw, const uint32_t* cl)
{
    /* symbol order: most significant byte first (cpuencode.cpp:28) */
    uint32_t s3 = v >> 24, s2 = (v >> 16) & 255u, s1 = (v >> 8) & 255u, s0 = v & 255u;
    bitw_put(b, cw[s3], cl[s3]);
    bitw_put(b, cw[s2], cl[s2]);
    bitw_put(b, cw[s1], cl[s1]);
    bitw_put(b, cw[s0], cl[s0]);
}

uint64_t oracle_vlc_encode64(const uint32_t* indata, uint64_t num_elements,
                             uint32_t* outdata,
                             const uint32_t* codewords, const uint32_t* codewordlens)
{
    bitw_t b = { outdata, 0, 0, 0 };
    for (uint64_t k = 0; k < num_elements; ++k)
        put_word(&b, indata[k], codewords, codewordlens);
    return bitw_close(&b);
}

void oracle_vlc_encode(const uint32_t* indata, uint32_t num_elements,
                       uint32_t* outdata, uint32_t* outsize,
                       const uint32_t* codewords, const uint32_t* codewordlens)
{
    uint64_t bits = oracle_vlc_encode64(indata, num_elements, outdata, codewords, codewordlens);
    /* cpuencode.cpp:44-45: whole words * 4 + remaining bits rounded up to bytes */
    *outsize = (uint32_t)((bits >> 5) * 4u + (((uint32_t)bits & 31u) + 7u) / 8u);
}

uint64_t oracle_segment_offsets(const uint32_t* indata, uint64_t num_elements,
                                uint32_t seg_words,
                                const uint32_t* codewordlens,
                                uint32_t* counts, uint64_t* offsets)
{
    uint64_t pos = 0, seg = 0;
    for (uint64_t k0 = 0; k0 < num_elements; k0 += seg_words, ++seg) {
        uint64_t k1 = k0 + seg_words < num_elements ? k0 + seg_words : num_elements;
        uint32_t c = 0;
        for (uint64_t k = k0; k < k1; ++k) {
            uint32_t v = indata[k];
            c += codewordlens[v >> 24] + codewordlens[(v >> 16) & 255u]
               + codewordlens[(v >> 8) & 255u] + codewordlens[v & 255u];
        }
        if (offsets) offsets[seg] = pos;      /* scan.cu:228 semantics: exclusive */
        if (counts)  counts[seg]  = c;        /* vlc_kernel_sm64huff.cu:120 */
        pos += c;
    }
    return pos;
}

void oracle_generate_codewords(uint32_t* codewords, uint32_t* codewordlens,
                               uint32_t num_symbols)
{
    static const uint32_t pattern[8] = { 1, 2, 3, 4, 4, 5, 6, 7 };   /* testdatagen.h:45-54 */
    for (uint32_t s = 0; s < num_symbols; ++s) {
        codewordlens[s] = pattern[s & 7u];
        codewords[s]    = 1u << (codewordlens[s] - 1u);              /* testdatagen.h:57-59 */
    }
}

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline uint8_t cdf_pick(const uint32_t* cdf32, uint32_t u)
{
    /* smallest s with cdf32[s] > u; cdf32[255] counts as 2^32 */
    uint32_t lo = 0, hi = 255;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cdf32[mid] > u) hi = mid; else lo = mid + 1;
    }
    return (uint8_t)lo;
}

void oracle_gen_symbols(uint8_t* out, uint64_t n_bytes, uint64_t seed,
                        uint64_t first_draw, const uint32_t* cdf32)
{
    for (uint64_t j = 0; j < n_bytes / 2; ++j) {
        uint64_t r = splitmix64(seed + first_draw + j);
        out[2 * j]     = cdf_pick(cdf32, (uint32_t)(r >> 32));
        out[2 * j + 1] = cdf_pick(cdf32, (uint32_t)r);
    }
}

void oracle_histogram(const uint8_t* data, uint64_t n_bytes, uint64_t* hist)
{
    memset(hist, 0, 256 * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_bytes; ++i) hist[data[i]]++;
}

uint64_t oracle_fnv1a64(const void* data, uint64_t n_bytes)
{
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < n_bytes; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

typedef struct {
    const uint32_t* in; uint64_t n; uint32_t* out;
    const uint32_t* cw; const uint32_t* cl; uint64_t bits;
} mt_job_t;

static void* mt_worker(void* p)
{
    mt_job_t* j = (mt_job_t*)p;
    j->bits = oracle_vlc_encode64(j->in, j->n, j->out, j->cw, j->cl);
    return NULL;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

uint64_t oracle_vlc_encode_mt(const uint32_t* indata, uint64_t num_elements,
                              uint32_t* scratch, uint64_t scratch_words,
                              const uint32_t* codewords, const uint32_t* codewordlens,
                              int threads, double* seconds)
{
    if (threads < 1) threads = 1;
    mt_job_t*  jobs = (mt_job_t*)calloc((size_t)threads, sizeof(mt_job_t));
    pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    /* contiguous chunks, multiples of one 256-word segment */
    uint64_t segs = (num_elements + 255u) / 256u;
    uint64_t per  = (segs + (uint64_t)threads - 1u) / (uint64_t)threads * 256u;
    uint64_t off  = 0;
    uint64_t stride = scratch_words / (uint64_t)threads;   /* caller sizes it for its table */
    for (int t = 0; t < threads; ++t) {
        uint64_t n = off < num_elements ? (num_elements - off < per ? num_elements - off : per) : 0;
        jobs[t] = (mt_job_t){ indata + off, n, scratch + stride * (uint64_t)t, codewords, codewordlens, 0 };
        off += n;
    }
    double t0 = now_s();
    for (int t = 1; t < threads; ++t) pthread_create(&tids[t], NULL, mt_worker, &jobs[t]);
    mt_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
    double t1 = now_s();
    uint64_t bits = 0;
    for (int t = 0; t < threads; ++t) bits += jobs[t].bits;
    if (seconds) *seconds = t1 - t0;
    free(jobs); free(tids);
    return bits;
}

/* ---- whole-stream comparison in chunks, multi-threaded (checker for the 4 GiB configurations) ----------------- */
typedef struct {
    const uint32_t* in; uint64_t n; const uint32_t* cw; const uint32_t* cl;
    const uint32_t* stream; uint64_t stream_words; const uint64_t* offs; uint64_t total_bits;
    uint64_t chunk_segs; int tid, threads; uint64_t bad, first_bad;
} cmp_job_t;

static uint32_t stream_get32(const uint32_t* s, uint64_t nwords, uint64_t pos)
{
    const uint64_t w = pos >> 5; const uint32_t sh = (uint32_t)(pos & 31u);
    const uint32_t hi = w < nwords ? s[w] : 0u, lo = w + 1 < nwords ? s[w + 1] : 0u;
    return sh ? ((hi << sh) | (lo >> (32u - sh))) : hi;
}

static void* cmp_worker(void* p)
{
    cmp_job_t* j = (cmp_job_t*)p;
    const uint64_t S = (j->n + 255u) / 256u;
    const uint64_t chunk_words = j->chunk_segs * 256u;
    uint32_t* buf   = (uint32_t*)malloc((size_t)(chunk_words * 4u * 31u / 32u + 4u) * sizeof(uint32_t));
    uint32_t* cnt   = (uint32_t*)malloc((size_t)j->chunk_segs * sizeof(uint32_t));
    uint64_t* loff  = (uint64_t*)malloc((size_t)j->chunk_segs * sizeof(uint64_t));
    j->bad = 0; j->first_bad = ~(uint64_t)0;
    for (uint64_t c = (uint64_t)j->tid; c * j->chunk_segs < S; c += (uint64_t)j->threads) {
        const uint64_t seg0 = c * j->chunk_segs, seg1 = seg0 + j->chunk_segs < S ? seg0 + j->chunk_segs : S;
        const uint64_t w0 = seg0 * 256u, w1 = seg1 * 256u < j->n ? seg1 * 256u : j->n;
        const uint64_t bits = oracle_vlc_encode64(j->in + w0, w1 - w0, buf, j->cw, j->cl);
        const uint64_t start = j->offs[seg0], end = seg1 < S ? j->offs[seg1] : j->total_bits;
        int ok = (end >= start) && (end - start == bits);
        if (ok) {
            (void)oracle_segment_offsets(j->in + w0, w1 - w0, 256u, j->cl, cnt, loff);
            for (uint64_t i = 0; ok && i < seg1 - seg0; ++i) ok = j->offs[seg0 + i] == start + loff[i];
        }
        const uint64_t full = bits >> 5;
        for (uint64_t i = 0; ok && i < full; ++i) ok = stream_get32(j->stream, j->stream_words, start + 32u * i) == buf[i];
        if (ok && (bits & 31u)) {
            const uint32_t mask = ~0u << (32u - (uint32_t)(bits & 31u));
            ok = ((stream_get32(j->stream, j->stream_words, start + 32u * full) ^ buf[full]) & mask) == 0u;
        }
        if (!ok) { j->bad++; if (seg0 < j->first_bad) j->first_bad = seg0; }
    }
    free(buf); free(cnt); free(loff);
    return NULL;
}

uint64_t oracle_compare_stream_mt(const uint32_t* indata, uint64_t num_elements,
                                  const uint32_t* codewords, const uint32_t* codewordlens,
                                  const uint32_t* stream, uint64_t stream_words,
                                  const uint64_t* seg_offsets, uint64_t total_bits,
                                  uint64_t chunk_segments, int threads, uint64_t* first_bad_segment)
{
    if (threads < 1) threads = 1;
    if (chunk_segments < 1) chunk_segments = 1;
    cmp_job_t* jobs = (cmp_job_t*)calloc((size_t)threads, sizeof(cmp_job_t));
    pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; ++t)
        jobs[t] = (cmp_job_t){ indata, num_elements, codewords, codewordlens, stream, stream_words, seg_offsets, total_bits,
                               chunk_segments, t, threads, 0, ~(uint64_t)0 };
    for (int t = 1; t < threads; ++t) pthread_create(&tids[t], NULL, cmp_worker, &jobs[t]);
    cmp_worker(&jobs[0]);
    for (int t = 1; t < threads; ++t) pthread_join(tids[t], NULL);
    uint64_t bad = 0, first = ~(uint64_t)0;
    for (int t = 0; t < threads; ++t) { bad += jobs[t].bad; if (jobs[t].first_bad < first) first = jobs[t].first_bad; }
    if (first_bad_segment) *first_bad_segment = first;
    free(jobs); free(tids);
    return bad;
}

/* ---- bit-exact append (checker for hvlc_merge_shard) ------------------------------ */
uint64_t oracle_bit_append(uint32_t* dst, uint64_t dst_bits, const uint32_t* src, uint64_t nbits)
{
    for (uint64_t i = 0; i < nbits; ++i) {
        const uint32_t bit = (src[i >> 5] >> (31u - (uint32_t)(i & 31u))) & 1u;
        const uint64_t d = dst_bits + i;
        if (bit) dst[d >> 5] |= 1u << (31u - (uint32_t)(d & 31u));
    }
    return dst_bits + nbits;
}

/* ---- reference decoder (checker for hvlc_decode) ---------------------------------- */
uint64_t oracle_vlc_decode(const uint32_t* bits, uint64_t start_bit, uint32_t* outdata, uint64_t num_elements,
                           const uint32_t* codewords, const uint32_t* codewordlens)
{
    uint64_t pos = start_bit;
    for (uint64_t w = 0; w < num_elements; ++w) {
        uint32_t word = 0;
        for (int b = 0; b < 4; ++b) {
            uint32_t acc = 0, len = 0;
            int found = -1;
            while (found < 0 && len < 31u) {
                acc = (acc << 1) | ((bits[pos >> 5] >> (31u - (uint32_t)(pos & 31u))) & 1u);
                ++pos; ++len;
                for (int s = 0; s < 256; ++s)
                    if (codewordlens[s] == len && codewords[s] == acc) { found = s; break; }
            }
            if (found < 0) return ~(uint64_t)0;
            word = (word << 8) | (uint32_t)found;
        }
        outdata[w] = word;
    }
    return pos - start_bit;
}
