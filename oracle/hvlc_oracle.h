/*
 * hvlc_oracle.h -- CPU ORACLE for the VLC/Huffman encode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so, and there only as the checker / the reported CPU
 * baseline, never as the thing measured or shipped.  libhvlc.so does not link
 * against, include, or fall back to anything in this directory.
 *
 * Parity status: PINNED.  oracle_vlc_encode() is differential-tested against
 * the reference's own cpuencode.cpp compiled from where it lies
 * (oracle/Makefile -> oracle/_ref/libref_cpuencode.so) and against the golden
 * vectors of SURVEY.md section 4 (G1, G1p, K1-K3) kept under tests/golden/.
 *
 * Each function cites the reference file:line whose behaviour it restates
 * (paths relative to the reference checkout).
 */
#ifndef HVLC_ORACLE_H_
#define HVLC_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Restates cpu_vlc_encode (cpuencode.cpp:12-46, declared cpuencode.h:4-7).
 * Same argument order and meaning: indata[num_elements] uint32 words, symbols
 * taken from the most significant byte first; codewords[256]/codewordlens[256]
 * the flattened table; outdata receives the MSB-first bit stream in native
 * uint32 words and must hold floor(bits/32)+1 words; *outsize = bytes, rounded
 * up to whole bytes.  Lengths 0..31 (0 = symbol emits nothing). */
void oracle_vlc_encode(const uint32_t* indata, uint32_t num_elements,
                       uint32_t* outdata, uint32_t* outsize,
                       const uint32_t* codewords, const uint32_t* codewordlens);

/* Same arithmetic with 64-bit sizes (the reference's unsigned-int sizes stop
 * at 4 GiB of output; BASELINE configs 3-5 need more).  Returns total bits.
 * outdata may be NULL to only count. */
uint64_t oracle_vlc_encode64(const uint32_t* indata, uint64_t num_elements,
                             uint32_t* outdata,
                             const uint32_t* codewords, const uint32_t* codewordlens);

/* Per-segment bit counts and exclusive bit offsets: what
 * vlc_encode_kernel_sm64huff writes to outidx[] (vlc_kernel_sm64huff.cu:119-122)
 * followed by prescanArray (scan.cu:228) produces, widened to 64 bit and with
 * a ragged last segment allowed.  seg_words = 256 in the reference
 * (main_test_cu.cu:43).  counts/offsets may be NULL.  Returns total bits. */
uint64_t oracle_segment_offsets(const uint32_t* indata, uint64_t num_elements,
                                uint32_t seg_words,
                                const uint32_t* codewordlens,
                                uint32_t* counts, uint64_t* offsets);

/* Test-table generator restating generateCodewords (testdatagen.h:42-60):
 * lengths {1,2,3,4,4,5,6,7} repeating, codeword = 1 << (len-1). */
void oracle_generate_codewords(uint32_t* codewords, uint32_t* codewordlens,
                               uint32_t num_symbols);

/* Deterministic synthetic byte stream (the build's own
 * generator, no reference counterpart): draw j = first_draw .. uses
 * r = splitmix64(seed + j); two symbols per draw, u32 = r>>32 then r&0xffffffff;
 * symbol = min{ s : cdf32[s] > u32 } over a 256-entry cumulative table
 * (cdf32[255] is treated as 2^32).  Bytes are stored in file order.
 * n_bytes must be even. */
void oracle_gen_symbols(uint8_t* out, uint64_t n_bytes, uint64_t seed,
                        uint64_t first_draw, const uint32_t* cdf32);

/* 256-bin byte histogram of a buffer (helper for table construction in tests). */
void oracle_histogram(const uint8_t* data, uint64_t n_bytes, uint64_t* hist);

/* SURVEY.md section 8(f) #1, checker for the seam merge: append the first nbits bits of src (uint32 words, bit 31
 * first) to a stream that already holds dst_bits bits; dst words from the open word on must be zero or hold
 * earlier appended bits.  What pack2 (pack_kernels.cu:34-50) does word by word, restated bit-exactly with
 * 64-bit positions.  Returns the new bit count. */
uint64_t oracle_bit_append(uint32_t* dst, uint64_t dst_bits, const uint32_t* src, uint64_t nbits);

/* SURVEY.md section 8(f) #3, checker for the decoder (the reference has none): decode num_elements words
 * (4 symbols each, most significant byte first) from a stream produced by oracle_vlc_encode with a prefix-free
 * table, by matching the code of every symbol bit by bit.  Returns the number of bits consumed, or ~0 if no code
 * matches. */
uint64_t oracle_vlc_decode(const uint32_t* bits, uint64_t start_bit, uint32_t* outdata, uint64_t num_elements,
                           const uint32_t* codewords, const uint32_t* codewordlens);

/* FNV-1a 64-bit over bytes (fixture hashes). */
uint64_t oracle_fnv1a64(const void* data, uint64_t n_bytes);

/* Multi-threaded CPU baseline: T threads, each
 * encoding one contiguous chunk (multiple of seg_words words) independently
 * into its own scratch buffer; returns total bits; *seconds gets the wall time
 * of the encode region only.  scratch is split into `threads` equal slices; the
 * caller sizes it so one slice holds one chunk's output + 1 word. */
uint64_t oracle_vlc_encode_mt(const uint32_t* indata, uint64_t num_elements,
                              uint32_t* scratch, uint64_t scratch_words,
                              const uint32_t* codewords, const uint32_t* codewordlens,
                              int threads, double* seconds);

/* Whole-stream checker for inputs too large for one sequential pass (the 4 GiB BASELINE configs): the input is
 * cut into chunks of `chunk_segments` 256-word segments; every chunk is encoded on its own from bit 0
 * (oracle_vlc_encode64 -- the encoding is concatenative, cpuencode.cpp:21-41 carries no state but the bit cursor)
 * and compared, word for word, with the bits of `stream` that start at seg_offsets[first segment of the chunk];
 * the chunk's bit count and the offsets of all its segments are compared too.  `threads` host threads take the
 * chunks round robin.  Returns the number of chunks that differ (0 = identical); *first_bad_segment = first
 * segment of the first such chunk. */
uint64_t oracle_compare_stream_mt(const uint32_t* indata, uint64_t num_elements,
                                  const uint32_t* codewords, const uint32_t* codewordlens,
                                  const uint32_t* stream, uint64_t stream_words,
                                  const uint64_t* seg_offsets, uint64_t total_bits,
                                  uint64_t chunk_segments, int threads, uint64_t* first_bad_segment);

#ifdef __cplusplus
}
#endif
#endif
