/*
 * hvlc.h -- C ABI of libhvlc.so: MI355X-native (gfx950) variable-length / Huffman
 * ENCODER, the drop-in for the reference's GPU encode path
 *
 *     vlc_encode_kernel_sm64huff  (vlc_kernel_sm64huff.cu:37-160)
 *   + prescanArray                (scan.cu:228, scanLargeArray_kernel.cu:216-258)
 *   + pack2                       (pack_kernels.cu:19-52)
 *
 * as they are driven from runVLCTest (main_test_cu.cu:130-168), with
 * cpu_vlc_encode (cpuencode.cpp:12-46) as the bit-exact oracle.
 *
 * Plain C: pointers and sizes only, no torch/HIP types in any signature (a
 * hipStream_t is passed as void*; NULL = the default stream).  All device
 * pointers are ordinary hipMalloc'd (or torch-allocated) HBM addresses on the
 * context's device.  Every call returns an hvlc_status (no exit(), unlike the
 * reference's CUDA_SAFE_CALL, cutil.h:781-789).
 *
 * Data formats are the reference's:
 *   input    uint32 words; symbols are the bytes of each word, MOST significant
 *            byte first (cpuencode.cpp:28, vlc_kernel_sm64huff.cu:67)
 *   table    two uint32[256] arrays: codeword (right-aligned in its low `len`
 *            bits) and length in bits, 0..31; 0 = symbol emits nothing
 *            (load_data.h:40-47)
 *   stream   uint32 words, bit 31 first (cpuencode.cpp:34-36); word-for-word what
 *            cpu_vlc_encode writes, including the zeroed open word at the end
 *   offsets  exclusive BIT offset of every 256-word (1024-byte) segment, i.e.
 *            prescanArray's output d_cindex2 (main_test_cu.cu:164) widened to
 *            uint64 (the reference's uint32 overflows at 512 MiB of output)
 */
#ifndef HVLC_H_
#define HVLC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVLC_SEGMENT_WORDS 256u   /* the reference's block: 256 threads x 1 word (main_test_cu.cu:43) */
#define HVLC_MAX_CODE_LEN  31u    /* cpuencode.cpp:34: (1<<numbits) limits lengths to 31 */

typedef enum hvlc_status {
    HVLC_OK            =  0,
    HVLC_ERR_ARG       = -1,   /* NULL / misaligned pointer, bad size */
    HVLC_ERR_HIP       = -2,   /* a HIP runtime call failed (hvlc_last_hip_error) */
    HVLC_ERR_NOMEM     = -3,
    HVLC_ERR_TABLE     = -4,   /* a length > 31 or a codeword >= 2^len (precondition of cpuencode.cpp:35) */
    HVLC_ERR_CAPACITY  = -5,   /* stream would not fit out_capacity_words; nothing past it was written */
    HVLC_ERR_TIMEOUT   = -6,   /* an inter-workgroup wait gave up (bounded wait, safety net); output invalid */
    HVLC_ERR_NO_DEVICE = -7
} hvlc_status;

typedef struct hvlc_ctx hvlc_ctx;   /* workspace: replaces preallocBlockSums/deallocBlockSums (scan.cu:67,98) */

/* Create / destroy a context bound to `device`.  max_words sizes the workspace
 * (2 x 20 B per 32 KiB of input); it grows on demand if a later call is larger
 * (a re-allocation: synchronises the device, and HIP graphs captured on this
 * context before it must not be replayed afterwards -- size it up front). */
int hvlc_create(hvlc_ctx** ctx, int device, uint64_t max_words);
int hvlc_destroy(hvlc_ctx* ctx);

/* Words the caller must allocate for d_out_bits for `num_words` of input when no
 * table length exceeds max_len: floor(bits_max/32) + 1, as cpu_vlc_encode needs. */
uint64_t hvlc_out_capacity_words(uint64_t num_words, uint32_t max_len);
/* Number of 1024-byte segments (= entries of d_out_offsets) for num_words. */
uint64_t hvlc_num_segments(uint64_t num_words);

/*
 * encode(input, codeword_table, out_bits, out_offsets): the whole GPU path of
 * main_test_cu.cu:141-168 in one asynchronous call on `stream`.
 *
 *   d_in            [num_words] uint32, 16-byte aligned; any num_words >= 0 (the
 *                   reference requires a multiple of 4096 words, load_data.h:19-21)
 *   d_codewords     [256] uint32   } device copies of the table, as the reference
 *   d_codewordlens  [256] uint32   } passes them (main_test_cu.cu:142)
 *   d_out_bits      [out_capacity_words] uint32: receives floor(bits/32)+1 words,
 *                   identical to cpu_vlc_encode's outdata.  NO zero-fill is needed
 *                   (the reference memsets it, main_test_cu.cu:162): every word is
 *                   written exactly once with a plain store.
 *   d_out_offsets   [hvlc_num_segments] uint64 exclusive bit offsets, or NULL
 *   d_total_bits    uint64 on the device, or NULL (hvlc_finish returns it too)
 *
 * ONE kernel launch, nothing else (no allocation, no synchronisation, no
 * memset) once the context is large enough: capturable into a HIP graph.
 * Errors found on the device (table, capacity, timeout) are reported by
 * hvlc_finish.  Not re-entrant per context (one encode in flight per ctx).
 * Device requirement: an input of more than one tile (40 or 32 KiB) per resident
 * workgroup needs at least 64 workgroups resident at once (two per compute
 * unit: 32 CUs) -- fewer, e.g. a small compute partition, is HVLC_ERR_ARG.
 */
int hvlc_encode(hvlc_ctx* ctx,
                const uint32_t* d_in, uint64_t num_words,
                const uint32_t* d_codewords, const uint32_t* d_codewordlens,
                uint32_t* d_out_bits, uint64_t out_capacity_words,
                uint64_t* d_out_offsets, uint64_t* d_total_bits,
                void* stream);

/* Wait for the last hvlc_encode (or hvlc_exclusive_scan) of this context that was LAUNCHED on `stream`,
 * return its total (bits / sum) and its device-side status
 * (HVLC_OK / _TABLE / _CAPACITY / _TIMEOUT).  After a _TIMEOUT the outputs of that call are
 * incomplete and must not be used.  An empty call (num_words == 0, n == 0) counts: it reports
 * HVLC_OK and a total of 0, never the result of an earlier call; a call that failed before its
 * launch (an error return) leaves what hvlc_finish reports on unchanged. */
int hvlc_finish(hvlc_ctx* ctx, void* stream, uint64_t* total_bits);

/*
 * Same signature and meaning as the reference's cpu_vlc_encode (cpuencode.h:4-7,
 * called at main_test_cu.cu:122) but computed on the GPU: HOST buffers in and
 * out, synchronous.  outdata must hold floor(bits/32)+1 words; *outsize = bytes
 * rounded up (cpuencode.cpp:44-45).  Returns an hvlc_status (the reference
 * returns void).  Uses a process-wide context on the current device.
 */
int gpu_vlc_encode(unsigned int* indata, unsigned int num_elements,
                   unsigned int* outdata, unsigned int* outsize,
                   unsigned int* codewords, unsigned int* codewordlens);

/* ---- the stages of the reference pipeline as separate entry points --------- */

/* Per-segment bit counts: what vlc_encode_kernel_sm64huff stores to outidx[]
 * (vlc_kernel_sm64huff.cu:119-122; d_cindex at main_test_cu.cu:146).
 * d_counts [hvlc_num_segments] uint32. */
int hvlc_segment_bits(hvlc_ctx* ctx, const uint32_t* d_in, uint64_t num_words,
                      const uint32_t* d_codewordlens, uint32_t* d_counts, void* stream);

/* Exclusive prefix sum uint32[n] -> uint64[n] (+ optional total): replaces
 * preallocBlockSums + prescanArray + deallocBlockSums (main_test_cu.cu:160-168)
 * with a single-launch wavefront-hierarchical scan.  Asynchronous like hvlc_encode:
 * a look-back that gives up (bounded wait) is reported by hvlc_finish as
 * HVLC_ERR_TIMEOUT.  Unlike hvlc_encode the scan numbers its tiles statically: its
 * grid (what the occupancy query reports as resident, at most four workgroups of
 * 256 threads per compute unit) must be resident at once, i.e. the device should
 * not be shared with a long-running kernel while it runs. */
int hvlc_exclusive_scan(hvlc_ctx* ctx, const uint32_t* d_counts, uint64_t n,
                        uint64_t* d_offsets, uint64_t* d_total, void* stream);

/* Multi-GPU stitch: add a shard's base bit offset (the exclusive prefix of the
 * per-shard totals exchanged over RCCL) to its local segment offsets. */
int hvlc_rebase_offsets(hvlc_ctx* ctx, uint64_t* d_offsets, uint64_t n,
                        uint64_t base_bits, void* stream);

/* Same, with the base computed on the device: base = sum of d_shard_totals[0..rank),
 * where d_shard_totals is the all-gathered array of per-shard total bit counts (one
 * uint64 per rank, e.g. the output of ncclAllGather over xGMI).  No host round trip. */
int hvlc_rebase_offsets_dev(hvlc_ctx* ctx, uint64_t* d_offsets, uint64_t n,
                            const uint64_t* d_shard_totals, int rank, void* stream);

/* ---- utilities used by the harness / benchmarks ---------------------------- */

/* Kernel timing with HIP events on the launch stream: after hvlc_profile_begin every
 * hvlc_encode brackets its k_encode launch (the dominant kernel) with an event pair, up
 * to max_launches pairs.  hvlc_profile_collect synchronises those events, writes each
 * launch's duration in milliseconds to ms[0..*count) and stops profiling. */
int hvlc_profile_begin(hvlc_ctx* ctx, int max_launches);
int hvlc_profile_collect(hvlc_ctx* ctx, float* ms, int* count);

/* Measured memory ceilings for the roofline: times a kernel that does nothing but stream
 * n_bytes from d_in (4 x dwordx4 non-temporal loads per lane per 4 KiB, the encoder's load shape) and, for
 * store_shape != 0, write out_words_per_4k words per 4 KiB read to a contiguous stream in d_out (the encoder's
 * traffic mix: 1024 * bits_per_symbol / 32 words per 4 KiB).  store_shape 0 = read only, 1 = one dword per lane
 * per store instruction, 2 = dwordx4 per lane (out_words_per_4k % 4 == 0), 3 / 4 = the same two with non-temporal
 * stores, 5 / 6 = dwordx4 non-temporal stores in bursts: a wave writes the output of 10 steps (a tile's worth: the
 * shape of the encoder's store wave) / of 32 steps as one contiguous stretch (d_out must hold a region per wave and
 * burst: about twice the stream), 7 = a plain 1:1 copy of the input (out_bytes >= n_bytes; out_words_per_4k
 * ignored): the known-good reference of a read/write mix, 8 / 9 / 10 = the encoder's division of labour in the
 * geometry the memory system likes best: one workgroup of 15 reader waves and a writer wave per compute unit, a
 * reader with one / two / three 4 KiB steps in flight, the writer storing four rounds' output as one stretch, 11 = the same in the encoder's own geometry (two workgroups of 10
 * readers with one step in flight and a writer per compute unit) (no wave has loads queued behind its own stores; d_out: a region
 * per workgroup and burst, about the stream's size plus a burst per workgroup).  The ceiling of a
 * workload's mix is the BEST of the shapes.
 * Synchronous on the default stream; runs `iters` launches and returns the best and the median duration in ms. */
int hvlc_probe_bandwidth(hvlc_ctx* ctx, const void* d_in, uint64_t n_bytes, void* d_out, uint64_t out_bytes,
                         uint32_t out_words_per_4k, int store_shape, int iters, float* ms_best, float* ms_median);

/* Deterministic synthetic symbol stream on the device:
 * r = splitmix64(seed + first_draw + j), two symbols per draw through a 256-entry
 * 32-bit cumulative table d_cdf32.  n_bytes even, d_out 4-byte aligned. */
int hvlc_gen_symbols(hvlc_ctx* ctx, uint8_t* d_out, uint64_t n_bytes, uint64_t seed,
                     uint64_t first_draw, const uint32_t* d_cdf32, void* stream);

/* 256-bin byte histogram on the device (input to code-table construction; the
 * reference's histo_kernel, hist.cu:34-52, without its windowing bug :98-102).
 * d_hist [256] uint64, overwritten. */
int hvlc_histogram(hvlc_ctx* ctx, const uint8_t* d_data, uint64_t n_bytes,
                   uint64_t* d_hist, void* stream);

/* SURVEY.md section 8(f) #2 -- code table for a byte histogram (HOST function, no device needed; hvlc_histogram
 * produces the histogram on the device).  Replaces the reference's host-side table construction (huffTree.h:55-94,
 * load_data.h:40-47): Huffman code lengths (symbols with count 0 get length 0 = emit nothing); when the Huffman tree is
 * deeper than max_len (8..31) the OPTIMAL code with lengths <= max_len (package-merge); canonical codewords
 * right-aligned in `length` bits -- the table format hvlc_encode takes.  A Huffman code has the same total cost as the
 * reference's table whenever the reference's own tree fits max_len; the codewords themselves differ.
 * HVLC_ERR_ARG if the counts do not sum below 2^64. */
int hvlc_build_table(const uint64_t* hist, uint32_t max_len, uint32_t* codewords, uint32_t* codewordlens);

/* SURVEY.md section 8(f) #1 -- seam merge / contiguous stream materialisation.  Copies the `nbits` bits of a
 * stream that starts at bit 0 of d_src_bits (a shard's output of hvlc_encode; may be P2P-mapped memory of a peer
 * GPU) to bit position dst_bit_offset of d_dst_bits: what pack2 (pack_kernels.cu:19-52) does for 256-word
 * blocks, for whole shards.  Interior destination words are overwritten; the first and the last one, which the
 * neighbouring shards share, are merged with an atomic OR and must be ZERO before the first merge that touches
 * them (zero-fill the seam words or all of d_dst_bits once).  Shards may be merged in any order, concurrently --
 * also by several GPUs into one destination: d_dst_bits may be peer-mapped memory of another GPU (the two seam
 * words are merged with SYSTEM-scope atomics, so that a seam word OR-ed from two devices is resolved at its home).
 * Merging every shard at its base (hvlc_rebase_offsets[_dev]'s base) into a zeroed buffer of
 * floor(total/32)+1 words reproduces cpu_vlc_encode's stream of the whole input. */
int hvlc_merge_shard(hvlc_ctx* ctx, const uint32_t* d_src_bits, uint64_t nbits, uint32_t* d_dst_bits,
                     uint64_t dst_capacity_words, uint64_t dst_bit_offset, void* stream);

/* SURVEY.md section 8(f) #3 -- decoder (the reference has none; this one exists to prove round trips).
 * Inverse of hvlc_encode for PREFIX-FREE tables in which every symbol that occurs has length >= 1: one thread per
 * 1024-byte segment, started at d_offsets[segment].  Writes num_words words (4 symbols each, most significant byte
 * first).  An invalid stream or table yields wrong symbols, never an out-of-bounds access. */
int hvlc_decode(hvlc_ctx* ctx, const uint32_t* d_bits, uint64_t total_bits, const uint64_t* d_offsets,
                const uint32_t* d_codewords, const uint32_t* d_codewordlens, uint32_t* d_out_words,
                uint64_t num_words, void* stream);

/* TEST HOOKS (used by tests/ only; a production caller never calls this).  They live in the context, are
 * set explicitly and cost the launch path nothing (no environment variables are read by hvlc_encode):
 *   HVLC_HOOK_STALL_TILE  value = a tile number whose bit count is never published, so that every later tile
 *                         waits until the bounded waits give up with HVLC_ERR_TIMEOUT; value < 0 = off
 *   HVLC_HOOK_FORCE_GRID  value = number of workgroups to launch instead of what is resident (a grid that is
 *                         not co-resident must still finish with the exact stream); 0 = off */
enum { HVLC_HOOK_STALL_TILE = 1, HVLC_HOOK_FORCE_GRID = 2 };
int hvlc_test_hook(hvlc_ctx* ctx, int which, int64_t value);

/* The encoder has two workgroup geometries: ten data waves with 40 KiB tiles, and eight data waves with larger
 * staging rings and 32 KiB tiles for dense output (more than ~3.4 bits per symbol: a ring of the first geometry holds
 * only three parts then; the second is 9 % faster at 4 bits per symbol and 9 % slower at 2.2).  A context starts with
 * the first and follows the density of its last FINISHED encode (hvlc_finish after an encode of at least 4 MiB: up at
 * 3.4 bits per symbol, back at 3.2); a caller that never calls hvlc_finish stays with the first.  The output is the
 * same bit for bit either way.  HVLC_GEOMETRY=10|8 in the environment of hvlc_create fixes it (tests).
 * Returns the data waves per workgroup of the last encode launched and of the next one. */
int hvlc_get_geometry(hvlc_ctx* ctx, int* data_waves_last, int* data_waves_next);

/* Introspection: launch geometry chosen for this device (grid, blocks per CU,
 * CUs) and the last HIP error code seen by this context. */
int hvlc_get_launch_info(hvlc_ctx* ctx, int* grid_blocks, int* blocks_per_cu, int* compute_units);
int hvlc_last_hip_error(hvlc_ctx* ctx);
/* The shader clock the GPU held during the last hvlc_encode that hvlc_finish has reported on: workgroup 0 counts its
 * own lifetime in shader clocks (s_memtime) and in ticks of the constant 100 MHz counter (s_memrealtime); clock in GHz
 * = shader_clocks / realtime_ticks / 10.  MI355X lowers its clock under load, so a launch's time is set by the clock it
 * gets as much as by its instruction count.  Zeros before the first finished encode. */
int hvlc_get_clock_info(hvlc_ctx* ctx, uint64_t* shader_clocks, uint64_t* realtime_ticks);
const char* hvlc_status_string(int status);
const char* hvlc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HVLC_H_ */
