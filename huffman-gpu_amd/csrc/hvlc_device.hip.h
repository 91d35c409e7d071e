// hvlc_device.hip.h -- gfx950 (MI355X, wave64) device code of libhvlc.
//
// One fused, single-pass kernel -- ONE launch per hvlc_encode call -- replaces the reference's three GPU stages
// (vlc_encode_kernel_sm64huff, vlc_kernel_sm64huff.cu:37-160; prescanArray,
// scan.cu:228; pack2, pack_kernels.cu:19-52): the input is read from HBM once and
// every word of the final contiguous bit stream is written once, with a plain
// store, at its final position.  No padded intermediate copy, no memset of the
// output, no global atomics on the stream.
//
// Decomposition (all names follow the reference's domain):
//   segment  256 input words = 1024 symbols  = ONE WAVE-ROW (64 lanes x 16 bytes);
//            the reference's block (main_test_cu.cu:43); unit of out_offsets
//   part     4 segments = 4 KiB              = what ONE data wave encodes per tile
//   tile     40 segments = 40 KiB            = the 10 parts of a workgroup's 10 data waves;
//            data wave w owns the tile's segments 4w..4w+3; unit of the look-back
//   group    64 tiles = 2.5 MiB              = unit of the second scan level
//
// Workgroup = 10 DATA waves + 1 SCOUT wave + 1 STORE wave (three waves per SIMD), two workgroups per CU.
// The data waves never wait for each other (no barrier after the LDS set-up) and never touch the output.
// A data wave, per tile: (1) coalesced non-temporal dwordx4 loads (next tile prefetched); (2) per-lane LDS
// table lookups (one SDWA op per address) and 64-bit concatenation of 16 symbols; (3) wave64 DPP
// prefix scan of the bit lengths per segment; (4) it ARRIVES at the tile (one LDS atomic add of
// its bit count): the last wave to arrive PUBLISHES the tile's bit count (one self-validating
// word, relaxed agent-scope store); (5) "pabio": each lane ORs its <=64 bits into the wave's
// OWN LDS staging ring with 3 ds_or_b32 at PART-LOCAL bit positions (so packing waits for
// nothing).  Store wave: (7) several tiles later, once the scout has seen the tile's global bit offset,
// it walks the tile's stretch of the stream in 256-byte blocks aligned by ADDRESS, funnel-shifts the ring
// words of the part a word belongs to (v_alignbit_b32) and stores one block per instruction -- the block in
// which one part ends and the next begins is ONE instruction fed from two rings -- so that a tile leaves as one
// burst of whole lines; it writes the segment offsets (uint64), zeroes the ring words and hands them back.
// Scout wave: draws the workgroup's tiles from ticket counters, and (6) polls ONE 16-byte record
// per tile in flight (global offset + the predecessor's <=31 carry bits).  The offsets come
// from a two-level scan: the scout of the LAST tile of each 64-tile group reads the group's 64
// bit counts, publishes the group aggregate, walks the group granules down to the nearest
// inclusive prefix and writes the offsets of all 64 tiles into their records.  The scout exists
// because VMEM retires in order per wave: a data wave polling other workgroups' words would
// drain its own prefetch and stores first.
//
// Forward progress: tiles are drawn only by RUNNING workgroups (dynamic tickets, 64 classes so
// that a workgroup never holds two tiles of one group), every wave works through the workgroup's
// tiles in increasing order, and a tile depends only on tiles with lower numbers plus its
// group's last tile -- so the encoder finishes whatever part of the grid is resident.  Every
// wait is still bounded (s_memrealtime) and ends in HVLC_ERR_TIMEOUT instead of a hang.
// Inter-workgroup data are single naturally aligned 4- or 8-byte words carrying their own valid
// bit, written once with ONE store and read by ONE lane with ONE load of the same width
// (relaxed agent-scope, sc1) -- the form the CDNA4 guide lists as needing no release/acquire
// fence -- and zero at the start of each launch (by the launch before it: EncodeParams).
// This header is included TWICE by hvlc_api.hip (no include guard): as namespace hvlc with ten data waves per encoder
// workgroup, and as namespace hvlc8 with eight and larger rings (HVLC_NS, HVLC_DATA_WAVES, HVLC_WAVE_RING_WORDS) -- the
// geometry for streams of more than ~3.4 bits per symbol, whose parts a 1752-word ring holds only three of.

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef HVLC_NS
#define HVLC_NS hvlc
#endif
namespace HVLC_NS {

constexpr int      kThreads     = 256;                     // block size of the small kernels
#ifndef HVLC_DATA_WAVES
#define HVLC_DATA_WAVES 10
#endif
constexpr int      kDataWaves   = HVLC_DATA_WAVES;         // data waves per encoder workgroup (+1 scout wave, +1 store wave)
constexpr int      kDataThreads = 64 * kDataWaves;
constexpr int      kEncThreads  = kDataThreads + 128;      // k_encode: 10+1+1 waves = three waves per SIMD whatever the placement
#ifndef HVLC_ROWS
#define HVLC_ROWS 4
#endif
constexpr int      kRows        = HVLC_ROWS;               // 16-byte rows per lane per tile
constexpr int      kTileWords   = kDataThreads * kRows * 4;   // 10240 words = 40 KiB
constexpr int      kSegWords    = 256;
constexpr int      kSegsPerTile = kTileWords / kSegWords;  // 40: wave w owns segments 4w..4w+3
static_assert(kSegsPerTile <= 64, "the store wave writes a tile's segment offsets with one lane per segment");
#ifndef HVLC_GROUP_TILES
#define HVLC_GROUP_TILES 64
#endif
constexpr int      kGroupTiles  = HVLC_GROUP_TILES;       // <= 64 (one lane per tile in the leader's scan) and <= kTicketClasses
constexpr int      kScanTile    = 4096;                    // elements per workgroup pass of k_exclusive_scan
#ifndef HVLC_TICKET_CLASSES
#define HVLC_TICKET_CLASSES 64
#endif
constexpr int      kTicketClasses = HVLC_TICKET_CLASSES;    // ticket counters (workgroup b draws from counter b % kTicketClasses)
constexpr int      kTicketStride  = 32;                    // uint32 words between counters (one 128-byte line each)
constexpr uint64_t kValid       = 1ull << 63;
constexpr uint64_t kStAgg       = 1ull << 62;              // group granule: aggregate of the group
constexpr uint64_t kStPre       = 2ull << 62;              // group granule: inclusive prefix through the group
constexpr uint64_t kValMask     = (1ull << 62) - 1;

// What a tile waits for, in one 16-byte record = two 8-byte words, each written by ONE store and polled by ONE
// lane with ONE 8-byte load:
//   P    = kValid | global exclusive bit offset of the tile, written by the LAST tile of the tile's
//          group (the group leader) for all tiles of the group at once;
//   tail = kValid32 | the last 31 bits of the stream before the tile, written by the tile before it
//          (record 0: by the scout that draws tile 0); `pad` is never written.
struct alignas(16) TileRec { uint64_t P; uint32_t tail; uint32_t pad; };
constexpr uint32_t kValid32 = 1u << 31;

enum : uint32_t { kFlagTable = 1u, kFlagCapacity = 2u, kFlagTimeout = 4u, kFlagLayout = 8u };

// Workspace of a context, one allocation: [control block 64 B][granule area, half 0][granule area, half 1].
// ONE kernel launch per hvlc_encode: consecutive calls alternate between the two halves, and every launch, while it works
// in its own half, zeroes what the call before it used of the OTHER half (each workgroup a slice, at its start) -- so
// every polled word of a call starts as zero without a reset kernel (and its launch gap) in front.  Which half a launch
// uses is decided on the device from the context's call counter, so a captured HIP graph can be replayed.
// Control block, as uint32 words:
//   [0] [1]  status flags of even / odd calls (atomic RMWs only)      [2..3]  uint64 total bits of the last encode
//   [4]      calls completed (incremented by the last wave to leave)  [5]     waves that have left the running launch
//   [6] [7]  bytes the last call in half 0 / 1 used (what the next launch has to zero)
//   [8]      status flags of the last hvlc_exclusive_scan              [10..11] uint64 its total
//   [12..15] lifetime of the last encode's workgroup 0 in shader clocks and in 100 MHz ticks (two uint64)
// The kernels get ONE base pointer per area and byte offsets instead of eight pointers (fewer live scalar registers in
// k_encode: with eight, the compiler spilled pointers to VGPR lanes and reloaded them in front of the polls).
constexpr uint32_t kWsCtlBytes   = 64;
constexpr uint32_t kCtlResult64  = 1;              // uint64 index
constexpr uint32_t kCtlCalls     = 4, kCtlExits = 5, kCtlUsed = 6, kCtlScanStatus = 8;
constexpr uint32_t kCtlScanResult64 = 5;           // uint64 index
constexpr uint32_t kCtlClock     = 12;             // [12..13] shader clocks, [14..15] 100 MHz ticks of workgroup 0's scout
constexpr uint32_t kWsOffBits32  = 64 * 32 * 4;    // granule area: tickets [64 x 128 B] | bits32 | recs | group_desc

struct EncodeParams {
    const uint4*    in;          // input words, 16-byte aligned
    uint64_t        num_words;
    uint64_t        num_tiles;
    uint64_t        num_segs;
    const uint32_t* cw;          // the caller's table (device): codewords, lengths (encoder only)
    const uint32_t* cl;
    uint32_t*       out;         // final stream
    uint64_t        out_cap;     // words
    uint64_t*       offsets;     // per-segment exclusive bit offsets (nullable)
    uint64_t*       total_bits;  // user's device word (nullable)
    uint8_t*        ws;          // the context's control block (see above)
    uint8_t*        gran0;       // encoder: granule area, half 0 (half 1 follows half_bytes behind); scan: the scan's own area
    uint32_t        half_bytes;
    uint32_t        used_bytes;  // encoder: bytes of its half this call uses (layout of num_tiles tiles)
    uint32_t        off_recs;    // encoder: byte offset of recs[] in the granule area
    uint32_t        off_groups;  // byte offset of group_desc[] in the granule area (encoder and scan)
    uint32_t        half;        // encoder: set by the KERNEL (call counter & 1); scan: 0
    uint32_t        scan;        // 1: k_exclusive_scan's status / result words
    uint64_t        timeout_ticks;   // s_memrealtime ticks (100 MHz)
    uint32_t        stall_tile;      // TEST HOOK: this tile's bit count is never published (~0: none); see hvlc_api.hip
    uint32_t        dbg;             // ablation flags, honoured only in -DHVLC_EXPERIMENT builds (tools/)
    uint64_t*       dbg_tl;          // per-tile timeline [tiles][8] (100 MHz ticks), -DHVLC_EXPERIMENT builds only
    uint64_t*       dbg_ws;          // per-wave time accounting [grid][16 waves][16] (shader clocks), -DHVLC_EXPERIMENT builds only

    __device__ __forceinline__ uint32_t* ctl() const { return (uint32_t*)ws; }
    __device__ __forceinline__ uint32_t* status() const { return (uint32_t*)ws + (scan ? kCtlScanStatus : half); }   // flag word (atomic RMWs only)
    __device__ __forceinline__ uint64_t* result() const { return (uint64_t*)ws + (scan ? kCtlScanResult64 : kCtlResult64); }
    __device__ __forceinline__ uint8_t*  gran() const { return gran0 + (half ? half_bytes : 0u); }
    // encoder: kTicketClasses counters, kTicketStride words apart
    __device__ __forceinline__ uint32_t* tickets() const { return (uint32_t*)gran(); }
    // encoder: word per tile, kValid32 | bits of the tile (group g = words 64g..64g+63)
    __device__ __forceinline__ uint32_t* bits32() const { return (uint32_t*)(gran() + kWsOffBits32); }
    // encoder: record per tile (+1)
    __device__ __forceinline__ struct TileRec* recs() const { return (struct TileRec*)(gran() + off_recs); }
    // granule per group: status | value (encoder and scan)
    __device__ __forceinline__ uint64_t* group_desc() const { return (uint64_t*)(gran() + off_groups); }
    // scan kernel only: granule per scan tile, kValid | sum
    __device__ __forceinline__ uint64_t* tile_bits() const { return (uint64_t*)gran(); }
};

// Ablation switches for tools/ablate.py.  In the product build HVLC_DBG() is a compile-time false.
enum : uint32_t { kDbgNoLookback = 1u, kDbgNoStore = 2u, kDbgNoPack = 4u, kDbgNoOffsets = 16u };
#ifdef HVLC_EXPERIMENT
#define HVLC_DBG(p, bit) (((p).dbg & (bit)) != 0u)
#define TL(p, tile, k, cond) do { if ((p).dbg_tl && (cond)) (p).dbg_tl[(uint64_t)(tile) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TLV(p, tile, k, v, cond) do { if ((p).dbg_tl && (cond)) (p).dbg_tl[(uint64_t)(tile) * 8 + (k)] = (v); } while (0)
// stamps of workgroup 0's prologue and first tile, in the row behind the last tile's (tools/one_tile_timeline.py)
#define TLX(p, k, cond) do { if ((p).dbg_tl && blockIdx.x == 0 && (cond)) (p).dbg_tl[(p).num_tiles * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-wave time accounting of a data wave (tools/wave_stats.py): shader clocks per phase and per cause of waiting
enum { kWsEncode = 0, kWsStore, kWsArrive, kWsPack, kWsWaitTicket, kWsWaitDrain, kWsWaitSlot, kWsWaitRing, kWsWaitSlow, kWsWaitTail,
       kWsTotal, kWsTiles, kWsNWaits };
struct WaveStats {
    uint64_t t[16]; uint64_t last;
    __device__ __forceinline__ void init() { for (int i = 0; i < 16; ++i) t[i] = 0; last = __builtin_amdgcn_s_memtime(); t[kWsTotal] = last; }
    __device__ __forceinline__ void lap(int i) { const uint64_t now = __builtin_amdgcn_s_memtime(); t[i] += now - last; last = now; }
    __device__ __forceinline__ void dump(uint64_t* buf, int wave) {
        t[kWsTotal] = __builtin_amdgcn_s_memtime() - t[kWsTotal];
        if (buf) for (int i = 0; i < 16; ++i) buf[((uint64_t)blockIdx.x * 16 + wave) * 16 + i] = t[i];
    }
};
#define WS_DECL WaveStats ws; ws.init()
#define WS_LAP(i) ws.lap(i)
#define WS_COUNT(i) (ws.t[i] += 1)
#define WS_DUMP(p, wave, lane) do { if ((lane) == 0) ws.dump((p).dbg_ws, wave); } while (0)
#else
#define TL(p, tile, k, cond)
#define TLV(p, tile, k, v, cond)
#define TLX(p, k, cond)
#ifdef HVLC_ABL_MASK
#define HVLC_DBG(p, bit) (((HVLC_ABL_MASK) & (bit)) != 0u)      // (tools/: a compile-time ablation without the instrumentation)
#else
#define HVLC_DBG(p, bit) false
#endif
#define WS_DECL
#define WS_LAP(i)
#define WS_COUNT(i)
#define WS_DUMP(p, wave, lane)
#endif

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Inter-workgroup words: written once per launch with ONE relaxed agent-scope (sc1, write-through) store,
// zero at the start of every launch (EncodeParams), each carrying its own valid bit ("the data is the flag").
__device__ __forceinline__ uint64_t ld_granule(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(uint64_t* p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A value every lane of the wave read from the same LDS / kernarg word: tell the compiler it is wave-uniform
// (it then lives in an SGPR and feeds scalar address arithmetic instead of per-lane 64-bit VGPR math).
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x)
{
    return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | (uint64_t)uni((uint32_t)x);
}

// The lane number, recomputed where it is called (two VALU instructions) instead of kept in a register from the prologue --
// and, being volatile, not an invariant the compiler could hoist out of a loop together with everything computed from it:
// cold paths take their lane number from here, or their per-lane addresses and masks end up in (spilled) registers of
// the hot loop around them.
__device__ __forceinline__ int lane_id_fresh()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ void st_word(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v)
{
    (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t x)
{
    // lanes without a valid source (or in a masked-off row) receive `old` = 0
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}

// Inclusive prefix sum over the 64 lanes of a wave, DPP only (no LDS):
// row_shr:1,2,4,8 inside each row of 16, then row_bcast:15 / row_bcast:31 across rows.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    x = dpp_add<0x111, 0xf>(x);
    x = dpp_add<0x112, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 -> rows 1,3
    x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 -> rows 2,3
    return x;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)x, o);
        const uint32_t hi = __shfl_xor((uint32_t)(x >> 32), o);
        x += ((uint64_t)hi << 32) | lo;
    }
    return x;
}

// Bounded wait: ~every 1024 polls look at the 100 MHz real-time counter and at the global
// status word; give up (and make everybody else give up) after timeout_ticks.
#ifndef HVLC_WAIT_SLEEP
#define HVLC_WAIT_SLEEP 64       // x64 clocks per look of a WAITING wave at its LDS words (s_wakeup ends the sleep early).
                                 // (16 or 8, or a few short sleeps before the long ones against wake-ups that found nobody
                                 // asleep yet: 0.4-3 % SLOWER -- looks cost LDS time and issue slots: tools/ab.py, round 3)
#endif
struct Spin {
    uint64_t t0 = 0;
    uint32_t n  = 0;
    __device__ __forceinline__ bool wait(const EncodeParams& p)
    {
        __builtin_amdgcn_s_sleep(HVLC_WAIT_SLEEP);
        if ((++n & 1023u) == 0u) {
            const uint64_t now = __builtin_amdgcn_s_memrealtime();
            const uint32_t st  = atomicOr(p.status(), 0u);     // RMW: coherent with the other workgroups' atomicOr
            if (t0 == 0) t0 = now;
            if ((st & kFlagTimeout) || now - t0 > p.timeout_ticks) {
                atomicOr(p.status(), kFlagTimeout);
                return false;
            }
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// bit I/O on the LDS staging buffer ("pabio", pabio_kernels_v2.cu:17-33 generalised to a
// word-crossing <=64-bit string as vlc_kernel_sm64huff.cu:130-154 does).
// `st` points at the staging word that holds local bit 0; bit q of the tile-local stream lives
// in word q/32 at bit 31-(q%32).  All shifts are guarded for AMD's 5-bit shift counts
// (v_alignbit_b32 takes its count modulo 32, and count 0 returns the low operand).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_bits64(uint32_t* st, uint32_t q, uint64_t acc, uint32_t n)
{
    if (n == 0u) return;
    const uint64_t val = acc << ((64u - n) & 63u);            // left-align the n bits
    const uint32_t hi = (uint32_t)(val >> 32), lo = (uint32_t)val;
    const uint32_t sh = q & 31u;
    uint32_t* w = st + (q >> 5);
    lds_or(w, hi >> sh);
    if (sh + n > 32u) lds_or(w + 1, __builtin_amdgcn_alignbit(hi, lo, sh));
    if (sh + n > 64u) lds_or(w + 2, __builtin_amdgcn_alignbit(lo, 0u, sh));
}

// Last min(e,31) bits of the local stream that ends at local bit e (right-aligned).
__device__ __forceinline__ uint32_t extract_tail(const uint32_t* st, uint32_t e)
{
    const uint32_t wi = e >> 5, sh = e & 31u;
    const uint32_t lo = st[wi];
    const uint32_t hi = wi ? st[wi - 1] : 0u;
    const uint32_t x  = sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;   // the 32 bits ending at e
    return e >= 31u ? (x & 0x7fffffffu) : (x & ((1u << e) - 1u));
}

// ---------------------------------------------------------------------------------------------
// symbol -> codeword lookup and concatenation of one lane-row (16 bytes = 4 words = 16 symbols,
// most significant byte of each word first: cpuencode.cpp:28, vlc_kernel_sm64huff.cu:67).
// acc is only meaningful while tot <= 64 (the reference's cw64 limit is per WORD,
// vlc_kernel_sm64huff.cu:48,70; here it is per 16 symbols and checked, with a pairwise cold path).
// ---------------------------------------------------------------------------------------------
//
// The table sits in LDS at slot(s) = s ^ (s >> 4): symbol sets with a power-of-two stride (e.g. the
// 16 symbols 16j+3 of BASELINE config 3) would otherwise hit two of the 32 ds_read_b64 bank pairs
// and serialise 8-way.  hash4() applies the permutation to the four symbols of a word at once.
__device__ __forceinline__ uint32_t lut_slot(uint32_t s) { return s ^ (s >> 4); }
__device__ __forceinline__ uint32_t hash4(uint32_t w) { return w ^ ((w >> 4) & 0x0f0f0f0fu); }
// k_encode decides per table whether the permutation is needed at all (its prologue): where the
// identity layout conflicts no more than the permuted one, the hot loop saves the two instructions per word.
// Cold paths select at run time:
__device__ __forceinline__ uint32_t hash4_if(uint32_t w, bool permuted) { return permuted ? hash4(w) : w; }

// LDS byte offset (slot * 8) of byte B of the hashed word h in ONE VALU op: SDWA selects the byte,
// the shift scales it to the 8-byte table entries.
template <int B>
__device__ __forceinline__ uint32_t slot_offset(uint32_t h)
{
    uint32_t off;
    const uint32_t three = 3u;
    if (B == 3)      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "s"(three), "v"(h));
    else if (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "s"(three), "v"(h));
    else if (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "s"(three), "v"(h));
    else             asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "s"(three), "v"(h));
    return off;
}
// The table is the FIRST member of the workgroup's only LDS object, i.e. it sits at LDS address 0
// (checked once per launch in k_encode): the slot offset is the whole ds_read_b64 address.
__device__ __forceinline__ uint2 lut_at(const uint2* /*lut*/, uint32_t byte_off)
{
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;
    const u32x2 e = *reinterpret_cast<lds_u32x2*>((uintptr_t)byte_off);
    return make_uint2(e.x, e.y);
}
// One lane-row of a FULL tile: 16 lookups, a 16-step shift-or chain for the bits and the lengths summed three at a
// time (v_add3_u32): 3.5 vector instructions per symbol, or 3 when the table needs no slot permutation (HASH false).
template <bool HASH>
__device__ __forceinline__ void encode16(const uint4& v, const uint2* lut, uint64_t& acc, uint32_t& tot)
{
    const uint32_t w[4] = { HASH ? hash4(v.x) : v.x, HASH ? hash4(v.y) : v.y, HASH ? hash4(v.z) : v.z, HASH ? hash4(v.w) : v.w };
    acc = 0; tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h = w[k];
        const uint2 e3 = lut_at(lut, slot_offset<3>(h));
        const uint2 e2 = lut_at(lut, slot_offset<2>(h));
        const uint2 e1 = lut_at(lut, slot_offset<1>(h));
        const uint2 e0 = lut_at(lut, slot_offset<0>(h));
        acc = (acc << e3.y) | e3.x;
        acc = (acc << e2.y) | e2.x;
        acc = (acc << e1.y) | e1.x;
        acc = (acc << e0.y) | e0.x;
        tot = (tot + e3.y) + (e2.y + e1.y + e0.y);
    }
}
// The same when the four codewords of every input word fit 32 bits together (the common case: 8 bits per symbol and
// less): each word's codewords are joined by three 32-bit shift-or instructions (v_lshl_or_b32), only the four word
// strings of the lane-row by 64-bit shifts -- 2.9 vector instructions per symbol instead of 3.5.  `wmax` returns the
// longest word string: above 32 the row's `acc` is wrong and the caller encodes the part again with encode16.
template <bool HASH>
__device__ __forceinline__ void encode16_w32(const uint4& v, const uint2* lut, uint64_t& acc, uint32_t& tot, uint32_t& wmax)
{
    const uint32_t w[4] = { HASH ? hash4(v.x) : v.x, HASH ? hash4(v.y) : v.y, HASH ? hash4(v.z) : v.z, HASH ? hash4(v.w) : v.w };
    uint32_t cw[4], wl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h = w[k];
        const uint2 e3 = lut_at(lut, slot_offset<3>(h));
        const uint2 e2 = lut_at(lut, slot_offset<2>(h));
        const uint2 e1 = lut_at(lut, slot_offset<1>(h));
        const uint2 e0 = lut_at(lut, slot_offset<0>(h));
        uint32_t c = (e3.x << e2.y) | e2.x;                  // (lengths <= 31: the shifts are defined; v_lshl_or_b32)
        c = (c << e1.y) | e1.x;
        c = (c << e0.y) | e0.x;
        cw[k] = c;
        wl[k] = (e3.y + e2.y + e1.y) + e0.y;
    }
    acc = cw[0];
    acc = (acc << wl[1]) | cw[1];                             // (a word string of 32 bits: the shift is a 64-bit one)
    acc = (acc << wl[2]) | cw[2];
    acc = (acc << wl[3]) | cw[3];
    tot = (wl[0] + wl[1] + wl[2]) + wl[3];
    wmax = max(max(wl[0], wl[1]), max(wl[2], wl[3]));
}

// The hot-path form: the <=64 bits of `acc` (right-aligned, nothing above them) END at part-local bit e (exclusive);
// always three ORs, no per-lane branches, no left-alignment: with r = e % 32 and W = e / 32 the words W-2, W-1, W
// take hi >> r, {hi,lo} >> r and lo << (32-r).  Words in front of the string (down to two words in front of st[0])
// only ever receive an OR with zero.
__device__ __forceinline__ void put_bits64_end(uint32_t* st, uint32_t e, uint64_t acc);
__device__ __forceinline__ void put_row_pairs_mem(uint32_t* st, uint32_t q, const uint32_t* src, const uint2* lut, bool permuted);
// A lane-row of more than 64 bits, re-read from memory (cold): as two HALVES of eight symbols when each fits 64 bits
// (anything up to 8 bits per symbol on average: 4.5 bits per symbol ran at 0.9 TB/s through the pairs, tools/
// geometry_crossover.py), else pair by pair.
__device__ __forceinline__ void put_row_halves_mem(uint32_t* st, uint32_t q, const uint32_t* src, const uint2* lut, bool permuted)
{
    // (rolled loops, lengths first: cold code that must not cost the hot loop registers)
    uint32_t l0 = 0, l1 = 0;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = hash4_if(src[k], permuted);
        const uint32_t n = lut[w >> 24].y + lut[(w >> 16) & 255u].y + lut[(w >> 8) & 255u].y + lut[w & 255u].y;
        if (k < 2) l0 += n; else l1 += n;
    }
    if (l0 > 64u || l1 > 64u) { put_row_pairs_mem(st, q, src, lut, permuted); return; }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        uint64_t acc = 0;
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            const uint32_t w = hash4_if(src[2 * h + k], permuted);
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                const uint2 e = lut[(w >> (8 * b)) & 255u];
                acc = (acc << e.y) | e.x;                          // (lengths <= 31: the shift is defined)
            }
        }
        q += h ? l1 : l0;
        put_bits64_end(st, q, acc);
    }
}
__device__ __forceinline__ void put_bits64_end(uint32_t* st, uint32_t e, uint64_t acc)
{
    const uint32_t hi = (uint32_t)(acc >> 32), lo = (uint32_t)acc;
    const uint32_t r = e & 31u;
    uint32_t* w = st + (e >> 5);
#if defined(HVLC_ABL_PACK_PLAIN)
    w[-2] = hi >> r; w[-1] = __builtin_amdgcn_alignbit(hi, lo, r); w[0] = __builtin_amdgcn_alignbit(lo, 0u, r);   // (tools/: wrong output)
    asm volatile("" ::: "memory");
#elif defined(HVLC_ABL_PACK_ONE)
    lds_or(w - 1, (hi >> r) ^ __builtin_amdgcn_alignbit(hi, lo, r) ^ __builtin_amdgcn_alignbit(lo, 0u, r));        // (tools/: wrong output)
#else
    lds_or(w - 2, hi >> r);
    lds_or(w - 1, __builtin_amdgcn_alignbit(hi, lo, r));           // r == 0: lo
    lds_or(w,     __builtin_amdgcn_alignbit(lo, 0u, r));           // r == 0: 0
#endif
}

// A lane-row whose 16 codewords exceed 64 bits (lengths up to 31 allow 496): two symbols (<=62 bits)
// per put_bits64.  Rare with real tables, so it runs as a divergent branch of the fast path.
__device__ __forceinline__ void put_row_pairs(uint32_t* st, uint32_t q, const uint32_t (&wraw)[4], int nvalid,
                                              const uint2* lut, bool permuted)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nvalid) break;
        const uint32_t w = hash4_if(wraw[k], permuted);
#pragma unroll
        for (int b = 3; b >= 1; b -= 2) {
            const uint2 e0 = lut[(w >> (8 * b)) & 255u];
            const uint2 e1 = lut[(w >> (8 * (b - 1))) & 255u];
            put_bits64(st, q, ((uint64_t)e0.x << e1.y) | e1.x, e0.y + e1.y);
            q += e0.y + e1.y;
        }
    }
}

// The same for a lane-row of a FULL tile read back from memory, as a rolled loop (cold code: keep it small).
__device__ __forceinline__ void put_row_pairs_mem(uint32_t* st, uint32_t q, const uint32_t* src, const uint2* lut, bool permuted)
{
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {                             // pair k = bytes 3,2 (k even) or 1,0 (k odd) of word k/2
        const uint32_t w  = hash4_if(src[k >> 1], permuted);
        const uint32_t sh = (k & 1) ? 0u : 16u;
        const uint2 e0 = lut[(w >> (sh + 8u)) & 255u];
        const uint2 e1 = lut[(w >> sh) & 255u];
        put_bits64(st, q, ((uint64_t)e0.x << e1.y) | e1.x, e0.y + e1.y);
        q += e0.y + e1.y;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS of an encoder workgroup.
//
// The DATA WAVES NEVER WAIT FOR EACH OTHER and never touch the output: wave w owns segments 4w..4w+3 of every tile
// of its workgroup (its PART of the tile), packs them into its OWN staging ring at part-local bit positions and
// goes on to the next tile.  The workgroup's STORE WAVE takes a tile once the scout has seen its global bit offset:
// it stores the tile's stretch of the stream out of the ten rings, zeroes the ring words and hands them back
// (Smem::n_freed).  What the waves of a workgroup tell each other goes through small LDS records, written once per
// (tile, wave) and never waited for on the data waves' fast path:
//   WaveRec  per (slot, wave)   bits of the part, where it sits in the ring; its last 31 stream bits where a slow
//                               part needs them
//   TileSlot per slot           arrival counter + bit sum (the LAST wave to arrive publishes the tile's bit count
//                               and every part's base), and the scout's result (global offset, predecessor tail)
// A workgroup's n-th tile uses slot n % kSlots; a slot is reused only after the store wave has freed the slot's
// previous tile, so the fastest data wave runs at most kSlots-1 tiles ahead of the store wave.
// ---------------------------------------------------------------------------------------------
#ifndef HVLC_SLOTS
#define HVLC_SLOTS 8
#endif
// Issue priorities (s_setprio; tools/ab.py, round 3).  The store wave runs at 3 (one short serial instruction stream that ten
// data waves wait for: +12 %); the scout at 1 (its rounds are short and twenty data waves' tiles wait for them: +0.6 % H2.2,
// +1.5 % H4.0, +1.1 % H1.0 over 0, the same at 2 or 3); a data wave at 1 while it works on the workgroup's OLDEST
// unpublished tile and at 0 when it is ahead (a tile is published by its slowest wave, and which wave that is is decided
// by who wins the SIMD's issue arbitration -- the older waves, left alone: +2.1-2.6 % H2.2, 0...+3 % H1.0), but only
// while the rings hold more than three parts (at H4.0 they do not, the waves wait for ring words and the priority
// costs 2-4 %).  Priority 0 again once the wave has arrived (-1 %), 2 for a wave that finds half of the others arrived
// already (+-0), 2 or 3 for a scout on leader's duty (+-0), three levels by the age of the wave's tile (2 / 1 / 0: -1 %, +-0 with
// the scout at 2): measured, not taken.
#ifndef HVLC_PRIO_SCOUT
#define HVLC_PRIO_SCOUT 1
#endif
constexpr int      kSlots   = HVLC_SLOTS;
constexpr int      kTq      = HVLC_SLOTS <= 8 ? 16 : 32;                         // queue of drawn tile numbers: workgroup-local tile n at [n % kTq]
#ifndef HVLC_TICKETS_AHEAD
#define HVLC_TICKETS_AHEAD 2
#endif
constexpr int      kTixAhead = HVLC_TICKETS_AHEAD;         // tiles the scout draws ahead of the workgroup's slowest data wave (first guess)
constexpr int      kTixAheadMax = 3;
static_assert(kTq >= kSlots + kTixAheadMax + 2, "a tile number must stay in the queue until the store wave has stored the tile");
#ifndef HVLC_WAVE_RING_WORDS
#define HVLC_WAVE_RING_WORDS 1784        // six slots of 296 words (H2.2's parts: 292-296) and a word of eight to spare
#endif
constexpr int      kWideMax = 12;                          // lane-rows of more than 64 bits a wave keeps per part (the rest is re-read from memory)
#ifndef HVLC_RING_SLOTS
#define HVLC_RING_SLOTS 8                // a part takes a slot of the largest part's size so far, rounded up to this many words
#endif
constexpr uint32_t kWaveRing = HVLC_WAVE_RING_WORDS;       // staging ring of ONE data wave: 10 x 6.8 KiB per workgroup, two workgroups per CU
constexpr uint32_t kNoRoom   = 0xffffffffu;
constexpr uint32_t kRowWordsMax = 1024u * 31u / 32u + 8u;  // slow path: one segment at 31 bits per symbol + carry word + read-ahead
static_assert(kWaveRing % 4 == 0 && kWaveRing >= kRowWordsMax, "the slow path stages one segment in the (empty) ring");
constexpr uint32_t kBlockWords = 64;                       // the store wave's unit: one dword per lane = 256 bytes per instruction

struct alignas(16) WaveRec {
    uint32_t bits;            // T_w: bits of this wave's part of the tile
    uint32_t tail;            // kValid32 | the last 31 stream bits up to the end of the part; 0 until known (slow paths only)
    uint32_t start, need;     // ring words [start, start + need) hold the packed part; start = its carry word
    uint32_t base;            // bits of the tile in front of the part (written by the tile's last arriver)
    uint32_t rb1, rb2, rb3;   // bit offsets of segments 1..3 inside the part (segment 0 starts at 0)
};
struct alignas(16) TileSlot {
    uint64_t prefix;          // scout: global exclusive bit offset of the tile
    uint32_t pred_tail;       // scout: the last 31 bits of the stream before the tile
    uint32_t arrive;          // data waves: (arrivals << 24) | bits so far (a tile holds < 2^21 bits)
};
static_assert((uint64_t)kTileWords * 4u * 31u < (1ull << 24), "the arrival counter's bit sum");

struct alignas(16) Smem {
    uint2     lut[256];                       // 2 KiB: {codeword, length}; MUST be first (lut_at() addresses it absolutely)
    uint32_t  ring[kDataWaves][kWaveRing];    // per-wave staging rings
    WaveRec   wrec[kSlots][kDataWaves];
    TileSlot  ts[kSlots];
    uint32_t  packed[16];                     // data wave w: parts packed (or, slow parts, stored) so far -> store wave
    uint32_t  tq[kTq];                        // scout -> data, store: tile number of the workgroup's n-th tile
    uint32_t  n_tickets;                      // scout -> data: tiles drawn so far (release)
    uint32_t  n_final;                        // scout -> data, store: the workgroup's tile count once the tickets ran out, else ~0
    uint32_t  n_resolved;                     // scout -> store: tiles whose global offset is known (in order)
    uint32_t  n_bits_pub;                     // data -> scout: tiles whose bit count is published (in order; atomic max)
    uint32_t  n_freed;                        // store -> data: tiles stored, their ring words zeroed (in order)
    uint32_t  n_exited;                       // waves of the workgroup that have left (the last one reports to the context)
    uint32_t  abort_flag;
    uint32_t  tix_ahead;                      // data -> scout: tiles to draw ahead of the slowest data wave (from the ring's capacity in parts)
    uint8_t   len_tmp[256];                   // prologue only: the table's lengths, for the layout decision (not zeroed)
    uint4     wide[kDataWaves][kWideMax];     // the input words of lane-rows of more than 64 bits, kept from the encode to the pack
                                              // phase (the registers take the next tile's input in between); not zeroed
};
static_assert(kDataWaves <= 16, "Smem::packed[] and the arrival counter hold up to 16 data waves");
static_assert(2 * sizeof(Smem) <= 160 * 1024, "two workgroups per CU");

// Two-level decoupled look-back, executed by ONE wave with all 64 lanes active (k_exclusive_scan).
// Level 1: lanes 0..r-1 read the sums of the r predecessors inside the tile's group.
// Level 2: lanes read the granules of up to 64 preceding groups; the nearest inclusive prefix
//          plus the aggregates after it give the group's exclusive prefix.  The last tile of a
//          group (the "leader") publishes the group's aggregate as soon as level 1 is done and its
//          inclusive prefix after level 2, so one hop covers 64 tiles / 64 groups.
// Every granule is ONE naturally aligned 8-byte word read by ONE lane with ONE 8-byte load.
__device__ __forceinline__ bool lookback(const EncodeParams& p, uint64_t tile, uint64_t T, int lane, uint64_t& P_out)
{
    const uint32_t r = (uint32_t)(tile & (kGroupTiles - 1));
    const uint64_t g = tile / kGroupTiles;
    const bool leader = (r == kGroupTiles - 1) || (tile == p.num_tiles - 1);
    Spin spin;

    uint64_t d = 0;
    const bool need = (uint32_t)lane < r;
    const uint64_t* src = p.tile_bits() + (need ? tile - 1 - (uint64_t)lane : 0);
    for (;;) {
        if (!__any(need && !(d >> 63))) break;
        if (need && !(d >> 63)) d = ld_granule(src);
        if (!__any(need && !(d >> 63))) break;
        if (!spin.wait(p)) return false;
    }
    const uint64_t insum = wave_sum_u64(need ? (d & kValMask) : 0ull);
    if (leader && lane == 0) st_granule(&p.group_desc()[g], kStAgg | ((insum + T) & kValMask));

    uint64_t excl = 0;
    for (int64_t gb = (int64_t)g; gb > 0; gb -= 64) {
        const int64_t gi = gb - 1 - lane;
        uint64_t v = gi < 0 ? kStPre : 0ull;                    // before the stream: inclusive prefix 0
        int fp;
        for (;;) {
            const uint64_t pm = __ballot((v >> 62) == 2);
            fp = pm ? __builtin_ctzll(pm) : 64;
            const uint64_t xm  = __ballot((v >> 62) == 0);
            const uint64_t low = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
            if ((xm & low) == 0) break;
            if (gi >= 0 && (v >> 62) == 0) v = ld_granule(&p.group_desc()[gi]);
            const uint64_t pm2 = __ballot((v >> 62) == 2);
            fp = pm2 ? __builtin_ctzll(pm2) : 64;
            const uint64_t xm2  = __ballot((v >> 62) == 0);
            const uint64_t low2 = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
            if ((xm2 & low2) == 0) break;
            if (!spin.wait(p)) return false;
        }
        excl += wave_sum_u64(lane <= fp ? (v & kValMask) : 0ull);
        if (fp < 64) break;
    }
    const uint64_t P = excl + insum;
    if (leader && lane == 0) st_granule(&p.group_desc()[g], kStPre | ((P + T) & kValMask));
    P_out = P;
    return true;
}

// Loads of one wave's part of a tile: lane-row r = segment 4w+r, 16 bytes per lane, 1 KiB per wave-instruction,
// non-temporal (the input is read exactly once).  `full` must be wave-uniform (a scalar branch): the
// four loads of a full tile are issued back to back with nothing between them.
__device__ __forceinline__ void load_tile(const EncodeParams& p, uint32_t tile, bool full, int wave, int lane,
                                          uint4 (&raw)[kRows])
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (full) {
        // scalar 64-bit tile base + a 32-bit per-lane offset (no per-lane 64-bit pointer kept alive across the loop)
        const char* base = (const char*)p.in + (uint64_t)tile * (uint64_t)(kTileWords * 4);
        const uint32_t off = (uint32_t)(wave * kRows * 64 + lane) * 16u;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const u32x4 t = __builtin_nontemporal_load((const u32x4*)(base + off + (uint32_t)r * 1024u));
            raw[r] = make_uint4(t.x, t.y, t.z, t.w);
        }
    } else {
        const uint32_t* in32 = (const uint32_t*)p.in;
        const int cl = lane_id_fresh();      // (cold: not to be hoisted out of the data wave's loop)
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const uint64_t wbase = (uint64_t)tile * kTileWords + (uint64_t)((wave * kRows + r) * 64 + cl) * 4u;
            raw[r].x = wbase + 0 < p.num_words ? in32[wbase + 0] : 0u;
            raw[r].y = wbase + 1 < p.num_words ? in32[wbase + 1] : 0u;
            raw[r].z = wbase + 2 < p.num_words ? in32[wbase + 2] : 0u;
            raw[r].w = wbase + 3 < p.num_words ? in32[wbase + 3] : 0u;
        }
    }
}

// Does wave `wave`'s part (kRows segments) of `tile` lie completely inside the input?  (wave-uniform)
__device__ __forceinline__ bool part_is_full(const EncodeParams& p, uint32_t tile, int wave)
{
    return (uint64_t)tile * kTileWords + (uint64_t)(wave + 1) * (kRows * kSegWords) <= p.num_words;
}

// Does it lie completely behind the end of the input?  (Only parts of the last tile can: the ones behind the part the
// input ends in.  They hold no bits and nobody packs, stores or waits for them.)
__device__ __forceinline__ bool part_is_empty(const EncodeParams& p, uint32_t tile, int wave)
{
    return (uint64_t)tile * kTileWords + (uint64_t)wave * (kRows * kSegWords) >= p.num_words;
}
// Parts of `tile` that hold input: kDataWaves, fewer for the last tile.  The last of them ends the stream (it stores
// the open word and the total).
__device__ __forceinline__ int parts_of_tile(const EncodeParams& p, uint32_t tile)
{
    if (tile != (uint32_t)(p.num_tiles - 1)) return kDataWaves;
    const uint32_t words = (uint32_t)(p.num_words - (uint64_t)tile * kTileWords);      // 1 .. kTileWords
    return (int)((words + (uint32_t)(kRows * kSegWords) - 1u) / (uint32_t)(kRows * kSegWords));
}

__device__ __forceinline__ int valid_words(const EncodeParams& p, uint64_t wbase)
{
    return wbase >= p.num_words ? 0 : (p.num_words - wbase >= 4 ? 4 : (int)(p.num_words - wbase));
}

// Funnel-shift staged words by s = (global bit offset % 32) and store `nown` finished words at out[ow..], by ONE
// wave.  sb[0] is the carry slot (low s bits = last s bits of the preceding stream).
__device__ __forceinline__ void store_words(const EncodeParams& p, const uint32_t* sb, uint32_t s,
                                            uint64_t ow, uint32_t nown, int lane)
{
    // capacity: decided once per call (all arguments are wave-uniform), not per word
    if (ow + nown > p.out_cap) {
        if (lane == 0) atomicOr(p.status(), kFlagCapacity);
        nown = ow < p.out_cap ? (uint32_t)(p.out_cap - ow) : 0u;
    }
    if (HVLC_DBG(p, kDbgNoStore)) return;
    uint32_t* const dst = p.out + ow;                           // scalar base, 32-bit per-lane offsets
    // one word per lane and trip: consecutive lanes read consecutive LDS words (no bank conflicts) and store 256
    // contiguous bytes per wave-instruction.  Non-temporal: the stream is written once and not read again by this kernel;
    // at the memory system's limit for this read/write mix that is worth 2-4 % (plain stores: 3.49-3.53 TB/s, 3.25, 4.30
    // on H2.2 / H4.0 / H1.0 against 3.59-3.61, 3.32, 4.47 in one A/B)
    for (uint32_t k = (uint32_t)lane; k < nown; k += 64u)
        __builtin_nontemporal_store(__builtin_amdgcn_alignbit(sb[k], sb[k + 1], s), dst + k);
}

__device__ __forceinline__ void zero_words(uint32_t* base, uint32_t n_vec4, int lane)
{
    uint4* z = (uint4*)base;
    for (uint32_t i = (uint32_t)lane; i < n_vec4; i += 64u) z[i] = make_uint4(0, 0, 0, 0);
}

// LDS words shared between the waves of a workgroup: relaxed / acquire / release at WORKGROUP scope (LDS
// operations of one wave complete in order; the scopes only keep the compiler from reordering them).
__device__ __forceinline__ uint32_t lds_acq(const uint32_t* p) { return uni(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ uint32_t lds_rlx(const uint32_t* p) { return uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void lds_rel(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// Waiting data waves sleep (s_sleep) between looks at their LDS words; s_wakeup ends the sleep of every other wave of
// the workgroup at once, so the scout's news (a tile number, a resolved tile) is seen without short polling intervals.
__device__ __forceinline__ void wake_workgroup()
{
#ifndef HVLC_NO_WAKEUP
    asm volatile("s_wakeup" ::: "memory");
#endif
}

// One bounded waiting step of a data wave: false = the launch is being aborted (timeout somewhere).
struct WaveWait {
    Spin spin;
    __device__ __forceinline__ bool step(const EncodeParams& p, Smem& sm, int lane)
    {
        if (lds_rlx(&sm.abort_flag)) return false;
        if (!spin.wait(p)) {
            if (lane == 0) __hip_atomic_store(&sm.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return false;
        }
        return true;
    }
};

// The tail of a part once the tail of what precedes it is known: its own last 31 bits, or, if it is shorter,
// the predecessor's bits shifted up with its own appended.
__device__ __forceinline__ uint32_t chain_tail(uint32_t pred_tail, uint32_t own, uint32_t T)
{
    return T >= 31u ? own : (((pred_tail << T) | own) & 0x7fffffffu);
}

// Publish the stream tail behind (tile, wave)'s part: to the next wave of the workgroup through LDS, and, for the
// tile's last part, to the next tile's record (read by that tile's scout).
template <bool LATE>
__device__ __forceinline__ void publish_tail(const EncodeParams& p, Smem& sm, uint32_t slot, int wave, uint32_t tile,
                                             uint32_t tail, int lane)
{
    if (lane == 0) {
        lds_rel(&sm.wrec[slot][wave].tail, kValid32 | tail);
        int last = wave;
        if (LATE) {
            // A late tail is published when the tile is resolved, so every part's bit count is known: EMPTY parts behind
            // this one end in the same 31 bits -- publish for them too instead of handing the tail from wave to wave
            // (the end of the input: the parts behind the one the input ends in).  They may publish the same value again.
            while (last + 1 < kDataWaves && sm.wrec[slot][last + 1].bits == 0u) {
                ++last;
                lds_rel(&sm.wrec[slot][last].tail, kValid32 | tail);
            }
        }
        if (last == kDataWaves - 1) st_word(&p.recs()[(uint64_t)tile + 1].tail, kValid32 | tail);
    }
    if (LATE) wake_workgroup();                       // the next wave may be sleeping on it (an early tail is never waited for)
}

// The last 31 stream bits in front of (slot, wave)'s part of a RESOLVED tile: the tile's predecessor tail for wave 0,
// else what wave-1 published for the tile (normally long ago, at pack time; it is late only when wave-1's part is
// shorter than 31 bits or went through the slow path).  False = abort.
__device__ __forceinline__ bool pred_tail_of(const EncodeParams& p, Smem& sm, uint32_t slot, int wave, int lane, uint32_t& pt)
{
    if (wave == 0) { pt = uni(sm.ts[slot].pred_tail); return true; }
    WaveWait w;
    for (;;) {
        const uint32_t v = lds_acq(&sm.wrec[slot][wave - 1].tail);
        if (v & kValid32) { pt = v & 0x7fffffffu; return true; }
        if (!w.step(p, sm, lane)) return false;
    }
}

// The tail (last 31 stream bits) of a FULL part that will take the slow path, computed at pack time from its last
// segment alone -- in registers: every lane folds its 16 symbols into (last <=31 bits, their count), a wave scan with
// the same operator joins the lanes.  Published like the early tail of a packed part, so that nothing behind a slow
// part waits for it to be stored: a stream of slow parts only (a table that EXPANDS its input beyond ~13 bits per
// byte) otherwise runs one part at a time, each waiting for the tail of the one in front.  False: the last segment
// holds fewer than 31 bits (zero-length codes), the tail follows when the part is stored.
__device__ __forceinline__ bool publish_slow_tail_early(const EncodeParams& p, Smem& sm, uint32_t slot, int wave, uint32_t tile,
                                                        int /*lane*/, bool permuted)
{
    const int lane = lane_id_fresh();      // cold code: nothing here may be hoisted into the data wave's loop as an invariant
    const uint64_t seg = (uint64_t)tile * kSegsPerTile + (uint64_t)(wave * kRows + (kRows - 1));
    const uint4 v = ((const uint4*)p.in)[seg * (kSegWords / 4u) + (uint64_t)lane];
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
    uint32_t tl = 0, ln = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t h = hash4_if(w[k], permuted);
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            const uint2 e = sm.lut[(h >> (8 * b)) & 255u];             // lengths <= 31: the shift is defined
            tl = ((tl << e.y) | e.x) & 0x7fffffffu;
            ln = min(31u, ln + e.y);
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {                                 // inclusive scan: (A, B) -> B behind A
        const uint32_t tl_a = (uint32_t)__shfl_up((int)tl, d), ln_a = (uint32_t)__shfl_up((int)ln, d);
        if (lane >= d && ln < 31u) {
            tl = ((tl_a << ln) | tl) & 0x7fffffffu;
            ln = min(31u, ln_a + ln);
        }
    }
    const uint32_t tail = (uint32_t)__builtin_amdgcn_readlane((int)tl, 63);
    if ((uint32_t)__builtin_amdgcn_readlane((int)ln, 63) < 31u) return false;
    publish_tail<false>(p, sm, slot, wave, tile, tail, lane);
    wake_workgroup();                                                  // unlike a packed part's, this tail may be waited for already
    return true;
}

// Slow path for ONE wave's part of a tile: the part does not fit the wave's (empty) ring (more than ~13 bits per
// symbol), or the tile is the one the input ends in.  Runs with the ring drained and the tile resolved, segment by
// segment: the segment is re-read (valid words only), packed two symbols (<=62 bits) per put_bits64 into the ring
// taken as one buffer, funnel-shifted and stored; the carry runs through the segments exactly as it runs through
// parts.  Leaves the ring zeroed.  `tail_published`: publish_slow_tail_early has done its work.
__device__ __forceinline__ bool slow_part(const EncodeParams& p, Smem& sm, uint32_t* ring, uint32_t j, uint32_t tile,
                                          int wave, int /*lane*/, bool permuted, bool tail_published)
{
    const int lane = lane_id_fresh();      // cold code: see publish_slow_tail_early
    const uint32_t slot = j % kSlots;
    uint64_t G = uni64(sm.ts[slot].prefix) + (uint64_t)uni(sm.wrec[slot][wave].base);
    uint32_t carry = 0;
    zero_words(ring, (kRowWordsMax + 3u) / 4u, lane);          // (packed parts leave their words behind: the ring is empty, not clean)
    const bool last_part = (tile == (uint32_t)(p.num_tiles - 1)) && wave == parts_of_tile(p, tile) - 1;   // the part the input ends in
    const uint32_t* in32 = (const uint32_t*)p.in;
#pragma unroll 1
    for (int r = 0; r < kRows; ++r) {
        const uint64_t seg   = (uint64_t)tile * kSegsPerTile + (uint64_t)(wave * kRows + r);
        const uint64_t wbase = seg * kSegWords + (uint64_t)lane * 4u;
        const int n = valid_words(p, wbase);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = k < n ? in32[wbase + k] : 0u;
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) {
                const uint32_t h = hash4_if(w[k], permuted);
#pragma unroll
                for (int b = 3; b >= 0; --b) t += sm.lut[(h >> (8 * b)) & 255u].y;
            }
        const uint32_t incl = wave_incl_scan(t);
        const uint32_t Tq = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        put_row_pairs(ring + 1, incl - t, w, n, sm.lut, permuted);
        if (r == 0 && !pred_tail_of(p, sm, slot, wave, lane, carry)) return false;     // first needed here, behind the packing
        if (p.offsets && lane == 0 && seg < p.num_segs) p.offsets[seg] = G;
        if (lane == 0) ring[0] = carry;
        const uint32_t s = (uint32_t)G & 31u;
        const bool lastq = last_part && r == kRows - 1;
        uint32_t nown = (s + Tq) >> 5;
        if (lastq) nown = ((s + Tq + 31u) >> 5) + ((((s + Tq) & 31u) == 0u) ? 1u : 0u);
        store_words(p, ring, s, G >> 5, nown, lane);
        carry = chain_tail(carry, extract_tail(ring + 1, Tq), Tq);    // every lane computes the same carry
        zero_words(ring, (((Tq + 31u) >> 5) + 8u) / 4u, lane);
        G += Tq;
    }
    if (last_part && lane == 0) {
        p.result()[0] = G;
        if (p.total_bits) *p.total_bits = G;
    }
    if (!tail_published) publish_tail<true>(p, sm, slot, wave, tile, carry, lane);
    return true;
}

// Ring words a packed part of T bits takes: carry word + data + the two words behind the data that pack / store touch.
__device__ __forceinline__ uint32_t ring_need(uint32_t T) { return (((T + 31u) >> 5) + 3u + 3u) & ~3u; }
// Is (tile, wave)'s part of T bits packed into its wave's ring (the fast path), or does it take the slow path?
// Wave-uniform; the data wave and the store wave answer it the same way (T is in the part's record once it arrived).
__device__ __forceinline__ bool part_is_fast(const EncodeParams& p, uint32_t tile, int wave, uint32_t T)
{
    return (tile != (uint32_t)(p.num_tiles - 1) || part_is_full(p, tile, wave)) && ring_need(T) <= kWaveRing;
}

// Where `need` words fit in a data wave's staging ring: kNoRoom = nowhere for now.  Allocations are contiguous (a part
// that does not fit before the end of the ring starts over at word 0) and are released in order, a tile at a time, by
// the store wave: `freed` tiles of the workgroup are stored, this wave has `packed` parts in (or through) its ring.
__device__ __forceinline__ uint32_t ring_place(const Smem& sm, int wave, uint32_t head, uint32_t packed, uint32_t freed, uint32_t need)
{
    if (freed >= packed) return 0u;                                            // empty: start over at word 0 (need <= kWaveRing is the caller's business)
    const uint32_t tail = uni(sm.wrec[freed % kSlots][wave].start);            // oldest part still in the ring
    if (head > tail) {
        if (head + need <= kWaveRing) return head;
        return need <= tail ? 0u : kNoRoom;
    }
    return (head < tail && need <= tail - head) ? head : kNoRoom;              // head == tail with parts in flight: full
}

// ------------------------------------------------------------------------------------------------
// A DATA wave of the pipelined encoder.  One loop iteration = one tile of the workgroup, this wave's part:
//   1  wait for the part's input (loaded during the previous iteration), look up / concatenate its 4 x 16 symbols
//      per lane (registers only), DPP-scan the lane-row bit counts
//   2  issue the loads of the next tile's part into the registers the lookups just vacated
//   3  write the part's record and ARRIVE (the last wave to arrive publishes the tile's bit count)
//   4  take the part's words in the ring, pack (three branch-free ds_or_b32 per lane-row), hand the part to the store
//      wave (Smem::packed); the tile's last part also hands its last 31 bits to the next tile's record
// Waiting happens only for: a tile number (scout), ring space / a free slot (the store wave, i.e. resolution of older
// tiles).  The output is the store wave's business -- a data wave issues no store to the stream (the slow path aside).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_data_wave(const EncodeParams& p, Smem& sm, int lane, int wave, bool permuted)
{
    uint4 raw[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) raw[r] = make_uint4(0, 0, 0, 0);
    uint32_t* const ring = sm.ring[wave];
    uint32_t n_loaded = 0, n_packed = 0, head = 0;           // workgroup-local tile indices: input loaded / part packed; ring: where the next part's words start
#if HVLC_RING_SLOTS
    uint32_t slot_words = 0;
#endif
    uint32_t n_final = ~0u;
    const uint32_t last_tile = (uint32_t)(p.num_tiles - 1);
    // 32-bit word strings (encode16_w32): a part in which some word's four codewords exceed 32 bits is encoded again the
    // general way; a wave that keeps meeting such parts (score: +8 per failure, -1 per success) stops trying for a while
    constexpr uint32_t kW32Off = 24u;
    uint32_t w32_score = 0;
    WS_DECL;

    // issue the loads of the workgroup's tile number n_loaded if the scout has drawn it
    auto try_load = [&]() -> bool {
        if (lds_acq(&sm.n_tickets) <= n_loaded) return false;
        const uint32_t t = uni(sm.tq[n_loaded % kTq]);
        load_tile(p, t, t != last_tile || part_is_full(p, t, wave), wave, lane, raw);
        ++n_loaded;
        return true;
    };

    for (;;) {
        // ---- input of the next tile: normally prefetched; at the start and when the scout was behind, wait here
        if (n_loaded == n_packed) {
            WaveWait w;
            for (;;) {
                if (n_final == ~0u) n_final = lds_rlx(&sm.n_final);      // (read BEFORE the ticket count: once set, every ticket is visible)
                if (try_load()) { TLX(p, 3, wave == 0 && lane == 0 && n_loaded == 1u); break; }     // first tile number seen, loads issued
                if (n_final != ~0u && n_loaded >= n_final) { WS_LAP(kWsWaitTicket); WS_DUMP(p, wave, lane); return; }   // the tickets ran out: nothing left to encode
                if (!w.step(p, sm, lane)) return;
            }
            WS_LAP(kWsWaitTicket);
        }
        const uint32_t cur  = uni(sm.tq[n_packed % kTq]);
        const uint32_t slot = n_packed % kSlots;
        const bool full = cur != last_tile || part_is_full(p, cur, wave);   // (only the part the input ends in, and the empty ones behind it, are not)
#ifndef HVLC_PRIO_ADAPT
#define HVLC_PRIO_ADAPT 1
#endif
#if HVLC_PRIO_ADAPT
#ifndef HVLC_PRIO_ADAPT_MIN_AHEAD
#define HVLC_PRIO_ADAPT_MIN_AHEAD 2u
#endif
        // issue priority goes to the waves that work on the workgroup's OLDEST unpublished tile (see HVLC_PRIO_SCOUT above)
        if (n_packed <= lds_rlx(&sm.n_bits_pub) && lds_rlx(&sm.tix_ahead) >= HVLC_PRIO_ADAPT_MIN_AHEAD) {
            __builtin_amdgcn_s_setprio(HVLC_PRIO_ADAPT);
        }
        else __builtin_amdgcn_s_setprio(0);
#endif

        // ---- 1: lookups, concatenation, scan (registers only)
#ifdef HVLC_EXPERIMENT
        if (n_packed == 0u && wave == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TLX(p, 4, lane == 0); }   // the first tile's words are here
#ifdef HVLC_WS_LOADWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (wave_stats: how long a data wave waits for its input, by itself)
        WS_LAP(kWsWaitDrain);
#endif
#endif
        uint64_t acc[kRows];
        uint32_t tot[kRows], incl[kRows], rowbase[kRows];
        if (full) {
            // (row after row: the four rows advanced together, with the next step's table entries requested ahead --
            // eight to twelve LDS reads in flight instead of four -- took 5 % MORE time: tools/ab.py, round 3)
            bool wide = true;
#ifndef HVLC_NO_W32
            if (w32_score < kW32Off) {                        // the 32-bit word strings, unless they kept failing lately
                uint32_t wm[kRows];
                if (permuted) {
#pragma unroll
                    for (int r = 0; r < kRows; ++r) encode16_w32<true>(raw[r], sm.lut, acc[r], tot[r], wm[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < kRows; ++r) encode16_w32<false>(raw[r], sm.lut, acc[r], tot[r], wm[r]);
                }
                wide = __any(max(max(wm[0], wm[1]), max(wm[2], wm[3])) > 32u);
                w32_score = wide ? w32_score + 8u : (w32_score ? w32_score - 1u : 0u);
            } else if (++w32_score >= kW32Off + 64u) w32_score = kW32Off - 8u;     // every 64 parts another try
#endif
            if (wide) {
                if (permuted) {
#pragma unroll
                    for (int r = 0; r < kRows; ++r) encode16<true>(raw[r], sm.lut, acc[r], tot[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < kRows; ++r) encode16<false>(raw[r], sm.lut, acc[r], tot[r]);
                }
            }
        } else {
            // the part the input ends in (or an empty one behind it): bit counts only, the slow path packs it (cold code)
            const int cl = lane_id_fresh();
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const uint64_t wbase = (uint64_t)cur * kTileWords + (uint64_t)((wave * kRows + r) * 64 + cl) * 4u;
                const int nv = valid_words(p, wbase);
                const uint32_t wd[4] = { raw[r].x, raw[r].y, raw[r].z, raw[r].w };
                uint32_t t = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nv) {
                        const uint32_t h = hash4_if(wd[k], permuted);
                        t += sm.lut[h >> 24].y + sm.lut[(h >> 16) & 255u].y + sm.lut[(h >> 8) & 255u].y + sm.lut[h & 255u].y;
                    }
                tot[r] = t; acc[r] = 0;
            }
        }
        uint32_t T = 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            incl[r]    = wave_incl_scan(tot[r]);
            rowbase[r] = T;
            T         += (uint32_t)__builtin_amdgcn_readlane((int)incl[r], 63);
        }

        // lane-rows of more than 64 bits (H2.2: one in 3000, i.e. in 55 % of the tiles) are packed from their input words in
        // the pack phase: kept in LDS until then, a slot per row in lane order (the same count finds it again)
        if (full && __any(max(max(tot[0], tot[1]), max(tot[2], tot[3])) > 64u)) {
            const int cl = lane_id_fresh();
            uint32_t nw = 0;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const uint64_t m = __ballot(tot[r] > 64u);
                const uint32_t idx = nw + (uint32_t)__builtin_popcountll(m & ((1ull << cl) - 1ull));
                if (tot[r] > 64u && idx < (uint32_t)kWideMax) sm.wide[wave][idx] = raw[r];
                nw += (uint32_t)__builtin_popcountll(m);
            }
        }
        WS_LAP(kWsEncode);
        TLX(p, 5, wave == 0 && lane == 0 && n_packed == 0u);           // the first part is encoded and scanned
        // ---- 2: the next tile's loads.  (Issued EARLIER -- row by row between the lookups, each row into the registers its
        // lookups had just vacated, when the tile's number was already there -- they made the encoder 7-15 % slower:
        // tools/ab.py, round 3.  More requests in flight lengthen the queues of a memory system that is busy already.  Issued
        // LATER, behind the arrival: -0.6 % H2.2, -1.3 % H1.0.)
        (void)try_load();
        WS_LAP(kWsStore);

        // ---- 3: slot, record, arrival
        uint32_t freed = lds_acq(&sm.n_freed);
        if (n_packed >= (uint32_t)kSlots + freed) {           // the slot's previous tile must have been stored
            WaveWait w;
            while (n_packed >= (uint32_t)kSlots + (freed = lds_acq(&sm.n_freed))) {
                if (!w.step(p, sm, lane)) return;
            }
        }
        WS_LAP(kWsWaitSlot);
        const uint32_t need = ring_need(T);
        const bool slow = !full || need > kWaveRing;          // (== !part_is_fast: the store wave's view of this part)
        if (lane == 0) {
            WaveRec& rec = sm.wrec[slot][wave];
            *reinterpret_cast<uint2*>(&rec.bits) = make_uint2(T, 0u);                        // bits; tail not known yet
            *reinterpret_cast<uint4*>(&rec.base) = make_uint4(0u, rowbase[1], rowbase[2], rowbase[3]);
        }
        // ARRIVE before anything that may wait: other workgroups wait for the tile's bit count.  (The record is written
        // before the add: release; the last arriver reads the others' records: acquire.)
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(&sm.ts[slot].arrive, (1u << 24) | T, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = uni(old);
        if ((old >> 24) == (uint32_t)(kDataWaves - 1)) {
            // last to arrive: every part's base, the tile's bit count (what other workgroups wait for), slot counter reset
            // (the lane number is taken afresh instead of kept live from the prologue)
            const int ln = lane_id_fresh();
            const uint32_t b    = ln < kDataWaves ? sm.wrec[slot][ln].bits : 0u;
            const uint32_t incb = wave_incl_scan(b);
            if (ln < kDataWaves) sm.wrec[slot][ln].base = incb - b;
            const uint32_t Ttile = (uint32_t)__builtin_amdgcn_readlane((int)incb, kDataWaves - 1);
            if (lane == 0) {
#ifndef HVLC_FIXED_TICKETS_AHEAD
                // How far ahead tiles are worth drawing depends on how many parts a ring holds: a drawn tile whose part cannot
                // be packed (no ring words until older tiles are resolved) is published late, and every tile of its group
                // waits for it.  Measured (tools/ab.py): 1 tile ahead is best at 3 parts per ring (4 bits per symbol: +6 %
                // over 2), 3 at 9 parts (1.3 bits per symbol: +3 %), 2 in between.
#ifdef HVLC_PARTS_GUARANTEED
                // (allocations are contiguous: a part that does not fit before the ring's end starts over at word 0 and leaves
                // up to need - 1 words unused -- what a ring holds FOR SURE is one part less than its size suggests)
                const uint32_t need_avg = ring_need(Ttile / (uint32_t)kDataWaves);
                const uint32_t parts = (kWaveRing + 1u) / need_avg > 1u ? (kWaveRing + 1u) / need_avg - 1u : 1u;
#else
                const uint32_t parts = kWaveRing / ring_need(Ttile / (uint32_t)kDataWaves);
#endif
#ifndef HVLC_AHEAD3_PARTS
#define HVLC_AHEAD3_PARTS 8u
#endif
                sm.tix_ahead = parts <= 3u ? 1u : parts >= HVLC_AHEAD3_PARTS ? 3u : 2u;
#endif
                sm.ts[slot].arrive = 0u;
                (void)__hip_atomic_fetch_max(&sm.n_bits_pub, n_packed + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur != p.stall_tile) st_word(&p.bits32()[cur], kValid32 | Ttile);
            }
            TL(p, cur, 0, lane == 0);
        }

        WS_LAP(kWsArrive);
        if (!full && part_is_empty(p, cur, wave)) {
            // behind the end of the input (cold): the part has arrived with zero bits; nothing to pack, store or wait for
            if (lane == 0) *reinterpret_cast<uint2*>(&sm.wrec[slot][wave].start) = make_uint2(0u, 0u);
            ++n_packed;
            if (lane == 0) lds_rel(&sm.packed[wave], n_packed);
            head = 0;
        } else if (!slow) {
            // ---- 4: ring words, pack, hand over
            uint32_t start;
#if HVLC_RING_SLOTS
            // the words a part TAKES are a slot of the size the largest part so far needed (rounded up): parts of slightly
            // different sizes otherwise leave a ring that holds "four" of them with room for three whenever the one that
            // wraps round is a word or two larger than the one that left (3.25-3.75 bits per symbol: +20-28 %, tools/
            // geometry_crossover.py)
            if (need > slot_words || need + 64u < slot_words) slot_words = min((need + (uint32_t)(HVLC_RING_SLOTS) - 1u) & ~((uint32_t)(HVLC_RING_SLOTS) - 1u), kWaveRing);
            const uint32_t take = slot_words;
#else
            const uint32_t take = need;
#endif
            if ((start = ring_place(sm, wave, head, n_packed, freed, take)) == kNoRoom) {
                WaveWait w;
                while ((start = ring_place(sm, wave, head, n_packed, lds_acq(&sm.n_freed), take)) == kNoRoom)
                    if (!w.step(p, sm, lane)) return;
            }
            head = start + take;
            if (lane == 0) *reinterpret_cast<uint2*>(&sm.wrec[slot][wave].start) = make_uint2(start, need);
            zero_words(ring + start, need / 4u, lane);       // (what an older part left there; the store wave only reads)
            uint32_t* const st = ring + start + 1;
            if (!HVLC_DBG(p, kDbgNoPack)) {
                const uint32_t longest = max(max(tot[0], tot[1]), max(tot[2], tot[3]));
                if (!__any(longest > 64u)) {
#pragma unroll
                    for (int r = 0; r < kRows; ++r) put_bits64_end(st, rowbase[r] + incl[r], acc[r]);
                } else {
                    // cold: some lane-row holds more than 64 bits (rare with real tables): its `acc` is meaningless; re-read such
                    // a row (L2) and pack it pairwise
#pragma unroll
                    for (int r = 0; r < kRows; ++r) put_bits64_end(st, rowbase[r] + incl[r], tot[r] <= 64u ? acc[r] : 0ull);
                    const int cl = lane_id_fresh();
                    uint32_t nw = 0;
#pragma unroll
                    for (int r = 0; r < kRows; ++r) {
                        const uint64_t m = __ballot(tot[r] > 64u);
                        const uint32_t idx = nw + (uint32_t)__builtin_popcountll(m & ((1ull << cl) - 1ull));
                        nw += (uint32_t)__builtin_popcountll(m);
                        if (tot[r] > 64u) {
                            const char* base = (const char*)p.in + (uint64_t)cur * (uint64_t)(kTileWords * 4);
                            const uint32_t* src = idx < (uint32_t)kWideMax
                                                      ? reinterpret_cast<const uint32_t*>(&sm.wide[wave][idx])          // kept by the encode phase
                                                      : (const uint32_t*)(base + (uint32_t)((wave * kRows + r) * 64 + cl) * 16u);
                            put_row_halves_mem(st, rowbase[r] + incl[r] - tot[r], src, sm.lut, permuted);
                        }
                    }
                }
            } else if (acc[0] + acc[1] + acc[2] + acc[3] == 0x123456789ull) st[incl[0] & 1023u] = tot[0];
            // the tile's last part hands the stream's last 31 bits to the next tile (whose store wave may be waiting for
            // nothing else) as soon as they exist; a shorter part's tail follows from the store wave
            if (wave == kDataWaves - 1 && T >= 31u) {
                const uint32_t tl = extract_tail(st, T);
                if (lane == 0) st_word(&p.recs()[(uint64_t)cur + 1].tail, kValid32 | tl);
            }
            ++n_packed;
            if (lane == 0) lds_rel(&sm.packed[wave], n_packed);
            WS_LAP(kWsPack); WS_COUNT(kWsTiles);
        } else {
            // slow path (cold): the tail first if the part allows, then wait until the ring is empty and this tile's offset
            // is known, store the part directly
            const bool tail_out = full && publish_slow_tail_early(p, sm, slot, wave, cur, lane, permuted);
            if (lane == 0) *reinterpret_cast<uint2*>(&sm.wrec[slot][wave].start) = make_uint2(0u, 0u);
            WaveWait w;
            while (lds_acq(&sm.n_freed) < n_packed)          // every older tile stored: nothing of this wave is left in its ring
                if (!w.step(p, sm, lane)) return;
            while (lds_acq(&sm.n_resolved) <= n_packed)
                if (!w.step(p, sm, lane)) return;
            if (!slow_part(p, sm, ring, n_packed, cur, wave, lane, permuted, tail_out)) return;
            ++n_packed;
            if (lane == 0) lds_rel(&sm.packed[wave], n_packed);
            head = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// STORE wave: the only wave of a workgroup that writes the stream (slow parts aside).  Tile by tile, in the order the
// scout resolves them: the segment offsets (one lane per segment, one store), then the tile's stretch of the stream.
// Part w's words are global words [ow_w, ow_{w+1}) with ow_w = (tile offset + base_w) / 32: word g of part w is
// alignbit(sb[k], sb[k + 1], s_w), k = g - ow_w, where sb[0] is the part's carry word (the 32 stream bits in front of
// the part; the store wave takes them from the part in front as it goes) and sb[1..] the packed words.  The words are
// stored in BLOCKS of 64 (256 bytes) aligned by address: one dword per lane, consecutive lanes read consecutive LDS
// words (no bank conflicts).  The block in which part w ends and part w + 1 begins is stored once, its lanes fed from
// two rings (a `junction`); only a tile's first and last block are partial, and a block next to a slow part.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t lane_value64(uint64_t v, uint32_t l);
struct PartSrc {                 // where the words of a packed part come from (all wave-uniform)
    const uint32_t* sb;          // carry word; sb[1 + j] = the part's own word j
    uint64_t ow;                 // global index of the part's first word
    uint32_t s;                  // global bit offset % 32
    uint32_t need;               // ring words to zero once the part is stored
};

// The 32 stream bits that end where a part of T bits ends: its own last bits, preceded -- when it is shorter than
// that -- by what its carry word holds (only the last 31 are ever used).
__device__ __forceinline__ uint32_t carry_behind(const uint32_t* sb, uint32_t T)
{
    const uint32_t wi = T >> 5, sh = T & 31u;
    const uint32_t hi = uni(sb[wi]), lo = uni(sb[wi + 1]);
    return sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
}

// Words [g0, g1) of one part, g0 >= c.ow, any alignment, at most kBlockWords of them: one store instruction.
__device__ __forceinline__ void store_piece(const EncodeParams& p, const PartSrc& c, uint64_t g0, uint64_t g1, int lane)
{
    const uint32_t k0 = (uint32_t)(g0 - c.ow), n = (uint32_t)(g1 - g0);
    uint32_t* const dst = p.out + g0;
    const uint32_t* const src = c.sb + k0;
    if ((uint32_t)lane < n) __builtin_nontemporal_store(__builtin_amdgcn_alignbit(src[lane], src[lane + 1], c.s), dst + lane);
}

// Store tile number n of the workgroup (resolved), part after part, every scalar fetched from LDS when it is needed:
// the general form (slow parts, parts shorter than a word, the last tile).  False = abort.
__device__ __forceinline__ bool store_tile_generic(const EncodeParams& p, Smem& sm, uint32_t n, int /*lane*/)
{
    const int lane = lane_id_fresh();      // cold code: see publish_slow_tail_early
    const uint32_t slot = n % kSlots;
    const uint32_t tile = uni(sm.tq[n % kTq]);
    const uint64_t P    = uni64(sm.ts[slot].prefix);
    uint32_t carry      = uni(sm.ts[slot].pred_tail);         // the (up to 31) stream bits in front of the next part
    const bool last_tile = tile == (uint32_t)(p.num_tiles - 1);
    // address alignment of the stream: block boundaries are words g with (g + mis) % 64 == 0
    const uint32_t mis = (uint32_t)((uintptr_t)p.out >> 2) & (kBlockWords - 1u);
    const uint64_t cap = p.out_cap;

    // ---- segment offsets: lane l = segment l of the tile (parts' bases by the last arriver, rows by each data wave)
    if (p.offsets && lane < kSegsPerTile && !HVLC_DBG(p, kDbgNoOffsets)) {
        const WaveRec& r = sm.wrec[slot][lane / kRows];
        const uint32_t rr = (uint32_t)lane % (uint32_t)kRows;
        const uint32_t rel = r.base + (rr ? (&r.base)[rr] : 0u);
        const uint64_t seg = (uint64_t)tile * kSegsPerTile + (uint64_t)lane;
        if (seg < p.num_segs) __builtin_nontemporal_store(P + rel, &p.offsets[seg]);
    }

    PartSrc prev = { nullptr, 0, 0, 0 };
    bool pend = false;               // prev's last, partial block [block start, prev's end) waits for the next part's first words
    bool prev_fast = false;          // the part in front (of this tile) was a packed part: `carry` is mine to hand on
    WaveWait w;
    const int nparts = parts_of_tile(p, tile);                 // (the last tile: the parts behind the end of the input are nobody's business)
#pragma unroll 1
    for (int wv = 0; wv < nparts; ++wv) {
        const uint32_t T    = uni(sm.wrec[slot][wv].bits);
        const uint64_t G    = P + (uint64_t)uni(sm.wrec[slot][wv].base);
        if (!part_is_fast(p, tile, wv, T)) {
            // ---- a slow part stores itself (cold): give its wave the tail in front of it, take the tail behind it
            if (wv > 0 && prev_fast) {
                if (lane == 0) lds_rel(&sm.wrec[slot][wv - 1].tail, kValid32 | (carry & 0x7fffffffu));
                wake_workgroup();
            }
            for (;;) {
                const uint32_t v = lds_acq(&sm.wrec[slot][wv].tail);
                if (v & kValid32) { carry = v & 0x7fffffffu; break; }
                if (!w.step(p, sm, lane)) return false;
            }
            prev_fast = false;
            continue;
        }
        while (lds_acq(&sm.packed[wv]) <= n)                   // (normally packed long ago)
            if (!w.step(p, sm, lane)) return false;
        const uint2 sn = *reinterpret_cast<const uint2*>(&sm.wrec[slot][wv].start);
        PartSrc c;
        c.sb = sm.ring[wv] + uni(sn.x); c.need = uni(sn.y); c.ow = G >> 5; c.s = (uint32_t)G & 31u;
        if (lane == 0) const_cast<uint32_t*>(c.sb)[0] = carry;  // carry word (same-wave LDS operations complete in order)

        // the part's words: [lo, hi); the last part of the stream also stores the open word, zero padded, and the zero
        // word cpu_vlc_encode leaves behind when the stream ends on a word boundary (cpuencode.cpp:39)
        const bool last = last_tile && wv == nparts - 1;
        const uint64_t lo = c.ow;
        uint64_t hi = (G + T) >> 5;
        if (last) {
            hi = (G + T + 31u) >> 5;
            if (((G + T) & 31u) == 0u) hi += 1u;
            if (lane == 0) {
                p.result()[0] = G + T;
                if (p.total_bits) *p.total_bits = G + T;
            }
        }
        if (hi > cap) {                                          // capacity: decided per part, not per word
            if (lane == 0) atomicOr(p.status(), kFlagCapacity);
            hi = cap > lo ? cap : lo;
        }
        const bool more = wv + 1 < nparts && hi == ((G + T) >> 5) &&
                          part_is_fast(p, tile, wv + 1, uni(sm.wrec[slot][wv + 1].bits));   // the next part continues this one's last block
        if (!HVLC_DBG(p, kDbgNoStore)) {
            // first block: from the block boundary at or below lo
            const uint64_t B0 = lo - (((uint32_t)lo + mis) & (kBlockWords - 1u));
            const uint64_t e0 = hi < B0 + kBlockWords ? hi : B0 + kBlockWords;
            if (pend) {
                // junction: lanes below lo take prev's last words, the others this part's first words
                const uint32_t nprev = (uint32_t)(lo - B0), nall = (uint32_t)(e0 - B0);
                const bool isprev = (uint32_t)lane < nprev;
                const uint32_t* src = isprev ? prev.sb + (uint32_t)(B0 - prev.ow) : c.sb - nprev;
                const uint32_t  sh  = isprev ? prev.s : c.s;
                if ((uint32_t)lane < nall)
                    __builtin_nontemporal_store(__builtin_amdgcn_alignbit(src[lane], src[lane + 1], sh), p.out + B0 + lane);
            } else if (e0 > lo) {
                store_piece(p, c, lo, e0, lane);
            }
            // whole blocks
            uint64_t B = B0 + kBlockWords;
            if (B + kBlockWords <= hi) {
                uint32_t* dst = p.out + B;
                const uint32_t* src = c.sb + (uint32_t)(B - c.ow) + lane;
                const uint32_t nblk = (uint32_t)((hi - B) / kBlockWords);
#pragma unroll 1
                for (uint32_t i = 0; i < nblk; ++i) {
                    __builtin_nontemporal_store(__builtin_amdgcn_alignbit(src[0], src[1], c.s), dst + lane);
                    src += kBlockWords; dst += kBlockWords;
                }
                B += (uint64_t)nblk * kBlockWords;
            }
            // last, partial block [B, hi): left to the next part if that continues it, else stored now
            if (B < hi && B > lo && !more) store_piece(p, c, B, hi, lane);
        }
        carry = carry_behind(c.sb, T);
        const uint64_t Bl = hi - (((uint32_t)hi + mis) & (kBlockWords - 1u));
        pend = more && Bl > lo && Bl < hi && !HVLC_DBG(p, kDbgNoStore);
        if (pend) prev = c;
        prev_fast = true;
        // the tile's last part: the next tile's record takes the tail now if the data wave could not give it at pack time
        if (wv == kDataWaves - 1 && T < 31u && lane == 0)
            st_word(&p.recs()[(uint64_t)tile + 1].tail, kValid32 | (carry & 0x7fffffffu));
    }
    return true;
}


#ifdef HVLC_ABL_NO_GLOBAL_STORE
#define ABL_STORE(v, ptr) do { auto v_ = (v); asm volatile("" :: "v"(v_)); (void)(ptr); } while (0)   // (tools/: the store wave's work without its memory traffic)
#else
#define ABL_STORE(v, ptr) __builtin_nontemporal_store(v, ptr)
#endif
// Store tile number n of the workgroup (resolved).  The common case -- ten packed parts of at least a word each, not
// the stream's last tile -- keeps every part's record in the lane of the part's number (one vector read each), takes
// its scalars from there with v_readlane and counts in words RELATIVE to the tile's first word (32-bit scalar
// arithmetic); everything else goes through store_tile_generic.  The store wave is one serial instruction stream that
// ten data waves wait for, and what it spends its time on is LDS round trips: per part, the reads of the block the part
// starts in (a junction with the part in front, or the tile's ragged first block) and of its first four whole blocks
// go out TOGETHER, then the shifts, then the stores.  The data waves zero their ring words themselves (when they take
// them), so nothing is left to do behind the stores.  False = abort.
// LAST: the stream's last tile (a second copy of the code, so that the common one keeps its constant trip counts): its
// full parts go the same way -- the tail of every launch --; the part the input ends in, if it is not a full one, stores
// itself behind them (`tail_slow`), otherwise the last full part closes the stream.
template <bool LAST>
__device__ __forceinline__ bool store_tile_fast(const EncodeParams& p, Smem& sm, uint32_t n, int lane_in)
{
    const int lane = LAST ? lane_id_fresh() : lane_in;     // (the cold copy: nothing of it is an invariant of the store wave's loop)
    const uint32_t slot = n % kSlots;
    const uint32_t tile = uni(sm.tq[n % kTq]);
    int nfast = kDataWaves;
    bool tail_slow = false, stream_end = false;
    if (LAST) {
        const int np = parts_of_tile(p, tile);
        tail_slow  = !part_is_full(p, tile, np - 1);
        stream_end = !tail_slow;
        nfast      = tail_slow ? np - 1 : np;
        if (nfast == 0) return store_tile_generic(p, sm, n, lane);
    }
    const int wvl = lane < nfast ? lane : nfast - 1;
    const WaveRec& rec = sm.wrec[slot][wvl];
    const uint32_t Tl = rec.bits, basel = rec.base;
    uint32_t pk = __hip_atomic_load(&sm.packed[wvl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool plain = Tl >= 32u && ring_need(Tl) <= kWaveRing;
    if (!__all(plain)) return store_tile_generic(p, sm, n, lane);
    const uint64_t P  = uni64(sm.ts[slot].prefix);
    const uint32_t pt = uni(sm.ts[slot].pred_tail);
    if (__any(pk <= n)) {                 // every part packed?  (normally long ago)
        WaveWait w;
        while (__any(__hip_atomic_load(&sm.packed[wvl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= n))
            if (!w.step(p, sm, lane)) return false;
    }
#ifdef HVLC_ABL_SKIP_STORE
    return true;                                                            // (tools/: what the data waves alone can do; wrong output)
#endif
    const uint32_t startl = rec.start;                                      // valid once packed
    const uint64_t Gl = P + (uint64_t)basel;
    const uint64_t Gend = lane_value64(Gl + Tl, (uint32_t)(nfast - 1));
    // words of the stream up to the end of these parts; the stream's end also takes the open word, zero padded, and the
    // zero word cpu_vlc_encode leaves behind when the stream ends on a word boundary (cpuencode.cpp:39)
    uint64_t wend = Gend >> 5;
    if (LAST && stream_end) wend = ((Gend + 31u) >> 5) + ((Gend & 31u) == 0u ? 1u : 0u);
    if (wend > p.out_cap) return store_tile_generic(p, sm, n, lane);          // (capacity overrun: the general form reports it)
    uint32_t* const sbl = &sm.ring[wvl][startl];
    // the 32 stream bits behind every part (its own: it holds at least a word), handed to the part behind it as carry word
    uint32_t cout;
    {
        const uint32_t wi = Tl >> 5, sh = Tl & 31u;
        const uint32_t hi = sbl[wi], lo = sbl[wi + 1];
        cout = sh ? __builtin_amdgcn_alignbit(hi, lo, 32u - sh) : hi;
        uint32_t cin = (uint32_t)__shfl_up((int)cout, 1);
        if (lane == 0) cin = pt;
        if (lane < nfast) sbl[0] = cin;
    }
    // segment offsets: lane l = segment l of the tile
    if (p.offsets && lane < kSegsPerTile && !HVLC_DBG(p, kDbgNoOffsets)) {
        const WaveRec& r = sm.wrec[slot][lane / kRows];
        const uint32_t rr = (uint32_t)lane % (uint32_t)kRows;
        const uint32_t rel = r.base + (rr ? (&r.base)[rr] : 0u);
        const uint64_t seg = (uint64_t)tile * kSegsPerTile + (uint64_t)lane;
        if (!LAST || seg < p.num_segs) ABL_STORE(P + rel, &p.offsets[seg]);
    }
    if (HVLC_DBG(p, kDbgNoStore)) return true;
    // ---- the stream: positions in words relative to the tile's first word ow0
    const uint64_t ow0  = P >> 5;                                           // (base of part 0 is 0)
    uint32_t* const outp = p.out + ow0;
    const uint32_t mis0 = ((uint32_t)((uintptr_t)p.out >> 2) + (uint32_t)ow0) & (kBlockWords - 1u);   // block boundaries: r with (r + mis0) % 64 == 0
    const int32_t  rl   = (int32_t)(uint32_t)((Gl >> 5) - ow0);             // the part's first word
    const uint32_t sl   = (uint32_t)Gl & 31u;
    const int32_t  rend = (int32_t)(uint32_t)(wend - ow0);
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t sbaddr = (uint32_t)(uintptr_t)(lds_u32*)sbl;             // LDS byte address of the part's carry word
    auto lds_at = [](uint32_t byte_addr) { return (const uint32_t*)(lds_u32*)(uintptr_t)byte_addr; };

    // ---- software pipeline over the parts: the LDS reads of part w + 1 (the block it starts in -- a junction with part
    // w when w leaves its last block pending -- and up to eight whole blocks) are issued BEFORE the shifts and stores of
    // part w; left to itself the wave waits out an LDS round trip per part (ten per tile, ~800 clocks each under load).
    struct Batch {
        int32_t lo, hi, B0, e0, nblk, first; uint32_t sb, sh;               // scalars
        uint32_t a0, a1, a_sh;                                              // block 0 (per lane: it may be fed from two parts)
        uint32_t w[16];                                                     // whole blocks 1..8: word pairs
    };
    int32_t plo = 0; uint32_t psb = 0, psh = 0; bool pend = false;           // the part in front, while its last block is pending
    auto issue = [&](int wv, int32_t lo, Batch& q) {
        q.lo = lo;
        q.sb = (uint32_t)__builtin_amdgcn_readlane((int)sbaddr, wv);
        q.sh = (uint32_t)__builtin_amdgcn_readlane((int)sl, wv);
        q.hi = wv + 1 < nfast ? __builtin_amdgcn_readlane(rl, wv + 1) : rend;
        q.B0 = lo - (int32_t)(((uint32_t)lo + mis0) & (kBlockWords - 1u));  // block 0 = the block the part starts in: [B0, B0 + 64)
        q.e0 = q.hi < q.B0 + 64 ? q.hi : q.B0 + 64;
        q.nblk = q.hi >= q.B0 + 128 ? (q.hi - (q.B0 + 64)) / 64 : 0;        // whole blocks behind block 0
        q.first = pend ? q.B0 : lo;
        const int32_t k0 = q.B0 + lane;                                     // lane's word in block 0
        const bool isprev = pend && k0 < lo;
        const uint32_t* const pa = lds_at(isprev ? psb + (uint32_t)(k0 - plo) * 4u : q.sb + (uint32_t)(k0 - lo) * 4u);
        q.a_sh = isprev ? psh : q.sh;
        q.a0 = pa[0]; q.a1 = pa[1];
        const uint32_t* const pw = lds_at(q.sb + (uint32_t)(q.B0 + 64 - lo + lane) * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) { q.w[2 * j] = pw[64 * j]; q.w[2 * j + 1] = pw[64 * j + 1]; }
        if (q.nblk > 4) {
#pragma unroll
            for (int j = 4; j < 8; ++j) { q.w[2 * j] = pw[64 * j]; q.w[2 * j + 1] = pw[64 * j + 1]; }
        }
        // the part's last, partial block [Bl, hi) is left to the part behind it (a junction) when there is one and the
        // block lies behind block 0
        const int32_t Bl = q.hi - (int32_t)(((uint32_t)q.hi + mis0) & (kBlockWords - 1u));
        pend = wv + 1 < nfast && Bl > q.B0 && Bl < q.hi;
        plo = lo; psb = q.sb; psh = q.sh;
    };
    auto consume = [&](const Batch& c, bool more) {
        uint32_t* const dst = outp + c.B0;                                  // (scalar; part 0's B0 may lie below the tile: those lanes are masked)
        const int32_t k0 = c.B0 + lane;
        if (k0 >= c.first && k0 < c.e0) ABL_STORE(__builtin_amdgcn_alignbit(c.a0, c.a1, c.a_sh), dst + lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c.nblk > j) ABL_STORE(__builtin_amdgcn_alignbit(c.w[2 * j], c.w[2 * j + 1], c.sh), dst + 64 * (j + 1) + lane);
        if (c.nblk > 4) {
#pragma unroll
            for (int j = 4; j < 8; ++j)
                if (c.nblk > j) ABL_STORE(__builtin_amdgcn_alignbit(c.w[2 * j], c.w[2 * j + 1], c.sh), dst + 64 * (j + 1) + lane);
            if (c.nblk > 8) {                                               // (more than 2.3 KiB per part: 4.6 bits per symbol and up)
                const uint32_t* src = lds_at(c.sb + (uint32_t)(c.B0 + 576 - c.lo + lane) * 4u);
                uint32_t* d2 = dst + 576;
#pragma unroll 1
                for (int32_t i = 8; i < c.nblk; ++i) {
                    ABL_STORE(__builtin_amdgcn_alignbit(src[0], src[1], c.sh), d2 + lane);
                    src += 64; d2 += 64;
                }
            }
        }
        if (!more) {                                                        // the tile's last part stores its last, partial block itself
            const int32_t Bl = c.hi - (int32_t)(((uint32_t)c.hi + mis0) & (kBlockWords - 1u));
            if (Bl > c.B0 && Bl < c.hi) {
                const uint32_t* ps = lds_at(c.sb + (uint32_t)(Bl - c.lo + lane) * 4u);
                if (Bl + lane < c.hi) ABL_STORE(__builtin_amdgcn_alignbit(ps[0], ps[1], c.sh), outp + Bl + lane);
            }
        }
    };
    Batch A, B;                                                              // two register sets take turns
    issue(0, 0, A);
#pragma unroll 1
    for (int wv = 0; wv < nfast; wv += 2) {
        const bool second = wv + 1 < nfast;
        if (second) issue(wv + 1, A.hi, B);
        consume(A, second);
        if (!second) break;
        const bool more = wv + 2 < nfast;
        if (more) issue(wv + 2, B.hi, A);
        consume(B, more);
    }
    if (LAST && stream_end && lane == 0) {
        p.result()[0] = Gend;
        if (p.total_bits) *p.total_bits = Gend;
    }
    if (LAST && tail_slow) {
        // the part the input ends in stores itself (slow_part): give its wave the 31 bits in front of it, wait for it
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cout, nfast - 1);
        if (lane == 0) lds_rel(&sm.wrec[slot][nfast - 1].tail, kValid32 | (c & 0x7fffffffu));
        wake_workgroup();
        WaveWait w;
        while (!(lds_acq(&sm.wrec[slot][nfast].tail) & kValid32))
            if (!w.step(p, sm, lane)) return false;
    }
    return true;
}

__device__ __forceinline__ bool store_tile(const EncodeParams& p, Smem& sm, uint32_t n, int lane)
{
    if (uni(sm.tq[n % kTq]) == (uint32_t)(p.num_tiles - 1)) return store_tile_fast<true>(p, sm, n, lane);
    return store_tile_fast<false>(p, sm, n, lane);
}

__device__ __forceinline__ void encode_store_wave(const EncodeParams& p, Smem& sm, int lane)
{
    uint32_t n = 0, n_final = ~0u;
    WS_DECL;
    for (;;) {
        WaveWait w;
        for (;;) {
            if (lds_acq(&sm.n_resolved) > n) break;
            if (n_final == ~0u) n_final = lds_rlx(&sm.n_final);
            if (n_final != ~0u && n >= n_final) { WS_LAP(kWsWaitTicket); WS_DUMP(p, kDataWaves + 1, lane); return; }
            if (!w.step(p, sm, lane)) return;
        }
        WS_LAP(kWsWaitTicket);
        TL(p, uni(sm.tq[n % kTq]), 5, lane == 0);
        if (!store_tile(p, sm, n, lane)) return;
        TL(p, uni(sm.tq[n % kTq]), 6, lane == 0);
        WS_LAP(kWsStore); WS_COUNT(kWsTiles);
        ++n;
        if (lane == 0) lds_rel(&sm.n_freed, n);
        wake_workgroup();                                      // a data wave may be waiting for ring space or a slot
    }
}

// ------------------------------------------------------------------------------------------------
// SCOUT wave: everything that crosses workgroups.  Its decisions are scalar code: what a tile waits for is
// one 16-byte record (TileRec), polled as two 8-byte words by two lanes (an aligned 8-byte word read by one lane
// is one request: never a mix of two values); a leader's window of group granules rides in the same vector load.
//
// Offsets come from a two-level scan over the granules: the LAST tile of every 64-tile group is the
// group's leader.  Its scout (LeaderDuty) reads the 64 bit counts of the group (one vector load),
// publishes the group's aggregate, walks the group granules downwards (16 per round) to the nearest
// inclusive prefix, publishes the group's inclusive prefix and then the global offset of EVERY tile of the
// group (one 64-lane store into the records).  So a tile's offset costs its own scout one poll per round,
// and the chain published -> leader -> records is three hand-offs whatever the input size.
// The scout also draws the workgroup's tiles from the ticket counters.  It free-runs and hands results over
// through LDS.
//
// Tiles are handed out DYNAMICALLY: every tile a workgroup works on is drawn by its scout from one of
// kTicketClasses atomic counters (draw k of class c is tile 64k + (c + k) % 64, so the classes interleave, the
// group-leader role rotates over them and a workgroup never holds two tiles of one group).  A workgroup draws a
// tile only when it is about to work on it, so (1) no tile ever waits for a workgroup that is not running: forward
// progress does not depend on the grid being co-resident, and (2) a slow workgroup just draws fewer tiles instead of
// setting everybody's pace.  Tile numbers are limited to 28 bits by the host (32-bit record offsets).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxPoll = 6;        // records the scout polls per round (lanes 0..11, 8 bytes each)
constexpr int kWindow  = 16;       // group granules a leader looks at per round (lanes 32..47, 8 bytes each)
static_assert(kMaxPoll < kSlots, "a polled tile's slot must still belong to it");

template <bool WINDOW>
__device__ __forceinline__ uint64_t poll_round(const EncodeParams& p, uint32_t rec_off, const void* w, int lane)
{
    const char* src = (const char*)p.recs() + rec_off;
    if (WINDOW && lane >= 32) src = (const char*)w + (uint32_t)(lane - 32) * 8u;
    uint64_t v = 0;
    if (lane < 2 * kMaxPoll || (WINDOW && lane >= 32 && lane < 32 + kWindow)) v = ld_granule((const uint64_t*)src);
    return v;
}
__device__ __forceinline__ uint64_t lane_value64(uint64_t v, uint32_t l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l) << 32) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
}

struct LeaderDuty {
    bool     active = false, have_bits = false, win_done = false;
    uint32_t g = 0, n = 0;      // group, tiles in it
    uint32_t A = 0;             // aggregate of the group (64 tiles x < 2^20 bits)
    uint64_t E = 0;             // exclusive prefix of the group, collected so far
    int32_t  j = 0;             // next group granule to look at, walking down; -1: start of the stream
    uint32_t excl_in = 0;       // per lane: bits of the group's tiles before tile 64g+lane
};

__device__ __forceinline__ void encode_scout_wave(const EncodeParams& p, Smem& sm, int lane, uint32_t first_ticket, bool first_published)
{
    // lanes 2q, 2q+1 poll the record of the tile in poll slot q (workgroup-local tile n uses slot n % kMaxPoll):
    // byte offset from p.recs() of the 8 bytes this lane reads (record 0 until a slot is started)
    uint32_t rec_off = (uint32_t)(lane & 1) * 8u;
    uint32_t n_started = 0, n_done = 0; // tiles whose record is being polled / handed to the data waves
    uint32_t n_duty = 0;                // tiles already checked for the group-leader role
    LeaderDuty d;
    uint32_t rounds = 0;
    uint64_t t_progress = 0;
    const uint32_t ntiles = (uint32_t)p.num_tiles, last_tile = ntiles - 1u;
    // tickets: draw k of class cls is tile C*k + (cls + k) % C
    const uint32_t cls = blockIdx.x % (uint32_t)kTicketClasses;
    uint32_t n_tix = first_published ? 1u : 0u, n_final = ~0u;  // tiles drawn; the total once the tickets ran out
    bool     tix_inflight = !first_published;   // (the first draw was made in the kernel's prologue, and handed on there if it was a tile)
    uint32_t tix_k = first_ticket;      // lane 0: the value a ticket draw returned (valid one round later)

    while (n_done < n_final) {
        bool progress = false;

        // (0) the ticket drawn in the previous round has arrived
        if (tix_inflight) {
            const uint32_t k = uni(tix_k);
            tix_inflight = false;
            progress = true;
            if ((uint64_t)k * kTicketClasses >= (uint64_t)ntiles) {
                n_final = n_tix;
                if (lane == 0) __hip_atomic_store(&sm.n_final, n_final, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                wake_workgroup();
            } else {
                const uint32_t t = k * (uint32_t)kTicketClasses + ((cls + k) % (uint32_t)kTicketClasses);
                if (t < ntiles) {       // (a draw in the last, partial block may fall behind the end: draw again)
                    if (lane == 0) {
                        if (t == 0u) st_word(&p.recs()[0].tail, kValid32);      // nothing precedes tile 0: its predecessor's tail is empty
                        sm.tq[n_tix % kTq] = t;
                        lds_rel(&sm.n_tickets, n_tix + 1u);
                    }
                    wake_workgroup();
                    ++n_tix;
                }
            }
        }

        // (1) this round's requests: the leader's bit counts (vector, asynchronous), then the records of the
        //     tiles in flight and, for a leader, 16 group granules, in one vector load
        uint32_t bv = kValid32;
        if (d.active && !d.have_bits && (uint32_t)lane < d.n)
            bv = __hip_atomic_load(&p.bits32()[(uint64_t)d.g * kGroupTiles + (uint32_t)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t pv;                     // lanes 0-11: the records of the tiles in flight; lanes 32-47: the leader's window
        if (d.active && !d.win_done) {
            const int32_t c0 = d.j & ~(kWindow - 1);
            pv = poll_round<true>(p, rec_off, p.group_desc() + c0, lane);
            bool stop = false;
#pragma unroll
            for (int i = kWindow - 1; i >= 0; --i) {
                if (!stop && c0 + i <= d.j) {
                    const uint64_t v  = lane_value64(pv, 32u + (uint32_t)i);
                    const uint32_t st = (uint32_t)(v >> 62);
                    if (st == 0u) stop = true;                            // not published yet: again next round
                    else {
                        d.E += v & kValMask;
                        d.j  = c0 + i - 1;
                        progress = true;
                        if (st == 2u) { d.win_done = true; stop = true; }  // an inclusive prefix ends the walk
                    }
                }
            }
            if (!d.win_done && d.j < 0) d.win_done = true;                // walked down to the start of the stream
        } else {
            pv = poll_round<false>(p, rec_off, nullptr, lane);
        }

        // draw the next tile while the workgroup's SLOWEST data wave (a tile is published when its last wave has arrived)
        // is within kTixAhead tiles of the last one drawn: a wave that ran further ahead would only fill its ring with
        // parts that cannot be resolved before the slowest wave has arrived at their tiles, and the tiles of a group,
        // drawn within a microsecond of each other, would be published over many iterations.  The atomic is issued
        // BEHIND this round's polls (in-order return) and its value is used at the top of the next round
        if (n_final == ~0u && !tix_inflight && n_tix < lds_rlx(&sm.n_bits_pub) + lds_rlx(&sm.tix_ahead) &&
            n_tix - n_duty < (uint32_t)(kTq - 1)) {                 // (a tile number stays queued until the duty scan has seen it)
            if (lane == 0)
                tix_k = __hip_atomic_fetch_add(&p.tickets()[cls * (uint32_t)kTicketStride], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tix_inflight = true;
        }

        // (2) the tiles in flight, in order: resolved when the leader's offset and the predecessor's tail are both there
        {
            const uint64_t vP = __ballot((pv >> 63) != 0ull);                     // even lanes: P's valid bit
            const uint64_t vT = __ballot((((uint32_t)pv) >> 31) != 0u);           // odd lanes: the tail's valid bit
            uint32_t nd = n_done;
            bool go = true;
#pragma unroll
            for (int i = 0; i < kMaxPoll; ++i) {
                if (go && nd < n_started) {
                    const uint32_t q = nd % (uint32_t)kMaxPoll;
                    const bool ok = (((vP >> (2u * q)) & 1ull) != 0ull && ((vT >> (2u * q + 1u)) & 1ull) != 0ull) ||
                                    HVLC_DBG(p, kDbgNoLookback);
                    if (ok) {
                        uint64_t P  = lane_value64(pv, 2u * q) & kValMask;
                        uint32_t tl = (uint32_t)lane_value64(pv, 2u * q + 1u) & 0x7fffffffu;
                        if (HVLC_DBG(p, kDbgNoLookback)) { P = (uint64_t)uni(sm.tq[nd % kTq]) * 36400ull; tl = 0u; }
                        if (lane == 0) {
                            sm.ts[nd % kSlots].prefix    = P;
                            sm.ts[nd % kSlots].pred_tail = tl;
                        }
                        TL(p, uni(sm.tq[nd % kTq]), 4, lane == 0);
                        ++nd;
                    } else go = false;
                }
            }
            if (nd != n_done) {
                n_done = nd;
                progress = true;
                if (lane == 0) lds_rel(&sm.n_resolved, n_done);
                wake_workgroup();
            }
        }

        // (3) the leader's duty
        if (d.active && !d.have_bits) {
            if (__ballot((bv >> 31) != 0u) == ~0ull) {
                const uint32_t b    = (uint32_t)lane < d.n ? (bv & 0x7fffffffu) : 0u;
                const uint32_t incl = wave_incl_scan(b);
                d.excl_in   = incl - b;
                d.A         = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                d.have_bits = true;
                progress    = true;
                TL(p, (uint64_t)d.g * kGroupTiles + d.n - 1u, 2, lane == 0);
                if (lane == 0 && !d.win_done) st_granule(&p.group_desc()[d.g], kStAgg | (uint64_t)d.A);
            }
        }
        if (d.active && d.have_bits && d.win_done) {
            if (lane == 0) st_granule(&p.group_desc()[d.g], kStPre | (d.E + (uint64_t)d.A));   // later leaders wait for this
            if ((uint32_t)lane < d.n)
                st_granule(&p.recs()[(uint64_t)d.g * kGroupTiles + (uint32_t)lane].P, kValid | (d.E + (uint64_t)d.excl_in));
            TL(p, (uint64_t)d.g * kGroupTiles + d.n - 1u, 3, lane == 0);
            d.active = false;
            progress = true;

        }

        // (4) newly published tiles: start polling their records; the last tile of a group starts that group's duty
        const uint32_t nb = lds_acq(&sm.n_bits_pub);
        while (n_started < nb && n_started - n_done < (uint32_t)kMaxPoll) {
            const uint32_t q = n_started % (uint32_t)kMaxPoll;
            const uint32_t t = uni(sm.tq[n_started % kTq]);
            if ((uint32_t)(lane >> 1) == q) rec_off = t * (uint32_t)sizeof(TileRec) + (uint32_t)(lane & 1) * 8u;
            TL(p, t, 1, lane == 0);
            ++n_started;
            progress = true;
        }
        if (HVLC_DBG(p, kDbgNoLookback)) n_duty = nb;
        while (!d.active && n_duty < nb) {
            const uint32_t t = uni(sm.tq[n_duty % kTq]);
            ++n_duty;
            if ((t & (uint32_t)(kGroupTiles - 1)) == (uint32_t)(kGroupTiles - 1) || t == last_tile) {
                d.active = true; d.have_bits = false;
                d.g = t / kGroupTiles; d.n = (t & (uint32_t)(kGroupTiles - 1)) + 1u;
                d.E = 0; d.j = (int32_t)d.g - 1; d.win_done = (d.g == 0u);
                progress = true;
            }
        }

        if (lds_rlx(&sm.abort_flag)) return;
#ifdef HVLC_SCOUT_SLEEP
        if (!progress) __builtin_amdgcn_s_sleep(HVLC_SCOUT_SLEEP);
#endif
        // watchdog: nothing moved for timeout_ticks -> give up (and make every other workgroup give up)
        if (progress) { rounds = 0; t_progress = 0; }
        else {
            if (n_started == n_done && !d.active && !tix_inflight) __builtin_amdgcn_s_sleep(4);   // nothing in flight: wait for the data waves
            if ((++rounds & 255u) == 0u) {
                const uint64_t now = __builtin_amdgcn_s_memrealtime();
                const uint32_t st  = atomicOr(p.status(), 0u);  // RMW: coherent with the other workgroups' atomicOr
                if (t_progress == 0) t_progress = now;
                if ((st & kFlagTimeout) || now - t_progress > p.timeout_ticks) {
                    atomicOr(p.status(), kFlagTimeout);
                    if (lane == 0) __hip_atomic_store(&sm.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return;
                }
            }
        }
    }
}

// The persistent, pipelined encoder.  A workgroup = kDataWaves (10) DATA waves + 1 SCOUT wave + 1 STORE wave (three
// waves per SIMD whatever the placement).  It draws tiles from the ticket counters; every data wave keeps several packed
// parts in its LDS ring while the scout waits for their tiles' global bit offsets and the store wave writes resolved
// tiles, so HBM latency, the inter-workgroup hand-offs, the LDS bit packing and the stores of several tiles overlap
// inside a workgroup, on top of the overlap between the two workgroups of a CU.
#ifndef HVLC_WAVES_PER_EU
#define HVLC_WAVES_PER_EU 6
#endif
// LDS is passed as dynamic shared memory (sizeof(Smem)) so that the register allocator honours the
// waves-per-SIMD request below instead of the occupancy the static LDS size would imply: 80 VGPRs
// = 6 waves per SIMD = two 12-wave workgroups per CU.
__global__ void __launch_bounds__(kEncThreads) __attribute__((amdgpu_waves_per_eu(HVLC_WAVES_PER_EU, 8)))
k_encode(EncodeParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hvlc_smem_raw[];
    Smem& sm = *reinterpret_cast<Smem*>(hvlc_smem_raw);
    const int tid   = (int)threadIdx.x;
    const int lane  = tid & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which half of the workspace this launch works in: the context's call counter (stable while the launch runs:
    // only the last wave to leave increments it); zero this workgroup's slice of what the previous call used of the
    // OTHER half -- the next launch starts there
    uint32_t* const ctl = p.ctl();
    TL(p, 0, 7, blockIdx.x == 0 && tid == 0);                 // (experiment builds: workgroup 0 enters the kernel)
    // (the table's loads go out first: they do not depend on the control block, whose load everything else waits for)
    uint32_t len = 0, code = 0;
    if (tid < 256) { len = p.cl[tid]; code = p.cw[tid]; }
    {
        // (calls, exits, used[0], used[1]) in one load: written by the launch before this one, constant while this one runs
        // (the exits word aside, which nobody reads here)
        const uint4 c4 = *reinterpret_cast<const uint4*>(ctl + kCtlCalls);
        p.half = uni(c4.x) & 1u;
        const uint32_t nvec = uni(p.half ? c4.z : c4.w) / 16u;
        uint4* const z = reinterpret_cast<uint4*>(p.gran0 + (p.half ? 0u : p.half_bytes));
        for (uint32_t i = blockIdx.x * (uint32_t)kEncThreads + (uint32_t)tid; i < nvec; i += gridDim.x * (uint32_t)kEncThreads)
            z[i] = make_uint4(0, 0, 0, 0);
        if (blockIdx.x == 0 && tid == 0) {
            __hip_atomic_store(&ctl[kCtlUsed + p.half], p.used_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every access to a status word is an atomic RMW (set by atomicOr from any XCD, read back the same way): a
            // plain store was observed to leave a stale flag visible to the next launch
            (void)atomicExch(&ctl[p.half ^ 1u], 0u);
        }
    }

    // the scout draws the workgroup's first tile right away: the atomic is back by the time LDS is set up
    uint32_t first_ticket = 0;
    if (wave == kDataWaves && lane == 0)
        first_ticket = __hip_atomic_fetch_add(&p.tickets()[(blockIdx.x % (uint32_t)kTicketClasses) * (uint32_t)kTicketStride], 1u,
                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // lut_at() addresses the table absolutely: the workgroup's LDS object must start at LDS address 0
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const bool lds_ok = (uint32_t)reinterpret_cast<uintptr_t>((lds_byte*)hvlc_smem_raw) == 0u;
    if (!lds_ok) {
        if (tid == 0) atomicOr(p.status(), kFlagLayout);
    } else {
        // ---- the table, straight from the caller's two arrays: validated (workgroup 0 reports), masked, packed into
        // {codeword, length} pairs.  LDS layout: identity, or slot(s) = s ^ (s >> 4).  A ds_read_b64 serves a group of
        // 32 lanes in one cycle unless DIFFERENT entries fall into the same pair of banks (slot mod 32).  Expected
        // conflicting mass with symbol probabilities 2^-len: per bank pair, everything except its most probable entry;
        // when the identity layout conflicts no more than the permuted one, the hot loop saves two instructions per
        // word.  Every wave decides for itself, in registers: lane l looks at the four slots l + 64k of either layout
        // (lut_slot is an involution: slot t of the permuted layout holds symbol lut_slot(t)); lanes l and l ^ 32 share
        // a bank pair.
        if (tid < 256) {
            bool bad = false;
            if (len > 31u) { bad = true; len = 0u; }
            else if (len > 0u && (code >> len) != 0u) bad = true;    // cpuencode.cpp:35 precondition
            if (bad && blockIdx.x == 0) atomicOr(p.status(), kFlagTable);
            sm.len_tmp[tid] = (uint8_t)len;                          // (one load of the table per workgroup, not per wave)
        }
        TLX(p, 0, tid == 0);                                         // the table's loads are back
        __syncthreads();
        TLX(p, 1, tid == 0);
        uint32_t c0, c1;
        {
            uint32_t m0 = 0, t0 = 0, m1 = 0, t1 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t sl = (uint32_t)lane + 64u * (uint32_t)k;
                const uint32_t l0 = sm.len_tmp[sl], l1 = sm.len_tmp[lut_slot(sl)];
                const uint32_t w0 = (l0 >= 1u && l0 <= 24u) ? (1u << (24u - l0)) : 0u;
                const uint32_t w1 = (l1 >= 1u && l1 <= 24u) ? (1u << (24u - l1)) : 0u;
                m0 += w0; t0 = max(t0, w0); m1 += w1; t1 = max(t1, w1);
            }
            m0 += (uint32_t)__shfl_xor((int)m0, 32); t0 = max(t0, (uint32_t)__shfl_xor((int)t0, 32));
            m1 += (uint32_t)__shfl_xor((int)m1, 32); t1 = max(t1, (uint32_t)__shfl_xor((int)t1, 32));
            // (both halves of the wave now hold the same 32 sums: the totals count every bank pair twice)
            c0 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(m0 - t0), 63) / 2u;
            c1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(m1 - t1), 63) / 2u;
        }
        const bool permuted = !(c0 <= c1 || c0 <= (1u << 24) / 32u);
        if (tid < 256)
            sm.lut[permuted ? lut_slot((uint32_t)tid) : (uint32_t)tid] = make_uint2(len ? (code & ((1u << len) - 1u)) : 0u, len);
        {   // records and counters start as zero.  (Not the rings: a data wave zeroes the words it takes before it packs
            // into them, the slow path its scratch words -- 70 KiB less to write per workgroup and launch.)
            static_assert(offsetof(Smem, wrec) % 16 == 0 && offsetof(Smem, len_tmp) % 16 == 0, "zeroed as 16-byte vectors");
            uint4* z = reinterpret_cast<uint4*>(&sm.wrec[0][0]);
            const uint32_t nvec = (uint32_t)((offsetof(Smem, len_tmp) - offsetof(Smem, wrec)) / 16u);
            for (uint32_t i = (uint32_t)tid; i < nvec; i += (uint32_t)kEncThreads) z[i] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();                                  // the only workgroup-wide barriers (three): LDS initialised
        if (tid == 0) { sm.n_final = ~0u; sm.tix_ahead = (uint32_t)kTixAhead; }
        // the workgroup's first tile number goes to the data waves here, so that they find it at their first look (they
        // would sleep through the scout's first round otherwise: ~1 us at the start of every launch)
        bool first_published = false;
        if (wave == kDataWaves) {
            const uint32_t k   = uni(first_ticket);
            const uint32_t cls = blockIdx.x % (uint32_t)kTicketClasses;
            const uint32_t t   = k * (uint32_t)kTicketClasses + ((cls + k) % (uint32_t)kTicketClasses);
            if ((uint64_t)k * kTicketClasses < p.num_tiles && (uint64_t)t < p.num_tiles) {
                first_published = true;
                if (lane == 0) {
                    if (t == 0u) st_word(&p.recs()[0].tail, kValid32);          // nothing precedes tile 0: its predecessor's tail is empty
                    sm.tq[0] = t;
                    sm.n_tickets = 1u;
                }
            }
        }
        __syncthreads();
        TLX(p, 2, tid == 0);                                         // LDS is set up: the roles part
#ifdef HVLC_PRIO_YOUNG
        if (wave >= 4 && wave < kDataWaves) __builtin_amdgcn_s_setprio(HVLC_PRIO_YOUNG);
#endif
#ifdef HVLC_PRIO_TIERS
        // instruction issue on a SIMD goes to the oldest wave first: the data waves dispatched later lose to their elders
        if (wave >= 8 && wave < kDataWaves) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4 && wave < kDataWaves) __builtin_amdgcn_s_setprio(1);
#endif
        if (wave == kDataWaves) __builtin_amdgcn_s_setprio(HVLC_PRIO_SCOUT);
        if (wave == kDataWaves) {
            // workgroup 0's scout lives about as long as the launch: its lifetime in shader clocks and in 100 MHz ticks
            // tells the clock the GPU held (hvlc_get_clock_info)
            const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
            encode_scout_wave(p, sm, lane, first_ticket, first_published);
            if (blockIdx.x == 0 && lane == 0) {
                uint64_t* const clk = reinterpret_cast<uint64_t*>(ctl + kCtlClock);
                clk[0] = __builtin_amdgcn_s_memtime() - c0;
                clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
        }
        else if (wave == kDataWaves + 1) {
#ifndef HVLC_NO_STORE_PRIO
#ifndef HVLC_PRIO_STORE
#define HVLC_PRIO_STORE 3
#endif
            __builtin_amdgcn_s_setprio(HVLC_PRIO_STORE);       // one short, serial instruction stream that ten data waves wait for
#endif
            encode_store_wave(p, sm, lane);
        }
        else                             encode_data_wave(p, sm, lane, wave, permuted);
    }

    // ---- every wave of the grid leaves through here, finished or giving up; the last wave of a workgroup reports for all
    // eight (one atomic per workgroup on the same word, not eight), the last workgroup closes the call.  (Relaxed atomics:
    // the counters order nothing inside the launch -- an agent-scope release here is an L2 write-back per wave -- and
    // the end of the kernel publishes them to the next one.)
    if (lane == 0) {
        constexpr uint32_t kWaves = (uint32_t)(kEncThreads / 64);
        uint32_t report = 1u;                        // (no usable LDS: every wave for itself)
        if (lds_ok) report = __hip_atomic_fetch_add(&sm.n_exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u == kWaves ? kWaves : 0u;
        if (report) {
            const uint32_t left = __hip_atomic_fetch_add(&ctl[kCtlExits], report, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + report;
            if (left == gridDim.x * kWaves) {
                __hip_atomic_store(&ctl[kCtlExits], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(&ctl[kCtlCalls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The empty input: the stream is one zero word (cpuencode.cpp:17), nothing else runs.
__global__ void k_encode_empty(uint32_t* out, uint64_t* total)
{
    out[0] = 0u;
    if (total) *total = 0ull;
}

// The workspace (or a part of it) back to zero: at allocation, and hvlc_exclusive_scan's own area before every scan
// (its status word only ever sees atomic RMWs, like the encoder's).
__global__ void __launch_bounds__(kThreads) k_zero(uint4* area, uint64_t vecs, uint32_t* status, uint64_t* result)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < vecs; i += (uint64_t)gridDim.x * kThreads)
        area[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (status) { (void)atomicExch(&status[0], 0u); (void)atomicExch(&status[1], 0u); }   // (a pair: the encoder's two; the scan's second is unused)
        if (result) result[0] = 0ull;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage-level entry points of the reference pipeline, kept as separate kernels.
// ---------------------------------------------------------------------------------------------

// Per-segment bit counts (vlc_kernel_sm64huff.cu:119-122's outidx): one wave per segment.
__global__ void __launch_bounds__(kThreads) k_segment_bits(const uint32_t* in, uint64_t num_words,
                                                           const uint32_t* cl, uint32_t* counts, uint64_t num_segs)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t lens[256];
    lens[threadIdx.x] = cl[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwave = (uint64_t)gridDim.x * 4;
    const uint64_t full  = num_words / kSegWords;                  // segments that are whole
    auto bits_of = [&](uint32_t w) { return lens[w >> 24] + lens[(w >> 16) & 255u] + lens[(w >> 8) & 255u] + lens[w & 255u]; };
    // four whole segments per trip: the four coalesced 16-byte loads go out together (non-temporal: read once)
    uint64_t seg = wave0;
    for (; seg + 3 * nwave < full; seg += 4 * nwave) {
        u32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_nontemporal_load((const u32x4*)(in + (seg + r * nwave) * kSegWords) + lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t t = wave_sum_u32(bits_of(v[r].x) + bits_of(v[r].y) + bits_of(v[r].z) + bits_of(v[r].w));
            if (lane == 0) counts[seg + r * nwave] = t;
        }
    }
    for (; seg < num_segs; seg += nwave) {                          // what is left, and the segment the input ends in
        const uint64_t wbase = seg * kSegWords + (uint64_t)lane * 4u;
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (wbase + k < num_words) t += bits_of(in[wbase + k]);
        t = wave_sum_u32(t);
        if (lane == 0) counts[seg] = t;
    }
}

// Exclusive scan uint32 -> uint64, single launch, wavefront-hierarchical.  A workgroup pass covers kScanTile = 4096
// elements as four rows of 1024; thread t owns elements 4t..4t+3 of every row, so every load is one coalesced 16-byte
// vector per lane and every store two.  Per row: four elements summed in the lane, DPP scan across the wave (exact
// 64-bit sums from two 32-bit scans), the (row, wave) totals through LDS; workgroup totals through the same two-level
// look-back granules as the encoder (replaces scan.cu:67-231 + scanLargeArray_kernel.cu).
// Only tile_bits / group_desc / status / timeout_ticks / num_tiles of `p` are used.
__global__ void __launch_bounds__(kThreads) k_exclusive_scan(const uint32_t* in, uint64_t n, uint64_t* out,
                                                             uint64_t* total, EncodeParams p)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    static_assert(kScanTile == 16 * kThreads, "four rows of four elements per thread");
    __shared__ uint64_t wtot[4][4];                      // [row][wave]
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_abort;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_abort = 0u;
    for (uint64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const uint64_t tbase = tile * (uint64_t)kScanTile;
        const bool whole = tbase + kScanTile <= n;
        u32x4 v[4];
        uint64_t incl[4], sum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t e = tbase + (uint64_t)r * (4 * kThreads) + (uint64_t)tid * 4u;
            if (whole) v[r] = __builtin_nontemporal_load((const u32x4*)(in + e));
            else { v[r].x = e < n ? in[e] : 0u; v[r].y = e + 1 < n ? in[e + 1] : 0u; v[r].z = e + 2 < n ? in[e + 2] : 0u; v[r].w = e + 3 < n ? in[e + 3] : 0u; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sum[r] = (uint64_t)v[r].x + v[r].y + v[r].z + v[r].w;
            // exact 64-bit wave scan from two 32-bit DPP scans: sum < 2^34, so (sum >> 16) < 2^18 and
            // (sum & 0xffff) < 2^16 stay below 2^32 after adding 64 lanes
            const uint32_t a = wave_incl_scan((uint32_t)(sum[r] >> 16));
            const uint32_t b = wave_incl_scan((uint32_t)(sum[r] & 0xffffu));
            incl[r] = ((uint64_t)a << 16) + b;
        }
        __syncthreads();                                  // previous tile's wtot / s_prefix consumed
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wtot[r][wave] = incl[r];
        }
        __syncthreads();
        uint64_t T = 0, before[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w == wave) before[r] = T;             // everything in front of (row r, this wave)
                T += wtot[r][w];
            }
        }
        if (tid == 0 && tile != (uint64_t)p.stall_tile) st_granule(&p.tile_bits()[tile], kValid | (T & kValMask));   // (stall_tile: test hook)
        if (wave == 0) {
            uint64_t P = 0;
            const bool ok = lookback(p, tile, T, lane, P);
            if (lane == 0) { s_prefix = P; if (!ok) s_abort = 1u; }
        }
        __syncthreads();
        if (s_abort) return;
        const uint64_t P = s_prefix;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t e = tbase + (uint64_t)r * (4 * kThreads) + (uint64_t)tid * 4u;
            const uint64_t o0 = P + before[r] + (incl[r] - sum[r]), o1 = o0 + v[r].x, o2 = o1 + v[r].y, o3 = o2 + v[r].z;
            if (whole) {
                u64x2* dst = (u64x2*)(out + e);
                __builtin_nontemporal_store((u64x2){o0, o1}, dst); __builtin_nontemporal_store((u64x2){o2, o3}, dst + 1);
            } else {
                if (e < n) out[e] = o0;
                if (e + 1 < n) out[e + 1] = o1;
                if (e + 2 < n) out[e + 2] = o2;
                if (e + 3 < n) out[e + 3] = o3;
            }
        }
        if (tile == p.num_tiles - 1 && tid == 0) { p.result()[0] = P + T; if (total) *total = P + T; }
    }
}

__global__ void __launch_bounds__(kThreads) k_rebase(uint64_t* offs, uint64_t n, uint64_t base)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads)
        offs[i] += base;
}

// base = exclusive prefix of the all-gathered per-shard totals, computed by every thread (rank <= a few)
__global__ void __launch_bounds__(kThreads) k_rebase_dev(uint64_t* offs, uint64_t n, const uint64_t* totals, int rank)
{
    uint64_t base = 0;
    for (int r = 0; r < rank; ++r) base += totals[r];
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads)
        offs[i] += base;
}

// ---------------------------------------------------------------------------------------------
// SURVEY.md section 8(f) #1 -- seam merge: funnel-shift copy of a whole stream (a shard's) to an arbitrary bit
// position of the job's contiguous stream; pack2 (pack_kernels.cu:19-52) generalised from 256-word blocks to
// shards.  Destination word j takes bits from source words j-1 and j (v_alignbit_b32).  Interior words are
// stored; the first and last destination words, which the neighbouring shards may also write, are merged with
// an atomic OR (they must be zero before the first merge).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t merge_word(const uint32_t* __restrict__ src, uint64_t k, uint32_t s, uint64_t nd,
                                               uint64_t ns, uint64_t nbits)
{
    const uint32_t hi = (k >= 1 && k - 1 < ns) ? src[k - 1] : 0u;
    const uint32_t lo = k < ns ? src[k] : 0u;
    uint32_t v = s ? __builtin_amdgcn_alignbit(hi, lo, s) : lo;          // ({hi,lo} >> s): hi's low s bits, then lo's top bits
    if (k == nd - 1) {                                                    // bits past the end of the shard are not ours
        const uint32_t used = (uint32_t)((s + nbits) & 31u);
        if (used) v &= ~0u << (32u - used);
    }
    return v;
}

__global__ void __launch_bounds__(kThreads) k_merge_shard(const uint32_t* __restrict__ src, uint64_t nbits,
                                                          uint32_t* __restrict__ dst, uint64_t dst_bit_offset)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t s  = (uint32_t)dst_bit_offset & 31u;
    const uint64_t w0 = dst_bit_offset >> 5;
    const uint64_t nd = (s + nbits + 31u) >> 5;                   // destination words touched
    const uint64_t ns = (nbits + 31u) >> 5;                       // source words holding bits (zero padded by the encoder)
    // interior destination words in 16-byte aligned groups of four: relative k in [kg, kg + 4 * ng)
    uint64_t kg = ((w0 + 1 + 3) & ~3ull) - w0;                    // first group starts behind the (atomic) first word
    uint64_t ng = 0;
    if (nd >= 2 && kg + 4 <= nd - 1) {
        ng = (nd - 1 - kg) / 4;
        while (ng > 0 && kg + 4 * ng > ns) --ng;                  // every source word of a group must exist (k + 3 < ns)
    }
    if (ng == 0) kg = nd;                                         // no vector part: everything goes word by word
    const uint64_t tid0 = (uint64_t)blockIdx.x * kThreads + threadIdx.x, nthr = (uint64_t)gridDim.x * kThreads;
    for (uint64_t t = tid0; t < ng; t += nthr) {
        const uint64_t k = kg + 4 * t;
        const uint32_t hi = src[k - 1];
        const u32x4 lo = __builtin_nontemporal_load((const u32x4*)(src + k));      // 4-byte aligned is enough for global loads
        u32x4 v;
        v.x = s ? __builtin_amdgcn_alignbit(hi,   lo.x, s) : lo.x;
        v.y = s ? __builtin_amdgcn_alignbit(lo.x, lo.y, s) : lo.y;
        v.z = s ? __builtin_amdgcn_alignbit(lo.y, lo.z, s) : lo.z;
        v.w = s ? __builtin_amdgcn_alignbit(lo.z, lo.w, s) : lo.w;
        __builtin_nontemporal_store(v, (u32x4*)(dst + w0 + k));
    }
    // the words in front of and behind the vector part (the first and the last one of the shard: atomic)
    const uint64_t head = kg < nd ? kg : nd, tail0 = kg + 4 * ng;
    const uint64_t nscalar = head + (tail0 < nd ? nd - tail0 : 0);
    for (uint64_t t = tid0; t < nscalar; t += nthr) {
        const uint64_t k = t < head ? t : tail0 + (t - head);
        const uint32_t v = merge_word(src, k, s, nd, ns, nbits);
        // (system scope: the word may be peer memory that another GPU ORs its own seam bits into -- an agent-scope
        // atomic could be resolved in the issuing GPU's L2 and lose the other's bits when the line is written back)
        if (k == 0 || k == nd - 1) { if (v) (void)__hip_atomic_fetch_or(&dst[w0 + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
        else dst[w0 + k] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// SURVEY.md section 8(f) #3 -- decoder (the reference has none; it exists to prove round trips and to read the format
// back).  One LANE per 1024-byte segment, started at the segment's bit offset; a wave owns 64 consecutive segments
// (64 KiB of output, one contiguous span of the stream).
//   stream    every lane keeps the next words of ITS part of the span in a 16-dword ring in LDS (dword-major:
//             ring[slot][lane], conflict-free whatever slot each lane is at).  Every 16 symbols the wave tops the rings up:
//             a lane whose ring holds fewer than 8 words ahead loads the next aligned 16 bytes -- every stream byte is
//             loaded once, and neighbouring lanes' parts are adjacent, so the lines are shared through L2.
//   symbols   four symbols per 64-bit bit buffer (4 x kDecBits <= 64), straight-line and branch-free: codes of up to
//             kDecBits bits resolve with ONE LDS lookup of the next kDecBits stream bits; a longer code reads as length 0,
//             is noticed once per word, and that lane decodes the word again symbol by symbol (long codes: a list in
//             order of length, searched linearly)
//   output    16 words per lane are staged in LDS (odd row stride: conflict-free writes) and flushed with dwordx4 stores:
//             one instruction writes 64 contiguous bytes of each of 16 segments
// One 16-wave workgroup per CU shares the 16 KiB table.  Precondition: a prefix-free table; symbols with length 0 cannot
// be decoded.  An invalid stream or table yields wrong symbols, never an access outside the stream / output buffers.
// ---------------------------------------------------------------------------------------------
constexpr int kDecBits    = 13;                // first-level table: 8192 x 2 bytes
constexpr int kDecWaves   = 16;
constexpr int kDecThreads = 64 * kDecWaves;
constexpr int kDecTop     = 4;                 // words (x4 symbols) between two top-ups of the rings
constexpr int kDecChunk   = 16;                // words per lane between two flushes
constexpr int kDecRing    = 16;                // ring dwords per lane (+2 mirror slots: slots 16, 17 repeat slots 0, 1)
constexpr int kDecAhead   = 8;                 // words a ring must hold ahead of the read position at a top-up
constexpr int kDecIssue   = 12;                // ... below which the next 16 bytes are requested for the top-up after this one
constexpr int kDecStride  = 17;                // LDS words per lane in the output staging area
static_assert((31 + (kDecTop * 4 - 4) * kDecBits) / 32 + 2 < kDecAhead, "short codes cannot outrun the ring between two top-ups");
static_assert(kDecAhead + 3 + 4 <= kDecRing && kDecIssue + 4 <= kDecRing, "a top-up never overwrites words still to be read");

struct alignas(16) DecSmem {
    uint16_t fast[1 << kDecBits];              // (symbol << 8) | length, 0 = a longer code.  The length sits in the LOW byte: a
                                               // 64-bit shift takes its count from the low 6 bits of the entry as it stands
    uint32_t lcode[256];                       // long codes, left-aligned in 32 bits, in order of length
    uint8_t  llen[256], lsym[256];
    uint32_t nlong, pad_[3];
    uint32_t ring[kDecWaves][kDecRing + 2][64];    // stream words, dword-major; [16], [17] mirror [0], [1]: three consecutive
                                                   // words are three consecutive slots whatever the read position
    uint32_t stage[kDecWaves][64][kDecStride]; // decoded words
};
static_assert(sizeof(DecSmem) <= 160 * 1024, "one decoder workgroup per CU");

__device__ __forceinline__ uint32_t dec_word(const uint32_t* __restrict__ bits, uint64_t total_words, uint64_t wi)
{
    return wi < total_words ? bits[wi] : 0u;
}

// The hot loop, per output word (4 symbols): the 64 stream bits at the lane's read position from three consecutive
// ring slots (one ds_read2st64_b32 + one ds_read_b32; two 64-bit shifts), then four times {table index from the top 13
// bits (two instructions), ds_read_u16, v_perm (symbol into the word), 64-bit shift by the entry itself, SDWA add of its
// low byte to the position}: 8 vector instructions per symbol where round 2's loop took 15.7.
__global__ void __launch_bounds__(kDecThreads) k_decode(const uint32_t* __restrict__ bits, uint64_t total_bits,
                                                        const uint64_t* __restrict__ offsets, uint64_t num_segs,
                                                        const uint32_t* __restrict__ cw, const uint32_t* __restrict__ cl,
                                                        uint32_t* __restrict__ out, uint64_t num_words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dec_smem_raw[];
    DecSmem& sm = *reinterpret_cast<DecSmem*>(dec_smem_raw);
    // ---- tables.  Short codes: every table index that starts with the code gets (symbol << 8) | length; the indices
    // of one symbol are filled by all threads together (a 1-bit code owns half the table).  Long codes: a list ordered
    // by length (the likelier first), searched linearly.
    __shared__ uint32_t t_len[256], t_code[256];
    for (int i = threadIdx.x; i < (1 << kDecBits); i += kDecThreads) sm.fast[i] = 0;
    if (threadIdx.x < 256) {
        const uint32_t len = cl[threadIdx.x];
        t_len[threadIdx.x]  = len <= 31u ? len : 0u;
        t_code[threadIdx.x] = len >= 1 && len <= 31u ? (cw[threadIdx.x] & ((1u << len) - 1u)) : 0u;   // an invalid table must not index out of range
    }
    __syncthreads();
    for (int sy = 0; sy < 256; ++sy) {
        const uint32_t len = t_len[sy];
        if (len >= 1u && len <= (uint32_t)kDecBits) {
            const uint32_t base = t_code[sy] << (kDecBits - len), n = 1u << (kDecBits - len);
            for (uint32_t i = threadIdx.x; i < n; i += kDecThreads) sm.fast[base + i] = (uint16_t)(((uint32_t)sy << 8) | len);
        }
    }
    if (threadIdx.x < 256) {
        const uint32_t len = t_len[threadIdx.x];
        uint32_t before = 0, total = 0;
        for (int o = 0; o < 256; ++o) {
            const uint32_t lo = t_len[o];
            if (lo > (uint32_t)kDecBits) { ++total; if (lo < len || (lo == len && o < (int)threadIdx.x)) ++before; }
        }
        if (len > (uint32_t)kDecBits) {
            sm.lcode[before] = t_code[threadIdx.x] << (32u - len); sm.llen[before] = (uint8_t)len; sm.lsym[before] = (uint8_t)threadIdx.x;
        }
        if (threadIdx.x == 0) sm.nlong = total;
    }
    __syncthreads();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t total_words = (total_bits >> 5) + 1;
    const uint32_t nl = sm.nlong;
    uint32_t (*const ring)[64] = sm.ring[wave];
    uint32_t (*const my)[kDecStride] = sm.stage[wave];
    typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
    typedef __attribute__((address_space(3))) const uint16_t lds_cu16;
    const uint32_t ring_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&ring[0][lane];   // LDS byte address of slot 0
    const uint32_t fast_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)&sm.fast[0];
    // a symbol of a long code (or of an invalid stream: then symbol 0, one bit) from the 32 bits at the read position
    auto slow_symbol = [&](uint32_t win, uint32_t& sym, uint32_t& len) {
        const uint32_t e = sm.fast[win >> (32 - kDecBits)];
        sym = e >> 8; len = e & 255u;
        if (len == 0u) {
            len = 1u; sym = 0u;
            for (uint32_t q = 0; q < nl; ++q) {
                const uint32_t L = sm.llen[q];
                if (((win ^ sm.lcode[q]) >> (32u - L)) == 0u) { len = L; sym = sm.lsym[q]; break; }
            }
        }
    };
    // a wave owns 64 consecutive segments; all its lanes run the same number of chunks (a wave-uniform loop)
    for (uint64_t seg0 = ((uint64_t)blockIdx.x * kDecWaves + wave) * 64; seg0 < num_segs; seg0 += (uint64_t)gridDim.x * kDecWaves * 64) {
        const uint64_t seg = seg0 + (uint64_t)lane;
        const uint64_t pos = seg < num_segs ? offsets[seg] : 0ull;
        const uint64_t wbase = (pos >> 5) & ~3ull;   // ring word 0 (16-byte aligned in the stream); positions are relative to it
        uint32_t pb   = (uint32_t)(pos - wbase * 32u);   // the lane's read position in bits (a segment holds < 2^15 bits)
        uint32_t have = 0;                           // words [have - kDecRing, have) are in the ring (a multiple of 4)
        // the 64 stream bits at bit q, left-aligned: from the ring, or (a lane that long codes took past it) from memory
        auto bits64_at = [&](uint32_t q) -> uint64_t {
            const uint32_t kk = q >> 5, sh = q & 31u;
            uint32_t w0, w1, w2;
            if (kk + 2u < have) { w0 = ring[kk & (kDecRing - 1)][lane]; w1 = ring[(kk + 1) & (kDecRing - 1)][lane]; w2 = ring[(kk + 2) & (kDecRing - 1)][lane]; }
            else {
                w0 = dec_word(bits, total_words, wbase + kk); w1 = dec_word(bits, total_words, wbase + kk + 1);
                w2 = dec_word(bits, total_words, wbase + kk + 2);
            }
            return ((((uint64_t)w0 << 32) | (uint64_t)w1) << sh) | (((uint64_t)w2 << sh) >> 32);
        };
        // the ring is topped up ONE STEP AHEAD: the 16 bytes a lane will need are requested here and written into the ring
        // at the next top-up (`pend`), so the trip to memory is covered by the decoding of four words -- a wave that waits
        // for its load on the spot waits at almost every top-up, because some lane of its 64 always needs one
        u32x4 pv = {0u, 0u, 0u, 0u};
        bool pend = false;
        auto load16 = [&](uint32_t at) -> u32x4 {
            const uint64_t wi = wbase + at;
            u32x4 v;
            if (wi + 4 <= total_words) v = *(const u32x4*)(bits + wi);
            else { v.x = dec_word(bits, total_words, wi); v.y = dec_word(bits, total_words, wi + 1);
                   v.z = dec_word(bits, total_words, wi + 2); v.w = dec_word(bits, total_words, wi + 3); }
            return v;
        };
        auto commit16 = [&](const u32x4& v) {
            const uint32_t sl = have & (kDecRing - 1);
            ring[sl][lane] = v.x; ring[sl + 1][lane] = v.y; ring[sl + 2][lane] = v.z; ring[sl + 3][lane] = v.w;
            if (sl == 0u) { ring[kDecRing][lane] = v.x; ring[kDecRing + 1][lane] = v.y; }       // the mirror slots
            have += 4;
        };
        for (int chunk = 0; chunk < kSegWords / kDecChunk; ++chunk) {
#pragma unroll 1
            for (int top = 0; top < kDecChunk / kDecTop; ++top) {
                // ---- top up.  What was requested at the previous top-up goes into the ring; a lane whose ring will hold
                // fewer than kDecIssue words ahead requests its next aligned 16 bytes; a lane that even so holds fewer than
                // kDecAhead (the start of a segment; codes of 9 bits and more on average) gets them on the spot
                if (have + kDecRing < (pb >> 5)) { have = (pb >> 5) & ~3u; pend = false; }   // (long codes took the lane far ahead: skip what it will never read)
                if (pend) { commit16(pv); pend = false; }
                while (__any(have < (pb >> 5) + (uint32_t)kDecAhead)) {
                    if (have < (pb >> 5) + (uint32_t)kDecAhead) commit16(load16(have));
                }
                if (have < (pb >> 5) + (uint32_t)kDecIssue) { pv = load16(have); pend = true; }
                // (same lane, same wave: LDS operations complete in order, no barrier)
#pragma unroll
                for (int w = 0; w < kDecTop; ++w) {
                    // four symbols per bit buffer, straight-line and branch-free: a long code reads as length 0
                    const uint32_t p0 = pb, kk = pb >> 5, sh = pb & 31u;
                    lds_cu32* const rp = (lds_cu32*)(uintptr_t)(ring_lane + ((kk & (uint32_t)(kDecRing - 1)) << 8));
                    const uint32_t w0 = rp[0], w1 = rp[64], w2 = rp[128];
                    uint64_t buf = ((((uint64_t)w0 << 32) | (uint64_t)w1) << sh) & 0xffffffff00000000ull;
                    buf |= ((((uint64_t)w1 << 32) | (uint64_t)w2) << sh) >> 32;
                    uint32_t word = 0, e = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const uint32_t idx2 = ((uint32_t)(buf >> 32) >> (31 - kDecBits)) & ((2u << kDecBits) - 2u);   // 2 x the top 13 bits
                        e    = *(lds_cu16*)(uintptr_t)(fast_base + idx2);
                        word = __builtin_amdgcn_perm(word, e, 0x06050401u);   // (word << 8) | symbol: most significant byte first (cpuencode.cpp:28)
                        buf <<= (e & 63u);                                    // (lengths <= 13: the low 6 bits of the entry are the length)
                        pb += e & 255u;
                    }
                    // a long code leaves the buffer where it is, so every lookup behind it reads the same entry: the LAST
                    // entry is 0 exactly when any of the four was
                    const bool redo = e == 0u || kk + 2u >= have;            // a long code, or the lane is past its ring
                    if (__any(redo)) {
                        const bool far = redo && kk + 6u >= have;            // (even a redo from the ring could run out of it)
                        if (redo && !far) {          // cold: this lane decodes the word again, symbol by symbol, from its ring
                            pb = p0; word = 0;
#pragma unroll 1
                            for (int b = 0; b < 4; ++b) {
                                const uint32_t k2 = pb >> 5, s2 = pb & 31u;
                                const uint32_t x0 = ring[k2 & (kDecRing - 1)][lane], x1 = ring[(k2 + 1) & (kDecRing - 1)][lane];
                                const uint32_t win2 = (uint32_t)(((((uint64_t)x0 << 32) | (uint64_t)x1) << s2) >> 32);   // 32 bits: a whole code
                                uint32_t sym, len;
                                slow_symbol(win2, sym, len);
                                pb += len;
                                word = (word << 8) | sym;
                            }
                        }
                        if (__any(far)) {
                            if (far) {               // colder: the same with reads that may go to memory
                                pb = p0; word = 0;
#pragma unroll 1
                                for (int b = 0; b < 4; ++b) {
                                    uint32_t sym, len;
                                    slow_symbol((uint32_t)(bits64_at(pb) >> 32), sym, len);
                                    pb += len;
                                    word = (word << 8) | sym;
                                }
                            }
                        }
                    }
                    my[lane][top * kDecTop + w] = word;
                }
            }
            // flush: instruction t writes words 4(l%4)..4(l%4)+3 of segments 16t + l/4: 64 contiguous bytes per segment
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = 16 * t + (lane >> 2), q4 = (lane & 3) * 4;
                const uint4 v = make_uint4(my[j][q4], my[j][q4 + 1], my[j][q4 + 2], my[j][q4 + 3]);
                const uint64_t wo = (seg0 + (uint64_t)j) * kSegWords + (uint64_t)chunk * kDecChunk + (uint64_t)q4;
                if (wo + 3 < num_words) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)(out + wo));
                else {
                    if (wo     < num_words) out[wo]     = v.x;
                    if (wo + 1 < num_words) out[wo + 1] = v.y;
                    if (wo + 2 < num_words) out[wo + 2] = v.z;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Bandwidth probe (hvlc_probe_bandwidth; SURVEY.md section 8(d): "a measured streaming-read ceiling from a trivial
// dwordx4 read kernel in the same run").  No table, no bit work: what the memory system gives a kernel with the
// encoder's ACCESS PATTERN -- every wave reads 4 KiB (4 x dwordx4 nt per lane) per step, two steps in flight --
// and, for the mix, writes `wpt` words per 4 KiB read to a contiguous stream (the stream words of the step before,
// so that the stores never wait for the loads just issued).
//   STORE 0: read only   1: one dword per lane per store instruction   2: dwordx4 per lane
//         3, 4: the same two with non-temporal stores
//         5, 6: dwordx4 non-temporal stores in BURSTS: a wave writes the output of `burst` steps (10: a tile's worth, what
//               the encoder's store wave does; 32) as one contiguous stretch every `burst`-th step
//         8:    (k_probe_roles) the encoder's DIVISION OF LABOUR, in the geometry the memory system likes best (tools/
//               ubench_mem.hip, profiles/r03_ubench_roles.txt): ONE workgroup of 15 reader waves + 1 writer wave per CU; a
//               reader only reads (4 KiB steps, one -- shape 8 --, two -- shape 9 -- or three -- shape 10 -- in flight), the writer only writes -- the output of four
//               rounds of its workgroup as one stretch (dwordx4, non-temporal), paced by the readers through a counter in
//               LDS.  No wave has loads queued behind its own stores (vector memory operations of a wave return in order),
//               and few waves keep the queues short: 6.2 TB/s of bytes moved where the per-step shapes reach 5.1-5.4
//         11:   the same in the encoder's OWN geometry: two workgroups of 10 reader waves (one step in flight) + 1 writer wave
//               per CU -- the better shape on some boxes of the pool at low write shares (tools/ubench_mem.hip mode 4)
// k_probe_copy: a plain 1:1 copy (dwordx4 non-temporal loads and stores), the known-good reference of a read/write mix.
// ---------------------------------------------------------------------------------------------
template <int STORE>
__global__ void __launch_bounds__(kThreads) k_probe_rw(const uint4* __restrict__ in, uint64_t tiles4k, uint32_t* __restrict__ out,
                                                       uint32_t wpt, uint32_t burst, uint32_t* sink)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, w0 = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const u32x4* src = (const u32x4*)in;
    uint32_t acc = 0, prev = 0;
    u32x4 a, b, c, d;
    uint64_t t = w0, it = 0;
    auto store = [&](uint32_t* dst, uint32_t n) {
        if (STORE == 1)      { for (uint32_t k = lane; k < n; k += 64) dst[k] = prev; }
        else if (STORE == 2) { for (uint32_t k = lane * 4; k + 3 < n; k += 256) *(u32x4*)(dst + k) = (u32x4){prev, prev, prev, prev}; }
        else if (STORE == 3) { for (uint32_t k = lane; k < n; k += 64) __builtin_nontemporal_store(prev, dst + k); }
        else                 { for (uint32_t k = lane * 4; k + 3 < n; k += 256) __builtin_nontemporal_store((u32x4){prev, prev, prev, prev}, (u32x4*)(dst + k)); }
    };
    if (t < tiles4k) {
        const u32x4* q = src + t * 256u + lane;
        a = __builtin_nontemporal_load(q); b = __builtin_nontemporal_load(q + 64); c = __builtin_nontemporal_load(q + 128); d = __builtin_nontemporal_load(q + 192);
    }
    for (; t < tiles4k; t += nw, ++it) {
        // the stores of the PREVIOUS step first, then the next step's loads, then the wait for this step's data
        if (STORE && STORE < 5 && t != w0) store(out + (t - nw) * (uint64_t)wpt, wpt);
        if ((STORE == 5 || STORE == 6) && it && (it % burst) == 0u) store(out + ((it / burst - 1u) * nw + w0) * (uint64_t)(wpt * burst), wpt * burst);

        u32x4 na = a, nb = b, nc = c, nd = d;
        if (t + nw < tiles4k) {
            const u32x4* q = src + (t + nw) * 256u + lane;
            na = __builtin_nontemporal_load(q); nb = __builtin_nontemporal_load(q + 64); nc = __builtin_nontemporal_load(q + 128); nd = __builtin_nontemporal_load(q + 192);
        }
        prev = a.x ^ b.y ^ c.z ^ d.w;
        acc += prev;
        a = na; b = nb; c = nc; d = nd;
    }
    if (STORE && STORE < 5 && tiles4k > w0) {          // the last step's words
        const uint64_t tl = w0 + ((tiles4k - 1 - w0) / nw) * nw;
        store(out + tl * (uint64_t)wpt, wpt);
    }
    if ((STORE == 5 || STORE == 6) && it) {            // what is left of the last burst
        const uint64_t done = (it - 1u) / burst * burst, left = it - done;
        store(out + (((it - 1u) / burst) * nw + w0) * (uint64_t)(wpt * burst), (uint32_t)left * wpt);
    }
    if (acc == 0x12345678u) *sink = acc;
}

constexpr int kProbeReaders = 15, kProbeRounds = 4;        // k_probe_roles: reader waves per workgroup (+1 writer), rounds per writer burst
constexpr int kProbeReaders2 = 10;                          // ... in the encoder's own geometry (shape 11): two workgroups of 10 readers + 1 writer per CU
template <int DEPTH>                                        // 4 KiB steps a reader keeps in flight
__global__ void __launch_bounds__(64 * (kProbeReaders + 1)) k_probe_roles(const uint4* __restrict__ in, uint64_t tiles4k,
                                                                          uint32_t* __restrict__ out, uint32_t wpt, uint32_t* sink, int R)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr uint32_t rounds = kProbeRounds;
    __shared__ uint32_t steps_read;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) steps_read = 0u;
    __syncthreads();
    const uint64_t nr = (uint64_t)gridDim.x * R;                  // reader waves of the grid; the workgroup's round k: steps first + k * nr + {0..R-1}
    const uint64_t first = (uint64_t)blockIdx.x * R;
    if (wave < R) {
        const u32x4* src = (const u32x4*)in;
        uint32_t acc = 0;
        u32x4 r[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t t = first + (uint64_t)wave + (uint64_t)d * nr;
            if (t < tiles4k) {
                const u32x4* q = src + t * 256u + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j);
            }
        }
        for (uint64_t t = first + (uint64_t)wave; t < tiles4k; t += (uint64_t)DEPTH * nr) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint64_t td = t + (uint64_t)d * nr;
                if (td < tiles4k) {
                    acc += r[d][0].x ^ r[d][1].y ^ r[d][2].z ^ r[d][3].w;       // (waits for step td's data only)
                    if (lane == 0) __hip_atomic_fetch_add(&steps_read, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint64_t tn = td + (uint64_t)DEPTH * nr;
                    if (tn < tiles4k) {
                        const u32x4* q = src + tn * 256u + lane;
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j);
                    }
                }
            }
        }
        if (acc == 0x12345678u) *sink = acc;
    } else {
        // a burst = the output of `rounds` rounds of the workgroup, one stretch, in a region of its own (the rounds of a
        // workgroup are not adjacent in the stream)
        uint32_t k = 0;
        for (uint64_t t0 = first; t0 < tiles4k; t0 += nr * rounds, k += rounds) {
            uint32_t nst = 0;
            for (uint32_t j = 0; j < rounds; ++j) {
                const uint64_t tj = t0 + (uint64_t)j * nr;
                if (tj < tiles4k) nst += (uint32_t)(tiles4k - tj < (uint64_t)R ? tiles4k - tj : (uint64_t)R);
            }
            // wait until the readers have these rounds' data (the readers cannot fail)
            while (__hip_atomic_load(&steps_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)R * k + nst) __builtin_amdgcn_s_sleep(2);
            uint32_t* dst = out + ((uint64_t)(k / rounds) * gridDim.x + blockIdx.x) * (uint64_t)(R * rounds) * wpt;
            const uint32_t n = nst * wpt;
            for (uint32_t q = (uint32_t)lane * 4u; q + 3u < n; q += 256u) __builtin_nontemporal_store((u32x4){k, k, k, k + 1u}, (u32x4*)(dst + q));
        }
    }
}

__global__ void __launch_bounds__(kThreads) k_probe_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t nvec)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load((const u32x4*)in + i), (u32x4*)out + i);
}

// ---------------------------------------------------------------------------------------------
// Harness utilities: synthetic symbol generator and byte histogram.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ uint32_t cdf_pick(const uint32_t* cdf, uint32_t u)
{
    uint32_t lo = 0, hi = 255;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lo < hi) { if (cdf[mid] > u) hi = mid; else lo = mid + 1; }
    }
    return lo;
}

// one uint32 (4 symbols = 2 draws) per thread-iteration, stored in file order
__global__ void __launch_bounds__(kThreads) k_gen_symbols(uint32_t* out, uint64_t n_words, uint64_t tail_bytes,
                                                          uint64_t seed, uint64_t first_draw, const uint32_t* cdf32)
{
    __shared__ uint32_t cdf[256];
    cdf[threadIdx.x] = cdf32[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * kThreads) {
        const uint64_t r0 = splitmix64(seed + first_draw + 2 * i);
        const uint64_t r1 = splitmix64(seed + first_draw + 2 * i + 1);
        const uint32_t b0 = cdf_pick(cdf, (uint32_t)(r0 >> 32)), b1 = cdf_pick(cdf, (uint32_t)r0);
        const uint32_t b2 = cdf_pick(cdf, (uint32_t)(r1 >> 32)), b3 = cdf_pick(cdf, (uint32_t)r1);
        out[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    if (tail_bytes && blockIdx.x == 0 && threadIdx.x == 0) {      // n_bytes % 4 == 2
        const uint64_t r0 = splitmix64(seed + first_draw + 2 * n_words);
        uint8_t* o8 = (uint8_t*)(out + n_words);
        o8[0] = (uint8_t)cdf_pick(cdf, (uint32_t)(r0 >> 32));
        o8[1] = (uint8_t)cdf_pick(cdf, (uint32_t)r0);
    }
}

// Byte histogram, one HBM pass.  Skewed data makes most lanes of a wave hit the same few bins, and LDS atomics on
// one address serialise: the workgroup keeps kHistCopies copies of the histogram, lane % kHistCopies picks one (two
// lanes of a wave per copy), staggered by one word so that the same bin of the 32 copies falls into 32 different banks.
constexpr int kHistCopies = 32;
__global__ void __launch_bounds__(kThreads) k_histogram(const uint8_t* data, uint64_t n_bytes, unsigned long long* hist)
{
    __shared__ uint32_t h[kHistCopies][257];
    for (int i = threadIdx.x; i < kHistCopies * 257; i += kThreads) (&h[0][0])[i] = 0u;
    __syncthreads();
    uint32_t* mine = h[threadIdx.x & (kHistCopies - 1)];
    const uint64_t nvec = n_bytes / 16;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* v = (const u32x4*)data;
    uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    auto count16 = [&](const u32x4 x) {
        const uint32_t w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&mine[w[k] & 255u], 1u); atomicAdd(&mine[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 16) & 255u], 1u); atomicAdd(&mine[w[k] >> 24], 1u);
        }
    };
    // (host sizes the grid so a 32-bit bin cannot overflow)  Two vectors per trip, both loads in flight before the first
    // LDS atomic (one load per trip left the memory pipeline idle while the 16 atomics per lane drained: +7..15 %; four
    // per trip gain nothing more -- the LDS atomics bound it)
    for (; i + stride < nvec; i += 2 * stride) {
        const u32x4 x0 = __builtin_nontemporal_load(v + i), x1 = __builtin_nontemporal_load(v + i + stride);
        count16(x0); count16(x1);
    }
    for (; i < nvec; i += stride) count16(__builtin_nontemporal_load(v + i));
    if (blockIdx.x == 0)
        for (uint64_t j = nvec * 16 + threadIdx.x; j < n_bytes; j += kThreads) atomicAdd(&mine[data[j]], 1u);
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int c2 = 0; c2 < kHistCopies; ++c2) s += h[c2][threadIdx.x];
    if (s) atomicAdd(&hist[threadIdx.x], (unsigned long long)s);
}

}  // namespace HVLC_NS
