// hvlc_device.hip.h -- gfx950 (MI355X, wave64) device code of libhvlc.
//
// One fused, single-pass kernel replaces the reference's three GPU stages
// (vlc_encode_kernel_sm64huff, vlc_kernel_sm64huff.cu:37-160; prescanArray,
// scan.cu:228; pack2, pack_kernels.cu:19-52): the input is read from HBM once and
// every word of the final contiguous bit stream is written once, with a plain
// store, at its final position.  No padded intermediate copy, no memset of the
// output, no global atomics on the stream.
//
// Decomposition (all names follow the reference's domain):
//   segment  256 input words = 1024 symbols  = ONE WAVE-ROW (64 lanes x 16 bytes);
//            the reference's block (main_test_cu.cu:43); unit of out_offsets
//   tile     16 segments = 16 KiB            = one pass of a 256-thread workgroup;
//            wave w owns the tile's segments 4w..4w+3 (4 KiB contiguous)
//   group    64 tiles = 1 MiB                = unit of the second scan level
//
// Per tile:  (1) coalesced dwordx4 loads; (2) per-lane LDS table lookups and
// 64-bit codeword concatenation of 16 symbols; (3) wave64 DPP prefix scan of the
// bit lengths per segment, wave totals through LDS; (4) the tile's bit count is
// PUBLISHED (8-byte self-validating granule, relaxed agent-scope store);
// (5) "pabio": each lane ORs its <=64 bits into an LDS staging buffer with <=3
// ds_or_b32 at TILE-LOCAL bit positions (so packing does not wait for the global
// offset); (6) wave 0 resolves the tile's global bit offset by a two-level
// decoupled look-back (64 tile granules + 64 group granules per hop, all lanes of
// the wave in parallel) and fetches the <=31 carry bits of the predecessor;
// (7) the staging buffer is funnel-shifted (v_alignbit_b32) by offset%32 and
// stored with coalesced dword stores; segment offsets are written as uint64.
//
// Forward progress: the grid is PERSISTENT and sized to be co-resident (host side,
// hvlc_api.hip); tile t is processed by workgroup t % gridDim.x, tiles of one
// workgroup in increasing order, so every tile a wait depends on belongs to a
// running workgroup whatever the dispatch order or XCD placement.  Every wait is
// bounded (s_memrealtime) and ends in HVLC_ERR_TIMEOUT instead of a hang.
// Inter-workgroup data are single 8-byte granules carrying their own valid bit,
// written once and read with relaxed agent-scope (sc1) accesses -- the form the
// CDNA4 guide lists as needing no release/acquire fence -- and zeroed before each
// launch by k_prepare.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvlc {

constexpr int      kThreads     = 256;
constexpr int      kRows        = 4;                       // 16-byte rows per lane per tile
constexpr int      kTileWords   = kThreads * kRows * 4;    // 4096 words = 16 KiB
constexpr int      kTileVec     = kThreads * kRows;        // 1024 uint4
constexpr int      kSegWords    = 256;
constexpr int      kSegsPerTile = kTileWords / kSegWords;  // 16
constexpr int      kGroupTiles  = 64;
constexpr uint32_t kFastBits    = 65536;                   // bits one staging buffer holds (2048 words)
constexpr int      kStageWords  = 2064;                    // carry slot + 2048 (64 bit x 1024 lanes-rows) + pad, x2 buffers
constexpr uint64_t kValid       = 1ull << 63;
constexpr uint64_t kStAgg       = 1ull << 62;              // group granule: aggregate of the group
constexpr uint64_t kStPre       = 2ull << 62;              // group granule: inclusive prefix through the group
constexpr uint64_t kValMask     = (1ull << 62) - 1;

enum : uint32_t { kFlagTable = 1u, kFlagCapacity = 2u, kFlagTimeout = 4u };

struct EncodeParams {
    const uint4*    in;          // input words, 16-byte aligned
    uint64_t        num_words;
    uint64_t        num_tiles;
    uint64_t        num_segs;
    const uint2*    lut;         // {codeword (masked), length} x 256, built by k_prepare
    uint32_t*       out;         // final stream
    uint64_t        out_cap;     // words
    uint64_t*       offsets;     // per-segment exclusive bit offsets (nullable)
    uint64_t*       total_bits;  // user's device word (nullable)
    uint64_t*       result;      // workspace: [0] total bits
    uint32_t*       status;      // workspace: flag word
    uint64_t*       tile_bits;   // granule per tile: kValid | bits
    uint64_t*       tile_tail;   // granule per tile: kValid | last 31 bits of the stream through this tile
    uint64_t*       group_desc;  // granule per group: status | value
    uint64_t        timeout_ticks;   // s_memrealtime ticks (100 MHz)
};

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ld_granule(const uint64_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(uint64_t* p, uint64_t v)
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
struct Spin {
    uint64_t t0 = 0;
    uint32_t n  = 0;
    __device__ __forceinline__ bool wait(const EncodeParams& p)
    {
        __builtin_amdgcn_s_sleep(1);
        if ((++n & 1023u) == 0u) {
            const uint64_t now = __builtin_amdgcn_s_memrealtime();
            const uint32_t st  = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t0 == 0) t0 = now;
            if ((st & kFlagTimeout) || now - t0 > p.timeout_ticks) {
                atomicOr(p.status, kFlagTimeout);
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
// vlc_kernel_sm64huff.cu:48,70; here it is per 16 symbols and checked, with a general path).
// ---------------------------------------------------------------------------------------------
//
// The table sits in LDS at slot(s) = s ^ (s >> 4): symbol sets with a power-of-two stride (e.g. the
// 16 symbols 16j+3 of BASELINE config 3) would otherwise hit two of the 32 ds_read_b64 bank pairs
// and serialise 8-way.  hash4() applies the permutation to the four symbols of a word at once.
__device__ __forceinline__ uint32_t lut_slot(uint32_t s) { return s ^ (s >> 4); }
__device__ __forceinline__ uint32_t hash4(uint32_t w) { return w ^ ((w >> 4) & 0x0f0f0f0fu); }

template <bool FULL>
__device__ __forceinline__ void encode16(const uint4& v, int nvalid, const uint2* lut,
                                         uint64_t& acc, uint32_t& tot)
{
    const uint32_t w[4] = { hash4(v.x), hash4(v.y), hash4(v.z), hash4(v.w) };
    acc = 0; tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!FULL && k >= nvalid) break;
#pragma unroll
        for (int b = 3; b >= 0; --b) {
            const uint2 e = lut[(w[k] >> (8 * b)) & 255u];
            acc = (acc << e.y) | e.x;
            tot += e.y;
        }
    }
}

// A lane-row whose 16 codewords exceed 64 bits (lengths up to 31 allow 496): two symbols (<=62 bits)
// per put_bits64.  Rare with real tables, so it runs as a divergent branch of the fast path.
__device__ __forceinline__ void put_row_pairs(uint32_t* st, uint32_t q, const uint32_t (&wraw)[4], int nvalid,
                                              const uint2* lut)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nvalid) break;
        const uint32_t w = hash4(wraw[k]);
#pragma unroll
        for (int b = 3; b >= 1; b -= 2) {
            const uint2 e0 = lut[(w >> (8 * b)) & 255u];
            const uint2 e1 = lut[(w >> (8 * (b - 1))) & 255u];
            put_bits64(st, q, ((uint64_t)e0.x << e1.y) | e1.x, e0.y + e1.y);
            q += e0.y + e1.y;
        }
    }
}

// Per-tile state that lives from the iteration that packs a tile to the iteration that stores it.
struct TileState {
    uint64_t prefix;        // global exclusive bit offset (look-back result)
    uint32_t bits;          // T: bits of the tile
    uint32_t pred_tail;     // last 31 bits of the stream before the tile
    uint32_t flags;         // kTsEarlyTail | kTsPrefixReady
    uint32_t seg_rel[kSegsPerTile];   // tile-local bit offset of each segment
};
enum : uint32_t { kTsEarlyTail = 1u, kTsPrefixReady = 2u };

struct alignas(16) Smem {
    uint2     lut[256];                 // 2 KiB: {codeword, length}
    uint32_t  stage[2][kStageWords];    // double-buffered staging; [b][0] is the carry slot
    uint32_t  wave_tot[2][4];           // per-wave bit totals (bit 31: a lane-row exceeded 64 bits)
    TileState ts[2];
    uint32_t  qtot[4];                  // general path: per-wave totals of a quarter tile
    uint32_t  abort_flag;
};

// Granule values a look-back has in flight (wave 0 only).  Issued one iteration before they are
// consumed, so the round trip to the other XCDs' writes overlaps the next tile's encode.
struct LbRegs {
    uint64_t d = 0;     // tile_bits of predecessor tile-1-lane (inside the group)
    uint64_t v = 0;     // group_desc of group g-1-lane
    uint64_t t = 0;     // tile_tail of tile-1 (lane 0)
};

__device__ __forceinline__ void lookback_issue(const EncodeParams& p, uint64_t tile, int lane, LbRegs& L)
{
    const uint32_t r = (uint32_t)(tile & (kGroupTiles - 1));
    const int64_t  gi = (int64_t)(tile / kGroupTiles) - 1 - lane;
    L.d = (uint32_t)lane < r ? ld_granule(p.tile_bits + (tile - 1 - (uint64_t)lane)) : 0ull;
    L.v = gi >= 0 ? ld_granule(&p.group_desc[gi]) : 0ull;
    L.t = (lane == 0 && tile > 0) ? ld_granule(&p.tile_tail[tile - 1]) : 0ull;
}

// Two-level decoupled look-back, executed by ONE wave with all 64 lanes active.
// Level 1: lanes 0..r-1 read the bit counts of the r predecessors inside the tile's group.
// Level 2: lanes read the granules of up to 64 preceding groups; the nearest inclusive prefix
//          plus the aggregates after it give the group's exclusive prefix.  The last tile of a
//          group (the "leader") publishes the group's aggregate as soon as level 1 is done and its
//          inclusive prefix after level 2, so one hop covers 64 tiles / 64 groups.
// `L` may carry values loaded earlier by lookback_issue (or zeros: everything is polled here).
template <bool WIDE, bool TAIL>
__device__ __forceinline__ bool lookback(const EncodeParams& p, uint64_t tile, uint64_t T, int lane, LbRegs L,
                                         uint64_t& P_out, uint32_t& pred_tail)
{
    const uint32_t r = (uint32_t)(tile & (kGroupTiles - 1));
    const uint64_t g = tile / kGroupTiles;
    const bool leader = (r == kGroupTiles - 1) || (tile == p.num_tiles - 1);
    Spin spin;

    uint64_t d = L.d;
    const bool need = (uint32_t)lane < r;
    const uint64_t* src = p.tile_bits + (need ? tile - 1 - (uint64_t)lane : 0);
    for (;;) {
        if (!__any(need && !(d >> 63))) break;
        if (need && !(d >> 63)) d = ld_granule(src);
        if (!__any(need && !(d >> 63))) break;
        if (!spin.wait(p)) return false;
    }
    // encoder tiles hold < 2^20 bits each (32-bit sum is exact); the scan's tiles may hold 62 bits
    const uint64_t insum = WIDE ? wave_sum_u64(need ? (d & kValMask) : 0ull)
                                : (uint64_t)wave_sum_u32(need ? (uint32_t)d : 0u);
    if (leader && lane == 0) st_granule(&p.group_desc[g], kStAgg | ((insum + T) & kValMask));

    uint64_t excl = 0;
    bool first = true;
    for (int64_t gb = (int64_t)g; gb > 0; gb -= 64) {
        const int64_t gi = gb - 1 - lane;
        uint64_t v = gi < 0 ? kStPre : (first ? L.v : 0ull);   // before the stream: inclusive prefix 0
        first = false;
        int fp;
        for (;;) {
            const uint64_t pm = __ballot((v >> 62) == 2);
            fp = pm ? __builtin_ctzll(pm) : 64;
            const uint64_t xm  = __ballot((v >> 62) == 0);
            const uint64_t low = fp >= 63 ? ~0ull : ((2ull << fp) - 1ull);
            if ((xm & low) == 0) break;
            if (gi >= 0 && (v >> 62) == 0) v = ld_granule(&p.group_desc[gi]);
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
    if (leader && lane == 0) st_granule(&p.group_desc[g], kStPre | ((P + T) & kValMask));

    uint32_t pt = 0;
    if (TAIL && tile > 0) {
        uint64_t t = L.t;
        for (;;) {
            if (!__any(lane == 0 && !(t >> 63))) break;
            if (lane == 0 && !(t >> 63)) t = ld_granule(&p.tile_tail[tile - 1]);
            if (!__any(lane == 0 && !(t >> 63))) break;
            if (!spin.wait(p)) return false;
        }
        pt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t) & 0x7fffffffu;
    }
    P_out = P;
    pred_tail = pt;
    return true;
}

// Funnel-shift the staging buffer by s = (global bit offset % 32) and store `nown` finished words
// at out[ow..].  sb[0] is the carry slot (low s bits = last s bits of the preceding stream).
__device__ __forceinline__ void store_words(const EncodeParams& p, const uint32_t* sb, uint32_t s,
                                            uint64_t ow, uint32_t nown, int tid)
{
    for (uint32_t k = (uint32_t)tid; k < nown; k += kThreads) {
        const uint32_t v = __builtin_amdgcn_alignbit(sb[k], sb[k + 1], s);
        const uint64_t idx = ow + k;
        if (idx < p.out_cap) p.out[idx] = v;
        else atomicOr(p.status, kFlagCapacity);
    }
}

__device__ __forceinline__ void zero_words(uint32_t* base, uint32_t n_vec4, int tid)
{
    uint4* z = (uint4*)base;
    for (uint32_t i = (uint32_t)tid; i < n_vec4; i += kThreads) z[i] = make_uint4(0, 0, 0, 0);
}

// Loads of one tile: lane-row r of wave w = segment 4w+r, 16 bytes per lane, 1 KiB per wave-instruction.
__device__ __forceinline__ void load_tile(const EncodeParams& p, uint64_t tile, int wave, int lane, uint4 (&raw)[kRows])
{
    const bool full = (tile + 1) * (uint64_t)kTileWords <= p.num_words;
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const uint32_t vi = (uint32_t)((wave * kRows + r) * 64 + lane);     // uint4 index inside the tile
        if (full) {
            {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = __builtin_nontemporal_load((const u32x4*)&p.in[tile * kTileVec + vi]);
                raw[r] = make_uint4(t.x, t.y, t.z, t.w);
            }
        } else {
            const uint64_t wbase = tile * (uint64_t)kTileWords + (uint64_t)vi * 4u;
            const uint32_t* in32 = (const uint32_t*)p.in;
            raw[r].x = wbase + 0 < p.num_words ? in32[wbase + 0] : 0u;
            raw[r].y = wbase + 1 < p.num_words ? in32[wbase + 1] : 0u;
            raw[r].z = wbase + 2 < p.num_words ? in32[wbase + 2] : 0u;
            raw[r].w = wbase + 3 < p.num_words ? in32[wbase + 3] : 0u;
        }
    }
}

__device__ __forceinline__ int valid_words(const EncodeParams& p, uint64_t wbase)
{
    return wbase >= p.num_words ? 0 : (p.num_words - wbase >= 4 ? 4 : (int)(p.num_words - wbase));
}

// Store the finished words of a packed tile (fast path) and everything that needs its global
// offset: segment offsets, the total on the last tile, the tail granule when it could not be
// published early.
__device__ __forceinline__ void finish_tile(const EncodeParams& p, Smem& sm, uint64_t tile, uint32_t par, int tid)
{
    const TileState& ts = sm.ts[par];
    const uint64_t P = ts.prefix;
    const uint32_t T = ts.bits;
    const bool last = (tile == p.num_tiles - 1);
    const uint32_t* sb = sm.stage[par];
    if (p.offsets && tid < kSegsPerTile) {
        const uint64_t seg = tile * kSegsPerTile + (uint64_t)tid;
        if (seg < p.num_segs) p.offsets[seg] = P + ts.seg_rel[tid];
    }
    if (last && tid == 0) {
        p.result[0] = P + T;
        if (p.total_bits) *p.total_bits = P + T;
    }
    // the last tile also stores the open word, zero padded, and the zero word cpu_vlc_encode leaves
    // behind when the stream ends on a word boundary (cpuencode.cpp:39)
    const uint32_t s = (uint32_t)P & 31u;
    uint32_t nown = (s + T) >> 5;
    if (last) nown = ((s + T + 31u) >> 5) + ((((s + T) & 31u) == 0u) ? 1u : 0u);
    store_words(p, sb, s, P >> 5, nown, tid);
    if (!(ts.flags & kTsEarlyTail) && tid == 0) {
        const uint32_t own  = extract_tail(sb + 1, T);
        const uint32_t tail = T >= 31u ? own : (((ts.pred_tail << T) | own) & 0x7fffffffu);
        st_granule(&p.tile_tail[tile], kValid | (uint64_t)tail);
    }
}

// General path for one tile (some lane-row holds more than 64 bits; lengths up to 31 allow 496):
// the tile is re-read (L2) and packed quarter by quarter (4 KiB = wave-contiguous segments 4q..4q+3,
// at most 3968 words) into both staging buffers taken as one, two symbols (<=62 bits) per
// put_bits64.  The carry runs through the quarters exactly as it runs through tiles.  Blocking:
// the caller has no other tile in flight.  Leaves both staging buffers zeroed.
__device__ __forceinline__ void general_tile(const EncodeParams& p, Smem& sm, uint64_t tile, uint32_t par, int tid,
                                             int lane, int wave)
{
    const TileState& ts = sm.ts[par];
    const bool last = (tile == p.num_tiles - 1);
    const uint64_t word0 = tile * (uint64_t)kTileWords;
    uint32_t* const big = &sm.stage[0][0];
    uint64_t Pq    = ts.prefix;
    uint32_t carry = ts.pred_tail;
    const uint32_t T = ts.bits;
    if (p.offsets && tid < kSegsPerTile) {
        const uint64_t seg = tile * kSegsPerTile + (uint64_t)tid;
        if (seg < p.num_segs) p.offsets[seg] = Pq + ts.seg_rel[tid];
    }
    if (last && tid == 0) {
        p.result[0] = Pq + T;
        if (p.total_bits) *p.total_bits = Pq + T;
    }
    for (int qd = 0; qd < 4; ++qd) {
        __syncthreads();                            // previous quarter fully stored
        zero_words(big, 2u * kStageWords / 4u, tid);
        const uint64_t wbase = word0 + (uint64_t)qd * 1024u + (uint64_t)tid * 4u;
        const uint32_t* in32 = (const uint32_t*)p.in;
        const int n = valid_words(p, wbase);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = k < n ? in32[wbase + k] : 0u;
        uint32_t tq = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) {
                const uint32_t h = hash4(w[k]);
#pragma unroll
                for (int b = 3; b >= 0; --b) tq += sm.lut[(h >> (8 * b)) & 255u].y;
            }
        const uint32_t incl = wave_incl_scan(tq);
        if (lane == 63) sm.qtot[wave] = incl;
        __syncthreads();
        const uint32_t q0 = sm.qtot[0], q1 = sm.qtot[1], q2 = sm.qtot[2], q3 = sm.qtot[3];
        const uint32_t Tq = q0 + q1 + q2 + q3;
        const uint32_t pos = (incl - tq) + (wave == 0 ? 0u : wave == 1 ? q0 : wave == 2 ? q0 + q1 : q0 + q1 + q2);
        if (tid == 0) big[0] = carry;
        put_row_pairs(big + 1, pos, w, n, sm.lut);
        __syncthreads();
        const uint32_t s = (uint32_t)Pq & 31u;
        const bool lastq = last && qd == 3;
        uint32_t nown = (s + Tq) >> 5;
        if (lastq) nown = ((s + Tq + 31u) >> 5) + ((((s + Tq) & 31u) == 0u) ? 1u : 0u);
        store_words(p, big, s, Pq >> 5, nown, tid);
        const uint32_t own = extract_tail(big + 1, Tq);     // every thread computes the same carry
        carry = Tq >= 31u ? own : (((carry << Tq) | own) & 0x7fffffffu);
        Pq += Tq;
    }
    __syncthreads();
    if (tid == 0) st_granule(&p.tile_tail[tile], kValid | (uint64_t)carry);
    zero_words(big, 2u * kStageWords / 4u, tid);
}

// The persistent, software-pipelined encoder.  One loop iteration of a workgroup:
//   phase X  issue the loads of its NEXT tile; look up / concatenate / scan the CURRENT tile
//            (registers only); wave 0 resolves the PREVIOUS tile's global offset from the granules
//            it requested one iteration ago; zero the staging buffer the current tile will use
//   -- barrier 1 --
//   phase Y  publish the current tile's bit count; store the previous tile (funnel shift from its
//            staging buffer); pack the current tile into the other buffer; publish its tail;
//            wave 0 requests the current tile's look-back granules (the last tile of a group
//            resolves and publishes its group's prefix right away instead)
//   -- barrier 2 --
// so HBM latency, the granule round trip and the LDS bit packing of three different tiles overlap
// inside one workgroup, on top of the overlap between the workgroups of a CU.
__global__ void __launch_bounds__(kThreads) k_encode(EncodeParams p)
{
    __shared__ Smem sm;
    const int tid  = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    sm.lut[lut_slot((uint32_t)tid)] = p.lut[tid];
    zero_words(&sm.stage[0][0], 2u * kStageWords / 4u, tid);
    if (tid == 0) { sm.abort_flag = 0u; sm.ts[0].bits = 0u; sm.ts[1].bits = 0u; }

    uint64_t cur = blockIdx.x;
    if (cur >= p.num_tiles) return;
    const uint64_t stride = gridDim.x;
    uint4 raw[kRows];
    load_tile(p, cur, wave, lane, raw);
    __syncthreads();

    bool     has_prev = false;
    uint64_t prev = 0;
    uint32_t par = 0;                    // staging buffer / TileState slot of `cur`
    LbRegs   lb;                         // wave 0: granules requested for `prev`

    for (;;) {
        const uint64_t nxt = cur + stride;
        const bool has_nxt = nxt < p.num_tiles;
        const bool full = (cur + 1) * (uint64_t)kTileWords <= p.num_words;

        // ---------------- phase X ----------------
        uint64_t acc[kRows];
        uint32_t tot[kRows], excl[kRows], rowbase[kRows];
        uint32_t wavetot = 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            if (full) {
                encode16<true>(raw[r], 4, sm.lut, acc[r], tot[r]);
            } else {
                const uint64_t wbase = cur * (uint64_t)kTileWords + (uint64_t)((wave * kRows + r) * 64 + lane) * 4u;
                encode16<false>(raw[r], valid_words(p, wbase), sm.lut, acc[r], tot[r]);
            }
            const uint32_t incl = wave_incl_scan(tot[r]);
            excl[r]    = incl - tot[r];
            rowbase[r] = wavetot;
            wavetot   += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) sm.wave_tot[par][wave] = wavetot;
        if (has_nxt) load_tile(p, nxt, wave, lane, raw);      // raw is dead: its registers take the next tile

        if (has_prev && wave == 0 && !(sm.ts[par ^ 1u].flags & kTsPrefixReady)) {
            uint64_t P = 0; uint32_t pt = 0;
            const bool ok = lookback<false, true>(p, prev, (uint64_t)sm.ts[par ^ 1u].bits, lane, lb, P, pt);
            if (lane == 0) {
                sm.ts[par ^ 1u].prefix    = P;
                sm.ts[par ^ 1u].pred_tail = pt;
                sm.stage[par ^ 1u][0]     = pt;           // carry slot of prev's staging buffer
                if (!ok) sm.abort_flag = 1u;
            }
        }
        // the buffer `cur` will be packed into was last used two tiles ago (stored one iteration ago)
        zero_words(sm.stage[par], (sm.ts[par].bits >> 7) + 2u, tid);
        __syncthreads();                                                      // barrier 1
        if (sm.abort_flag) return;

        // ---------------- phase Y ----------------
        const uint32_t c0 = sm.wave_tot[par][0], c1 = sm.wave_tot[par][1],
                       c2 = sm.wave_tot[par][2], c3 = sm.wave_tot[par][3];
        const uint32_t T = c0 + c1 + c2 + c3;
        const bool slow = T > kFastBits;                 // does not fit one staging buffer: general path
        const uint32_t wavebase = wave == 0 ? 0u : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
        const bool early_tail = !slow && c3 >= 31u;     // the tile's last 31 bits are all written by wave 3
        const uint32_t rtile = (uint32_t)(cur & (kGroupTiles - 1));
        const bool leader = (rtile == kGroupTiles - 1) || (cur == p.num_tiles - 1);

        if (tid == 0) {
            st_granule(&p.tile_bits[cur], kValid | (uint64_t)T);   // publish before anything else
            sm.ts[par].bits  = T;
            sm.ts[par].flags = early_tail ? kTsEarlyTail : 0u;
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) sm.ts[par].seg_rel[wave * kRows + r] = wavebase + rowbase[r];
        }

        if (has_prev) finish_tile(p, sm, prev, par ^ 1u, tid);

        if (!slow) {
            uint32_t* const st = sm.stage[par] + 1;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const uint32_t q = wavebase + rowbase[r] + excl[r];
                if (tot[r] <= 64u) {
                    put_bits64(st, q, acc[r], tot[r]);
                } else {                                   // rare: re-read this lane's 16 bytes (L2) and go pairwise
                    const uint64_t wbase = cur * (uint64_t)kTileWords + (uint64_t)((wave * kRows + r) * 64 + lane) * 4u;
                    const uint32_t* in32 = (const uint32_t*)p.in;
                    const int n = valid_words(p, wbase);
                    uint32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = k < n ? in32[wbase + k] : 0u;
                    put_row_pairs(st, q, w, n, sm.lut);
                }
            }
            if (early_tail && wave == 3 && lane == 0)   // same-wave LDS ops complete in order
                st_granule(&p.tile_tail[cur], kValid | (uint64_t)extract_tail(st, T));
        }

        if (wave == 0) {
            if (leader || slow) {
                // resolve now: a group leader publishes its group's aggregate / prefix without waiting an
                // iteration; the general path needs its offset before it can store anything
                uint64_t P = 0; uint32_t pt = 0;
                const bool ok = lookback<false, true>(p, cur, (uint64_t)T, lane, LbRegs(), P, pt);
                if (lane == 0) {
                    sm.ts[par].prefix    = P;
                    sm.ts[par].pred_tail = pt;
                    sm.ts[par].flags    |= kTsPrefixReady;
                    sm.stage[par][0]     = pt;
                    if (!ok) sm.abort_flag = 1u;
                }
            } else {
                lookback_issue(p, cur, lane, lb);
            }
        }
        __syncthreads();                                                      // barrier 2
        if (sm.abort_flag) return;

        if (slow) {
            general_tile(p, sm, cur, par, tid, lane, wave);
            if (tid == 0) { sm.ts[0].bits = 0u; sm.ts[1].bits = 0u; }   // both buffers are clean again
            has_prev = false;
        } else {
            has_prev = true;
            prev = cur;
        }
        if (!has_nxt) break;
        cur = nxt;
        par ^= 1u;
        if (slow) __syncthreads();       // general_tile's final zeroing / ts reset before the next phase X
    }

    if (has_prev) {                      // drain: the last packed tile of this workgroup
        const uint32_t pp = par;         // `par` was not flipped after the final iteration
        if (wave == 0 && !(sm.ts[pp].flags & kTsPrefixReady)) {
            uint64_t P = 0; uint32_t pt = 0;
            const bool ok = lookback<false, true>(p, prev, (uint64_t)sm.ts[pp].bits, lane, lb, P, pt);
            if (lane == 0) {
                sm.ts[pp].prefix    = P;
                sm.ts[pp].pred_tail = pt;
                sm.stage[pp][0]     = pt;
                if (!ok) sm.abort_flag = 1u;
            }
        }
        __syncthreads();
        if (sm.abort_flag) return;
        finish_tile(p, sm, prev, pp, tid);
    }
}

// ---------------------------------------------------------------------------------------------
// k_prepare: block 0 validates the table and packs it into {codeword, length} pairs; all blocks
// zero the granule arrays (every polled word is re-initialised on every call).  With
// num_words == 0 it also writes the empty stream (one zero word, cpuencode.cpp:17).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_prepare(const uint32_t* cw, const uint32_t* cl, uint2* lut,
                                                      uint32_t* status, uint64_t* result,
                                                      uint4* granules, uint64_t granule_vecs,
                                                      uint32_t* out_if_empty, uint64_t out_cap,
                                                      uint64_t* total_if_empty, int empty)
{
    if (blockIdx.x == 0) {
        __shared__ uint32_t bad;
        if (threadIdx.x == 0) bad = 0u;
        __syncthreads();
        uint32_t len = cl[threadIdx.x], c = cw[threadIdx.x];
        if (len > 31u) { atomicOr(&bad, kFlagTable); len = 0u; }
        else if (len > 0u && (c >> len) != 0u) atomicOr(&bad, kFlagTable);   // cpuencode.cpp:35 precondition
        lut[threadIdx.x] = make_uint2(len ? (c & ((1u << len) - 1u)) : 0u, len);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t f = bad;
            result[0] = 0ull;
            if (empty) {
                if (out_cap >= 1) out_if_empty[0] = 0u; else f |= kFlagCapacity;
                if (total_if_empty) *total_if_empty = 0ull;
            }
            status[0] = f;
        }
    }
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < granule_vecs; i += (uint64_t)gridDim.x * kThreads)
        granules[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Stage-level entry points of the reference pipeline, kept as separate kernels.
// ---------------------------------------------------------------------------------------------

// Per-segment bit counts (vlc_kernel_sm64huff.cu:119-122's outidx): one wave per segment.
__global__ void __launch_bounds__(kThreads) k_segment_bits(const uint32_t* in, uint64_t num_words,
                                                           const uint32_t* cl, uint32_t* counts, uint64_t num_segs)
{
    __shared__ uint32_t lens[256];
    lens[threadIdx.x] = cl[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwave = (uint64_t)gridDim.x * 4;
    for (uint64_t seg = wave0; seg < num_segs; seg += nwave) {
        const uint64_t wbase = seg * kSegWords + (uint64_t)lane * 4u;
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (wbase + k < num_words) {
                const uint32_t w = in[wbase + k];
                t += lens[w >> 24] + lens[(w >> 16) & 255u] + lens[(w >> 8) & 255u] + lens[w & 255u];
            }
        t = wave_sum_u32(t);
        if (lane == 0) counts[seg] = t;
    }
}

// Exclusive scan uint32 -> uint64, single launch, wavefront-hierarchical: 16 elements per lane
// serially, DPP scan across the wave, wave totals through LDS, workgroup totals through the same
// two-level look-back granules as the encoder (replaces scan.cu:67-231 + scanLargeArray_kernel.cu).
// Only tile_bits / group_desc / status / timeout_ticks / num_tiles of `p` are used.
__global__ void __launch_bounds__(kThreads) k_exclusive_scan(const uint32_t* in, uint64_t n, uint64_t* out,
                                                             uint64_t* total, EncodeParams p)
{
    __shared__ uint64_t wtot[4];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_abort;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) s_abort = 0u;
    for (uint64_t tile = blockIdx.x; tile < p.num_tiles; tile += gridDim.x) {
        const uint64_t base = tile * (uint64_t)kTileWords + (uint64_t)tid * 16u;
        uint32_t v[16];
        uint64_t sum = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
        // exact 64-bit wave scan from two 32-bit DPP scans: sum < 2^36, so (sum >> 16) < 2^20 and
        // (sum & 0xffff) < 2^16 stay below 2^32 after adding 64 lanes
        const uint32_t a = wave_incl_scan((uint32_t)(sum >> 16));
        const uint32_t b = wave_incl_scan((uint32_t)(sum & 0xffffu));
        const uint64_t incl = ((uint64_t)a << 16) + b;
        __syncthreads();                                  // previous tile's wtot / s_prefix consumed
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        const uint64_t w0 = wtot[0], w1 = wtot[1], w2 = wtot[2], w3 = wtot[3];
        const uint64_t T = w0 + w1 + w2 + w3;
        const uint64_t wavebase = wave == 0 ? 0ull : wave == 1 ? w0 : wave == 2 ? w0 + w1 : w0 + w1 + w2;
        if (tid == 0) st_granule(&p.tile_bits[tile], kValid | (T & kValMask));
        if (wave == 0) {
            uint64_t P = 0; uint32_t unused = 0;
            const bool ok = lookback<true, false>(p, tile, T, lane, LbRegs(), P, unused);
            if (lane == 0) { s_prefix = P; if (!ok) s_abort = 1u; }
        }
        __syncthreads();
        if (s_abort) return;
        uint64_t run = s_prefix + wavebase + (incl - sum);
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
        if (total && tile == p.num_tiles - 1 && tid == kThreads - 1) *total = run;
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

__global__ void __launch_bounds__(kThreads) k_histogram(const uint8_t* data, uint64_t n_bytes, unsigned long long* hist)
{
    __shared__ uint32_t h[4][256];          // one private histogram per wave: 4x fewer LDS atomic collisions
    for (int i = threadIdx.x; i < 1024; i += kThreads) (&h[0][0])[i] = 0u;
    __syncthreads();
    uint32_t* mine = h[threadIdx.x >> 6];
    const uint64_t nvec = n_bytes / 16;
    const uint4* v = (const uint4*)data;
    uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * kThreads;
    for (; i < nvec; i += stride) {      // host sizes the grid so a 32-bit bin cannot overflow
        const uint4 x = v[i];
        const uint32_t w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&mine[w[k] & 255u], 1u); atomicAdd(&mine[(w[k] >> 8) & 255u], 1u);
            atomicAdd(&mine[(w[k] >> 16) & 255u], 1u); atomicAdd(&mine[w[k] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0)
        for (uint64_t j = nvec * 16 + threadIdx.x; j < n_bytes; j += kThreads) atomicAdd(&mine[data[j]], 1u);
    __syncthreads();
    const uint32_t s = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
    if (s) atomicAdd(&hist[threadIdx.x], (unsigned long long)s);
}

}  // namespace hvlc
