// tools/ubench_datapath.hip -- developer-only experiment (not part of the product): the CEILING of the encoder's
// per-wave data path on MI355X when nothing couples the waves.  Every wave works on its own 4 KiB tiles:
// 4 x dwordx4 nt loads (next tile prefetched), LDS table lookups + 64-bit concatenation (encode16), DPP scan,
// ds_or packing into a wave-private LDS slot, funnel-shifted dword stores of the packed words to a FAKE position
// (tile * fixed stride), slot cleared.  No tickets, no look-back, no barriers, no scout -- the output is not a
// valid stream.  What it measures: how fast this instruction mix can run at all, i.e. how much of the gap between
// k_encode and the HBM roofline is the data path itself and how much is coupling (barriers, look-back, tickets).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I huffman-gpu_amd/csrc tools/ubench_datapath.hip -o build_tmp/ubench_datapath
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "hvlc_device.hip.h"
using namespace hvlc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kSlotWords = 1024 + 64;     // a wave's 4 KiB tile at up to 8 bits per symbol

// experiment variants of the symbol loop
template <int MODE>   // 0: product encode16, 1: no slot hash, 2: eight lookups in flight (inline asm), 3: both
__device__ __forceinline__ void enc16(const uint4& v, const uint2* lut, uint64_t& acc, uint32_t& tot)
{
    if (MODE == 0) { encode16<true>(v, 4, lut, acc, tot); return; }
    const bool nohash = (MODE & 1) != 0;
    const uint32_t w[4] = { nohash ? v.x : hash4(v.x), nohash ? v.y : hash4(v.y), nohash ? v.z : hash4(v.z), nohash ? v.w : hash4(v.w) };
    acc = 0; tot = 0;
    if (MODE & 2) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t ha = w[2 * half], hb = w[2 * half + 1];
            const uint32_t o0 = slot_offset<3>(ha), o1 = slot_offset<2>(ha), o2 = slot_offset<1>(ha), o3 = slot_offset<0>(ha);
            const uint32_t o4 = slot_offset<3>(hb), o5 = slot_offset<2>(hb), o6 = slot_offset<1>(hb), o7 = slot_offset<0>(hb);
            u32x2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                         "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                         : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "v"(o5), "v"(o6), "v"(o7) : "memory");
            const u32x2 e[8] = { r0, r1, r2, r3, r4, r5, r6, r7 };
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc = (acc << e[i].y) | e[i].x; tot += e[i].y; }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t h = w[k];
            { const uint2 e = lut_at(lut, slot_offset<3>(h)); acc = (acc << e.y) | e.x; tot += e.y; }
            { const uint2 e = lut_at(lut, slot_offset<2>(h)); acc = (acc << e.y) | e.x; tot += e.y; }
            { const uint2 e = lut_at(lut, slot_offset<1>(h)); acc = (acc << e.y) | e.x; tot += e.y; }
            { const uint2 e = lut_at(lut, slot_offset<0>(h)); acc = (acc << e.y) | e.x; tot += e.y; }
        }
    }
}

template <int WAVES, bool PACK, int STORE, int MODE = 0>
__global__ void __launch_bounds__(64 * WAVES) k_datapath(const uint4* in, uint64_t num_tiles4k, const uint2* lut_g, uint32_t* out,
                                                         uint32_t out_stride_words, unsigned long long* total)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw_smem[];
    uint2* lut = (uint2*)raw_smem;                                       // LDS address 0, as lut_at() needs
    uint32_t* slots = (uint32_t*)(raw_smem + 2048);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 256) lut[(MODE & 1) ? (uint32_t)tid : lut_slot((uint32_t)tid)] = lut_g[tid];
    uint32_t* slot = slots + wave * kSlotWords;
    for (int i = lane; i < kSlotWords; i += 64) slot[i] = 0;
    __syncthreads();
    const uint64_t nw = (uint64_t)gridDim.x * WAVES, w0 = (uint64_t)blockIdx.x * WAVES + wave;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint4 raw[4];
    uint64_t t = w0;
    if (t < num_tiles4k) {
        const u32x4* src = (const u32x4*)in + t * 256 + lane;
        for (int r = 0; r < 4; ++r) { const u32x4 v = __builtin_nontemporal_load(src + r * 64); raw[r] = make_uint4(v.x, v.y, v.z, v.w); }
    }
    unsigned long long bits_sum = 0;
    for (; t < num_tiles4k; t += nw) {
        uint64_t acc[4]; uint32_t tot[4], excl[4], rowbase[4], wavetot = 0;
        for (int r = 0; r < 4; ++r) enc16<MODE>(raw[r], lut, acc[r], tot[r]);
        for (int r = 0; r < 4; ++r) {
            const uint32_t incl = wave_incl_scan(tot[r]);
            excl[r] = incl - tot[r]; rowbase[r] = wavetot;
            wavetot += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (STORE == 5) {                      // stores first, then the prefetch (loads younger than stores)
            const uint32_t nown = (wavetot >> 5) & ~3u;
            uint32_t* dst = out + t * out_stride_words;
            for (uint32_t k = lane * 4; k < nown; k += 256) *(uint4*)(dst + k) = make_uint4(tot[0], tot[1], tot[2], tot[3]);
        }
        if (t + nw < num_tiles4k) {
            const u32x4* src = (const u32x4*)in + (t + nw) * 256 + lane;
            for (int r = 0; r < 4; ++r) { const u32x4 v = __builtin_nontemporal_load(src + r * 64); raw[r] = make_uint4(v.x, v.y, v.z, v.w); }
        }
        bits_sum += wavetot;
        if (PACK) {
            for (int r = 0; r < 4; ++r) {
                const bool fits = tot[r] <= 64u;
                put_bits64_flat(slot + 1, rowbase[r] + excl[r], fits ? acc[r] : 0ull, fits ? tot[r] : 0u);
            }
        } else if (acc[0] + acc[1] + acc[2] + acc[3] == 0x123456789ull) slot[lane] = tot[0];
        if (STORE == 2) {                      // the same bytes as dwordx4 stores straight from registers (no LDS traffic)
            const uint32_t nown = (wavetot >> 5) & ~3u;
            uint32_t* dst = out + t * out_stride_words;
            for (uint32_t k = lane * 4; k < nown; k += 256) *(uint4*)(dst + k) = make_uint4(tot[0], tot[1], tot[2], tot[3]);
        } else if (STORE == 4) {               // dwordx4 NON-TEMPORAL stores from registers
            const uint32_t nown = (wavetot >> 5) & ~3u;
            uint32_t* dst = out + t * out_stride_words;
            typedef uint32_t u32x4s __attribute__((ext_vector_type(4)));
            for (uint32_t k = lane * 4; k < nown; k += 256) { u32x4s v = { tot[0], tot[1], tot[2], tot[3] }; __builtin_nontemporal_store(v, (u32x4s*)(dst + k)); }
        } else if (STORE == 5) {
        } else if (STORE == 3) {               // dword stores from registers (no LDS traffic)
            const uint32_t nown = wavetot >> 5;
            uint32_t* dst = out + t * out_stride_words;
            for (uint32_t k = lane; k < nown; k += 64) dst[k] = tot[0];
        } else if (STORE) {
            const uint32_t s = (uint32_t)(t * 7u) & 31u, nown = (s + wavetot) >> 5;
            uint32_t* dst = out + t * out_stride_words;
            for (uint32_t k = lane; k < nown; k += 64) dst[k] = __builtin_amdgcn_alignbit(slot[k], slot[k + 1], s);
            const uint32_t nz = (nown + 4) / 4;
            for (uint32_t k = lane; k < nz; k += 64) ((uint4*)slot)[k] = make_uint4(0, 0, 0, 0);
        }
    }
    bits_sum = wave_sum_u64(bits_sum);
    if (lane == 0 && bits_sum) atomicAdd(total, bits_sum);
}

template <int WAVES, bool PACK, int STORE, int MODE = 0>
static void run(const char* name, const uint4* d_in, uint64_t bytes, const uint2* d_lut, uint32_t* d_out, unsigned long long* d_total,
                int blocks_per_cu)
{
    const uint64_t tiles = bytes / 4096;
    const size_t lds = 2048 + (size_t)WAVES * kSlotWords * 4;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(d_total, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_datapath<WAVES, PACK, STORE, MODE>), dim3(grid), dim3(64 * WAVES), lds, 0, d_in, tiles, d_lut, d_out, 640u, d_total);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    unsigned long long tot; CK(hipMemcpy(&tot, d_total, 8, hipMemcpyDeviceToHost));
    printf("datapath %-34s %d waves x %d blocks/CU: %.3f ms  %.0f GB/s input  (%.3f bits/symbol)\n", name, WAVES, blocks_per_cu, best,
           bytes / best / 1e6, (double)tot / bytes);
}

int main(int argc, char** argv)
{
    const uint64_t bytes = 1ull << 30;
    // synthetic skewed bytes + a plausible prefix table: lengths 1..8 by symbol class
    std::vector<uint8_t> h(bytes);
    uint64_t x = 88172645463325252ull;
    for (uint64_t i = 0; i < bytes; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint32_t r = (uint32_t)(x >> 40) & 0xffff;
        h[i] = r < 32768 ? 0 : r < 49152 ? 1 : r < 57344 ? 2 : r < 61440 ? 3 : (uint8_t)(4 + (r & 63));
    }
    std::vector<uint2> lut(256);
    for (int s = 0; s < 256; ++s) { const uint32_t len = s == 0 ? 1 : s == 1 ? 2 : s == 2 ? 3 : s == 3 ? 4 : 10; lut[s] = make_uint2(s < 4 ? (1u << len) - 2u : (0x3c0u | (uint32_t)(s & 63)), len); }
    uint4* d_in; uint2* d_lut; uint32_t* d_out; unsigned long long* d_total;
    CK(hipMalloc(&d_in, bytes)); CK(hipMalloc(&d_lut, 2048)); CK(hipMalloc(&d_out, (bytes / 4096) * 640ull * 4ull)); CK(hipMalloc(&d_total, 8));
    CK(hipMemcpy(d_in, h.data(), bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(d_lut, lut.data(), 2048, hipMemcpyHostToDevice));
    run<4, true, 1>("load+encode+scan+pack+store", d_in, bytes, d_lut, d_out, d_total, 6);
    run<8, true, 1>("load+encode+scan+pack+store", d_in, bytes, d_lut, d_out, d_total, 3);
    run<4, true, 2>("...+pack+dwordx4 reg stores", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, true, 3>("...+pack+dword reg stores", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, true, 0>("load+encode+scan+pack", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 0>("load+encode+scan", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 2>("load+encode+scan+dwordx4 reg stores", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 4>("load+encode+scan+dwordx4 NT stores", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 0, 1>("load+encode+scan  NO HASH", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 0, 2>("load+encode+scan  8 LOOKUPS IN FLIGHT", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, true, 1, 1>("full path  NO HASH", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, true, 1, 2>("full path  8 LOOKUPS IN FLIGHT", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, true, 1, 3>("full path  BOTH", d_in, bytes, d_lut, d_out, d_total, 6);
    run<4, false, 5>("load+encode+scan+stores BEFORE prefetch", d_in, bytes, d_lut, d_out, d_total, 6);
    return 0;
}
