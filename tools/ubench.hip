// tools/ubench.hip -- developer-only micro-benchmarks behind DESIGN.md section 6 (not part of the product,
// not built by __graft_entry__.build()).  hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench
//
//   1. ticket: returning atomic adds on ONE address from every workgroup of a co-resident grid
//      (agent scope = sc1, and workgroup scope = executed in the XCD's L2) -- rate and uniqueness
//   2. ping-pong: one-way latency of an 8-byte granule between two workgroups, same / other XCD,
//      for the store/load flavours a look-back could use
//   3. VALU issue cost of the instructions the symbol loop is made of
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// ---------------------------------------------------------------- 1. tickets
template <int SCOPE>
__global__ void k_ticket(uint32_t* ctr /* 8 counters, 256 B apart */, uint32_t* got, uint32_t* xcc, int iters, int per_xcd,
                         uint64_t* ticks)
{
    if (threadIdx.x != 0) return;
    const uint32_t x = xcc_id();
    xcc[blockIdx.x] = x;
    uint32_t* c = ctr + (per_xcd ? x * 64u : 0u);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t last = 0;
    for (int i = 0; i < iters; ++i) {
        last = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, SCOPE);
        got[(size_t)blockIdx.x * iters + i] = last;
    }
    ticks[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

// ---------------------------------------------------------------- 2. ping-pong
enum { kStoreSc1 = 0, kStorePlain = 1 };
enum { kLoadSc1 = 0, kLoadScalarGlc = 1 };

__device__ __forceinline__ void st_flag(uint64_t* p, uint64_t v, int mode)
{
    if (mode == kStoreSc1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else                   asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint64_t ld_flag(uint64_t* p, int mode)
{
    if (mode == kLoadSc1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t v;
    asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// blocks a and b play; everyone else leaves.  result[0] = ticks (100 MHz) for `iters` round trips, result[1] = failures
__global__ void k_pingpong(uint64_t* flags, int a, int b, int iters, int smode, int lmode, uint64_t* result, uint32_t* xcc)
{
    if (threadIdx.x != 0) return;
    xcc[blockIdx.x] = xcc_id();
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    uint64_t* f0 = flags;          // a -> b
    uint64_t* f1 = flags + 32;     // b -> a  (other 128-B line)
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t fail = 0;
    for (int i = 1; i <= iters && !fail; ++i) {
        if ((int)blockIdx.x == a) {
            st_flag(f0, (uint64_t)i, smode);
            uint32_t n = 0;
            while (ld_flag(f1, lmode) != (uint64_t)i) if (++n > 2000000u) { fail = i; break; }
        } else {
            uint32_t n = 0;
            while (ld_flag(f0, lmode) != (uint64_t)i) if (++n > 2000000u) { fail = i; break; }
            st_flag(f1, (uint64_t)i, smode);
        }
    }
    if ((int)blockIdx.x == a) { result[0] = __builtin_amdgcn_s_memrealtime() - t0; result[1] = fail; }
    else                      { result[2] = fail; }
}

// ---------------------------------------------------------------- 2b. ping-pong under streaming load
// 256-thread workgroups, co-resident grid: waves 0..2 of EVERY workgroup stream a big buffer (4 x 16 B per lane in
// flight) until told to stop; wave 3 of workgroups a and b plays ping-pong with the chosen load flavour.
__global__ void __launch_bounds__(256) k_pingpong_loaded(const uint4* big, uint64_t nvec, uint64_t* flags, int a, int b, int iters,
                                                         int lmode, uint64_t* result, uint32_t* sink, uint32_t* stop)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 3) {
        uint32_t acc = 0;
        uint64_t i = ((uint64_t)blockIdx.x * 3 + wave) * 256 + lane;
        const uint64_t stride = (uint64_t)gridDim.x * 3 * 256;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* src = (const u32x4*)big;
        for (int it = 0; it < 100000; ++it) {
            u32x4 v0 = __builtin_nontemporal_load(src + (i % nvec));
            u32x4 v1 = __builtin_nontemporal_load(src + ((i + 64) % nvec));
            u32x4 v2 = __builtin_nontemporal_load(src + ((i + 128) % nvec));
            u32x4 v3 = __builtin_nontemporal_load(src + ((i + 192) % nvec));
            acc += v0.x + v1.y + v2.z + v3.w;
            i += stride;
            if ((it & 15) == 15 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
        if (acc == 0x12345u) sink[0] = acc;
        return;
    }
    if (lane != 0) return;
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    uint64_t* f0 = flags;
    uint64_t* f1 = flags + 32;
    __builtin_amdgcn_s_sleep(127);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t fail = 0;
    for (int i = 1; i <= iters && !fail; ++i) {
        if ((int)blockIdx.x == a) {
            st_flag(f0, (uint64_t)i, kStoreSc1);
            uint32_t n = 0;
            while (ld_flag(f1, lmode) != (uint64_t)i) if (++n > 2000000u) { fail = i; break; }
        } else {
            uint32_t n = 0;
            while (ld_flag(f0, lmode) != (uint64_t)i) if (++n > 2000000u) { fail = i; break; }
            st_flag(f1, (uint64_t)i, kStoreSc1);
        }
    }
    if ((int)blockIdx.x == a) {
        result[0] = __builtin_amdgcn_s_memrealtime() - t0; result[1] = fail;
        __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else result[2] = fail;
}

// ---------------------------------------------------------------- 3. VALU issue cost
// N independent chains per lane would hide latency; here 8 independent accumulators, 64 ops each per loop trip.
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k_valu(uint32_t* out, int trips, uint64_t* cycles)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint64_t b0 = a0 | 1, b1 = a1 | 1, b2 = a2 | 1, b3 = a3 | 1;
    uint32_t sh = (threadIdx.x & 3) + 1, three = 3;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
        if (OP == 0) {          // v_lshlrev_b32
            REP8(asm volatile("v_lshlrev_b32 %0, %4, %0\n\tv_lshlrev_b32 %1, %4, %1\n\tv_lshlrev_b32 %2, %4, %2\n\tv_lshlrev_b32 %3, %4, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));)
        } else if (OP == 1) {   // v_lshlrev_b64
            REP8(asm volatile("v_lshlrev_b64 %0, %4, %0\n\tv_lshlrev_b64 %1, %4, %1\n\tv_lshlrev_b64 %2, %4, %2\n\tv_lshlrev_b64 %3, %4, %3"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(sh));)
        } else if (OP == 2) {   // SDWA byte extract + shift
            REP8(asm volatile("v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                              "v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                              "v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
                              "v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(three));)
        } else if (OP == 3) {   // v_mad_u64_u32
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                              "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                              : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(sh), "v"(three) : "vcc");)
        } else if (OP == 4) {   // v_bfe_u32
            REP8(asm volatile("v_bfe_u32 %0, %0, 8, 8\n\tv_bfe_u32 %1, %1, 8, 8\n\tv_bfe_u32 %2, %2, 8, 8\n\tv_bfe_u32 %3, %3, 8, 8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (OP == 5) {   // v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %0, %4, %4\n\tv_add3_u32 %1, %1, %4, %4\n\tv_add3_u32 %2, %2, %4, %4\n\tv_add3_u32 %3, %3, %4, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));)
        } else if (OP == 6) {   // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, %4\n\tv_alignbit_b32 %1, %1, %2, %4\n\tv_alignbit_b32 %2, %2, %3, %4\n\tv_alignbit_b32 %3, %3, %0, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));)
        } else if (OP == 7) {   // v_lshl_or_b32
            REP8(asm volatile("v_lshl_or_b32 %0, %0, %4, %1\n\tv_lshl_or_b32 %1, %1, %4, %2\n\tv_lshl_or_b32 %2, %2, %4, %3\n\tv_lshl_or_b32 %3, %3, %4, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));)
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(b0 + b1 + b2 + b3);
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int OP>
static void run_valu(const char* name, uint32_t* d_out, uint64_t* d_cyc, int waves_per_simd)
{
    const int trips = 2000;
    // timing is per wave (wave 0 of block 0) and chip-wide (events); occupancy = waves sharing a SIMD
    const int threads = waves_per_simd >= 4 ? 1024 : 256 * waves_per_simd;
    const int grid = 256 * (waves_per_simd >= 4 ? waves_per_simd / 4 : 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(threads), 0, 0, d_out, trips, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_valu<OP>, dim3(grid), dim3(threads), 0, 0, d_out, trips, d_cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cyc;
    CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
    const double per = (double)cyc / (trips * 32.0);
    const double winstr = (double)grid * (threads / 64) * trips * 32.0;
    printf("valu %-18s waves/SIMD=%d  %.2f s_memtime ticks per wave-instruction (wave view); chip: %.1f G wave-instr/s = %.3f per SIMD per ns\n",
           name, waves_per_simd, per, winstr / ms / 1e6, winstr / ms / 1e6 / 1024.0);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);

    // ---- 1. tickets
    {
        const int grid = 1535, iters = 256;
        uint32_t *d_ctr, *d_got, *d_xcc; uint64_t* d_ticks;
        CK(hipMalloc(&d_ctr, 8 * 256)); CK(hipMalloc(&d_got, (size_t)grid * iters * 4)); CK(hipMalloc(&d_xcc, grid * 4));
        CK(hipMalloc(&d_ticks, grid * 8));
        for (int variant = 0; variant < 3; ++variant) {
            CK(hipMemset(d_ctr, 0, 8 * 256));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (variant == 0)      hipLaunchKernelGGL(k_ticket<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(64), 0, 0, d_ctr, d_got, d_xcc, iters, 0, d_ticks);
            else if (variant == 1) hipLaunchKernelGGL(k_ticket<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(64), 0, 0, d_ctr, d_got, d_xcc, iters, 1, d_ticks);
            else                   hipLaunchKernelGGL(k_ticket<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(64), 0, 0, d_ctr, d_got, d_xcc, iters, 1, d_ticks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint32_t> got((size_t)grid * iters), xcc(grid), ctr(8 * 64);
            std::vector<uint64_t> ticks(grid);
            CK(hipMemcpy(got.data(), d_got, got.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(xcc.data(), d_xcc, grid * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ctr.data(), d_ctr, 8 * 256, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ticks.data(), d_ticks, grid * 8, hipMemcpyDeviceToHost));
            // uniqueness per counter
            int dup = 0, cnt[8] = {0};
            for (int x = 0; x < 8; ++x) {
                std::vector<uint32_t> v;
                for (int b = 0; b < grid; ++b) if (variant == 0 ? x == 0 : (int)xcc[b] == x)
                    for (int i = 0; i < iters; ++i) v.push_back(got[(size_t)b * iters + i]);
                if (variant != 0) cnt[x] = (int)v.size() / iters;
                std::sort(v.begin(), v.end());
                for (size_t i = 0; i < v.size(); ++i) if (v[i] != (uint32_t)i) { ++dup; }
                if (variant == 0) break;
            }
            double avg = 0; for (auto t : ticks) avg += (double)t; avg /= grid;
            const char* nm = variant == 0 ? "agent scope, ONE counter" : variant == 1 ? "agent scope, per-XCD counters" : "workgroup scope, per-XCD counters";
            printf("ticket %-34s: %d x %d adds in %.3f ms = %.1f M/s chip-wide, %.2f us per add per caller, mismatches %d, final ctr[0]=%u\n",
                   nm, grid, iters, ms, grid * (double)iters / ms / 1e3, avg / 100.0 / iters, dup, ctr[0]);
            if (variant == 1) { printf("   workgroups per XCD:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("\n"); }
        }
    }

    // ---- 2. ping-pong
    {
        uint64_t *d_flags, *d_res; uint32_t* d_xcc;
        CK(hipMalloc(&d_flags, 4096)); CK(hipMalloc(&d_res, 64)); CK(hipMalloc(&d_xcc, 64 * 4));
        const int iters = 2000;
        struct { const char* name; int a, b, sm, lm; } cases[] = {
            { "other XCD: sc1 store, sc1 load", 0, 1, kStoreSc1, kLoadSc1 },
            { "same  XCD: sc1 store, sc1 load", 0, 8, kStoreSc1, kLoadSc1 },
            { "same  XCD: plain store, sc1 load", 0, 8, kStorePlain, kLoadSc1 },
            { "other XCD: plain store, sc1 load", 0, 1, kStorePlain, kLoadSc1 },
            { "other XCD: sc1 store, s_load glc", 0, 1, kStoreSc1, kLoadScalarGlc },
            { "same  XCD: sc1 store, s_load glc", 0, 8, kStoreSc1, kLoadScalarGlc },
            { "same  XCD: plain store, s_load glc", 0, 8, kStorePlain, kLoadScalarGlc },
        };
        for (auto& c : cases) {
            CK(hipMemset(d_flags, 0, 4096)); CK(hipMemset(d_res, 0, 64));
            hipLaunchKernelGGL(k_pingpong, dim3(16), dim3(64), 0, 0, d_flags, c.a, c.b, iters, c.sm, c.lm, d_res, d_xcc);
            CK(hipDeviceSynchronize());
            uint64_t res[3]; uint32_t xcc[16];
            CK(hipMemcpy(res, d_res, 24, hipMemcpyDeviceToHost)); CK(hipMemcpy(xcc, d_xcc, 64, hipMemcpyDeviceToHost));
            printf("pingpong %-36s xcc %u/%u: %.3f us one-way (idle chip), fail a=%llu b=%llu\n", c.name, xcc[c.a], xcc[c.b],
                   res[0] / 100.0 / iters / 2.0, (unsigned long long)res[1], (unsigned long long)res[2]);
        }
    }

    // ---- 2b. ping-pong under load
    {
        uint64_t *d_flags, *d_res; uint32_t *d_sink, *d_stop; uint4* d_big;
        const uint64_t nvec = (1ull << 30) / 16;
        CK(hipMalloc(&d_flags, 4096)); CK(hipMalloc(&d_res, 64)); CK(hipMalloc(&d_sink, 64)); CK(hipMalloc(&d_stop, 64));
        CK(hipMalloc(&d_big, nvec * 16)); CK(hipMemset(d_big, 1, nvec * 16));
        const int iters = 500;
        for (int grid : {1024, 1535}) for (int lm = 0; lm < 2; ++lm) for (int pair = 0; pair < 2; ++pair) {
            CK(hipMemset(d_flags, 0, 4096)); CK(hipMemset(d_res, 0, 64)); CK(hipMemset(d_stop, 0, 64));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_pingpong_loaded, dim3(grid), dim3(256), 0, 0, d_big, nvec, d_flags, 0, pair ? 8 : 1, iters, lm, d_res, d_sink, d_stop);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t res[3];
            CK(hipMemcpy(res, d_res, 24, hipMemcpyDeviceToHost));
            printf("pingpong LOADED grid %d (%s XCD) %-12s: %.3f us one-way, fail a=%llu b=%llu (kernel %.2f ms)\n", grid, pair ? "same " : "other",
                   lm ? "s_load glc" : "sc1 load", res[0] / 100.0 / iters / 2.0, (unsigned long long)res[1], (unsigned long long)res[2], ms);
        }
    }

    // ---- 3. VALU
    {
        uint32_t* d_out; uint64_t* d_cyc;
        CK(hipMalloc(&d_out, 512 * 1024 * 4)); CK(hipMalloc(&d_cyc, 8));
        for (int w = 1; w <= 8; w *= 2) {
            run_valu<0>("v_lshlrev_b32", d_out, d_cyc, w);
            run_valu<1>("v_lshlrev_b64", d_out, d_cyc, w);
            run_valu<2>("v_lshlrev_b32_sdwa", d_out, d_cyc, w);
            run_valu<3>("v_mad_u64_u32", d_out, d_cyc, w);
            run_valu<4>("v_bfe_u32", d_out, d_cyc, w);
            run_valu<5>("v_add3_u32", d_out, d_cyc, w);
            run_valu<6>("v_alignbit_b32", d_out, d_cyc, w);
            run_valu<7>("v_lshl_or_b32", d_out, d_cyc, w);
        }
    }
    return 0;
}
