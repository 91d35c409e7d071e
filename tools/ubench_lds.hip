// tools/ubench_lds.hip -- developer-only: what the LDS instructions of k_encode cost per wave-instruction when every CU
// runs them from many waves at once (round 3: the encoder turned out to be bound by LDS work, not by vector instructions).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds.hip -o tools/ubench_lds && tools/ubench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kIters = 2048, kUnroll = 8;

// MODE: 0 ds_or_b32, lane i -> word (i * stride16 / 16) (stride16 = 16: consecutive words; 18: the H2.2 pattern, 35 bits
//          per lane; 10: H1.0, 21 bits; 32: H4.0), three instructions at w-2, w-1, w like put_bits64_end
//       1 the same with ds_or_b64 x2 at 8-byte aligned pairs
//       2 ds_write_b32 x3      3 ds_or_b32 x3 with only the lanes whose value is nonzero active (1/4 of them on the first)
//       4 ds_read_b64 table lookup, symbol per lane from a skewed draw (`skew`: 0 uniform over 256, 1 H2.2-like)
//       5 ds_read2_b32 consecutive (the store wave's read)   6 ds_read_b32 consecutive
//       7 ds_write_b128 zeroing
template <int MODE>
__global__ void __launch_bounds__(768) k_lds(uint32_t* out, int stride16, int skew, uint64_t* clocks)
{
    __shared__ uint32_t buf[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 768) buf[i] = i * 2654435761u;
    __syncthreads();
    uint32_t* const mine = buf + wave * 1280 + 64;
    uint32_t w = (uint32_t)(lane * stride16) / 16u + 2u;
    uint32_t acc = 0;
    // symbol draw for the lookups
    uint32_t sym[kUnroll];
    for (int j = 0; j < kUnroll; ++j) {
        uint32_t r = (lane * 747796405u + j * 2891336453u + wave * 12345u) >> 8;
        uint32_t s;
        if (!skew) s = r & 255u;
        else { const uint32_t z = __builtin_clz((r & 0xffffu) | 1u) - 16u; static const uint8_t tab[17] = {116,129,164,37,99,138,131,25,171,176,111,241,210,112,231,55,101}; s = tab[z > 16u ? 16u : z]; }
        sym[j] = s * 8u;
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            if (MODE == 0) {
                __hip_atomic_fetch_or(mine + w - 2, acc | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(mine + w - 1, acc | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(mine + w,     acc | 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 1) {
                unsigned long long* q = (unsigned long long*)(mine + ((w - 2) & ~1u));
                __hip_atomic_fetch_or(q,     (unsigned long long)(acc | 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(q + 1, (unsigned long long)(acc | 2u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 2) {
                mine[w - 2] = acc | 1u; mine[w - 1] = acc | 2u; mine[w] = acc | 4u;
                asm volatile("" ::: "memory");
            } else if (MODE == 3) {
                if ((lane & 3) == 0) __hip_atomic_fetch_or(mine + w - 2, acc | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(mine + w - 1, acc | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_or(mine + w,     acc | 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 4) {
                const uint2 e = *(const uint2*)((const char*)buf + sym[j]);
                acc += e.x ^ e.y;
            } else if (MODE == 5) {
                acc += mine[lane + 64 * (j & 3)] ^ mine[lane + 64 * (j & 3) + 1];
            } else if (MODE == 6) {
                acc += mine[lane + 64 * (j & 3)];
            } else {
                ((uint4*)mine)[lane + 64 * (j & 1)] = make_uint4(0, 0, 0, acc & 1u);
                asm volatile("" ::: "memory");
            }
        }
        if (MODE <= 3 || MODE == 7) acc += it;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clocks[blockIdx.x * 12 + wave] = t1 - t0;
    if (acc == 0x12345u) out[0] = acc + mine[lane];
}

template <int MODE> static void run(const char* name, int stride16, int skew, int per_step)
{
    uint32_t* d_out; uint64_t* d_clk;
    const int grid = 512;
    CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_clk, grid * 12 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(768), 0, 0, d_out, stride16, skew, d_clk);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds<MODE>, dim3(grid), dim3(768), 0, 0, d_out, stride16, skew, d_clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    static uint64_t clk[512 * 12];
    CK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
    double mean = 0; for (int i = 0; i < grid * 12; ++i) mean += (double)clk[i]; mean /= grid * 12;
    const double insts_per_wave = (double)kIters * kUnroll * per_step;
    // 24 waves per CU share one LDS: LDS clocks per wave-instruction = wave clocks / instructions / 24
    printf("%-58s %8.1f wave-clocks per instruction = %5.2f LDS clocks per instruction at 24 waves per CU  (%.3f ms)\n", name,
           mean / insts_per_wave, mean / insts_per_wave / 24.0, ms);
    CK(hipFree(d_out)); CK(hipFree(d_clk));
}

int main()
{
    run<0>("ds_or_b32 x3, consecutive words (32 bits per lane)", 16, 0, 3);
    run<0>("ds_or_b32 x3, 35 bits per lane (H2.2)", 18, 0, 3);
    run<0>("ds_or_b32 x3, 21 bits per lane (H1.0)", 10, 0, 3);
    run<0>("ds_or_b32 x3, 64 bits per lane (H4.0)", 32, 0, 3);
    run<1>("ds_or_b64 x2, 35 bits per lane", 18, 0, 2);
    run<1>("ds_or_b64 x2, 64 bits per lane", 32, 0, 2);
    run<2>("ds_write_b32 x3, 35 bits per lane", 18, 0, 3);
    run<3>("ds_or_b32 x3, first one on a quarter of the lanes, 35 bits", 18, 0, 3);
    run<4>("ds_read_b64 lookup, uniform symbols", 16, 0, 1);
    run<4>("ds_read_b64 lookup, H2.2-like symbols", 16, 1, 1);
    run<5>("ds_read2_b32 consecutive", 16, 0, 1);
    run<6>("ds_read_b32 consecutive", 16, 0, 1);
    run<7>("ds_write_b128 consecutive", 16, 0, 1);
    return 0;
}
