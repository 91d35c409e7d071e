// tools/hwid_probe.hip -- developer-only: where the hardware puts the waves of k_encode-shaped workgroups (768 threads, 73.5 KiB
// of LDS: two per CU).  Per wave: HW_ID (wave slot, SIMD, CU, SH, SE, workgroup slot) and XCC_ID.
//   hipcc -O2 --offload-arch=gfx950 tools/hwid_probe.hip -o tools/hwid_probe && tools/hwid_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(768) k_probe(uint32_t* out, int spin)
{
    extern __shared__ uint32_t lds[];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 12 + wave) * 2] = hw; out[(blockIdx.x * 12 + wave) * 2 + 1] = xcc; }
    // stay resident long enough for the whole grid to be placed
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
    if (spin < 0) lds[threadIdx.x] = hw;
}

int main()
{
    const int grid = 512;
    uint32_t* d; CK(hipMalloc(&d, grid * 12 * 8));
    static uint32_t h[512 * 12 * 2];
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 75232));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(grid), dim3(768), 75232, 0, d, 2000);   // 20 us
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13], tg_id [19:16]
        int simd_of_wave[12][4]; memset(simd_of_wave, 0, sizeof(simd_of_wave));
        int tg_hist[16] = {0};
        int cu_count[8 * 8 * 2 * 16]; memset(cu_count, 0, sizeof(cu_count));
        int cu_tgmask[8 * 8 * 2 * 16]; memset(cu_tgmask, 0, sizeof(cu_tgmask));
        int same_simd_base = 0;
        for (int b = 0; b < grid; ++b) {
            const uint32_t hw0 = h[(b * 12) * 2], xcc0 = h[(b * 12) * 2 + 1] & 0xf;
            const int cu = (hw0 >> 8) & 15, sh = (hw0 >> 12) & 1, se = (hw0 >> 13) & 7, tg = (hw0 >> 16) & 15;
            const int key = ((xcc0 * 8 + se) * 2 + sh) * 16 + cu;
            cu_count[key]++; cu_tgmask[key] |= 1 << tg; tg_hist[tg]++;
            for (int w = 0; w < 12; ++w) simd_of_wave[w][(h[(b * 12 + w) * 2] >> 4) & 3]++;
            if (((hw0 >> 4) & 3) == 0) same_simd_base++;
        }
        printf("rep %d: workgroups whose wave 0 sits on SIMD 0: %d of %d\n", rep, same_simd_base, grid);
        for (int w = 0; w < 12; ++w) printf("  wave %2d on SIMD 0/1/2/3: %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
        printf("  tg_id histogram:"); for (int i = 0; i < 16; ++i) if (tg_hist[i]) printf(" [%d]=%d", i, tg_hist[i]); printf("\n");
        int cus = 0, two = 0, distinct = 0;
        for (int k = 0; k < 8 * 8 * 2 * 16; ++k) if (cu_count[k]) { cus++; if (cu_count[k] == 2) { two++; if (__builtin_popcount(cu_tgmask[k]) == 2) distinct++; } }
        printf("  CUs used %d, with exactly two workgroups %d, of which with two different tg_id %d\n", cus, two, distinct);
        if (rep == 0) { printf("  first 8 workgroups (block: xcc se sh cu tg | simd of waves 0..11):\n");
            for (int b = 0; b < 8; ++b) { const uint32_t hw0 = h[(b * 12) * 2]; printf("   %3d: %u %u %u %2u %u |", b, h[(b * 12) * 2 + 1] & 0xf, (hw0 >> 13) & 7, (hw0 >> 12) & 1, (hw0 >> 8) & 15, (hw0 >> 16) & 15);
                for (int w = 0; w < 12; ++w) printf(" %u", (h[(b * 12 + w) * 2] >> 4) & 3); printf("\n"); } }
    }
    return 0;
}
