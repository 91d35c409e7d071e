// tools/ubench_rw.hip -- developer-only: what the memory system gives a kernel that READS N bytes and WRITES 0.28 N
// (the encoder's traffic mix at H~2.2) with no compute at all, for several store shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every wave: 4 KiB in (4 x dwordx4 nt per lane), `wpt` words out per 4 KiB tile
template <int MODE>   // 0 no stores, 1 dword stores, 2 dwordx4 stores, 3 dwordx4 nt stores, 4 dword stores to a CONTIGUOUS stream
__global__ void __launch_bounds__(256) k_rw(const u32x4* in, uint64_t tiles, uint32_t* out, uint32_t wpt, uint32_t stride, unsigned* sink)
{
    const int lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * 4, w0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t acc = 0;
    for (uint64_t t = w0; t < tiles; t += nw) {
        const u32x4* src = in + t * 256 + lane;
        u32x4 a = __builtin_nontemporal_load(src), b = __builtin_nontemporal_load(src + 64), c = __builtin_nontemporal_load(src + 128), d = __builtin_nontemporal_load(src + 192);
        const uint32_t v = a.x ^ b.y ^ c.z ^ d.w;
        acc += v;
        uint32_t* dst = out + (MODE == 4 ? t * (uint64_t)wpt : t * (uint64_t)stride);
        if (MODE == 1 || MODE == 4) { for (uint32_t k = lane; k < wpt; k += 64) dst[k] = v; }
        if (MODE == 2) { for (uint32_t k = lane * 4; k + 3 < wpt; k += 256) *(u32x4*)(dst + k) = (u32x4){v, v, v, v}; }
        if (MODE == 3) { for (uint32_t k = lane * 4; k + 3 < wpt; k += 256) __builtin_nontemporal_store((u32x4){v, v, v, v}, (u32x4*)(dst + k)); }
        if (MODE >= 5) {                       // bursts: write BURST tiles' worth of output at once, every BURST-th tile
            const uint32_t burst = MODE == 5 ? 8u : 32u;
            const uint64_t it = (t - w0) / nw;
            if ((it % burst) == burst - 1) {
                uint32_t* d2 = out + ((it / burst) * nw + w0) * (uint64_t)(wpt * burst);   // a region of its own per wave and burst
                for (uint32_t k = lane * 4; k + 3 < wpt * burst; k += 256) *(u32x4*)(d2 + k) = (u32x4){v, v, v, v};
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int MODE> static void run(const char* name, const u32x4* d_in, uint64_t bytes, uint32_t* d_out, unsigned* d_sink, int bpc)
{
    const uint64_t tiles = bytes / 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rw<MODE>, dim3(256 * bpc), dim3(256), 0, 0, d_in, tiles, d_out, 288u, 640u, d_sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double wr = MODE ? 288.0 * 4 / 4096 : 0.0;
    printf("rw %-40s %d blocks/CU: %.3f ms  input %.0f GB/s  total %.0f GB/s\n", name, bpc, best, bytes / best / 1e6, bytes * (1 + wr) / best / 1e6);
}

int main()
{
    const uint64_t bytes = 4ull << 30;
    u32x4* d_in; uint32_t* d_out; unsigned* d_sink;
    CK(hipMalloc(&d_in, bytes)); CK(hipMemset(d_in, 1, bytes)); CK(hipMalloc(&d_out, (bytes / 4096) * 640ull * 4)); CK(hipMalloc(&d_sink, 4));
    for (int bpc : {4, 8}) {
        run<0>("read only", d_in, bytes, d_out, d_sink, bpc);
        run<1>("+ dword stores (strided tiles)", d_in, bytes, d_out, d_sink, bpc);
        run<4>("+ dword stores (contiguous stream)", d_in, bytes, d_out, d_sink, bpc);
        run<2>("+ dwordx4 stores", d_in, bytes, d_out, d_sink, bpc);
        run<3>("+ dwordx4 nt stores", d_in, bytes, d_out, d_sink, bpc);
        run<5>("+ dwordx4 stores in 9 KiB bursts", d_in, bytes, d_out, d_sink, bpc);
        run<6>("+ dwordx4 stores in 36 KiB bursts", d_in, bytes, d_out, d_sink, bpc);
    }
    return 0;
}
