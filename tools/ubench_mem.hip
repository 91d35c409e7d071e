// tools/ubench_mem.hip -- developer-only: what the MI355X memory system gives the encoder's traffic mix (read N bytes,
// write 0.28 N bytes of a contiguous stream) with no compute, over the store shapes and pipeline depths a kernel could
// use.  Answers (round 3): how far is the mix from a plain copy, what do unaligned store instructions and shared seam
// lines cost, what do wider stores, deeper pipelines and longer bursts buy.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mem.hip -o tools/ubench_mem && tools/ubench_mem [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- plain copy / fill / read
template <int POL>   // 0 plain stores, 1 nt stores
__global__ void __launch_bounds__(256) k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nvec; i += stride) {
        const u32x4 v = __builtin_nontemporal_load(in + i);
        if (POL) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
template <int POL>
__global__ void __launch_bounds__(256) k_fill(u32x4* __restrict__ out, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    const u32x4 v = (u32x4){1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nvec; i += stride) {
        if (POL) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

// ---------------------------------------------------------------- the mix
// Every wave reads 4 KiB per step (4 x dwordx4 nt per lane; step t = wave index + k * waves) and writes chunk t of the
// output stream: words [off + t * wpt, off + (t + 1) * wpt).  DEPTH steps of loads are in flight.
//   SHAPE 0: one dword per lane per instruction, from the chunk's first word (the encoder's round-2 shape: an
//            instruction starts at an arbitrary word)
//   SHAPE 1: dword per lane, a ragged head first so that every following instruction starts on a 256-byte boundary
//   SHAPE 2: dwordx4 per lane: dword head to the next 1 KiB boundary... no: to the next 16-byte boundary, 1 KiB
//            instructions from there, dword tail
//   SHAPE 3: dword head to the next 256-byte boundary, dwordx4 body (aligned 1 KiB pieces at 256-byte boundaries), dword tail
//   SHAPE 4: whole 128-byte lines only, dwordx4: the chunk's span is widened to line boundaries DOWNWARDS at both ends
//            (the line a chunk starts in belongs to it, the line it ends in belongs to the next chunk): what a kernel
//            that hands seam lines from part to part through LDS would store
template <int SHAPE, int POL>
__device__ __forceinline__ void store_chunk(uint32_t* __restrict__ out, uint64_t o, uint32_t n, uint32_t v, int lane)
{
    auto st1 = [&](uint32_t* p) { if (POL) __builtin_nontemporal_store(v, p); else *p = v; };
    auto st4 = [&](uint32_t* p) { const u32x4 x = (u32x4){v, v, v, v}; if (POL) __builtin_nontemporal_store(x, (u32x4*)p); else *(u32x4*)p = x; };
    uint32_t* const dst = out + o;
    if (SHAPE == 0) {
        for (uint32_t k = (uint32_t)lane; k < n; k += 64u) st1(dst + k);
    } else if (SHAPE == 1) {
        const uint32_t head = min(n, (uint32_t)((0u - (uint32_t)o) & 63u));
        if ((uint32_t)lane < head) st1(dst + lane);
        for (uint32_t k = head + (uint32_t)lane; k < n; k += 64u) st1(dst + k);
    } else if (SHAPE == 2 || SHAPE == 3) {
        const uint32_t amask = SHAPE == 2 ? 3u : 63u;
        const uint32_t head = min(n, (uint32_t)((0u - (uint32_t)o) & amask));
        if ((uint32_t)lane < head) st1(dst + lane);
        const uint32_t nb = (n - head) / 4u;                       // 16-byte pieces of the body
        for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) st4(dst + head + 4u * k);
        const uint32_t done = head + 4u * nb;
        if (done + (uint32_t)lane < n) st1(dst + done + lane);
    } else if (SHAPE == 4) {
        const uint64_t a = o & ~31ull, b = (o + n) & ~31ull;       // whole 128-byte lines [a, b)
        const uint32_t nb = (uint32_t)(b - a) / 4u;
        for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) st4(out + a + 4u * k);
    } else if (SHAPE == 5 || SHAPE == 6 || SHAPE == 8) {
        // whole 64-byte sectors (5) / 128-byte lines (6) / 256-byte blocks (8), one dword per lane per instruction
        const uint64_t m = SHAPE == 5 ? 15ull : SHAPE == 6 ? 31ull : 63ull;
        const uint64_t a = o & ~m, b = (o + n) & ~m;
        const uint32_t nb = (uint32_t)(b - a);
        for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) st1(out + a + k);
    } else if (SHAPE == 7) {
        // whole 128-byte lines, dword per lane, but every 7th seam (the seam between two tiles) stays ragged
        const uint64_t t = (o - 1u) / n;                            // (off = 1)
        const uint64_t a = (t % 7u) == 0u ? o : (o & ~31ull), b = (t % 7u) == 6u ? o + n : ((o + n) & ~31ull);
        const uint32_t nb = (uint32_t)(b - a);
        for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) st1(out + a + k);
    } else {
        // SHAPE 9: whole 128-byte lines, dwordx2 per lane (512 bytes per instruction)
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const uint64_t a = o & ~31ull, b = (o + n) & ~31ull;
        const uint32_t nb = (uint32_t)(b - a) / 2u;
        for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) {
            const u32x2 x = (u32x2){v, v};
            if (POL) __builtin_nontemporal_store(x, (u32x2*)(out + a + 2u * k)); else *(u32x2*)(out + a + 2u * k) = x;
        }
    }
}

template <int SHAPE, int POL, int DEPTH>
__global__ void __launch_bounds__(256) k_mix(const u32x4* __restrict__ in, uint64_t steps, uint32_t* __restrict__ out,
                                             uint32_t wpt, uint32_t off, uint32_t burst, uint32_t* sink)
{
    const int lane = threadIdx.x & 63;
    const uint64_t nw = (uint64_t)gridDim.x * 4u, w0 = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    u32x4 r[DEPTH][4];
    uint32_t acc = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        const uint64_t t = w0 + (uint64_t)d * nw;
        if (t < steps) {
            const u32x4* q = in + t * 256u + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j);
        }
    }
    uint64_t it = 0;
    for (uint64_t t = w0; t < steps; t += (uint64_t)DEPTH * nw) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t td = t + (uint64_t)d * nw;
            if (td < steps) {
                const uint32_t v = r[d][0].x ^ r[d][1].y ^ r[d][2].z ^ r[d][3].w;     // waits for step td's loads only
                acc += v;
                // stores first (as the encoder does), then the refill of the registers just consumed
                if (burst <= 1u) store_chunk<SHAPE, POL>(out, (uint64_t)off + td * (uint64_t)wpt, wpt, v, lane);
                else if ((it % burst) == burst - 1u)
                    store_chunk<SHAPE, POL>(out, (uint64_t)off + ((it / burst) * nw + w0) * (uint64_t)(wpt * burst), wpt * burst, v, lane);
                const uint64_t tn = td + (uint64_t)DEPTH * nw;
                if (tn < steps) {
                    const u32x4* q = in + tn * 256u + lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j);
                }
                ++it;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

// ---------------------------------------------------------------- host
static float time_best(const std::function<void()>& launch, float* median = nullptr)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    launch(); CK(hipDeviceSynchronize());
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    if (median) *median = ms[ms.size() / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms[0];
}

struct Ctx { const u32x4* in; uint32_t* out; uint32_t* sink; uint64_t bytes; };

template <int SHAPE, int POL, int DEPTH>
static void mix(const Ctx& c, const char* name, int bpc, uint32_t wpt, uint32_t off, uint32_t burst = 1)
{
    const uint64_t steps = c.bytes / 4096;
    float med;
    const float best = time_best([&] {
        hipLaunchKernelGGL((k_mix<SHAPE, POL, DEPTH>), dim3(256 * bpc), dim3(256), 0, 0, c.in, steps, c.out, wpt, off, burst, c.sink);
    }, &med);
    const double wr = (double)wpt * 4.0 / 4096.0;
    printf("mix %-34s %s d%d bpc%-2d wpt%-4u off%-3u burst%-3u: %.3f ms (med %.3f)  input %5.0f GB/s  total %5.0f GB/s\n", name,
           POL ? "nt   " : "plain", DEPTH, bpc, wpt, off, burst, best, med, c.bytes / best / 1e6, c.bytes * (1 + wr) / best / 1e6);
    fflush(stdout);
}


// ---------------------------------------------------------------- division of labour (the encoder's round-3 shape)
// Workgroups of R reader waves + 1 writer wave: a reader only reads (one 4 KiB step at a time, DEPTH steps in flight), the
// writer only writes -- every `rounds` rounds the output of its workgroup's R * rounds steps as one stretch.
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_roles(const u32x4* __restrict__ in, uint64_t steps, uint32_t* __restrict__ out, uint32_t wpt,
                                               int R, uint32_t rounds, uint32_t* sink, int work = 0)
{
    __shared__ uint32_t steps_read;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) steps_read = 0u;
    __syncthreads();
    const uint64_t nr = (uint64_t)gridDim.x * R, first = (uint64_t)blockIdx.x * R;
    if (wave < R) {
        u32x4 r[DEPTH][4];
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint64_t t = first + wave + (uint64_t)d * nr;
            if (t < steps) { const u32x4* q = in + t * 256u + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j); }
        }
        for (uint64_t t = first + wave; t < steps; t += (uint64_t)DEPTH * nr) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint64_t td = t + (uint64_t)d * nr;
                if (td < steps) {
                    acc += r[d][0].x ^ r[d][1].y ^ r[d][2].z ^ r[d][3].w;
                    for (int w = 0; w < work; ++w) acc = acc * 2654435761u + (uint32_t)w;      // (mode 4: a reader that also computes, like an encoder's data wave)
                    if (lane == 0) __hip_atomic_fetch_add(&steps_read, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint64_t tn = td + (uint64_t)DEPTH * nr;
                    if (tn < steps) { const u32x4* q = in + tn * 256u + lane;
#pragma unroll
                        for (int j = 0; j < 4; ++j) r[d][j] = __builtin_nontemporal_load(q + 64 * j); }
                }
            }
        }
        if (acc == 0x12345678u) *sink = acc;
    } else if (wave == R) {
        // round k of the workgroup = steps first + k * nr + {0..R-1}; a burst = `rounds` rounds, stored as one stretch of the
        // output (bursts of one workgroup are NOT adjacent in the stream unless rounds == 1: each burst gets a region of its own)
        uint32_t k = 0;
        for (uint64_t t0 = first; t0 < steps; t0 += nr * rounds, k += rounds) {
            uint32_t nst = 0;
            for (uint32_t j = 0; j < rounds; ++j) { const uint64_t tj = t0 + (uint64_t)j * nr; if (tj < steps) nst += (uint32_t)(steps - tj < (uint64_t)R ? steps - tj : (uint64_t)R); }
            while (__hip_atomic_load(&steps_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)R * k + nst) __builtin_amdgcn_s_sleep(2);
            uint32_t* dst = out + ((uint64_t)(k / rounds) * gridDim.x + blockIdx.x) * (uint64_t)(R * rounds) * wpt;
            const uint32_t n = nst * wpt;
            for (uint32_t q = (uint32_t)lane * 4u; q + 3u < n; q += 256u) __builtin_nontemporal_store((u32x4){k, k, k, k + 1u}, (u32x4*)(dst + q));
        }
    }
}
template <int DEPTH>
static void roles(const Ctx& c, int wgs_per_cu, int R, uint32_t rounds, uint32_t wpt, int work = 0)
{
    const uint64_t steps = c.bytes / 4096;
    float med;
    const float best = time_best([&] {
        hipLaunchKernelGGL((k_roles<DEPTH>), dim3(256 * wgs_per_cu), dim3(64 * (R + 1)), 0, 0, c.in, steps, c.out, wpt, R, rounds, c.sink, work);
    }, &med);
    const double wr = (double)wpt * 4.0 / 4096.0;
    if (work) printf("work %4d: ", work);
    printf("roles %2d readers + 1 writer x %d per CU, d%d, writer bursts of %2u rounds (%5.1f KiB), wpt %3u: %.3f ms (med %.3f)  input %5.0f GB/s  total %5.0f GB/s\n",
           R, wgs_per_cu, DEPTH, rounds, R * rounds * wpt * 4 / 1024.0, wpt, best, med, c.bytes / best / 1e6, c.bytes * (1 + wr) / best / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const uint64_t gib = argc > 1 ? strtoull(argv[1], 0, 10) : 4;
    const uint64_t bytes = gib << 30;
    u32x4* d_in; uint32_t* d_out; uint32_t* d_sink;
    CK(hipMalloc(&d_in, bytes)); CK(hipMemset(d_in, 1, bytes));
    const uint64_t out_bytes = bytes / 2 + (1u << 20);            // room for wpt <= 512 and the copy of half the input
    CK(hipMalloc(&d_out, out_bytes)); CK(hipMemset(d_out, 0, out_bytes)); CK(hipMalloc(&d_sink, 4));
    Ctx c{d_in, d_out, d_sink, bytes};

    if (argc > 2 && atoi(argv[2]) == 3) {
        for (uint32_t wpt : {172u, 288u, 512u}) {
            mix<2, 1, 2>(c, "dwordx4 per step", 4, wpt, 0);
            mix<2, 1, 3>(c, "dwordx4 per step", 4, wpt, 0);
            mix<2, 1, 2>(c, "burst dwordx4", 4, wpt, 0, 32);
            for (int wgs : {1, 2}) for (int R : {7, 10, 15}) for (uint32_t rounds : {1u, 4u}) {
                if (wgs * (R + 1) > 32) continue;
                roles<1>(c, wgs, R, rounds, wpt);
                roles<2>(c, wgs, R, rounds, wpt);
            }
            roles<2>(c, 4, 3, 1, wpt); roles<2>(c, 4, 3, 4, wpt); roles<2>(c, 4, 7, 1, wpt); roles<1>(c, 4, 7, 1, wpt);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 4) {                         // readers that also compute (`work` multiply-adds per step)
        for (uint32_t wpt : {172u, 288u}) for (int work : {0, 150, 300, 600}) {
            roles<2>(c, 1, 15, 4, wpt, work); roles<1>(c, 2, 10, 4, wpt, work); roles<2>(c, 2, 10, 4, wpt, work); roles<1>(c, 2, 11, 4, wpt, work);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 2) {
        for (int rep = 0; rep < 2; ++rep)
        for (int bpc : {4, 6}) {
            mix<0, 1, 2>(c, "ragged: dword from chunk start", bpc, 287, 1);
            mix<1, 1, 2>(c, "ragged: dword, 256B-aligned instr", bpc, 287, 1);
            mix<5, 1, 2>(c, "ragged: whole 64B sectors, dword", bpc, 287, 1);
            mix<6, 1, 2>(c, "ragged: whole 128B lines, dword", bpc, 287, 1);
            mix<8, 1, 2>(c, "ragged: whole 256B blocks, dword", bpc, 287, 1);
            mix<7, 1, 2>(c, "ragged: 128B lines, 1/7 seams ragged", bpc, 287, 1);
            mix<9, 1, 2>(c, "ragged: whole 128B lines, dwordx2", bpc, 287, 1);
            mix<4, 1, 2>(c, "ragged: whole 128B lines, dwordx4", bpc, 287, 1);
            mix<6, 1, 1>(c, "ragged: whole 128B lines, dword", bpc, 287, 1);
            mix<6, 1, 3>(c, "ragged: whole 128B lines, dword", bpc, 287, 1);
            mix<6, 0, 2>(c, "ragged: whole 128B lines, dword", bpc, 287, 1);
            mix<5, 0, 2>(c, "ragged: whole 64B sectors, dword", bpc, 287, 1);
        }
        return 0;
    }
    // ---- 1. copy, fill, read: the known-good references (guide: float4 copy 6.29 TB/s)
    for (int bpc : {4, 8}) {
        const uint64_t nvec = bytes / 2 / 16;
        for (int pol = 0; pol < 2; ++pol) {
            float med;
            float b = time_best([&] { if (pol) hipLaunchKernelGGL(k_copy<1>, dim3(256 * bpc), dim3(256), 0, 0, d_in, (u32x4*)d_out, nvec);
                                      else     hipLaunchKernelGGL(k_copy<0>, dim3(256 * bpc), dim3(256), 0, 0, d_in, (u32x4*)d_out, nvec); }, &med);
            printf("copy %llu GiB -> %llu GiB %s bpc%d: %.3f ms (med %.3f)  read+written %5.0f GB/s\n", (unsigned long long)gib / 2, (unsigned long long)gib / 2,
                   pol ? "nt   " : "plain", bpc, b, med, 2.0 * nvec * 16 / b / 1e6);
            b = time_best([&] { if (pol) hipLaunchKernelGGL(k_fill<1>, dim3(256 * bpc), dim3(256), 0, 0, (u32x4*)d_out, nvec);
                                else     hipLaunchKernelGGL(k_fill<0>, dim3(256 * bpc), dim3(256), 0, 0, (u32x4*)d_out, nvec); }, &med);
            printf("fill %llu GiB %s bpc%d: %.3f ms (med %.3f)  written %5.0f GB/s\n", (unsigned long long)gib / 2, pol ? "nt   " : "plain", bpc, b, med,
                   1.0 * nvec * 16 / b / 1e6);
        }
    }
    // ---- 2. the mix: read only, then the store shapes at the encoder's volume (288 words per 4 KiB = 0.28)
    for (int bpc : {4, 8}) {
        mix<0, 0, 2>(c, "read only", bpc, 0, 0);
        mix<0, 0, 3>(c, "read only", bpc, 0, 0);
    }
    for (int bpc : {4, 8}) {
        mix<0, 0, 2>(c, "dword, chunks 128B-aligned", bpc, 288, 0);
        mix<0, 1, 2>(c, "dword, chunks 128B-aligned", bpc, 288, 0);
        mix<2, 0, 2>(c, "dwordx4, chunks 128B-aligned", bpc, 288, 0);
        mix<2, 1, 2>(c, "dwordx4, chunks 128B-aligned", bpc, 288, 0);
        mix<2, 1, 1>(c, "dwordx4, chunks 128B-aligned", bpc, 288, 0);
        mix<2, 1, 3>(c, "dwordx4, chunks 128B-aligned", bpc, 288, 0);
        mix<2, 0, 3>(c, "dwordx4, chunks 128B-aligned", bpc, 288, 0);
    }
    // ---- 3. ragged chunks (287 words: every alignment occurs, seam lines are shared by two chunks)
    for (int bpc : {4, 8}) {
        mix<0, 0, 2>(c, "ragged: dword from chunk start", bpc, 287, 1);
        mix<0, 1, 2>(c, "ragged: dword from chunk start", bpc, 287, 1);
        mix<1, 0, 2>(c, "ragged: dword, 256B-aligned instr", bpc, 287, 1);
        mix<1, 1, 2>(c, "ragged: dword, 256B-aligned instr", bpc, 287, 1);
        mix<2, 0, 2>(c, "ragged: dwordx4 16B-aligned", bpc, 287, 1);
        mix<2, 1, 2>(c, "ragged: dwordx4 16B-aligned", bpc, 287, 1);
        mix<3, 0, 2>(c, "ragged: dwordx4 256B-aligned", bpc, 287, 1);
        mix<3, 1, 2>(c, "ragged: dwordx4 256B-aligned", bpc, 287, 1);
        mix<4, 0, 2>(c, "ragged: whole lines only x4", bpc, 287, 1);
        mix<4, 1, 2>(c, "ragged: whole lines only x4", bpc, 287, 1);
        mix<4, 1, 3>(c, "ragged: whole lines only x4", bpc, 287, 1);
    }
    // ---- 4. bursts: one wave writes `burst` chunks' worth of contiguous output every `burst` steps
    for (int bpc : {4, 8}) {
        for (uint32_t burst : {2u, 4u, 7u, 16u, 32u}) {
            mix<2, 0, 2>(c, "burst dwordx4", bpc, 288, 0, burst);
            mix<2, 1, 2>(c, "burst dwordx4", bpc, 288, 0, burst);
        }
    }
    // ---- 5. other write shares (H4.0: 512 words, H1.0: 172 words per 4 KiB)
    for (int bpc : {4, 8}) {
        mix<2, 1, 2>(c, "dwordx4 H4.0 volume", bpc, 512, 0);
        mix<4, 1, 2>(c, "whole lines H4.0 volume", bpc, 511, 1);
        mix<2, 1, 2>(c, "dwordx4 H1.0 volume", bpc, 172, 0);
        mix<4, 1, 2>(c, "whole lines H1.0 volume", bpc, 171, 1);
    }
    return 0;
}
