// What the stand-alone quantizer's LAUNCH SHAPE costs before a single distance is computed (round 4, second session).
// The kernel is one workgroup per CU (8 or 16 waves, 160 KiB of LDS); a wave owns units of 32 / 64 rows (256 B each).
// Stages, each timed by the dispatch's own events (hipExtLaunchKernelGGL), best / median of ITERS launches:
//   empty      nothing (the dispatch itself with this grid / LDS size)
//   image      64 KiB codebook image -> LDS by LDS-DMA + barrier                     (every workgroup reads the same lines)
//   rows       + every wave reads its units' rows (and folds them into one value)    = all of z once
//   copy       + and writes them back                                                 = the 520 B / row of the real kernel
//   copy+hist  + 512 global atomics per workgroup at the end
//   spin<T>    copy with T microseconds of dependent arithmetic between a unit's loads and its stores (what overlap there is)
//   flat       the same bytes as a plain grid-stride copy (2048 x 256 threads)       = what the memory system gives this size
// hipcc -O3 --offload-arch=gfx950 vq_floor.hip -o vq_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { M_IMAGE = 1, M_ROWS = 2, M_STORE = 4, M_HIST = 8, M_BARFIRST = 16 };

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

// RU rows per unit (32 / 64); spin: 100 MHz ticks of busy-wait per unit between loads and stores
template <int NW, int RU, int MODE>
__global__ __launch_bounds__(NW * 64) void k_stage(const u32x4 *__restrict__ img, const float *__restrict__ z, float *__restrict__ zq,
                                                  int *__restrict__ hist, long long nunits, int spin, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NL = RU / 4;                      // 16-byte loads per lane and unit
    long long p = (long long)blockIdx.x * NW + wave;
    const long long pstride = (long long)gridDim.x * NW;
    f32x4 F[NL];
    auto load = [&](long long q) {
        const float *b = z + (size_t)q * RU * 64 + lane * 4;
#pragma unroll
        for (int i = 0; i < NL; ++i) F[i] = *reinterpret_cast<const f32x4 *>(b + i * 256);
    };
    if ((MODE & M_ROWS) && !(MODE & M_BARFIRST) && p < nunits) load(p);
    if (MODE & M_IMAGE) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
        const int rot = (int)((blockIdx.x * 97u) % 64u);
        for (int pc = wave; pc < 64; pc += NW) {
            const int sp = (pc + rot) % 64;
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)(dst + sp * 64));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(img + sp * 64 + lane), "s"(lds) : "memory");
        }
        if (MODE & M_BARFIRST) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if ((MODE & M_ROWS) && p < nunits) load(p);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float acc = 0.f;
    while ((MODE & M_ROWS) && p < nunits) {
        if (spin > 0) {
            const unsigned long long t0 = wall();
#pragma unroll
            for (int i = 0; i < NL; ++i) acc += F[i].x;          // (waits for the loads first)
            while (wall() - t0 < (unsigned long long)spin) acc = acc * 1.0000001f + 1e-9f;
        }
        if (MODE & M_STORE) {
            float *o = zq + (size_t)p * RU * 64 + lane * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4 *>(o + i * 256) = F[i];
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc += F[i].x + F[i].w;
        }
        p += pstride;
        if (p < nunits) load(p);
    }
    if (acc == 1234.5678f) sink[tid] = acc;
    if (MODE & M_HIST) {
        __syncthreads();
        for (int k = tid; k < 512; k += NW * 64) atomicAdd(&hist[k], 1 + (k & 1));
    }
    if (MODE & M_IMAGE) { if (smem[tid] == 77 && sink) sink[1] = 1.f; }
}

__global__ __launch_bounds__(256) void k_flat(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long long n16) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n16) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n16) out[i + u * stride] = v[u];
    }
}

static int ITERS = 25;
template <class L> static void timeit(const char *name, long long N, double bytes, L &&launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int i = 0; i < ITERS + 3; ++i) {
        launch(e0, e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("N=%-8lld %-34s best %7.2f us  median %7.2f us", N, name, ts[0], ts[ts.size() / 2]);
    if (bytes > 0) printf("   %.2f TB/s at best", bytes / ts[0] / 1e6);
    printf("\n"); fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

template <int NW, int RU, int MODE> static void stage(const char *name, long long N, const u32x4 *img, const float *z, float *zq, int *hist, float *sink, int spin = 0) {
    auto kfn = k_stage<NW, RU, MODE>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long nunits = N / RU;
    long long grid = (nunits + NW - 1) / NW; if (grid > 256) grid = 256;
    const double bytes = (MODE & M_ROWS) ? N * 256.0 * ((MODE & M_STORE) ? 2 : 1) : 0;
    char nm[96]; snprintf(nm, sizeof nm, "%s NW=%d RU=%d", name, NW, RU);
    timeit(nm, N, bytes, [&](hipEvent_t e0, hipEvent_t e1) {
        hipExtLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NW * 64), 158 * 1024, 0, e0, e1, 0, img, z, zq, hist, nunits, spin, sink);
    });
}

int main(int argc, char **argv) {
    if (argc > 1) ITERS = atoi(argv[1]);
    u32x4 *img; float *sink; int *hist;
    (void)hipMalloc(&img, 65536 + 4096); (void)hipMemset(img, 1, 65536 + 4096);
    (void)hipMalloc(&sink, 4096 * 4); (void)hipMalloc(&hist, 2048); (void)hipMemset(hist, 0, 2048);
    for (long long N : {65536LL, 262144LL, 2097152LL}) {
        float *z, *zq;
        (void)hipMalloc(&z, N * 256); (void)hipMalloc(&zq, N * 256); (void)hipMemset(z, 0, N * 256); (void)hipMemset(zq, 0, N * 256);
        if (N == 65536) {
            stage<8, 64, 0>("empty", N, img, z, zq, hist, sink);
            stage<16, 32, 0>("empty", N, img, z, zq, hist, sink);
            stage<16, 32, M_IMAGE>("image", N, img, z, zq, hist, sink);
            stage<8, 64, M_IMAGE>("image", N, img, z, zq, hist, sink);
            stage<16, 32, M_HIST>("hist only", N, img, z, zq, hist, sink);
        }
        stage<8, 64, M_IMAGE | M_ROWS>("image+rows", N, img, z, zq, hist, sink);
        stage<16, 32, M_IMAGE | M_ROWS>("image+rows", N, img, z, zq, hist, sink);
        stage<16, 32, M_IMAGE | M_ROWS | M_BARFIRST>("image|barrier|rows", N, img, z, zq, hist, sink);
        stage<8, 64, M_IMAGE | M_ROWS | M_STORE>("image+copy", N, img, z, zq, hist, sink);
        stage<16, 32, M_IMAGE | M_ROWS | M_STORE>("image+copy", N, img, z, zq, hist, sink);
        stage<16, 32, M_ROWS | M_STORE>("copy (no image)", N, img, z, zq, hist, sink);
        stage<16, 32, M_IMAGE | M_ROWS | M_STORE | M_HIST>("image+copy+hist", N, img, z, zq, hist, sink);
        // per-unit compute stand-ins: a 32-row unit of the real kernel is ~2.8 us of SIMD issue, four waves share a SIMD -> ~11 us per
        // unit and wave when all four run (16-wave form); 64-row units on two waves per SIMD likewise
        for (int spin : {300, 600, 1100}) {
            char nm[64]; snprintf(nm, sizeof nm, "image+copy spin %.1f us/unit", spin / 100.0);
            stage<16, 32, M_IMAGE | M_ROWS | M_STORE>(nm, N, img, z, zq, hist, sink, spin);
        }
        stage<8, 64, M_IMAGE | M_ROWS | M_STORE>("image+copy spin 11.0 us/unit", N, img, z, zq, hist, sink, 1100);
        {
            const long long n16 = N * 16;
            timeit("flat copy 2048x256", N, N * 512.0, [&](hipEvent_t e0, hipEvent_t e1) {
                hipExtLaunchKernelGGL(k_flat, dim3(2048), dim3(256), 0, 0, e0, e1, 0, reinterpret_cast<const f32x4 *>(z), reinterpret_cast<f32x4 *>(zq), n16);
            });
            timeit("flat copy 8192x256", N, N * 512.0, [&](hipEvent_t e0, hipEvent_t e1) {
                hipExtLaunchKernelGGL(k_flat, dim3(8192), dim3(256), 0, 0, e0, e1, 0, reinterpret_cast<const f32x4 *>(z), reinterpret_cast<f32x4 *>(zq), n16);
            });
        }
        (void)hipFree(z); (void)hipFree(zq);
    }
    return 0;
}
