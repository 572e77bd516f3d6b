// The fused conv kernels' front-conv stage in isolation (gfx950): per stage one workgroup barrier, 8 ds_read_b128 of pixel
// operands, then 8 groups of { 2 ds_read_b128 of the NEXT group's weights ; 6 v_mfma_f32_32x32x16_f16 } -- 48 MFMAs =
// 1536 matrix-pipe cycles per wave and stage.  Two 4-wave workgroups per CU (two waves per SIMD), so a stage pair is
// MFMA-bound at 3072 cycles per SIMD.  Variants: BAR (barrier yes / no), LDS (0 none, 1 weights only, 2 weights + pixels),
// DMA (each wave also issues 4 global_load_lds pieces per stage and waits vmcnt(0) at the barrier), WPS (waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 mfma_lds_stage.hip -o mfma_lds_stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define HF(v) __builtin_bit_cast(f16x8, v)
template <int BAR, int LDS, int DMA, int IL = 0>
__global__ __launch_bounds__(256, 2) void k(const u32x4 *__restrict__ src, float *out, int stages) {
    __shared__ u32x4 Xs[4 * 520];
    __shared__ u32x4 Ws[2 * 1152];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 520; i += 256) Xs[i] = src[i];
    for (int i = tid; i < 2 * 1152; i += 256) Ws[i] = src[i + 4096];
    __syncthreads();
    f32x16 Y[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) Y[a][b][r] = 0.0f;
    const u32x4 *xp = Xs + wave * 520 + lane;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    u32x4 X[2][2][2];
    for (int t = 0; t < 2; ++t) for (int m = 0; m < 2; ++m) for (int q = 0; q < 2; ++q) X[t][m][q] = src[lane + 64 * (t * 4 + m * 2 + q)];
    u32x4 W0 = src[lane + 1024], W1 = src[lane + 1088];
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
        if (LDS >= 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) { X[t][m][0] = xp[(t * 2) * 130 + m * 32 * 0 + ((s + m) & 1)]; X[t][m][1] = xp[(t * 2 + 1) * 130 + ((s + m) & 1)]; }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __syncthreads();
        if (DMA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)(Ws + ((s + 1) & 1) * 1152 + (wave_u * 4 + j) * 64));
                const u32x4 *g = src + 8192 + (size_t)(s & 63) * 1024 + (wave_u * 4 + j) * 64 + lane;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
            }
        }
        const u32x4 *wt = Ws + (s & 1) * 1152 + lane;
        u32x4 Wc0 = W0, Wc1 = W1;
        if (LDS >= 1) { Wc0 = wt[0]; Wc1 = wt[64]; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int t = g >> 2, nt = g & 3;
            u32x4 Wn0 = Wc0, Wn1 = Wc1;
            if (LDS >= 1 && g + 1 < 8) { Wn0 = wt[(g + 1) * 128]; Wn1 = wt[(g + 1) * 128 + 64]; }
            __builtin_amdgcn_sched_barrier(0);
            if (IL) {                                      // four accumulators in flight: groups g and g ^ 1 interleaved
                const int n2 = nt ^ 1;
                Y[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][0][1]), Y[0][nt], 0, 0, 0);
                Y[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][1][1]), Y[1][nt], 0, 0, 0);
                Y[0][n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc1), HF(X[t][0][0]), Y[0][n2], 0, 0, 0);
                Y[1][n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc1), HF(X[t][1][0]), Y[1][n2], 0, 0, 0);
                Y[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][0][0]), Y[0][nt], 0, 0, 0);
                Y[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][1][0]), Y[1][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                Wc0 = Wn0; Wc1 = Wn1;
                continue;
            }
            Y[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][0][1]), Y[0][nt], 0, 0, 0);
            Y[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][1][1]), Y[1][nt], 0, 0, 0);
            Y[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc1), HF(X[t][0][0]), Y[0][nt], 0, 0, 0);
            Y[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc1), HF(X[t][1][0]), Y[1][nt], 0, 0, 0);
            Y[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][0][0]), Y[0][nt], 0, 0, 0);
            Y[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][1][0]), Y[1][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Wc0 = Wn0; Wc1 = Wn1;
        }
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += Y[a][b][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int BAR, int LDS, int DMA, int IL = 0> void run(const u32x4 *src, float *out, int blocks_per_cu, const char *tag = "") {
    const int stages = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BAR, LDS, DMA, IL><<<blocks, 256>>>(src, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<BAR, LDS, DMA, IL><<<blocks, 256>>>(src, out, stages); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_stage = ms * 1e3 / stages;
    // MFMA-pipe time of a stage: 48 MFMAs x 32 cycles x waves per SIMD
    printf("%s il=%d bar=%d lds=%d dma=%d waves/simd=%d: %.3f ms, %.3f us per stage; matrix-pipe floor at 2.4 GHz %.3f us -> %.1f %% of it (clock unknown)\n",
           tag, IL, BAR, LDS, DMA, blocks_per_cu, ms, us_stage, 48 * 32 * blocks_per_cu / 2400.0, 100.0 * 48 * 32 * blocks_per_cu / 2400.0 / us_stage);
}

// Same stage with ONE pixel tile per wave and eight waves per workgroup (two waves share an image's planes): 24 MFMAs per wave and
// stage, four waves per SIMD at 128 registers.  Is the matrix pipe fuller with four half-size waves than with two?
template <int BAR, int LDS, int DMA>
__global__ __launch_bounds__(512, 2) void k1(const u32x4 *__restrict__ src, float *out, int stages) {
    __shared__ u32x4 Xs[4 * 520];
    __shared__ u32x4 Ws[2 * 1152];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * 520; i += 512) Xs[i] = src[i];
    for (int i = tid; i < 2 * 1152; i += 512) Ws[i] = src[i + 4096];
    __syncthreads();
    f32x16 Y[4];
    for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) Y[b][r] = 0.0f;
    const u32x4 *xp = Xs + (wave >> 1) * 520 + (wave & 1) * 32 + (lane & 31) + (lane >> 5) * 65;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    u32x4 X[2][2];
    for (int t = 0; t < 2; ++t) for (int q = 0; q < 2; ++q) X[t][q] = src[lane + 64 * (t * 2 + q)];
    u32x4 W0 = src[lane + 1024], W1 = src[lane + 1088];
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
        if (LDS >= 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) { X[t][0] = xp[(t * 2) * 130 + (s & 1)]; X[t][1] = xp[(t * 2 + 1) * 130 + (s & 1)]; }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) __syncthreads();
        if (DMA) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)(Ws + ((s + 1) & 1) * 1152 + (wave_u * 2 + j) * 64));
                const u32x4 *g = src + 8192 + (size_t)(s & 63) * 1024 + (wave_u * 2 + j) * 64 + lane;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
            }
        }
        const u32x4 *wt = Ws + (s & 1) * 1152 + lane;
        u32x4 Wc0 = W0, Wc1 = W1;
        if (LDS >= 1) { Wc0 = wt[0]; Wc1 = wt[64]; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int t = g >> 2, nt = g & 3;
            u32x4 Wn0 = Wc0, Wn1 = Wc1;
            if (LDS >= 1 && g + 1 < 8) { Wn0 = wt[(g + 1) * 128]; Wn1 = wt[(g + 1) * 128 + 64]; }
            __builtin_amdgcn_sched_barrier(0);
            Y[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][1]), Y[nt], 0, 0, 0);
            Y[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc1), HF(X[t][0]), Y[nt], 0, 0, 0);
            Y[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(Wc0), HF(X[t][0]), Y[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Wc0 = Wn0; Wc1 = Wn1;
        }
    }
    float sum = 0;
    for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) sum += Y[b][r];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}
template <int BAR, int LDS, int DMA> void run1(const u32x4 *src, float *out, const char *tag = "") {
    const int stages = 4000, blocks = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k1<BAR, LDS, DMA><<<blocks, 512>>>(src, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k1<BAR, LDS, DMA><<<blocks, 512>>>(src, out, stages); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us_stage = ms * 1e3 / stages;
    printf("%s four half-size waves per SIMD: bar=%d lds=%d dma=%d: %.3f ms, %.3f us per stage; matrix-pipe floor at 2.4 GHz %.3f us -> %.1f %% of it\n",
           tag, BAR, LDS, DMA, ms, us_stage, 24 * 32 * 4 / 2400.0, 100.0 * 24 * 32 * 4 / 2400.0 / us_stage);
}
int main() {
    size_t n = 8192 + 64 * 1024 + 4096;
    u32x4 *src; float *out;
    unsigned *h = (unsigned *)malloc(n * 16);
    for (size_t i = 0; i < n * 4; ++i) { _Float16 a = (_Float16)((rand() / (float)RAND_MAX - 0.5f)), b = (_Float16)((rand() / (float)RAND_MAX - 0.5f)); unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2); h[i] = ua | ((unsigned)ub << 16); }
    hipMalloc(&src, n * 16); hipMalloc(&out, 512 * 512 * 4); hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
    u32x4 *zsrc; hipMalloc(&zsrc, n * 16); hipMemset(zsrc, 0, n * 16);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0, 0, 0>(zsrc, out, bpc, "zeros");
        run<0, 0, 0, 1>(src, out, bpc, "interleaved");
        run<1, 2, 1>(zsrc, out, bpc, "zeros");
        run<0, 0, 0>(src, out, bpc);
        run<1, 0, 0>(src, out, bpc);
        run<0, 1, 0>(src, out, bpc);
        run<1, 1, 0>(src, out, bpc);
        run<1, 2, 0>(src, out, bpc);
        run<1, 2, 1>(src, out, bpc);
    }
    run1<0, 0, 0>(src, out);
    run1<1, 1, 0>(src, out);
    run1<1, 2, 0>(src, out);
    run1<1, 2, 1>(src, out);
    run1<1, 2, 1>(zsrc, out, "zeros");
    return 0;
}
