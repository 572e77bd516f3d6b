// How many independent VALU ops hide under one v_mfma_f32_32x32x2_f32 (2 waves/SIMD)?
// ACC=0: accumulators wherever hipcc puts them (VGPRs here); ACC=1: forced into AGPRs via inline asm.
// DEP=1: the VALU ops READ accumulator registers of an older (finished) MFMA chain, like the VQ argmin does.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int ACC, int DEP>
__global__ __launch_bounds__(512, 2) void k(const float *src, float *out, int iters) {
    const int tid = threadIdx.x;
    float zb[32], za[32];
    for (int s = 0; s < 32; ++s) { zb[s] = src[tid * 64 + s]; za[s] = src[tid * 64 + 32 + s]; }
    f32x16 acc[2], old[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) { acc[a][r] = 0; old[a][r] = src[tid + r + a * 16]; }
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = src[tid + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (ACC) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[a]) : "v"(za[s]), "v"(zb[s]));
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(za[s], zb[s], acc[a], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const int i = (s * 2 + a + q) & 7;
                    if (DEP == 1) v[i] = __builtin_fmaf(old[a][(s + q) & 15], 1.0001f, v[i]);
                    else if (DEP == 2) { unsigned u = __float_as_uint(v[i]); u = (u ^ (u >> 3)) + 0x9e37u; v[i] = __uint_as_float(u); }   // integer VALU: xor-shift + add (2 ops)
                    else if (DEP == 3) { bool lt = v[i] < v[(i + 1) & 7]; v[i] = lt ? v[(i + 2) & 7] : v[i]; }                         // v_cmp + v_cndmask
                    else v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                }
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + tid] = s;
}
template <int NV, int ACC, int DEP> void run(const float *src, float *out) {
    int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, ACC, DEP><<<256, 512>>>(src, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<NV, ACC, DEP><<<256, 512>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("valu/mfma=%d acc=%s dep=%d: %.3f ms %.1f TFLOP/s\n", NV, ACC ? "agpr" : "auto", DEP, ms, 256.0 * 8 * iters * 64 * 4096.0 / ms / 1e9);
}
int main() {
    float *src, *out; size_t n = 512 * 64 + 8192;
    float *h = (float *)malloc(n * 4); for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    run<0, 0, 0>(src, out); run<2, 0, 0>(src, out); run<4, 0, 0>(src, out); run<8, 0, 0>(src, out); run<12, 0, 0>(src, out);
    run<2, 0, 1>(src, out); run<4, 0, 1>(src, out);
    run<0, 1, 0>(src, out); run<4, 1, 0>(src, out); run<8, 1, 0>(src, out);
    run<2, 1, 2>(src, out); run<4, 1, 2>(src, out); run<2, 1, 3>(src, out); run<4, 1, 3>(src, out);
    return 0;
}
