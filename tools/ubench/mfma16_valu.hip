// What does one wave64 vector op cost beside (or without) v_mfma_f32_32x32x16_f16 on gfx950?
//   MF = 1: every group is one MFMA followed by NV vector ops of KIND; MF = 0: the vector ops alone.
//   KIND 0 v_fma_f32, 1 v_max_f32 (VOP2), 2 v_med3_f32 (three VGPR sources), 3 v_and_or_b32, 4 v_med3 chain as in the
//   VQ key update (m3, m2, m1 dependent on each other through one key).
// Threads per block 256 (one wave per SIMD) or 512 (two); one block per CU.  Output: cycles per group at 2.0 GHz nominal
// and the implied cycles per vector op.   hipcc -O3 --offload-arch=gfx950 mfma16_valu.hip -o mfma16_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NV, int KIND, int MF>
__global__ __launch_bounds__(512, 2) void k(const float *src, float *out, int iters) {
    const int tid = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)src[tid + i]; b[i] = (_Float16)src[tid + 8 + i]; }
    f32x16 acc[2];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = src[tid + r + 16 * q];
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = src[tid + i];
    float pinf = __builtin_inff(); asm volatile("" : "+v"(pinf));
    const unsigned mask = 0xfffffc00u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (MF) acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[s & 1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i = (s + q) & 7;
                if (KIND == 0) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
                else if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
                else if (KIND == 2) v[i] = __builtin_amdgcn_fmed3f(v[i], v[(i + 1) & 7], v[(i + 2) & 7]);
                else if (KIND == 3) v[i] = __uint_as_float((__float_as_uint(v[i]) & mask) | (unsigned)(q | 16));
                else if (KIND == 4) {
                    if (q % 3 == 0) v[2] = __builtin_amdgcn_fmed3f(v[1], v[2], v[3 + (q & 3)]);
                    else if (q % 3 == 1) v[1] = __builtin_amdgcn_fmed3f(v[0], v[1], v[3 + (q & 3)]);
                    else v[0] = __builtin_amdgcn_fmed3f(v[0], v[3 + (q & 3)], pinf);
                }
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 512 + tid] = s;
}
template <int NV, int KIND, int MF> void run(const float *src, float *out, int threads) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, KIND, MF><<<256, threads>>>(src, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<NV, KIND, MF><<<256, threads>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double cyc_group = ms * 1e-3 * 2.0e9 / (iters * 16.0) / waves_per_simd;      // per group per wave slot, SIMD-serial view
    printf("mfma=%d kind=%d nv=%2d waves/simd=%.0f: %.3f ms  %.1f cyc/group(SIMD) ", MF, KIND, NV, waves_per_simd, ms, cyc_group);
    if (NV) printf(" (%.2f cyc/op beyond %s)", (cyc_group - (MF ? 32.0 : 0.0)) / NV, MF ? "32" : "0");
    printf("\n");
}
int main() {
    float *src, *out; size_t n = 8192;
    float *h = (float *)malloc(n * 4); for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    for (int th = 256; th <= 512; th += 256) {
        run<0, 0, 1>(src, out, th);
        run<8, 0, 0>(src, out, th); run<8, 1, 0>(src, out, th); run<8, 2, 0>(src, out, th); run<8, 3, 0>(src, out, th); run<9, 4, 0>(src, out, th);
        run<4, 2, 1>(src, out, th); run<8, 2, 1>(src, out, th); run<12, 2, 1>(src, out, th); run<16, 2, 1>(src, out, th);
        run<8, 0, 1>(src, out, th); run<8, 1, 1>(src, out, th); run<8, 3, 1>(src, out, th); run<9, 4, 1>(src, out, th); run<12, 4, 1>(src, out, th);
    }
    return 0;
}
