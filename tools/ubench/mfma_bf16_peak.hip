// bf16 MFMA issue-rate probe: NACC independent accumulator chains per wave, operand data zero / random,
// 1..3 waves per SIMD.  Prints TFLOP/s (dense 2.5 PF peak = 32 cycles per 32x32x16 MFMA per SIMD at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ src, float *__restrict__ out, int iters) {
    const int tid = threadIdx.x;
    u32x4 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = src[(tid + 64 * i) & 1023]; b[i] = src[(tid * 3 + 17 * i + 5) & 1023]; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int n = 0; n < NACC; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[p % 3]),
                                                                 __builtin_bit_cast(bf16x8, b[(p + n) % 3]), acc[n], 0, 0, 0);
        }
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    if (s == 123.456f) out[tid] = s;
}

template <int NACC>
void run(const u32x4 *d, float *o, int wgs_per_cu, const char *tag) {
    const int iters = 2000 / NACC * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = (double)grid * 4 * iters * 6 * NACC * 32768.0;
    printf("%-8s nacc=%d waves/simd=%d: %.3f ms  %.0f TF\n", tag, NACC, wgs_per_cu, ms, flop / ms / 1e9);
}

int main() {
    u32x4 *d; float *o;
    hipMalloc(&d, 1024 * 16); hipMalloc(&o, 4096);
    unsigned *h = (unsigned *)malloc(1024 * 16);
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 4096; ++i) {
            if (pass == 0) h[i] = 0;
            else {   // two random bf16 in [-2,2): random mantissa/sign, exponent 126..127
                unsigned lo = 0x3F00u | (rand() & 0x80FFu) | ((rand() & 1) << 7), hi = 0x3F00u | (rand() & 0x80FFu);
                h[i] = lo | (hi << 16);
            }
        }
        hipMemcpy(d, h, 1024 * 16, hipMemcpyHostToDevice);
        const char *tag = pass ? "random" : "zeros";
        for (int w = 1; w <= 3; ++w) { run<1>(d, o, w, tag); run<2>(d, o, w, tag); run<4>(d, o, w, tag); }
    }
    return 0;
}
