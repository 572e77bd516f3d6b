// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate and the shader clock it runs at.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_peak.hip -o mfma_f32_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, unsigned long long *cyc, unsigned long long *wall, int rnd) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    float x = threadIdx.x * 0.001f + 0.5f, y = 0.25f + blockIdx.x * 1e-6f;
    if (rnd) { unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u; s ^= s >> 13; s *= 0x5bd1e995u; s ^= s >> 15;
               x = __uint_as_float(0x3f000000u | (s & 0x7fffffu)) - 0.75f; s = s * 1664525u + 1013904223u; y = __uint_as_float(0x3f000000u | (s & 0x7fffffu)) - 0.75f; }
    unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { *cyc = c1 - c0; *wall = w1 - w0; }
}

template <int NACC>
void run(int waves_per_simd, int iters, int rnd = 0) {
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
    float *out; unsigned long long *cyc, *wall;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8); hipMalloc(&wall, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, 10, cyc, wall, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, cyc, wall, rnd);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc, hw; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hw, wall, 8, hipMemcpyDeviceToHost);
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("rnd=%d nacc=%d waves/simd=%d iters=%d: %.3f ms  %.1f TFLOP/s  cycles/mfma(per wave)=%.1f  shader clock=%.3f GHz (wall_clock 100MHz)\n",
           rnd, NACC, waves_per_simd, iters, ms, flop / ms / 1e9, (double)hc / ((double)iters * 8 * NACC), (double)hc / ((double)hw * 10.0));
    hipFree(out); hipFree(cyc); hipFree(wall);
}

int main() {
    run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000); run<4>(2, 5000); run<4>(3, 5000); run<8>(2, 2500);
    run<4>(2, 5000, 1); run<8>(2, 2500, 1); run<4>(3, 5000, 1);
    return 0;
}
