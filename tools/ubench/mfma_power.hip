// What the fp16 matrix cores of an MI355X sustain on ZERO vs RANDOM operands, with the clock they run at (gfx950).
// A bare stream of v_mfma_f32_32x32x16_f16 on every SIMD (two waves per SIMD, four independent accumulators per wave) for a
// few seconds per operand set; every wave reads the shader clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) around its loop, so the effective shader clock is measured, not assumed.  tools/r03_power.sh runs this
// next to a rocm-smi sampler (power, sclk) -- the evidence behind DESIGN.md's "power, not issue slots, is the matrix
// ceiling on this data".
//   hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k(const u32x4 *__restrict__ src, float *out, unsigned long long *clk, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    const f16x8 a0 = __builtin_bit_cast(f16x8, src[lane]), a1 = __builtin_bit_cast(f16x8, src[lane + 64]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, src[lane + 128]), b1 = __builtin_bit_cast(f16x8, src[lane + 192]);
    f32x16 y[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        y[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, y[0], 0, 0, 0);
        y[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, y[1], 0, 0, 0);
        y[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, y[2], 0, 0, 0);
        y[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, y[3], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += y[j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int blocks = 512, iters = 200000;                 // 512 x 4 waves = 2 waves per SIMD on 256 CUs
    u32x4 *zsrc, *rsrc; float *out; unsigned long long *clk;
    unsigned *h = (unsigned *)malloc(256 * 16);
    for (int i = 0; i < 256 * 4; ++i) {
        _Float16 a = (_Float16)(rand() / (float)RAND_MAX - 0.5f), b = (_Float16)(rand() / (float)RAND_MAX - 0.5f);
        unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2); h[i] = ua | ((unsigned)ub << 16);
    }
    hipMalloc(&zsrc, 256 * 16); hipMemset(zsrc, 0, 256 * 16);
    hipMalloc(&rsrc, 256 * 16); hipMemcpy(rsrc, h, 256 * 16, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    unsigned long long hc[2 * 512];
    for (int pass = 0; pass < 2; ++pass) {
        const u32x4 *src = pass ? rsrc : zsrc;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const auto t0 = std::chrono::steady_clock::now();
        double ms_last = 0; int launches = 0;
        do {
            hipEventRecord(e0); k<<<blocks, 256>>>(src, out, clk, iters); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); ms_last = ms; ++launches;
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds);
        hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
        double ghz = 0; for (int b = 0; b < blocks; ++b) ghz += (double)hc[2 * b] / ((double)hc[2 * b + 1] * 10e-9) / 1e9;
        ghz /= blocks;
        const double flop = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        const double tf = flop / (ms_last * 1e-3) / 1e12;
        // issue floor: 32 cycles per MFMA and SIMD, two waves per SIMD -> 8 MFMAs per iteration per SIMD
        printf("%-7s operands: %d launches over %.1f s, last launch %.2f ms -> %.0f TFLOP/s = %.1f %% of 2500; shader clock %.3f GHz; "
               "MFMA issue efficiency at that clock %.1f %%\n", pass ? "random" : "zero", launches, seconds, ms_last, tf, tf / 25.0, ghz,
               100.0 * (8.0 * iters * 32 / (ghz * 1e9)) / (ms_last * 1e-3));
        fflush(stdout);
    }
    return 0;
}
