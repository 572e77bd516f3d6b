// Does operand variety / LDS-fed operands slow v_mfma_f32_32x32x2_f32?  (2 waves/SIMD, 512-thread WGs)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: constant operands. 1: B operand cycles through 32 VGPRs. 2: A and B cycle through 32 VGPRs each.
// 3: A from LDS (ds_read_b128 per 8 MFMAs), B cycles.  4: like 3 but LDS read result unused by MFMA (A cycles regs).
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float *src, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 512) lds[i] = src[i];
    float zb[32], za[32];
    for (int s = 0; s < 32; ++s) { zb[s] = src[tid * 64 + s]; za[s] = src[tid * 64 + 32 + s]; }
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    float sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 a4;
            if (MODE >= 3) a4 = *(const f32x4 *)(lds + ((j * 64 + (tid & 63)) * 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float A = MODE == 0 ? za[0] : (MODE == 1 ? za[0] : (MODE == 3 ? a4[i] : za[4 * j + i]));
                float B = MODE == 0 ? zb[0] : zb[4 * j + i];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc[1], 0, 0, 0);
            }
            if (MODE == 4) sink += a4.x + a4.y + a4.z + a4.w;
        }
    }
    float s = sink;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 512 + tid] = s;
}
template <int MODE> void run(const float *src, float *out) {
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512>>>(src, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<256, 512>>>(src, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms %.1f TFLOP/s\n", MODE, ms, 256.0 * 8 * iters * 64 * 4096.0 / ms / 1e9);
}
int main() {
    float *src, *out; size_t n = 512 * 64 + 8192;
    float *h = (float *)malloc(n * 4); for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    run<0>(src, out); run<1>(src, out); run<2>(src, out); run<3>(src, out); run<4>(src, out);
    return 0;
}
