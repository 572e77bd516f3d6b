// Micro-benchmark of the product's vq_tile_step (text extracted from vqvae_amd/csrc/vq_exact.hip at build time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int D, int RT, bool DO_MFMA, bool DO_ARG>
__device__ __forceinline__ void vq_tile_step(const float *__restrict__ ap, size_t jstride,
                                             const float (&zr)[RT][D / 2], f32x16 (&accM)[RT],
                                             const float *__restrict__ ee_t, int code0,
                                             const f32x16 (&accA)[RT], const float (&zz)[RT],
                                             float (&bd)[RT], int (&bk)[RT]) {
    constexpr int NJ = D / 8;
    if (DO_MFMA) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accM[t][r] = 0.0f;
    }
    f32x4 e4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (DO_MFMA) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + (size_t)j * jstride);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    accM[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zr[t][4 * j + i], accM[t], 0, 0, 0);
        }
        if (DO_ARG) {
            // accumulator registers [r0, r1) of the previous tile are folded in during this j
            const int r0 = (16 * j) / NJ, r1 = (16 * (j + 1)) / NJ;
#pragma unroll
            for (int r = r0; r < r1; ++r) {
                if ((r & 3) == 0) e4 = *reinterpret_cast<const f32x4 *>(ee_t + 8 * (r >> 2));
                const int code = code0 + 8 * (r >> 2) + (r & 3);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const float tt = zz[t] + e4[r & 3];
                    const float d = __builtin_fmaf(-2.0f, accA[t][r], tt);
                    const bool lt = d < bd[t];
                    bd[t] = lt ? d : bd[t];
                    bk[t] = lt ? code : bk[t];
                }
            }
        }
    }
}


template <int RT, bool ARG>
__global__ __launch_bounds__(512, 2) void k(const float *img, const float *zsrc, float *out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float Es[];
    constexpr int D = 64, KC = 512;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < KC * D / 4; i += 512) ((f32x4 *)Es)[i] = ((const f32x4 *)img)[i];
    float *ee_s = Es + KC * D;
    for (int i = tid; i < KC; i += 512) ee_s[i] = img[i] * img[i];
    float zr[RT][32], zz[RT], bd[RT]; int bk[RT];
    for (int t = 0; t < RT; ++t) { for (int s = 0; s < 32; ++s) zr[t][s] = zsrc[(blockIdx.x * 512 + tid) * 64 + t * 32 + s]; zz[t] = zr[t][0]; bd[t] = 1e30f; bk[t] = 0; }
    __syncthreads();
    const size_t jstride = (size_t)2 * KC * 4;
    const float *ap0 = Es + ((size_t)h * KC + l31) * 4;
    const float *ee0 = ee_s + 4 * h;
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 accA[RT], accB[RT];
        zz[0] += 1e-9f;
        vq_tile_step<D, RT, true, false>(ap0, jstride, zr, accA, ee0, 0, accB, zz, bd, bk);
        int ct = 1;
        for (; ct + 1 < 16; ct += 2) {
            vq_tile_step<D, RT, true, ARG>(ap0 + ct * 128, jstride, zr, accB, ee0 + (ct - 1) * 32, (ct - 1) * 32, accA, zz, bd, bk);
            vq_tile_step<D, RT, true, ARG>(ap0 + (ct + 1) * 128, jstride, zr, accA, ee0 + ct * 32, ct * 32, accB, zz, bd, bk);
        }
        vq_tile_step<D, RT, true, ARG>(ap0 + ct * 128, jstride, zr, accB, ee0 + (ct - 1) * 32, (ct - 1) * 32, accA, zz, bd, bk);
        vq_tile_step<D, RT, false, true>(ap0, jstride, zr, accA, ee0 + ct * 32, ct * 32, accB, zz, bd, bk);
        if (!ARG) for (int t = 0; t < RT; ++t) bd[t] += accA[t][3];
    }
    float s = 0; for (int t = 0; t < RT; ++t) s += bd[t] + bk[t];
    out[blockIdx.x * 512 + tid] = s;
}
template <int RT, bool ARG> void run(const float *img, const float *z, float *out, int reps) {
    auto kk = k<RT, ARG>;
    hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t lds = 512 * 64 * 4 + 512 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kk<<<256, 512, lds>>>(img, z, out, 2); hipDeviceSynchronize();
    hipEventRecord(e0); kk<<<256, 512, lds>>>(img, z, out, reps); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("RT=%d ARG=%d: %.3f ms %.1f TFLOP/s\n", RT, (int)ARG, ms, 256.0 * 8 * reps * 16 * 32 * RT * 4096.0 / ms / 1e9);
}
int main() {
    float *img, *z, *out; size_t ni = 512 * 64 + 4096, nz = 256 * 512 * 64;
    float *h = (float *)malloc(nz * 4); for (size_t i = 0; i < nz; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    hipMalloc(&img, ni * 4); hipMalloc(&z, nz * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(img, h, ni * 4, hipMemcpyHostToDevice); hipMemcpy(z, h, nz * 4, hipMemcpyHostToDevice);
    run<2, false>(img, z, out, 40); run<2, true>(img, z, out, 40); run<1, false>(img, z, out, 80); run<1, true>(img, z, out, 80);
    return 0;
}
