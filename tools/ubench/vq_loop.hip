// Micro-benchmark isolating the inner loop of the fused VQ kernel (8 waves/CU, LDS-resident codebook image):
//   V0 MFMA chain on register operands (random data)   V1 + ds_read_b128 A operands from LDS
//   V2 + interleaved running-argmin VALU work           V3 V2 but argmin after (not interleaved)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int RT>
__global__ __launch_bounds__(512, 2) void vqloop(const float *img, const float *zsrc, float *out, int reps) {
    extern __shared__ __attribute__((aligned(16))) float Es[];
    constexpr int D = 64, KC = 512;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < KC * D / 4; i += 512) ((f32x4 *)Es)[i] = ((const f32x4 *)img)[i];
    float *ee_s = Es + KC * D;
    for (int i = tid; i < KC; i += 512) ee_s[i] = img[i] * img[i];
    float zr[RT][32], zz[RT], bd[RT]; int bk[RT];
    for (int t = 0; t < RT; ++t) { for (int s = 0; s < 32; ++s) zr[t][s] = zsrc[(blockIdx.x * 512 + tid) * 64 + t * 32 + s]; zz[t] = zr[t][0]; bd[t] = 1e30f; bk[t] = 0; }
    __syncthreads();
    for (int rep = 0; rep < reps; ++rep) {
        f32x16 accA[RT], accB[RT];
        for (int t = 0; t < RT; ++t) for (int r = 0; r < 16; ++r) { accA[t][r] = 0; accB[t][r] = 0; }
        for (int ct = 0; ct < 16; ++ct) {
            const float *ap = Es + ((size_t)h * KC + ct * 32 + l31) * 4;
            f32x4 e4 = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) accA[t][r] = (V == 1 || V == 4) ? accA[t][r] : 0.0f;
            f32x4 a_nx = *(const f32x4 *)(ap);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 a;
                if (V >= 4) { a = a_nx; if (j < 7) a_nx = *(const f32x4 *)(ap + (size_t)(j + 1) * 2 * KC * 4); }
                else if (V >= 1) a = *(const f32x4 *)(ap + (size_t)j * 2 * KC * 4);
                else { a.x = zr[0][j]; a.y = zr[0][j + 8]; a.z = zr[0][j + 16]; a.w = zr[0][j + 24]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < RT; ++t) accA[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zr[t][4 * j + i], accA[t], 0, 0, 0);
                if (V == 2 || V >= 5) {
#pragma unroll
                    for (int r = 2 * j; r < 2 * j + 2; ++r) {
                        if ((r & 3) == 0) e4 = *(const f32x4 *)(ee_s + ct * 32 + 8 * (r >> 2) + 4 * h);
#pragma unroll
                        for (int t = 0; t < RT; ++t) {
                            float tt = zz[t] + e4[r & 3]; float d = __builtin_fmaf(-2.0f, accB[t][r], tt);
                            bool lt = d < bd[t]; bd[t] = lt ? d : bd[t]; bk[t] = lt ? ct * 32 + r : bk[t];
                        }
                    }
                }
                if (V >= 5) {
#pragma unroll
                    for (int q = 0; q < 4 * RT; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, V == 5 ? 3 : 6, 0);
                    }
                }
            }
            if (V == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if ((r & 3) == 0) e4 = *(const f32x4 *)(ee_s + ct * 32 + 8 * (r >> 2) + 4 * h);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        float tt = zz[t] + e4[r & 3]; float d = __builtin_fmaf(-2.0f, accA[t][r], tt);
                        bool lt = d < bd[t]; bd[t] = lt ? d : bd[t]; bk[t] = lt ? ct * 32 + r : bk[t];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) accB[t] = accA[t];
        }
        for (int t = 0; t < RT; ++t) bd[t] += accB[t][0] + accB[t][5];
    }
    float s = 0; for (int t = 0; t < RT; ++t) s += bd[t] + bk[t];
    out[blockIdx.x * 512 + tid] = s;
}

template <int V, int RT> void run(const float *img, const float *z, float *out, int reps) {
    auto k = vqloop<V, RT>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t lds = 512 * 64 * 4 + 512 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512, lds>>>(img, z, out, 2); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, 512, lds>>>(img, z, out, reps); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = 256.0 * 8 * reps * 16 * 32 * RT * 4096.0;
    printf("V%d RT=%d: %.3f ms %.1f TFLOP/s (%s)\n", V, RT, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
}
int main() {
    float *img, *z, *out; size_t ni = 512 * 64 + 4096, nz = 256 * 512 * 64;
    float *h = (float *)malloc(nz * 4); for (size_t i = 0; i < nz; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    hipMalloc(&img, ni * 4); hipMalloc(&z, nz * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(img, h, ni * 4, hipMemcpyHostToDevice); hipMemcpy(z, h, nz * 4, hipMemcpyHostToDevice);
    run<0, 2>(img, z, out, 40); run<1, 2>(img, z, out, 40); run<2, 2>(img, z, out, 40); run<3, 2>(img, z, out, 40); run<4, 2>(img, z, out, 40); run<5, 2>(img, z, out, 40); run<6, 2>(img, z, out, 40);
    run<2, 1>(img, z, out, 80); run<3, 1>(img, z, out, 80); run<5, 1>(img, z, out, 80); run<6, 1>(img, z, out, 80);
    return 0;
}
