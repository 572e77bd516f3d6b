// Microbenchmark of the single-sweep fp16 screen of the VQ kernel (design study, round 2).
// Measures ONLY the inner loop: codebook image + seeds resident in LDS, one v_mfma_f32_32x32x16_f16 chain of four
// per (32-code tile x 32-row tile), then per accumulator element: key = (acc & mask) | r, top-K update with v_med3_f32.
//   NW   waves per workgroup (one workgroup per CU)      TPW  32-row tiles a wave sweeps together (share A / seed reads)
//   TOPK 2 or 3 tracked keys per lane                    CU   code tiles in flight per row tile (accumulator sets)
// Work is normalised to 96 row tiles per CU; the printed time is per 32 row tiles (= 1024 rows per CU = config 3).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW, int TPW, int TOPK, int CU>
__global__ __launch_bounds__(NW * 64) void k(const uint4 *__restrict__ img, const float *__restrict__ seeds,
                                              const uint4 *__restrict__ zsrc, float *__restrict__ out, int ntile, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *Eimg = reinterpret_cast<uint4 *>(smem);                    // [ntile][4][2][32] x 16 B
    float *sd = reinterpret_cast<float *>(Eimg + (size_t)ntile * 256);  // [ntile][2][16]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < ntile * 256; i += NW * 64) Eimg[i] = img[i];
    for (int i = tid; i < ntile * 32; i += NW * 64) sd[i] = seeds[i];
    __syncthreads();
    const uint4 *ap = Eimg + h * 32 + l31;
    const float *sp = sd + h * 16;
    const float inf = __builtin_inff();
    const unsigned mask = 0xfffffc00u;
    float sink = 0.0f;
    for (int it = 0; it < iters; ++it) {
        f16x8 zb[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                zb[t][q] = __builtin_bit_cast(f16x8, zsrc[((size_t)(it * TPW + t) * 4 + q) * 64 * 16 % 4096 + tid % 1024]);
        float m1[TPW], m2[TPW], m3[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) { m1[t] = -inf; m2[t] = -inf; m3[t] = -inf; }
        for (int ct = 0; ct < ntile; ct += CU) {
            f32x16 acc[CU][TPW];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                uint4 a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = ap[((ct + u) * 4 + q) * 64];
                f32x16 seed;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sp + (ct + u) * 32 + 4 * g);
                    seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                }
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[t][0], seed, 0, 0, 0);
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        acc[u][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[t][q], acc[u][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float key = __uint_as_float((__float_as_uint(acc[u][t][r]) & mask) | (unsigned)(r | 16));
                        if (TOPK >= 3) m3[t] = __builtin_amdgcn_fmed3f(m2[t], m3[t], key);
                        m2[t] = __builtin_amdgcn_fmed3f(m1[t], m2[t], key);
                        m1[t] = __builtin_amdgcn_fmed3f(m1[t], key, inf);
                    }
                    // fresh keys (bit 4 set) get their tile id: m += (m & 16) * (2 * tile - 1)
                    const unsigned f = (unsigned)(2 * (ct + u) - 1);
                    unsigned b1 = __float_as_uint(m1[t]), b2 = __float_as_uint(m2[t]);
                    b1 += (b1 & 16u) * f; b2 += (b2 & 16u) * f;
                    m1[t] = __uint_as_float(b1); m2[t] = __uint_as_float(b2);
                }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) sink += m1[t] + m2[t] + m3[t];
    }
    out[(size_t)blockIdx.x * NW * 64 + tid] = sink;
}

template <int NW, int TPW, int TOPK, int CU> void run(const uint4 *img, const float *seeds, const uint4 *z, float *out, int ntile) {
    const int iters = 96 / (NW * TPW) * 20;           // 20 x 96 row tiles per CU
    const size_t lds = (size_t)ntile * 4096 + (size_t)ntile * 128;
    auto kfn = k<NW, TPW, TOPK, CU>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kfn<<<256, NW * 64, lds>>>(img, seeds, z, out, ntile, 2); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); kfn<<<256, NW * 64, lds>>>(img, seeds, z, out, ntile, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    const double us32 = best * 1e3 / (iters * NW * TPW / 32.0);
    const double mfma_floor = ntile * 4 * 32.0 * 32 / 4 / 2.4e3;          // 32 row tiles x ntile x 4 MFMAs x 32 cyc over 4 SIMDs @2.4 GHz
    printf("NW=%2d TPW=%d TOPK=%d CU=%d K=%4d: %7.2f us per 1024 rows/CU   (MFMA floor %.2f us)  %s\n", NW, TPW, TOPK, CU, ntile * 32,
           us32, mfma_floor, e == hipSuccess ? "" : hipGetErrorString(e));
}


// Two VALU-light sweeps: sweep 1 row maximum (v_max3 tree, 8 ops per 16 elements), sweep 2 compare against max - delta
// (16 v_cmp per 16 elements, hits are rare and handled under a wave-uniform branch).
template <int NW, int TPW>
__global__ __launch_bounds__(NW * 64) void k2(const uint4 *__restrict__ img, const float *__restrict__ seeds,
                                              const uint4 *__restrict__ zsrc, float *__restrict__ out, int ntile, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4 *Eimg = reinterpret_cast<uint4 *>(smem);
    float *sd = reinterpret_cast<float *>(Eimg + (size_t)ntile * 256);
    int *cnt = reinterpret_cast<int *>(sd + ntile * 32);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < ntile * 256; i += NW * 64) Eimg[i] = img[i];
    for (int i = tid; i < ntile * 32; i += NW * 64) sd[i] = seeds[i];
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    const uint4 *ap = Eimg + h * 32 + l31;
    const float *sp = sd + h * 16;
    float sink = 0.0f;
    for (int it = 0; it < iters; ++it) {
        f16x8 zb[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                zb[t][q] = __builtin_bit_cast(f16x8, zsrc[((size_t)(it * TPW + t) * 4 + q) * 64 * 16 % 4096 + tid % 1024]);
        float best[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) best[t] = -__builtin_inff();
        auto cell = [&](int ct, f32x16(&acc)[TPW]) {
            uint4 a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = ap[(ct * 4 + q) * 64];
            f32x16 seed;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sp + ct * 32 + 4 * g);
                seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[t][0], seed, 0, 0, 0);
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[t][q], acc[t], 0, 0, 0);
            }
        };
        for (int ct = 0; ct < ntile; ++ct) {
            f32x16 acc[TPW];
            cell(ct, acc);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float m0 = fmaxf(fmaxf(acc[t][0], acc[t][1]), acc[t][2]), m1 = fmaxf(fmaxf(acc[t][3], acc[t][4]), acc[t][5]);
                const float m2 = fmaxf(fmaxf(acc[t][6], acc[t][7]), acc[t][8]), m3 = fmaxf(fmaxf(acc[t][9], acc[t][10]), acc[t][11]);
                const float m4 = fmaxf(fmaxf(acc[t][12], acc[t][13]), acc[t][14]);
                best[t] = fmaxf(best[t], fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), acc[t][15])));
            }
        }
        float thr[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            best[t] = fmaxf(best[t], __shfl_xor(best[t], 32));
            thr[t] = best[t] + 1.0e-3f * __builtin_fabsf(best[t]) + 1.0f;   // no hits: the clean sweep cost
        }
        for (int ct = 0; ct < ntile; ++ct) {
            f32x16 acc[TPW];
            cell(ct, acc);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                bool any = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) any = any || (acc[t][r] >= thr[t]);
                if (__builtin_amdgcn_ballot_w64(any)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (acc[t][r] >= thr[t]) { const int s = atomicAdd(&cnt[0], 1); sink += (float)(s & 1); }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) sink += best[t];
    }
    out[(size_t)blockIdx.x * NW * 64 + tid] = sink;
}

template <int NW, int TPW> void run2(const uint4 *img, const float *seeds, const uint4 *z, float *out, int ntile) {
    const int iters = 96 / (NW * TPW) * 20;
    const size_t lds = (size_t)ntile * 4096 + (size_t)ntile * 128 + 64;
    auto kfn = k2<NW, TPW>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kfn<<<256, NW * 64, lds>>>(img, seeds, z, out, ntile, 2); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); kfn<<<256, NW * 64, lds>>>(img, seeds, z, out, ntile, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("two-sweep NW=%2d TPW=%d K=%4d: %7.2f us per 1024 rows/CU  %s\n", NW, TPW, ntile * 32, best * 1e3 / (iters * NW * TPW / 32.0),
           e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
    const int ntile = 16;
    size_t nimg = (size_t)ntile * 256 * 16, nz = 4096 * 16 + 1024 * 16;
    unsigned short *hi = (unsigned short *)malloc(nimg), *hz = (unsigned short *)malloc(nz);
    for (size_t i = 0; i < nimg / 2; ++i) hi[i] = (unsigned short)(0x3000 + (rand() & 0x8fff));     // random-sign fp16 around 0.1-1
    for (size_t i = 0; i < nz / 2; ++i) hz[i] = (unsigned short)(0x3000 + (rand() & 0x8fff));
    float *hs = (float *)malloc(ntile * 32 * 4); for (int i = 0; i < ntile * 32; ++i) hs[i] = -(rand() / (float)RAND_MAX);
    uint4 *img, *z; float *seeds, *out;
    hipMalloc(&img, nimg); hipMalloc(&z, nz); hipMalloc(&seeds, ntile * 32 * 4); hipMalloc(&out, 256 * 1024 * 4);
    hipMemcpy(img, hi, nimg, hipMemcpyHostToDevice); hipMemcpy(z, hz, nz, hipMemcpyHostToDevice);
    hipMemcpy(seeds, hs, ntile * 32 * 4, hipMemcpyHostToDevice);
    run2<8, 2>(img, seeds, z, out, ntile);
    run2<12, 2>(img, seeds, z, out, ntile);
    run2<16, 2>(img, seeds, z, out, ntile);
    run2<8, 1>(img, seeds, z, out, ntile);
    run<8, 2, 3, 1>(img, seeds, z, out, ntile);
    run<8, 2, 2, 1>(img, seeds, z, out, ntile);
    run<8, 2, 3, 2>(img, seeds, z, out, ntile);
    run<8, 1, 3, 2>(img, seeds, z, out, ntile);
    run<12, 1, 3, 1>(img, seeds, z, out, ntile);
    run<12, 1, 3, 2>(img, seeds, z, out, ntile);
    run<12, 1, 2, 2>(img, seeds, z, out, ntile);
    run<12, 2, 3, 1>(img, seeds, z, out, ntile);
    run<12, 2, 2, 1>(img, seeds, z, out, ntile);
    run<16, 1, 3, 1>(img, seeds, z, out, ntile);
    run<16, 1, 3, 2>(img, seeds, z, out, ntile);
    run<16, 1, 2, 1>(img, seeds, z, out, ntile);
    run<16, 1, 2, 2>(img, seeds, z, out, ntile);
    run<16, 2, 3, 1>(img, seeds, z, out, ntile);
    run<16, 2, 2, 1>(img, seeds, z, out, ntile);
    return 0;
}
