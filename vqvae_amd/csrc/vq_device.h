// Device helpers shared by the exact and the filter-and-refine VectorQuantizer kernels.
#pragma once
#include "common.h"

namespace vqvae {

// Raw buffer descriptor over [p, p + 4 GiB): stride 0, no swizzle.  p must be
// wave-uniform; lanes address it with 32-bit byte offsets.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0xFFFFFFFFu, 0x00020000);
}

// ---------------------------------------------------------------------------
// ATen cascade_sum order for one row of D squares held fully by one thread.
template <int D>
__device__ __forceinline__ float aten_sqsum_full(const float (&sq)[D]) {
    static_assert(D % 8 == 0 && D / 32 < 16, "D must be a multiple of 8 below 512");
    constexpr int NV = D / 8, NI = NV / 4;
    float part[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 8; ++t) part[q][t] = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 8; ++t) part[q][t] = part[q][t] + sq[(4 * i + q) * 8 + t];
#pragma unroll
    for (int v = NI * 4; v < NV; ++v)
#pragma unroll
        for (int t = 0; t < 8; ++t) part[0][t] = part[0][t] + sq[v * 8 + t];
    float fin = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float a = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
        fin = fin + a;
    }
    return fin;
}

// ---------------------------------------------------------------------------
// torch.argmin semantics for one row, scalar, used only for rows whose distances
// may be non-finite (zz or some ||e||^2 not < 1e38): NaN is minimal, first wins.
template <int D, bool ROWMAJOR>
__device__ __noinline__ int vq_slow_argmin(const float *__restrict__ z, size_t zbase, size_t zstride,
                                           const float *__restrict__ cb, const float *__restrict__ ee,
                                           int K, float zz) {
    if (zz != zz) return 0;          // every t = zz + ee is NaN -> first index
    int best = 0;
    float bd = 0.0f;
    for (int k = 0; k < K; ++k) {
        float m = 0.0f;
        for (int c = 0; c < D; ++c)
            m = __builtin_fmaf(z[zbase + (size_t)c * zstride], cb[(size_t)k * D + c], m);
        const float t = zz + ee[k];
        const float u = 2.0f * m;
        const float d = t - u;
        const bool dn = d != d, bn = bd != bd;
        const bool better = (k == 0) || (dn ? !bn : (!bn && d < bd));
        if (better) { best = k; bd = d; }
    }
    return best;
}


}  // namespace vqvae
