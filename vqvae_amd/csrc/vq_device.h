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

// The same argmin by the WHOLE WAVE for one row (round 6): lane l takes the codes l, l + 64, ... two at a time, each the c-ordered fmaf
// chain on the row's fp32 data (zbase / zstride wave-uniform: the row's elements come through the scalar path), (distance, index) keys
// with NaN below every distance, folded across the wave.  What the kernels call when a row's candidate list overflows: a trained
// checkpoint's dead codes -- hundreds of near-identical codes at the origin -- send rows there, and one lane running K x D serial fmaf
// (vq_slow_argmin) made those launches 20 x slower (profiles/r06_vq_trained.txt).  Every lane of the wave must call it.
template <int D, bool ROWMAJOR>
__device__ __forceinline__ int vq_wave_argmin(const float *__restrict__ z, size_t zbase, size_t zstride, const float *__restrict__ cb,
                                              const float *__restrict__ ee, int K, float zz, int lane) {
    if (zz != zz) return 0;          // every t = zz + ee is NaN -> first index
    unsigned long long key = ~0ull;
    const float *zr = z + zbase;
    for (int kb = lane; kb < K; kb += 128) {
        const int k1 = kb + 64 < K ? kb + 64 : kb;               // (clamped: the same key again)
        const float *e0 = cb + (size_t)kb * D, *e1 = cb + (size_t)k1 * D;
        float m0 = 0.0f, m1 = 0.0f;
#pragma unroll 4
        for (int c4 = 0; c4 < D / 4; ++c4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(e0 + 4 * c4), b = *reinterpret_cast<const f32x4 *>(e1 + 4 * c4);
            float zv[4];
            if (ROWMAJOR) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(zr + 4 * c4);
                zv[0] = q.x; zv[1] = q.y; zv[2] = q.z; zv[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) zv[i] = zr[(size_t)(4 * c4 + i) * zstride];
            }
            m0 = __builtin_fmaf(zv[3], a.w, __builtin_fmaf(zv[2], a.z, __builtin_fmaf(zv[1], a.y, __builtin_fmaf(zv[0], a.x, m0))));
            m1 = __builtin_fmaf(zv[3], b.w, __builtin_fmaf(zv[2], b.z, __builtin_fmaf(zv[1], b.y, __builtin_fmaf(zv[0], b.x, m1))));
        }
        const float d0 = (zz + ee[kb]) - 2.0f * m0, d1 = (zz + ee[k1]) - 2.0f * m1;
        const unsigned u0 = __float_as_uint(d0), u1 = __float_as_uint(d1);
        const unsigned s0 = (d0 != d0) ? 0u : ((u0 >> 31) ? ~u0 : (u0 | 0x80000000u)), s1 = (d1 != d1) ? 0u : ((u1 >> 31) ? ~u1 : (u1 | 0x80000000u));
        const unsigned long long q0 = ((unsigned long long)s0 << 32) | (unsigned)kb, q1 = ((unsigned long long)s1 << 32) | (unsigned)k1;
        key = q0 < key ? q0 : key;
        key = q1 < key ? q1 : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)key, o), hi = __shfl_xor((unsigned)(key >> 32), o);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        key = other < key ? other : key;
    }
    return (int)(unsigned)key;
}

}  // namespace vqvae
