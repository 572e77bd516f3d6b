// Conv2d / ConvTranspose2d / residual-layer forwards for gfx950 as implicit GEMMs
// on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, 157 TF peak).
//
// Replaces the nn.Conv2d / nn.ConvTranspose2d / ResidualLayer calls of
// models/encoder.py:28-43, models/residual.py:18-29,47-51, models/vqvae.py:33 and
// models/decoder.py:27-39.  Activations are row-major (B,H,W,C) between layers
// (one contiguous C-vector per pixel), NCHW only at the image boundaries.
//
// GEMM view of every layer: out[pixel][co] = sum_{tap,ci} in[pixel+tap][ci] * w[tap][ci][co].
//   M side (MFMA A operand)  = 32 output pixels per wave tile; lane l holds pixel (l&31)
//                              and the reduction slots k = l>>5 of each MFMA step.
//   N side (MFMA B operand)  = 32 output channels per tile.
//   reduction                = taps x 32-channel chunks.  fp32 MFMA results only need
//                              tolerance-level parity for convs (oneDNN's order is opaque,
//                              SURVEY.md A.2), so the k-slot assignment is free: within a
//                              chunk lane-half h owns channels [16h, 16h+16), which makes
//                              the A operand four contiguous float4 loads per lane straight
//                              from HBM/L2 -- no LDS staging, no transposition.
//   weights are pre-packed once per layer into the B-operand image
//        [phase][tap*cpt + chunk][n_tile][4][2][32][4]   (j', h, n, i): ci = 32*chunk + 16h + 4j' + i
//   so a workgroup streams them linearly through double-buffered LDS and every lane reads its
//   operands with conflict-free ds_read_b128.
// ConvTranspose2d(k=4,s=2,p=1) runs as 4 sub-pixel phases of 2x2 taps (no zero-stuffing);
// ConvTranspose2d(k=3,s=1,p=1) is a 3x3 conv with mirrored taps.
#include <string.h>

#include <type_traits>

#include "common.h"
#include "vq_unit.h"

namespace vqvae {

// Ordering of a wave's own LDS writes and reads.  LDS operations of one wave are performed in issue order, so only the
// compiler has to be kept from reordering them.  (A workgroup-scope release fence lowers to s_waitcnt vmcnt(0) lgkmcnt(0):
// it would also drain every outstanding global prefetch and every store of the previous output tile -- ~2 us each.)
__device__ __forceinline__ void lds_order_wave() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }


struct ConvGeom {
    int B, Hin, Win, Cin;
    int Hg, Wg;                 // output pixel grid per phase
    int Hout, Wout, Cout;
    int istride, ostride;       // iy = gy*istride + dy ; oy = gy*ostride + opy
    int ntaps, nphase, cpt;     // cpt = ceil(Cin/32) chunks per tap
    int ntile;                  // ceil(Cout/32)
    int flags;
    signed char dy[4][16], dx[4][16];   // [phase][tap]
    signed char kyx[4][16];             // [phase][tap] -> ky*kw + kx in the torch weight
    signed char opy[4], opx[4];
    unsigned long long dymask[4], dxmask[4];   // 4 bits per tap: (dy + 8), (dx + 8) -- scalar decode
    int kk;                     // kh*kw
    int transposed;             // weight is (Cin,Cout,kh,kw)
    int s2d;                    // pack only: space-to-depth chunk order of the 4x4 s2 conv (conv_tile8_bf3_kernel<., true>)
    // epilogue of the data-gradient launches (vqvae_conv_forward_ep_f32; both NULL otherwise), output layout, row-major:
    //   out = (ep_mask > 0) ? conv + ep_add : 0      -- the skip gradient of a residual layer and the ReLU mask of the layer below
    const float *ep_add, *ep_mask;
};

// (a 16-byte group of the output, `off` floats into it)
__device__ __forceinline__ f32x4 ep_apply4(const ConvGeom &g, long long off, f32x4 a) {
    if (g.ep_add) a += *reinterpret_cast<const f32x4 *>(g.ep_add + off);
    if (g.ep_mask) {
        const f32x4 m = *reinterpret_cast<const f32x4 *>(g.ep_mask + off);
        a.x = m.x > 0.0f ? a.x : 0.0f; a.y = m.y > 0.0f ? a.y : 0.0f; a.z = m.z > 0.0f ? a.z : 0.0f; a.w = m.w > 0.0f ? a.w : 0.0f;
    }
    return a;
}
__device__ __forceinline__ float ep_apply1(const ConvGeom &g, long long off, float v) {
    if (g.ep_add) v += g.ep_add[off];
    if (g.ep_mask) v = g.ep_mask[off] > 0.0f ? v : 0.0f;
    return v;
}

constexpr int kFlagReluIn = 1, kFlagReluOut = 2;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;      // >= num_records of every descriptor (and no wrap when the
                                                  // chunk / float4 offsets are added): the load returns 0

// Buffer descriptor over the activation tensor starting at `p` (wave-uniform), `bytes` long: loads
// past the end -- and lanes whose offset is forced to kOobOffset (padding taps) -- read as zero, so
// the im2col border handling costs one select per tap instead of per-load predication.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const float *p, unsigned long long bytes) {
    const unsigned n = bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, n, 0x00020000);
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
// v_max_f32 / v_max3_f32 as ONE instruction each (fmaxf costs two: hipcc puts a canonicalising v_max in front; the
// hardware instruction already returns the other operand for a NaN, which is fmaxf's rule)
__device__ __forceinline__ float vmax(float a, float b) {
    float o;
    asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ void vmax3_abs(float &m, float a, float b) {       // m = max(m, |a|, |b|)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vmax3(float &m, float a, float b) {           // m = max(m, a, b)
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b));
}
// (a0, a1) <- max((a0, a1) * d + (b0, b1), 0), m <- max(m, a0, a1): one packed FMA (d a power of two: the product is exact, so
// the fused form rounds once like the separate add), two v_max, one v_max3 for the pair -- five instructions; twelve as hipcc
// emits the C form (multiply, add, two canonicalising v_max in front of the two maxima, per value)
__device__ __forceinline__ f32x2v scale_bias_relu2(float a0, float a1, float d, float b0, float b1, float &m) {
    // (two scalar FMAs, not a packed one: v_pk_fma_f32 wants aligned register pairs, and with 128 accumulator registers live
    // that constraint cost conv_res_pair8_h2_kernel<2, true> 500 spilled registers)
    const f32x2v r = {vmax(__builtin_fmaf(a0, d, b0), 0.0f), vmax(__builtin_fmaf(a1, d, b1), 0.0f)};
    vmax3(m, r.x, r.y);
    return r;
}
// the same with one scale per value (round 4: the weight rows' own powers of two)
#define SCALE2_BIAS_RELU2(A0, A1, D0, D1, B0, B1, M)                             \
    do {                                                                         \
        const float r0_ = vmax(__builtin_fmaf((A0), (D0), (B0)), 0.0f);          \
        const float r1_ = vmax(__builtin_fmaf((A1), (D1), (B1)), 0.0f);          \
        vmax3((M), r0_, r1_);                                                    \
        (A0) = r0_;                                                              \
        (A1) = r1_;                                                              \
    } while (0)
#define SCALE_BIAS_RELU2(A0, A1, D, B0, B1, M)                                   \
    do {                                                                         \
        const f32x2v r_ = scale_bias_relu2((A0), (A1), (D), (B0), (B1), (M));    \
        (A0) = r_.x;                                                             \
        (A1) = r_.y;                                                             \
    } while (0)
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = vmax(v.x, 0.0f); v.y = vmax(v.y, 0.0f); v.z = vmax(v.z, 0.0f); v.w = vmax(v.w, 0.0f);
    return v;
}

// Epilogue helper: move one 32-pixel x 32-channel accumulator tile (this wave's) through a wave-private
// 32 x 32-float LDS tile so that lane L of pass k holds channels 4 (L % 8) .. +3 of pixel L / 8 + 8 k, and finish
// it there with 16-byte accesses: each instruction then covers eight pixels x 128 contiguous bytes (whole cache
// lines; both the LDS write in accumulator layout and the linear 16-byte read-back are conflict-free).
// fin(p, n, v, k): pixel row p of the tile (0..31), first channel n, four accumulator values, pass k (0..3).
// Dword stores straight from the accumulator layout cost ~6x more per byte.
template <typename Fin>
__device__ __forceinline__ void tile_epilogue(float *tile, const float (&v)[16], int lane, int nbase, Fin fin) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = v[r];
    lds_order_wave();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 q = *reinterpret_cast<const f32x4 *>(tile + k * 256 + lane * 4);
        fin((lane >> 3) + 8 * k, nbase + 4 * (lane & 7), q, k);
    }
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Weight packing (once per layer / weight version): conv_pack_images_kernel, below the split helpers.

// ---------------------------------------------------------------------------
// Generic implicit-GEMM kernel.  Workgroup = 4 waves x (MT x 32 pixels) x (NT x 32 channels).
// Each loop iteration covers KC = 2 reduction chunks (64 channels of one tap): one barrier per
// 2*16*MT*NT MFMAs per wave.  The A operand lives in a 2-deep register ring: the registers of a
// chunk are re-loaded for chunk+2 right after that chunk's MFMAs were issued, so the loads fly
// under the other chunk's MFMAs.  Weights for the next iteration are fetched to registers at the
// top of the iteration and written to the other LDS buffer at its end.
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float *__restrict__ in,
                                                         const float *__restrict__ wimg,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ out, ConvGeom g) {
    constexpr int KC = 2;
    __shared__ __attribute__((aligned(16))) float Bs[2][KC][NT * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int phase = blockIdx.y % g.nphase, nb = blockIdx.y / g.nphase;
    const long long M = (long long)g.B * g.Hg * g.Wg;
    const int nchunk = g.ntaps * g.cpt;
    const bool relu_in = g.flags & kFlagReluIn;
    const unsigned long long dym = g.dymask[phase], dxm = g.dxmask[phase];

    // Per-lane pixel bookkeeping, done once: byte offset of the (tap 0,0) input pixel relative to
    // the first image this workgroup touches, and one validity bit per tap (image-border padding).
    const long long img_px = (long long)g.Hg * g.Wg;
    const long long b_first = ((long long)blockIdx.x * (128 * MT)) / img_px;
    const unsigned long long in_img_bytes = (unsigned long long)g.Hin * g.Win * g.Cin * 4ull;
    const auto in_rs = act_rsrc(in + (size_t)b_first * g.Hin * g.Win * g.Cin,
                                (unsigned long long)(g.B - b_first) * in_img_bytes);
    unsigned pbase[MT], tapmask[MT];
    long long myoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT) + mt * 32 + l31;
        const bool valid = p < M;
        const long long pc = valid ? p : 0;
        const long long b = pc / img_px;
        const int rem = (int)(pc - b * img_px);
        const int gy = rem / g.Wg, gx = rem - gy * g.Wg;
        const int iy0 = gy * g.istride, ix0 = gx * g.istride;
        pbase[mt] = (unsigned)((((b - b_first) * g.Hin + iy0) * g.Win + ix0) * g.Cin * 4 + 64 * h);
        unsigned m = 0;
        for (int t = 0; t < g.ntaps; ++t) {
            const int iy = iy0 + (int)((dym >> (4 * t)) & 15) - 8, ix = ix0 + (int)((dxm >> (4 * t)) & 15) - 8;
            if (valid && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) m |= 1u << t;
        }
        tapmask[mt] = m;
        myoff[mt] = valid ? ((b * g.Hout + gy * g.ostride + g.opy[phase]) * g.Wout +
                             gx * g.ostride + g.opx[phase]) * (long long)g.Cout
                          : -1;
    }
    const float *wbase = wimg + ((size_t)phase * nchunk * g.ntile + (size_t)nb * NT) * 1024;
    const size_t wchunk = (size_t)g.ntile * 1024;

    f32x4 a[KC][MT][4], b_nxt[KC][NT];
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const bool ragged_c = (g.Cin & 31) != 0;
    auto load_a = [&](int c, f32x4(&dst)[MT][4]) {
        const int tap = c / g.cpt, cc = c - tap * g.cpt;
        const int dy = (int)((dym >> (4 * tap)) & 15) - 8, dx = (int)((dxm >> (4 * tap)) & 15) - 8;
        const int tapbytes = (dy * g.Win + dx) * g.Cin * 4;          // scalar
        const unsigned soff = (unsigned)cc * 128u;                    // scalar: 32-channel chunk
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned vo = ((tapmask[mt] >> tap) & 1u) ? pbase[mt] + (unsigned)tapbytes : kOobOffset;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, vo + 16 * j, soff, 0));
        }
    };
    // rare fix-ups, applied once per chunk right before its MFMAs (never between the loads):
    // input ReLU (standalone residual modules) and channel counts that are not a multiple of 32
    const bool needs_fix = relu_in || ragged_c;
    auto fix_a = [&](int c, f32x4(&dst)[MT][4]) {
        const int cc = c % g.cpt;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = dst[mt][j];
                if (cc * 32 + 16 * h + 4 * j >= g.Cin) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                dst[mt][j] = relu_in ? relu4(v) : v;
            }
    };
    auto load_b = [&](int c0) {
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (c0 + k < nchunk) {
                const f32x4 *src = reinterpret_cast<const f32x4 *>(wbase + (size_t)(c0 + k) * wchunk);
#pragma unroll
                for (int q = 0; q < NT; ++q) b_nxt[k][q] = src[tid + 256 * q];
            }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(Bs[buf][k]);
#pragma unroll
            for (int q = 0; q < NT; ++q) dst[tid + 256 * q] = b_nxt[k][q];
        }
    };

#pragma unroll
    for (int k = 0; k < KC; ++k)
        if (k < nchunk) load_a(k, a[k]);
    load_b(0);
    store_b(0);
    __syncthreads();
    const int niter = (nchunk + KC - 1) / KC;
    for (int it = 0; it < niter; ++it) {
        const int c0 = it * KC;
        const bool more = it + 1 < niter;
        if (more) load_b(c0 + KC);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (c0 + k < nchunk) {
                if (needs_fix) fix_a(c0 + k, a[k]);
                const f32x4 *bs = reinterpret_cast<const f32x4 *>(Bs[it & 1][k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 b4[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b4[nt] = bs[((nt * 4 + j) * 2 + h) * 32 + l31];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][mt][j][i], b4[nt][i],
                                                                                   acc[mt][nt], 0, 0, 0);
                }
                if (c0 + k + KC < nchunk) load_a(c0 + k + KC, a[k]);
            }
        }
        if (more) store_b((it + 1) & 1);
        __syncthreads();
    }

    // epilogue: lane holds channel n = tile*32 + l31 of the 16 pixels (r&3)+8(r>>2)+4h
    const bool relu_out = g.flags & kFlagReluOut;
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (nb * NT + nt) * 32 + l31;
        bv[nt] = (bias && n < g.Cout) ? bias[n] : 0.0f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int src = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long long off = __shfl(myoff[mt], src);
            if (off >= 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = (nb * NT + nt) * 32 + l31;
                    if (n < g.Cout) {
                        float v = acc[mt][nt][r] + bv[nt];
                        if (relu_out) v = fmaxf(v, 0.0f);
                        out[off + n] = v;
                    }
                }
            }
        }
}


// ===========================================================================
// Split-bf16 ("bf16x3") implicit GEMM: fp32-grade products on the bf16 matrix cores.
//
// gfx950's exact-fp32 MFMA runs at the vector rate (157 TF) and blocks the VALU while it does;
// its bf16 MFMA is 16x faster per reduction element.  Every fp32 operand is split exactly into
// three bf16 terms, x = x1 + x2 + x3 (8 significand bits each, round-to-nearest, remainders are
// exact in fp32), and a product keeps the six term pairs whose weight is >= 2^-16 relative:
//     x*w ~ x1w1 + (x1w2 + x2w1) + (x1w3 + x2w2 + x3w1)          (dropped pairs are <= 2^-24 |xw|)
// accumulated in fp32 inside v_mfma_f32_32x32x16_bf16.  The per-product error (<= 3*2^-24 relative: three bf16 terms carry
// all 24 significand bits exactly, only the three smallest of the nine term pairs are dropped)
// is the size of fp32's own product rounding, so results stay inside the conv parity tolerance
// (tests/test_conv_gpu.py, tests/test_model_gpu.py: z_e atol 2e-6, no index flips on the goldens)
// while the reduction costs 6 x 2 = 12 matrix-pipe cycles per element pair instead of 32.
// Weights are split once at pack time; activations are split in registers (11 VALU ops per pair,
// which overlap with the matrix pipe -- bf16 MFMA does not occupy the VALU).
// Weight image per (phase, chunk, n_tile): [term 3][step 2][half 2][n 32] x 16 B, element i of a
// 16-B group = channel 32*chunk + 16*half + 8*step + i.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk_bf16_rne(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// split two fp32 values into three packed-bf16 term pairs
__device__ __forceinline__ void split2(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = cvt_pk_bf16_rne(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16_rne(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = cvt_pk_bf16_rne(sa, sb);
}
// split 8 consecutive fp32 channels (two float4) into three bf16x8 terms
__device__ __forceinline__ void split8(const f32x4 &u, const f32x4 &v, u32x4 &t1, u32x4 &t2, u32x4 &t3) {
    unsigned a1, a2, a3, b1, b2, b3, c1, c2, c3, d1, d2, d3;
    split2(u.x, u.y, a1, a2, a3);
    split2(u.z, u.w, b1, b2, b3);
    split2(v.x, v.y, c1, c2, c3);
    split2(v.z, v.w, d1, d2, d3);
    t1 = u32x4{a1, b1, c1, d1};
    t2 = u32x4{a2, b2, c2, d2};
    t3 = u32x4{a3, b3, c3, d3};
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    if (f != f) return 0x7FC0;
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// The six significant term products of one 16-deep MFMA step for two pixel tiles (s*, t*) against one weight
// operand (w1..w3), smallest terms first, the two accumulators interleaved:
//   accA += s (x) w,  accB += t (x) w      with  x (x) w = x3 w1 + x2 w2 + x1 w3 + x2 w1 + x1 w2 + x1 w1
__device__ __forceinline__ void prod6x2(const u32x4 &s1, const u32x4 &s2, const u32x4 &s3, const u32x4 &t1,
                                        const u32x4 &t2, const u32x4 &t3, const u32x4 &w1, const u32x4 &w2,
                                        const u32x4 &w3, f32x16 &accA, f32x16 &accB) {
#define BF(v) __builtin_bit_cast(bf16x8, v)
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s3), BF(w1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t3), BF(w1), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s2), BF(w2), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t2), BF(w2), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s1), BF(w3), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t1), BF(w3), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s2), BF(w1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t2), BF(w1), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s1), BF(w2), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t1), BF(w2), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(s1), BF(w1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(t1), BF(w1), accB, 0, 0, 0);
#undef BF
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-term fp16 products (round 2; the 8x8-map kernels).  fp16 carries 11 significand bits + a signed remainder:
// x = h1 + h2 + r with h1 = fp16(x), h2 = fp16(x - h1) (the difference is exact in fp32): |x - h1| <= 2^-11 |x|, and the
// rounded remainder leaves |r| <= 2^-23 |x|.  The product keeps three of the four term pairs,
//     x*w ~ h1 g1 + h1 g2 + h2 g1        dropped: h2 g2 (<= 2^-22 |xw|) + r w + x s (<= 2^-23 |xw| each),
// i.e. at most 2^-21 |xw| per product -- EIGHT times fp32's own 2^-24 and 2.7x the six-product three-term bf16 scheme
// above (3 * 2^-24), at HALF that scheme's matrix work.  (Round 2 documented 3 * 2^-24 here; that was wrong, VERDICT r2.
// tests/test_conv_gpu.py::test_fp16_two_term_product_bound_on_aligned_operands drives every product of an output to
// that maximum in the same direction and checks 2^-22 <= error <= 2^-21 + accumulation against an fp64 conv.)  The
// parity tiers (z_e atol 2e-6, x_hat 1e-5 + 1e-4 |x_hat|) hold with it: typical operands err by ~2^-24 per product with
// random signs.  What bf16 gave for free and fp16 does not is range: operands are scaled by exact powers of two
// -- weights once per layer at pack time (largest |w| -> [2^14, 2^15)), activations once per IMAGE by the wave that
// owns the image (largest |x| of the image -> [2^14, 2^15)) -- and the accumulator is scaled back in the epilogue.
// Elements more than 2^17 below the image's maximum lose RELATIVE precision (their h2 is a fp16 subnormal, absolute
// error 2^-25 in scaled units = 2^-40 of the maximum), which is invisible next to the fp32 accumulation itself.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_h(float a, float b, unsigned &p1, unsigned &p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};                      // v_cvt_pk_f16_f32, round to nearest even
    const f16x2 r = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, r);
}
// The same on a * sc, b * sc (sc a power of two: the products are exact) in FIVE instructions instead of the ten hipcc
// emits for the C form: the mixed-precision FMAs convert (v_fma_mixlo / mixhi_f16: fp16(a * sc) into one half of the
// register) and subtract (v_fma_mix_f32 with the fp16 half as its addend: a * sc - h, exact) in one step each.  Same bits.
__device__ __forceinline__ void split2_hs(float a, float b, float sc, unsigned &p1, unsigned &p2) {
    unsigned h;
    float ra, rb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "v"(sc));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(sc), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(sc), "v"(h));
    const f16x2 r = {(_Float16)ra, (_Float16)rb};
    p1 = h;
    p2 = __builtin_bit_cast(unsigned, r);
}
// split 8 consecutive fp32 channels (two float4), multiplied by the image's scale, into two fp16x8 terms
__device__ __forceinline__ void split8_h(const f32x4 &u, const f32x4 &v, float sc, u32x4 &t1, u32x4 &t2) {
    unsigned a1, a2, b1, b2, c1, c2, d1, d2;
    split2_hs(u.x, u.y, sc, a1, a2);
    split2_hs(u.z, u.w, sc, b1, b2);
    split2_hs(v.x, v.y, sc, c1, c2);
    split2_hs(v.z, v.w, sc, d1, d2);
    t1 = u32x4{a1, b1, c1, d1};
    t2 = u32x4{a2, b2, c2, d2};
}
// the three significant term products of one 16-deep step for two pixel tiles, smallest terms first
__device__ __forceinline__ void prod3x2(const u32x4 &s1, const u32x4 &s2, const u32x4 &t1, const u32x4 &t2,
                                        const u32x4 &w1, const u32x4 &w2, f32x16 &accA, f32x16 &accB) {
#define HF(v) __builtin_bit_cast(f16x8, v)
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(s2), HF(w1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(t2), HF(w1), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(s1), HF(w2), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(t1), HF(w2), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(s1), HF(w1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(t1), HF(w1), accB, 0, 0, 0);
#undef HF
}
// largest value of the wave -> the power of two that puts it into [2^14, 2^15) (0 for an all-zero or non-finite image)
__device__ __forceinline__ int wave_scale_exp(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    int e = 15;
    if (m > 0.0f && m < 3.0e38f) (void)__builtin_frexpf(m, &e);
    e = 15 - e;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return __builtin_amdgcn_readfirstlane(e);
}

// Producer side of the per-image activation scale: the wave's largest |output| of image `img` goes to out_amax[img]
// (non-negative floats order like signed ints; the array starts at -1 = "not provided").  A consumer that finds a value
// there skips its own pass over the image.
__device__ __forceinline__ void publish_amax(int *out_amax, long long img, float om, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) om = fmaxf(om, __shfl_xor(om, o));
    if (lane == 0) atomicMax(out_amax + img, __float_as_int(om));
}

// The same where exactly ONE wave ever produces image `img` (the one-wave-per-image kernels): a plain store, and the array
// needs no -1 fill in front of the launch.
__device__ __forceinline__ void publish_amax_exclusive(int *out_amax, long long img, float om, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) om = fmaxf(om, __shfl_xor(om, o));
    if (lane == 0) out_amax[img] = __float_as_int(om);
}

// Weight scales of a layer (round 4: one power of two per OUTPUT CHANNEL -- a trained checkpoint's channel norms differ by
// orders of magnitude, and a row 2^17 below the tensor's maximum would lose the bits fp32 keeps).  Header in
// front of the layer's two-term fp16 image, h2_header_bytes(ntile) long:
//   int   [0 .. 63]              misc ([1]: the first layer's L1 bound, conv_in_hdr_kernel)
//   float [64 + c]               dw[c] = 2^-kw[c]: what an accumulator of output channel c is multiplied with (1.0 for the
//                                padding channels of the last 32-channel tile)
//   int   [64 + 32 ntile + c]    kw[c]: row c of the weights is packed as fp16 terms of w * 2^kw[c], largest |w| of the row
//                                -> [2^14, 2^15)
// One block per output channel (the row's Cin * kh * kw elements; `transposed`: the tensor is (Cin, Cout, kh, kw)).
__device__ __forceinline__ const float *h2_dw(const int *hdr) { return reinterpret_cast<const float *>(hdr) + 64; }
// TRANSPOSED accumulator tiles (lane = pixel, register 4 g + q = channel c0 + 8 g + 4 h + q): the four channel scales
// 2^-kw[.] of registers 4 g .. 4 g + 3, times d (the activation side's 2^-kx)
// (tab: the dw table, in the fused kernels a copy in LDS -- one address register (h) and an immediate offset per read, no
// pointer pair kept live next to 128 accumulator registers)
__device__ __forceinline__ f32x4 h2_dw4(const float *tab, int c0, int g, int h, float d) {
    __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise hoists every group's read to the top of the epilogue: 16+ more live registers
    const f32x4 t = *reinterpret_cast<const f32x4 *>(tab + c0 + 8 * g + 4 * h);
    return f32x4{t.x * d, t.y * d, t.z * d, t.w * d};
}
__device__ __forceinline__ int h2_scale_exp(float mm) {
    int e = 15;
    if (mm > 0.0f && mm < 3.0e38f) (void)__builtin_frexpf(mm, &e);
    e = 15 - e;
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__global__ __launch_bounds__(64) void conv_wscale_kernel(const float *__restrict__ w, int Cin, int Cout, int kk, int transposed,
                                                         int ntile, int *__restrict__ hdr) {
    const int co = blockIdx.x, lane = threadIdx.x;
    float m = 0.0f;
    if (co < Cout) {
        if (!transposed) {
            const float *row = w + (size_t)co * Cin * kk;
            for (int i = lane; i < Cin * kk; i += 64) m = fmaxf(m, __builtin_fabsf(row[i]));
        } else {
            for (int i = lane; i < Cin * kk; i += 64) {
                const int ci = i / kk, k = i - ci * kk;
                m = fmaxf(m, __builtin_fabsf(w[((size_t)ci * Cout + co) * kk + k]));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        const int e = co < Cout ? h2_scale_exp(m) : 0;
        reinterpret_cast<float *>(hdr)[64 + co] = __builtin_ldexpf(1.0f, -e);
        hdr[64 + 32 * ntile + co] = e;
    }
}

// Per-image maximum of |x| for a tensor whose producer did not publish one (z_q in front of the decoder on large maps,
// the generic first-layer kernel): grid (parts, B), one more read of the tensor.
__global__ __launch_bounds__(256) void act_absmax_kernel(const float *__restrict__ x, long long elems_per_image,
                                                         int *__restrict__ amax) {
    __shared__ float red[4];
    const float *p = x + (size_t)blockIdx.y * elems_per_image;
    float m = 0.0f;
    const long long n4 = elems_per_image >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(p)[i];
        m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < elems_per_image; i += 256) m = fmaxf(m, __builtin_fabsf(p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax + blockIdx.y, __float_as_int(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// Every image of a layer's weights in ONE launch (a training step packs every layer again after the optimizer, and again for
// its data-gradient conv: the launches, not the bytes, are what that costs): element e of
//   img32   the fp32 B-operand image ([(phase, chunk, n_tile)][k-group 4][half][n 32][4]; NULL: not in this pass),
//   img_bf  three bf16 terms per element,
//   img_h   two fp16 terms of w * 2^kw[co] (kw per output channel from conv_wscale_kernel, which runs first),
// the two 16-bit images in the chunk order g.s2d selects.
__global__ __launch_bounds__(256) void conv_pack_images_kernel(const float *__restrict__ w, float *__restrict__ img32,
                                                               unsigned short *__restrict__ img_bf,
                                                               unsigned short *__restrict__ img_h, ConvGeom g,
                                                               long long total, const int *__restrict__ hdr) {
    const int *kwtab = hdr + 64 + 32 * g.ntile;
    const int nchunk = g.ntaps * g.cpt;
    auto wat = [&](int ci, int co, int kyx) {
        return g.transposed ? w[((size_t)ci * g.Cout + co) * g.kk + kyx] : w[((size_t)co * g.Cin + ci) * g.kk + kyx];
    };
    // one thread per (phase, chunk, ntile, step, half, n, i) element
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        if (img32) {
            const int i = e & 3, n = (e >> 2) & 31, h = (e >> 7) & 1, j = (e >> 8) & 3;
            long long t = e >> 10;
            const int nt = (int)(t % g.ntile); t /= g.ntile;
            const int chunk = (int)(t % nchunk);
            const int phase = (int)(t / nchunk);
            const int tap = chunk / g.cpt, cc = chunk - tap * g.cpt;
            const int ci = cc * 32 + 16 * h + 4 * j + i, co = nt * 32 + n;
            img32[e] = (ci < g.Cin && co < g.Cout) ? wat(ci, co, g.kyx[phase][tap]) : 0.0f;
        }
        const int i = e & 7, n = (e >> 3) & 31, hh = (e >> 8) & 1, t = (e >> 9) & 1;
        long long r = e >> 10;
        const int nt = (int)(r % g.ntile); r /= g.ntile;
        const int chunk = (int)(r % nchunk);
        const int phase = (int)(r / nchunk);
        int ci, kyx;
        const int co = nt * 32 + n;
        if (g.s2d) {
            // chunk = cc * 4 + vt: cc = (input sub-position (py,px) of the 2x2 block) * cpt + 32-channel slice,
            // vt = (block offset ay + py, ax + px) in {0,1}^2;  ky = 2*ay + py + 1 (same for x)
            const int cc = chunk >> 2, vt = chunk & 3;
            const int sub = cc / g.cpt, sl = cc - sub * g.cpt;
            const int py = sub >> 1, px = sub & 1;
            const int ay = (vt >> 1) - py, ax = (vt & 1) - px;
            kyx = (2 * ay + py + 1) * 4 + (2 * ax + px + 1);
            ci = sl * 32 + 16 * hh + 8 * t + i;
        } else {
            const int tap = chunk / g.cpt, cc = chunk - tap * g.cpt;
            ci = cc * 32 + 16 * hh + 8 * t + i;
            kyx = g.kyx[phase][tap];
        }
        const float v = (ci < g.Cin && co < g.Cout) ? wat(ci, co, kyx) : 0.0f;
        const size_t pos = (size_t)((t * 2 + hh) * 32 + n) * 8 + i;
        const size_t cell = (size_t)(phase * nchunk + chunk) * g.ntile + nt;
        {
            const float vs = v * __builtin_ldexpf(1.0f, kwtab[co]);
            const _Float16 g1 = (_Float16)vs;
            const _Float16 g2 = (_Float16)(vs - (float)g1);
            img_h[cell * 2048 + pos] = __builtin_bit_cast(unsigned short, g1);
            img_h[cell * 2048 + 1024 + pos] = __builtin_bit_cast(unsigned short, g2);
        }
        {
            const unsigned short b1 = f32_to_bf16_rne(v);
            const float r1 = v - __uint_as_float((unsigned)b1 << 16);
            const unsigned short b2 = f32_to_bf16_rne(r1);
            const float r2 = r1 - __uint_as_float((unsigned)b2 << 16);
            const unsigned short b3 = f32_to_bf16_rne(r2);
            // image: [(phase*nchunk + chunk)*ntile + nt][term][t][hh][n][i]
            img_bf[cell * 3072 + pos] = b1;
            img_bf[cell * 3072 + 1024 + pos] = b2;
            img_bf[cell * 3072 + 2048 + pos] = b3;
        }
    }
}

// The waves-per-SIMD hint of 2 keeps the accumulators in VGPRs: without it the compiler parks them in AGPRs
// and copies all of them to VGPRs and back once per loop iteration (128 v_accvgpr moves per 96 MFMAs).
// H2: two-term fp16 products (split8_h); every pixel row carries the power-of-two scale of ITS image (taps never cross
// images, so the scale factors out of a row's accumulators); in_amax must then hold every image's maximum.
template <int NT, bool H2 = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf3_kernel(const float *__restrict__ in,
                                                             const u32x4 *__restrict__ wimg,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ out, ConvGeom g,
                                                             const int *__restrict__ whdr, const int *__restrict__ in_amax,
                                                             int *__restrict__ out_amax) {
    constexpr int TERMS = H2 ? 2 : 3;
    constexpr int CH4 = NT * 128 * TERMS;              // uint4 per chunk of this n-block (2 KiB per term and n-tile)
    __shared__ u32x4 Bs[2][CH4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int phase = blockIdx.y % g.nphase, nb = blockIdx.y / g.nphase;
    const long long M = (long long)g.B * g.Hg * g.Wg;
    const int nchunk = g.ntaps * g.cpt;
    const bool relu_in = g.flags & kFlagReluIn;
    const unsigned long long dym = g.dymask[phase], dxm = g.dxmask[phase];

    const long long img_px = (long long)g.Hg * g.Wg;
    const long long b_first = ((long long)blockIdx.x * 128) / img_px;
    const unsigned long long in_img_bytes = (unsigned long long)g.Hin * g.Win * g.Cin * 4ull;
    const auto in_rs = act_rsrc(in + (size_t)b_first * g.Hin * g.Win * g.Cin,
                                (unsigned long long)(g.B - b_first) * in_img_bytes);
    unsigned pbase, tapmask;
    long long myoff;
    long long myimg = -1;                              // image of this lane's pixel row (-1: past the end)
    float xsc = 1.0f, dsc = 1.0f;                      // H2: the image's scale 2^kx and its inverse (the weight rows' 2^-kw[n] join in the epilogue)
    {
        const long long p = (long long)blockIdx.x * 128 + wave * 32 + l31;
        const bool valid = p < M;
        const long long pc = valid ? p : 0;
        const long long b = pc / img_px;
        const int rem = (int)(pc - b * img_px);
        const int gy = rem / g.Wg, gx = rem - gy * g.Wg;
        const int iy0 = gy * g.istride, ix0 = gx * g.istride;
        pbase = (unsigned)((((b - b_first) * g.Hin + iy0) * g.Win + ix0) * g.Cin * 4 + 64 * h);
        unsigned m = 0;
        for (int t = 0; t < g.ntaps; ++t) {
            const int iy = iy0 + (int)((dym >> (4 * t)) & 15) - 8, ix = ix0 + (int)((dxm >> (4 * t)) & 15) - 8;
            if (valid && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) m |= 1u << t;
        }
        tapmask = m;
        myoff = valid ? ((b * g.Hout + gy * g.ostride + g.opy[phase]) * g.Wout + gx * g.ostride + g.opx[phase]) *
                            (long long)g.Cout
                      : -1;
        if (valid) myimg = b;
        if (H2 && valid) {
            const float mx = __int_as_float(in_amax[b]);
            int e = 15;
            if (mx > 0.0f && mx < 3.0e38f) (void)__builtin_frexpf(mx, &e);
            int kx = 15 - e;
            kx = kx > 100 ? 100 : (kx < -100 ? -100 : kx);
            xsc = __builtin_ldexpf(1.0f, kx);
            dsc = __builtin_ldexpf(1.0f, -kx);
        }
    }
    const u32x4 *wbase = wimg + ((size_t)phase * nchunk * g.ntile + (size_t)nb * NT) * (128 * TERMS);
    const size_t wchunk = (size_t)g.ntile * (128 * TERMS);

    // Software pipeline over 32-channel chunks (static register names, loop unrolled by two):
    //   raw A ring of depth 2: the registers of chunk c+1 are split while chunk c's MFMAs run, then
    //   immediately re-loaded with chunk c+3; weights for chunk c+1 go global -> registers at the top of
    //   iteration c and registers -> the other LDS buffer at its end (one barrier per chunk).
    f32x4 ra0[4], ra1[4];
    u32x4 b_nxt[CH4 / 256 > 0 ? CH4 / 256 : 1];
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;

    auto load_a = [&](int c, f32x4(&dst)[4]) {
        const int tap = c / g.cpt, cc = c - tap * g.cpt;
        const int dy = (int)((dym >> (4 * tap)) & 15) - 8, dx = (int)((dxm >> (4 * tap)) & 15) - 8;
        const int tapbytes = (dy * g.Win + dx) * g.Cin * 4;
        const unsigned soff = (unsigned)cc * 128u;
        const unsigned vo = ((tapmask >> tap) & 1u) ? pbase + (unsigned)tapbytes : kOobOffset;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dst[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, vo + 16 * j, soff, 0));
    };
    constexpr int NBQ = CH4 / 256, BREM = CH4 % 256;    // 384 uint4 per n-tile (three terms): NT = 1 leaves a 128-thread tail
    u32x4 b_tail = {0, 0, 0, 0};
    auto load_b = [&](int c) {
        const u32x4 *src = wbase + (size_t)c * wchunk;
#pragma unroll
        for (int q = 0; q < NBQ; ++q) b_nxt[q] = src[tid + 256 * q];
        if (BREM && tid < BREM) b_tail = src[tid + 256 * NBQ];
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NBQ; ++q) Bs[buf][tid + 256 * q] = b_nxt[q];
        if (BREM && tid < BREM) Bs[buf][tid + 256 * NBQ] = b_tail;
    };
    // split this lane's 16 channels of a chunk: MFMA step t covers channels 16h + 8t .. +7
    auto split_a = [&](f32x4(&raw)[4], u32x4(&S1)[2], u32x4(&S2)[2], u32x4(&S3)[2]) {
        if (relu_in) {
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = relu4(raw[j]);
        }
        if constexpr (H2) {
            split8_h(raw[0], raw[1], xsc, S1[0], S2[0]);
            split8_h(raw[2], raw[3], xsc, S1[1], S2[1]);
        } else {
            split8(raw[0], raw[1], S1[0], S2[0], S3[0]);
            split8(raw[2], raw[3], S1[1], S2[1], S3[1]);
        }
    };
    // one chunk: MFMAs of chunk c from (S1,S2,S3); meanwhile split chunk c+1 (raw `rn`) into (T1,T2,T3)
    // and re-issue `rn`'s loads for chunk c+3
    auto chunk_step = [&](int c, const u32x4(&S1)[2], const u32x4(&S2)[2], const u32x4(&S3)[2], f32x4(&rn)[4],
                          u32x4(&T1)[2], u32x4(&T2)[2], u32x4(&T3)[2]) {
        const bool more = c + 1 < nchunk;
        if (more) load_b(c + 1);
        const u32x4 *bs = Bs[c & 1];
        constexpr int NP = NT >= 2 ? 2 : 1;               // n-tiles interleaved per product chain
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (H2) {
                const f16x8 h1 = __builtin_bit_cast(f16x8, S1[t]), h2 = __builtin_bit_cast(f16x8, S2[t]);
#pragma unroll
                for (int n0 = 0; n0 < NT; n0 += NP) {
                    f16x8 G1[NP], G2[NP];
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const u32x4 *bp = bs + (n0 + u) * 256 + (t * 2 + h) * 32 + l31;
                        G1[u] = __builtin_bit_cast(f16x8, bp[0]);
                        G2[u] = __builtin_bit_cast(f16x8, bp[128]);
                    }
#pragma unroll
                    for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, G1[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, G2[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, G1[u], acc[n0 + u], 0, 0, 0);
                }
                if (t == 0 && more) split_a(rn, T1, T2, T3);
                continue;
            }
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, S1[t]), a2 = __builtin_bit_cast(bf16x8, S2[t]),
                         a3 = __builtin_bit_cast(bf16x8, S3[t]);
#pragma unroll
            for (int n0 = 0; n0 < NT; n0 += NP) {
                bf16x8 B1[NP], B2[NP], B3[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const u32x4 *bp = bs + (n0 + u) * 384 + (t * 2 + h) * 32 + l31;
                    B1[u] = __builtin_bit_cast(bf16x8, bp[0]);
                    B2[u] = __builtin_bit_cast(bf16x8, bp[128]);
                    B3[u] = __builtin_bit_cast(bf16x8, bp[256]);
                }
                // smallest terms first; consecutive MFMAs alternate accumulators (8-pass MFMAs have a
                // dependent latency above their issue interval)
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, B1[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B2[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B3[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, B1[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B2[u], acc[n0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < NP; ++u) acc[n0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, B1[u], acc[n0 + u], 0, 0, 0);
            }
            if (t == 0 && more) split_a(rn, T1, T2, T3);          // VALU work in the shadow of the MFMAs
        }
        if (more) {
            if (c + 3 < nchunk) load_a(c + 3, rn);
            store_b((c + 1) & 1);
        }
        __syncthreads();
    };

    u32x4 P1[2], P2[2], P3[2], Q1[2], Q2[2], Q3[2];
    load_a(0, ra0);
    if (nchunk > 1) load_a(1, ra1);
    load_b(0);
    store_b(0);
    split_a(ra0, P1, P2, P3);
    if (nchunk > 2) load_a(2, ra0);
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {
        chunk_step(c, P1, P2, P3, ra1, Q1, Q2, Q3);                       // splits chunk c+1 from ra1
        if (c + 1 < nchunk) chunk_step(c + 1, Q1, Q2, Q3, ra0, P1, P2, P3);   // splits chunk c+2 from ra0
    }

    const bool relu_out = g.flags & kFlagReluOut;
    float bv[NT], wd[NT];                              // bias and (H2) weight-row scale 2^-kw[n] of this lane's output channels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (nb * NT + nt) * 32 + l31;
        bv[nt] = (bias && n < g.Cout) ? bias[n] : 0.0f;
        wd[nt] = H2 ? h2_dw(whdr)[n] : 1.0f;
    }
    // maxima for the next layer: one image per wave in the common case (one wave-wide reduction), per pixel row otherwise
    const long long img0 = __shfl(myimg, 0);
    const bool one_img = out_amax && __builtin_amdgcn_ballot_w64(myimg != img0) == 0 && img0 >= 0;
    float omax = 0.0f;
    // data-gradient epilogue (ep_add / ep_mask): every value this lane will need is requested up front -- one memory round trip
    // for the whole tile instead of one per store
    const bool ep = g.ep_add || g.ep_mask;
    float ea[16][NT], em[16][NT];
    if (ep) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long off = __shfl(myoff, (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = (nb * NT + nt) * 32 + l31;
                const bool ok = off >= 0 && n < g.Cout;
                ea[r][nt] = (g.ep_add && ok) ? g.ep_add[off + n] : 0.0f;
                em[r][nt] = (g.ep_mask && ok) ? g.ep_mask[off + n] : 1.0f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int src = (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long off = __shfl(myoff, src);
        const float drow = H2 ? __shfl(dsc, src) : 1.0f;
        float rmax = 0.0f;
        if (off >= 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = (nb * NT + nt) * 32 + l31;
                if (n < g.Cout) {
                    float v = (H2 ? acc[nt][r] * drow * wd[nt] : acc[nt][r]) + bv[nt];
                    if (relu_out) v = fmaxf(v, 0.0f);
                    rmax = fmaxf(rmax, __builtin_fabsf(v));
                    if (ep) v = em[r][nt] > 0.0f ? v + ea[r][nt] : 0.0f;
                    out[off + n] = v;
                }
            }
        }
        if (out_amax && !one_img) {
            const long long rimg = __shfl(myimg, src);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o));
            if (l31 == 0 && rimg >= 0) atomicMax(out_amax + rimg, __float_as_int(rmax));
        } else {
            omax = fmaxf(omax, rmax);
        }
    }
    if (one_img) publish_amax(out_amax, img0, omax, lane);
}


// ---------------------------------------------------------------------------
// Split-bf16 implicit GEMM for layers whose INPUT map is 8x8 and is sampled at stride 1 (the reference's
// enc conv 3x3, dec convT 3x3, dec convT 4x4 s2 phases, 1x1 pre-quantisation conv at 32x32 images).
// One wave owns one whole input image (64 pixels = two 32-pixel MFMA tiles) and 64 output channels:
//   for each 32-channel chunk:   (reduction order: chunk outer, tap inner)
//       the image's chunk is loaded ONCE (one contiguous 128 B per pixel), ReLU'd, split ONCE into three bf16
//       terms and parked in a wave-private LDS tile with an all-zero padding pixel;
//       every tap reads its A operands from that tile at a shifted pixel index (ds_read_b128).
//   Weights: the same three-term chunk images as conv_igemm_bf3_kernel, streamed through double-buffered LDS
//   and shared by the workgroup's four images (256 pixels x 64 channels per workgroup and chunk).
// Versus conv_igemm_bf3_kernel: activation traffic through L1/TA and the split VALU work drop by the number
// of taps (9x / 4x), and the per-load tap decode disappears.
// S2D: the 4x4 stride-2 conv on a 16x16 map, read as a conv over the 8x8 grid of 2x2 input blocks: a chunk is
// (sub-position (py,px) of the block, 32-channel slice) and meets four block offsets ("virtual taps"), so every
// input element is still split once and used four times (weights in the s2d chunk order, conv_pack_images_kernel).
// H2: two-term fp16 products with per-image activation scale and per-layer weight scale (see split8_h) instead of the
// three-term bf16 products; whdr = the weight image's header {kw}.
template <int NT, bool S2D, int NW, int WB = 2, bool H2 = false>
__global__ __launch_bounds__(NW * 64, 2) void conv_tile8_bf3_kernel(const float *__restrict__ in,
                                                                const u32x4 *__restrict__ wimg,
                                                                const float *__restrict__ bias,
                                                                float *__restrict__ out, ConvGeom g, int ny, const int *__restrict__ whdr,
                                                                const int *__restrict__ in_amax, int *__restrict__ out_amax) {
    constexpr int MT = 2, PX = 64, PLANE = (PX + 1) * 2;        // u32x4 per (k-step, term) plane: [half][pixel + zero]
    constexpr int HP = PX + 1;                                   // (consecutive lanes = consecutive 16 B: no bank conflicts)
    constexpr int TERMS = H2 ? 2 : 3;
    constexpr int TILE4 = 2 * TERMS * PLANE;                     // [k-step 2][term][PLANE]
    constexpr int CH4 = NT * 128 * TERMS;
    __shared__ u32x4 Bs[WB][CH4];                               // WB = 1: one weight buffer, two barriers per iteration (two workgroups per CU)
    __shared__ u32x4 As_all[NW * TILE4];                        // NW waves = NW images per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    // 1-D grid of (image group bx) x (phase / n-block by) workgroups.  The ny workgroups that read the same four
    // images are adjacent slots of ONE XCD's dispatch sequence (workgroup i goes to XCD i % 8), so the re-reads of
    // an image by its other output phases / n-blocks hit that XCD's L2 instead of going back to the fabric.
    unsigned bx, by;
    {
        const unsigned id = blockIdx.x, nxb = gridDim.x / (unsigned)ny;
        if ((nxb & 7u) == 0) {
            const unsigned slot = id >> 3;
            by = slot % (unsigned)ny;
            bx = (slot / (unsigned)ny) * 8 + (id & 7u);
        } else {
            by = id % (unsigned)ny;
            bx = id / (unsigned)ny;
        }
    }
    const int phase = by % g.nphase, nb = by / g.nphase;
    const bool relu_in = g.flags & kFlagReluIn, relu_out = g.flags & kFlagReluOut;
    const unsigned long long dym = g.dymask[phase], dxm = g.dxmask[phase];
    const int ntaps = S2D ? 4 : g.ntaps, cpt = S2D ? 4 * g.cpt : g.cpt, nchunk = ntaps * cpt;

    if (lane < 4 * TERMS) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};     // padding pixels

    const long long img = (long long)bx * NW + wave;
    const bool img_ok = img < g.B;
    // this lane's pixel row (S2D: the top-left pixel of this lane's 2x2 input block)
    const float *src = S2D ? in + (((size_t)(img_ok ? img : 0) * 16 + 2 * (lane >> 3)) * 16 + 2 * (lane & 7)) * g.Cin
                           : in + ((size_t)(img_ok ? img : 0) * PX + lane) * g.Cin;

    int spx[MT];
    unsigned tapok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, x = spx[mt] & 7;
        unsigned m = 0;
        if (S2D) {
            // bit sub*4 + vt: block offset (vt>>1) - py, (vt&1) - px
            for (int q = 0; q < 16; ++q) {
                const int yy = y + ((q >> 1) & 1) - (q >> 3), xx = x + (q & 1) - ((q >> 2) & 1);
                if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << q;
            }
        } else {
            for (int t = 0; t < ntaps; ++t) {
                const int yy = y + (int)((dym >> (4 * t)) & 15) - 8, xx = x + (int)((dxm >> (4 * t)) & 15) - 8;
                if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << t;
            }
        }
        tapok[mt] = m;
    }

    const u32x4 *wbase = wimg + ((size_t)phase * nchunk * g.ntile + (size_t)nb * NT) * (128 * TERMS);
    const size_t wchunk = (size_t)g.ntile * (128 * TERMS);
    constexpr int NBQ = CH4 / (NW * 64);               // u32x4 of the weight chunk per thread
    static_assert(CH4 % (NW * 64) == 0, "weight chunk must divide over the workgroup");
    u32x4 b_nxt[NBQ];
    auto load_b = [&](int c) {
        const u32x4 *p = wbase + (size_t)c * wchunk;
#pragma unroll
        for (int q = 0; q < NBQ; ++q) b_nxt[q] = p[tid + NW * 64 * q];
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NBQ; ++q) Bs[buf][tid + NW * 64 * q] = b_nxt[q];
    };
    float xscale = 1.0f, descale = 1.0f;             // H2: image scale 2^kx and its inverse (x the weight row's 2^-kw[n] in the epilogue)
    f32x4 raw[8];
    auto load_raw = [&](int cc) {
        const float *q = src + 32 * cc;
        if (S2D) {
            const int sub = cc / g.cpt, sl = cc - sub * g.cpt;
            q = src + ((sub >> 1) * 16 + (sub & 1)) * g.Cin + 32 * sl;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4 *>(q + 4 * j);
    };
    // park chunk `raw`: k-step t, operand half hh hold channels 16*hh + 8*t + [0,8) (the weight image's order)
    auto stage = [&]() {
        if (relu_in) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = relu4(raw[j]);
        }
        u32x4 *dst = As + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr (H2) {
                    u32x4 t1, t2;
                    split8_h(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], xscale, t1, t2);
                    dst[(t * 2 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 2 + 1) * PLANE + hh * HP] = t2;
                } else {
                    u32x4 t1, t2, t3;
                    split8(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], t1, t2, t3);
                    dst[(t * 3 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 3 + 1) * PLANE + hh * HP] = t2;
                    dst[(t * 3 + 2) * PLANE + hh * HP] = t3;
                }
            }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    if constexpr (H2) {
        // the image's largest |x| (after the input ReLU) -> its power-of-two scale; the image is read again below (L2)
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;        // the producer's maximum of this image, if any
        if (given >= 0) m = __int_as_float(given);
        else for (int c2 = 0; c2 < cpt; ++c2) {
            load_raw(c2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x4 v = raw[j];
                if (relu_in) v = relu4(v);
                m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
            }
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        xscale = __builtin_ldexpf(1.0f, kx);
        descale = __builtin_ldexpf(1.0f, -kx);
    }

    // iteration it = cc * ntaps + tap  ->  weight chunk tap * cpt + cc
    const int niter = nchunk;
    load_raw(0);
    load_b(0);
    store_b(0);
    if (niter > 1) load_b(S2D ? 1 : (ntaps > 1 ? cpt : 1));
    int cc = 0, tap = 0;
    for (int it = 0; it < niter; ++it) {
        if (tap == 0) {
            // the tile is wave-private and a wave's LDS operations execute in order: the previous chunk's
            // operand reads are behind us, no barrier needed to overwrite it
            stage();
            if (cc + 1 < cpt) load_raw(cc + 1);
        }
#if !defined(TILE8_KNOB) || TILE8_KNOB != 4      // 4 = no workgroup barrier in the main loop (races: timing only)
        __syncthreads();                                   // weights of this iteration + (tap 0) the fresh tile
#else
        lds_order_wave();
#endif
        const u32x4 *bs = Bs[WB == 2 ? (it & 1) : 0];
        int shift, okbit;
        if (S2D) {
            const int sub = cc / g.cpt;
            shift = ((tap >> 1) - (sub >> 1)) * 8 + ((tap & 1) - (sub & 1));
            okbit = sub * 4 + tap;
        } else {
            shift = ((int)((dym >> (4 * tap)) & 15) - 8) * 8 + ((int)((dxm >> (4 * tap)) & 15) - 8);
            okbit = tap;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if constexpr (H2) {
                u32x4 A1[MT], A2[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = ((tapok[mt] >> okbit) & 1u) ? spx[mt] + shift : PX;
                    const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                    A1[mt] = ap[0];
                    A2[mt] = ap[PLANE];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#if defined(TILE8_KNOB) && TILE8_KNOB == 3      // 3 = the weight operands of n-tile 0 for every n-tile (12 instead of 24 LDS reads per step pair)
                    const u32x4 *bp = bs + (t * 2 + h) * 32 + l31;
#else
                    const u32x4 *bp = bs + nt * 256 + (t * 2 + h) * 32 + l31;
#endif
#if defined(TILE8_KNOB) && TILE8_KNOB == 1      // timing-only knock-outs (wrong results): 1 = no MFMAs (operands still read)
                    asm volatile("" :: "v"(A1[0]), "v"(A2[0]), "v"(A1[1]), "v"(A2[1]), "v"(bp[0]), "v"(bp[128]));
                    continue;
#endif
                    prod3x2(A1[0], A2[0], A1[1], A2[1], bp[0], bp[128], acc[0][nt], acc[1][nt]);
                }
                continue;
            }
            bf16x8 A[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = ((tapok[mt] >> okbit) & 1u) ? spx[mt] + shift : PX;
                const u32x4 *ap = As + (t * 3) * PLANE + h * HP + p;
                A[mt][0] = __builtin_bit_cast(bf16x8, ap[0]);
                A[mt][1] = __builtin_bit_cast(bf16x8, ap[PLANE]);
                A[mt][2] = __builtin_bit_cast(bf16x8, ap[2 * PLANE]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 *bp = bs + nt * 384 + (t * 2 + h) * 32 + l31;
                const bf16x8 B1 = __builtin_bit_cast(bf16x8, bp[0]), B2 = __builtin_bit_cast(bf16x8, bp[128]),
                             B3 = __builtin_bit_cast(bf16x8, bp[256]);
                // smallest terms first; the two pixel tiles alternate accumulators
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][2], B1, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], B2, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B3, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][1], B1, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B2, acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][0], B1, acc[mt][nt], 0, 0, 0);
            }
        }
        // next iteration's weights: registers -> the other LDS buffer (its last readers passed the barrier above);
        // then fetch the iteration after that
        int ntap = tap + 1, ncc = cc;
        if (ntap == ntaps) { ntap = 0; ++ncc; }
#if defined(TILE8_KNOB) && TILE8_KNOB == 2      // 2 = weights loaded once (no weight stream through global -> LDS after the first two chunks)
        if (it >= 1) { tap = ntap; cc = ncc; continue; }
#endif
        if (it + 1 < niter) {
            if (WB == 1) __syncthreads();                  // single buffer: every wave is done reading this iteration's weights
            store_b(WB == 2 ? ((it + 1) & 1) : 0);
            int t2 = ntap + 1, c2 = ncc;
            if (t2 == ntaps) { t2 = 0; ++c2; }
            if (it + 2 < niter) load_b(S2D ? c2 * 4 + t2 : t2 * cpt + c2);
        }
        tap = ntap; cc = ncc;
    }

    float bv[NT], wd[NT];                            // bias, (H2) accumulator scale 2^-(kx + kw[n]) of this lane's output channels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (nb * NT + nt) * 32 + l31;
        bv[nt] = (bias && n < g.Cout) ? bias[n] : 0.0f;
        wd[nt] = H2 ? descale * h2_dw(whdr)[n] : 1.0f;
    }
    float omax = 0.0f;
    if (img_ok && (g.Cout & 7) == 0) {
        // the operand tile is free now (wave-private): stage the outputs through it, 16-byte stores
        float *tile = reinterpret_cast<float *>(As);
        // pass k of tile (mt, nt): pixel 32 mt + (lane >> 3) + 8 k, channels n0 .. n0 + 3
        auto out_off = [&](int mt, int k) {
            const int px = 32 * mt + (lane >> 3) + 8 * k;
            const int gy = px >> 3, gx = px & 7;
            return ((img * g.Hout + gy * g.ostride + g.opy[phase]) * g.Wout + gx * g.ostride + g.opx[phase]) * (long long)g.Cout;
        };
        // data-gradient epilogue (ep_add / ep_mask): the NEXT tile's sixteen-byte groups are requested while this tile goes
        // through the LDS tile, so that the stores do not wait a memory round trip each
        // (the two-tile kernels request the tile's own groups in front of its staging instead: one set of registers fewer keeps
        // them at three waves per SIMD)
        const bool ep = g.ep_add || g.ep_mask;
        constexpr int AHEAD = NT >= 4 ? 1 : 0;
        f32x4 pa[1 + AHEAD][4], pm[1 + AHEAD][4];
        auto ep_fetch = [&](int tix, int slot) {
            const int mt = tix / NT, nt = tix % NT, n0 = (nb * NT + nt) * 32 + 4 * (lane & 7);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long long o = out_off(mt, k) + n0;
                pa[slot][k] = (g.ep_add && n0 < g.Cout) ? *reinterpret_cast<const f32x4 *>(g.ep_add + o) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                pm[slot][k] = (g.ep_mask && n0 < g.Cout) ? *reinterpret_cast<const f32x4 *>(g.ep_mask + o) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
            }
        };
        if (ep && AHEAD) ep_fetch(0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int tix = mt * NT + nt;
                if (ep && AHEAD && tix + 1 < MT * NT) ep_fetch(tix + 1, (tix + 1) & 1);
                if (ep && !AHEAD) ep_fetch(tix, 0);
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = (H2 ? acc[mt][nt][r] * wd[nt] : acc[mt][nt][r]) + bv[nt];
                    if (relu_out) v[r] = fmaxf(v[r], 0.0f);
                    omax = fmaxf(omax, __builtin_fabsf(v[r]));
                }
                tile_epilogue(tile, v, lane, (nb * NT + nt) * 32, [&](int p, int n, f32x4 a, int k) {
                    (void)p;
                    const long long off = out_off(mt, k);
                    if (ep) {
                        const f32x4 m = pm[tix & AHEAD][k];
                        a += pa[tix & AHEAD][k];
                        a.x = m.x > 0.0f ? a.x : 0.0f; a.y = m.y > 0.0f ? a.y : 0.0f; a.z = m.z > 0.0f ? a.z : 0.0f; a.w = m.w > 0.0f ? a.w : 0.0f;
                    }
                    if (n < g.Cout)                            // Cout % 8 == 0: a 4-channel group is all in or all out
                        *reinterpret_cast<f32x4 *>(out + off + n) = a;
                });
            }
    } else if (img_ok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gy = px >> 3, gx = px & 7;
                const long long off = ((img * g.Hout + gy * g.ostride + g.opy[phase]) * g.Wout + gx * g.ostride +
                                       g.opx[phase]) * (long long)g.Cout;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = (nb * NT + nt) * 32 + l31;
                    if (n < g.Cout) {
                        float v = (H2 ? acc[mt][nt][r] * wd[nt] : acc[mt][nt][r]) + bv[nt];
                        if (relu_out) v = fmaxf(v, 0.0f);
                        omax = fmaxf(omax, __builtin_fabsf(v));
                        out[off + n] = ep_apply1(g, off + n, v);
                    }
                }
            }
    }
    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
}

// ---------------------------------------------------------------------------
// The tile-resident scheme of conv_tile8_bf3_kernel on maps LARGER than 8x8 (round 3; BASELINE configs 4 / 5: 56x56 and 64x64
// latent maps): one wave owns one 8x8 TILE of one image's pixel grid plus a one-pixel halo -- a 10x10 input patch -- and NT
// 32-channel output tiles.  Layers whose input is sampled at stride 1 on a grid that is a multiple of 8 both ways: the 3x3
// conv, the 3x3 conv-transpose, the four phases of the 4x4 stride-2 conv-transpose, and (S2D) the 4x4 stride-2 conv read
// as a 2x2 conv over the grid of 2x2 input blocks.  Per SIXTEEN-channel slice the patch is loaded ONCE (64 contiguous bytes
// per pixel; pixels outside the image read as zero through the buffer descriptor), ReLU'd, split once into its two fp16
// terms and parked in the wave's LDS tile (6.5 KiB); every tap then reads its operands at a shifted patch index -- no
// per-tap reload / re-split as in conv_igemm_bf3_kernel, no border masks.  Two-term fp16 products only (split8_h): the scale
// is the image's maximum handed over by the producing layer (in_amax), or the patch's own maximum where none is given (any
// power of two that covers the patch is exact).
// Weights stream through two LDS buffers by LDS-DMA (no staging registers), one STAGE = TPS taps of one slice for the NT
// output tiles (TPS * NT * 2 KiB), requested a stage ahead right behind the stage barrier; the DMA gathers the slice's
// units out of the packed image's 32-channel chunks ([term][k-step t][half h][32 lanes]: the 16-channel slice s' is the
// units (t = lane half, h = s')), so the image conv_igemm_bf3_kernel reads serves unchanged.  Four waves per workgroup and
// TWO workgroups per CU (<= 75 KiB of LDS each): their prologues, conversions, barriers and epilogues fall into each
// other's matrix phases -- with one eight-wave workgroup per CU these were 45 % of a wave's time with the CU's matrix
// pipes idle (in-kernel stamps, profiles/r03_notes.txt).
template <int NT, int TPS, bool S2D = false, int NPH = 1, int HALO = 1>
__global__ __launch_bounds__(256, 2) void conv_halo8_h2_kernel(const float *__restrict__ in, const u32x4 *__restrict__ wimg,
                                                              const float *__restrict__ bias, float *__restrict__ out,
                                                              ConvGeom g, int ny, const int *__restrict__ whdr,
                                                              const int *__restrict__ in_amax, int *__restrict__ out_amax) {
    // HALO = 0: the 1x1 conv (one tap, no border): the patch is the tile itself
    constexpr int NW = 4, MT = 2, PW = 8 + 2 * HALO, PP = PW * PW;         // patch: 10 x 10 pixels
    constexpr int HP = PP + 1, PLANE = HP * 2;                   // u32x4 per term plane: [half][patch pixel]
    constexpr int TILE = 2 * PLANE;                              // [term 2][PLANE]: one 16-channel slice of the patch
    // NPH = 2: two output PHASES of the 4x4 stride-2 conv-transpose per wave, (py, 0) and (py, 1) (they read the same patch:
    // one load / ReLU / split for both); a phase's NT channel tiles are then tiles i * NT .. of NV "virtual" ones
    constexpr int NV = NPH * NT;
    constexpr int NPIECE = TPS * NV * 2;                         // 1 KiB pieces of a weight stage: [tap][phase][nt][term]
    constexpr int WST = NPIECE * 64;
    static_assert(NPH == 1 || (NPH == 2 && !S2D), "phase pairs are the conv-transpose's");
    static_assert(NPIECE % NW == 0, "a stage's pieces divide over the waves");
    static_assert(TILE * 16 >= 32 * 32 * 4, "the epilogue stages a 32 x 32 float tile in the wave's operand tile");
    __shared__ u32x4 Bs[2 * WST];
    __shared__ u32x4 As_all[NW * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    u32x4 *As = As_all + wave * TILE;
    unsigned bx, by;
    {
        const unsigned id = blockIdx.x, nxb = gridDim.x / (unsigned)ny;
        if ((nxb & 7u) == 0) {
            const unsigned slot = id >> 3;
            by = slot % (unsigned)ny;
            bx = (slot / (unsigned)ny) * 8 + (id & 7u);
        } else {
            by = id % (unsigned)ny;
            bx = id / (unsigned)ny;
        }
    }
    const int nps = g.nphase / NPH;                              // phase sets
    const int phase0 = (int)(by % (unsigned)nps) * NPH, nb = by / nps;
    const bool relu_in = g.flags & kFlagReluIn, relu_out = g.flags & kFlagReluOut;
    unsigned long long dym[NPH], dxm[NPH];
#pragma unroll
    for (int i = 0; i < NPH; ++i) {
        dym[i] = g.dymask[phase0 + i];
        dxm[i] = g.dxmask[phase0 + i];
    }
    // S2D: the 4x4 stride-2 conv read as a conv over the grid of 2x2 input blocks (conv_tile8_bf3_kernel<., true>): a patch
    // "pixel" is a block, a 32-channel chunk = (sub-position of the block, 32-channel slice) meets four block offsets
    const int ntaps = S2D ? 4 : g.ntaps, cpt = S2D ? 4 * g.cpt : g.cpt;
    const int nslice = 2 * cpt, ngrp = ntaps / TPS, nstage = nslice * ngrp;

    // this wave's tile
    const int tx_n = g.Wg >> 3, ty_n = g.Hg >> 3;
    const long long tile_id = (long long)bx * NW + wave_u;       // (wave-uniform: a lane-derived one costs a waterfall loop per load)
    const long long ntile_all = (long long)g.B * ty_n * tx_n;
    const bool img_ok = tile_id < ntile_all;
    const long long tq = img_ok ? tile_id : 0;
    const long long img = tq / (ty_n * tx_n);
    const int trem = (int)(tq - img * (ty_n * tx_n));
    const int y0 = (trem / tx_n) * 8, x0 = (trem % tx_n) * 8;
    // patch pixels of this lane: q = lane and q = 64 + lane (< 100); byte offset inside the image, out of range = zero
    const auto rs = act_rsrc(in + (size_t)img * g.Hin * g.Win * g.Cin, img_ok ? (unsigned long long)g.Hin * g.Win * g.Cin * 4ull : 0ull);
    unsigned poff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = 64 * k + lane;
        const int iy = y0 - HALO + q / PW, ix = x0 - HALO + q % PW;   // pixel (S2D: block) coordinates
        if (S2D) poff[k] = (q < PP && iy >= 0 && 2 * iy < g.Hin && ix >= 0 && 2 * ix < g.Win) ? (unsigned)((2 * iy * g.Win + 2 * ix) * g.Cin) * 4u : kOobOffset;
        else poff[k] = (q < PP && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win) ? (unsigned)((iy * g.Win + ix) * g.Cin) * 4u : kOobOffset;
    }
    int spx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = 32 * mt + l31;
        spx[mt] = ((p >> 3) + HALO) * PW + (p & 7) + HALO;       // the tile pixel's place in the patch
    }

    // weight stage s = (slice s / ngrp, tap group s % ngrp): piece p = (tap of the group, nt, term), this wave's are
    // p = wave + 4 j; unit of lane (h', l31) in the packed chunk (32-channel chunk sl >> 1, output tile): term * 128 + h' * 64 +
    // (sl & 1) * 32 + l31
    // (scalar base + this lane's constant byte offset: no vector instruction per piece, no address register to wait for)
    const u32x4 *wbase = wimg + ((size_t)phase0 * ntaps * cpt * g.ntile + (size_t)nb * NT) * 256;
    const size_t wphase = (size_t)ntaps * cpt * g.ntile * 256;
    const unsigned wlane = (unsigned)(h * 64 + l31) * 16u;
    const size_t wchunk = (size_t)g.ntile * 256;
    const unsigned bs_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)Bs);
    auto dma = [&](const u32x4 *src_uniform, unsigned lds) {      // lds: byte address of the 1 KiB piece
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(wlane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // this wave's pieces p = wave + 4 j: their part of the source offset that does not depend on the stage (tap within the
    // group, phase, channel tile, term) is worked out ONCE -- scalar instructions have a shared issue slot too
    const size_t wtap = (S2D ? (size_t)1 : (size_t)cpt) * wchunk;            // units from a tap to the next
    size_t wpiece[NPIECE / NW];
#pragma unroll
    for (int j = 0; j < NPIECE / NW; ++j) {
        const int p = wave_u + NW * j;
        const int tl = p / (NV * 2), ph = (p / (NT * 2)) % NPH, nt = (p >> 1) % NT, term = p & 1;
        wpiece[j] = (size_t)tl * wtap + (size_t)ph * wphase + (size_t)nt * 256 + term * 128;
    }
    auto dma_stage = [&](int sl, int grp, int buf) {
        // stage (slice sl, tap group grp): chunk = tap * cpt + (sl >> 1) (S2D: (sl >> 1) * 4 + tap), k-step half sl & 1
        const u32x4 *sp = wbase + (size_t)(sl >> 1) * (S2D ? 4 * wchunk : wchunk) + (size_t)(grp * TPS) * wtap + (sl & 1) * 32;
        const unsigned dp = bs_lds + (unsigned)(buf * WST + wave_u * 64) * 16u;
#pragma unroll
        for (int j = 0; j < NPIECE / NW; ++j) dma(sp + wpiece[j], dp + (unsigned)(j * NW * 64) * 16u);
    };
    dma_stage(0, 0, 0);

    float xscale = 1.0f, descale = 1.0f;
    f32x4 raw[2][4];
    auto load_raw = [&](int sl) {                          // slice sl: 16 channels = 64 contiguous bytes per pixel
        unsigned co = (unsigned)(16 * sl) * 4u;
        if (S2D) {
            const int c32 = sl >> 1, sub = c32 / g.cpt, s32 = c32 - sub * g.cpt;
            co = (unsigned)(((sub >> 1) * g.Win + (sub & 1)) * g.Cin + 32 * s32 + 16 * (sl & 1)) * 4u;
        }
#pragma unroll
        for (int k = 0; k < (PP > 64 ? 2 : 1); ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                raw[k][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k] + (unsigned)(4 * j) * 4u, co, 0));
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < (PP > 64 ? 2 : 1); ++k) {
            if (k == 1 && lane >= PP - 64) break;
            if (relu_in) {
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[k][j] = relu4(raw[k][j]);
            }
            u32x4 *dst = As + 64 * k + lane;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                u32x4 t1, t2;
                split8_h(raw[k][2 * hh], raw[k][2 * hh + 1], xscale, t1, t2);
                dst[hh * HP] = t1;
                dst[PLANE + hh * HP] = t2;
            }
        }
    };

    f32x16 acc[MT][NV];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NV; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    {
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;        // the producer's maximum of this image, if any
        if (given >= 0) m = __int_as_float(given);
        else for (int s2 = 0; s2 < nslice; ++s2) {                        // else the patch's own (one more pass over it)
            load_raw(s2);
#pragma unroll
            for (int k = 0; k < (PP > 64 ? 2 : 1); ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = raw[k][j];
                    if (relu_in) v = relu4(v);
                    m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
                }
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        xscale = __builtin_ldexpf(1.0f, kx);
        descale = __builtin_ldexpf(1.0f, -kx);
    }

    load_raw(0);
    int sl = 0, grp = 0;
#pragma unroll 1
    for (int s = 0; s < nstage; ++s) {
        if (grp == 0) stage();                         // (wave-private tile, LDS operations of a wave execute in order)
        // operand offsets of the stage's taps; the first tap's operands are requested in front of the barrier
        int shift[TPS][NPH];
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {
            const int tap = grp * TPS + tl;
#pragma unroll
            for (int i = 0; i < NPH; ++i) {
                if (S2D) {
                    const int sub = (sl >> 1) / g.cpt;
                    shift[tl][i] = ((tap >> 1) - (sub >> 1)) * PW + ((tap & 1) - (sub & 1));
                } else {
                    shift[tl][i] = ((int)((dym[i] >> (4 * tap)) & 15) - 8) * PW + ((int)((dxm[i] >> (4 * tap)) & 15) - 8);
                }
            }
        }
        // (one tap's operands ahead; with two phases per pass the registers allow the current tap's only)
        constexpr int NAB = NPH == 1 ? 2 : 1;
        u32x4 A1[NAB][NPH][MT], A2[NAB][NPH][MT];
        auto ldA = [&](int tl) {
#pragma unroll
            for (int i = 0; i < NPH; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const u32x4 *ap = As + h * HP + spx[mt] + shift[tl][i];
                    A1[tl % NAB][i][mt] = ap[0];
                    A2[tl % NAB][i][mt] = ap[PLANE];
                }
        };
        ldA(0);
        // this stage's weights are in (a slice's load_raw behind the previous barrier may still be in flight: its eight
        // loads are the youngest), everyone is done with the other buffer
        const bool raw_behind = s > 0 && grp == (ngrp > 1 ? 1 : 0) && (ngrp > 1 ? sl : sl - 1) + 1 < nslice;
        if (raw_behind && PP > 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (raw_behind) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < nstage) dma_stage(grp + 1 == ngrp ? sl + 1 : sl, grp + 1 == ngrp ? 0 : grp + 1, (s + 1) & 1);
        if (grp == 0 && sl + 1 < nslice) load_raw(sl + 1);
        const u32x4 *bs = Bs + (s & 1) * WST + lane;
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {
            if (NAB == 2 && tl + 1 < TPS) ldA(tl + 1);
            if (NAB == 1 && tl > 0) ldA(tl);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const u32x4 *bp = bs + (tl * NV + v) * 128;
                const int i = v / NT;
                prod3x2(A1[tl % NAB][i][0], A2[tl % NAB][i][0], A1[tl % NAB][i][1], A2[tl % NAB][i][1], bp[0], bp[64], acc[0][v], acc[1][v]);
            }
        }
        if (++grp == ngrp) { grp = 0; ++sl; }
    }

    float bv[NT], wd[NT];                              // bias and accumulator scale 2^-(kx + kw[n]) of this lane's output channels
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (nb * NT + nt) * 32 + l31;
        bv[nt] = (bias && n < g.Cout) ? bias[n] : 0.0f;
        wd[nt] = descale * h2_dw(whdr)[n];
    }
    float omax = 0.0f;
    if (img_ok) {
        // the operand tile is free now (wave-private): the outputs go through it (accumulator layout in, whole 128-byte
        // pixel rows out) and leave as 16-byte stores at a scalar row base + this lane's constant offset.  Few vector
        // instructions on purpose: whatever a wave issues here waits behind the other workgroup's MFMAs.
        float *tile = reinterpret_cast<float *>(As);
        float *obase[NPH];
#pragma unroll
        for (int i = 0; i < NPH; ++i)
            obase[i] = out + ((img * g.Hout + (long long)y0 * g.ostride + g.opy[phase0 + i]) * g.Wout + (long long)x0 * g.ostride +
                              g.opx[phase0 + i]) * (long long)g.Cout + (size_t)nb * NT * 32;
        const size_t orow = (size_t)g.ostride * g.Wout * g.Cout;                  // floats from a tile row to the next
        const unsigned olane = (unsigned)((lane >> 3) * g.ostride * g.Cout + 4 * (lane & 7)) * 4u;
        const bool nok = (nb * NT) * 32 + 4 * (lane & 7) < g.Cout;               // Cout % 32 == 0 (ntile even): all tiles alike
        auto finish = [&](auto RO) {                       // (one straight-line copy per ReLU flag: no branches inside)
            constexpr bool ro = decltype(RO)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int nt = v % NT, i = v / NT;
                    const f32x2v d2 = {wd[nt], wd[nt]}, b2 = {bv[nt], bv[nt]};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        // acc * 2^-k + bias: the product is exact, so the fused form rounds once like the separate add
                        const f32x2v y = __builtin_elementwise_fma(f32x2v{acc[mt][v][r], acc[mt][v][r + 1]}, d2, b2);
                        float v0 = y.x, v1 = y.y;
                        if (ro) { v0 = vmax(v0, 0.0f); v1 = vmax(v1, 0.0f); }
                        vmax3_abs(omax, v0, v1);
                        tile[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = v0;
                        tile[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * h) * 32 + l31] = v1;
                    }
                    lds_order_wave();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 q = *reinterpret_cast<const f32x4 *>(tile + k * 256 + lane * 4);
                        float *urow = obase[i] + (size_t)(4 * mt + k) * orow + nt * 32;   // wave-uniform
                        if (nok) *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(urow) + olane) = q;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
        };
        if (relu_out) finish(std::true_type{});
        else finish(std::false_type{});
    }
    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
}

// ---------------------------------------------------------------------------
// Fused residual layer on the split-bf16 product path (same math and layout as res_layer_kernel below;
// see conv_igemm_bf3_kernel for the split).  GEMM1 (3x3, C -> 32 hidden) is barrier-free: each wave
// reads its 6-KiB weight chunk (three bf16 terms) straight from L1/L2 next to its A operands.
// GEMM2 (1x1, 32 -> C) takes the three-term W2 image from LDS.
// H2: two-term fp16 products; every pixel row carries its image's scale in the 3x3 GEMM (in_amax holds the maxima) and
// its OWN scale (largest of its 32 hidden values) in the 1x1 GEMM, whose rows are independent.
template <int NT2, bool H2 = false>
__global__ __launch_bounds__(256, 2) void res_layer_bf3_kernel(const float *__restrict__ in,
                                                            const u32x4 *__restrict__ w1img,
                                                            const u32x4 *__restrict__ w2img,
                                                            float *__restrict__ out, int B, int H, int W,
                                                            int C, int flags, const int *__restrict__ hdr1,
                                                            const int *__restrict__ hdr2, const int *__restrict__ in_amax,
                                                            int *__restrict__ out_amax) {
    constexpr int MT = 2, TERMS = H2 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) float smem_res[NT2 * 1536 + 4 * MT * 32 * 33];
    u32x4 *W2s = reinterpret_cast<u32x4 *>(smem_res);                       // [NT2][384]
    float(*Hs)[MT][32 * 33] = reinterpret_cast<float(*)[MT][32 * 33]>(smem_res + NT2 * 1536);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const long long M = (long long)B * H * W;
    const int cpt = (C + 31) / 32;
    const int nchunk = 9 * cpt;
    const bool relu_in = flags & kFlagReluIn, relu_out = flags & kFlagReluOut;

    for (int i = tid; i < NT2 * 128 * TERMS; i += 256) W2s[i] = w2img[i];

    const long long wbase = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT);
    const long long img_px = (long long)H * W;
    const long long b_first = ((long long)blockIdx.x * (128 * MT)) / img_px;
    const auto in_rs = act_rsrc(in + (size_t)b_first * H * W * C, (unsigned long long)(B - b_first) * H * W * C * 4ull);
    unsigned pbase[MT], tapmask[MT];
    long long myimg[MT];
    float xsc[MT], d1[MT];                              // H2: image scale 2^kx and its inverse of this lane's pixel rows
    const float w1d = H2 ? h2_dw(hdr1)[l31] : 1.0f;     // H2: 2^-kw1[n] of this lane's hidden channel
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = wbase + mt * 32 + l31;
        const bool valid = p < M;
        const long long pc = valid ? p : 0;
        const long long b = pc / img_px;
        const int rem = (int)(pc - b * img_px);
        const int gy = rem / W, gx = rem - gy * W;
        pbase[mt] = (unsigned)((((b - b_first) * H + gy) * W + gx) * C * 4 + 64 * h);
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = gy + t / 3 - 1, ix = gx + t % 3 - 1;
            if (valid && iy >= 0 && iy < H && ix >= 0 && ix < W) m |= 1u << t;
        }
        tapmask[mt] = m;
        myimg[mt] = valid ? b : -1;
        xsc[mt] = 1.0f; d1[mt] = 1.0f;
        if (H2 && valid) {
            const float mx = __int_as_float(in_amax[b]);
            int e = 15;
            if (mx > 0.0f && mx < 3.0e38f) (void)__builtin_frexpf(mx, &e);
            int kx = 15 - e;
            kx = kx > 100 ? 100 : (kx < -100 ? -100 : kx);
            xsc[mt] = __builtin_ldexpf(1.0f, kx);
            d1[mt] = __builtin_ldexpf(1.0f, -kx);
        }
    }

    constexpr int KC = 2;
    f32x4 a[KC][MT][4];
    u32x4 bq[KC][6];                                   // [term*2 + step] for this lane's half
    f32x16 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;

    const u32x4 *w1v = w1img + h * 32 + l31;          // + chunk*384 + (term*2 + step)*64
    auto load_ab = [&](int c, f32x4(&dst)[MT][4], u32x4(&bd)[6]) {
        const int tap = c / cpt, cc = c - tap * cpt;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int tapbytes = (dy * W + dx) * C * 4;
        const unsigned soff = (unsigned)cc * 128u;
#pragma unroll
        for (int q = 0; q < 2 * TERMS; ++q) bd[q] = w1v[(size_t)c * (128 * TERMS) + q * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned vo = ((tapmask[mt] >> tap) & 1u) ? pbase[mt] + (unsigned)tapbytes : kOobOffset;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, vo + 16 * j, soff, 0));
        }
    };

#pragma unroll
    for (int k = 0; k < KC; ++k)
        if (k < nchunk) load_ab(k, a[k], bq[k]);
    for (int c0 = 0; c0 < nchunk; c0 += KC) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (c0 + k < nchunk) {
                u32x4 S1[MT][2], S2[MT][2], S3[MT][2], bw[6];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (relu_in) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[k][mt][j] = relu4(a[k][mt][j]);
                    }
                    if constexpr (H2) {
                        split8_h(a[k][mt][0], a[k][mt][1], xsc[mt], S1[mt][0], S2[mt][0]);
                        split8_h(a[k][mt][2], a[k][mt][3], xsc[mt], S1[mt][1], S2[mt][1]);
                    } else {
                        split8(a[k][mt][0], a[k][mt][1], S1[mt][0], S2[mt][0], S3[mt][0]);
                        split8(a[k][mt][2], a[k][mt][3], S1[mt][1], S2[mt][1], S3[mt][1]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2 * TERMS; ++q) bw[q] = bq[k][q];
                if (c0 + k + KC < nchunk) load_ab(c0 + k + KC, a[k], bq[k]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {         // the two pixel tiles share the weights, separate accumulators
                    if constexpr (H2)
                        prod3x2(S1[0][t], S2[0][t], S1[1][t], S2[1][t], bw[t], bw[2 + t], acc1[0], acc1[1]);
                    else
                        prod6x2(S1[0][t], S2[0][t], S3[0][t], S1[1][t], S2[1][t], S3[1][t], bw[t], bw[2 + t], bw[4 + t],
                                acc1[0], acc1[1]);
                }
            }
        }
    }
    __syncthreads();          // W2 image (copied at kernel start) is complete

    // hidden tile: relu, accumulator layout -> [pixel][hidden] in LDS (stride 33: conflict-free)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float dr = H2 ? __shfl(d1[mt], prow) * w1d : 1.0f;  // the row's 3x3 accumulator scale 2^-(kx + kw1[n])
            Hs[wave][mt][prow * 33 + l31] = fmaxf(H2 ? acc1[mt][r] * dr : acc1[mt][r], 0.0f);
        }
    lds_order_wave();
    u32x4 H1[MT][2], Hb[MT][2], H3[MT][2];
    float d2[MT];                                       // H2: 1x1 accumulator scale of this lane's pixel rows
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float a2[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a2[q] = Hs[wave][mt][l31 * 33 + 16 * h + q];
        d2[mt] = 1.0f;
        if constexpr (H2) {
            float m = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaxf(m, a2[q]);
            m = fmaxf(m, __shfl_xor(m, 32));             // the pixel's other sixteen hidden values
            int e = 15;
            if (m > 0.0f && m < 3.0e38f) (void)__builtin_frexpf(m, &e);
            int kh = 15 - e;
            kh = kh > 100 ? 100 : (kh < -100 ? -100 : kh);
            const float hsc = __builtin_ldexpf(1.0f, kh);
            d2[mt] = __builtin_ldexpf(1.0f, -kh);
            split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, hsc, H1[mt][0], Hb[mt][0]);
            split8_h(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, hsc, H1[mt][1], Hb[mt][1]);
        } else {
            split8(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, H1[mt][0], Hb[mt][0], H3[mt][0]);
            split8(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, H1[mt][1], Hb[mt][1],
                   H3[mt][1]);
        }
    }
    // maxima for the next layer: one image per wave in the common case, per pixel row otherwise
    const long long img0 = __shfl(myimg[0], 0);
    const bool one_img = out_amax && img0 >= 0 && __builtin_amdgcn_ballot_w64(myimg[0] != img0 || myimg[1] != img0) == 0;
    float omax = 0.0f;

    // second GEMM, one n-tile at a time (two pixel tiles = two interleaved accumulators)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        f32x16 acc2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 *bp = W2s + nt * (128 * TERMS) + (t * 2 + h) * 32 + l31;
            if constexpr (H2) {
                prod3x2(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], bp[0], bp[128], acc2[0], acc2[1]);
            } else {
                const u32x4 w1 = bp[0], w2 = bp[128], w3 = bp[256];
                prod6x2(H1[0][t], Hb[0][t], H3[0][t], H1[1][t], Hb[1][t], H3[1][t], w1, w2, w3, acc2[0], acc2[1]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int src = (r & 3) + 8 * (r >> 2) + 4 * h;
                const long long prow = wbase + mt * 32 + src;
                const int n = nt * 32 + l31;
                const float dr = H2 ? __shfl(d2[mt], src) * h2_dw(hdr2)[n] : 1.0f;     // 2^-(kh + kw2[n])
                float rmax = 0.0f;
                if (prow < M && n < C) {
                    float u = in[prow * C + n];
                    if (relu_in) u = fmaxf(u, 0.0f);
                    float v = u + (H2 ? acc2[mt][r] * dr : acc2[mt][r]);
                    if (relu_out) v = fmaxf(v, 0.0f);
                    rmax = __builtin_fabsf(v);
                    out[prow * C + n] = v;
                }
                if (out_amax && !one_img) {
                    const long long rimg = __shfl(myimg[mt], src);
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o));
                    if (l31 == 0 && rimg >= 0) atomicMax(out_amax + rimg, __float_as_int(rmax));
                } else {
                    omax = fmaxf(omax, rmax);
                }
            }
    }
    if (one_img) publish_amax(out_amax, img0, omax, lane);
}

// ---------------------------------------------------------------------------
// Fused residual layer for 8x8 feature maps (the reference's 32x32 images: both residual stacks run at 8x8).
// One wave owns one whole image, so every 3x3 tap of every pixel lives inside the wave's own tile:
//   for each 16-channel slice of the input (one MFMA k-step of the packed weight image; slice outer, tap inner):
//       load the slice of the image once, apply the in-place ReLU, split it ONCE into its three bf16
//       terms and park them in a wave-private LDS tile (64 pixels + one all-zero "padding" pixel);
//       the nine taps then read their A operands from that tile with ds_read_b128 at shifted pixel indices.
// Compared with res_layer_bf3_kernel (A re-loaded from L2 and re-split for each of the 9 taps) this cuts
// the L1/TA traffic and the split VALU work of the 3x3 GEMM 9x; no workgroup barrier in the reduction.
// The hidden tile and the 1x1 GEMM / skip / ReLU epilogue are the same as in res_layer_bf3_kernel.
// H2: two-term fp16 products (split8_h): per-image scale for x, a second one for the hidden tile, per-layer weight scales
// in the headers hdr1 / hdr2 of the two weight images.
template <int NT2, bool H2 = false>
__global__ __launch_bounds__(256, H2 ? 4 : 3) void res_tile8_bf3_kernel(const float *__restrict__ in,
                                                               const u32x4 *__restrict__ w1img,
                                                               const u32x4 *__restrict__ w2img,
                                                               float *__restrict__ out, int B, int C, int flags,
                                                               const int *__restrict__ hdr1, const int *__restrict__ hdr2,
                                                               const int *__restrict__ in_amax, int *__restrict__ out_amax,
                                                               float *__restrict__ hid_out) {
    constexpr int TERMS = H2 ? 2 : 3;
    // u32x4 per wave tile: [term][half][pixel + zero], at least the 32 x 33 floats of the hidden tile that aliases it.
    // H2: 4.1 KiB per wave + 16 KiB of W2 = 33 KiB per workgroup -> four workgroups (16 waves) per CU, and the 1024
    // workgroups of a B = 4096 layer are all resident at once (no second, part-filled round)
    constexpr int MT = 2, PX = 64, TILE4 = H2 ? 264 : 3 * (PX + 1) * 2;
    constexpr int HP = PX + 1;                                     // (consecutive lanes = consecutive 16 B: no bank conflicts)
    static_assert(TILE4 * 16 >= 32 * 33 * 4, "the hidden tile aliases the operand tile");
    __shared__ u32x4 W2s[NT2 * 128 * TERMS];
    __shared__ u32x4 As_all[4 * TILE4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    const bool relu_in = flags & kFlagReluIn, relu_out = flags & kFlagReluOut;
    const int cpt = C >> 5, nslice = C >> 4;

    for (int i = tid; i < NT2 * 128 * TERMS; i += 256) W2s[i] = w2img[i];
    if (lane < 2 * TERMS) As[(lane >> 1) * (HP * 2) + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};        // padding pixel

    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;
    const float *src = in + (size_t)(img_ok ? img : 0) * PX * C + (size_t)lane * C;   // this lane's pixel row

    // operand pixel index per (tap, m-tile): the shifted pixel, or the zero pixel outside the image
    int spx[MT];
    unsigned tapok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, x = spx[mt] & 7;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << t;
        }
        tapok[mt] = m;
    }

    f32x16 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;

    // slice sl = k-step (sl & 1) of 32-channel chunk (sl >> 1) of the packed weight image: channels
    // 32*chunk + 8*step + [0,8) for the h = 0 operand half and 32*chunk + 16 + 8*step + [0,8) for h = 1
    auto load_raw = [&](int sl, f32x4(&r)[4]) {
        const float *q = src + 32 * (sl >> 1) + 8 * (sl & 1);
        r[0] = *reinterpret_cast<const f32x4 *>(q);
        r[1] = *reinterpret_cast<const f32x4 *>(q + 4);
        r[2] = *reinterpret_cast<const f32x4 *>(q + 16);
        r[3] = *reinterpret_cast<const f32x4 *>(q + 20);
    };
    // weights of (tap, slice): 16 k x 32 hidden x 3 terms, this lane's 8 k of each term.  The image is the
    // conv_pack_bf3 layout: chunk = tap*cpt + slice/2, k-step = slice & 1
    const u32x4 *w1v = w1img + h * 32 + l31;
    auto load_w = [&](int tap, int sl, u32x4(&bw)[3]) {
        const u32x4 *p = w1v + (size_t)(tap * cpt + (sl >> 1)) * (128 * TERMS) + (sl & 1) * 64;
        bw[0] = p[0]; bw[1] = p[128];
        if constexpr (!H2) bw[2] = p[256];
    };

    f32x4 raw[4];
    u32x4 bw[2][3];
    float xscale = 1.0f, d1 = 1.0f;                  // H2: image scale 2^kx, GEMM1 accumulator scale 2^-(kx + kw1)
    if constexpr (H2) {
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;        // the producer's maximum of this image, if any
        if (given >= 0) m = __int_as_float(given);
        else for (int sl = 0; sl < nslice; ++sl) {
            load_raw(sl, raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = raw[j];
                if (relu_in) v = relu4(v);
                m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
            }
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        xscale = __builtin_ldexpf(1.0f, kx);
        d1 = __builtin_ldexpf(1.0f, -kx) * h2_dw(hdr1)[l31];          // this lane's hidden channel: 2^-(kx + kw1[n])
    }
    load_raw(0, raw);
    load_w(0, 0, bw[0]);
    // one 16-channel slice; PAR = slice parity (nine taps per slice flip which weight register set is "current")
    auto slice = [&](int sl, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        // ---- stage this slice: ReLU, split once, park the three terms (the tile is wave-private) ----
        {
            if (relu_in) {
#pragma unroll
                for (int j = 0; j < 4; ++j) raw[j] = relu4(raw[j]);
            }
            u32x4 t1a, t2a, t3a, t1b, t2b, t3b;
            if constexpr (H2) {
                split8_h(raw[0], raw[1], xscale, t1a, t2a);
                split8_h(raw[2], raw[3], xscale, t1b, t2b);
            } else {
                split8(raw[0], raw[1], t1a, t2a, t3a);
                split8(raw[2], raw[3], t1b, t2b, t3b);
            }
            if (sl + 1 < nslice) load_raw(sl + 1, raw);
            __builtin_amdgcn_wave_barrier();                  // all taps of the previous slice have been read
            u32x4 *dst = As + lane;
            dst[0] = t1a; dst[HP] = t1b;
            dst[HP * 2] = t2a; dst[HP * 3] = t2b;
            if constexpr (!H2) { dst[HP * 4] = t3a; dst[HP * 5] = t3b; }
            lds_order_wave();
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = (tap + par) & 1;
            if (tap + 1 < 9) load_w(tap + 1, sl, bw[cur ^ 1]);
            else if (sl + 1 < nslice) load_w(0, sl + 1, bw[cur ^ 1]);
            const int shift = (tap / 3 - 1) * 8 + (tap % 3 - 1);
            u32x4 S[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = ((tapok[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                const u32x4 *ap = As + h * HP + p;
                S[mt][0] = ap[0]; S[mt][1] = ap[HP * 2];
                if constexpr (!H2) S[mt][2] = ap[HP * 4];
            }
            if constexpr (H2)
                prod3x2(S[0][0], S[0][1], S[1][0], S[1][1], bw[cur][0], bw[cur][1], acc1[0], acc1[1]);
            else
                prod6x2(S[0][0], S[0][1], S[0][2], S[1][0], S[1][1], S[1][2], bw[cur][0], bw[cur][1], bw[cur][2], acc1[0],
                        acc1[1]);
        }
    };
    for (int sl = 0; sl < nslice; sl += 2) {                  // C % 32 == 0: an even number of slices
        slice(sl, std::integral_constant<int, 0>{});
        slice(sl + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();          // W2 image (copied at kernel start) is complete; operand tile no longer read

    // hidden tile: relu, accumulator layout -> [pixel][hidden] in LDS (stride 33), one m-tile at a time in the
    // (now free) operand tile
    float *Hs = reinterpret_cast<float *>(As);
    u32x4 H1[MT][2], Hb[MT][2], H3[MT][2];
    float hscale = 1.0f, d2 = 1.0f;                  // H2: hidden-tile scale 2^kh, GEMM2 accumulator scale 2^-(kh + kw2)
    if constexpr (H2) {
        float m = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) SCALE_BIAS_RELU2(acc1[mt][r], acc1[mt][r + 1], d1, 0.0f, 0.0f, m);
        const int kh = wave_scale_exp(m);
        hscale = __builtin_ldexpf(1.0f, kh);
        d2 = __builtin_ldexpf(1.0f, -kh);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
            Hs[prow * 33 + l31] = fmaxf(acc1[mt][r], 0.0f);
        }
        lds_order_wave();
        float a2[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a2[q] = Hs[l31 * 33 + 16 * h + q];
        if (hid_out && img_ok) {                            // training: the hidden activation (B, 8, 8, 32) for backward
            f32x4 *hp = reinterpret_cast<f32x4 *>(hid_out + ((size_t)img * PX + mt * 32 + l31) * 32 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) hp[q] = f32x4{a2[4 * q], a2[4 * q + 1], a2[4 * q + 2], a2[4 * q + 3]};
        }
        if constexpr (H2) {
            split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, hscale, H1[mt][0], Hb[mt][0]);
            split8_h(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, hscale, H1[mt][1], Hb[mt][1]);
        } else {
            split8(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, H1[mt][0], Hb[mt][0], H3[mt][0]);
            split8(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, H1[mt][1], Hb[mt][1],
                   H3[mt][1]);
        }
        __builtin_amdgcn_wave_barrier();
    }

    const long long wbase = img * PX;
    float omax = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        f32x16 acc2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 *bp = W2s + nt * (128 * TERMS) + (t * 2 + h) * 32 + l31;
            if constexpr (H2) {
                prod3x2(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], bp[0], bp[128], acc2[0], acc2[1]);
            } else {
                const u32x4 w1 = bp[0], w2 = bp[128], w3 = bp[256];
                prod6x2(H1[0][t], Hb[0][t], H3[0][t], H1[1][t], Hb[1][t], H3[1][t], w1, w2, w3, acc2[0], acc2[1]);
            }
        }
        if (img_ok) {
            // skip connection, activation and store in the staged layout: 16-byte loads and stores
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v[16];
                const float d2n = H2 ? d2 * h2_dw(hdr2)[nt * 32 + l31] : 1.0f;       // 2^-(kh + kw2[n]) of this lane's channel
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = H2 ? acc2[mt][r] * d2n : acc2[mt][r];
                // the four skip values of this lane are requested before the tile goes through LDS
                f32x4 u[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    u[k] = *reinterpret_cast<const f32x4 *>(in + (wbase + mt * 32 + (lane >> 3) + 8 * k) * C + nt * 32 +
                                                            4 * (lane & 7));
                tile_epilogue(Hs, v, lane, nt * 32, [&](int p, int n, f32x4 a4, int k) {
                    f32x4 u0 = u[k];
                    if (relu_in) u0 = relu4(u0);
                    f32x4 y0 = u0 + a4;
                    if (relu_out) y0 = relu4(y0);
                    omax = fmaxf(omax, fmaxf(fmaxf(__builtin_fabsf(y0.x), __builtin_fabsf(y0.y)), fmaxf(__builtin_fabsf(y0.z), __builtin_fabsf(y0.w))));
                    *reinterpret_cast<f32x4 *>(out + (wbase + mt * 32 + p) * C + n) = y0;
                });
            }
        }
    }
    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
}

// ---------------------------------------------------------------------------
// res_tile8_bf3_kernel<., true> on maps LARGER than 8x8 (round 3; BASELINE configs 4 / 5): one wave owns one 8x8 tile of
// one image's map plus a one-pixel halo (a 10x10 patch).  Per 16-channel slice the patch is loaded once (pixels outside the
// image read as zero through the buffer descriptor), ReLU'd, split once into its two fp16 terms and parked in the wave's
// LDS tile; the nine taps read their operands at shifted patch indices, no masks.  Hidden tile, 1x1 GEMM, skip, ReLU and
// the staged stores are res_tile8_bf3_kernel's (the skip re-reads the tile's 64 centre pixels).  Two-term fp16 products;
// the scale of x is the image's maximum from the producing layer (in_amax), or the patch's own where none is given.
template <int NT2>
__global__ __launch_bounds__(256, 3) void res_halo8_h2_kernel(const float *__restrict__ in, const u32x4 *__restrict__ w1img,
                                                             const u32x4 *__restrict__ w2img, float *__restrict__ out, int B,
                                                             int H, int W, int C, int flags, const int *__restrict__ hdr1,
                                                             const int *__restrict__ hdr2, const int *__restrict__ in_amax,
                                                             int *__restrict__ out_amax) {
    constexpr int MT = 2, PW = 10, PP = PW * PW, HP = PP + 1, TILE4 = 4 * HP;      // [term 2][half 2][patch pixel]
    static_assert(TILE4 * 16 >= 32 * 33 * 4, "the hidden tile aliases the operand tile");
    __shared__ u32x4 W2s[NT2 * 256];
    __shared__ u32x4 As_all[4 * TILE4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    u32x4 *As = As_all + wave * TILE4;
    const bool relu_in = flags & kFlagReluIn, relu_out = flags & kFlagReluOut;
    const int cpt = C >> 5, nslice = C >> 4;

    for (int i = tid; i < NT2 * 256; i += 256) W2s[i] = w2img[i];

    const int tx_n = W >> 3, ty_n = H >> 3;
    // (wave-uniform: a lane-derived tile index costs a waterfall loop around every buffer load)
    const long long tile_id = (long long)blockIdx.x * 4 + wave_u, ntile_all = (long long)B * ty_n * tx_n;
    const bool img_ok = tile_id < ntile_all;
    const long long tq = img_ok ? tile_id : 0;
    const long long img = tq / (ty_n * tx_n);
    const int trem = (int)(tq - img * (ty_n * tx_n));
    const int y0 = (trem / tx_n) * 8, x0 = (trem % tx_n) * 8;
    const float *img_base = in + (size_t)img * H * W * C;
    const auto rs = act_rsrc(img_base, img_ok ? (unsigned long long)H * W * C * 4ull : 0ull);
    unsigned poff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = 64 * k + lane;
        const int iy = y0 - 1 + q / PW, ix = x0 - 1 + q % PW;
        poff[k] = (q < PP && iy >= 0 && iy < H && ix >= 0 && ix < W) ? (unsigned)((iy * W + ix) * C) * 4u : kOobOffset;
    }
    int spx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = 32 * mt + l31;
        spx[mt] = ((p >> 3) + 1) * PW + (p & 7) + 1;
    }

    f32x16 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;

    // slice sl = k-step (sl & 1) of 32-channel chunk (sl >> 1): channels 32 chunk + 8 step + [0, 8) (h = 0), + 16 (h = 1)
    f32x4 raw[2][4];
    auto load_raw = [&](int sl) {
        const unsigned co = (unsigned)(32 * (sl >> 1) + 8 * (sl & 1)) * 4u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                      // (the slice's offset rides in the scalar offset: no vector instruction)
            raw[k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k], co, 0));
            raw[k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k] + 16u, co, 0));
            raw[k][2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k] + 64u, co, 0));
            raw[k][3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k] + 80u, co, 0));
        }
    };
    const u32x4 *w1v = w1img + h * 32 + l31;
    auto load_w = [&](int tap, int sl, u32x4(&bw)[2]) {
        const u32x4 *p = w1v + (size_t)(tap * cpt + (sl >> 1)) * 256 + (sl & 1) * 64;
        bw[0] = p[0]; bw[1] = p[128];
    };
    u32x4 bw[3][2];                                    // three taps' weights: two in flight behind the one in use
    float xscale, d1;
    {
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;
        if (given >= 0) m = __int_as_float(given);
        else for (int sl = 0; sl < nslice; ++sl) {
            load_raw(sl);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = raw[k][j];
                    if (relu_in) v = relu4(v);
                    m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
                }
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        xscale = __builtin_ldexpf(1.0f, kx);
        d1 = __builtin_ldexpf(1.0f, -kx) * h2_dw(hdr1)[l31];          // this lane's hidden channel: 2^-(kx + kw1[n])
    }
    load_raw(0);
    load_w(0, 0, bw[0]);
    load_w(1, 0, bw[1]);
    auto slice = [&](int sl) {
        {
            u32x4 t1a[2], t2a[2], t1b[2], t2b[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (relu_in) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) raw[k][j] = relu4(raw[k][j]);
                }
                split8_h(raw[k][0], raw[k][1], xscale, t1a[k], t2a[k]);
                split8_h(raw[k][2], raw[k][3], xscale, t1b[k], t2b[k]);
            }
            if (sl + 1 < nslice) load_raw(sl + 1);
            __builtin_amdgcn_wave_barrier();                  // all taps of the previous slice have been read
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && lane >= PP - 64) break;
                u32x4 *dst = As + 64 * k + lane;
                dst[0] = t1a[k]; dst[HP] = t1b[k];
                dst[HP * 2] = t2a[k]; dst[HP * 3] = t2b[k];
            }
            lds_order_wave();
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {            // (nine taps: the ring position of tap 0 is the same for every slice)
            const int cur = tap % 3, nxt = (tap + 2) % 3;
            if (tap + 2 < 9) load_w(tap + 2, sl, bw[nxt]);
            else if (sl + 1 < nslice) load_w(tap + 2 - 9, sl + 1, bw[nxt]);
            const int shift = (tap / 3 - 1) * PW + (tap % 3 - 1);
            u32x4 S[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const u32x4 *ap = As + h * HP + spx[mt] + shift;
                S[mt][0] = ap[0]; S[mt][1] = ap[HP * 2];
            }
            prod3x2(S[0][0], S[0][1], S[1][0], S[1][1], bw[cur][0], bw[cur][1], acc1[0], acc1[1]);
        }
    };
#pragma unroll 1
    for (int sl = 0; sl < nslice; ++sl) slice(sl);
    __syncthreads();          // W2 image (copied at kernel start) is complete; operand tile no longer read

    float *Hs = reinterpret_cast<float *>(As);
    u32x4 H1[MT][2], Hb[MT][2];
    float hscale, d2;
    {
        float m = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc1[mt][r] = vmax(acc1[mt][r] * d1, 0.0f);
                m = vmax(m, acc1[mt][r]);
            }
        const int kh = wave_scale_exp(m);
        hscale = __builtin_ldexpf(1.0f, kh);
        d2 = __builtin_ldexpf(1.0f, -kh);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
            Hs[prow * 33 + l31] = acc1[mt][r];
        }
        lds_order_wave();
        float a2[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a2[q] = Hs[l31 * 33 + 16 * h + q];
        split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, hscale, H1[mt][0], Hb[mt][0]);
        split8_h(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, hscale, H1[mt][1], Hb[mt][1]);
        __builtin_amdgcn_wave_barrier();
    }

    float omax = 0.0f;
    if (img_ok) {
        // Per 32-channel tile: the 1x1 GEMM, then per pixel tile the skip and the output rows -- a scalar row base + this
        // lane's constant offset, packed multiplies, single-instruction max: whatever a wave issues here waits behind the
        // other waves' MFMAs.  The skip values of the NEXT (channel tile, pixel tile) step are requested a step ahead.  One
        // straight-line copy per ReLU flag pair (branches inside would cut it into blocks with a full wait at every join).
        const unsigned olane = (unsigned)((lane >> 3) * C + 4 * (lane & 7)) * 4u;
        auto urow0 = [&](int step) { return ((size_t)(y0 + 4 * (step & 1)) * W + x0) * C + (step >> 1) * 32; };     // wave-uniform
        auto skip_load = [&](int step, f32x4(&u)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                u[k] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(img_base + urow0(step) + (size_t)k * W * C) + olane);
        };
        auto finish = [&](auto RI, auto RO) {
            constexpr bool ri = decltype(RI)::value, ro = decltype(RO)::value;
            f32x4 u[2][4];
            skip_load(0, u[0]);
#pragma unroll 1
            for (int nt = 0; nt < NT2; ++nt) {
                f32x16 acc2[MT];
#pragma unroll
                for (int m2 = 0; m2 < MT; ++m2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m2][r] = 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const u32x4 *bp = W2s + nt * 256 + (t * 2 + h) * 32 + l31;
                    prod3x2(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], bp[0], bp[128], acc2[0], acc2[1]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int step = 2 * nt + mt;
                    if (step + 1 < 2 * NT2) skip_load(step + 1, u[mt ^ 1]);
                    const float d2n = d2 * h2_dw(hdr2)[nt * 32 + l31];                // 2^-(kh + kw2[n]) of this lane's channel
                    const f32x2v dd = {d2n, d2n};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2v v = f32x2v{acc2[mt][r], acc2[mt][r + 1]} * dd;
                        Hs[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = v.x;
                        Hs[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * h) * 32 + l31] = v.y;
                    }
                    lds_order_wave();
                    float *orow0 = out + (size_t)img * H * W * C + urow0(step);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 q = *reinterpret_cast<const f32x4 *>(Hs + k * 256 + lane * 4);
                        f32x4 uu = u[mt][k];
                        if (ri) uu = relu4(uu);
                        f32x4 yv = uu + q;
                        if (ro) yv = relu4(yv);
                        vmax3_abs(omax, yv.x, yv.y);
                        vmax3_abs(omax, yv.z, yv.w);
                        *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(orow0 + (size_t)k * W * C) + olane) = yv;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
        if (relu_in && relu_out) finish(std::true_type{}, std::true_type{});
        else if (relu_in) finish(std::true_type{}, std::false_type{});
        else if (relu_out) finish(std::false_type{}, std::true_type{});
        else finish(std::false_type{}, std::false_type{});
    }
    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
}

// ---------------------------------------------------------------------------
// TWO residual layers of a stack in one kernel (models/residual.py:47-51: the layers of a stack share their weights), 8x8
// maps, two-term fp16 products.  One wave owns one image; the first layer's output never leaves the chip:
//   layer 1:  as res_tile8_bf3_kernel<., true>, but the skip relu(x) is added in the ACCUMULATOR layout (dword loads, 128
//             contiguous bytes per pixel row) and y1' = relu(relu(x) + W2 h1) -- the ReLU is the second layer's in-place
//             one -- stays in 128 registers per lane, Y[m-tile][n-tile][16];
//   layer 2:  per 32-channel chunk Y goes accumulator layout -> [pixel][channel] through the wave's LDS tile (the hidden
//             tile's path), is split with the image's scale (maximum taken from the registers) and parked as the 3x3 GEMM's
//             operands; the skip of the second 1x1 GEMM comes straight from Y.
// HBM-side traffic per pair of layers: x read twice (reduction + skip), y2 written once -- 3 maps instead of 6.
// NT3 > 0: a 1x1 conv (C -> 32 NT3 channels, + bias; the encoder's pre-quantisation conv, models/vqvae.py:33) consumes the
// pair's output straight from the registers: y2 is not stored at all, out3 receives the conv's result.
template <int NT2, int NT3 = 0>
__global__ __launch_bounds__(256, 2) void res_pair8_h2_kernel(const float *__restrict__ in, const u32x4 *__restrict__ w1img,
                                                              const u32x4 *__restrict__ w2img, float *__restrict__ out,
                                                              int B, int C, int flags, const int *__restrict__ hdr1,
                                                              const int *__restrict__ hdr2, const int *__restrict__ in_amax,
                                                              int *__restrict__ out_amax, const u32x4 *__restrict__ w3img,
                                                              const int *__restrict__ hdr3, const float *__restrict__ bias3,
                                                              float *__restrict__ out3) {
    constexpr int MT = 2, PX = 64, TILE4 = 264, HP = PX + 1;
    __shared__ u32x4 W2s[NT2 * 256];
    __shared__ u32x4 As_all[4 * TILE4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    float *Hs = reinterpret_cast<float *>(As);
    const bool relu_in = flags & kFlagReluIn, relu_out = flags & kFlagReluOut;
    const int cpt = C >> 5, nslice = C >> 4;

    for (int i = tid; i < NT2 * 256; i += 256) W2s[i] = w2img[i];
    if (lane < 4) As[(lane >> 1) * (HP * 2) + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};        // padding pixel

    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;
    const float *src = in + (size_t)(img_ok ? img : 0) * PX * C + (size_t)lane * C;   // this lane's pixel row
    const float w1d = h2_dw(hdr1)[l31];                  // 2^-kw1[n] of this lane's hidden channel

    int spx[MT];
    unsigned tapok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, x = spx[mt] & 7;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << t;
        }
        tapok[mt] = m;
    }
    const u32x4 *w1v = w1img + h * 32 + l31;
    auto load_w = [&](int tap, int sl, u32x4(&bw)[2]) {
        const u32x4 *p = w1v + (size_t)(tap * cpt + (sl >> 1)) * 256 + (sl & 1) * 64;
        bw[0] = p[0]; bw[1] = p[128];
    };
    // nine taps of one parked 16-channel slice into acc1; PAR = which weight register set is current at tap 0
    u32x4 bw[2][2];
    f32x16 acc1[MT];
    auto taps = [&](int sl, auto PAR) {
        constexpr int par = decltype(PAR)::value;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int cur = (tap + par) & 1;
            if (tap + 1 < 9) load_w(tap + 1, sl, bw[cur ^ 1]);
            else if (sl + 1 < nslice) load_w(0, sl + 1, bw[cur ^ 1]);
            const int shift = (tap / 3 - 1) * 8 + (tap % 3 - 1);
            u32x4 S[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = ((tapok[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                const u32x4 *ap = As + h * HP + p;
                S[mt][0] = ap[0]; S[mt][1] = ap[HP * 2];
            }
            prod3x2(S[0][0], S[0][1], S[1][0], S[1][1], bw[cur][0], bw[cur][1], acc1[0], acc1[1]);
        }
    };
    // hidden tile: relu, scale, accumulator layout -> A operands of the 1x1 GEMM; returns that GEMM's accumulator scale
    u32x4 H1[MT][2], Hb[MT][2];
    auto hidden = [&](float d1) -> float {
        float m = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) SCALE_BIAS_RELU2(acc1[mt][r], acc1[mt][r + 1], d1, 0.0f, 0.0f, m);
        const int kh = wave_scale_exp(m);
        const float hscale = __builtin_ldexpf(1.0f, kh);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Hs[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc1[mt][r];
            lds_order_wave();
            float a2[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a2[q] = Hs[l31 * 33 + 16 * h + q];
            split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, hscale, H1[mt][0], Hb[mt][0]);
            split8_h(f32x4{a2[8], a2[9], a2[10], a2[11]}, f32x4{a2[12], a2[13], a2[14], a2[15]}, hscale, H1[mt][1], Hb[mt][1]);
            __builtin_amdgcn_wave_barrier();
        }
        return __builtin_ldexpf(1.0f, -kh);               // (x the 1x1 rows' 2^-kw2[n] at the use)
    };
    auto gemm2 = [&](int nt, f32x16(&acc2)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4 *bp = W2s + nt * 256 + (t * 2 + h) * 32 + l31;
            prod3x2(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], bp[0], bp[128], acc2[0], acc2[1]);
        }
    };

    // =========================================== layer 1 ===========================================
    auto load_raw = [&](int sl, f32x4(&r)[4]) {
        const float *q = src + 32 * (sl >> 1) + 8 * (sl & 1);
        r[0] = *reinterpret_cast<const f32x4 *>(q);
        r[1] = *reinterpret_cast<const f32x4 *>(q + 4);
        r[2] = *reinterpret_cast<const f32x4 *>(q + 16);
        r[3] = *reinterpret_cast<const f32x4 *>(q + 20);
    };
    f32x4 raw[4];
    float xscale, d1;
    {
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;
        if (given >= 0) m = __int_as_float(given);
        else for (int sl = 0; sl < nslice; ++sl) {
            load_raw(sl, raw);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = raw[j];
                if (relu_in) v = relu4(v);
                m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
            }
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        xscale = __builtin_ldexpf(1.0f, kx);
        d1 = __builtin_ldexpf(1.0f, -kx) * w1d;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;
    load_raw(0, raw);
    load_w(0, 0, bw[0]);
    auto slice1 = [&](int sl, auto PAR) {
        if (relu_in) {
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[j] = relu4(raw[j]);
        }
        u32x4 t1a, t2a, t1b, t2b;
        split8_h(raw[0], raw[1], xscale, t1a, t2a);
        split8_h(raw[2], raw[3], xscale, t1b, t2b);
        if (sl + 1 < nslice) load_raw(sl + 1, raw);
        __builtin_amdgcn_wave_barrier();                  // all taps of the previous slice have been read
        u32x4 *dst = As + lane;
        dst[0] = t1a; dst[HP] = t1b;
        dst[HP * 2] = t2a; dst[HP * 3] = t2b;
        lds_order_wave();
        taps(sl, PAR);
    };
    for (int sl = 0; sl < nslice; sl += 2) {
        slice1(sl, std::integral_constant<int, 0>{});
        slice1(sl + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();          // W2 image (copied at kernel start) is complete
    float Y[MT][NT2][16];
    {
        const float d2 = hidden(d1);
        const float *xb = in + (size_t)(img_ok ? img : 0) * PX * C;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            f32x16 acc2[MT];
            // the skip values in accumulator layout: element r of m-tile mt = pixel 32 mt + (r&3) + 8 (r>>2) + 4 h, channel
            // 32 nt + l31 (the first m-tile's are requested before the GEMM, the second's behind it: 16 live registers)
            float u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = xb[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * C + nt * 32 + l31];
            gemm2(nt, acc2);
            const float d2n = d2 * h2_dw(hdr2)[nt * 32 + l31];            // 2^-(kh + kw2[n]) of this lane's channel
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float u0 = relu_in ? fmaxf(u[r], 0.0f) : u[r];
                    Y[mt][nt][r] = fmaxf(u0 + acc2[mt][r] * d2n, 0.0f);      // + the second layer's in-place ReLU
                }
                if (mt + 1 < MT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[r] = xb[(size_t)(32 * (mt + 1) + (r & 3) + 8 * (r >> 2) + 4 * h) * C + nt * 32 + l31];
                }
            }
        }
    }

    // =========================================== layer 2 ===========================================
    {
        float m = 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, Y[mt][nt][r]);
        const int kx = wave_scale_exp(m);
        xscale = __builtin_ldexpf(1.0f, kx);
        d1 = __builtin_ldexpf(1.0f, -kx) * w1d;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;
    load_w(0, 0, bw[0]);
#pragma unroll
    for (int c = 0; c < NT2; ++c) {
        // chunk c of Y: accumulator layout -> lane (pixel l31 of tile mt, half h) holds channels 32 c + 16 h + [0, 16); the
        // transposition runs once per 16-channel slice (eight of the sixteen values each time: LDS traffic is cheaper
        // than sixteen more live registers next to Y)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 t1[MT], t2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                __builtin_amdgcn_wave_barrier();          // the previous slice's taps / the previous tile's reads are behind us
#pragma unroll
                for (int r = 0; r < 16; ++r) Hs[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = Y[mt][c][r];
                lds_order_wave();
                float a2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a2[q] = Hs[l31 * 33 + 16 * h + 8 * s2 + q];
                split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, xscale, t1[mt], t2[mt]);
            }
            __builtin_amdgcn_wave_barrier();              // scratch reads are done
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                As[(0 * 2 + h) * HP + 32 * mt + l31] = t1[mt];
                As[(1 * 2 + h) * HP + 32 * mt + l31] = t2[mt];
            }
            if (lane < 4) As[lane * HP + PX] = u32x4{0, 0, 0, 0};      // the padding pixel was under the transposition scratch
            lds_order_wave();
            if (s2 == 0) taps(2 * c, std::integral_constant<int, 0>{});
            else taps(2 * c + 1, std::integral_constant<int, 1>{});
        }
    }
    {
        const float d2 = hidden(d1);
        const long long wbase = img * PX;
        float omax = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            f32x16 acc2[MT];
            gemm2(nt, acc2);
            const float d2n = d2 * h2_dw(hdr2)[nt * 32 + l31];
            if (img_ok) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        v[r] = Y[mt][nt][r] + acc2[mt][r] * d2n;
                        if (relu_out) v[r] = fmaxf(v[r], 0.0f);
                        omax = fmaxf(omax, __builtin_fabsf(v[r]));
                    }
                    if constexpr (NT3 > 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[mt][nt][r] = v[r];       // stays on chip for the 1x1 conv below
                    } else {
                        tile_epilogue(Hs, v, lane, nt * 32, [&](int p, int n, f32x4 a4, int) {
                            *reinterpret_cast<f32x4 *>(out + (wbase + mt * 32 + p) * C + n) = a4;
                        });
                    }
                }
            }
        }
        if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);

        if constexpr (NT3 > 0) {
            // ================================ 1x1 conv on y2 (same operand order as conv_tile8_bf3_kernel) ================
            const int kx3 = wave_scale_exp(img_ok ? omax : 0.0f);
            const float xs3 = __builtin_ldexpf(1.0f, kx3), d3 = __builtin_ldexpf(1.0f, -kx3);
            f32x16 acc3[MT][NT3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc3[mt][n3][r] = 0.0f;
            const u32x4 *w3v = w3img + h * 32 + l31;
#pragma unroll
            for (int c = 0; c < NT2; ++c) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 t1[MT], t2[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r = 0; r < 16; ++r) Hs[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = Y[mt][c][r];
                        lds_order_wave();
                        float a2[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) a2[q] = Hs[l31 * 33 + 16 * h + 8 * s2 + q];
                        split8_h(f32x4{a2[0], a2[1], a2[2], a2[3]}, f32x4{a2[4], a2[5], a2[6], a2[7]}, xs3, t1[mt], t2[mt]);
                    }
                    // lane (pixel l31 of tile mt, half h) holds the A operands of ITS pixel row: exactly the MFMA A layout
#pragma unroll
                    for (int n3 = 0; n3 < NT3; ++n3) {
                        const u32x4 *bp = w3v + (size_t)(c * NT3 + n3) * 256 + s2 * 64;
                        prod3x2(t1[0], t2[0], t1[1], t2[1], bp[0], bp[128], acc3[0][n3], acc3[1][n3]);
                    }
                }
            }
            if (img_ok) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int n3 = 0; n3 < NT3; ++n3) {
                        const float bv = bias3 ? bias3[n3 * 32 + l31] : 0.0f, d3n = d3 * h2_dw(hdr3)[n3 * 32 + l31];
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = acc3[mt][n3][r] * d3n + bv;
                        __builtin_amdgcn_wave_barrier();
                        tile_epilogue(Hs, v, lane, n3 * 32, [&](int p, int n, f32x4 a4, int) {
                            *reinterpret_cast<f32x4 *>(out3 + (wbase + mt * 32 + p) * (32 * NT3) + n) = a4;
                        });
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// A 3x3 conv / 3x3 conv-transpose (stride 1) IN FRONT of a residual pair, all in one kernel (8x8 maps, two-term fp16
// products): models/encoder.py:35-38 (conv 3x3 + ReLU -> ResidualStack) and models/decoder.py:28-30 (conv-transpose 3x3 ->
// ResidualStack).  One wave owns one image.  The front conv accumulates straight into the registers that hold the
// residual layers' map (Y[m-tile][n-tile], accumulator layout: 128 channels x 64 pixels = 128 registers per lane); BOTH
// residual layers then take their 3x3 operands from Y through the in-LDS transposition of res_pair8_h2_kernel's second
// layer and their skip from Y itself.  The conv's output map and the first layer's output map never exist in memory.
// Operand order per accumulator = the separate kernels' (chunk, tap, k-step for the front conv as in
// conv_tile8_bf3_kernel; slice, tap for the residual 3x3): results are bitwise those of the separate launches.
// NT3 > 0: the 1x1 conv behind the pair as in res_pair8_h2_kernel.
struct FrontConv {
    const u32x4 *wimg;             // two-term fp16 image of the front conv (vqvae_conv_pack_f32), phase 0
    const int *hdr;                // {kw}
    const float *bias;
    unsigned long long dym, dxm;   // 4 bits per tap: dy + 8, dx + 8 (ConvGeom)
    int Cin;                       // multiple of 32
};

// x[lanes 32..63] <-> z[lanes 0..31] (v_permlane32_swap_b32)
__device__ __forceinline__ void swap_halves(float &x, float &z) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(z), false, false);
    x = __uint_as_float(r[0]);
    z = __uint_as_float(r[1]);
}
// A TRANSPOSED accumulator tile (weights as the A operand: lane = pixel l31, register r = channel (r & 3) + 8 (r >> 2) + 4 h
// of the 32-channel tile) turned into the B operands of the next GEMM's two 16-deep k-steps, in the weight images' channel
// order (k-step t, half h, element q = channel 16 h + 8 t + q): four half-wave register swaps per k-step bring channels
// 16 h + 8 t + [0, 4) and + [4, 8) into one lane; no trip through LDS.
__device__ __forceinline__ void acc_to_ksteps(const f32x16 &a, float sc, u32x4 (&t1)[2], u32x4 (&t2)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float P[4], Q[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            P[q] = a[4 * t + q];
            Q[q] = a[8 + 4 * t + q];
            swap_halves(P[q], Q[q]);
        }
        split8_h(f32x4{P[0], P[1], P[2], P[3]}, f32x4{Q[0], Q[1], Q[2], Q[3]}, sc, t1[t], t2[t]);
    }
}
// prod3x2 with the operands exchanged: acc^T += W^T x^T (same products, same k order, transposed result)
__device__ __forceinline__ void prod3x2t(const u32x4 &s1, const u32x4 &s2, const u32x4 &t1, const u32x4 &t2,
                                         const u32x4 &w1, const u32x4 &w2, f32x16 &accA, f32x16 &accB) {
#define HF(v) __builtin_bit_cast(f16x8, v)
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w1), HF(s2), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w1), HF(t2), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w2), HF(s1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w2), HF(t1), accB, 0, 0, 0);
    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w1), HF(s1), accA, 0, 0, 0);
    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(HF(w1), HF(t1), accB, 0, 0, 0);
#undef HF
}

// Everything is computed TRANSPOSED (weights = A operand, pixels = B operand): an accumulator lane then owns one pixel and
// its registers run over channels, which is the B-operand layout of the next GEMM up to half-wave swaps (acc_to_ksteps) --
// the 1x1 GEMMs take their inputs straight from registers and the 3x3 slices go registers -> fp16 planes in LDS without
// the accumulator -> LDS -> transposed read round trip of res_pair8_h2_kernel.
#ifndef CRP_NW
#define CRP_NW 4        // waves (= images) per workgroup of conv_res_pair8_h2_kernel.  8 (one workgroup per CU, weights shared by eight
                        // images, every stage barrier spanning all waves of the CU; tools/build_variant.py nw8 -DCRP_NW=8) is 17 us per
                        // step SLOWER: 0.589 vs 0.572 ms for the two launches
#endif
#ifndef CRP_MINW
#define CRP_MINW 2      // waves per SIMD the register allocation must allow (tools/build_variant.py crp1 -DCRP_MINW=1: 388 registers, no
                        // scratch, one workgroup per CU)
#endif
template <int NT3, bool VQ = false>
__global__ __launch_bounds__(CRP_NW * 64, CRP_MINW) void conv_res_pair8_h2_kernel(const float *__restrict__ in, FrontConv fc,
                                                                   const u32x4 *__restrict__ w1img, const u32x4 *__restrict__ w2img,
                                                                   float *__restrict__ out, int B, int flags,
                                                                   const int *__restrict__ hdr1, const int *__restrict__ hdr2,
                                                                   const int *__restrict__ in_amax, int *__restrict__ out_amax,
                                                                   const u32x4 *__restrict__ w3img, const int *__restrict__ hdr3,
                                                                   const float *__restrict__ bias3, float *__restrict__ out3,
                                                                   int *__restrict__ zero_buf, int zero_n, VqFuse vq) {
    static_assert(NT3 == 0 || NT3 == 1 || NT3 == 2 || NT3 == 4, "the 1x1 post conv streams through NT3 weight stages of 16 KiB");
    static_assert(!VQ || (NT3 == 2 && CRP_NW == 4), "the fused quantizer takes the 64-channel z_e of four images per workgroup");
    constexpr int NT2 = 4, C = 128, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // front conv: [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int RBUF = 4 * HP;                           // residual slice: [term 2][half 2][pixel + zero]; two buffers = TILE4
    __shared__ u32x4 As_all[CRP_NW * TILE4];
    // Weights stream through two 18 KiB LDS buffers shared by the workgroup's four images, filled by LDS-DMA (no staging
    // registers) one stage ahead; one workgroup barrier per stage.  Stages: one (tap, chunk) of the front conv (16 pieces
    // of 1 KiB); then per residual layer the nine taps of each 16-channel slice of the 3x3 (18 pieces) x 8 and the 1x1
    // (16 pieces); then the 1x1 post conv in NT3 parts of 4 / NT3 channel tiles (16 pieces each).  Per-wave loads straight from L2 cost 87 + 51 us
    // per step in exposed latency (knock-outs, profiles/r02_vq_stream.txt).
    constexpr int WBUF = 18 * 64;
    // stages after the front conv: 18 (two residual layers) + NT3 (post conv) + (VQ) one per four 32-code tiles of the codebook
    const int nvq = VQ ? (vq.K32 >> 7) : 0;
    const int NSTAGE = 18 + NT3 + nvq;
    __shared__ u32x4 Wb_all[2 * WBUF];
    // fused quantizer: per-wave tables (vq_unit.h), the workgroup's histogram and loss partials
    __shared__ __attribute__((aligned(16))) unsigned char vq_tab_all[VQ ? CRP_NW * 1040 : 16];
    __shared__ int vq_hist_s[VQ ? 512 : 1];
    // the four weight tensors' per-output-channel scales 2^-kw[c]: front conv [0, 128), residual 3x3 [128, 160), residual 1x1
    // [160, 288), post conv [288, 288 + 32 NT3) (a stage barrier precedes every use)
    __shared__ __attribute__((aligned(16))) float dw_s[288 + 32 * (NT3 > 0 ? NT3 : 1)];
    for (int i = threadIdx.x; i < 288 + 32 * NT3; i += CRP_NW * 64)
        dw_s[i] = i < 128 ? h2_dw(fc.hdr)[i] : (i < 160 ? h2_dw(hdr1)[i - 128] : (i < 288 ? h2_dw(hdr2)[i - 160] : h2_dw(hdr3)[i - 288]));
    __shared__ double vq_red_s[VQ ? CRP_NW : 1];
    if constexpr (VQ) {
        for (int i = threadIdx.x; i < vq.K; i += CRP_NW * 64) vq_hist_s[i] = 0;        // (a stage barrier precedes every use)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    float *Hs = reinterpret_cast<float *>(As);
    const bool relu_out = flags & kFlagReluOut;            // of the SECOND residual layer (the stack's final ReLU)
    constexpr int cpt = C >> 5;

    const long long img = (long long)blockIdx.x * CRP_NW + wave;
    const bool img_ok = img < B;
    // a buffer the NEXT kernel of the stream wants zeroed (the quantizer's histogram: saves a fill launch per step)
    if (zero_buf && blockIdx.x == 0)
        for (int i = tid; i < zero_n; i += CRP_NW * 64) zero_buf[i] = 0;

    // pixel bookkeeping: the residual 3x3 (taps t/3-1, t%3-1) and the front conv (taps from the geometry masks)
    int spx[MT];
    unsigned tapok[MT], tapok0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, x = spx[mt] & 7;
        unsigned m = 0, m0 = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << t;
            const int y0 = y + (int)((fc.dym >> (4 * t)) & 15) - 8, x0 = x + (int)((fc.dxm >> (4 * t)) & 15) - 8;
            if (y0 >= 0 && y0 < 8 && x0 >= 0 && x0 < 8) m0 |= 1u << t;
        }
        tapok[mt] = m;
        tapok0[mt] = m0;
    }

    // Y[mt][nt][r]: channel 32 nt + (r & 3) + 8 (r >> 2) + 4 h of pixel 32 mt + l31
    f32x16 Y[MT][NT2];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // one 1 KiB piece global -> LDS: every lane's 16 bytes land at dst + 16 lane
    // Issued as inline assembly: for the builtin hipcc puts s_waitcnt vmcnt(0) in front of every later LDS read that it
    // cannot prove disjoint from the destination -- i.e. it waits for the NEXT stage's pieces before reading this stage's.
    // The waits are explicit here (dma_wait_sync); the compiler's own vmcnt waits stay correct (loads return in order and an
    // uncounted outstanding load only makes a counted wait longer).
    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // the nine taps of slice sl of the residual 3x3 -> buffer `buf`: piece p = tap * 2 + term
    auto dma_slice = [&](int sl, int buf) {
        const u32x4 *base = w1img + (size_t)(sl >> 1) * 256 + (sl & 1) * 64;
        for (int p = wave_u; p < 18; p += CRP_NW)
            dma(base + (size_t)(p >> 1) * cpt * 256 + (p & 1) * 128, Wb_all + buf * WBUF + p * 64);
    };
    // 16 KiB of an image as it lies (the 1x1 GEMMs): four pieces per wave
    auto dma_linear = [&](const u32x4 *src, int buf) {
#pragma unroll
        for (int j = 0; j < 16 / CRP_NW; ++j) dma(src + (wave_u * (16 / CRP_NW) + j) * 64, Wb_all + buf * WBUF + (wave_u * (16 / CRP_NW) + j) * 64);
    };
    // stage k after the front conv: 9 LI + slice (3x3 of layer LI), 9 LI + 8 (its 1x1), 18 + j (part j of the post conv)
    auto dma_stage = [&](int k, int buf) {
        if (VQ && k >= 18 + NT3) {
            // four 32-code tiles of the codebook's fp16 image (16 pieces) + their seeds -A ee / 2 (512 bytes of piece 16)
            const int j = k - (18 + NT3);
            dma_linear(reinterpret_cast<const u32x4 *>(vq.imgf) + (size_t)j * 1024, buf);
            if (wave_u == 0) dma(reinterpret_cast<const u32x4 *>(vq.seeds) + (size_t)j * 32, Wb_all + buf * WBUF + 16 * 64);
        }
        else if (k >= 18) dma_linear(w3img + (size_t)(k - 18) * 1024, buf);
        else if (k % 9 == 8) dma_linear(w2img, buf);
        else dma_slice(k % 9, buf);
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    float ymax = 0.0f;                                     // largest |Y| (the next consumer's scale)
    int wstage = 0;                                        // weight stages of the front conv (LDS buffer parity)
    // =========================================== front conv ===========================================
    {
        const int cpt0 = fc.Cin >> 5;
        const float *src = in + ((size_t)(img_ok ? img : 0) * PX + lane) * fc.Cin;     // this lane's pixel row
        f32x4 raw[8];
        auto load_raw0 = [&](int cc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4 *>(src + 32 * cc + 4 * j);
        };
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;
        if (given >= 0) m = __int_as_float(given);
        else for (int cc = 0; cc < cpt0; ++cc) {
            load_raw0(cc);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(raw[j].x), __builtin_fabsf(raw[j].y)), fmaxf(__builtin_fabsf(raw[j].z), __builtin_fabsf(raw[j].w))));
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        const float xs = __builtin_ldexpf(1.0f, kx), d0 = __builtin_ldexpf(1.0f, -kx);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[mt][nt][r] = 0.0f;
        // padding pixels of the eight planes (the residual slices' two plane buffers have theirs at the same units)
        if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};
        // stage (cc, tap) = 16 pieces: piece p = nt * 4 + t * 2 + term, four per wave
        auto dma_front = [&](int cc, int tap, int buf) {
            const u32x4 *base = fc.wimg + (size_t)(tap * cpt0 + cc) * (NT2 * 256);
#pragma unroll
            for (int j = 0; j < 16 / CRP_NW; ++j) {
                const int p = wave_u * (16 / CRP_NW) + j;
                dma(base + (p >> 2) * 256 + ((p >> 1) & 1) * 64 + (p & 1) * 128, Wb_all + buf * WBUF + p * 64);
            }
        };
        load_raw0(0);
        dma_front(0, 0, 0);
        for (int cc = 0; cc < cpt0; ++cc) {
            // park the chunk: k-step t, half hh hold channels 16 hh + 8 t + [0, 8) (conv_tile8_bf3_kernel's stage())
            __builtin_amdgcn_wave_barrier();
            u32x4 *dst = As + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    u32x4 t1, t2;
                    split8_h(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], xs, t1, t2);
                    dst[(t * 2 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 2 + 1) * PLANE + hh * HP] = t2;
                }
            lds_order_wave();
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap, ++wstage) {
                // the pixel operands do not depend on the stage buffer: read them before the barrier
                const int shift = ((int)((fc.dym >> (4 * tap)) & 15) - 8) * 8 + ((int)((fc.dxm >> (4 * tap)) & 15) - 8);
                u32x4 X[2][MT][2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int p = ((tapok0[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                        const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                        X[t][mt][0] = ap[0];
                        X[t][mt][1] = ap[PLANE];
                    }
                // this stage's weights are in; everyone is done with the other buffer.  The next chunk's eight activation loads
                // go out BEHIND tap 1's weights and may stay in flight across tap 1's wait (they are its youngest requests):
                // in front of tap 0's wait, as before, every chunk sat out their whole latency at that barrier
                if (tap == 1 && cc + 1 < cpt0) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __syncthreads();
                } else dma_wait_sync();
                if (tap + 1 < 9) dma_front(cc, tap + 1, (wstage + 1) & 1);
                else if (cc + 1 < cpt0) dma_front(cc + 1, 0, (wstage + 1) & 1);
                else dma_stage(0, (wstage + 1) & 1);           // the first slice of the first residual layer
                if (tap == 0 && cc + 1 < cpt0) load_raw0(cc + 1);
                const u32x4 *wt = Wb_all + (wstage & 1) * WBUF + lane;
                // group g = (t, nt): weights one group ahead of the matrix instructions
                u32x4 Wc0 = wt[0], Wc1 = wt[64];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int t = g >> 2, nt = g & 3;
                    u32x4 Wn0 = Wc0, Wn1 = Wc1;
                    if (g + 1 < 8) {
                        const u32x4 *bp = wt + (((g + 1) & 3) * 4 + ((g + 1) >> 2) * 2) * 64;
                        Wn0 = bp[0];
                        Wn1 = bp[64];
                    }
                    __builtin_amdgcn_sched_barrier(0);         // hipcc otherwise sinks the reads to just before their use
                    prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, Y[0][nt], Y[1][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    Wc0 = Wn0;
                    Wc1 = Wn1;
                }
            }
        }
        // bias + ReLU (encoder.py:36 / the stack's first in-place ReLU applied by the producer, decoder.py:29-30)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (fc.bias) bv = *reinterpret_cast<const f32x4 *>(fc.bias + nt * 32 + 8 * g + 4 * h);
                const f32x4 dv = h2_dw4(dw_s, nt * 32, g, h, d0);          // 2^-(kx + kw[c]) of registers 4 g .. 4 g + 3
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; q += 2)
                        SCALE2_BIAS_RELU2(Y[mt][nt][4 * g + q], Y[mt][nt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], ymax);
            }
    }
    lds_order_wave();

    // =========================================== residual layers from Y ===========================================
    // stage k: wait for its weights, start the next stage's, return this lane's column of its buffer
    auto stage_sync = [&](int k) -> const u32x4 * {
        dma_wait_sync();
        if (k + 1 < NSTAGE) dma_stage(k + 1, (wstage + k + 1) & 1);
        return Wb_all + ((wstage + k) & 1) * WBUF + lane;
    };
    f32x16 acc1[MT];
    // nine taps of one slice: wb = this lane's column of the stage buffer ([tap * 2 + term] x 64 units), pl = this half-wave's
    // planes of the slice; operands of tap + 1 are read while tap's products run
    auto taps = [&](const u32x4 *wb, const u32x4 *pl) {
        u32x4 Xc[MT][2], Wc[2];
        auto ld = [&](int tap, u32x4(&X)[MT][2], u32x4(&W)[2]) {
            const int shift = (tap / 3 - 1) * 8 + (tap % 3 - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = ((tapok[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                X[mt][0] = pl[p];
                X[mt][1] = pl[p + HP * 2];
            }
            W[0] = wb[tap * 128];
            W[1] = wb[tap * 128 + 64];
        };
        ld(0, Xc, Wc);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            u32x4 Xn[MT][2], Wn[2];
            if (tap + 1 < 9) ld(tap + 1, Xn, Wn);
            __builtin_amdgcn_sched_barrier(0);
            prod3x2t(Xc[0][0], Xc[0][1], Xc[1][0], Xc[1][1], Wc[0], Wc[1], acc1[0], acc1[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { Xc[mt][0] = Xn[mt][0]; Xc[mt][1] = Xn[mt][1]; }
                Wc[0] = Wn[0]; Wc[1] = Wn[1];
            }
        }
    };
    // one k-step's operands of both pixel tiles -> plane buffer `buf`
    auto put_planes = [&](int buf, const u32x4(&T1)[MT][2], const u32x4(&T2)[MT][2], int t) {
        u32x4 *pb = As + buf * RBUF + h * HP + l31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            pb[32 * mt] = T1[mt][t];
            pb[32 * mt + 2 * HP] = T2[mt][t];
        }
    };
    // Y <- [relu](Y + W2 relu(W1 (*) Y)); ymax in: largest Y, out: largest new Y.  always_inline: hipcc does not inline a
    // lambda this size twice by itself, and Y (captured by reference) then lives in scratch memory -- 5 ms per launch,
    // measured; a two-iteration loop around the body instead spills 275 registers
    auto layer = [&](auto LT, bool relu_after) __attribute__((always_inline)) {
        constexpr int LI = decltype(LT)::value;                // 0 or 1: stages 9 LI ..
        const int kx = wave_scale_exp(img_ok ? ymax : 0.0f);
        const float xscale = __builtin_ldexpf(1.0f, kx), d1 = __builtin_ldexpf(1.0f, -kx);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;
        u32x4 T1[MT][2], T2[MT][2];                            // [pixel tile][k-step] of the current 32-channel tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][0], xscale, T1[mt], T2[mt]);
        __builtin_amdgcn_wave_barrier();
        put_planes(0, T1, T2, 0);
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            // slice (c, 0) from plane buffer 0; (c, 1)'s planes go to buffer 1 (last read by slice (c - 1, 1))
            __builtin_amdgcn_wave_barrier();
            put_planes(1, T1, T2, 1);
            lds_order_wave();
            taps(stage_sync(9 * LI + 2 * c), As + h * HP);
            // slice (c, 1); the next tile's operands are made now and its first k-step goes to buffer 0
            if (c + 1 < NT2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][c + 1], xscale, T1[mt], T2[mt]);
                __builtin_amdgcn_wave_barrier();
                put_planes(0, T1, T2, 0);
            }
            lds_order_wave();
            taps(stage_sync(9 * LI + 2 * c + 1), As + RBUF + h * HP);
        }
        // hidden tile -> B operands of the 1x1 GEMM, in registers
        float m = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 dv = h2_dw4(dw_s + 128, 0, g, h, d1);            // 2^-(kx + kw1[j]) of the hidden channels 8 g + 4 h + [0, 4)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; q += 2)
                    SCALE2_BIAS_RELU2(acc1[mt][4 * g + q], acc1[mt][4 * g + q + 1], dv[q], dv[q + 1], 0.0f, 0.0f, m);
        }
        const int kh = wave_scale_exp(m);
        const float hscale = __builtin_ldexpf(1.0f, kh), d2 = __builtin_ldexpf(1.0f, -kh);
        u32x4 H1[MT][2], Hb[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(acc1[mt], hscale, H1[mt], Hb[mt]);
        const u32x4 *wb = stage_sync(9 * LI + 8);              // [nt][term][k-step] x 64 units
        float nmax = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            // (no read-ahead here: four short GEMMs per layer, and 32 more live registers spill next to Y, H and acc2)
            u32x4 Wc[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { Wc[t][0] = wb[nt * 256 + t * 64]; Wc[t][1] = wb[nt * 256 + t * 64 + 128]; }
            f32x16 acc2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                prod3x2t(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], Wc[t][0], Wc[t][1], acc2[0], acc2[1]);
            // Y <- [relu](Y + acc2 * 2^-k), nmax: FMA (exact product), single-instruction max; the ReLU flag is wave-uniform and
            // decided once per tile, not per value
            // (the 1x1 rows' own scales: 2^-(kh + kw2[c]) of registers 4 g .. 4 g + 3, four at a time)
            if (relu_after) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dv = h2_dw4(dw_s + 160, nt * 32, g, h, d2);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const int r = 4 * g + q;
                            const float y0 = vmax(__builtin_fmaf(acc2[mt][r], dv[q], Y[mt][nt][r]), 0.0f);
                            const float y1 = vmax(__builtin_fmaf(acc2[mt][r + 1], dv[q + 1], Y[mt][nt][r + 1]), 0.0f);
                            Y[mt][nt][r] = y0;
                            Y[mt][nt][r + 1] = y1;
                            vmax3(nmax, y0, y1);
                        }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dv = h2_dw4(dw_s + 160, nt * 32, g, h, d2);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const int r = 4 * g + q;
                            const float y0 = __builtin_fmaf(acc2[mt][r], dv[q], Y[mt][nt][r]), y1 = __builtin_fmaf(acc2[mt][r + 1], dv[q + 1], Y[mt][nt][r + 1]);
                            Y[mt][nt][r] = y0;
                            Y[mt][nt][r + 1] = y1;
                            vmax3_abs(nmax, y0, y1);
                        }
                }
            }
        }
        ymax = nmax;
    };
    layer(std::integral_constant<int, 0>{}, true);         // the second layer's in-place ReLU is applied by its producer
    layer(std::integral_constant<int, 1>{}, relu_out);
    if (out_amax && img_ok) publish_amax_exclusive(out_amax, img, ymax, lane);

    // one transposed 32-pixel x 32-channel tile -> rows of `ld` floats at dst (pixel-major), whole 128-byte lines per
    // eight lanes: registers -> wave-private LDS tile [pixel][36] -> linear 16-byte reads
    auto store_tile = [&](const float(&v)[16], float *dst, int ld) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(Hs + l31 * 36 + 8 * g + 4 * h) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        lds_order_wave();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = (lane >> 3) + 8 * k;
            *reinterpret_cast<f32x4 *>(dst + (size_t)p * ld + 4 * (lane & 7)) = *reinterpret_cast<const f32x4 *>(Hs + p * 36 + 4 * (lane & 7));
        }
    };
    const long long wbase = img * PX;
    if constexpr (NT3 > 0) {
        const int kx3 = wave_scale_exp(img_ok ? ymax : 0.0f);
        const float xs3 = __builtin_ldexpf(1.0f, kx3), d3 = __builtin_ldexpf(1.0f, -kx3);
        f32x16 acc3[MT][NT3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mt][n3][r] = 0.0f;
        const u32x4 *wb = nullptr;
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            u32x4 T1[MT][2], T2[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][c], xs3, T1[mt], T2[mt]);
            constexpr int CPS = 4 / NT3;                       // channel tiles per stage: [c % CPS][n3][term][k-step] x 64 units
            if (c % CPS == 0) wb = stage_sync(18 + c / CPS);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) {
                    const u32x4 *bp = wb + ((c % CPS) * NT3 + n3) * 256 + t * 64;
                    prod3x2t(T1[0][t], T2[0][t], T1[1][t], T2[1][t], bp[0], bp[128], acc3[0][n3], acc3[1][n3]);
                }
        }
        if constexpr (VQ) {
            // ================= the quantizer on z_e = this image's 64 rows, straight from the accumulators =================
            // (models/vqvae.py:33-34: z_e is never written.)  acc3 <- z_e: lane = row 32 mt + l31, register = channel
            // 32 n3 + (r & 3) + 8 (r >> 2) + 4 h.  The codebook's fp16 image streams through the weight stages (four 32-code
            // tiles + their seeds per stage, the image in THIS kernel's channel order: vq_prepare16_kernel's `imgf`); the sweep,
            // the trackers and everything behind them are vq_track.hip's (vq_track.h / vq_unit.h).
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias3) bv = *reinterpret_cast<const f32x4 *>(bias3 + n3 * 32 + 8 * g + 4 * h);
                        const f32x4 dv = h2_dw4(dw_s + 288, n3 * 32, g, h, d3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc3[mt][n3][4 * g + q] = acc3[mt][n3][4 * g + q] * dv[q] + bv[q];
                    }
            // fp16 B operands of the screen: k-step ks = 2 n3 + t, this half's channels 32 n3 + 16 h + 8 t + [0, 8)
            u32x4 zb[MT][4];
            float zn2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float sq = 0.0f;
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float P[4], Q[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            P[q] = acc3[mt][n3][4 * t + q];
                            Q[q] = acc3[mt][n3][8 + 4 * t + q];
                            swap_halves(P[q], Q[q]);
                        }
                        u32x4 v;
                        v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{P[0], P[1]}), f16x2));
                        v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{P[2], P[3]}), f16x2));
                        v.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{Q[0], Q[1]}), f16x2));
                        v.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{Q[2], Q[3]}), f16x2));
                        zb[mt][2 * n3 + t] = v;
                        sq = sqsum8_f16(v.x, v.y, v.z, v.w, sq);       // (not four fdot2 builtins: miscompiled, common.h)
                    }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
                zn2[mt] = sq + __uint_as_float(h ? sw[0] : sw[1]);
            }
            const float inf = __builtin_inff();
            float pinf = inf, ninf = -inf;
            unsigned keymask = trk::kKeyMask;
            asm volatile("" : "+v"(pinf), "+v"(ninf), "+v"(keymask));
            trk::Lane L[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) trk::init(L[mt], ninf);
            // one stage = four code tiles: [tile][k-step 4][half 2][code 32] x 16 bytes, seeds [tile][half][16] floats in piece 16
            auto sweep_stage = [&](const u32x4 *wb, const float *sd, int j, auto &&use) {
#pragma unroll
                for (int ctl = 0; ctl < 4; ++ctl) {
                    f32x16 seed;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sd + ctl * 32 + h * 16 + 4 * g);
                        seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                    }
                    u32x4 a[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) a[ks] = wb[ctl * 256 + ks * 64];
                    use(4 * j + ctl, a, seed);
                }
            };
            for (int j = 0; j < nvq; ++j) {
                const u32x4 *wb = stage_sync(18 + NT3 + j);
                const float *sd = reinterpret_cast<const float *>(wb - lane + 16 * 64);
                sweep_stage(wb, sd, j, [&](int ct, const u32x4(&a)[4], const f32x16 &seed) {
                    f32x16 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, zb[mt][0]), seed, 0, 0, 0);
#pragma unroll
                        for (int ks = 1; ks < 4; ++ks)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks]), __builtin_bit_cast(f16x8, zb[mt][ks]), acc[mt], 0, 0, 0);
                    }
                    unsigned cell0 = (unsigned)(2 * ct), cell1 = cell0 + 1u;
                    asm volatile("" : "+s"(cell0), "+s"(cell1));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) trk::tile(L[mt], acc[mt], cell0, cell1, keymask, ninf, pinf);
                });
            }
            // ---- verdicts; exact tasks
            vqu::Tables tb = vqu::tables(vq_tab_all + wave_u * 1040);
            const vqu::Bound bound = vqu::load_bound(vq.flags);
            vqu::Rows R;
            R.valid[0] = img_ok;
            R.valid[1] = img_ok;
            vqu::classify(L, zn2, bound, vq.K, lane, ninf, tb.task_s, R);
            vqu::Flagged FL = vqu::exact_begin(R, lane, tb);
            int ntasks = FL.ndirect;
            // rows whose candidates the stream x cell products do not cover (~0.01 %) need the codebook image once more: the
            // workgroup votes, and if any of its waves has one, all four stream the stages again (the others only keep the barriers)
            const bool rescan_me = FL.hmask && FL.ndirect <= 64;
            if (__syncthreads_or(rescan_me ? 1 : 0)) {
                dma_stage(18 + NT3, 0);
                for (int j = 0; j < nvq; ++j) {
                    dma_wait_sync();
                    if (j + 1 < nvq) dma_stage(18 + NT3 + j + 1, (j + 1) & 1);
                    if (rescan_me) {
                        const u32x4 *wb = Wb_all + (j & 1) * WBUF + lane;
                        const float *sd = reinterpret_cast<const float *>(Wb_all + (j & 1) * WBUF + 16 * 64);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            if ((unsigned)(FL.hmask >> (32 * mt))) {
                                const float thr_t = R.hardf[mt] ? R.thr[mt] : inf;
                                sweep_stage(wb, sd, j, [&](int ct, const u32x4(&a)[4], const f32x16 &seed) {
                                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, zb[mt][0]), seed, 0, 0, 0);
#pragma unroll
                                    for (int ks = 1; ks < 4; ++ks)
                                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks]), __builtin_bit_cast(f16x8, zb[mt][ks]), acc, 0, 0, 0);
                                    vqu::rescan_tile(acc, thr_t, ct, mt, lane, vq.K, FL.ndirect, ninf, tb);
                                });
                            }
                    }
                }
                if (rescan_me) {
                    lds_order_wave();
                    ntasks = FL.ndirect + tb.cnt_s[0];
                }
            }
            __syncthreads();                        // every wave is done with the weight buffers: they hold half of the rows now
            // ---- z_e rows (fp32) -> LDS: rows 0..31 in this wave's plane region, 32..63 in its quarter of the weight buffers;
            // 256 bytes per row, the 16-byte chunk c of row r at slot c ^ (r & 15)
            unsigned char *zlo = reinterpret_cast<unsigned char *>(As);
            unsigned char *zhi = reinterpret_cast<unsigned char *>(Wb_all) + (size_t)wave_u * 9216;
            auto zchunk = [&](int row, int c16) -> f32x4 * {
                unsigned char *b = row < 32 ? zlo + row * 256 : zhi + (row - 32) * 256;
                return reinterpret_cast<f32x4 *>(b + (((unsigned)c16 ^ ((unsigned)row & 15u)) << 4));
            };
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *zchunk(32 * mt + l31, 8 * n3 + 2 * g + h) = f32x4{acc3[mt][n3][4 * g], acc3[mt][n3][4 * g + 1], acc3[mt][n3][4 * g + 2], acc3[mt][n3][4 * g + 3]};
            lds_order_wave();
            vqu::exact_end(R, FL, ntasks, lane, tb, vq.cb, vq.ee, vq.K,
                           [&](int rr, int jc) { return *zchunk(rr, jc); },
                           [&](int rr, int c) { return reinterpret_cast<const float *>(zchunk(rr, c >> 2))[c & 3]; });
            const int j16 = lane & 15, g4 = lane >> 4;
            const float sacc = vqu::epilogue(R, lane, vq.cb, vq.K, [&](int t, int i) { return *zchunk(32 * t + 4 * i + g4, j16); },
                                             (img_ok && vq.zq) ? vq.zq + (size_t)img * PX * 64 : nullptr, img_ok ? PX : 0,
                                             vq.idx + (size_t)(img_ok ? img : 0) * PX, vq_hist_s);
            // loss partial and histogram of the workgroup (fixed order: run-to-run bitwise loss / perplexity)
            double dacc = img_ok ? (double)sacc : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
            if (lane == 0) vq_red_s[wave_u] = dacc;
            __syncthreads();
            if (tid == 0) {
                double sum = 0.0;
                for (int w = 0; w < CRP_NW; ++w) sum += vq_red_s[w];
                vq.partials[blockIdx.x] = sum;
            }
            for (int k = tid; k < vq.K; k += CRP_NW * 64) {
                const int c = vq_hist_s[k];
                if (c) atomicAdd(&vq.hist[k], c);
            }
        } else
        if (img_ok) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) {
                    float v[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias3) bv = *reinterpret_cast<const f32x4 *>(bias3 + n3 * 32 + 8 * g + 4 * h);
                        const f32x4 dv = h2_dw4(dw_s + 288, n3 * 32, g, h, d3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[4 * g + q] = acc3[mt][n3][4 * g + q] * dv[q] + bv[q];
                    }
                    store_tile(v, out3 + (wbase + mt * 32) * (32 * NT3) + n3 * 32, 32 * NT3);
                }
        }
    } else if (img_ok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = Y[mt][nt][r];
                store_tile(v, out + (wbase + mt * 32) * C + nt * 32, C);
            }
    }
}

// ---------------------------------------------------------------------------
// Encoder front in one launch (models/encoder.py:29-34): Conv2d(3 -> 64, 4x4 s2 p1) + ReLU + Conv2d(64 -> 128, 4x4 s2 p1)
// + ReLU on 32x32 images; the 16x16x64 map between them (64 KiB per image, written and read back by the separate
// kernels: 536 MB per 4096 images) never exists.  One wave owns one image; everything is computed transposed as in
// conv_res_pair8_h2_kernel.  The second conv runs as in conv_tile8_bf3_kernel<.., S2D>: a conv over the 8x8 grid of 2x2
// blocks of the 16x16 map, chunk = (block sub-position s, 32-channel slice), four block offsets (taps) per chunk.  Per
// sub-position the wave builds ITS OWN operand slice: the 64 block pixels' 4x4x3 input patches are gathered from the NCHW
// image (two 16-byte loads per channel and lane half: rows ky = 2h, 2h+1), split into fp16 terms and multiplied with the
// first layer's weights on the matrix cores (3 k-steps of 16 = the 48 taps; +9 % matrix work), + bias, ReLU, and the
// accumulator becomes the second conv's B operands by half-wave swaps (acc_to_ksteps).
// Scales: the image's largest |x| is measured; the first layer's outputs are bounded by L1 * max|x| + max|b| (L1 = the
// largest absolute row sum of its weights, in the header) -- a power of two up to ~8x above the true maximum, which costs
// the second term's range three bits at the very bottom and nothing where it matters (see split8_h).
#ifndef EF_MINW
#define EF_MINW 2
#endif
template <int CIN>
__global__ __launch_bounds__(256, EF_MINW) void enc_front8_h2_kernel(const float *__restrict__ x, const u32x4 *__restrict__ w0img,
                                                               const int *__restrict__ hdr0, const float *__restrict__ bias0,
                                                               const u32x4 *__restrict__ w2img, const int *__restrict__ hdr2,
                                                               const float *__restrict__ bias2, float *__restrict__ out, int B,
                                                               int *__restrict__ out_amax, int *__restrict__ zero_buf, int zero_n) {
    constexpr int NT = 4, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2, C0 = 64, C = 128;
    // ints a LATER kernel of the stream wants zeroed (the quantizer's histogram when the encoder's last kernel quantizes)
    if (zero_buf && blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_buf[i] = 0;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int WBUF = 16 * 64, NSTAGE = 32;             // a stage = one (chunk, tap) of the second conv: 16 pieces of 1 KiB
    __shared__ u32x4 As_all[4 * TILE4];
    __shared__ u32x4 Wb_all[2 * WBUF];
    __shared__ u32x4 W0s[2 * CIN * 2 * 64];                // first layer: [slice 2][ci][term 2] x 64 lanes
    __shared__ __attribute__((aligned(16))) float dw_s[64 + 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    float *Hs = reinterpret_cast<float *>(As);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;

    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // stage k = chunk * 4 + tap, chunk = 2 s + slice: 16 KiB as it lies in the space-to-depth image
    auto dma_stage = [&](int k, int buf) {
        const u32x4 *src = w2img + (size_t)k * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(src + (wave_u * 4 + j) * 64, Wb_all + buf * WBUF + (wave_u * 4 + j) * 64);
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    dma_stage(0, 0);
    for (int i = tid; i < 2 * CIN * 2 * 64; i += 256) W0s[i] = w0img[i];
    // both layers' per-output-channel weight scales 2^-kw[c]: first layer [0, 64), second [64, 192) (behind the W0s barrier)
    if (tid < C0 + C) dw_s[tid] = tid < C0 ? h2_dw(hdr0)[tid] : h2_dw(hdr2)[tid - C0];
    if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};       // padding pixels of the four planes

    // block-pixel bookkeeping: bit sub * 4 + tap of tapok = block offset ((tap >> 1) - (sub >> 1), (tap & 1) - (sub & 1)) is inside
    int spx[MT];
    unsigned tapok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, xx0 = spx[mt] & 7;
        unsigned m = 0;
        for (int q = 0; q < 16; ++q) {
            const int yy = y + ((q >> 1) & 1) - (q >> 3), xx = xx0 + (q & 1) - ((q >> 2) & 1);
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << q;
        }
        tapok[mt] = m;
    }

    // scales: the image's largest |x| -> the first layer's operand scale; the bound on its outputs -> the second layer's
    const float *ximg = x + (size_t)(img_ok ? img : 0) * (CIN * 1024);
    float xm = 0.0f;
#pragma unroll
    for (int j = 0; j < CIN * 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ximg + 4 * lane + 256 * j);
        xm = fmaxf(xm, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xm = fmaxf(xm, __shfl_xor(xm, o));
    const int kx0 = wave_scale_exp(img_ok ? xm : 0.0f);
    const float xs0 = __builtin_ldexpf(1.0f, kx0), d0 = __builtin_ldexpf(1.0f, -kx0);       // (x the weight rows' 2^-kw[c] at the use)
    float bm = bias0 ? __builtin_fabsf(bias0[lane]) : 0.0f;                                 // C0 = 64 channels
    const float bound = (__int_as_float(hdr0[1]) * xm + bm) * 1.0001f;
    const int k1 = wave_scale_exp(img_ok ? bound : 0.0f);                                     // (reduces bm over the wave)
    const float xs1 = __builtin_ldexpf(1.0f, k1), d2 = __builtin_ldexpf(1.0f, -k1);

    f32x16 Y[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[mt][nt][r] = 0.0f;
    // (built from the wave-uniform image index: with a lane-derived one hipcc wraps every gather in a waterfall loop)
    const long long img_u = (long long)blockIdx.x * 4 + wave_u;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)(img_u < B ? img_u : 0) * (CIN * 1024)), 0,
                                                       (unsigned)(CIN * 4096), 0x00020000);
    __syncthreads();                                       // W0s

    // the four taps of chunk cc (its operand planes are in the wave's tile): stages 4 cc .. 4 cc + 3
    auto taps = [&](int cc) {
        const int sub = cc >> 1;
#pragma unroll 1
        for (int tap = 0; tap < 4; ++tap) {
            const int k = cc * 4 + tap;
            const int shift = ((tap >> 1) - (sub >> 1)) * 8 + ((tap & 1) - (sub & 1)), okbit = sub * 4 + tap;
            u32x4 X[2][MT][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = ((tapok[mt] >> okbit) & 1u) ? spx[mt] + shift : PX;
                    const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                    X[t][mt][0] = ap[0];
                    X[t][mt][1] = ap[PLANE];
                }
            dma_wait_sync();                               // this stage's weights are in; everyone is done with the other buffer
            if (k + 1 < NSTAGE) dma_stage(k + 1, (k + 1) & 1);
            const u32x4 *wt = Wb_all + (k & 1) * WBUF + lane;      // [nt][term][k-step] x 64 units
            u32x4 Wc0 = wt[0], Wc1 = wt[128];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int t = g >> 2, nt = g & 3;
                u32x4 Wn0 = Wc0, Wn1 = Wc1;
                if (g + 1 < 8) {
                    const u32x4 *bp = wt + ((g + 1) & 3) * 256 + ((g + 1) >> 2) * 64;
                    Wn0 = bp[0];
                    Wn1 = bp[128];
                }
                __builtin_amdgcn_sched_barrier(0);         // hipcc otherwise sinks the reads to just before their use
                prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, Y[0][nt], Y[1][nt]);
                __builtin_amdgcn_sched_barrier(0);
                Wc0 = Wn0;
                Wc1 = Wn1;
            }
        }
    };
    auto put_planes = [&](const u32x4(&T1)[MT][2], const u32x4(&T2)[MT][2]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                As[(t * 2 + 0) * PLANE + h * HP + 32 * mt + l31] = T1[mt][t];
                As[(t * 2 + 1) * PLANE + h * HP + 32 * mt + l31] = T2[mt][t];
            }
        lds_order_wave();
    };
    // first layer on the patches XB for output channels 32 sl .. +31: + bias, ReLU, -> the second layer's operands
    auto first_layer = [&](int sl, const u32x4(&XB)[MT][CIN][2], u32x4(&T1)[MT][2], u32x4(&T2)[MT][2]) {
        f32x16 acc0[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[mt][r] = 0.0f;
        const u32x4 *wp = W0s + (sl * CIN) * 128 + lane;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            prod3x2t(XB[0][ci][0], XB[0][ci][1], XB[1][ci][0], XB[1][ci][1], wp[ci * 128], wp[ci * 128 + 64], acc0[0], acc0[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias0) bv = *reinterpret_cast<const f32x4 *>(bias0 + sl * 32 + 8 * g + 4 * h);
            const f32x4 dv = h2_dw4(dw_s, sl * 32, g, h, d0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    float unused = 0.0f;
                    SCALE2_BIAS_RELU2(acc0[mt][4 * g + q], acc0[mt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], unused);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(acc0[mt], xs1, T1[mt], T2[mt]);
    };

#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int sy = s >> 1, sx = s & 1;
        // patches of the 64 block pixels at sub-position s: lane half h holds rows ky = 2h, 2h + 1 (4 columns each) per channel
        u32x4 XB[MT][CIN][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int by = spx[mt] >> 3, bx = spx[mt] & 7;
            const int col0 = 4 * bx + 2 * sx - 1;
            const int adj = col0 < 0 ? 1 : (col0 + 3 > 31 ? -1 : 0);       // edge lanes load one column off and shift
            f32x4 pv[CIN][2];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int row = 4 * by + 2 * sy - 1 + 2 * h + rr;
                    const unsigned off = (row >= 0 && row < 32) ? (unsigned)(((ci * 32 + row) * 32 + col0 + adj) * 4) : kOobOffset;
                    pv[ci][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
                }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const f32x4 v = pv[ci][rr];
                    f32x4 o;
                    o.x = adj > 0 ? 0.0f : (adj < 0 ? v.y : v.x);
                    o.y = adj > 0 ? v.x : (adj < 0 ? v.z : v.y);
                    o.z = adj > 0 ? v.y : (adj < 0 ? v.w : v.z);
                    o.w = adj > 0 ? v.z : (adj < 0 ? 0.0f : v.w);
                    pv[ci][rr] = o;
                }
                split8_h(pv[ci][0], pv[ci][1], xs0, XB[mt][ci][0], XB[mt][ci][1]);
            }
        }
        u32x4 T1[MT][2], T2[MT][2], U1[MT][2], U2[MT][2];
        first_layer(0, XB, T1, T2);
        put_planes(T1, T2);
        first_layer(1, XB, U1, U2);
        taps(2 * s);
        put_planes(U1, U2);
        taps(2 * s + 1);
    }

    // bias + ReLU (encoder.py:32-34), the image's maximum for the next layer, whole-line stores
    float ymax = 0.0f;
    const long long wbase = img * PX;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (bias2) bv = *reinterpret_cast<const f32x4 *>(bias2 + nt * 32 + 8 * g + 4 * h);
                const f32x4 dv = h2_dw4(dw_s + C0, nt * 32, g, h, d2);
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    v[4 * g + q] = Y[mt][nt][4 * g + q];
                    v[4 * g + q + 1] = Y[mt][nt][4 * g + q + 1];
                    SCALE2_BIAS_RELU2(v[4 * g + q], v[4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], ymax);
                }
            }
            if (img_ok) {
                float *dst = out + (wbase + mt * 32) * C + nt * 32;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(Hs + l31 * 36 + 8 * g + 4 * h) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                lds_order_wave();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p = (lane >> 3) + 8 * k;
                    *reinterpret_cast<f32x4 *>(dst + (size_t)p * C + 4 * (lane & 7)) = *reinterpret_cast<const f32x4 *>(Hs + p * 36 + 4 * (lane & 7));
                }
            }
        }
    if (out_amax && img_ok) publish_amax_exclusive(out_amax, img, ymax, lane);
    (void)C0;
}

// header of the first layer's two-term image, slot [1]: float bits of the largest absolute row sum of w (one block; the
// per-channel scales come from conv_wscale_kernel)
__global__ __launch_bounds__(256) void conv_in_hdr_kernel(const float *__restrict__ w, int per, int Cout, int *__restrict__ hdr) {
    __shared__ float red1[256];
    float l1 = 0.0f;
    for (int co = threadIdx.x; co < Cout; co += 256) {
        float s = 0.0f;
        for (int i = 0; i < per; ++i) s += __builtin_fabsf(w[(size_t)co * per + i]);
        l1 = fmaxf(l1, s);
    }
    red1[threadIdx.x] = l1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red1[threadIdx.x] = fmaxf(red1[threadIdx.x], red1[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) hdr[1] = __float_as_int(red1[0] * 1.0001f);
}
// two-term fp16 A-operand image of the first layer's weights * 2^kw[co]: [n_tile][ci][term] x 64 lanes x 16 B; lane (n, h),
// element q = tap (ky = 2h + (q >> 2), kx = q & 3)
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_pack_h2_kernel(const float *__restrict__ w, u32x4 *__restrict__ img, int Cout,
                                                              int ntile, const int *__restrict__ hdr) {
    const int *kwtab = hdr + 64 + 32 * ntile;
    const int total = ntile * CIN * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int lane = e & 63, t = e >> 6, ci = t % CIN, n = t / CIN;
        const int co = n * 32 + (lane & 31), hh = lane >> 5;
        const float sc = __builtin_ldexpf(1.0f, kwtab[co]);
        float v[8];
        for (int q = 0; q < 8; ++q) v[q] = co < Cout ? w[((co * CIN + ci) * 4 + 2 * hh + (q >> 2)) * 4 + (q & 3)] : 0.0f;
        u32x4 t1, t2;
        split8_h(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, sc, t1, t2);
        img[(size_t)((n * CIN + ci) * 2) * 64 + lane] = t1;
        img[(size_t)((n * CIN + ci) * 2 + 1) * 64 + lane] = t2;
    }
}

// ---------------------------------------------------------------------------
// Decoder tail in one launch (models/decoder.py:31-35): ConvTranspose2d(128 -> 64, 4x4 s2 p1) + ReLU +
// ConvTranspose2d(64 -> 3, 4x4 s2 p1) on 8x8 maps -> 32x32 NCHW images; the 16x16x64 map between them (64 KiB per image)
// never exists.  One wave owns one image and walks the four output phases (py, px) of the first layer, two per pass (the
// phases (py, 0) and (py, 1) share the parked input planes); everything is computed transposed as in conv_res_pair8_h2_kernel.
//   layer 1, phase (py, px): u[2y + py][2x + px][c] = relu(b + sum over 2x2 taps and 128 channels), as in
//       conv_tile8_bf3_kernel (4 chunks of 32 channels parked as fp16 planes, tap = shifted plane read), weights streamed
//       by LDS-DMA two taps per stage; the accumulator (lane = block pixel, registers = channels) becomes the second
//       layer's B operands by half-wave swaps (acc_to_ksteps), scaled by the phase tile's own maximum;
//   layer 2 in its GEMM + col2im form: T[co * 16 + tap][pixel] = sum_c w4[c][co][tap] u[pixel][c] (48 rows = two A tiles),
//       and out[co][4y + 2py - 1 + ky][4x + 2px - 1 + kx] += T: every output element receives exactly ONE term per phase.
//       The two phases of a pass are combined in REGISTERS: a lane holds its pixel's terms of both phases for the output
//       rows 4y + 2py - 1 + h and + 2; its own 16-byte quad of such a row (columns 4x .. 4x + 3) is its six inner terms plus one
//       term each of its left and right pixel, fetched by DPP row shifts -- eight lanes then store one whole 128-byte output
//       row straight from registers: out = bias + quad in pass 0, out += quad in pass 1 with 16-byte read-modify-writes of
//       the wave's own 12 KiB image (L2-resident; plain accesses by the wave that wrote them, its write-through L1 keeps no
//       stale copy, and a pass's stores are complete -- sixteen s_waitcnt vmcnt(0) later -- before the next pass's loads are
//       issued): a fixed summation order, no atomics, no accumulation tile.  (Round 2 scattered the terms into a zeroed
//       wave-private 34 x 42 LDS tile per channel and read it back: 420 LDS operations per image, 28 % of a wave's time;
//       scattered 4-byte read-modify-writes straight from the accumulator layout were measured first: 570 us instead of 330
//       for the two separate kernels -- L2 request bound.)
struct TailGeom {
    unsigned long long dym[4], dxm[4];             // 4 bits per tap: dy + 8, dx + 8 (ConvGeom) of each phase
};

#ifndef DT_MINW
#define DT_MINW 2
#endif
__global__ __launch_bounds__(256, DT_MINW) void dec_tail8_h2_kernel(const float *__restrict__ in, const u32x4 *__restrict__ w2img,
                                                              const int *__restrict__ hdr2, const float *__restrict__ bias2,
                                                              TailGeom tg, const u32x4 *__restrict__ w4img,
                                                              const int *__restrict__ hdr4, const float *__restrict__ bias4,
                                                              float *__restrict__ out, int B, const int *__restrict__ in_amax) {
    constexpr int NT = 2, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2, CIN = 128, CPT = CIN / 32, CO = 3;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int WBUF = 16 * 64, NSTAGE = 34;             // per pass: 4 chunks x 4 tap pairs (16 pieces each) + the second layer's image
    __shared__ u32x4 As_all[4 * TILE4];
    __shared__ u32x4 Wb_all[2 * WBUF];
    __shared__ __attribute__((aligned(16))) float dw_s[64];      // the first layer's per-output-channel weight scales 2^-kw[c]
    if (threadIdx.x < 64) dw_s[threadIdx.x] = h2_dw(hdr2)[threadIdx.x];      // (stage barriers precede every use)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;

    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // A PASS covers the two phases (py, 0) and (py, 1): they share the parked planes, and where their taps read the same
    // input offset (dx = 0) also the operand reads.  stage k = 17 py + i: i < 16: chunk i >> 2, tap pair i & 3 = (ty, kind) of
    // the first layer -- kind 0: the two phases' dx = 0 taps (tx = 0 of px = 0, tx = 1 of px = 1), kind 1: the other two;
    // pieces 0..7 = phase (py, 0)'s tap, 8..15 = phase (py, 1)'s ([nt][term][k-step] each); i = 16: the second layer's A image
    auto dma_stage = [&](int k, int buf) {
        const int py = k / 17, i = k - 17 * py;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave_u * 4 + j, half = p >> 3;
            const int ty = (i >> 1) & 1, kind = i & 1, tap = ty * 2 + (half ? 1 - kind : kind);
            const u32x4 *src = i == 16 ? w4img + p * 64
                                       : w2img + (size_t)((2 * py + half) * 16 + tap * CPT + (i >> 2)) * 512 + (p & 7) * 64;
            dma(src, Wb_all + buf * WBUF + p * 64);
        }
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    dma_stage(0, 0);
    if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};       // padding pixels of the four planes

    // (built from the wave-uniform image index: with a lane-derived one hipcc wraps every access in a waterfall loop)
    const long long img_u = (long long)blockIdx.x * 4 + wave_u;
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(img_u < B ? img_u : 0) * (CO * 1024), 0, img_u < B ? (unsigned)(CO * 4096) : 0u, 0x00020000);
    int spx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) spx[mt] = 32 * mt + l31;
    const float *src = in + ((size_t)(img_ok ? img : 0) * PX + lane) * CIN;                 // this lane's pixel row
    f32x4 raw[8];
    auto load_raw = [&](int cc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4 *>(src + 32 * cc + 4 * j);
    };
    float m = 0.0f;
    const int given = (in_amax && img_ok) ? in_amax[img] : -1;
    if (given >= 0) m = __int_as_float(given);
    else for (int cc = 0; cc < CPT; ++cc) {
        load_raw(cc);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(raw[j].x), __builtin_fabsf(raw[j].y)), fmaxf(__builtin_fabsf(raw[j].z), __builtin_fabsf(raw[j].w))));
    }
    const int kx = wave_scale_exp(img_ok ? m : 0.0f);
    const float xs = __builtin_ldexpf(1.0f, kx), d1 = __builtin_ldexpf(1.0f, -kx);      // (x the weight rows' 2^-kw[c] at the use)
    load_raw(0);

#pragma unroll 1
    for (int py = 0; py < 2; ++py) {
        unsigned long long dym[2], dxm[2];
        unsigned tapok[2][MT];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            dym[px] = tg.dym[2 * py + px];
            dxm[px] = tg.dxm[2 * py + px];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int y = spx[mt] >> 3, x = spx[mt] & 7;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int yy = y + (int)((dym[px] >> (4 * t)) & 15) - 8, xx = x + (int)((dxm[px] >> (4 * t)) & 15) - 8;
                    if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) mk |= 1u << t;
                }
                tapok[px][mt] = mk;
            }
        }
        f32x16 acc[2][MT][NT];                              // [px][pixel tile][channel tile]
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[px][mt][nt][r] = 0.0f;
        // ----------------------------- layer 1, both phases of the pass -----------------------------
#pragma unroll 1
        for (int cc = 0; cc < CPT; ++cc) {
            __builtin_amdgcn_wave_barrier();
            u32x4 *dst = As + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    u32x4 t1, t2;
                    split8_h(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], xs, t1, t2);
                    dst[(t * 2 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 2 + 1) * PLANE + hh * HP] = t2;
                }
            lds_order_wave();
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int k = py * 17 + cc * 4 + i;
                const int ty = i >> 1, kind = i & 1;
                const int tapA = ty * 2 + kind, tapB = ty * 2 + 1 - kind;     // of phase px = 0 / px = 1
                u32x4 X[2][MT][2];                          // [k-step][pixel tile][term] of the half-stage in flight
                // k-step t of tap `tap` of phase px (a k-step's registers are reloaded for the second half as soon as the first
                // half's groups that read them have been issued)
                auto ldX = [&](int t, int px, int tap) {
                    const int shift = ((int)((dym[px] >> (4 * tap)) & 15) - 8) * 8 + ((int)((dxm[px] >> (4 * tap)) & 15) - 8);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int p = ((tapok[px][mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                        const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                        X[t][mt][0] = ap[0];
                        X[t][mt][1] = ap[PLANE];
                    }
                };
                ldX(0, 0, tapA);
                ldX(1, 0, tapA);
                // this stage's weights are in; everyone is done with the other buffer.  (The next chunk's eight activation loads
                // go out behind stage i = 1's weights and may stay in flight across its wait: see conv_res_pair8_h2_kernel.)
                if (i == 1) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __syncthreads();
                } else dma_wait_sync();
                dma_stage(k + 1, (k + 1) & 1);
                if (i == 0) load_raw(cc + 1 < CPT ? cc + 1 : 0);       // (the next pass starts over at chunk 0)
                const u32x4 *wt = Wb_all + (k & 1) * WBUF + lane;      // [phase of the pair][nt][term][k-step] x 64 units
                u32x4 Wc0 = wt[0], Wc1 = wt[128];
#pragma unroll
                for (int g = 0; g < 8; ++g) {              // group g = (phase of the pair, k-step, nt)
                    const int px = g >> 2, t = (g >> 1) & 1, nt = g & 1;
                    u32x4 Wn0 = Wc0, Wn1 = Wc1;
                    if (g + 1 < 8) {
                        const int g1 = g + 1;
                        const u32x4 *bp = wt + (g1 >> 2) * 512 + (g1 & 1) * 256 + ((g1 >> 1) & 1) * 64;
                        Wn0 = bp[0];
                        Wn1 = bp[128];
                    }
                    if (g == 2) ldX(0, 1, tapB);           // groups 0, 1 (the readers of k-step 0) are behind us
                    if (g == 4) ldX(1, 1, tapB);           // groups 2, 3 likewise
                    __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise sinks the reads to just before their use
                    prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, acc[px][0][nt], acc[px][1][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    Wc0 = Wn0;
                    Wc1 = Wn1;
                }
            }
        }
        // ----------------------------- per phase: bias + ReLU, its scale, T = W4 u -----------------------------
        // row rho = 32 m + (r & 3) + 8 (r >> 2) + 4 h of T is (co = rho >> 4, tap = rho & 15 = ky * 4 + kx): register r of A tile m
        // holds co = 2 m + (r >> 3), ky = h + 2 ((r >> 2) & 1), kx = r & 3; rows >= 48 (m = 1, r >= 8) are padding
        f32x16 T[2][2][MT];                                 // [px][A tile][pixel tile]
        float d4[2];
        const u32x4 *wt4 = nullptr;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            float um = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (bias2) bv = *reinterpret_cast<const f32x4 *>(bias2 + nt * 32 + 8 * g + 4 * h);
                    const f32x4 dv = h2_dw4(dw_s, nt * 32, g, h, d1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2)
                            SCALE2_BIAS_RELU2(acc[px][mt][nt][4 * g + q], acc[px][mt][nt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], um);
                }
            const int ku = wave_scale_exp(img_ok ? um : 0.0f);
            const float us = __builtin_ldexpf(1.0f, ku);
            d4[px] = __builtin_ldexpf(1.0f, -ku);
            u32x4 U1[MT][NT][2], U2[MT][NT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_to_ksteps(acc[px][mt][nt], us, U1[mt][nt], U2[mt][nt]);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[px][m2][mt][r] = 0.0f;
            if (px == 0) {
                const int k = py * 17 + 16;
                dma_wait_sync();
                if (k + 1 < NSTAGE) dma_stage(k + 1, (k + 1) & 1);
                wt4 = Wb_all + (k & 1) * WBUF + lane;       // [m][k-step 4][term] x 64 units
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const u32x4 *bp = wt4 + (m2 * 4 + kk) * 128;
                    prod3x2t(U1[0][kk >> 1][kk & 1], U2[0][kk >> 1][kk & 1], U1[1][kk >> 1][kk & 1], U2[1][kk >> 1][kk & 1], bp[0], bp[64],
                             T[px][m2][0], T[px][m2][1]);
                }
        }
        // ----------------------------- col2im of the pass, in registers -----------------------------
        // Lane (pixel (y, x), half h) holds for every output channel the eight terms of its pixel's two phases for the two
        // output rows oy = 4y + 2py - 1 + h (kernel row ky = h) and oy + 2 (ky = h + 2): a[kx] of phase (py, 0) lands at column
        // 4x - 1 + kx, b[kx] of phase (py, 1) at 4x + 1 + kx.  The lane's own 16-byte quad of a row, columns 4x .. 4x + 3, is
        //     { a1 + b3 of the LEFT pixel,  a2 + b0,  a3 + b1,  b2 + a0 of the RIGHT pixel }
        // -- the two neighbour terms come by DPP row shifts inside the 8-lane pixel row (nothing at the image's left / right
        // edge: those taps fall outside) -- so the eight lanes of a pixel row write one whole 128-byte output row straight
        // from registers: pass 0 stores bias + quad, pass 1 adds to what pass 0 stored (row 31 gets its only term in pass 1).
        // Same terms in the same order as the LDS-tile form this replaces (phase 0's term first, pass 0 first): same bits; no
        // LDS tile to zero, scatter into and read back (420 LDS operations per image), 28 % of a wave's time before.
        const bool xl = (lane & 7) != 0, xr = (lane & 7) != 7;
        unsigned roff[MT][2];                              // byte offset of the lane's quad in rows oy / oy + 2 of channel 0, or out of range
        bool only1[MT][2];                                 // pass 1: the row got nothing in pass 0 (row 31)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int row = 4 * (spx[mt] >> 3) + 2 * py - 1 + h + 2 * g;
                roff[mt][g] = (row >= 0 && row < 32) ? (unsigned)((row * 32 + 4 * (lane & 7)) * 4) : kOobOffset;
                only1[mt][g] = row == 31;
            }
        f32x4 ov[CO][MT][2];
        if (py > 0) {
#pragma unroll
            for (int co = 0; co < CO; ++co)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        ov[co][mt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, roff[mt][g], (unsigned)co * 4096u, 0));
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float bv = bias4 ? bias4[co] : 0.0f;
            const float w4d = h2_dw(hdr4)[co];                  // the output channel's own weight scale 2^-kw4[co] (wave-uniform)
            const float d40 = d4[0] * w4d, d41 = d4[1] * w4d;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x16 &T0 = T[0][co >> 1][mt], &T1 = T[1][co >> 1][mt];
                    const int r0 = 8 * (co & 1) + 4 * g;
                    const float a0 = T0[r0] * d40, b0 = T1[r0] * d41, b1 = T1[r0 + 1] * d41, b2 = T1[r0 + 2] * d41, b3 = T1[r0 + 3] * d41;
                    // neighbours: row_shr:1 hands lane i the value of lane i - 1, row_shl:1 that of lane i + 1 (16-lane rows = two pixel rows)
                    float lb3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b3), 0x111, 0xf, 0xf, true));
                    float ra0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x101, 0xf, 0xf, true));
                    lb3 = xl ? lb3 : 0.0f;
                    ra0 = xr ? ra0 : 0.0f;
                    f32x4 e;
                    e.x = __builtin_fmaf(T0[r0 + 1], d40, lb3);            // (the products by 2^-k are exact: one rounding, as mul + add)
                    e.y = __builtin_fmaf(T0[r0 + 2], d40, b0);
                    e.z = __builtin_fmaf(T0[r0 + 3], d40, b1);
                    e.w = ra0 + b2;
                    f32x4 base = {bv, bv, bv, bv};
                    if (py > 0 && !only1[mt][g]) base = ov[co][mt][g];
                    const f32x4 v = base + e;
                    // (the channel's offset in the VECTOR offset: a scalar-offset store followed by an overwrite of its data registers is the
                    // hazard hipcc leaves unguarded, tools/hazard_scan.py)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, roff[mt][g] == kOobOffset ? kOobOffset : roff[mt][g] + (unsigned)co * 4096u, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};   // the planes' padding pixels again
    }
}

// A-operand image of the last layer's weights * 2^kw for dec_tail8_h2_kernel: [m 2][k-step 4][term 2] x 64 lanes x 16 B;
// lane (row rho - 32 m with rho = co * 16 + tap, h), element q = input channel 32 (k >> 1) + 16 h + 8 (k & 1) + q (acc_to_ksteps' order)
__global__ __launch_bounds__(256) void convt_out_pack_a_kernel(const float *__restrict__ w, u32x4 *__restrict__ img, int Cin, int Cout,
                                                               const int *__restrict__ hdr) {
    const int *kwtab = hdr + 64 + 32;                       // header of one 32-channel tile: kw[co], co < Cout <= 4
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 8 * 64; e += gridDim.x * 256) {
        const int lane = e & 63, kk = (e >> 6) & 3, m2 = e >> 8;
        const int rho = 32 * m2 + (lane & 31), hh = lane >> 5;
        const float sc = __builtin_ldexpf(1.0f, kwtab[rho < 16 * Cout ? (rho >> 4) : 0]);
        float v[8];
        for (int q = 0; q < 8; ++q) {
            const int c = 32 * (kk >> 1) + 16 * hh + 8 * (kk & 1) + q;
            v[q] = (rho < 16 * Cout && c < Cin) ? w[((size_t)c * Cout + (rho >> 4)) * 16 + (rho & 15)] : 0.0f;
        }
        u32x4 t1, t2;
        split8_h(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, sc, t1, t2);
        img[(size_t)((m2 * 4 + kk) * 2) * 64 + lane] = t1;
        img[(size_t)((m2 * 4 + kk) * 2 + 1) * 64 + lane] = t2;
    }
}

// ---------------------------------------------------------------------------
// Fused residual layer (models/residual.py:18-29):
//     y = [relu](u) + W2 (*) relu(W1 (*) [relu](u)),  then optional relu(y)
// W1: 3x3 pad 1, C -> Rh (<= 32), no bias;  W2: 1x1, Rh -> C = 32*NT2, no bias.
// The hidden 32-channel tile goes accumulator -> LDS -> A operand inside the wave.
template <int NT2>
__global__ __launch_bounds__(256, 2) void res_layer_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ w1img,
                                                        const float *__restrict__ w2img,
                                                        float *__restrict__ out, int B, int H, int W,
                                                        int C, int flags) {
    constexpr int MT = 2;
    // LDS: W2 image (shared, read-only after the first barrier) | per-wave hidden tiles
    __shared__ __attribute__((aligned(16))) float smem_res[NT2 * 1024 + 4 * MT * 32 * 33];
    float *W2s = smem_res;
    float(*Hs)[MT][32 * 33] = reinterpret_cast<float(*)[MT][32 * 33]>(smem_res + NT2 * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const long long M = (long long)B * H * W;
    const int cpt = (C + 31) / 32;
    const int nchunk = 9 * cpt;
    const bool relu_in = flags & kFlagReluIn, relu_out = flags & kFlagReluOut;

    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(w2img);
        f32x4 *dst = reinterpret_cast<f32x4 *>(W2s);
#pragma unroll
        for (int q = 0; q < NT2; ++q) dst[tid + 256 * q] = src[tid + 256 * q];
    }

    const long long wbase = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT);
    const long long img_px = (long long)H * W;
    const long long b_first = ((long long)blockIdx.x * (128 * MT)) / img_px;
    const auto in_rs = act_rsrc(in + (size_t)b_first * H * W * C, (unsigned long long)(B - b_first) * H * W * C * 4ull);
    unsigned pbase[MT], tapmask[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = wbase + mt * 32 + l31;
        const bool valid = p < M;
        const long long pc = valid ? p : 0;
        const long long b = pc / img_px;
        const int rem = (int)(pc - b * img_px);
        const int gy = rem / W, gx = rem - gy * W;
        pbase[mt] = (unsigned)((((b - b_first) * H + gy) * W + gx) * C * 4 + 64 * h);
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = gy + t / 3 - 1, ix = gx + t % 3 - 1;
            if (valid && iy >= 0 && iy < H && ix >= 0 && ix < W) m |= 1u << t;
        }
        tapmask[mt] = m;
    }

    // GEMM1 (3x3, C -> 32 hidden): barrier-free.  With a single 32-wide n-tile the weight chunk a
    // wave needs per step is only 4 KiB, so every wave reads its B operands straight from L1/L2
    // (coalesced float4, same image layout) next to its A operands: no LDS staging, no workgroup
    // barrier in the reduction loop, and the waves of a SIMD drift apart instead of stalling together.
    constexpr int KC = 2;
    f32x4 a[KC][MT][4], bq[KC][4];
    f32x16 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;

    const bool ragged_c = (C & 31) != 0;
    const f32x4 *w1v = reinterpret_cast<const f32x4 *>(w1img) + h * 32 + l31;     // + (chunk*4 + j)*64
    auto load_ab = [&](int c, f32x4(&dst)[MT][4], f32x4(&bd)[4]) {
        const int tap = c / cpt, cc = c - tap * cpt;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int tapbytes = (dy * W + dx) * C * 4;                   // scalar
        const unsigned soff = (unsigned)cc * 128u;
#pragma unroll
        for (int j = 0; j < 4; ++j) bd[j] = w1v[(size_t)(c * 4 + j) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned vo = ((tapmask[mt] >> tap) & 1u) ? pbase[mt] + (unsigned)tapbytes : kOobOffset;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dst[mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, vo + 16 * j, soff, 0));
        }
    };
    const bool needs_fix = relu_in || ragged_c;
    auto fix_a = [&](int c, f32x4(&dst)[MT][4]) {
        const int cc = c % cpt;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = dst[mt][j];
                if (cc * 32 + 16 * h + 4 * j >= C) v = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                dst[mt][j] = relu_in ? relu4(v) : v;
            }
    };

#pragma unroll
    for (int k = 0; k < KC; ++k)
        if (k < nchunk) load_ab(k, a[k], bq[k]);
    for (int c0 = 0; c0 < nchunk; c0 += KC) {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (c0 + k < nchunk) {
                if (needs_fix) fix_a(c0 + k, a[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][mt][j][i], bq[k][j][i], acc1[mt], 0, 0, 0);
                if (c0 + k + KC < nchunk) load_ab(c0 + k + KC, a[k], bq[k]);
            }
        }
    }
    __syncthreads();          // W2 image (copied at kernel start) is complete

    // hidden tile: relu, accumulator layout -> [pixel][hidden] in LDS (stride 33: conflict-free)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
            Hs[wave][mt][prow * 33 + l31] = fmaxf(acc1[mt][r], 0.0f);
        }
    lds_order_wave();
    float a2[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 16; ++s) a2[mt][s] = Hs[wave][mt][l31 * 33 + 16 * h + s];

    // second GEMM in groups of <= 2 n-tiles so the accumulators stay at 64 registers
    constexpr int NG = NT2 < 2 ? NT2 : 2;
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(W2s);
#pragma unroll
    for (int n0 = 0; n0 < NT2; n0 += NG) {
        f32x16 acc2[MT][NG];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NG; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][nt][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4[NG];
#pragma unroll
            for (int nt = 0; nt < NG; ++nt) b4[nt] = ws[(((n0 + nt) * 4 + j) * 2 + h) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NG; ++nt)
                        acc2[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[mt][4 * j + i], b4[nt][i],
                                                                            acc2[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long prow = wbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (prow < M) {
#pragma unroll
                    for (int nt = 0; nt < NG; ++nt) {
                        const int n = (n0 + nt) * 32 + l31;
                        if (n < C) {
                            float u = in[prow * C + n];
                            if (relu_in) u = fmaxf(u, 0.0f);
                            float v = u + acc2[mt][nt][r];
                            if (relu_out) v = fmaxf(v, 0.0f);
                            out[prow * C + n] = v;
                        }
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------
// First conv: nn.Conv2d(CIN, Cout, k=4, s=2, p=1) on the NCHW image, row-major out
// (models/encoder.py:29-31).  Reduction slot s = (ci*4 + ky)*2 + kxl with kx = 2h + kxl,
// so the two lane halves differ only by a +2 column offset in their gathers.
template <int CIN, int NT>
__global__ __launch_bounds__(256) void conv_in_kernel(const float *__restrict__ x,
                                                      const float *__restrict__ wimg,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ out, int B, int H, int W,
                                                      int Cout, int flags) {
    constexpr int MT = 2, S = CIN * 8, JG = (S + 3) / 4;
    __shared__ __attribute__((aligned(16))) float Ws[NT * JG * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int Hg = H / 2, Wg = W / 2;
    const long long M = (long long)B * Hg * Wg;
    for (int i = tid; i < NT * JG * 64; i += 256)
        reinterpret_cast<f32x4 *>(Ws)[i] = reinterpret_cast<const f32x4 *>(wimg)[i];

    float a[MT][JG * 4];
    const long long wbase = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = wbase + mt * 32 + l31;
        const bool valid = p < M;
        const long long pc = valid ? p : 0;
        const long long b = pc / ((long long)Hg * Wg);
        const int rem = (int)(pc - b * Hg * Wg);
        const int gy = rem / Wg, gx = rem - gy * Wg;
#pragma unroll
        for (int s = 0; s < JG * 4; ++s) {
            float v = 0.0f;
            if (s < S) {
                const int ci = s >> 3, ky = (s >> 1) & 3, kxl = s & 1;
                const int iy = 2 * gy - 1 + ky, ix = 2 * gx - 1 + 2 * h + kxl;
                if (valid && iy >= 0 && iy < H && ix >= 0 && ix < W)
                    v = x[((b * CIN + ci) * H + iy) * (long long)W + ix];
            }
            a[mt][s] = v;
        }
    }
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    const f32x4 *ws = reinterpret_cast<const f32x4 *>(Ws);
#pragma unroll
    for (int j = 0; j < JG; ++j) {
        f32x4 b4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b4[nt] = ws[((nt * JG + j) * 2 + h) * 32 + l31];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][4 * j + i], b4[nt][i],
                                                                       acc[mt][nt], 0, 0, 0);
    }
    const bool relu_out = flags & kFlagReluOut;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long prow = wbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (prow < M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = nt * 32 + l31;
                    if (n < Cout) {
                        float v = acc[mt][nt][r] + (bias ? bias[n] : 0.0f);
                        if (relu_out) v = fmaxf(v, 0.0f);
                        out[prow * Cout + n] = v;
                    }
                }
            }
        }
}

// Same layer when a workgroup's 256 output pixels are an R x TW TILE of one image's output grid (TW a power of two that
// divides Wg, R = 256 / TW rows that divide Hg: whole rows on 32x32 and 256x256 images, 16 x 16 tiles on 224x224): the
// (2R+2) x (2TW+2) input patch the tile needs is staged once in LDS with coalesced 16-byte loads (zero outside the image),
// and the 8*CIN gathers per pixel become LDS reads without
// bounds checks (the plain kernel issues them as predicated 4-byte global loads).
// BF3: the 8*CIN-deep reduction runs as CIN k-steps of exact three-term bf16 splits on the bf16 matrix cores
// (weights split at pack time: [n_tile][k-step][term][half][n] x 16 B; the gathered pixels are split in
// registers) instead of 4*CIN fp32 MFMAs -- 2.7x less matrix time, which is what this otherwise memory-bound
// layer was waiting on.
template <int CIN, int NT, bool BF3>
__global__ __launch_bounds__(256, 3) void conv_in_rows_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ wimg,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out, int B, int H, int W,
                                                           int Cout, int flags, int *__restrict__ out_amax, int tw_log2,
                                                           const float *__restrict__ ep_mask) {
    // ep_mask (row-major like out, or NULL): out = ep_mask > 0 ? conv : 0 -- the last layer's data gradient with the ReLU mask
    // of the layer below (vqvae_conv_in_forward_ep_f32)
    constexpr int MT = 2, S = CIN * 8, JG = (S + 3) / 4;
    constexpr int WF = BF3 ? NT * CIN * 768 : NT * JG * 256;     // floats of the weight image
    extern __shared__ __attribute__((aligned(16))) float smem_ci[];
    float *Ws = smem_ci;                                   // [WF]
    float *Xs = smem_ci + WF;                              // [CIN][2R + 2][2TW + 8], input column ix at 4 + ix - ix0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int Hg = H / 2, Wg = W / 2;
    const int TW = 1 << tw_log2, R = 256 >> tw_log2, NR = 2 * R + 2, XS = 2 * TW + 8;
    const int ntx = Wg >> tw_log2, nty = Hg / R;
    const long long band = blockIdx.x;                     // one R x TW tile of output pixels
    const long long b = band / (nty * ntx);
    const int trem = (int)(band - b * (nty * ntx));
    const int gy0 = (trem / ntx) * R, gx0 = (trem % ntx) << tw_log2;
    const int iy0 = 2 * gy0 - 1, ix0 = 2 * gx0;
    for (int i = tid; i < WF / 4; i += 256)
        reinterpret_cast<f32x4 *>(Ws)[i] = reinterpret_cast<const f32x4 *>(wimg)[i];
    const int w4 = XS / 4;                                 // 16-byte groups ix0 - 4 + 4 x4 ... of a patch row (ix0 % 4 == 0)
    for (int i = tid; i < CIN * NR * w4; i += 256) {
        const int x4 = i % w4, q = i / w4;
        const int r = q % NR, ci = q / NR;
        const int iy = iy0 + r, ix = ix0 - 4 + 4 * x4;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(x + ((b * CIN + ci) * H + iy) * (long long)W + ix);
        *reinterpret_cast<f32x4 *>(Xs + (ci * NR + r) * XS + 4 * x4) = v;
    }
    __syncthreads();

    float a[MT][JG * 4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = wave * (32 * MT) + mt * 32 + l31;    // pixel within the tile
        const int ly = p >> tw_log2, gx = p & (TW - 1);
        const float *base = Xs + (2 * ly) * XS + 2 * gx + 2 * h + 3;
#pragma unroll
        for (int s = 0; s < JG * 4; ++s) {
            float v = 0.0f;
            if (s < S) {
                const int ci = s >> 3, ky = (s >> 1) & 3, kxl = s & 1;
                v = base[(ci * NR + ky) * XS + kxl];
            }
            a[mt][s] = v;
        }
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    if constexpr (BF3) {
        const u32x4 *wb = reinterpret_cast<const u32x4 *>(Ws);
#pragma unroll
        for (int t = 0; t < CIN; ++t) {                    // k-step t: this lane half's values 8t .. 8t+7
            u32x4 s1[MT], s2[MT], s3[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                split8(f32x4{a[mt][8 * t], a[mt][8 * t + 1], a[mt][8 * t + 2], a[mt][8 * t + 3]},
                       f32x4{a[mt][8 * t + 4], a[mt][8 * t + 5], a[mt][8 * t + 6], a[mt][8 * t + 7]}, s1[mt], s2[mt],
                       s3[mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const u32x4 *bp = wb + ((nt * CIN + t) * 3) * 64 + h * 32 + l31;
                prod6x2(s1[0], s2[0], s3[0], s1[1], s2[1], s3[1], bp[0], bp[64], bp[128], acc[0][nt], acc[1][nt]);
            }
        }
    } else {
        const f32x4 *ws = reinterpret_cast<const f32x4 *>(Ws);
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            f32x4 b4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b4[nt] = ws[((nt * JG + j) * 2 + h) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][4 * j + i], b4[nt][i],
                                                                           acc[mt][nt], 0, 0, 0);
        }
    }
    const bool relu_out = flags & kFlagReluOut;
    // output pixel (row-major NHWC) of the tile's pixel p
    auto opix = [&](int p) { return (b * Hg + gy0 + (p >> tw_log2)) * (long long)Wg + gx0 + (p & (TW - 1)); };
    const int wbase = wave * (32 * MT);
    float omax = 0.0f;
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias && nt * 32 + l31 < Cout) ? bias[nt * 32 + l31] : 0.0f;
    if ((Cout & 7) == 0) {
        __syncthreads();                                   // every wave is done with Ws / Xs: reuse them as output tiles
        float *tile = smem_ci + wave * (32 * 36);
        // the image's base is scalar, this lane's eight output pixels (two pixel tiles x four row groups of the staged tile)
        // are byte offsets inside the image: no address arithmetic per store
        float *obase = out + (size_t)b * Hg * Wg * Cout;
        const float *mbase = ep_mask ? ep_mask + (size_t)b * Hg * Wg * Cout : nullptr;
        unsigned ooff[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = wbase + mt * 32 + (lane >> 3) + 8 * k;
                ooff[mt][k] = (unsigned)((((gy0 + (p >> tw_log2)) * Wg + gx0 + (p & (TW - 1))) * Cout + 4 * (lane & 7)) * 4);
            }
        auto finish = [&](auto RO) {                       // (one straight-line copy per ReLU flag)
            constexpr bool ro = decltype(RO)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bool nok = nt * 32 + 4 * (lane & 7) < Cout, cok = nt * 32 + l31 < Cout;
                    const f32x2v b2 = {bv[nt], bv[nt]};
                    f32x4 mk[4];                           // (ep_mask) requested before the staging: in flight under it
                    if (mbase && nok) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            mk[k] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(mbase + nt * 32) + ooff[mt][k]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2v y = f32x2v{acc[mt][nt][r], acc[mt][nt][r + 1]} + b2;
                        float v0 = y.x, v1 = y.y;
                        if (ro) { v0 = vmax(v0, 0.0f); v1 = vmax(v1, 0.0f); }
                        if (cok) vmax3_abs(omax, v0, v1);
                        tile[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = v0;
                        tile[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * h) * 32 + l31] = v1;
                    }
                    lds_order_wave();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        f32x4 q = *reinterpret_cast<const f32x4 *>(tile + k * 256 + lane * 4);
                        if (mbase && nok) {
                            const f32x4 m = mk[k];
                            q.x = m.x > 0.0f ? q.x : 0.0f; q.y = m.y > 0.0f ? q.y : 0.0f; q.z = m.z > 0.0f ? q.z : 0.0f; q.w = m.w > 0.0f ? q.w : 0.0f;
                        }
                        if (nok) *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(obase + nt * 32) + ooff[mt][k]) = q;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
        };
        if (relu_out) finish(std::true_type{});
        else finish(std::false_type{});
        if (out_amax) publish_amax(out_amax, b, omax, lane);       // the band's 256 pixels belong to image b
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long prow = opix(wbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = nt * 32 + l31;
                if (n < Cout) {
                    float v = acc[mt][nt][r] + bv[nt];
                    if (relu_out) v = fmaxf(v, 0.0f);
                    omax = fmaxf(omax, __builtin_fabsf(v));
                    if (ep_mask) v = ep_mask[prow * Cout + n] > 0.0f ? v : 0.0f;
                    out[prow * Cout + n] = v;
                }
            }
        }
    if (out_amax) publish_amax(out_amax, b, omax, lane);
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_in_pack_kernel(const float *__restrict__ w, float *__restrict__ img,
                                                           int Cout, int ntile) {
    constexpr int S = CIN * 8, JG = (S + 3) / 4;
    const int total = ntile * JG * 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e & 3, n = (e >> 2) & 31, h = (e >> 7) & 1;
        const int t = e >> 8, j = t % JG, nt = t / JG;
        const int s = 4 * j + i, co = nt * 32 + n;
        float v = 0.0f;
        if (s < S && co < Cout) {
            const int ci = s >> 3, ky = (s >> 1) & 3, kx = 2 * h + (s & 1);
            v = w[((co * CIN + ci) * 4 + ky) * 4 + kx];
        }
        img[e] = v;
    }
}

// split-bf16 image of the first layer's weights: [n_tile][k-step CIN][term 3][half 2][n 32] x 8 bf16; k-step t, half hh,
// slot i holds reduction index s = 8t + i of that half, i.e. (ci = t, ky = i >> 1, kx = 2 hh + (i & 1))
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_pack_bf3_kernel(const float *__restrict__ w, unsigned short *__restrict__ img,
                                                               int Cout, int ntile) {
    const int total = ntile * CIN * 512;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e & 7, n = (e >> 3) & 31, hh = (e >> 8) & 1;
        const int r = e >> 9, t = r % CIN, nt = r / CIN;
        const int co = nt * 32 + n, ky = i >> 1, kx = 2 * hh + (i & 1);
        const float v = co < Cout ? w[((co * CIN + t) * 4 + ky) * 4 + kx] : 0.0f;
        const unsigned short b1 = f32_to_bf16_rne(v);
        const float r1 = v - __uint_as_float((unsigned)b1 << 16);
        const unsigned short b2 = f32_to_bf16_rne(r1);
        const float r2 = r1 - __uint_as_float((unsigned)b2 << 16);
        const unsigned short b3 = f32_to_bf16_rne(r2);
        const size_t base = (size_t)((nt * CIN + t) * 3) * 512 + (size_t)(hh * 32 + n) * 8 + i;
        img[base] = b1;
        img[base + 512] = b2;
        img[base + 1024] = b3;
    }
}

// ---------------------------------------------------------------------------
// Last layer: nn.ConvTranspose2d(Cin, Cout<=4, k=4, s=2, p=1), row-major in, NCHW image out
// (models/decoder.py:34-35).  Cout = 3 cannot fill a 32-wide MFMA tile as an output-channel
// dimension, so the layer runs in its GEMM + col2im form inside ONE kernel:
//   T[pixel][tap*Cout + co] = sum_ci x[pixel][ci] * w[ci][co][tap]     (N = 16*Cout <= 64 on the MFMA)
//   out[co][oy][ox] = bias[co] + sum over the 4 (ky,kx) with matching parity of T[(oy+1-ky)/2][(ox+1-kx)/2][ky][kx][co]
// A workgroup owns a 16x16 region of input pixels (a 14x14 interior + 1-pixel halo, or the whole
// image when it is at most 16 wide/high), keeps T for the region in LDS and writes the interior's
// 2x upsampled outputs with coalesced NCHW stores.
// BF3: products from exact three-term bf16 splits on the bf16 matrix cores (weights split at pack time, image
// [chunk][n_tile][term][k-step][half][n] x 16 B; activations split in registers) instead of the fp32 MFMA.
// MODE 0: exact fp32 MFMA, 1: three-term bf16 products, 2: two-term fp16 products (in_amax: the images' maxima, whdr: {kw})
template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void convt_out_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ wimg,
                                                        const float *__restrict__ bias,
                                                        float *__restrict__ out, int B, int H, int W,
                                                        int Cin, int Cout, int TH, int TW, int halo_y,
                                                        int halo_x, int tiles_y, int tiles_x, const int *__restrict__ whdr,
                                                        const int *__restrict__ in_amax, int ntiles) {
    constexpr int MT = 2;
    constexpr bool BF3 = MODE == 1, H2 = MODE == 2;
    const int STRIDE = 16 * Cout + 1;              // T row: the 16*Cout used columns (odd stride: conflict-free)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cpt = (Cin + 31) / 32;
    constexpr int WCH = BF3 ? 1536 : 1024;          // floats per (chunk, n-tile) of the weight image (fp32: 1024 values, fp16: 2 x 1024 halves)
    // H2: the 16 KiB weight image is read straight from L1 / L2 (every workgroup reads the same bytes), which leaves 50 KiB of
    // LDS per workgroup -> three workgroups per CU instead of two
    float *Ws = smem;                               // [cpt][NT][WCH]
    float *Ts = H2 ? smem : smem + (size_t)cpt * NT * WCH;      // [256][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;

    // PERSISTENT workgroups: tile blockIdx.x, + gridDim.x, ...; the next tile's input (both 32-channel chunks where there are
    // two) is requested before the current tile's col2im, which has no global loads of its own -- the load latency that every
    // one-tile workgroup used to sit out in front of its first MFMA now runs under the col2im of the tile before
    long long b = 0;
    int y0 = 0, x0 = 0, ry = 0, rx = 0;

    if constexpr (!H2)
        for (int i = tid; i < cpt * NT * (WCH / 4); i += 256)
            reinterpret_cast<f32x4 *>(Ws)[i] = reinterpret_cast<const f32x4 *>(wimg)[i];

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    // this lane's input rows as byte offsets into the tile's IMAGE (descriptor per image: one image is below 2 GiB, the
    // tensor need not be); pixels outside the image get kOobOffset and read as zero -- no branch around any load
    unsigned aoff[MT];
    __amdgpu_buffer_rsrc_t img_rs = act_rsrc(in, 0);
    const unsigned long long img_bytes = (unsigned long long)H * W * Cin * 4;
    // tile t: its image / origin (kept by the caller where the previous tile's are still needed) and this lane's input rows
    auto setup = [&](int t, long long &tb, int &ty0, int &tx0) {
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        tb = t / tiles_y;
        ty0 = ty * TH; tx0 = tx * TW;
        img_rs = act_rsrc(in + (size_t)tb * H * W * Cin, img_bytes);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = wave * 64 + mt * 32 + l31;
            const int iy = ty0 - halo_y + (p >> 4), ix = tx0 - halo_x + (p & 15);
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            aoff[mt] = ok ? (unsigned)((iy * W + ix) * Cin + 16 * h) * 4u : kOobOffset;
        }
    };
    // A operands: chunk c+1 is in flight while chunk c multiplies (two register sets); chunk 0 is requested
    // before the barrier so its latency overlaps the weight copy
    auto load_a = [&](int c, f32x4(&a)[MT][4]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned vo = c * 32 + 16 * h + 4 * j < Cin ? aoff[mt] : kOobOffset;     // channel tail of a partial chunk
                a[mt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, vo + (unsigned)((c * 32 + 4 * j) * 4), 0, 0));
            }
    };
    float xsc = 1.0f, dsc = 1.0f;                   // H2: the image's scale 2^kx and the accumulator scale 2^-(kx + kw)
    auto scales = [&]() {
        if constexpr (H2) {
            const float mx = __int_as_float(in_amax[b]);
            int e = 15;
            if (mx > 0.0f && mx < 3.0e38f) (void)__builtin_frexpf(mx, &e);
            int kx = 15 - e;
            kx = kx > 100 ? 100 : (kx < -100 ? -100 : kx);
            xsc = __builtin_ldexpf(1.0f, kx);
            dsc = __builtin_ldexpf(1.0f, -kx);
        }
    };
    auto mma = [&](int c, const f32x4(&a)[MT][4]) {
        if constexpr (H2) {
            const u32x4 *wb = reinterpret_cast<const u32x4 *>(wimg + (size_t)c * NT * WCH);
            u32x4 S1[MT][2], S2[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                split8_h(a[mt][0], a[mt][1], xsc, S1[mt][0], S2[mt][0]);
                split8_h(a[mt][2], a[mt][3], xsc, S1[mt][1], S2[mt][1]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u32x4 *bp = wb + nt * 256 + (t * 2 + h) * 32 + l31;
                    prod3x2(S1[0][t], S2[0][t], S1[1][t], S2[1][t], bp[0], bp[128], acc[0][nt], acc[1][nt]);
                }
            return;
        }
        if (BF3) {
            const u32x4 *wb = reinterpret_cast<const u32x4 *>(Ws + (size_t)c * NT * WCH);
            u32x4 S1[MT][2], S2[MT][2], S3[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                split8(a[mt][0], a[mt][1], S1[mt][0], S2[mt][0], S3[mt][0]);
                split8(a[mt][2], a[mt][3], S1[mt][1], S2[mt][1], S3[mt][1]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u32x4 *bp = wb + nt * 384 + (t * 2 + h) * 32 + l31;
                    prod6x2(S1[0][t], S2[0][t], S3[0][t], S1[1][t], S2[1][t], S3[1][t], bp[0], bp[128], bp[256], acc[0][nt],
                            acc[1][nt]);
                }
            return;
        }
        const f32x4 *ws = reinterpret_cast<const f32x4 *>(Ws + (size_t)c * NT * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b4[nt] = ws[((nt * 4 + j) * 2 + h) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][j][i], b4[nt][i], acc[mt][nt],
                                                                           0, 0, 0);
        }
    };
    f32x4 a0[MT][4], a1[MT][4];
    int tcur = blockIdx.x;
    setup(tcur, b, y0, x0);
    load_a(0, a0);
    if (cpt > 1) load_a(1, a1);
    __syncthreads();
  for (;;) {
    ry = y0 - halo_y; rx = x0 - halo_x;
    scales();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    // on entry chunks 0 and 1 are in a0 / a1 (requested a tile ago)
    for (int c = 0; c < cpt; c += 2) {
        mma(c, a0);
        if (c + 2 < cpt) load_a(c + 2, a0);
        if (c + 1 < cpt) {
            mma(c + 1, a1);
            if (c + 3 < cpt) load_a(c + 3, a1);
        }
    }
    // (column test outermost: one exec mask per n-tile instead of one branch per store)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        if (nt * 32 + l31 < STRIDE - 1) {
            float *tcol = Ts + (wave * 64 + 4 * h) * STRIDE + nt * 32 + l31;
            // H2: column = tap * Cout + co -> the output channel's own weight scale 2^-kw[co] beside the image's 2^-kx
            const float dcol = H2 ? dsc * h2_dw(whdr)[(nt * 32 + l31) % Cout] : 1.0f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tcol[(mt * 32 + (r & 3) + 8 * (r >> 2)) * STRIDE] = H2 ? acc[mt][nt][r] * dcol : acc[mt][nt][r];
        }
    __syncthreads();
    // the next tile's input goes on its way now
    const int tnext = tcur + (int)gridDim.x;
    const bool more = tnext < ntiles;
    long long nb_ = b;
    int ny0 = y0, nx0 = x0;
    if (more) {
        setup(tnext, nb_, ny0, nx0);
        load_a(0, a0);
        if (cpt > 1) load_a(1, a1);
    }

    // col2im over the interior's outputs, ox fastest (coalesced NCHW rows)
    const int th = min(TH, H - y0), tw = min(TW, W - x0);
    const int OH = 2 * th, OW = 2 * tw, Ho = 2 * H, Wo = 2 * W;
    auto gather = [&](int co, int oy, int ox) -> float {
        float s = bias ? bias[co] : 0.0f;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int ky = ((oy + 1) & 1) + 2 * a2;
            const int iy = (oy + 1 - ky) >> 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int kx = ((ox + 1) & 1) + 2 * b2;
                const int ix = (ox + 1 - kx) >> 1;
                if (ix < 0 || ix >= W) continue;
                s += Ts[((iy - ry) * 16 + (ix - rx)) * STRIDE + (ky * 4 + kx) * Cout + co];
            }
        }
        return s;
    };
    if ((OW & 3) == 0 && (Wo & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        // four consecutive ox per thread: one 16-byte store per quad.  The quad (ox = 4 xq .. 4 xq + 3, ox even first)
        // reads input columns ixc - 1 .. ixc + 2 of two input rows; per output the taps are added in gather()'s
        // order (ky, then kx), with the index arithmetic hoisted out of the sixteen LDS reads.
        // thread -> (output row oyl = tid / 8 < OH <= 32, quad xq = tid % 8 < OW / 4 <= 8), channels in a loop: no division
        // per quad, and everything but the channel offset is worked out once per thread
        const int qw = OW >> 2;
        const int xq = tid & 7, oyl = tid >> 3;
        if (xq < qw && oyl < OH) {
            const int oy = 2 * y0 + oyl, ox = 2 * x0 + 4 * xq;
            const int ky0 = (oy + 1) & 1;
            const int iyA = (oy + 1 - ky0) >> 1, iyB = iyA - 1;            // rows of ky = ky0 and ky0 + 2
            const bool vA = iyA < H, vB = iyB >= 0;
            const float *TA0 = Ts + ((vA ? iyA - ry : 0) * 16 - rx) * STRIDE + (ky0 * 4) * Cout;
            const float *TB0 = Ts + ((vB ? iyB - ry : 0) * 16 - rx) * STRIDE + ((ky0 + 2) * 4) * Cout;
            const int ixc = ox >> 1;
            const bool vm = ixc - 1 >= 0, v1 = ixc + 1 < W, v2 = ixc + 2 < W;
            const int om = (vm ? ixc - 1 : ixc) * STRIDE, o0 = ixc * STRIDE, o1 = (v1 ? ixc + 1 : ixc) * STRIDE,
                      o2 = (v2 ? ixc + 2 : ixc) * STRIDE;
            float *orow = out + (b * Cout * Ho + oy) * (long long)Wo + ox;
            for (int co = 0; co < Cout; ++co) {
                const float bsv = bias ? bias[co] : 0.0f;
                const float *TA = TA0 + co, *TB = TB0 + co;
                // one output: row A taps (kx0 at column ca, kx0 + 2 at column cb), then row B taps.  Every address is inside
                // T -- rows and columns are clamped above -- so the sixteen reads are unconditional and a term outside the
                // image enters as + 0.0f: no branch per read
                auto one = [&](int kx0, int ca, bool va, int cb, bool vb) -> float {
                    const float a0 = TA[ca + kx0 * Cout], a1 = TA[cb + (kx0 + 2) * Cout];
                    const float b0 = TB[ca + kx0 * Cout], b1 = TB[cb + (kx0 + 2) * Cout];
                    float acc = bsv;
                    acc += vA && va ? a0 : 0.0f;
                    acc += vA && vb ? a1 : 0.0f;
                    acc += vB && va ? b0 : 0.0f;
                    acc += vB && vb ? b1 : 0.0f;
                    return acc;
                };
                f32x4 v;
                v.x = one(1, o0, true, om, vm);
                v.y = one(0, o1, v1, o0, true);
                v.z = one(1, o1, v1, o0, true);
                v.w = one(0, o2, v2, o1, v1);
                *reinterpret_cast<f32x4 *>(orow + (long long)co * Ho * Wo) = v;
            }
        }
    } else {
        const int total = Cout * OH * OW;
        for (int e = tid; e < total; e += 256) {
            const int oxl = e % OW;
            const int q = e / OW;
            const int oyl = q % OH, co = q / OH;
            const int oy = 2 * y0 + oyl, ox = 2 * x0 + oxl;
            out[((b * Cout + co) * Ho + oy) * (long long)Wo + ox] = gather(co, oy, ox);
        }
    }
    if (!more) break;
    __syncthreads();                               // everyone is done with T
    tcur = tnext; b = nb_; y0 = ny0; x0 = nx0;
  }
}

__global__ __launch_bounds__(256) void convt_out_pack_kernel(const float *__restrict__ w, float *__restrict__ img,
                                                             int Cin, int Cout, int ntile) {
    // w: (Cin, Cout, 4, 4) -> B-operand image [chunk][ntile][4][2][32][4], column n = tap*Cout + co
    const int cpt = (Cin + 31) / 32;
    const int total = cpt * ntile * 1024;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e & 3, n = (e >> 2) & 31, h = (e >> 7) & 1, j = (e >> 8) & 3;
        const int t = e >> 10;
        const int nt = t % ntile, chunk = t / ntile;
        const int ci = chunk * 32 + 16 * h + 4 * j + i, col = nt * 32 + n;
        const int tap = col / Cout, co = col - tap * Cout;
        img[e] = (ci < Cin && tap < 16) ? w[((size_t)ci * Cout + co) * 16 + tap] : 0.0f;
    }
}

// split-bf16 image of the same weights: [chunk][n_tile][term 3][k-step 2][half 2][n 32] x 8 bf16 (cf. conv_pack_bf3)
template <bool H2>
__global__ __launch_bounds__(256) void convt_out_pack_bf3_kernel(const float *__restrict__ w, unsigned short *__restrict__ img,
                                                                 int Cin, int Cout, int ntile, const int *__restrict__ hdr) {
    const int *kwtab = H2 ? hdr + 64 + 32 : nullptr;       // kw[co] of the (single-tile) header, co < Cout <= 4
    const int cpt = (Cin + 31) / 32;
    const int total = cpt * ntile * 1024;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int i = e & 7, n = (e >> 3) & 31, hh = (e >> 8) & 1, t = (e >> 9) & 1;
        const int r = e >> 10;
        const int nt = r % ntile, chunk = r / ntile;
        const int ci = chunk * 32 + 16 * hh + 8 * t + i, col = nt * 32 + n;
        const int tap = col / Cout, co = col - tap * Cout;
        const float v = (ci < Cin && tap < 16) ? w[((size_t)ci * Cout + co) * 16 + tap] : 0.0f;
        const size_t pos = (size_t)((t * 2 + hh) * 32 + n) * 8 + i;
        if (H2) {
            const float vs = v * __builtin_ldexpf(1.0f, kwtab[co]);
            const _Float16 g1 = (_Float16)vs;
            const _Float16 g2 = (_Float16)(vs - (float)g1);
            const size_t base = (size_t)(chunk * ntile + nt) * 2048;
            img[base + pos] = __builtin_bit_cast(unsigned short, g1);
            img[base + 1024 + pos] = __builtin_bit_cast(unsigned short, g2);
            continue;
        }
        const unsigned short b1 = f32_to_bf16_rne(v);
        const float r1 = v - __uint_as_float((unsigned)b1 << 16);
        const unsigned short b2 = f32_to_bf16_rne(r1);
        const float r2 = r1 - __uint_as_float((unsigned)b2 << 16);
        const unsigned short b3 = f32_to_bf16_rne(r2);
        const size_t base = (size_t)(chunk * ntile + nt) * 3072;
        img[base + pos] = b1;
        img[base + 1024 + pos] = b2;
        img[base + 2048 + pos] = b3;
    }
}

// ---------------------------------------------------------------------------
// The skip connection of a residual layer whose width the fused kernels do not cover (round 4: C not in {32, 64, 128} or
// more than 32 hidden channels -- main.py's --n_hiddens / --n_residual_hiddens are free parameters): y = r(x) + t, r = ReLU if
// relu_in (the in-place nn.ReLU(True) of residual.py:19 also rewrites the skip), then ReLU if relu_out.  t may alias y.
__global__ __launch_bounds__(256) void res_combine_kernel(const float *__restrict__ x, const float *t, float *y, long long n4,
                                                          int relu_in, int relu_out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 a = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 b = reinterpret_cast<const f32x4 *>(t)[i];
        if (relu_in) a = relu4(a);
        f32x4 o = a + b;
        if (relu_out) o = relu4(o);
        reinterpret_cast<f32x4 *>(y)[i] = o;
    }
}

// ---------------------------------------------------------------------------
// Batched 2-D transpose in[b][R][Cc] -> out[b][Cc][R] (NCHW <-> row-major at module boundaries).
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        int R, int Cc) {
    __shared__ float tile[32][33];
    const long long b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *src = in + b * (long long)R * Cc;
    float *dst = out + b * (long long)R * Cc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + 8 * k][tx] = src[(long long)r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < R && c < Cc) dst[(long long)c * R + r] = tile[tx][ty + 8 * k];
    }
}

// ---------------------------------------------------------------------------
// kind VQVAE_CONV_TAPS: a stride-1 convolution over an explicit tap list (the masked convolutions of the GatedPixelCNN prior,
// pixelcnn/models.py:45-58, without an im2col pass): the list is handed to make_geom by the vqvae_conv_taps_* entry points
// through this thread-local slot for the duration of the call.
struct TapSpec {
    int n;
    signed char dy[16], dx[16];
};
static thread_local const TapSpec *t_taps = nullptr;
struct TapScope {
    explicit TapScope(const TapSpec *t) { t_taps = t; }
    ~TapScope() { t_taps = nullptr; }
};

static int make_geom(int kind, long long B, int H, int W, int Cin, int Cout, int flags, ConvGeom &g) {
    memset(&g, 0, sizeof(g));
    g.B = (int)B; g.Hin = H; g.Win = W; g.Cin = Cin; g.Cout = Cout; g.flags = flags;
    g.cpt = (Cin + 31) / 32; g.ntile = (Cout + 31) / 32;
    g.nphase = 1; g.istride = 1; g.ostride = 1;
    auto conv_taps = [&](int k, int pad) {
        g.ntaps = k * k; g.kk = k * k;
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                g.dy[0][ky * k + kx] = (signed char)(ky - pad);
                g.dx[0][ky * k + kx] = (signed char)(kx - pad);
                g.kyx[0][ky * k + kx] = (signed char)(ky * k + kx);
            }
    };
    switch (kind) {
        case VQVAE_CONV_4x4_S2:
            if (H % 2 || W % 2) return VQVAE_ERR_UNSUPPORTED;
            conv_taps(4, 1); g.istride = 2; g.Hg = g.Hout = H / 2; g.Wg = g.Wout = W / 2; break;
        case VQVAE_CONV_3x3_S1:
            conv_taps(3, 1); g.Hg = g.Hout = H; g.Wg = g.Wout = W; break;
        case VQVAE_CONV_1x1:
            conv_taps(1, 0); g.Hg = g.Hout = H; g.Wg = g.Wout = W; break;
        case VQVAE_CONV_TAPS:                     // weight (Cout, Cin, n taps): tap t reads the input at (y + dy[t], x + dx[t])
            if (!t_taps || t_taps->n < 1 || t_taps->n > 16) return VQVAE_ERR_UNSUPPORTED;
            g.ntaps = g.kk = t_taps->n;
            for (int t = 0; t < t_taps->n; ++t) {
                g.dy[0][t] = t_taps->dy[t]; g.dx[0][t] = t_taps->dx[t]; g.kyx[0][t] = (signed char)t;
            }
            g.Hg = g.Hout = H; g.Wg = g.Wout = W; break;
        case VQVAE_CONVT_1x1:                     // = the data gradient of a 1x1 nn.Conv2d (weight read transposed)
            conv_taps(1, 0); g.transposed = 1; g.Hg = g.Hout = H; g.Wg = g.Wout = W; break;
        case VQVAE_CONVT_3x3_S1:
            g.transposed = 1; g.ntaps = 9; g.kk = 9;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    g.dy[0][ky * 3 + kx] = (signed char)(1 - ky);
                    g.dx[0][ky * 3 + kx] = (signed char)(1 - kx);
                    g.kyx[0][ky * 3 + kx] = (signed char)(ky * 3 + kx);
                }
            g.Hg = g.Hout = H; g.Wg = g.Wout = W; break;
        case VQVAE_CONVT_4x4_S2: {
            g.transposed = 1; g.ntaps = 4; g.kk = 16; g.nphase = 4; g.ostride = 2;
            g.Hg = H; g.Wg = W; g.Hout = 2 * H; g.Wout = 2 * W;
            // phase parity 0: (k=1,d=0),(k=3,d=-1); parity 1: (k=0,d=+1),(k=2,d=0)
            const int kk[2][2] = {{1, 3}, {0, 2}}, dd[2][2] = {{0, -1}, {1, 0}};
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px) {
                    const int ph = py * 2 + px;
                    g.opy[ph] = (signed char)py; g.opx[ph] = (signed char)px;
                    for (int ty = 0; ty < 2; ++ty)
                        for (int tx = 0; tx < 2; ++tx) {
                            const int t = ty * 2 + tx;
                            g.dy[ph][t] = (signed char)dd[py][ty];
                            g.dx[ph][t] = (signed char)dd[px][tx];
                            g.kyx[ph][t] = (signed char)(kk[py][ty] * 4 + kk[px][tx]);
                        }
                }
            break;
        }
        default: return VQVAE_ERR_UNSUPPORTED;
    }
    for (int ph = 0; ph < g.nphase; ++ph)
        for (int t = 0; t < g.ntaps; ++t) {
            g.dymask[ph] |= (unsigned long long)(g.dy[ph][t] + 8) << (4 * t);
            g.dxmask[ph] |= (unsigned long long)(g.dx[ph][t] + 8) << (4 * t);
        }
    return VQVAE_OK;
}

static size_t packed_floats(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 1024;
}
// split-bf16 image: 3 terms x 32x32 bf16 per (phase, chunk, n_tile) = 6 KiB
static size_t packed_bf3_bytes(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 3072 * sizeof(unsigned short);
}
// two-term fp16 image: 2 terms x 32x32 fp16 per (phase, chunk, n_tile) = 4 KiB, behind the header of conv_wscale_kernel
// (64 ints + one float and one int per output channel of the ntile 32-channel tiles)
static size_t h2_header_bytes(int ntile) { return 256 + (size_t)ntile * 256; }
static size_t packed_h2_bytes(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 2048 * sizeof(unsigned short);
}
// conv_halo8_h2_kernel: stride-1-sampled layers on pixel grids that are multiples of 8 both ways and larger than one tile
static bool conv_halo8_ok(const ConvGeom &g, int Cin, int flags) {
    return !(flags & (VQVAE_CONV_BF16_SPLIT | VQVAE_CONV_EXACT_FP32)) && g.istride == 1 && g.Hg == g.Hin && g.Wg == g.Win &&
           g.Hg % 8 == 0 && g.Wg % 8 == 0 && g.Hg * g.Wg > 64 && Cin % 32 == 0 && g.ntile % 2 == 0 && (g.ntaps == 1 || g.ntaps == 4 || g.ntaps == 9) &&
           (long long)g.Hin * g.Win * Cin * 4 < 0x7FFFFFF0ll;
}
// byte offset of the header from the start of a layer's packed weights
static size_t packed_h2_offset(const ConvGeom &g, int kind) {
    return packed_floats(g) * sizeof(float) + packed_bf3_bytes(g) * (kind == VQVAE_CONV_4x4_S2 ? 2 : 1);
}

}  // namespace vqvae

using namespace vqvae;

extern "C" {

size_t vqvae_conv_packed_bytes(int kind, int Cin, int Cout) {
    ConvGeom g;
    if (Cin < 1 || Cout < 1 || make_geom(kind, 1, 4, 4, Cin, Cout, 0, g) != VQVAE_OK) return 0;
    // [fp32 B-operand image][split-bf16 image][4x4 s2 only: split-bf16 image in space-to-depth chunk order]
    // [header {kw}][two-term fp16 image][4x4 s2 only: the same in space-to-depth chunk order]
    return packed_h2_offset(g, kind) + h2_header_bytes(g.ntile) + packed_h2_bytes(g) * (kind == VQVAE_CONV_4x4_S2 ? 2 : 1);
}

int vqvae_conv_pack_f32(int kind, const float *w, int Cin, int Cout, float *packed, vqvae_stream_t stream) {
    if (!w || !packed) return VQVAE_ERR_NULL;
    if (Cin < 1 || Cout < 1) return VQVAE_ERR_SHAPE;
    ConvGeom g;
    int rc = make_geom(kind, 1, 4, 4, Cin, Cout, 0, g);
    if (rc != VQVAE_OK) return rc;
    const long long total = (long long)packed_floats(g);
    long long grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // [fp32 image][split-bf16 image (1024 bf16 per term per (phase, chunk, n_tile))] ... [header {kw}][two-term fp16 image of w * 2^kw]:
    // the weight scale first, then every image in one launch (4x4 stride 2: one more for the space-to-depth chunk order)
    char *h2 = reinterpret_cast<char *>(packed) + packed_h2_offset(g, kind);
    int *hdr = reinterpret_cast<int *>(h2);
    hipLaunchKernelGGL(conv_wscale_kernel, dim3(32 * g.ntile), dim3(64), 0, st, w, Cin, Cout, g.kk, g.transposed, g.ntile, hdr);
    hipLaunchKernelGGL(conv_pack_images_kernel, dim3((unsigned)grid), dim3(256), 0, st, w, packed,
                       reinterpret_cast<unsigned short *>(packed + total), reinterpret_cast<unsigned short *>(h2 + h2_header_bytes(g.ntile)), g,
                       total, hdr);
    if (kind == VQVAE_CONV_4x4_S2) {
        g.s2d = 1;
        hipLaunchKernelGGL(conv_pack_images_kernel, dim3((unsigned)grid), dim3(256), 0, st, w, (float *)nullptr,
                           reinterpret_cast<unsigned short *>(packed + total) + packed_bf3_bytes(g) / 2,
                           reinterpret_cast<unsigned short *>(h2 + h2_header_bytes(g.ntile) + packed_h2_bytes(g)), g, total, hdr);
    }
    return (int)hipGetLastError();
}

int vqvae_conv_term_products(int kind, int H, int W, int Cin, int Cout, int flags) {
    ConvGeom g;
    if (Cin < 1 || Cout < 1 || H < 1 || W < 1 || make_geom(kind, 1, H, W, Cin, Cout, flags, g) != VQVAE_OK) return 0;
    if (flags & VQVAE_CONV_EXACT_FP32) return 1;
    const bool tile8 = g.Hin == 8 && g.Win == 8 && g.istride == 1 && g.Hg == 8 && g.Wg == 8 && Cin % 32 == 0 && g.ntile % 2 == 0;
    const bool s2d = !tile8 && kind == VQVAE_CONV_4x4_S2 && g.Hin == 16 && g.Win == 16 && Cin % 32 == 0 && g.ntile % 2 == 0;
    // VQVAE_CONV_QUERY_WHOLE_PATH: as launched by vqvae_forward_f32 / _encoder_f32 / _decoder_f32, which hand every layer its
    // images' maxima -- the generic kernels then run the two-term fp16 products on every map size (conv_forward_impl)
    const bool handed = flags & VQVAE_CONV_QUERY_WHOLE_PATH;
    return ((tile8 || s2d || handed) && !(flags & VQVAE_CONV_BF16_SPLIT)) ? 3 : 6;
}

int vqvae_conv_forward_f32(int kind, const float *x, const float *packed, const float *bias, int64_t B,
                           int H, int W, int Cin, int Cout, int flags, float *y, vqvae_stream_t stream) {
    return vqvae::conv_forward_impl(kind, x, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr, nullptr);
}

static int fill_taps(vqvae::TapSpec &t, int ntaps, const int8_t *dy, const int8_t *dx) {
    if (!dy || !dx) return VQVAE_ERR_NULL;
    if (ntaps < 1 || ntaps > 16) return VQVAE_ERR_UNSUPPORTED;
    t.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (dy[i] < -7 || dy[i] > 7 || dx[i] < -7 || dx[i] > 7) return VQVAE_ERR_UNSUPPORTED;     // four bits per tap in the kernels' masks
        t.dy[i] = dy[i]; t.dx[i] = dx[i];
    }
    return VQVAE_OK;
}

size_t vqvae_conv_taps_packed_bytes(int ntaps, int Cin, int Cout) {
    vqvae::TapSpec t;
    t.n = ntaps;
    for (int i = 0; i < 16; ++i) t.dy[i] = t.dx[i] = 0;
    if (ntaps < 1 || ntaps > 16) return 0;
    vqvae::TapScope scope(&t);
    return vqvae_conv_packed_bytes(VQVAE_CONV_TAPS, Cin, Cout);
}

int vqvae_conv_taps_pack_f32(const float *w, int ntaps, const int8_t *dy, const int8_t *dx, int Cin, int Cout, float *packed,
                             vqvae_stream_t stream) {
    vqvae::TapSpec t;
    const int rc = fill_taps(t, ntaps, dy, dx);
    if (rc != VQVAE_OK) return rc;
    vqvae::TapScope scope(&t);
    return vqvae_conv_pack_f32(VQVAE_CONV_TAPS, w, Cin, Cout, packed, stream);
}

int vqvae_conv_taps_forward_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin, int Cout,
                                int ntaps, const int8_t *dy, const int8_t *dx, int flags, float *y, vqvae_stream_t stream) {
    vqvae::TapSpec t;
    const int rc = fill_taps(t, ntaps, dy, dx);
    if (rc != VQVAE_OK) return rc;
    vqvae::TapScope scope(&t);
    return vqvae::conv_forward_impl(VQVAE_CONV_TAPS, x, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr,
                                    nullptr);
}

int vqvae_conv_forward_ep_f32(int kind, const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin,
                              int Cout, int flags, const float *addend, const float *mask, float *y, vqvae_stream_t stream) {
    if ((addend && addend == y) || (mask && mask == y)) return VQVAE_ERR_UNSUPPORTED;     // no in-place form: other waves still read them
    return vqvae::conv_forward_impl(kind, x, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr, nullptr,
                                    addend, mask);
}
}  // extern "C"

// in_amax / out_amax: per-image activation maxima handed from layer to layer inside the whole-path entry points
// (model.hip); NULL from the per-layer C entry points, where the consuming kernel measures its image itself.
int vqvae::conv_forward_impl(int kind, const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                             int Cin, int Cout, int flags, float *y, hipStream_t stream, const int *in_amax, int *out_amax,
                             const float *ep_add, const float *ep_mask) {
    if (!x || !packed || !y) return VQVAE_ERR_NULL;
    if (ep_add || ep_mask) {
        // the data-gradient epilogue lives in the split-product kernels of the per-layer entry (no maxima hand-over, no fp32-MFMA form)
        if (in_amax || out_amax || (flags & VQVAE_CONV_EXACT_FP32)) return VQVAE_ERR_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(ep_add) | reinterpret_cast<uintptr_t>(ep_mask)) & 15) return VQVAE_ERR_UNSUPPORTED;
    }
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return VQVAE_ERR_SHAPE;
    if (Cin % 4) return VQVAE_ERR_UNSUPPORTED;          // float4 activation loads
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return VQVAE_ERR_UNSUPPORTED;   // 16-byte accesses
    if (B * (int64_t)H * W * 4 > (int64_t)INT32_MAX * 4 || B > INT32_MAX) return VQVAE_ERR_OVERFLOW;
    ConvGeom g;
    int rc = make_geom(kind, B, H, W, Cin, Cout, flags, g);
    if (rc != VQVAE_OK) return rc;
    g.ep_add = ep_add;
    g.ep_mask = ep_mask;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = (long long)B * g.Hg * g.Wg;
    prof_begin(VQVAE_PROF_CONV_IGEMM, st);
    if (!(flags & VQVAE_CONV_EXACT_FP32)) {
        // default: split-bf16 products on the bf16 matrix cores (fp32-grade accuracy, ~2.7x the rate)
        const u32x4 *img3 = reinterpret_cast<const u32x4 *>(packed + packed_floats(g));
        const unsigned gx = (unsigned)((M + 127) / 128);
        const bool tile8 = g.Hin == 8 && g.Win == 8 && g.istride == 1 && g.Hg == 8 && g.Wg == 8 && Cin % 32 == 0 && g.ntile % 2 == 0;
        // 16x16 -> 8x8 (4x4 s2): the same kernel over 2x2 input blocks, weights in the s2d chunk order (third image)
        const bool S2D_ = !tile8 && kind == VQVAE_CONV_4x4_S2 && g.Hin == 16 && g.Win == 16 && Cin % 32 == 0 && g.ntile % 2 == 0;
        if (tile8 || S2D_)
        {
            // whole 8x8 input images per wave: operands split once per chunk and kept in LDS for all taps.  With four
            // output tiles per wave (all 128 channels: the image is read and split once) the workgroup has eight
            // waves, so that its weight chunks (2 x 24 KiB) and eight operand tiles still fit one CU's LDS.
            // ... unless that leaves most of the chip without a workgroup (round 4: the prior's sampler runs these layers at
            // B = 64): below one eight-wave workgroup per CU the four-wave form, two channel halves per image, spreads the same
            // images over four times the workgroups
            const long long wg_wide = ((B + 7) / 8) * (S2D_ ? 1 : g.nphase) * (g.ntile / 4);
            const bool wide = g.ntile % 4 == 0 && wg_wide >= num_cus();
            const int ny = (S2D_ ? 1 : g.nphase) * (g.ntile / (wide ? 4 : 2));
            const unsigned gxt = (unsigned)((B + (wide ? 7 : 3)) / (wide ? 8 : 4)) * ny;
            const bool h2 = !(flags & VQVAE_CONV_BF16_SPLIT);
            const char *h2base = reinterpret_cast<const char *>(packed) + packed_h2_offset(g, kind);
            const int *whdr = reinterpret_cast<const int *>(h2base);
            const u32x4 *wsel = h2 ? reinterpret_cast<const u32x4 *>(h2base + h2_header_bytes(g.ntile) + (S2D_ ? packed_h2_bytes(g) : 0))
                                   : (S2D_ ? img3 + packed_bf3_bytes(g) / sizeof(u32x4) : img3);
#define TILE8_LAUNCH(NT_, S2D__, NW_, H2_, THREADS_)                                                                   \
    hipLaunchKernelGGL((conv_tile8_bf3_kernel<NT_, S2D__, NW_, 2, H2_>), dim3(gxt), dim3(THREADS_), 0, st, x, wsel, bias, y, \
                       g, ny, whdr, in_amax, out_amax)
            if (wide) {
                if (S2D_) { if (h2) TILE8_LAUNCH(4, true, 8, true, 512); else TILE8_LAUNCH(4, true, 8, false, 512); }
                else      { if (h2) TILE8_LAUNCH(4, false, 8, true, 512); else TILE8_LAUNCH(4, false, 8, false, 512); }
            } else {
                if (S2D_) { if (h2) TILE8_LAUNCH(2, true, 4, true, 256); else TILE8_LAUNCH(2, true, 4, false, 256); }
                else      { if (h2) TILE8_LAUNCH(2, false, 4, true, 256); else TILE8_LAUNCH(2, false, 4, false, 256); }
            }
#undef TILE8_LAUNCH
        }
        else if (in_amax && kind == VQVAE_CONV_4x4_S2 && !(flags & VQVAE_CONV_BF16_SPLIT) && g.Hg % 8 == 0 && g.Wg % 8 == 0 &&
                 g.Hg * g.Wg > 64 && g.Hin == 2 * g.Hg && g.Win == 2 * g.Wg && Cin % 32 == 0 && g.ntile % 2 == 0 &&
                 (long long)g.Hin * g.Win * Cin * 4 < 0x7FFFFFF0ll) {
            // the 4x4 stride-2 conv on larger maps: 8x8 output tiles over the grid of 2x2 input blocks, with a one-block halo
            const char *h2base = reinterpret_cast<const char *>(packed) + packed_h2_offset(g, kind);
            const int *whdr = reinterpret_cast<const int *>(h2base);
            const u32x4 *wsel = reinterpret_cast<const u32x4 *>(h2base + h2_header_bytes(g.ntile) + packed_h2_bytes(g));      // space-to-depth chunk order
            const bool wide = g.ntile % 4 == 0;
            const long long tiles = (long long)B * (g.Hg / 8) * (g.Wg / 8);
            const int ny = g.nphase * (wide ? g.ntile / 4 : g.ntile / 2);
            const unsigned gxt = (unsigned)((tiles + 3) / 4) * ny;
            if (wide) hipLaunchKernelGGL((conv_halo8_h2_kernel<4, 2, true>), dim3(gxt), dim3(256), 0, st, x, wsel, bias, y, g, ny, whdr, in_amax, out_amax);
            else hipLaunchKernelGGL((conv_halo8_h2_kernel<2, 4, true>), dim3(gxt), dim3(256), 0, st, x, wsel, bias, y, g, ny, whdr, in_amax, out_amax);
        }
        else if (conv_halo8_ok(g, Cin, flags) && in_amax) {
            // larger maps whose grid is a multiple of 8 both ways, inside the whole-path entry points (maxima handed over):
            // 8x8 tiles with a one-pixel halo, one per wave (conv_halo8_h2_kernel)
            const char *h2base = reinterpret_cast<const char *>(packed) + packed_h2_offset(g, kind);
            const int *whdr = reinterpret_cast<const int *>(h2base);
            const u32x4 *wsel = reinterpret_cast<const u32x4 *>(h2base + h2_header_bytes(g.ntile));
            const bool wide = g.ntile % 4 == 0;
            const long long tiles = (long long)B * (g.Hg / 8) * (g.Wg / 8);
            // 64-channel conv-transpose phases go in pairs (one patch load / split for two phases)
            const bool pairs = !wide && g.nphase == 4;
            const int ny = (pairs ? 2 : g.nphase) * (wide ? g.ntile / 4 : g.ntile / 2);
            const unsigned gxt = (unsigned)((tiles + 3) / 4) * ny;
            // taps per weight stage: a kernel row of the 3x3 layers; two of a conv-transpose phase's four taps
#define HALO_LAUNCH(NT_, TPS_, NPH_) hipLaunchKernelGGL((conv_halo8_h2_kernel<NT_, TPS_, false, NPH_>), dim3(gxt), dim3(256), 0, st, x, wsel, bias, y, g, ny, whdr, in_amax, out_amax)
            if (g.ntaps == 1) {                        // 1x1: the tile without a border
                if (wide) hipLaunchKernelGGL((conv_halo8_h2_kernel<4, 1, false, 1, 0>), dim3(gxt), dim3(256), 0, st, x, wsel, bias, y, g, ny, whdr, in_amax, out_amax);
                else hipLaunchKernelGGL((conv_halo8_h2_kernel<2, 1, false, 1, 0>), dim3(gxt), dim3(256), 0, st, x, wsel, bias, y, g, ny, whdr, in_amax, out_amax);
            }
            else if (g.ntaps == 9) { if (wide) HALO_LAUNCH(4, 3, 1); else HALO_LAUNCH(2, 3, 1); }
            else if (pairs) HALO_LAUNCH(2, 2, 2);
            else { if (wide) HALO_LAUNCH(4, 2, 1); else HALO_LAUNCH(2, 4, 1); }
#undef HALO_LAUNCH
        }
        else {
            // generic maps: the two-term fp16 products need every image's maximum from the producing layer (in_amax); the
            // per-layer C entry points have none and use the three-term bf16 products
            const bool h2 = in_amax && !(flags & VQVAE_CONV_BF16_SPLIT);
            const char *h2base = reinterpret_cast<const char *>(packed) + packed_h2_offset(g, kind);
            const int *whdr = reinterpret_cast<const int *>(h2base);
            const u32x4 *wsel = h2 ? reinterpret_cast<const u32x4 *>(h2base + h2_header_bytes(g.ntile)) : img3;
#define IGEMM_LAUNCH(NT_, H2_, GY_)                                                                                     \
    hipLaunchKernelGGL((conv_igemm_bf3_kernel<NT_, H2_>), dim3(gx, GY_), dim3(256), 0, st, x, wsel, bias, y, g, whdr, in_amax, \
                       out_amax)
            if (g.ntile % 4 == 0) { if (h2) IGEMM_LAUNCH(4, true, g.nphase * (g.ntile / 4)); else IGEMM_LAUNCH(4, false, g.nphase * (g.ntile / 4)); }
            else if (g.ntile % 2 == 0) { if (h2) IGEMM_LAUNCH(2, true, g.nphase * (g.ntile / 2)); else IGEMM_LAUNCH(2, false, g.nphase * (g.ntile / 2)); }
            else { if (h2) IGEMM_LAUNCH(1, true, g.nphase * g.ntile); else IGEMM_LAUNCH(1, false, g.nphase * g.ntile); }
#undef IGEMM_LAUNCH
        }
    } else if (g.ntile % 4 == 0) {
        // exact-fp32 MFMA kernels: 32 pixels x 128 channels per wave when Cout fills it, else 64 x 64 / 32
        const unsigned gx = (unsigned)((M + 127) / 128);
        hipLaunchKernelGGL((conv_igemm_kernel<1, 4>), dim3(gx, g.nphase * (g.ntile / 4)), dim3(256), 0, st, x,
                           packed, bias, y, g);
    } else if (g.ntile % 2 == 0) {
        const unsigned gx = (unsigned)((M + 255) / 256);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 2>), dim3(gx, g.nphase * (g.ntile / 2)), dim3(256), 0, st, x,
                           packed, bias, y, g);
    } else {
        const unsigned gx = (unsigned)((M + 255) / 256);
        hipLaunchKernelGGL((conv_igemm_kernel<2, 1>), dim3(gx, g.nphase * g.ntile), dim3(256), 0, st, x, packed,
                           bias, y, g);
    }
    prof_end(VQVAE_PROF_CONV_IGEMM, st);
    return (int)hipGetLastError();
}

extern "C" {

int vqvae_res_layer_forward_f32(const float *x, const float *packed_w1, const float *packed_w2, int64_t B,
                                int H, int W, int C, int Rh, int flags, float *y, vqvae_stream_t stream) {
    return vqvae::res_layer_forward_impl(x, packed_w1, packed_w2, B, H, W, C, Rh, flags, y, static_cast<hipStream_t>(stream), nullptr, nullptr);
}

int vqvae_res_layer_forward_ws_f32(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C,
                                   int Rh, int flags, float *y, float *scratch, size_t scratch_bytes, vqvae_stream_t stream) {
    if (!vqvae::res_layer_fused_ok(C, Rh) && (!scratch || scratch_bytes < (size_t)B * H * W * Rh * sizeof(float))) return VQVAE_ERR_WORKSPACE;
    return vqvae::res_layer_forward_impl(x, packed_w1, packed_w2, B, H, W, C, Rh, flags, y, static_cast<hipStream_t>(stream), nullptr, nullptr,
                                         nullptr, scratch);
}

int vqvae_res_layer_forward_hidden_f32(const float *x, const float *packed_w1, const float *packed_w2, int64_t B,
                                       int H, int W, int C, int Rh, int flags, float *y, float *hidden,
                                       vqvae_stream_t stream) {
    if (!hidden) return VQVAE_ERR_NULL;
    return vqvae::res_layer_forward_impl(x, packed_w1, packed_w2, B, H, W, C, Rh, flags, y, static_cast<hipStream_t>(stream), nullptr, nullptr, hidden);
}
}  // extern "C"

bool vqvae::res_layer_fused_ok(int C, int Rh) { return Rh >= 1 && Rh <= 32 && (C == 32 || C == 64 || C == 128); }

int vqvae::res_layer_forward_impl(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H, int W,
                                  int C, int Rh, int flags, float *y, hipStream_t stream, const int *in_amax, int *out_amax,
                                  float *hidden, float *hid_scratch) {
    if (!x || !packed_w1 || !packed_w2 || !y) return VQVAE_ERR_NULL;
    // the hidden activation is written by the kernels that own whole 8x8 images only (full 32-wide hidden tile)
    if (hidden && (H != 8 || W != 8 || Rh != 32 || (flags & VQVAE_CONV_EXACT_FP32) || (reinterpret_cast<uintptr_t>(hidden) & 15)))
        return VQVAE_ERR_UNSUPPORTED;
    if (B < 1 || H < 1 || W < 1 || C < 1 || Rh < 1) return VQVAE_ERR_SHAPE;
    if (!res_layer_fused_ok(C, Rh)) {
        // widths outside the fused kernels: 3x3 conv -> 1x1 conv through the conv kernels + one combine pass; the hidden map
        // goes through the caller's scratch (vqvae_res_layer_forward_ws_f32 / the whole-path workspace)
        if (C % 4 || Rh % 4 || !hid_scratch || hidden) return VQVAE_ERR_UNSUPPORTED;
        if (x == y) return VQVAE_ERR_UNSUPPORTED;
        const int cf = flags & (VQVAE_CONV_BF16_SPLIT | VQVAE_CONV_EXACT_FP32);
        int rc = conv_forward_impl(VQVAE_CONV_3x3_S1, x, packed_w1, nullptr, B, H, W, C, Rh,
                                   (flags & VQVAE_CONV_RELU_IN) | VQVAE_CONV_RELU_OUT | cf, hid_scratch, stream, nullptr, nullptr);
        if (rc != 0) return rc;
        if ((rc = conv_forward_impl(VQVAE_CONV_1x1, hid_scratch, packed_w2, nullptr, B, H, W, Rh, C, cf, y, stream, nullptr, nullptr)) != 0) return rc;
        const long long n4 = (long long)B * H * W * C / 4;
        long long grid = (n4 + 255) / 256;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(res_combine_kernel, dim3((unsigned)grid), dim3(256), 0, stream, x, y, y, n4,
                           (flags & VQVAE_CONV_RELU_IN) ? 1 : 0, (flags & VQVAE_CONV_RELU_OUT) ? 1 : 0);
        return (int)hipGetLastError();
    }
    if (x == y) return VQVAE_ERR_UNSUPPORTED;           // 3x3 halo: not in place
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return VQVAE_ERR_UNSUPPORTED;   // 16-byte accesses
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = (long long)B * H * W;
    const unsigned gx = (unsigned)((M + 255) / 256);
    prof_begin(VQVAE_PROF_RES_LAYER, st);
    if (!(flags & VQVAE_CONV_EXACT_FP32)) {
        // split-bf16 images sit behind the fp32 ones in each packed buffer
        const int cpt = (C + 31) / 32;
        const u32x4 *w1b = reinterpret_cast<const u32x4 *>(packed_w1 + (size_t)9 * cpt * 1024);          // 3x3, C -> Rh
        const u32x4 *w2b = reinterpret_cast<const u32x4 *>(packed_w2 + (size_t)((C + 31) / 32) * 1024);   // 1x1, Rh -> C
        if (H == 8 && W == 8) {
            // whole 8x8 images per wave: operands split once and kept in LDS for all nine taps
            const unsigned gt = (unsigned)((B + 3) / 4);
            if (!(flags & VQVAE_CONV_BF16_SPLIT)) {
                // two-term fp16 images: [header {kw}][image] behind the bf16 ones (vqvae_conv_pack_f32)
                const char *h1 = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
                const char *h2 = reinterpret_cast<const char *>(packed_w2) + (size_t)((C + 31) / 32) * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
                const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1 + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes((C + 31) / 32));
                const int *hd1 = reinterpret_cast<const int *>(h1), *hd2 = reinterpret_cast<const int *>(h2);
                switch (C / 32) {
                    case 1: hipLaunchKernelGGL((res_tile8_bf3_kernel<1, true>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, hidden); break;
                    case 2: hipLaunchKernelGGL((res_tile8_bf3_kernel<2, true>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, hidden); break;
                    case 4: hipLaunchKernelGGL((res_tile8_bf3_kernel<4, true>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, hidden); break;
                }
            } else switch (C / 32) {
                case 1: hipLaunchKernelGGL((res_tile8_bf3_kernel<1>), dim3(gt), dim3(256), 0, st, x, w1b, w2b, y, (int)B, C, flags, nullptr, nullptr, nullptr, out_amax, hidden); break;
                case 2: hipLaunchKernelGGL((res_tile8_bf3_kernel<2>), dim3(gt), dim3(256), 0, st, x, w1b, w2b, y, (int)B, C, flags, nullptr, nullptr, nullptr, out_amax, hidden); break;
                case 4: hipLaunchKernelGGL((res_tile8_bf3_kernel<4>), dim3(gt), dim3(256), 0, st, x, w1b, w2b, y, (int)B, C, flags, nullptr, nullptr, nullptr, out_amax, hidden); break;
            }
        } else if (in_amax && !(flags & VQVAE_CONV_BF16_SPLIT) && H % 8 == 0 && W % 8 == 0 && C % 32 == 0 &&
                   (long long)H * W * C * 4 < 0x7FFFFFF0ll) {
            // larger maps that are multiples of 8 both ways, inside the whole-path entry points: 8x8 tiles with a halo
            const char *h1p = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
            const char *h2p = reinterpret_cast<const char *>(packed_w2) + (size_t)((C + 31) / 32) * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
            const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1p + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2p + h2_header_bytes((C + 31) / 32));
            const int *hd1 = reinterpret_cast<const int *>(h1p), *hd2 = reinterpret_cast<const int *>(h2p);
            const long long tiles = (long long)B * (H / 8) * (W / 8);
            const unsigned gt = (unsigned)((tiles + 3) / 4);
            switch (C / 32) {
                case 1: hipLaunchKernelGGL((res_halo8_h2_kernel<1>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, H, W, C, flags, hd1, hd2, in_amax, out_amax); break;
                case 2: hipLaunchKernelGGL((res_halo8_h2_kernel<2>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, H, W, C, flags, hd1, hd2, in_amax, out_amax); break;
                case 4: hipLaunchKernelGGL((res_halo8_h2_kernel<4>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, H, W, C, flags, hd1, hd2, in_amax, out_amax); break;
            }
        } else {
            // generic maps: two-term fp16 products when the producing layer handed over the images' maxima
            const bool h2 = in_amax && !(flags & VQVAE_CONV_BF16_SPLIT);
            const char *h1p = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
            const char *h2p = reinterpret_cast<const char *>(packed_w2) + (size_t)((C + 31) / 32) * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
            const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1p + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2p + h2_header_bytes((C + 31) / 32));
            const int *hd1 = reinterpret_cast<const int *>(h1p), *hd2 = reinterpret_cast<const int *>(h2p);
#define RES_GEN(NT_)                                                                                                             \
    do {                                                                                                                         \
        if (h2) hipLaunchKernelGGL((res_layer_bf3_kernel<NT_, true>), dim3(gx), dim3(256), 0, st, x, w1h, w2h, y, (int)B, H, W, C, \
                                   flags, hd1, hd2, in_amax, out_amax);                                                          \
        else hipLaunchKernelGGL((res_layer_bf3_kernel<NT_, false>), dim3(gx), dim3(256), 0, st, x, w1b, w2b, y, (int)B, H, W, C,  \
                                flags, nullptr, nullptr, nullptr, out_amax);                                                     \
    } while (0)
            switch (C / 32) {
            case 1: RES_GEN(1); break;
            case 2: RES_GEN(2); break;
            case 4: RES_GEN(4); break;
            }
#undef RES_GEN
        }
    } else {
        switch (C / 32) {
            case 1: hipLaunchKernelGGL((res_layer_kernel<1>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
            case 2: hipLaunchKernelGGL((res_layer_kernel<2>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
            case 4: hipLaunchKernelGGL((res_layer_kernel<4>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
        }
    }
    prof_end(VQVAE_PROF_RES_LAYER, st);
    return (int)hipGetLastError();
}

// Two layers of a residual stack (shared weights) in one launch: 8x8 maps on the two-term fp16 path only.  x == y is
// allowed (every wave reads its image completely before it writes it).  Returns VQVAE_ERR_UNSUPPORTED when the caller
// has to run the two layers separately.
bool vqvae::res_pair_supported(int H, int W, int C, int Rh, int flags) {
    return H == 8 && W == 8 && Rh >= 1 && Rh <= 32 && (C == 32 || C == 64 || C == 128) &&
           !(flags & (VQVAE_CONV_BF16_SPLIT | VQVAE_CONV_EXACT_FP32));
}

// post (optional): a 1x1 conv (+ bias) applied to the pair's output inside the same kernel; y is then not written.
bool vqvae::res_pair_post_supported(int C, int Cout) { return C == 128 && (Cout == 32 || Cout == 64 || Cout == 128); }

int vqvae::res_pair_forward_impl(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H, int W,
                                 int C, int Rh, int flags, float *y, hipStream_t st, const int *in_amax, int *out_amax,
                                 const ResPairPost *post) {
    if (!x || !packed_w1 || !packed_w2 || (!y && !post)) return VQVAE_ERR_NULL;
    if (B < 1 || !res_pair_supported(H, W, C, Rh, flags)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(post ? post->out : nullptr)) & 15) return VQVAE_ERR_UNSUPPORTED;
    const int cpt = (C + 31) / 32;
    const char *h1 = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const char *h2 = reinterpret_cast<const char *>(packed_w2) + (size_t)((C + 31) / 32) * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1 + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes((C + 31) / 32));
    const int *hd1 = reinterpret_cast<const int *>(h1), *hd2 = reinterpret_cast<const int *>(h2);
    const unsigned gt = (unsigned)((B + 3) / 4);
    // the checks that can refuse come BEFORE prof_begin: an early return behind it would leave an unmatched begin event
    ConvGeom g3;
    if (post && (!post->packed || !post->out || !res_pair_post_supported(C, post->Cout) ||
                 make_geom(VQVAE_CONV_1x1, 1, 8, 8, C, post->Cout, 0, g3) != VQVAE_OK)) return VQVAE_ERR_UNSUPPORTED;
    prof_begin(VQVAE_PROF_RES_LAYER, st);
    if (post) {
        const char *h3 = reinterpret_cast<const char *>(post->packed) + packed_h2_offset(g3, VQVAE_CONV_1x1);
        const u32x4 *w3h = reinterpret_cast<const u32x4 *>(h3 + h2_header_bytes(g3.ntile));
        const int *hd3 = reinterpret_cast<const int *>(h3);
#define PAIR_POST(NT3_)                                                                                                         \
    hipLaunchKernelGGL((res_pair8_h2_kernel<4, NT3_>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2,  \
                       in_amax, out_amax, w3h, hd3, post->bias, post->out)
        switch (post->Cout / 32) {
            case 1: PAIR_POST(1); break;
            case 2: PAIR_POST(2); break;
            case 4: PAIR_POST(4); break;
        }
#undef PAIR_POST
    } else switch (C / 32) {
        case 1: hipLaunchKernelGGL((res_pair8_h2_kernel<1>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, nullptr, nullptr, nullptr, nullptr); break;
        case 2: hipLaunchKernelGGL((res_pair8_h2_kernel<2>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, nullptr, nullptr, nullptr, nullptr); break;
        case 4: hipLaunchKernelGGL((res_pair8_h2_kernel<4>), dim3(gt), dim3(256), 0, st, x, w1h, w2h, y, (int)B, C, flags, hd1, hd2, in_amax, out_amax, nullptr, nullptr, nullptr, nullptr); break;
    }
    prof_end(VQVAE_PROF_RES_LAYER, st);
    return (int)hipGetLastError();
}

// A 3x3 conv / conv-transpose (stride 1, Cin -> 128, + bias + ReLU) and the two residual layers behind it in one launch
// (conv_res_pair8_h2_kernel); post as in res_pair_forward_impl.  x == y is not allowed (x has Cin channels).
bool vqvae::conv_res_pair_supported(int kind, int H, int W, int Cin, int C, int Rh) {
    return (kind == VQVAE_CONV_3x3_S1 || kind == VQVAE_CONVT_3x3_S1) && H == 8 && W == 8 && C == 128 && Cin >= 32 && Cin % 32 == 0 &&
           Cin <= 256 && Rh >= 1 && Rh <= 32;
}

int vqvae::conv_res_pair_forward_impl(int kind, const float *x, const float *packed_front, const float *bias_front, int Cin,
                                      const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C, int Rh,
                                      int flags, float *y, hipStream_t st, const int *in_amax, int *out_amax,
                                      const ResPairPost *post) {
    if (!x || !packed_front || !packed_w1 || !packed_w2 || (!y && !post)) return VQVAE_ERR_NULL;
    if (B < 1 || !conv_res_pair_supported(kind, H, W, Cin, C, Rh)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(post ? post->out : nullptr)) & 15)
        return VQVAE_ERR_UNSUPPORTED;
    ConvGeom g;
    if (make_geom(kind, B, H, W, Cin, C, 0, g) != VQVAE_OK || g.nphase != 1 || g.ntaps != 9) return VQVAE_ERR_UNSUPPORTED;
    const char *hf = reinterpret_cast<const char *>(packed_front) + packed_h2_offset(g, kind);
    FrontConv fc;
    fc.wimg = reinterpret_cast<const u32x4 *>(hf + h2_header_bytes(g.ntile));
    fc.hdr = reinterpret_cast<const int *>(hf);
    fc.bias = bias_front;
    fc.dym = g.dymask[0];
    fc.dxm = g.dxmask[0];
    fc.Cin = Cin;
    const int cpt = C / 32;
    const char *h1 = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const char *h2 = reinterpret_cast<const char *>(packed_w2) + (size_t)cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1 + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes((C + 31) / 32));
    const int *hd1 = reinterpret_cast<const int *>(h1), *hd2 = reinterpret_cast<const int *>(h2);
    const unsigned gtc = (unsigned)((B + CRP_NW - 1) / CRP_NW);
    // the checks that can refuse come BEFORE prof_begin: an early return behind it would leave an unmatched begin event
    ConvGeom g3;
    if (post && (!post->packed || !post->out || !res_pair_post_supported(C, post->Cout) ||
                 make_geom(VQVAE_CONV_1x1, 1, 8, 8, C, post->Cout, 0, g3) != VQVAE_OK)) return VQVAE_ERR_UNSUPPORTED;
    if (post && post->vq && (post->Cout != 64 || CRP_NW != 4 || post->vq->K32 % 128 || post->vq->K32 > 512 || !post->vq->partials))
        return VQVAE_ERR_UNSUPPORTED;
    prof_begin(VQVAE_PROF_RES_LAYER, st);
    if (post) {
        const char *h3 = reinterpret_cast<const char *>(post->packed) + packed_h2_offset(g3, VQVAE_CONV_1x1);
        const u32x4 *w3h = reinterpret_cast<const u32x4 *>(h3 + h2_header_bytes(g3.ntile));
        const int *hd3 = reinterpret_cast<const int *>(h3);
#define CRP_POST(NT3_)                                                                                                          \
    hipLaunchKernelGGL((conv_res_pair8_h2_kernel<NT3_>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags, hd1, hd2, \
                       in_amax, out_amax, w3h, hd3, post->bias, post->out, post->zero, post->zero_n, VqFuse{})
        if (post->vq) {
            // the quantizer rides behind the 1x1 conv: z_e is never written (post->out unused)
            hipLaunchKernelGGL((conv_res_pair8_h2_kernel<2, true>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags,
                               hd1, hd2, in_amax, out_amax, w3h, hd3, post->bias, post->out, post->zero, post->zero_n, *post->vq);
        } else
        switch (post->Cout / 32) {
            case 1: CRP_POST(1); break;
            case 2: CRP_POST(2); break;
            case 4: CRP_POST(4); break;
        }
#undef CRP_POST
    } else {
        hipLaunchKernelGGL((conv_res_pair8_h2_kernel<0>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags, hd1, hd2,
                           in_amax, out_amax, nullptr, nullptr, nullptr, nullptr, nullptr, 0, VqFuse{});
    }
    prof_end(VQVAE_PROF_RES_LAYER, st);
    return (int)hipGetLastError();
}

extern "C" {

size_t vqvae_conv_in_packed_bytes(int Cin, int Cout) {
    if (!(Cin == 1 || Cin == 3 || Cin == 4) || Cout < 1 || Cout > 128) return 0;
    const int S = Cin * 8, JG = (S + 3) / 4;
    // [fp32 B-operand image][split-bf16 image][header (kw per output channel, [1] = L1)][two-term fp16 A-operand image (enc_front8_h2_kernel)]
    return (size_t)((Cout + 31) / 32) * ((size_t)JG * 256 + (size_t)Cin * 768) * sizeof(float) + h2_header_bytes((Cout + 31) / 32) +
           (size_t)((Cout + 31) / 32) * Cin * 2048;
}

int vqvae_conv_in_pack_f32(const float *w, int Cin, int Cout, float *packed, vqvae_stream_t stream) {
    if (!w || !packed) return VQVAE_ERR_NULL;
    if (vqvae_conv_in_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int ntile = (Cout + 31) / 32;
    switch (Cin) {
        case 1: hipLaunchKernelGGL((conv_in_pack_kernel<1>), dim3(16), dim3(256), 0, st, w, packed, Cout, ntile); break;
        case 3: hipLaunchKernelGGL((conv_in_pack_kernel<3>), dim3(16), dim3(256), 0, st, w, packed, Cout, ntile); break;
        case 4: hipLaunchKernelGGL((conv_in_pack_kernel<4>), dim3(16), dim3(256), 0, st, w, packed, Cout, ntile); break;
    }
    unsigned short *img3 = reinterpret_cast<unsigned short *>(packed + (size_t)ntile * ((Cin * 8 + 3) / 4) * 256);
    switch (Cin) {
        case 1: hipLaunchKernelGGL((conv_in_pack_bf3_kernel<1>), dim3(16), dim3(256), 0, st, w, img3, Cout, ntile); break;
        case 3: hipLaunchKernelGGL((conv_in_pack_bf3_kernel<3>), dim3(16), dim3(256), 0, st, w, img3, Cout, ntile); break;
        case 4: hipLaunchKernelGGL((conv_in_pack_bf3_kernel<4>), dim3(16), dim3(256), 0, st, w, img3, Cout, ntile); break;
    }
    char *h2 = reinterpret_cast<char *>(packed) + (size_t)ntile * ((size_t)((Cin * 8 + 3) / 4) * 256 + (size_t)Cin * 768) * sizeof(float);
    int *hdr = reinterpret_cast<int *>(h2);
    u32x4 *img16 = reinterpret_cast<u32x4 *>(h2 + h2_header_bytes(ntile));
    hipLaunchKernelGGL(conv_wscale_kernel, dim3(32 * ntile), dim3(64), 0, st, w, Cin, Cout, 16, 0, ntile, hdr);
    hipLaunchKernelGGL(conv_in_hdr_kernel, dim3(1), dim3(256), 0, st, w, Cin * 16, Cout, hdr);
    switch (Cin) {
        case 1: hipLaunchKernelGGL((conv_in_pack_h2_kernel<1>), dim3(4), dim3(256), 0, st, w, img16, Cout, ntile, hdr); break;
        case 3: hipLaunchKernelGGL((conv_in_pack_h2_kernel<3>), dim3(4), dim3(256), 0, st, w, img16, Cout, ntile, hdr); break;
        case 4: hipLaunchKernelGGL((conv_in_pack_h2_kernel<4>), dim3(4), dim3(256), 0, st, w, img16, Cout, ntile, hdr); break;
    }
    return (int)hipGetLastError();
}

int vqvae_conv_in_forward_f32(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H,
                              int W, int Cin, int Cout, int flags, float *y, vqvae_stream_t stream) {
    return vqvae::conv_in_forward_impl(x_nchw, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr);
}

int vqvae_conv_in_forward_ep_f32(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H, int W, int Cin, int Cout,
                                 int flags, const float *mask, float *y, vqvae_stream_t stream) {
    return vqvae::conv_in_forward_impl(x_nchw, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr, mask);
}
}  // extern "C"

// The encoder's first two layers in one launch (enc_front8_h2_kernel): 32x32 images, 3 input channels, 64 -> 128 channels.
bool vqvae::enc_front_supported(int H, int W, int Cin, int C1, int C2) { return H == 32 && W == 32 && Cin == 3 && C1 == 64 && C2 == 128; }

int vqvae::enc_front_forward_impl(const float *x_nchw, const float *packed_in, const float *bias_in, const float *packed2,
                                  const float *bias2, int64_t B, int H, int W, int Cin, int C1, int C2, float *y, hipStream_t st,
                                  int *out_amax, int *zero_buf, int zero_n) {
    if (!x_nchw || !packed_in || !packed2 || !y) return VQVAE_ERR_NULL;
    if (B < 1 || !enc_front_supported(H, W, Cin, C1, C2)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_nchw) | reinterpret_cast<uintptr_t>(y)) & 15) return VQVAE_ERR_UNSUPPORTED;
    const int ntile0 = (C1 + 31) / 32;
    const char *h0 = reinterpret_cast<const char *>(packed_in) +
                     (size_t)ntile0 * ((size_t)((Cin * 8 + 3) / 4) * 256 + (size_t)Cin * 768) * sizeof(float);
    ConvGeom g;
    if (make_geom(VQVAE_CONV_4x4_S2, B, H / 2, W / 2, C1, C2, 0, g) != VQVAE_OK) return VQVAE_ERR_UNSUPPORTED;
    const char *h2 = reinterpret_cast<const char *>(packed2) + packed_h2_offset(g, VQVAE_CONV_4x4_S2);
    const u32x4 *w2s2d = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes(g.ntile) + packed_h2_bytes(g));     // space-to-depth chunk order
    prof_begin(VQVAE_PROF_CONV_IGEMM, st);
    hipLaunchKernelGGL((enc_front8_h2_kernel<3>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, x_nchw,
                       reinterpret_cast<const u32x4 *>(h0 + h2_header_bytes(ntile0)), reinterpret_cast<const int *>(h0), bias_in, w2s2d,
                       reinterpret_cast<const int *>(h2), bias2, y, (int)B, out_amax, zero_buf, zero_n);
    prof_end(VQVAE_PROF_CONV_IGEMM, st);
    return (int)hipGetLastError();
}

void vqvae::act_absmax_impl(const float *x, int64_t B, long long elems_per_image, int *amax, hipStream_t st) {
    long long parts = (elems_per_image + 256 * 4 * 16 - 1) / (256 * 4 * 16);
    parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
    hipLaunchKernelGGL(act_absmax_kernel, dim3((unsigned)parts, (unsigned)B), dim3(256), 0, st, x, elems_per_image, amax);
}

int vqvae::conv_in_forward_impl(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H, int W,
                                int Cin, int Cout, int flags, float *y, hipStream_t stream, int *out_amax, const float *ep_mask) {
    if (!x_nchw || !packed || !y) return VQVAE_ERR_NULL;
    if (ep_mask && (ep_mask == y || (reinterpret_cast<uintptr_t>(ep_mask) & 15) || out_amax)) return VQVAE_ERR_UNSUPPORTED;
    if (B < 1 || H < 2 || W < 2) return VQVAE_ERR_SHAPE;
    if (H % 2 || W % 2 || vqvae_conv_in_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = B * (long long)(H / 2) * (W / 2);
    const unsigned gx = (unsigned)((M + 255) / 256);
    const int ntile = (Cout + 31) / 32;
    // whole output rows per workgroup -> LDS-staged input band (conv_in_rows_kernel)
    const int Hg = H / 2, Wg = W / 2;
    // tile: the widest power of two TW <= 256 that divides Wg with 256 / TW rows dividing Hg
    int tw_log2 = -1;
    for (int t = 8; t >= 1; --t)
        if (Wg % (1 << t) == 0 && Hg % (256 >> t) == 0) { tw_log2 = t; break; }
    const bool rows = tw_log2 > 0 && W % 4 == 0 && (long long)Hg * Wg * Cout * 4 < 0xFFFFFFF0ll &&      // 32-bit byte offsets inside an image
                      ((reinterpret_cast<uintptr_t>(x_nchw) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    const int jg = (Cin * 8 + 3) / 4;
    const bool bf3 = !(flags & VQVAE_CONV_EXACT_FP32);     // split-bf16 products unless the fp32 MFMA is asked for
    const float *packed3 = packed + (size_t)ntile * jg * 256;
    size_t rows_lds = rows ? ((size_t)ntile * (bf3 ? Cin * 768 : jg * 256) +
                              (size_t)Cin * (2 * (256 >> tw_log2) + 2) * (2 * (1 << tw_log2) + 8)) * sizeof(float) : 0;
    if (rows_lds < 4 * 32 * 36 * sizeof(float)) rows_lds = 4 * 32 * 36 * sizeof(float);   // the epilogue's output tiles
#define CI_LAUNCH(CIN_, NT_)                                                                                       \
    do {                                                                                                           \
        if (rows && rows_lds <= 64 * 1024 && bf3)                                                                  \
            hipLaunchKernelGGL((conv_in_rows_kernel<CIN_, NT_, true>), dim3(gx), dim3(256), rows_lds, st, x_nchw,  \
                               packed3, bias, y, (int)B, H, W, Cout, flags, out_amax, tw_log2, ep_mask);           \
        else if (rows && rows_lds <= 64 * 1024)                                                                    \
            hipLaunchKernelGGL((conv_in_rows_kernel<CIN_, NT_, false>), dim3(gx), dim3(256), rows_lds, st, x_nchw, \
                               packed, bias, y, (int)B, H, W, Cout, flags, out_amax, tw_log2, ep_mask);            \
        else                                                                                                       \
        {                                                                                                          \
            hipLaunchKernelGGL((conv_in_kernel<CIN_, NT_>), dim3(gx), dim3(256), 0, st, x_nchw, packed, bias, y,    \
                               (int)B, H, W, Cout, flags);                                                         \
            if (out_amax) act_absmax_impl(y, B, (long long)(H / 2) * (W / 2) * Cout, out_amax, st);               \
        }                                                                                                          \
    } while (0)
#define CI_NT(CIN_)                                                       \
    switch (ntile) {                                                      \
        case 1: CI_LAUNCH(CIN_, 1); break;                                \
        case 2: CI_LAUNCH(CIN_, 2); break;                                \
        case 3: CI_LAUNCH(CIN_, 3); break;                                \
        default: CI_LAUNCH(CIN_, 4); break;                               \
    }
    if (ep_mask && !(rows && rows_lds <= 64 * 1024)) return VQVAE_ERR_UNSUPPORTED;     // the mask lives in the row-band kernel only
    prof_begin(VQVAE_PROF_CONV_IN, st);
    switch (Cin) {
        case 1: CI_NT(1); break;
        case 3: CI_NT(3); break;
        case 4: CI_NT(4); break;
    }
#undef CI_NT
#undef CI_LAUNCH
    prof_end(VQVAE_PROF_CONV_IN, st);
    return (int)hipGetLastError();
}

extern "C" {

size_t vqvae_convt_out_packed_bytes(int Cin, int Cout) {
    if (Cin < 4 || Cin % 4 || Cin > 256 || Cout < 1 || Cout > 4) return 0;
    const int ntile = (16 * Cout + 31) / 32;
    // [fp32 B-operand image][three-term bf16 image][header (one tile: kw per output channel)][two-term fp16 image]
    // ... [A-operand image of dec_tail8_h2_kernel: 16 KiB]
    return (size_t)((Cin + 31) / 32) * ntile * (1024 * sizeof(float) + 3072 * sizeof(unsigned short)) + h2_header_bytes(1) +
           (size_t)((Cin + 31) / 32) * ntile * 2048 * sizeof(unsigned short) + 16384;
}

int vqvae_convt_out_pack_f32(const float *w, int Cin, int Cout, float *packed, vqvae_stream_t stream) {
    if (!w || !packed) return VQVAE_ERR_NULL;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    const int ntile_p = (16 * Cout + 31) / 32;
    hipLaunchKernelGGL(convt_out_pack_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed,
                       Cin, Cout, ntile_p);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t cells = (size_t)((Cin + 31) / 32) * ntile_p;
    hipLaunchKernelGGL(convt_out_pack_bf3_kernel<false>, dim3(32), dim3(256), 0, st, w,
                       reinterpret_cast<unsigned short *>(packed + cells * 1024), Cin, Cout, ntile_p, (const int *)nullptr);
    char *h2 = reinterpret_cast<char *>(packed) + cells * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    int *hdr = reinterpret_cast<int *>(h2);
    hipLaunchKernelGGL(conv_wscale_kernel, dim3(32), dim3(64), 0, st, w, Cin, Cout, 16, 1, 1, hdr);
    hipLaunchKernelGGL(convt_out_pack_bf3_kernel<true>, dim3(32), dim3(256), 0, st, w,
                       reinterpret_cast<unsigned short *>(h2 + h2_header_bytes(1)), Cin, Cout, ntile_p, hdr);
    hipLaunchKernelGGL(convt_out_pack_a_kernel, dim3(2), dim3(256), 0, st, w,
                       reinterpret_cast<u32x4 *>(h2 + h2_header_bytes(1) + cells * 2048 * sizeof(unsigned short)), Cin, Cout, hdr);
    return (int)hipGetLastError();
}

int vqvae_convt_out_forward_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                                int Cin, int Cout, int flags, float *y_nchw, vqvae_stream_t stream) {
    return vqvae::convt_out_forward_impl(x, packed, bias, B, H, W, Cin, Cout, flags, y_nchw, static_cast<hipStream_t>(stream), nullptr);
}
}  // extern "C"

// The decoder's last two layers in one launch (dec_tail8_h2_kernel): 8x8 maps, 128 -> 64 -> 3 channels.
bool vqvae::dec_tail_supported(int h4, int w4, int C, int C1, int Cout) { return h4 == 8 && w4 == 8 && C == 128 && C1 == 64 && Cout == 3; }

int vqvae::dec_tail_forward_impl(const float *x, const float *packed2, const float *bias2, const float *packed4, const float *bias4,
                                 int64_t B, int h4, int w4, int C, int C1, int Cout, float *y_nchw, hipStream_t st, const int *in_amax) {
    if (!x || !packed2 || !packed4 || !y_nchw) return VQVAE_ERR_NULL;
    if (B < 1 || !dec_tail_supported(h4, w4, C, C1, Cout)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_nchw)) & 15) return VQVAE_ERR_UNSUPPORTED;
    ConvGeom g;
    if (make_geom(VQVAE_CONVT_4x4_S2, B, h4, w4, C, C1, 0, g) != VQVAE_OK) return VQVAE_ERR_UNSUPPORTED;
    const char *h2 = reinterpret_cast<const char *>(packed2) + packed_h2_offset(g, VQVAE_CONVT_4x4_S2);
    TailGeom tg;
    for (int ph = 0; ph < 4; ++ph) { tg.dym[ph] = g.dymask[ph]; tg.dxm[ph] = g.dxmask[ph]; }
    const int ntile4 = (16 * Cout + 31) / 32, cpt4 = (C1 + 31) / 32;
    const size_t cells = (size_t)cpt4 * ntile4;
    const char *h4p = reinterpret_cast<const char *>(packed4) + cells * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    prof_begin(VQVAE_PROF_CONV_OUT, st);
    hipLaunchKernelGGL(dec_tail8_h2_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, x,
                       reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes(g.ntile)), reinterpret_cast<const int *>(h2), bias2, tg,
                       reinterpret_cast<const u32x4 *>(h4p + h2_header_bytes(1) + cells * 2048 * sizeof(unsigned short)),
                       reinterpret_cast<const int *>(h4p), bias4, y_nchw, (int)B, in_amax);
    prof_end(VQVAE_PROF_CONV_OUT, st);
    return (int)hipGetLastError();
}

// in_amax: the input images' maxima from the producing layer (whole-path entry points) -> two-term fp16 products
int vqvae::convt_out_forward_impl(const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin,
                                  int Cout, int flags, float *y_nchw, hipStream_t stream, const int *in_amax) {
    if (!x || !packed || !y_nchw) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1) return VQVAE_ERR_SHAPE;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int halo_y = H > 16, halo_x = W > 16;
    const int TH = halo_y ? 14 : H, TW = halo_x ? 14 : W;
    const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
    const long long ntiles = B * (long long)tiles_y * tiles_x;
    if (ntiles > INT32_MAX) return VQVAE_ERR_OVERFLOW;
    if ((long long)H * W * Cin * 4 >= 0x7FFFFFF0ll) return VQVAE_ERR_OVERFLOW;          // one image per buffer descriptor
    const int ntile = (16 * Cout + 31) / 32, cpt = (Cin + 31) / 32;
    const bool h2 = in_amax && !(flags & (VQVAE_CONV_EXACT_FP32 | VQVAE_CONV_BF16_SPLIT));
    const bool bf3 = !h2 && !(flags & VQVAE_CONV_EXACT_FP32);     // split products unless the fp32 MFMA is asked for
    const size_t lds = ((h2 ? 0 : (size_t)cpt * ntile * (bf3 ? 1536 : 1024)) + 256 * (16 * Cout + 1)) * sizeof(float);
    const char *h2base = reinterpret_cast<const char *>(packed) + (size_t)cpt * ntile * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const int *whdr = reinterpret_cast<const int *>(h2base);
    const float *wimg = h2 ? reinterpret_cast<const float *>(h2base + h2_header_bytes(1)) : (bf3 ? packed + (size_t)cpt * ntile * 1024 : packed);
    // persistent workgroups, as many as fit on the chip at once (two per CU: ~230 registers per lane with a tile's input in flight)
    const long long resident = (long long)num_cus() * (lds <= 80 * 1024 ? 2 : 1);
    const long long grid = ntiles < resident ? ntiles : resident;
    prof_begin(VQVAE_PROF_CONV_OUT, st);
#define CTO_LAUNCH(NT_, BF_)                                                                                          \
    do {                                                                                                              \
        auto k = convt_out_kernel<NT_, BF_>;                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                  kLdsBytes);                                                                         \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, x, wimg, bias, y_nchw, (int)B, H, W, Cin,     \
                           Cout, TH, TW, halo_y, halo_x, tiles_y, tiles_x, whdr, in_amax, (int)ntiles);               \
    } while (0)
    if (ntile == 1) { if (h2) CTO_LAUNCH(1, 2); else if (bf3) CTO_LAUNCH(1, 1); else CTO_LAUNCH(1, 0); }
    else { if (h2) CTO_LAUNCH(2, 2); else if (bf3) CTO_LAUNCH(2, 1); else CTO_LAUNCH(2, 0); }
#undef CTO_LAUNCH
    prof_end(VQVAE_PROF_CONV_OUT, st);
    return (int)hipGetLastError();
}

extern "C" {

int vqvae_transpose_f32(const float *x, int64_t batch, int R, int Cc, float *y, vqvae_stream_t stream) {
    if (!x || !y) return VQVAE_ERR_NULL;
    if (batch < 1 || R < 1 || Cc < 1) return VQVAE_ERR_SHAPE;
    if (batch > 65535) return VQVAE_ERR_OVERFLOW;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, R, Cc);
    return (int)hipGetLastError();
}

}  // extern "C"
