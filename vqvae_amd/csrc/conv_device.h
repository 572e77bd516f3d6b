// Device-side helpers shared by the conv translation units (conv.hip, conv_res.hip, conv_fused.hip, conv_ends.hip): the layer
// geometry, epilogue pieces, the three-term bf16 / two-term fp16 operand splits and their MFMA groups, per-image maxima,
// the per-output-channel scale tables of the two-term images.  Everything here is __device__ __forceinline__, a type or a constant.
#pragma once
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
// ReLU that KEEPS a NaN, as nn.ReLU does (models/residual.py:19,22, encoder.py:31,34, decoder.py:33: torch's relu of a NaN is a
// NaN).  v_max_f32 returns the other operand for a quiet NaN and would flush it to 0; gfx950 has the IEEE-754-2019 `maximum`
// (v_maximum3_f32: any NaN operand makes the result NaN) at the same one-instruction cost.  The activation MAXIMA that make the
// per-image scales stay on v_max / v_max3 on purpose: a NaN pixel must not become its image's scale.
__device__ __forceinline__ float relu1(float a) {
    float o;
    asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(o) : "v"(a));
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
    const f32x2v r = {relu1(__builtin_fmaf(a0, d, b0)), relu1(__builtin_fmaf(a1, d, b1))};
    vmax3(m, r.x, r.y);
    return r;
}
// the same with one scale per value (round 4: the weight rows' own powers of two)
#define SCALE2_BIAS_RELU2(A0, A1, D0, D1, B0, B1, M)                             \
    do {                                                                         \
        const float r0_ = relu1(__builtin_fmaf((A0), (D0), (B0)));          \
        const float r1_ = relu1(__builtin_fmaf((A1), (D1), (B1)));          \
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
    v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w);
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

}  // namespace vqvae
