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
#include "conv_host.h"

namespace vqvae {
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
                        if (relu_out) v = relu1(v);
                        out[off + n] = v;
                    }
                }
            }
        }
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
                    if (relu_out) v = relu1(v);
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
                    if (relu_out) v[r] = relu1(v[r]);
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
                        if (relu_out) v = relu1(v);
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
                        if (ro) { v0 = relu1(v0); v1 = relu1(v1); }
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

}  // namespace vqvae

using namespace vqvae;

// one 64-thread block per output channel of the 32-channel tiles (the other conv units pack their images behind the same scales)
void vqvae::conv_wscale_launch(const float *w, int Cin, int Cout, int kk, int transposed, int ntile, int *hdr, hipStream_t st) {
    hipLaunchKernelGGL(conv_wscale_kernel, dim3(32 * ntile), dim3(64), 0, st, w, Cin, Cout, kk, transposed, ntile, hdr);
}

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
    conv_wscale_launch(w, Cin, Cout, g.kk, g.transposed, g.ntile, hdr, st);
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

int vqvae_conv_taps_forward_ep_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin, int Cout,
                                   int ntaps, const int8_t *dy, const int8_t *dx, int flags, const float *addend, const float *mask, float *y,
                                   vqvae_stream_t stream) {
    if ((addend && addend == y) || (mask && mask == y)) return VQVAE_ERR_UNSUPPORTED;
    vqvae::TapSpec t;
    const int rc = fill_taps(t, ntaps, dy, dx);
    if (rc != VQVAE_OK) return rc;
    vqvae::TapScope scope(&t);
    return vqvae::conv_forward_impl(VQVAE_CONV_TAPS, x, packed, bias, B, H, W, Cin, Cout, flags, y, static_cast<hipStream_t>(stream), nullptr,
                                    nullptr, addend, mask);
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

void vqvae::act_absmax_impl(const float *x, int64_t B, long long elems_per_image, int *amax, hipStream_t st) {
    long long parts = (elems_per_image + 256 * 4 * 16 - 1) / (256 * 4 * 16);
    parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
    hipLaunchKernelGGL(act_absmax_kernel, dim3((unsigned)parts, (unsigned)B), dim3(256), 0, st, x, elems_per_image, amax);
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
