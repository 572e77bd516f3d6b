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

#include "common.h"

namespace vqvae {

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
};

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

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
    return v;
}

// ---------------------------------------------------------------------------
// Weight packing (once per layer / weight version).
__global__ __launch_bounds__(256) void conv_pack_kernel(const float *__restrict__ w, float *__restrict__ img,
                                                        ConvGeom g, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = e & 3, n = (e >> 2) & 31, h = (e >> 7) & 1, j = (e >> 8) & 3;
        long long t = e >> 10;
        const int nt = (int)(t % g.ntile); t /= g.ntile;
        const int nchunk = g.ntaps * g.cpt;
        const int chunk = (int)(t % nchunk);
        const int phase = (int)(t / nchunk);
        const int tap = chunk / g.cpt, cc = chunk - tap * g.cpt;
        const int ci = cc * 32 + 16 * h + 4 * j + i, co = nt * 32 + n;
        float v = 0.0f;
        if (ci < g.Cin && co < g.Cout) {
            const int kyx = g.kyx[phase][tap];
            v = g.transposed ? w[((size_t)ci * g.Cout + co) * g.kk + kyx]
                             : w[((size_t)co * g.Cin + ci) * g.kk + kyx];
        }
        img[e] = v;
    }
}

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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // wave-private tile: LDS ops of a wave stay in order
    __builtin_amdgcn_wave_barrier();
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

// ---------------------------------------------------------------------------
// Last layer: nn.ConvTranspose2d(Cin, Cout<=4, k=4, s=2, p=1), row-major in, NCHW image out
// (models/decoder.py:34-35).  Cout = 3 cannot fill a 32-wide MFMA tile as an output-channel
// dimension, so the layer runs in its GEMM + col2im form inside ONE kernel:
//   T[pixel][tap*Cout + co] = sum_ci x[pixel][ci] * w[ci][co][tap]     (N = 16*Cout <= 64 on the MFMA)
//   out[co][oy][ox] = bias[co] + sum over the 4 (ky,kx) with matching parity of T[(oy+1-ky)/2][(ox+1-kx)/2][ky][kx][co]
// A workgroup owns a 16x16 region of input pixels (a 14x14 interior + 1-pixel halo, or the whole
// image when it is at most 16 wide/high), keeps T for the region in LDS and writes the interior's
// 2x upsampled outputs with coalesced NCHW stores.
template <int NT>
__global__ __launch_bounds__(256) void convt_out_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ wimg,
                                                        const float *__restrict__ bias,
                                                        float *__restrict__ out, int B, int H, int W,
                                                        int Cin, int Cout, int TH, int TW, int halo_y,
                                                        int halo_x, int tiles_y, int tiles_x) {
    constexpr int MT = 2, STRIDE = NT * 32 + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cpt = (Cin + 31) / 32;
    float *Ws = smem;                               // [cpt][NT][1024]
    float *Ts = smem + (size_t)cpt * NT * 1024;     // [256][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;

    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const long long b = t / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int ry = y0 - halo_y, rx = x0 - halo_x;

    for (int i = tid; i < cpt * NT * 256; i += 256)
        reinterpret_cast<f32x4 *>(Ws)[i] = reinterpret_cast<const f32x4 *>(wimg)[i];

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    bool ok[MT];
    const float *src[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = wave * 64 + mt * 32 + l31;
        const int iy = ry + (p >> 4), ix = rx + (p & 15);
        ok[mt] = iy >= 0 && iy < H && ix >= 0 && ix < W;
        src[mt] = in + ((b * H + iy) * (long long)W + ix) * Cin + 16 * h;
    }
    __syncthreads();
    for (int c = 0; c < cpt; ++c) {
        f32x4 a[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (ok[mt] && c * 32 + 16 * h + 4 * j < Cin)
                    v = *reinterpret_cast<const f32x4 *>(src[mt] + c * 32 + 4 * j);
                a[mt][j] = v;
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
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = wave * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Ts[p * STRIDE + nt * 32 + l31] = acc[mt][nt][r];
        }
    __syncthreads();

    // col2im over the interior's outputs, ox fastest (coalesced NCHW rows)
    const int th = min(TH, H - y0), tw = min(TW, W - x0);
    const int OH = 2 * th, OW = 2 * tw, Ho = 2 * H, Wo = 2 * W;
    const int total = Cout * OH * OW;
    for (int e = tid; e < total; e += 256) {
        const int oxl = e % OW;
        const int q = e / OW;
        const int oyl = q % OH, co = q / OH;
        const int oy = 2 * y0 + oyl, ox = 2 * x0 + oxl;
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
        out[((b * Cout + co) * Ho + oy) * (long long)Wo + ox] = s;
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

}  // namespace vqvae

using namespace vqvae;

extern "C" {

size_t vqvae_conv_packed_bytes(int kind, int Cin, int Cout) {
    ConvGeom g;
    if (Cin < 1 || Cout < 1 || make_geom(kind, 1, 4, 4, Cin, Cout, 0, g) != VQVAE_OK) return 0;
    return packed_floats(g) * sizeof(float);
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
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                       packed, g, total);
    return (int)hipGetLastError();
}

int vqvae_conv_forward_f32(int kind, const float *x, const float *packed, const float *bias, int64_t B,
                           int H, int W, int Cin, int Cout, int flags, float *y, vqvae_stream_t stream) {
    if (!x || !packed || !y) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return VQVAE_ERR_SHAPE;
    if (Cin % 4) return VQVAE_ERR_UNSUPPORTED;          // float4 activation loads
    if (B * (int64_t)H * W * 4 > (int64_t)INT32_MAX * 4 || B > INT32_MAX) return VQVAE_ERR_OVERFLOW;
    ConvGeom g;
    int rc = make_geom(kind, B, H, W, Cin, Cout, flags, g);
    if (rc != VQVAE_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = (long long)B * g.Hg * g.Wg;
    prof_begin(VQVAE_PROF_CONV_IGEMM, st);
    // tile shape: 32 pixels x 128 channels per wave when Cout fills it (3 waves/SIMD resident),
    // else 64 pixels x 64 / 32 channels
    if (g.ntile % 4 == 0) {
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

int vqvae_res_layer_forward_f32(const float *x, const float *packed_w1, const float *packed_w2, int64_t B,
                                int H, int W, int C, int Rh, int flags, float *y, vqvae_stream_t stream) {
    if (!x || !packed_w1 || !packed_w2 || !y) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || C < 1 || Rh < 1) return VQVAE_ERR_SHAPE;
    if (Rh > 32 || C % 4 || !(C == 32 || C == 64 || C == 128)) return VQVAE_ERR_UNSUPPORTED;
    if (x == y) return VQVAE_ERR_UNSUPPORTED;           // 3x3 halo: not in place
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = (long long)B * H * W;
    const unsigned gx = (unsigned)((M + 255) / 256);
    prof_begin(VQVAE_PROF_RES_LAYER, st);
    switch (C / 32) {
        case 1: hipLaunchKernelGGL((res_layer_kernel<1>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
        case 2: hipLaunchKernelGGL((res_layer_kernel<2>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
        case 4: hipLaunchKernelGGL((res_layer_kernel<4>), dim3(gx), dim3(256), 0, st, x, packed_w1, packed_w2, y, (int)B, H, W, C, flags); break;
    }
    prof_end(VQVAE_PROF_RES_LAYER, st);
    return (int)hipGetLastError();
}

size_t vqvae_conv_in_packed_bytes(int Cin, int Cout) {
    if (!(Cin == 1 || Cin == 3 || Cin == 4) || Cout < 1 || Cout > 128) return 0;
    const int S = Cin * 8, JG = (S + 3) / 4;
    return (size_t)((Cout + 31) / 32) * JG * 256 * sizeof(float);
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
    return (int)hipGetLastError();
}

int vqvae_conv_in_forward_f32(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H,
                              int W, int Cin, int Cout, int flags, float *y, vqvae_stream_t stream) {
    if (!x_nchw || !packed || !y) return VQVAE_ERR_NULL;
    if (B < 1 || H < 2 || W < 2) return VQVAE_ERR_SHAPE;
    if (H % 2 || W % 2 || vqvae_conv_in_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = B * (long long)(H / 2) * (W / 2);
    const unsigned gx = (unsigned)((M + 255) / 256);
    const int ntile = (Cout + 31) / 32;
    prof_begin(VQVAE_PROF_CONV_IN, st);
#define CI_LAUNCH(CIN_, NT_) hipLaunchKernelGGL((conv_in_kernel<CIN_, NT_>), dim3(gx), dim3(256), 0, st, x_nchw, packed, bias, y, (int)B, H, W, Cout, flags)
#define CI_NT(CIN_)                                                       \
    switch (ntile) {                                                      \
        case 1: CI_LAUNCH(CIN_, 1); break;                                \
        case 2: CI_LAUNCH(CIN_, 2); break;                                \
        case 3: CI_LAUNCH(CIN_, 3); break;                                \
        default: CI_LAUNCH(CIN_, 4); break;                               \
    }
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

size_t vqvae_convt_out_packed_bytes(int Cin, int Cout) {
    if (Cin < 4 || Cin % 4 || Cin > 256 || Cout < 1 || Cout > 4) return 0;
    const int ntile = (16 * Cout + 31) / 32;
    return (size_t)((Cin + 31) / 32) * ntile * 1024 * sizeof(float);
}

int vqvae_convt_out_pack_f32(const float *w, int Cin, int Cout, float *packed, vqvae_stream_t stream) {
    if (!w || !packed) return VQVAE_ERR_NULL;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(convt_out_pack_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed,
                       Cin, Cout, (16 * Cout + 31) / 32);
    return (int)hipGetLastError();
}

int vqvae_convt_out_forward_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                                int Cin, int Cout, float *y_nchw, vqvae_stream_t stream) {
    if (!x || !packed || !y_nchw) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1) return VQVAE_ERR_SHAPE;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int halo_y = H > 16, halo_x = W > 16;
    const int TH = halo_y ? 14 : H, TW = halo_x ? 14 : W;
    const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
    const long long ntiles = B * (long long)tiles_y * tiles_x;
    if (ntiles > INT32_MAX) return VQVAE_ERR_OVERFLOW;
    const int ntile = (16 * Cout + 31) / 32, cpt = (Cin + 31) / 32;
    const size_t lds = ((size_t)cpt * ntile * 1024 + 256 * (ntile * 32 + 1)) * sizeof(float);
    prof_begin(VQVAE_PROF_CONV_OUT, st);
    if (ntile == 1) {
        auto k = convt_out_kernel<1>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(256), lds, st, x, packed, bias, y_nchw, (int)B, H, W, Cin,
                           Cout, TH, TW, halo_y, halo_x, tiles_y, tiles_x);
    } else {
        auto k = convt_out_kernel<2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        hipLaunchKernelGGL(k, dim3((unsigned)ntiles), dim3(256), lds, st, x, packed, bias, y_nchw, (int)B, H, W, Cin,
                           Cout, TH, TW, halo_y, halo_x, tiles_y, tiles_x);
    }
    prof_end(VQVAE_PROF_CONV_OUT, st);
    return (int)hipGetLastError();
}

int vqvae_transpose_f32(const float *x, int64_t batch, int R, int Cc, float *y, vqvae_stream_t stream) {
    if (!x || !y) return VQVAE_ERR_NULL;
    if (batch < 1 || R < 1 || Cc < 1) return VQVAE_ERR_SHAPE;
    if (batch > 65535) return VQVAE_ERR_OVERFLOW;
    hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, R, Cc);
    return (int)hipGetLastError();
}

}  // extern "C"
