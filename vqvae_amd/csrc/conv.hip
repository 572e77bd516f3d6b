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
    int kk;                     // kh*kw
    int transposed;             // weight is (Cin,Cout,kh,kw)
};

constexpr int kFlagReluIn = 1, kFlagReluOut = 2;

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
template <int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float *__restrict__ in,
                                                         const float *__restrict__ wimg,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ out, ConvGeom g) {
    __shared__ __attribute__((aligned(16))) float Bs[2][NT * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int phase = blockIdx.y % g.nphase, nb = blockIdx.y / g.nphase;
    const long long M = (long long)g.B * g.Hg * g.Wg;
    const int nchunk = g.ntaps * g.cpt;
    const bool relu_in = g.flags & kFlagReluIn;

    bool valid[MT];
    int gy[MT], gx[MT];
    long long bimg[MT];
    long long myoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT) + mt * 32 + l31;
        valid[mt] = p < M;
        const long long pc = valid[mt] ? p : 0;
        const long long b = pc / ((long long)g.Hg * g.Wg);
        const int rem = (int)(pc - b * g.Hg * g.Wg);
        gy[mt] = rem / g.Wg;
        gx[mt] = rem - gy[mt] * g.Wg;
        bimg[mt] = b * g.Hin;
        myoff[mt] = valid[mt] ? ((b * g.Hout + gy[mt] * g.ostride + g.opy[phase]) * g.Wout +
                                 gx[mt] * g.ostride + g.opx[phase]) * (long long)g.Cout
                              : -1;
    }
    const float *wbase = wimg + ((size_t)phase * nchunk * g.ntile + (size_t)nb * NT) * 1024;
    const size_t wchunk = (size_t)g.ntile * 1024;

    f32x4 a_cur[MT][4], a_nxt[MT][4], b_nxt[NT];
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    auto load_a = [&](int c, f32x4(&dst)[MT][4]) {
        const int tap = c / g.cpt, cc = c - tap * g.cpt;
        const int dy = g.dy[phase][tap], dx = g.dx[phase][tap];
        const int ch0 = cc * 32 + 16 * h;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int iy = gy[mt] * g.istride + dy, ix = gx[mt] * g.istride + dx;
            const bool ok = valid[mt] && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            const float *src = in + ((bimg[mt] + iy) * g.Win + ix) * (long long)g.Cin + ch0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (ok && ch0 + 4 * j < g.Cin) v = *reinterpret_cast<const f32x4 *>(src + 4 * j);
                dst[mt][j] = relu_in ? relu4(v) : v;
            }
        }
    };
    auto load_b = [&](int c) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(wbase + (size_t)c * wchunk);
#pragma unroll
        for (int q = 0; q < NT; ++q) b_nxt[q] = src[tid + 256 * q];
    };
    auto store_b = [&](int buf) {
        f32x4 *dst = reinterpret_cast<f32x4 *>(Bs[buf]);
#pragma unroll
        for (int q = 0; q < NT; ++q) dst[tid + 256 * q] = b_nxt[q];
    };

    load_a(0, a_cur);
    load_b(0);
    store_b(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) {
            load_a(c + 1, a_nxt);
            load_b(c + 1);
        }
        const f32x4 *bs = reinterpret_cast<const f32x4 *>(Bs[c & 1]);
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
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][j][i], b4[nt][i],
                                                                           acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            store_b((c + 1) & 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) a_cur[mt][j] = a_nxt[mt][j];
        }
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
__global__ __launch_bounds__(256) void res_layer_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ w1img,
                                                        const float *__restrict__ w2img,
                                                        float *__restrict__ out, int B, int H, int W,
                                                        int C, int flags) {
    constexpr int MT = 2;
    __shared__ __attribute__((aligned(16))) float Bs[2][1024];
    __shared__ __attribute__((aligned(16))) float W2s[NT2 * 1024];
    __shared__ float Hs[4][MT][32 * 33];
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

    bool valid[MT];
    int gy[MT], gx[MT];
    long long bimg[MT];
    const long long wbase = (long long)blockIdx.x * (128 * MT) + wave * (32 * MT);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long p = wbase + mt * 32 + l31;
        valid[mt] = p < M;
        const long long pc = valid[mt] ? p : 0;
        const long long b = pc / ((long long)H * W);
        const int rem = (int)(pc - b * H * W);
        gy[mt] = rem / W;
        gx[mt] = rem - gy[mt] * W;
        bimg[mt] = b * H;
    }

    f32x4 a_cur[MT][4], a_nxt[MT][4], b_nxt;
    f32x16 acc1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;

    auto load_a = [&](int c, f32x4(&dst)[MT][4]) {
        const int tap = c / cpt, cc = c - tap * cpt;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int ch0 = cc * 32 + 16 * h;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int iy = gy[mt] + dy, ix = gx[mt] + dx;
            const bool ok = valid[mt] && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float *src = in + ((bimg[mt] + iy) * W + ix) * (long long)C + ch0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (ok && ch0 + 4 * j < C) v = *reinterpret_cast<const f32x4 *>(src + 4 * j);
                dst[mt][j] = relu_in ? relu4(v) : v;
            }
        }
    };

    load_a(0, a_cur);
    b_nxt = reinterpret_cast<const f32x4 *>(w1img)[tid];
    reinterpret_cast<f32x4 *>(Bs[0])[tid] = b_nxt;
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const bool more = c + 1 < nchunk;
        if (more) {
            load_a(c + 1, a_nxt);
            b_nxt = reinterpret_cast<const f32x4 *>(w1img + (size_t)(c + 1) * 1024)[tid];
        }
        const f32x4 *bs = reinterpret_cast<const f32x4 *>(Bs[c & 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b4 = bs[(j * 2 + h) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt][j][i], b4[i], acc1[mt], 0, 0, 0);
        }
        if (more) {
            reinterpret_cast<f32x4 *>(Bs[(c + 1) & 1])[tid] = b_nxt;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) a_cur[mt][j] = a_nxt[mt][j];
        }
        __syncthreads();
    }

    // hidden tile: relu, accumulator layout -> [pixel][hidden] in LDS (stride 33: conflict-free)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
            Hs[wave][mt][prow * 33 + l31] = fmaxf(acc1[mt][r], 0.0f);
        }
    __syncthreads();
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
// Last layer: nn.ConvTranspose2d(Cin, COUT<=4, k=4, s=2, p=1), row-major in, NCHW image out
// (models/decoder.py:34-35).  3 output channels cannot feed a 32-wide MFMA tile, so this one
// is a VALU kernel: one lane per input-grid pixel produces its 2x2 output pixels x COUT from
// the 3x3 input neighbourhood; weights are wave-uniform (scalar loads).
// Packed weights: [tap 9][ci][phase 4][COUT] with zeros where a phase does not use a tap.
template <int COUT>
__global__ __launch_bounds__(256) void convt_out_kernel(const float *__restrict__ in,
                                                        const float *__restrict__ wp,
                                                        const float *__restrict__ bias,
                                                        float *__restrict__ out, int B, int H, int W,
                                                        int Cin) {
    const long long M = (long long)B * H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool valid = p < M;
    const long long pc = valid ? p : 0;
    const long long b = pc / ((long long)H * W);
    const int rem = (int)(pc - b * H * W);
    const int gy = rem / W, gx = rem - gy * W;
    float acc[4][COUT];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[ph][co] = bias ? bias[co] : 0.0f;
    // phase (py,px) uses input rows {gy, gy-1} (py=0) or {gy+1, gy} (py=1); same for columns
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const int iy = gy + dy, ix = gx + dx;
        const bool ok = valid && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float *src = in + ((b * H + iy) * (long long)W + ix) * Cin;
        const float *wt = wp + (size_t)t * Cin * 4 * COUT;
        for (int c4 = 0; c4 < Cin; c4 += 4) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (ok) v = *reinterpret_cast<const f32x4 *>(src + c4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xv = v[q];
                const float *wq = wt + (size_t)(c4 + q) * 4 * COUT;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int py = ph >> 1, px = ph & 1;
                    const bool uses = (dy == 0 || (dy == -1 && py == 0) || (dy == 1 && py == 1)) &&
                                      (dx == 0 || (dx == -1 && px == 0) || (dx == 1 && px == 1));
                    if (uses) {
#pragma unroll
                        for (int co = 0; co < COUT; ++co)
                            acc[ph][co] = __builtin_fmaf(xv, wq[ph * COUT + co], acc[ph][co]);
                    }
                }
            }
        }
    }
    if (valid) {
        const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float2 v;
                v.x = acc[py * 2 + 0][co];
                v.y = acc[py * 2 + 1][co];
                *reinterpret_cast<float2 *>(out + ((b * COUT + co) * Ho + 2 * gy + py) * (long long)Wo + 2 * gx) = v;
            }
    }
}

__global__ __launch_bounds__(256) void convt_out_pack_kernel(const float *__restrict__ w, float *__restrict__ wp,
                                                             int Cin, int Cout) {
    // w: (Cin, Cout, 4, 4).  oy = 2*iy - 1 + ky  ->  for tap dy: py=0: dy=0->ky=1, dy=-1->ky=3;
    //                                                           py=1: dy=+1->ky=0, dy=0->ky=2
    const int total = 9 * Cin * 4 * Cout;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int co = e % Cout;
        int t = e / Cout;
        const int ph = t & 3; t >>= 2;
        const int ci = t % Cin, tap = t / Cin;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1, py = ph >> 1, px = ph & 1;
        const int ky = py == 0 ? (dy == 0 ? 1 : (dy == -1 ? 3 : -1)) : (dy == 1 ? 0 : (dy == 0 ? 2 : -1));
        const int kx = px == 0 ? (dx == 0 ? 1 : (dx == -1 ? 3 : -1)) : (dx == 1 ? 0 : (dx == 0 ? 2 : -1));
        wp[e] = (ky >= 0 && kx >= 0) ? w[((ci * Cout + co) * 4 + ky) * 4 + kx] : 0.0f;
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
    if (Cin < 4 || Cin % 4 || Cout < 1 || Cout > 4) return 0;
    return (size_t)9 * Cin * 4 * Cout * sizeof(float);
}

int vqvae_convt_out_pack_f32(const float *w, int Cin, int Cout, float *packed, vqvae_stream_t stream) {
    if (!w || !packed) return VQVAE_ERR_NULL;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(convt_out_pack_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed,
                       Cin, Cout);
    return (int)hipGetLastError();
}

int vqvae_convt_out_forward_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                                int Cin, int Cout, float *y_nchw, vqvae_stream_t stream) {
    if (!x || !packed || !y_nchw) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1) return VQVAE_ERR_SHAPE;
    if (vqvae_convt_out_packed_bytes(Cin, Cout) == 0) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long M = B * (long long)H * W;
    const unsigned gx = (unsigned)((M + 255) / 256);
    prof_begin(VQVAE_PROF_CONV_OUT, st);
    switch (Cout) {
        case 1: hipLaunchKernelGGL((convt_out_kernel<1>), dim3(gx), dim3(256), 0, st, x, packed, bias, y_nchw, (int)B, H, W, Cin); break;
        case 2: hipLaunchKernelGGL((convt_out_kernel<2>), dim3(gx), dim3(256), 0, st, x, packed, bias, y_nchw, (int)B, H, W, Cin); break;
        case 3: hipLaunchKernelGGL((convt_out_kernel<3>), dim3(gx), dim3(256), 0, st, x, packed, bias, y_nchw, (int)B, H, W, Cin); break;
        case 4: hipLaunchKernelGGL((convt_out_kernel<4>), dim3(gx), dim3(256), 0, st, x, packed, bias, y_nchw, (int)B, H, W, Cin); break;
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
