// First and last layer (NCHW image in / out: models/encoder.py:29-31, models/decoder.py:34-35) and their weight packing; split out
// of conv.hip in round 4 (shared device code: conv_device.h).
#include "conv_host.h"

namespace vqvae {
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
                        if (relu_out) v = relu1(v);
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
                        if (ro) { v0 = relu1(v0); v1 = relu1(v1); }
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
                    if (relu_out) v = relu1(v);
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

}  // namespace vqvae

using namespace vqvae;

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
    conv_wscale_launch(w, Cin, Cout, 16, 0, ntile, hdr, st);
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
    conv_wscale_launch(w, Cin, Cout, 16, 1, 1, hdr, st);
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
