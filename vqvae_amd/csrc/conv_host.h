// Host-side pieces shared by the conv translation units: layer geometry (make_geom), sizes / offsets inside a packed weight image.
#pragma once
#include "conv_device.h"

namespace vqvae {
// ---------------------------------------------------------------------------
// kind VQVAE_CONV_TAPS: a stride-1 convolution over an explicit tap list (the masked convolutions of the GatedPixelCNN prior,
// pixelcnn/models.py:45-58, without an im2col pass): the list is handed to make_geom by the vqvae_conv_taps_* entry points
// through this thread-local slot for the duration of the call.
struct TapSpec {
    int n;
    signed char dy[16], dx[16];
};
inline thread_local const TapSpec *t_taps = nullptr;
struct TapScope {
    explicit TapScope(const TapSpec *t) { t_taps = t; }
    ~TapScope() { t_taps = nullptr; }
};

inline int make_geom(int kind, long long B, int H, int W, int Cin, int Cout, int flags, ConvGeom &g) {
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

inline size_t packed_floats(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 1024;
}
// split-bf16 image: 3 terms x 32x32 bf16 per (phase, chunk, n_tile) = 6 KiB
inline size_t packed_bf3_bytes(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 3072 * sizeof(unsigned short);
}
// two-term fp16 image: 2 terms x 32x32 fp16 per (phase, chunk, n_tile) = 4 KiB, behind the header of conv_wscale_kernel
// (64 ints + one float and one int per output channel of the ntile 32-channel tiles)
inline size_t h2_header_bytes(int ntile) { return 256 + (size_t)ntile * 256; }
inline size_t packed_h2_bytes(const ConvGeom &g) {
    return (size_t)g.nphase * g.ntaps * g.cpt * g.ntile * 2048 * sizeof(unsigned short);
}
// conv_halo8_h2_kernel: stride-1-sampled layers on pixel grids that are multiples of 8 both ways and larger than one tile
inline bool conv_halo8_ok(const ConvGeom &g, int Cin, int flags) {
    return !(flags & (VQVAE_CONV_BF16_SPLIT | VQVAE_CONV_EXACT_FP32)) && g.istride == 1 && g.Hg == g.Hin && g.Wg == g.Win &&
           g.Hg % 8 == 0 && g.Wg % 8 == 0 && g.Hg * g.Wg > 64 && Cin % 32 == 0 && g.ntile % 2 == 0 && (g.ntaps == 1 || g.ntaps == 4 || g.ntaps == 9) &&
           (long long)g.Hin * g.Win * Cin * 4 < 0x7FFFFFF0ll;
}
// byte offset of the header from the start of a layer's packed weights
inline size_t packed_h2_offset(const ConvGeom &g, int kind) {
    return packed_floats(g) * sizeof(float) + packed_bf3_bytes(g) * (kind == VQVAE_CONV_4x4_S2 ? 2 : 1);
}


// kernels of conv.hip that the other units launch through these (a kernel is launched from the unit that defines it)
void conv_wscale_launch(const float *w, int Cin, int Cout, int kk, int transposed, int ntile, int *hdr, hipStream_t st);
}  // namespace vqvae
