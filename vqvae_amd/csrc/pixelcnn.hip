// Small kernels of the GatedPixelCNN prior (pixelcnn/models.py), gfx950.  The masked
// convolutions run as  im2col over their (causal) tap list  ->  the 1x1 split-bf16 GEMM of conv.hip, so the only
// kernels needed here are memory-bound glue on row-major (B,H,W,C) activations:
//   vqvae_gather_rows_f32        out[i][:] = table[idx[i]][:]                      nn.Embedding (:119-121, :68)
//   vqvae_im2col_rows_f32        out[b,y,x,t*C + c] = x[b, y+dy_t, x+dx_t, c] or 0  the taps of a masked conv
//   vqvae_gated_activation_f32   out = tanh(a) * sigmoid(g),  (a|g) = t1 [+ t2] + cond[b]    GatedActivation (:21-27)
//                                with the class-conditional embedding added as in :71, :77
//   vqvae_add_f32                out = a + b                                        the horizontal residual (:79)
#include "common.h"

namespace vqvae {

__global__ __launch_bounds__(256) void gather_rows_kernel(const long long *__restrict__ idx, const float *__restrict__ table,
                                                          long long n, int C, int rows, float *__restrict__ out) {
    const long long total = n * (C >> 2);
    const int c4 = C >> 2;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / c4;
        const int q = (int)(e - i * c4);
        long long k = idx[i];
        k = k < 0 ? 0 : (k >= rows ? rows - 1 : k);
        reinterpret_cast<f32x4 *>(out)[e] = reinterpret_cast<const f32x4 *>(table + (size_t)k * C)[q];
    }
}

struct TapList {
    signed char dy[32], dx[32];
    int n;
};

__global__ __launch_bounds__(256) void im2col_rows_kernel(const float *__restrict__ x, long long B, int H, int W, int C,
                                                          TapList taps, float *__restrict__ out) {
    const int c4 = C >> 2;
    const long long per_px = (long long)taps.n * c4;
    const long long total = B * H * W * per_px;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long px = e / per_px;
        const int r = (int)(e - px * per_px);
        const int t = r / c4, q = r - t * c4;
        const long long b = px / (H * W);
        const int rem = (int)(px - b * H * W);
        const int y = rem / W + taps.dy[t], xx = rem % W + taps.dx[t];
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (y >= 0 && y < H && xx >= 0 && xx < W)
            v = reinterpret_cast<const f32x4 *>(x + ((size_t)(b * H + y) * W + xx) * C)[q];
        reinterpret_cast<f32x4 *>(out)[e] = v;
    }
}

__global__ __launch_bounds__(256) void gated_activation_kernel(const float *__restrict__ t1, const float *__restrict__ t2,
                                                               const float *__restrict__ cond, long long B, int HW,
                                                               int dim, float *__restrict__ out) {
    const long long total = B * HW * dim;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long px = e / dim;
        const int c = (int)(e - px * dim);
        const long long b = px / HW;
        const size_t i = (size_t)px * 2 * dim + c;
        float a = t1[i], g = t1[i + dim];
        if (t2) { a = a + t2[i]; g = g + t2[i + dim]; }      // (v2h + h_horiz) first, then + h: the reference's order (:77)
        if (cond) { a = a + cond[(size_t)b * 2 * dim + c]; g = g + cond[(size_t)b * 2 * dim + dim + c]; }
        out[e] = tanhf(a) * (1.0f / (1.0f + expf(-g)));
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, const float *__restrict__ b, long long n4,
                                                  float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        reinterpret_cast<f32x4 *>(out)[i] = reinterpret_cast<const f32x4 *>(a)[i] + reinterpret_cast<const f32x4 *>(b)[i];
}

static unsigned grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace vqvae

using namespace vqvae;

extern "C" {

int vqvae_gather_rows_f32(const int64_t *idx, const float *table, int64_t n, int C, int rows, float *out,
                          vqvae_stream_t stream) {
    if (!idx || !table || !out) return VQVAE_ERR_NULL;
    if (n < 1 || C < 1 || rows < 1) return VQVAE_ERR_SHAPE;
    if (C % 4 || ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) & 15)) return VQVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * (C / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long *>(idx), table, (long long)n, C, rows, out);
    return (int)hipGetLastError();
}

int vqvae_im2col_rows_f32(const float *x, int64_t B, int H, int W, int C, int ntaps, const int8_t *dy, const int8_t *dx,
                          float *out, vqvae_stream_t stream) {
    if (!x || !dy || !dx || !out) return VQVAE_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || C < 1 || ntaps < 1) return VQVAE_ERR_SHAPE;
    if (ntaps > 32 || C % 4 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15))
        return VQVAE_ERR_UNSUPPORTED;
    TapList t;
    t.n = ntaps;
    for (int i = 0; i < 32; ++i) { t.dy[i] = i < ntaps ? dy[i] : 0; t.dx[i] = i < ntaps ? dx[i] : 0; }   // HOST arrays
    hipLaunchKernelGGL(im2col_rows_kernel, dim3(grid_for(B * H * W * ntaps * (C / 4))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, (long long)B, H, W, C, t, out);
    return (int)hipGetLastError();
}

int vqvae_gated_activation_f32(const float *t1, const float *t2, const float *cond, int64_t B, int HW, int dim, float *out,
                               vqvae_stream_t stream) {
    if (!t1 || !out) return VQVAE_ERR_NULL;
    if (B < 1 || HW < 1 || dim < 1) return VQVAE_ERR_SHAPE;
    hipLaunchKernelGGL(gated_activation_kernel, dim3(grid_for(B * HW * dim)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       t1, t2, cond, (long long)B, HW, dim, out);
    return (int)hipGetLastError();
}

int vqvae_add_f32(const float *a, const float *b, int64_t n, float *out, vqvae_stream_t stream) {
    if (!a || !b || !out) return VQVAE_ERR_NULL;
    if (n < 1) return VQVAE_ERR_SHAPE;
    if (n % 4 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15))
        return VQVAE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
                       (long long)(n / 4), out);
    return (int)hipGetLastError();
}

}  // extern "C"
