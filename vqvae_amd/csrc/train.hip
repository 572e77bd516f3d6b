// Training-step companions of the forward path, gfx950.
//
//   vqvae_vq_backward_f32      gradients of VectorQuantizer.forward as autograd derives them from
//                              models/quantizer.py:63-67:
//                                dL/dz   = g_zq + g_loss * 2 (z - e_idx) / (N D)         (:63 first term, :67)
//                                dL/dE_k = g_loss * 2 beta * sum_{i: idx_i = k} (e_k - z_i) / (N D)   (:63-64)
//                              The codebook gradient is a segmented sum over the rows of each code.  It is
//                              computed WITHOUT floating-point atomics: rows are sorted by code (stable radix
//                              sort, so row order inside a code is ascending), every 512-row chunk of a code is
//                              added in a fixed order in fp64 by one workgroup, and a code's chunks are combined in a
//                              fixed order -- the result is bit-reproducible from run to run, and the work is
//                              proportional to the rows however skewed the code histogram is.
//   vqvae_recon_loss_f32       main.py:75-76 and the three scalars of :81-83 packed into one 3-float buffer
//                              (one D2H copy per step instead of three).
//   vqvae_recon_loss_backward_f32   d/dx_hat of mean((x_hat - x)^2) / var.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace vqvae {

constexpr int kBwdChunk = 512;        // rows per workgroup of the segmented sum (a code owns ceil(count/512) units)
constexpr int kReconGrid = 1024;      // partial sums of the reconstruction loss

struct BwdPlan {
    size_t off_keys, off_keys_out, off_vals, off_vals_out, off_offsets, off_units, off_partials, off_sort, sort_bytes, total;
    long long max_units;
    int key_bits;
};

static BwdPlan bwd_plan(long long N, int K, int D) {
    BwdPlan p;
    p.key_bits = 1;
    while ((1 << p.key_bits) < K) ++p.key_bits;
    p.off_keys = 0;
    p.off_keys_out = align_up(p.off_keys + (size_t)N * 4, 256);
    p.off_vals = align_up(p.off_keys_out + (size_t)N * 4, 256);
    p.off_vals_out = align_up(p.off_vals + (size_t)N * 4, 256);
    p.off_offsets = align_up(p.off_vals_out + (size_t)N * 4, 256);
    p.off_units = align_up(p.off_offsets + (size_t)(K + 1) * 4, 256);
    p.max_units = N / kBwdChunk + K;                       // sum_k ceil(count_k / chunk) <= N/chunk + K
    p.off_partials = align_up(p.off_units + (size_t)(K + 1) * 4, 256);
    p.off_sort = align_up(p.off_partials + (size_t)p.max_units * D * sizeof(double), 256);
    size_t sb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, (const unsigned *)nullptr, (unsigned *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, (int)N, 0, p.key_bits, 0);
    p.sort_bytes = sb;
    p.total = align_up(p.off_sort + sb, 256);
    return p;
}

__global__ __launch_bounds__(256) void vqb_keys_kernel(const long long *__restrict__ idx, long long N, int K,
                                                       unsigned *__restrict__ keys, int *__restrict__ vals) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const long long k = idx[i];
        keys[i] = (unsigned)(k < 0 ? 0 : (k >= K ? K - 1 : k));
        vals[i] = (int)i;
    }
}

// offsets[k] = first sorted position whose key is >= k (k = 0..K): one binary search per code
__global__ __launch_bounds__(256) void vqb_offsets_kernel(const unsigned *__restrict__ keys, long long N, int K,
                                                          int *__restrict__ offsets) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k > K) return;
    long long lo = 0, hi = N;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned)k) lo = mid + 1; else hi = mid;
    }
    offsets[k] = (int)lo;
}

// unit_start[k] = number of work units (chunks of kBwdChunk sorted rows) owned by codes < k; one block scans K+1
__global__ __launch_bounds__(1024) void vqb_units_kernel(const int *__restrict__ offsets, int K, int *__restrict__ unit_start) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (K + 1023) / 1024;
    int local = 0;
    for (int j = 0; j < per; ++j) {
        const int k = tid * per + j;
        if (k < K) local += (offsets[k + 1] - offsets[k] + kBwdChunk - 1) / kBwdChunk;
    }
    part[tid] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;                             // exclusive prefix of this thread's codes
    for (int j = 0; j < per; ++j) {
        const int k = tid * per + j;
        if (k < K) {
            unit_start[k] = run;
            run += (offsets[k + 1] - offsets[k] + kBwdChunk - 1) / kBwdChunk;
        }
    }
    if (tid == 1023) unit_start[K] = part[1023];
}

// partial[unit][c] = sum over the unit's (<= kBwdChunk, sorted) rows of z_i[c]   (fp64, fixed order).  Work is
// proportional to the rows, however skewed the histogram (a freshly initialised codebook uses a handful of codes).
__global__ __launch_bounds__(256) void vqb_segsum_kernel(const float *__restrict__ z, const int *__restrict__ rows,
                                                         const int *__restrict__ offsets,
                                                         const int *__restrict__ unit_start, int K, int D, int HW,
                                                         int rowmajor, double *__restrict__ partial) {
    __shared__ double red[256];
    const int unit = blockIdx.x, tid = threadIdx.x;
    if (unit >= unit_start[K]) return;
    int lo_k = 0, hi_k = K;                                  // last k with unit_start[k] <= unit
    while (hi_k - lo_k > 1) {
        const int mid = (lo_k + hi_k) >> 1;
        if (unit_start[mid] <= unit) lo_k = mid; else hi_k = mid;
    }
    const int k = lo_k;
    const int a = offsets[k] + (unit - unit_start[k]) * kBwdChunk;
    const int b = a + kBwdChunk < offsets[k + 1] ? a + kBwdChunk : offsets[k + 1];
    // D <= 256: threads [0, G*D) are G row groups of D channels each
    const int G = 256 / D;
    const int g = tid / D, c = tid - g * D;
    double acc = 0.0;
    if (g < G) {
        auto zat = [&](long long r) {
            if (rowmajor) return z[(size_t)r * D + c];
            const long long bb = r / HW;
            const int hw = (int)(r - bb * HW);
            return z[((size_t)bb * D + c) * HW + hw];
        };
        int j = a + g;
        for (; j + 3 * G < b; j += 4 * G) {                 // four rows' (index, value) loads in flight; same summation order
            long long r[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = rows[j + q * G];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = zat(r[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += (double)v[q];
        }
        for (; j < b; j += G) acc += (double)zat(rows[j]);
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < D) {
        double t = 0.0;
        for (int q = 0; q < G; ++q) t += red[q * D + tid];
        partial[(size_t)unit * D + tid] = t;
    }
}

__global__ __launch_bounds__(256) void vqb_codebook_grad_kernel(const float *__restrict__ cb,
                                                                const int *__restrict__ offsets,
                                                                const int *__restrict__ unit_start,
                                                                const double *__restrict__ partial,
                                                                const float *__restrict__ g_loss, int K, int D,
                                                                double scale, float *__restrict__ grad_cb) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)K * D) return;
    const int k = (int)(e / D), c = (int)(e - (long long)k * D);
    // four interleaved running sums (a code that owns many rows has many units: their loads are in flight together
    // instead of one dependent add per load), combined in a fixed order
    double z4[4] = {0.0, 0.0, 0.0, 0.0};
    int u = unit_start[k];
    const int u1 = unit_start[k + 1];
    for (; u + 4 <= u1; u += 4)
#pragma unroll
        for (int j = 0; j < 4; ++j) z4[j] += partial[(size_t)(u + j) * D + c];
    for (int j = 0; u < u1; ++u, ++j) z4[j] += partial[(size_t)u * D + c];
    const double zsum = (z4[0] + z4[1]) + (z4[2] + z4[3]);
    const double cnt = (double)(offsets[k + 1] - offsets[k]);
    const double gl = g_loss ? (double)g_loss[0] : 1.0;
    grad_cb[e] = (float)(gl * scale * (cnt * (double)cb[e] - zsum));
}

// grad_z = g_zq + g_loss * s * (z - e_idx); one thread per element, row-major or NCHW addressing
__global__ __launch_bounds__(256) void vqb_gradz_kernel(const float *__restrict__ z, const float *__restrict__ cb,
                                                        const long long *__restrict__ idx,
                                                        const float *__restrict__ g_zq,
                                                        const float *__restrict__ g_loss, long long total, int D,
                                                        int HW, int rowmajor, float scale,
                                                        float *__restrict__ grad_z) {
    const float gs = (g_loss ? g_loss[0] : 1.0f) * scale;
    if (rowmajor < 0) {
        // row-major rows with D % 4 == 0 and 16-byte aligned tensors (the launch checks): four channels per thread
        for (long long e4 = (long long)blockIdx.x * 256 + threadIdx.x; e4 < (total >> 2); e4 += (long long)gridDim.x * 256) {
            const long long e = e4 << 2, row = e / D;
            const int c = (int)(e - row * D);
            const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + e);
            const f32x4 ev = *reinterpret_cast<const f32x4 *>(cb + (size_t)idx[row] * D + c);
            f32x4 o;
            o.x = gs * (zv.x - ev.x); o.y = gs * (zv.y - ev.y); o.z = gs * (zv.z - ev.z); o.w = gs * (zv.w - ev.w);
            if (g_zq) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(g_zq + e);
                o.x = gv.x + o.x; o.y = gv.y + o.y; o.z = gv.z + o.z; o.w = gv.w + o.w;
            }
            *reinterpret_cast<f32x4 *>(grad_z + e) = o;
        }
        return;
    }
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        long long row;
        int c;
        if (rowmajor) {
            row = e / D;
            c = (int)(e - row * D);
        } else {
            const long long plane = e / HW;          // b*D + c
            const int hw = (int)(e - plane * HW);
            const long long b = plane / D;
            c = (int)(plane - b * D);
            row = b * HW + hw;
        }
        const long long k = idx[row];
        const float d = z[e] - cb[(size_t)k * D + c];
        const float g = gs * d;
        grad_z[e] = g_zq ? g_zq[e] + g : g;
    }
}

__global__ __launch_bounds__(256) void recon_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            long long n, double *__restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    const long long n4 = n >> 2;
    const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a), *b4 = reinterpret_cast<const f32x4 *>(b);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 u = a4[i], v = b4[i];
        const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
        acc += (double)(d0 * d0) + (double)(d1 * d1) + (double)(d2 * d2) + (double)(d3 * d3);
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
            const float d = a[i] - b[i];
            acc += (double)(d * d);
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void recon_final_kernel(const double *__restrict__ partial, int np, long long n,
                                                          float inv_var, const float *__restrict__ embedding_loss,
                                                          const float *__restrict__ perplexity,
                                                          float *__restrict__ out3) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(red[0] / (double)n);          // torch.mean((x_hat - x)**2)   main.py:75
        const float recon = mse * inv_var;                       //   / x_train_var
        out3[0] = recon;
        out3[1] = recon + (embedding_loss ? embedding_loss[0] : 0.0f);   // loss = recon + embedding   :76
        out3[2] = perplexity ? perplexity[0] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void recon_backward_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             long long n, float scale,
                                                             const float *__restrict__ g_loss,
                                                             float *__restrict__ grad) {
    const float s = (g_loss ? g_loss[0] : 1.0f) * scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        grad[i] = s * (a[i] - b[i]);
}

}  // namespace vqvae

using namespace vqvae;

extern "C" {

size_t vqvae_vq_backward_workspace_bytes(int64_t N, int K, int D) {
    if (N < 1 || N > INT32_MAX || K < 1 || K > 16384 || D < 1 || D > 256) return 0;
    return bwd_plan(N, K, D).total;
}

int vqvae_vq_backward_f32(const float *z_e, const float *codebook, const int64_t *idx, const float *grad_zq,
                          const float *grad_loss, int64_t B, int D, int H, int W, int K, float beta, int flags,
                          float *grad_z, float *grad_codebook, void *workspace, size_t workspace_bytes,
                          vqvae_stream_t stream) {
    if (!z_e || !codebook || !idx || (!grad_z && !grad_codebook)) return VQVAE_ERR_NULL;
    if (B < 1 || D < 1 || H < 1 || W < 1 || K < 1) return VQVAE_ERR_SHAPE;
    if (D > 256 || K > 16384) return VQVAE_ERR_UNSUPPORTED;
    const long long HW = (long long)H * W, N = (long long)B * HW;
    if (N > INT32_MAX || N * D > ((long long)1 << 40)) return VQVAE_ERR_OVERFLOW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rowmajor = (flags & VQVAE_VQ_ROWMAJOR) ? 1 : 0;
    const double nd = (double)N * (double)D;
    if (grad_z) {
        const long long total = N * D;
        const bool vec4 = rowmajor && (D & 3) == 0 &&
                          !((reinterpret_cast<uintptr_t>(z_e) | reinterpret_cast<uintptr_t>(codebook) | reinterpret_cast<uintptr_t>(grad_z) |
                             reinterpret_cast<uintptr_t>(grad_zq)) & 15);
        long long grid = ((vec4 ? total >> 2 : total) + 255) / 256;
        if (grid > 65536) grid = 65536;
        hipLaunchKernelGGL(vqb_gradz_kernel, dim3((unsigned)grid), dim3(256), 0, st, z_e, codebook,
                           reinterpret_cast<const long long *>(idx), grad_zq, grad_loss, total, D, (int)HW, vec4 ? -1 : rowmajor,
                           (float)(2.0 / nd), grad_z);
    }
    if (grad_codebook) {
        const BwdPlan p = bwd_plan(N, K, D);
        if (!workspace || workspace_bytes < p.total) return VQVAE_ERR_WORKSPACE;
        char *ws = static_cast<char *>(workspace);
        unsigned *keys = reinterpret_cast<unsigned *>(ws + p.off_keys);
        unsigned *keys_out = reinterpret_cast<unsigned *>(ws + p.off_keys_out);
        int *vals = reinterpret_cast<int *>(ws + p.off_vals);
        int *vals_out = reinterpret_cast<int *>(ws + p.off_vals_out);
        int *offsets = reinterpret_cast<int *>(ws + p.off_offsets);
        double *partial = reinterpret_cast<double *>(ws + p.off_partials);
        long long grid = (N + 255) / 256;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(vqb_keys_kernel, dim3((unsigned)grid), dim3(256), 0, st,
                           reinterpret_cast<const long long *>(idx), N, K, keys, vals);
        size_t sb = p.sort_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(ws + p.off_sort, sb, keys, keys_out, vals, vals_out, (int)N, 0,
                                                          p.key_bits, st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(vqb_offsets_kernel, dim3((unsigned)((K + 1 + 255) / 256)), dim3(256), 0, st, keys_out, N, K,
                           offsets);
        int *unit_start = reinterpret_cast<int *>(ws + p.off_units);
        hipLaunchKernelGGL(vqb_units_kernel, dim3(1), dim3(1024), 0, st, offsets, K, unit_start);
        hipLaunchKernelGGL(vqb_segsum_kernel, dim3((unsigned)p.max_units), dim3(256), 0, st, z_e, vals_out, offsets,
                           unit_start, K, D, (int)HW, rowmajor, partial);
        hipLaunchKernelGGL(vqb_codebook_grad_kernel, dim3((unsigned)(((long long)K * D + 255) / 256)), dim3(256), 0,
                           st, codebook, offsets, unit_start, partial, grad_loss, K, D, 2.0 * (double)beta / nd,
                           grad_codebook);
    }
    return (int)hipGetLastError();
}

size_t vqvae_recon_loss_workspace_bytes(void) { return (size_t)kReconGrid * sizeof(double); }

int vqvae_recon_loss_f32(const float *x_hat, const float *x, int64_t n, float inv_var,
                         const float *embedding_loss, const float *perplexity, float *out3, void *workspace,
                         size_t workspace_bytes, vqvae_stream_t stream) {
    if (!x_hat || !x || !out3) return VQVAE_ERR_NULL;
    if (n < 1) return VQVAE_ERR_SHAPE;
    if (!workspace || workspace_bytes < vqvae_recon_loss_workspace_bytes()) return VQVAE_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(x_hat) | reinterpret_cast<uintptr_t>(x)) & 15) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    long long grid = ((n >> 2) + 255) / 256;
    if (grid > kReconGrid) grid = kReconGrid;
    if (grid < 1) grid = 1;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(recon_partial_kernel, dim3((unsigned)grid), dim3(256), 0, st, x_hat, x, (long long)n, partial);
    hipLaunchKernelGGL(recon_final_kernel, dim3(1), dim3(256), 0, st, partial, (int)grid, (long long)n, inv_var,
                       embedding_loss, perplexity, out3);
    return (int)hipGetLastError();
}

int vqvae_recon_loss_backward_f32(const float *x_hat, const float *x, int64_t n, float inv_var,
                                  const float *grad_loss, float *grad_x_hat, vqvae_stream_t stream) {
    if (!x_hat || !x || !grad_x_hat) return VQVAE_ERR_NULL;
    if (n < 1) return VQVAE_ERR_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    long long grid = (n + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(recon_backward_kernel, dim3((unsigned)grid), dim3(256), 0, st, x_hat, x, (long long)n,
                       (float)(2.0 * (double)inv_var / (double)n), grad_loss, grad_x_hat);
    return (int)hipGetLastError();
}

}  // extern "C"
