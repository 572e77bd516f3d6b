/*
 * vqvae_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, scalar, single-threaded CPU restatement of the VQ-VAE forward hot
 * path of the reference (MishaLaskin/vqvae).  It exists to CHECK the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link, import or execute it.  Nothing under vqvae_amd/ may depend on it.
 *
 * Parity pinning: the reference ships no tests or golden vectors for this path
 * (SURVEY.md section 4), so this restatement is pinned against the reference
 * itself, imported and executed in the build container by
 * oracle/gen_golden.py, whose outputs are committed under tests/golden/.
 *
 * Every function cites the reference file:line it follows.  The reference
 * delegates its arithmetic to PyTorch/ATen CPU kernels; where the rounding
 * ORDER of those kernels matters for bit-exactness (the quantizer's distance
 * matrix) the order is restated here and is verified bit-for-bit against live
 * torch by tests/test_oracle.py:
 *
 *   torch.matmul(z, E.t())          == k-ordered fmaf chain, acc starts at 0
 *                                      (valid for D <= 256 on this build)
 *   torch.sum(x**2, dim=1)          == ATen cascade_sum inner-dim order:
 *                                      8-lane vectors, 4-way ILP, lanes summed
 *                                      sequentially (aten/native/cpu/SumKernel)
 *   d = (zz + ee) - 2*m             == fl(fl(zz+ee) - fl(2*m))
 *   argmin                          == first minimal index, NaN counts as min
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VQO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* torch.sum(x, dim=-1) for one contiguous fp32 row of length n, in the exact
 * order ATen's cascade_sum uses for an inner-dimension reduction
 * (vectorized_inner_sum -> row_sum -> multi_row_sum) with Vec = 8 floats and
 * ilp_factor = 4.  For n/8/4 < 16 the cascade degenerates to plain sequential
 * accumulation per partial; larger rows use the 4-level cascade.             */
static void vqo_multi_row_sum8x4(const float *x, int64_t size_ilp, float acc_out[4][8])
{
    /* multi_row_sum<Vec8, nrows=4>: row stride = 4 vectors, col stride = 1 vec */
    enum { NUM_LEVELS = 4 };
    int64_t level_power = 4;
    {
        /* level_power = max(4, ceil_log2(size) / num_levels) */
        int64_t cl2 = 0;
        while (((int64_t)1 << cl2) < size_ilp) cl2++;
        int64_t lp = cl2 / NUM_LEVELS;
        if (lp > level_power) level_power = lp;
    }
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    static _Thread_local float acc[NUM_LEVELS][4][8];
    memset(acc, 0, sizeof(acc));
    int64_t i = 0;
    for (; i + level_step <= size_ilp;) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < 4; ++k)
                for (int t = 0; t < 8; ++t)
                    acc[0][k][t] += x[(i * 4 + k) * 8 + t];
        for (int j = 1; j < NUM_LEVELS; ++j) {
            for (int k = 0; k < 4; ++k)
                for (int t = 0; t < 8; ++t) {
                    acc[j][k][t] += acc[j - 1][k][t];
                    acc[j - 1][k][t] = 0.0f;
                }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 8; ++t)
                acc[0][k][t] += x[(i * 4 + k) * 8 + t];
    for (int j = 1; j < NUM_LEVELS; ++j)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 8; ++t)
                acc[0][k][t] += acc[j][k][t];
    memcpy(acc_out, acc[0], sizeof(acc[0]));
}

VQO_API float vqo_aten_row_sum(const float *x, int64_t n)
{
    if (n < 8) {
        /* shorter than one vector: cascade_sum takes scalar_inner_sum -> row_sum<float> (aten/native/cpu/SumKernel.cpp): four
         * partial sums over elements 4 i + k, leftovers into slot 0, then slots 1..3 into slot 0.  n = 5: ((x0 + x4) + x1 + x2) + x3,
         * NOT the sequential sum (round 5: the widths below 8 were never pinned before; tests/test_oracle.py d = 1 .. 7). */
        float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int64_t si = n / 4;
        for (int64_t i = 0; i < si; ++i)
            for (int k = 0; k < 4; ++k) p[k] += x[i * 4 + k];
        for (int64_t i = si * 4; i < n; ++i) p[0] += x[i];
        for (int k = 1; k < 4; ++k) p[0] += p[k];
        return p[0];
    }
    const int64_t vec_size = n / 8;          /* number of whole 8-float vectors */
    const int64_t size_ilp = vec_size / 4;
    float part[4][8];
    vqo_multi_row_sum8x4(x, size_ilp, part);
    for (int64_t v = size_ilp * 4; v < vec_size; ++v)
        for (int t = 0; t < 8; ++t) part[0][t] += x[v * 8 + t];
    for (int k = 1; k < 4; ++k)
        for (int t = 0; t < 8; ++t) part[0][t] += part[k][t];
    float final_acc = 0.0f;
    for (int64_t k = vec_size * 8; k < n; ++k) final_acc += x[k];
    for (int t = 0; t < 8; ++t) final_acc += part[0][t];
    return final_acc;
}

/* torch.sum(x**2, dim=1) -- models/quantizer.py:49-50 (both the z term and
 * the codebook term).  The square is a separate rounded op (pow), no FMA.    */
VQO_API void vqo_row_sqnorm(const float *x, int64_t rows, int d, float *out)
{
    float *sq = (float *)malloc(sizeof(float) * (size_t)(d > 0 ? d : 1));
    for (int64_t r = 0; r < rows; ++r) {
        for (int c = 0; c < d; ++c) {
            volatile float s = x[r * d + c] * x[r * d + c];
            sq[c] = s;
        }
        out[r] = vqo_aten_row_sum(sq, d);
    }
    free(sq);
}

/* torch.matmul(z_flattened, embedding.weight.t()) element -- quantizer.py:51.
 * Bitwise a c-ordered fmaf chain starting from 0.           */
static inline float vqo_dot_chain(const float *a, const float *b, int d)
{
    float acc = 0.0f;
    for (int c = 0; c < d; ++c) acc = fmaf(a[c], b[c], acc);
    return acc;
}

/* LessOrNan ordering used by torch.argmin (aten/native/cpu/ReduceOpsKernel):
 * NaN is smaller than everything; ties -> lower index.                      */
static inline int vqo_less_or_nan(float a, float b)
{
    if (isnan(a)) return !isnan(b);           /* strictly better only if b not NaN */
    return a < b;
}

/* ------------------------------------------------------------------------ */
/* VectorQuantizer.forward -- models/quantizer.py:29-76.
 *  z_nchw   (B,D,H,W) fp32 contiguous            in
 *  codebook (K,D) fp32 row-major                 in   (embedding.weight, :26)
 *  zq_nchw  (B,D,H,W)                            out  (:67 then :74)
 *  idx      (N) int64, N = B*H*W, row order (b,h,w)   (:54)
 *  hist     (K) int32 counts                     out  (column sums of :55-57)
 *  loss, perplexity  scalars                     out  (:63-64, :70-71)
 *  dist     optional (N,K) distance matrix       out  (:49-51) or NULL
 * Returns 0.                                                                */
VQO_API int vqo_vq_forward(const float *z_nchw, const float *codebook,
                           int64_t B, int D, int H, int W, int K, float beta,
                           float *zq_nchw, int64_t *idx, int32_t *hist,
                           float *loss, float *perplexity, float *dist)
{
    const int64_t HW = (int64_t)H * W;
    const int64_t N = B * HW;
    float *ee = (float *)malloc(sizeof(float) * (size_t)K);
    float *row = (float *)malloc(sizeof(float) * (size_t)D);
    vqo_row_sqnorm(codebook, K, D, ee);                       /* :50 */
    memset(hist, 0, sizeof(int32_t) * (size_t)K);
    double sq_sum = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        const int64_t b = n / HW, hw = n % HW;
        for (int c = 0; c < D; ++c)                           /* :45-46 permute+view */
            row[c] = z_nchw[(b * D + c) * HW + hw];
        float zz;
        vqo_row_sqnorm(row, 1, D, &zz);                       /* :49 */
        int64_t best = 0;
        float best_d = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float m = vqo_dot_chain(row, codebook + (int64_t)k * D, D);   /* :51 */
            volatile float t = zz + ee[k];                    /* :49-50 add  */
            volatile float u = 2.0f * m;                      /* :50 2 * mm  */
            volatile float dk = t - u;                        /* :50 sub     */
            if (dist) dist[n * K + k] = dk;
            if (k == 0 || vqo_less_or_nan(dk, best_d)) { best = k; best_d = dk; }
        }
        idx[n] = best;                                        /* :54 */
        hist[best] += 1;                                      /* :55-57 */
        const float *e = codebook + best * D;                 /* :60 one-hot @ E */
        for (int c = 0; c < D; ++c) {
            volatile float diff = e[c] - row[c];              /* z_q - z      */
            volatile float sq = diff * diff;                  /* :63 (.)**2   */
            sq_sum += (double)sq;
            volatile float st = row[c] + diff;                /* :67 z + (z_q - z) */
            zq_nchw[(b * D + c) * HW + hw] = st;              /* :74 back to NCHW */
        }
    }
    {
        /* :63-64  loss = mean(.) + beta * mean(.) ; the two means are the same
         * number in forward.  The mean itself is order-sensitive only at the
         * 3e-8 level -- accumulated in double here, compared
         * with rtol 1e-6 in the tests.                                        */
        const float mse = (float)(sq_sum / ((double)N * (double)D));
        volatile float bm = beta * mse;
        volatile float l = mse + bm;
        *loss = l;
    }
    {
        /* :70-71 perplexity = exp(-sum(p * log(p + 1e-10))), p = count/N      */
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const float p = (float)hist[k] / (float)N;
            const float lg = logf(p + 1e-10f);
            s += (double)(p * lg);
        }
        *perplexity = expf(-(float)s);
    }
    free(ee);
    free(row);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* min_encoding_indices only, for row-major rows, FAST (round 6): the checker for
 * BASELINE configs 4 / 5 at full size (1.6 M rows x K = 1024; 4.2 M rows x
 * K = 8192, D = 128 -- 4.4 TFLOP, hours for the scalar loop above).  The SAME
 * arithmetic as vqo_vq_forward (models/quantizer.py:45-54): per (row, code) one
 * c-ordered fmaf chain from 0 -- here eight codes per AVX2 register (vfmadd231ps
 * is fmaf in every lane: one rounding), four registers in flight, on a transposed
 * copy of the codebook --, then fl(fl(zz + ee_k) - fl(2 m)) and the first-index /
 * NaN-minimal argmin, scalar and in code order.  tests/test_oracle.py holds it
 * against vqo_vq_forward and the reference's goldens bit for bit.  The caller
 * (oracle/c_oracle.py: vq_indices_rows) spreads row slabs over host threads.     */
#include <immintrin.h>
VQO_API int vqo_vq_indices_rows(const float *z_rows, const float *codebook,
                                int64_t N, int D, int K, int64_t *idx)
{
    const int Kp = (K + 31) & ~31;
    float *et = (float *)aligned_alloc(32, sizeof(float) * (size_t)D * (size_t)Kp);
    float *ee = (float *)malloc(sizeof(float) * (size_t)K);
    float *m = (float *)aligned_alloc(32, sizeof(float) * (size_t)Kp);
    if (!et || !ee || !m) { free(et); free(ee); free(m); return -1; }
    memset(et, 0, sizeof(float) * (size_t)D * (size_t)Kp);
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < D; ++c)                           /* block-major: [k / 32][c][k % 32], a block is 32 x D contiguous floats */
            et[((size_t)(k >> 5) * D + c) * 32 + (k & 31)] = codebook[(size_t)k * D + c];
    vqo_row_sqnorm(codebook, K, D, ee);                       /* :50 */
    /* code blocks outside, rows inside (a block's 32 x D transposed codes stay in L1 across the rows); blocks in ascending order and a
     * strict "better" keep torch.argmin's first-index rule                                                                          */
    float *zz = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    float *bd = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    if (!zz || !bd) { free(et); free(ee); free(m); free(zz); free(bd); return -1; }
    vqo_row_sqnorm(z_rows, N, D, zz);                         /* :49 */
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        const int kend = k0 + 32 < K ? k0 + 32 : K;
        for (int64_t n = 0; n < N; ++n) {
            const float *row = z_rows + n * D;
            __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
            const float *e = et + (size_t)(k0 >> 5) * D * 32;
            for (int c = 0; c < D; ++c, e += 32) {            /* :51 -- every lane its own chain, c ascending */
                const __m256 zb = _mm256_broadcast_ss(row + c);
                a0 = _mm256_fmadd_ps(zb, _mm256_load_ps(e), a0);
                a1 = _mm256_fmadd_ps(zb, _mm256_load_ps(e + 8), a1);
                a2 = _mm256_fmadd_ps(zb, _mm256_load_ps(e + 16), a2);
                a3 = _mm256_fmadd_ps(zb, _mm256_load_ps(e + 24), a3);
            }
            _mm256_store_ps(m, a0);
            _mm256_store_ps(m + 8, a1);
            _mm256_store_ps(m + 16, a2);
            _mm256_store_ps(m + 24, a3);
            int64_t best = k0 ? idx[n] : 0;
            float best_d = k0 ? bd[n] : 0.0f;
            for (int k = k0; k < kend; ++k) {
                volatile float t = zz[n] + ee[k];             /* :49-50 add  */
                volatile float u = 2.0f * m[k - k0];          /* :50 2 * mm  */
                volatile float dk = t - u;                    /* :50 sub     */
                if (k == 0 || vqo_less_or_nan(dk, best_d)) { best = k; best_d = dk; }
            }
            idx[n] = best;                                    /* :54 */
            bd[n] = best_d;
        }
    }
    free(zz); free(bd);
    free(et); free(ee); free(m);
    return 0;
}

/* min_encodings one-hot (N,K) fp32 -- models/quantizer.py:55-57.            */
VQO_API void vqo_onehot(const int64_t *idx, int64_t N, int K, float *onehot)
{
    memset(onehot, 0, sizeof(float) * (size_t)N * (size_t)K);
    for (int64_t n = 0; n < N; ++n) onehot[n * K + idx[n]] = 1.0f;
}

/* idx -> z_q (B,D,H,W): the notebook's generate_samples path
 * (visualization.ipynb:358-365: one-hot @ embedding.weight, view, permute). */
VQO_API void vqo_decode_indices(const int64_t *idx, const float *codebook,
                                int64_t B, int D, int H, int W, float *zq_nchw)
{
    const int64_t HW = (int64_t)H * W;
    for (int64_t n = 0; n < B * HW; ++n) {
        const int64_t b = n / HW, hw = n % HW;
        for (int c = 0; c < D; ++c)
            zq_nchw[(b * D + c) * HW + hw] = codebook[idx[n] * D + c];
    }
}

/* ------------------------------------------------------------------------ */
/* nn.Conv2d forward, NCHW fp32, weight (Cout,Cin,kh,kw) -- the layer type of
 * models/encoder.py:29-36, models/residual.py:20-24, models/vqvae.py:16-17.
 * oneDNN's summation order is opaque, so the oracle pins the
 * correctly-rounded value: accumulate in double, round once.  flags bit0:
 * apply ReLU to the INPUT (in-place nn.ReLU(True) feeding a conv), bit1: ReLU
 * on the output.                                                            */
VQO_API void vqo_conv2d(const float *x, const float *w, const float *bias,
                        int64_t B, int Cin, int H, int W, int Cout, int kh, int kw,
                        int stride, int pad, int flags, float *y)
{
    const int Ho = (H + 2 * pad - kh) / stride + 1;
    const int Wo = (W + 2 * pad - kw) / stride + 1;
    for (int64_t b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < kh; ++ky) {
                            const int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < kw; ++kx) {
                                const int ix = ox * stride - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                float v = x[((b * Cin + ci) * H + iy) * W + ix];
                                if ((flags & 1) && v < 0.0f) v = 0.0f;
                                acc += (double)v *
                                       (double)w[((co * Cin + ci) * kh + ky) * kw + kx];
                            }
                        }
                    float r = (float)acc;
                    if ((flags & 2) && r < 0.0f) r = 0.0f;
                    y[((b * Cout + co) * Ho + oy) * Wo + ox] = r;
                }
}

/* nn.ConvTranspose2d forward, NCHW fp32, weight (Cin,Cout,kh,kw) --
 * models/decoder.py:28-35.  Gather form: out[oy] += x[iy]*w[ky] where
 * oy = iy*stride - pad + ky.                                                */
VQO_API void vqo_conv_transpose2d(const float *x, const float *w, const float *bias,
                                  int64_t B, int Cin, int H, int W, int Cout,
                                  int kh, int kw, int stride, int pad, int flags,
                                  float *y)
{
    const int Ho = (H - 1) * stride - 2 * pad + kh;
    const int Wo = (W - 1) * stride - 2 * pad + kw;
    for (int64_t b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < kh; ++ky) {
                            const int ty = oy + pad - ky;
                            if (ty < 0 || ty % stride) continue;
                            const int iy = ty / stride;
                            if (iy >= H) continue;
                            for (int kx = 0; kx < kw; ++kx) {
                                const int tx = ox + pad - kx;
                                if (tx < 0 || tx % stride) continue;
                                const int ix = tx / stride;
                                if (ix >= W) continue;
                                float v = x[((b * Cin + ci) * H + iy) * W + ix];
                                if ((flags & 1) && v < 0.0f) v = 0.0f;
                                acc += (double)v *
                                       (double)w[((ci * Cout + co) * kh + ky) * kw + kx];
                            }
                        }
                    float r = (float)acc;
                    if ((flags & 2) && r < 0.0f) r = 0.0f;
                    y[((b * Cout + co) * Ho + oy) * Wo + ox] = r;
                }
}

/* ResidualStack.forward -- models/residual.py:47-51 with the two reference
 * quirks: ONE ResidualLayer aliased n times (:44-45) and
 * nn.ReLU(True) in-place, so the skip carries relu(x) (:19,:28):
 *     t <- x ; repeat n: t <- relu(t) + W2 (*) relu(W1 (*) relu(t)) ; relu(t)
 * w1 (res_h, C, 3, 3) no bias, w2 (C, res_h, 1, 1) no bias.  x is NOT mutated
 * here (the in-place side effect is unobservable through Encoder/Decoder).  */
VQO_API void vqo_residual_stack(const float *x, const float *w1, const float *w2,
                                int64_t B, int C, int H, int W, int res_h, int n_layers,
                                float *y)
{
    const int64_t n = B * C * H * W, nh = B * (int64_t)res_h * H * W;
    float *t = (float *)malloc(sizeof(float) * (size_t)n);
    float *h = (float *)malloc(sizeof(float) * (size_t)nh);
    float *r = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(t, x, sizeof(float) * (size_t)n);
    for (int l = 0; l < n_layers; ++l) {
        for (int64_t i = 0; i < n; ++i) t[i] = t[i] < 0.0f ? 0.0f : t[i]; /* :19 in place */
        vqo_conv2d(t, w1, NULL, B, C, H, W, res_h, 3, 3, 1, 1, 2, h);  /* :20-22 */
        vqo_conv2d(h, w2, NULL, B, res_h, H, W, C, 1, 1, 1, 0, 0, r);  /* :23-24 */
        for (int64_t i = 0; i < n; ++i) {
            volatile float s = t[i] + r[i];                            /* :28 */
            t[i] = s;
        }
    }
    for (int64_t i = 0; i < n; ++i) y[i] = t[i] < 0.0f ? 0.0f : t[i];  /* :50 */
    free(t); free(h); free(r);
}

/* Weights of one VQVAE, as flat fp32 pointers in state_dict layout
 * (SURVEY.md 8b): conv weights (Cout,Cin,kh,kw), conv-transpose weights
 * (Cin,Cout,kh,kw).                                                         */
typedef struct {
    int h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, in_ch;
    float beta;
    const float *enc0_w, *enc0_b, *enc2_w, *enc2_b, *enc4_w, *enc4_b;
    const float *enc_res_w1, *enc_res_w2;
    const float *pre_w, *pre_b;
    const float *codebook;
    const float *dec0_w, *dec0_b;
    const float *dec_res_w1, *dec_res_w2;
    const float *dec2_w, *dec2_b, *dec4_w, *dec4_b;
} vqo_weights;

/* Encoder.forward + pre_quantization_conv -- models/encoder.py:28-43,
 * models/vqvae.py:31-33.  x (B,in_ch,H,W) -> z_e (B,D,H/4,W/4).             */
VQO_API void vqo_encode(const vqo_weights *m, const float *x, int64_t B, int H, int W,
                        float *z_e)
{
    const int h = m->h_dim, h2 = m->h_dim / 2, D = m->embedding_dim;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    float *a0 = (float *)malloc(sizeof(float) * (size_t)(B * h2 * H2 * W2));
    float *a1 = (float *)malloc(sizeof(float) * (size_t)(B * h * H4 * W4));
    float *a2 = (float *)malloc(sizeof(float) * (size_t)(B * h * H4 * W4));
    vqo_conv2d(x, m->enc0_w, m->enc0_b, B, m->in_ch, H, W, h2, 4, 4, 2, 1, 2, a0);  /* :29-31 */
    vqo_conv2d(a0, m->enc2_w, m->enc2_b, B, h2, H2, W2, h, 4, 4, 2, 1, 2, a1);      /* :32-34 */
    vqo_conv2d(a1, m->enc4_w, m->enc4_b, B, h, H4, W4, h, 3, 3, 1, 1, 0, a2);       /* :35-36 */
    vqo_residual_stack(a2, m->enc_res_w1, m->enc_res_w2, B, h, H4, W4,
                       m->res_h_dim, m->n_res_layers, a1);                          /* :37-38 */
    vqo_conv2d(a1, m->pre_w, m->pre_b, B, h, H4, W4, D, 1, 1, 1, 0, 0, z_e);        /* vqvae.py:33 */
    free(a0); free(a1); free(a2);
}

/* Decoder.forward -- models/decoder.py:27-39.  z_q (B,D,h,w) -> (B,3,4h,4w). */
VQO_API void vqo_decode(const vqo_weights *m, const float *z_q, int64_t B, int h4, int w4,
                        float *x_hat)
{
    const int h = m->h_dim, h2 = m->h_dim / 2, D = m->embedding_dim;
    float *a0 = (float *)malloc(sizeof(float) * (size_t)(B * h * h4 * w4));
    float *a1 = (float *)malloc(sizeof(float) * (size_t)(B * h * h4 * w4));
    float *a2 = (float *)malloc(sizeof(float) * (size_t)(B * h2 * h4 * 2 * w4 * 2));
    vqo_conv_transpose2d(z_q, m->dec0_w, m->dec0_b, B, D, h4, w4, h, 3, 3, 1, 1, 0, a0);      /* :28-29 */
    vqo_residual_stack(a0, m->dec_res_w1, m->dec_res_w2, B, h, h4, w4,
                       m->res_h_dim, m->n_res_layers, a1);                                     /* :30 */
    vqo_conv_transpose2d(a1, m->dec2_w, m->dec2_b, B, h, h4, w4, h2, 4, 4, 2, 1, 2, a2);      /* :31-33 */
    vqo_conv_transpose2d(a2, m->dec4_w, m->dec4_b, B, h2, h4 * 2, w4 * 2, m->in_ch,
                         4, 4, 2, 1, 0, x_hat);                                                /* :34-35 */
    free(a0); free(a1); free(a2);
}

/* VQVAE.forward -- models/vqvae.py:29-44: returns (embedding_loss, x_hat,
 * perplexity); z_e / z_q / idx are exposed for the parity tests.            */
VQO_API int vqo_forward(const vqo_weights *m, const float *x, int64_t B, int H, int W,
                        float *x_hat, float *loss, float *perplexity,
                        float *z_e_out, float *z_q_out, int64_t *idx_out)
{
    const int D = m->embedding_dim, K = m->n_embeddings;
    const int h4 = H / 4, w4 = W / 4;
    const int64_t nz = B * D * h4 * w4, N = B * h4 * w4;
    float *z_e = z_e_out ? z_e_out : (float *)malloc(sizeof(float) * (size_t)nz);
    float *z_q = z_q_out ? z_q_out : (float *)malloc(sizeof(float) * (size_t)nz);
    int64_t *idx = idx_out ? idx_out : (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int32_t *hist = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
    vqo_encode(m, x, B, H, W, z_e);                                        /* :31-33 */
    vqo_vq_forward(z_e, m->codebook, B, D, h4, w4, K, m->beta, z_q, idx, hist,
                   loss, perplexity, NULL);                                /* :34-35 */
    vqo_decode(m, z_q, B, h4, w4, x_hat);                                  /* :36 */
    if (!z_e_out) free(z_e);
    if (!z_q_out) free(z_q);
    if (!idx_out) free(idx);
    free(hist);
    return 0;
}
