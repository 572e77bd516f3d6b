// Fused VectorQuantizer forward for gfx950 -- exact-fp32 variant.
//
// Replaces models/quantizer.py:45-76 of the reference with ONE pass over z_e:
//   distance  d[n,k] = fl(fl(||z_n||^2 + ||e_k||^2) - 2*m[n,k]),  m = z_n . e_k
//   argmin (first index, NaN counts as minimum), gather e_k, z + (e_k - z),
//   sum((e_k - z)^2), per-code histogram.
//
// Numerics (pinned by oracle/ and tests/):
//   * m[n,k] is the c-ordered fp32 fmaf chain torch.matmul produces on CPU.  On
//     gfx950 v_mfma_f32_32x32x2_f32 is bit-for-bit that chain (k = 0 then 1 per
//     instruction, instructions chained through the accumulator), so the whole
//     N x K x D contraction runs on the matrix cores with no rounding difference.
//   * ||.||^2 follows ATen's cascade_sum order (8-lane vectors x 4-way ILP).
//
// Mapping: codes on the MFMA M side, latent rows on the N side.  A wave owns
// RT tiles of 32 rows; lane l holds row (l & 31) and the channels c = 2s + (l>>5)
// (exactly the B-operand layout), keeps them in registers for the whole sweep,
// and receives 16 codes x 1 row per 32x32 tile in its accumulator, so the
// running (min, argmin) is per-lane VALU work and only one cross-lane fold
// (l <-> l+32) is needed per row.  The codebook lives in LDS as the A-operand
// image [c/8][c&1][code][(c%8)/2], read conflict-free with ds_read_b128.
#include "common.h"
#include "vq_device.h"

namespace vqvae {

// ---------------------------------------------------------------------------
// Prepare: one thread per (padded) code.  Writes ||e_k||^2 in ATen order, the
// LDS image of the codebook, and raises cb_bad if any norm is not < 1e38.
template <int D>
__global__ __launch_bounds__(64) void vq_prepare_kernel(const float *__restrict__ cb, int K, int KC,
                                                        int K_pad, float *__restrict__ ee,
                                                        float *__restrict__ img,
                                                        int *__restrict__ flags, int K32,
                                                        unsigned short *__restrict__ img16,
                                                        float *__restrict__ neh) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K_pad && k >= K32) return;
    if (k >= K_pad) {                      // only the bf16 image is padded further than the fp32 one
        for (int c = 0; c < D; ++c) img16[(((c % (D / 2)) / 8 * 2 + c / (D / 2)) * (size_t)K32 + k) * 8 + (c & 7)] = 0;
        neh[k] = -__builtin_inff();
        return;
    }
    const int ch = k / KC, kl = k - ch * KC;
    float *dst = img + (size_t)ch * KC * D;
    float e[D];
    if (k < K) {
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(cb + (size_t)k * D + c);
            e[c] = v.x; e[c + 1] = v.y; e[c + 2] = v.z; e[c + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) e[c] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const int j = c >> 3, i = (c & 7) >> 1, h = c & 1;
        dst[((size_t)(j * 2 + h) * KC + kl) * 4 + i] = e[c];
    }
    float n2 = __builtin_inff();     // padded codes can never win
    if (k < K) {
        float sq[D];
#pragma unroll
        for (int c = 0; c < D; ++c) sq[c] = e[c] * e[c];
        n2 = aten_sqsum_full<D>(sq);
        if (!(n2 < 1.0e38f)) atomicOr(flags, 1);
        // statistics for the screening bounds: max ||e||^2 and max |e_kc| (non-negative floats order like ints)
        atomicMax(flags + 1, __float_as_int(n2));
        float am = 0.0f;
#pragma unroll
        for (int c = 0; c < D; ++c) am = fmaxf(am, __builtin_fabsf(e[c]));
        atomicMax(flags + 2, __float_as_int(am));
    }
    ee[k] = n2;
    if (k < K32) {
        // bf16 (round-to-nearest-even) A-operand image [c'/8][c/(D/2)][code][8], c' = c mod D/2
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const unsigned u = __float_as_uint(e[c]);
            const unsigned r = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
            const unsigned short bf = (e[c] != e[c]) ? 0x7FC0 : (unsigned short)r;
            img16[(((c % (D / 2)) / 8 * 2 + c / (D / 2)) * (size_t)K32 + k) * 8 + (c & 7)] = bf;
        }
        neh[k] = k < K ? -0.5f * n2 : -__builtin_inff();
    }
}

// ---------------------------------------------------------------------------
// One software-pipeline step of the codebook sweep: issue the D/2 MFMAs of code tile M into accM
// while the running (min, argmin) absorbs the finished accumulators accA of the previous tile.
// Program order interleaves the two so the VALU work issues in the shadow of this wave's own MFMAs
// (the two waves of a SIMD run in lock-step, so they cannot cover for each other).
template <int D, int RT, bool DO_MFMA, bool DO_ARG>
__device__ __forceinline__ void vq_tile_step(const float *__restrict__ ap, size_t jstride,
                                             const float (&zr)[RT][D / 2], f32x16 (&accM)[RT],
                                             const float *__restrict__ ee_t, int code0,
                                             const f32x16 (&accA)[RT], const float (&zz)[RT],
                                             float (&bd)[RT], int (&bk)[RT]) {
    constexpr int NJ = D / 8;
    if (DO_MFMA) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accM[t][r] = 0.0f;
    }
    f32x4 e4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (DO_MFMA) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + (size_t)j * jstride);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    accM[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], zr[t][4 * j + i], accM[t], 0, 0, 0);
        }
        if (DO_ARG) {
            // accumulator registers [r0, r1) of the previous tile are folded in during this j
            const int r0 = (16 * j) / NJ, r1 = (16 * (j + 1)) / NJ;
#pragma unroll
            for (int r = r0; r < r1; ++r) {
                if ((r & 3) == 0) e4 = *reinterpret_cast<const f32x4 *>(ee_t + 8 * (r >> 2));
                const int code = code0 + 8 * (r >> 2) + (r & 3);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const float tt = zz[t] + e4[r & 3];
                    const float d = __builtin_fmaf(-2.0f, accA[t][r], tt);
                    const bool lt = d < bd[t];
                    bd[t] = lt ? d : bd[t];
                    bk[t] = lt ? code : bk[t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
template <int D, int RT, bool ROWMAJOR>
__global__ __launch_bounds__(512, 2) void vq_exact_kernel(
    const float *__restrict__ z, const float *__restrict__ cb, const float *__restrict__ img,
    const float *__restrict__ ee_g, const int *__restrict__ flags, long long N, int HW, int K,
    int KC, int nchunks, long long nblocks, float *__restrict__ zq, long long *__restrict__ idx,
    int *__restrict__ hist, double *__restrict__ partials) {
    constexpr int S = D / 2;                 // MFMA k-steps (2 channels each)
    constexpr int ROWS_WG = 8 * 32 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Es = smem;                         // [D/8][2][KC][4]
    float *ee_s = Es + (size_t)KC * D;        // [KC]
    int *hist_s = reinterpret_cast<int *>(ee_s + KC);   // [K]
    double *red = reinterpret_cast<double *>(hist_s + ((K + 1) & ~1));

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cb_bad = flags[0];
    for (int k = tid; k < K; k += 512) hist_s[k] = 0;

    auto stage = [&](int ch) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(img + (size_t)ch * KC * D);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Es);
        const int n4 = KC * D / 4;
        for (int i = tid; i < n4; i += 512) dst[i] = src[i];
        for (int i = tid; i < KC; i += 512) ee_s[i] = ee_g[ch * KC + i];
    };
    if (nchunks == 1) stage(0);
    __syncthreads();

    double dacc = 0.0;

    for (long long rb = blockIdx.x; rb < nblocks; rb += gridDim.x) {
        // ---- load this wave's rows into the MFMA B-operand layout --------------
        float zr[RT][S];
        float zz[RT];
        long long row[RT];
        size_t zbase[RT];
        size_t img0[RT];
        unsigned voff[RT];
        bool valid[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            row[t] = rb * ROWS_WG + wave * (32 * RT) + t * 32 + l31;
            valid[t] = row[t] < N;
            const long long rc = valid[t] ? row[t] : N - 1;
            if (ROWMAJOR) {
                zbase[t] = (size_t)rc * D;
#pragma unroll
                for (int q = 0; q < D / 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(z + zbase[t] + 4 * q);
                    zr[t][2 * q] = h ? v.y : v.x;
                    zr[t][2 * q + 1] = h ? v.w : v.z;
                }
            } else {
                // NCHW: element (row, c) sits at ((b*D + c)*HW + hw).  One buffer
                // descriptor per tile (base = first image the tile touches, wave
                // uniform), one 32-bit VGPR offset per lane, and the channel step
                // 2s*HW as a scalar offset: no per-load 64-bit address registers.
                const long long r0 = rb * ROWS_WG + wave_u * (32 * RT) + t * 32;
                const long long b0 = (r0 < N ? r0 : N - 1) / HW;
                const long long b = rc / HW;
                const int hw = (int)(rc - b * HW);
                zbase[t] = (size_t)b * D * HW + hw;
                img0[t] = (size_t)b0 * D * HW;
                voff[t] = (unsigned)((((b - b0) * D + h) * HW + hw) * 4);
                const auto rs = make_rsrc(z + img0[t]);
#pragma unroll
                for (int s = 0; s < S; ++s)
                    zr[t][s] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[t], (unsigned)(2 * s) * HW * 4u, 0));
            }
        }
        // ---- ||z||^2 in ATen order: this lane owns elements t = 2u + h of every
        //      8-vector; the partner lane (l ^ 32) owns the other parity ----------
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            constexpr int NV = D / 8, NI = NV / 4;
            float A[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float P[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = zr[t][4 * (4 * i + q) + u];
                        P[q] = P[q] + x * x;
                    }
#pragma unroll
                for (int v = NI * 4; v < NV; ++v) {
                    const float x = zr[t][4 * v + u];
                    P[0] = P[0] + x * x;
                }
                A[u] = ((P[0] + P[1]) + P[2]) + P[3];
            }
            float fin = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float other = __shfl_xor(A[u], 32);
                const float ev = h ? other : A[u];   // element t = 2u
                const float od = h ? A[u] : other;   // element t = 2u + 1
                fin = fin + ev;
                fin = fin + od;
            }
            zz[t] = fin;
        }

        float bd[RT];
        int bk[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) { bd[t] = __builtin_inff(); bk[t] = 0; }

        // ---- sweep the codebook -------------------------------------------------
        for (int ch = 0; ch < nchunks; ++ch) {
            if (nchunks > 1) {
                __syncthreads();
                stage(ch);
                __syncthreads();
            }
            const int ncode = min(KC, K - ch * KC);
            const int ntile = (ncode + 31) >> 5;
            const size_t jstride = (size_t)2 * KC * 4;
            const float *ap0 = Es + ((size_t)h * KC + l31) * 4;      // + ct*128 per code tile
            const float *ee0 = ee_s + 4 * h;                          // + ct*32 per code tile
            const int cbase = ch * KC + 4 * h;
            f32x16 accA[RT], accB[RT];
            vq_tile_step<D, RT, true, false>(ap0, jstride, zr, accA, ee0, cbase, accB, zz, bd, bk);
            int ct = 1;
            for (; ct + 1 < ntile; ct += 2) {
                vq_tile_step<D, RT, true, true>(ap0 + ct * 128, jstride, zr, accB, ee0 + (ct - 1) * 32,
                                                cbase + (ct - 1) * 32, accA, zz, bd, bk);
                vq_tile_step<D, RT, true, true>(ap0 + (ct + 1) * 128, jstride, zr, accA, ee0 + ct * 32,
                                                cbase + ct * 32, accB, zz, bd, bk);
            }
            if (ct < ntile) {
                vq_tile_step<D, RT, true, true>(ap0 + ct * 128, jstride, zr, accB, ee0 + (ct - 1) * 32,
                                                cbase + (ct - 1) * 32, accA, zz, bd, bk);
                vq_tile_step<D, RT, false, true>(ap0, jstride, zr, accA, ee0 + ct * 32, cbase + ct * 32, accB, zz,
                                                 bd, bk);
            } else {
                vq_tile_step<D, RT, false, true>(ap0, jstride, zr, accB, ee0 + (ct - 1) * 32,
                                                 cbase + (ct - 1) * 32, accA, zz, bd, bk);
            }
        }

        // ---- fold the two half-waves, fix up non-finite rows, epilogue ----------
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float pd = __shfl_xor(bd[t], 32);
            const int pk = __shfl_xor(bk[t], 32);
            const bool take = (pd < bd[t]) || (pd == bd[t] && pk < bk[t]);
            int k = take ? pk : bk[t];

            const bool bad = valid[t] && (cb_bad || !(zz[t] < 1.0e38f));
            if (__any(bad)) {
                int ks = 0;
                if (bad && h == 0)
                    ks = vq_slow_argmin<D, ROWMAJOR>(z, zbase[t], ROWMAJOR ? 1 : (size_t)HW, cb, ee_g,
                                                     K, zz[t]);
                ks = __shfl(ks, l31);
                if (bad) k = ks;
            }

            float sq = 0.0f;
            const float *e = cb + (size_t)k * D;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float ev = e[2 * s + h];
                const float diff = ev - zr[t][s];
                sq = sq + diff * diff;
                zr[t][s] = zr[t][s] + diff;          // z + (z_q - z), models/quantizer.py:67
            }
            if (valid[t]) {
                dacc += (double)sq;
                if (zq) {
                    if (ROWMAJOR) {
#pragma unroll
                        for (int q = 0; q < D / 4; ++q) {
                            const float o0 = __shfl_xor(zr[t][2 * q], 32);
                            const float o1 = __shfl_xor(zr[t][2 * q + 1], 32);
                            if ((q & 1) == h) {
                                f32x4 v;
                                v.x = h ? o0 : zr[t][2 * q];
                                v.y = h ? zr[t][2 * q] : o0;
                                v.z = h ? o1 : zr[t][2 * q + 1];
                                v.w = h ? zr[t][2 * q + 1] : o1;
                                *reinterpret_cast<f32x4 *>(zq + zbase[t] + 4 * q) = v;
                            }
                        }
                    }
                }
                if (h == 0) {
                    idx[row[t]] = k;
                    atomicAdd(&hist_s[k], 1);
                }
            }
            if (!ROWMAJOR && zq && valid[t]) {
                const auto rs = make_rsrc(zq + img0[t]);
#pragma unroll
                for (int s = 0; s < S; ++s)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zr[t][s]), rs,
                                                          voff[t], (unsigned)(2 * s) * HW * 4u, 0);
            }
        }
    }

    // ---- per-workgroup results: squared-error partial + histogram flush ---------
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
    __syncthreads();
    if (lane == 0) red[wave] = dacc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < 8; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
    for (int k = tid; k < K; k += 512) {
        const int c = hist_s[k];
        if (c) atomicAdd(&hist[k], c);
    }
}

// ---------------------------------------------------------------------------
// loss = m + beta*m with m = sum((z_q-z)^2)/(N*D) (models/quantizer.py:63-64);
// perplexity = exp(-sum p log(p + 1e-10)), p = hist/N (:70-71).
__global__ __launch_bounds__(256) void vq_finalize_kernel(const double *__restrict__ partials, int nparts,
                                                          const int *__restrict__ hist, int K,
                                                          long long N, int D, float beta,
                                                          float *__restrict__ loss,
                                                          float *__restrict__ perplexity) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    const float fn = (float)N;
    for (int k = tid; k < K; k += 256) {
        const float p = (float)hist[k] / fn;
        const float lg = (float)log((double)(p + 1e-10f));
        s += (double)(p * lg);
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double ent = red[0];
    __syncthreads();
    // squared-error partials: fixed-order tree (deterministic run to run)
    double q = 0.0;
    for (int i = tid; i < nparts; i += 256) q += partials[i];
    red[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        *perplexity = (float)exp(-(double)(float)ent);
        const float m = (float)(red[0] / ((double)N * (double)D));
        const float bm = beta * m;
        *loss = m + bm;
    }
}

__global__ __launch_bounds__(256) void vq_onehot_kernel(const long long *__restrict__ idx, long long N,
                                                        int K, float *__restrict__ onehot) {
    const long long total = N * (long long)K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const long long n = i / K;
        const int k = (int)(i - n * K);
        onehot[i] = idx[n] == k ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(256) void vq_decode_indices_kernel(const long long *__restrict__ idx,
                                                                const float *__restrict__ cb,
                                                                long long total, int D, int HW, int K,
                                                                float *__restrict__ zq) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % D);
        const long long b = bc / D;
        // an index outside [0, K) never reads the codebook: the element becomes NaN (the reference's embedding
        // lookup raises there; the Python front end does too, before the launch)
        const long long k = idx[b * HW + hw];
        zq[i] = (k >= 0 && k < K) ? cb[(size_t)k * D + c] : __builtin_nanf("");
    }
}

// ---------------------------------------------------------------------------
// codebook images and bound statistics into the workspace (every kernel family's)
template <int D>
static int launch_vq_prepare(const float *cb, int K, char *ws, hipStream_t st) {
    const VqPlan p = vq_plan(K, D);
    int *wflags = reinterpret_cast<int *>(ws + p.off_flags);
    hipError_t e;
    if ((e = hipMemsetAsync(wflags, 0, 256 + kVqTicketBytes, st)) != hipSuccess) return (int)e;
    const int kmax = p.K_pad > p.K32 ? p.K_pad : p.K32;
    hipLaunchKernelGGL(vq_prepare_kernel<D>, dim3((kmax + 63) / 64), dim3(64), 0, st, cb, K, p.KC, p.K_pad,
                       reinterpret_cast<float *>(ws + p.off_ee), reinterpret_cast<float *>(ws + p.off_img), wflags, p.K32,
                       reinterpret_cast<unsigned short *>(ws + p.off_img16), reinterpret_cast<float *>(ws + p.off_neh));
    if (vq_track_ok(K, D) || vq_chunk_ok(K, D)) launch_vq_prepare16(cb, K, D, ws, st);
    return (int)hipGetLastError();
}

template <int D>
static int launch_vq(const float *z, const float *cb, long long N, int HW, int K, float beta,
                     int flags, float *zq, long long *idx, int *hist, float *loss, float *ppl,
                     char *ws, hipStream_t st, bool hist_zeroed, int *zq_amax, bool *zq_amax_done) {
    const VqPlan p = vq_plan(K, D);
    int *wflags = reinterpret_cast<int *>(ws + p.off_flags);
    float *ee = reinterpret_cast<float *>(ws + p.off_ee);
    float *img = reinterpret_cast<float *>(ws + p.off_img);
    double *partials = reinterpret_cast<double *>(ws + p.off_partials);

    hipError_t e;
    // (hist_zeroed: the kernel in front of this one in the stream has cleared it -- vqvae_forward_f32's fused path)
    if (!hist_zeroed && (e = hipMemsetAsync(hist, 0, sizeof(int) * (size_t)K, st)) != hipSuccess) return (int)e;
    if (!(flags & VQVAE_VQ_CODEBOOK_PREPARED)) {
        const int rc = launch_vq_prepare<D>(cb, K, ws, st);
        if (rc != 0) return rc;
    }
    const bool rowmajor = flags & VQVAE_VQ_ROWMAJOR;
    if constexpr (D == 64) {
        // NCHW maps whose pixel count is a multiple of 32 (a unit = 64 or 32 positions of one image): the stream-tracker kernel reads
        // and writes the reference's own layout (round 4); other NCHW maps stay on the two-sweep kernel below
        const bool track_nchw = !rowmajor && vq_track_nchw_ok(K, D, HW) && !(flags & (VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER));
        if (track_nchw || (rowmajor && vq_track_ok(K, D) && !(flags & (VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER)))) {
            int fgrid = 0;
            const int rc = launch_vq_track_d64(z, cb, N, K, zq, idx, hist, ws, st, &fgrid, HW, track_nchw,
                                               vq_form_of_flags(flags));
            if (rc != 0) return rc;
            hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partials, fgrid, hist, K, N, D,
                               beta, loss, ppl);
            return (int)hipGetLastError();
        }
    }
    if constexpr (D == 64 || D == 128) {
        // larger codebooks / D = 128: the same fp16 screen with the codebook image streamed through LDS (vq_chunk.hip)
        if (rowmajor && vq_chunk_ok(K, D) && !vq_track_ok(K, D) && !(flags & (VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER))) {
            int fgrid = 0;
            prof_begin(VQVAE_PROF_VQ_MAIN, st);
            const int rc = launch_vq_chunked(z, cb, N, K, D, zq, idx, hist, ws, st, &fgrid, zq_amax, HW);
            if (zq_amax_done) *zq_amax_done = zq_amax != nullptr && zq != nullptr;
            prof_end(VQVAE_PROF_VQ_MAIN, st);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partials, fgrid, hist, K, N, D,
                               beta, loss, ppl);
            return (int)hipGetLastError();
        }
    }
    if constexpr (D == 64) {
        if (p.filter_ok && !(flags & VQVAE_VQ_EXACT_SWEEP)) {
            int fgrid = 0;
            prof_begin(VQVAE_PROF_VQ_MAIN, st);
            const int rc = launch_vq_filter_d64(z, cb, N, HW, K, rowmajor, zq, idx, hist, ws, st, &fgrid);
            prof_end(VQVAE_PROF_VQ_MAIN, st);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partials, fgrid, hist, K, N, D,
                               beta, loss, ppl);
            return (int)hipGetLastError();
        }
    }
    const int cus = num_cus();
    // two row tiles per wave when that still leaves every CU at least two row blocks
    int rt = (D <= 64 && (N + 511) / 512 >= 2LL * cus) ? 2 : 1;
    const long long rows_wg = 256LL * rt;
    const long long nblocks = (N + rows_wg - 1) / rows_wg;
    long long grid = nblocks < cus ? nblocks : cus;
    if (grid > kVqMaxGrid) grid = kVqMaxGrid;

    prof_begin(VQVAE_PROF_VQ_MAIN, st);
#define VQ_LAUNCH(RT_, RM_)                                                                        \
    do {                                                                                           \
        auto kfn = vq_exact_kernel<D, RT_, RM_>;                                                   \
        /* per device, so set on every launch (a process may drive several GPUs) */                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);            \
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), p.lds_bytes, st, z, cb, img, ee,  \
                           wflags, N, HW, K, p.KC, p.nchunks, nblocks, zq, idx, hist, partials);   \
    } while (0)

    if (rt == 2) {
        if constexpr (D <= 64) {
            if (rowmajor) VQ_LAUNCH(2, true); else VQ_LAUNCH(2, false);
        }
    } else {
        if (rowmajor) VQ_LAUNCH(1, true); else VQ_LAUNCH(1, false);
    }
#undef VQ_LAUNCH
    prof_end(VQVAE_PROF_VQ_MAIN, st);
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partials, (int)grid, hist, K, N,
                       D, beta, loss, ppl);
    return (int)hipGetLastError();
}

}  // namespace vqvae

using namespace vqvae;

extern "C" {

static bool vq_mfma_dim(int D) { return D == 32 || D == 64 || D == 128 || D == 256; }

const char *vqvae_vq_kernel_name(int K, int D, int flags) {
    if (K < 1 || K > 16384 || (flags & VQVAE_VQ_REMOVED_FLAGS)) return "unsupported";
    // any other width: exact fp32 -- on the matrix cores since round 6 (zero-padded to a multiple of eight channels), round 5's
    // vector-unit kernel behind VQVAE_VQ_BF16_FILTER for A/B runs
    if (!vq_mfma_dim(D)) return vq_generic_ok(K, D) ? ((flags & VQVAE_VQ_BF16_FILTER) ? "vq_generic_kernel" : "vq_anyd_kernel") : "unsupported";
    if (D == 64 && !(flags & VQVAE_VQ_EXACT_SWEEP)) {
        if ((flags & VQVAE_VQ_ROWMAJOR) && vq_track_ok(K, D) && !(flags & VQVAE_VQ_BF16_FILTER)) return "vq_track_kernel_d64";
        if ((flags & VQVAE_VQ_ROWMAJOR) && vq_chunk_ok(K, D) && !(flags & VQVAE_VQ_BF16_FILTER)) return "vq_stream_sweep_kernel";
        // NCHW (the module boundary): the stream-tracker kernel on maps whose pixel count is a multiple of 32 (8x8, 56x56, 64x64
        // ...: what this function answers for); other NCHW maps run vq_filter_kernel_d64
        if (!(flags & (VQVAE_VQ_ROWMAJOR | VQVAE_VQ_BF16_FILTER)) && vq_track_nchw_ok(K, D, 64))
            return "vq_track_kernel_d64";
        if (vq_plan(K, D).filter_ok) return "vq_filter_kernel_d64";
    }
    if (D == 128 && (flags & VQVAE_VQ_ROWMAJOR) && !(flags & (VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER)) && vq_chunk_ok(K, D))
        return "vq_stream_sweep_kernel";
    return "vq_exact_kernel";
}

int vqvae_vq_launch_form(int64_t n_rows, int K, int D, int HW, int flags, int *waves, int *unit_rows, int *pool_pct) {
    if (n_rows < 1 || K < 1 || HW < 1) return VQVAE_ERR_SHAPE;
    const char *n = vqvae_vq_kernel_name(K, D, flags);
    VqTrackForm f;
    const bool nchw = !(flags & VQVAE_VQ_ROWMAJOR);
    if (D != 64 || n[3] != 't' || !vq_track_form(n_rows, K, HW, nchw, vq_form_of_flags(flags),
                                                  num_cus(), f))
        return VQVAE_ERR_UNSUPPORTED;
    if (waves) *waves = f.waves;
    if (unit_rows) *unit_rows = f.unit_rows;
    if (pool_pct) *pool_pct = f.pool_pct;
    return VQVAE_OK;
}

int vqvae_vq_screen_sweeps(int K, int D, int flags) {
    const char *n = vqvae_vq_kernel_name(K, D, flags);
    return (n[3] == 's' || n[3] == 't') ? 1 : (n[3] == 'f' ? 2 : 0);
}

size_t vqvae_vq_workspace_bytes(int64_t n_rows, int K, int D) {
    if (K < 1 || K > 16384) return 0;
    (void)n_rows;
    if (!vq_mfma_dim(D)) return vq_generic_ok(K, D) ? vq_generic_workspace_bytes(K, D) : 0;
    return vq_plan(K, D).total;
}

int vqvae_vq_forward_f32(const float *z_e, const float *codebook, int64_t B, int D, int H, int W,
                         int K, float beta, int flags, float *z_q, int64_t *idx, int32_t *hist,
                         float *loss, float *perplexity, void *workspace, size_t workspace_bytes,
                         vqvae_stream_t stream) {
    return vqvae::vq_forward_impl(z_e, codebook, B, D, H, W, K, beta, flags, z_q, idx, hist, loss, perplexity, workspace,
                                  workspace_bytes, stream, false);
}
}  // extern "C"

// ---- the quantizer inside the encoder's last kernel (conv_fused.hip): what vqvae_forward_f32 does around that launch ---------
bool vqvae::vq_fuse_ok(int K, int D, int64_t B, int flags) {
    const VqPlan p = vq_plan(K > 0 ? K : 1, 64);
    (void)B;
    return D == 64 && K >= 1 && K <= 1024 && p.K32 % 128 == 0 && vq_track_ok(K, D) &&
           !(flags & (VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER | VQVAE_VQ_UNFUSED));
}

int vqvae::vq_prepare_impl(const float *codebook, int K, int D, int flags, void *workspace, size_t workspace_bytes, hipStream_t st) {
    if (!codebook || !workspace) return VQVAE_ERR_NULL;
    if (D != 64 || K < 1 || K > 16384) return VQVAE_ERR_UNSUPPORTED;
    if (workspace_bytes < vqvae_vq_workspace_bytes(0, K, D)) return VQVAE_ERR_WORKSPACE;
    if (flags & VQVAE_VQ_CODEBOOK_PREPARED) return VQVAE_OK;
    return launch_vq_prepare<64>(codebook, K, static_cast<char *>(workspace), st);
}

vqvae::VqFuse vqvae::vq_fuse_args(const float *codebook, int K, void *workspace, float *z_q, int64_t *idx, int32_t *hist) {
    const VqPlan p = vq_plan(K, 64);
    char *ws = static_cast<char *>(workspace);
    VqFuse f;
    f.imgf = reinterpret_cast<const uint4 *>(ws + p.off_imgf);
    f.seeds = reinterpret_cast<const float *>(ws + p.off_seeds);
    f.ee = reinterpret_cast<const float *>(ws + p.off_ee);
    f.flags = reinterpret_cast<const int *>(ws + p.off_flags);
    f.cb = codebook;
    f.K = K;
    f.K32 = p.K32;
    f.zq = z_q;
    f.idx = reinterpret_cast<long long *>(idx);
    f.hist = hist;
    f.partials = reinterpret_cast<double *>(ws + p.off_partials);
    return f;
}

int vqvae::vq_finalize_impl(const double *partials, int grid, int32_t *hist, int K, int64_t n_rows, int D, float beta, float *loss,
                            float *perplexity, hipStream_t st) {
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(256), 0, st, partials, grid, hist, K, (long long)n_rows, D, beta, loss, perplexity);
    return (int)hipGetLastError();
}

int vqvae::vq_forward_impl(const float *z_e, const float *codebook, int64_t B, int D, int H, int W,
                           int K, float beta, int flags, float *z_q, int64_t *idx, int32_t *hist,
                           float *loss, float *perplexity, void *workspace, size_t workspace_bytes,
                           vqvae_stream_t stream, bool hist_zeroed, int *zq_amax, bool *zq_amax_done) {
    if (zq_amax_done) *zq_amax_done = false;
    if (!z_e || !codebook || !idx || !hist || !loss || !perplexity) return VQVAE_ERR_NULL;
    if (B < 1 || D < 1 || H < 1 || W < 1 || K < 1) return VQVAE_ERR_SHAPE;
    const bool generic = !(D == 32 || D == 64 || D == 128 || D == 256);
    if (K > 16384 || (generic && !vq_generic_ok(K, D))) return VQVAE_ERR_UNSUPPORTED;
    if ((int64_t)H * W > (int64_t)1 << 30) return VQVAE_ERR_OVERFLOW;
    const int64_t N = B * (int64_t)H * W;
    if (N / ((int64_t)H * W) != B || N > ((int64_t)1 << 40)) return VQVAE_ERR_OVERFLOW;
    if (flags & VQVAE_VQ_REMOVED_FLAGS) return VQVAE_ERR_UNSUPPORTED;     // round 2's tracker kernel and its A/B flags are gone (round 4)
    const size_t need = vqvae_vq_workspace_bytes(N, K, D);
    if (!workspace || workspace_bytes < need) return VQVAE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    const int HW = H * W;
    long long *idx_ll = reinterpret_cast<long long *>(idx);
    // any other embedding width (main.py:21 leaves it free): exact fp32 on the vector units, the same bits (vq_generic.hip); the
    // kernel-selection flags have nothing to select there
    if (generic)
        return launch_vq_generic(z_e, codebook, N, HW, K, D, beta, (flags & VQVAE_VQ_ROWMAJOR) != 0, z_q, idx_ll, hist, loss, perplexity, ws, st,
                                 hist_zeroed, (flags & VQVAE_VQ_BF16_FILTER) != 0, (flags & VQVAE_VQ_CODEBOOK_PREPARED) != 0);
    switch (D) {
        case 32:  return launch_vq<32>(z_e, codebook, N, HW, K, beta, flags, z_q, idx_ll, hist, loss, perplexity, ws, st, hist_zeroed, zq_amax, zq_amax_done);
        case 64:  return launch_vq<64>(z_e, codebook, N, HW, K, beta, flags, z_q, idx_ll, hist, loss, perplexity, ws, st, hist_zeroed, zq_amax, zq_amax_done);
        case 128: return launch_vq<128>(z_e, codebook, N, HW, K, beta, flags, z_q, idx_ll, hist, loss, perplexity, ws, st, hist_zeroed, zq_amax, zq_amax_done);
        case 256: return launch_vq<256>(z_e, codebook, N, HW, K, beta, flags, z_q, idx_ll, hist, loss, perplexity, ws, st, hist_zeroed, zq_amax, zq_amax_done);
    }
    return VQVAE_ERR_UNSUPPORTED;
}

extern "C" {

int vqvae_vq_onehot_f32(const int64_t *idx, int64_t N, int K, float *onehot, vqvae_stream_t stream) {
    if (!idx || !onehot) return VQVAE_ERR_NULL;
    if (N < 1 || K < 1) return VQVAE_ERR_SHAPE;
    if (N > INT64_MAX / K) return VQVAE_ERR_OVERFLOW;
    const long long total = N * (long long)K;
    long long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(vq_onehot_kernel, dim3((unsigned)grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const long long *>(idx), N, K,
                       onehot);
    return (int)hipGetLastError();
}

int vqvae_vq_decode_indices_f32(const int64_t *idx, const float *codebook, int64_t B, int D, int H,
                                int W, int K, float *z_q, vqvae_stream_t stream) {
    if (!idx || !codebook || !z_q) return VQVAE_ERR_NULL;
    if (B < 1 || D < 1 || H < 1 || W < 1 || K < 1) return VQVAE_ERR_SHAPE;
    const long long total = B * (long long)D * H * W;
    long long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(vq_decode_indices_kernel, dim3((unsigned)grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const long long *>(idx),
                       codebook, total, D, H * W, K, z_q);
    return (int)hipGetLastError();
}

}  // extern "C"
