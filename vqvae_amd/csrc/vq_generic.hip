// VectorQuantizer.forward for ANY embedding width (models/quantizer.py:29-76; main.py:21 --embedding_dim is free): the exact fp32
// path for shapes none of the MFMA kernels take (D not in {32, 64, 128, 256}).  Round 5: the reference "just runs" with e.g.
// --embedding_dim 48; until now this library answered VQVAE_ERR_UNSUPPORTED.
//
// Bit-exact by construction, with the rounding order the reference delegates to ATen (restated on the CPU by the tests' oracle and verified against
// live torch for D = 1 .. 4096, tests/test_oracle.py):
//   z @ E^T            one c-ordered fmaf chain per (row, code), accumulator from 0            (quantizer.py:51; what the reference's
//                      sgemm computes up to D = 383 -- beyond, MKL blocks the reduction: hence D <= 256 here, the pinned range)
//   sum(x ** 2, dim=1) ATen's cascade_sum inner-dimension order for the row length D           (quantizer.py:49-50)
//   d                  fl(fl(zz + ee) - fl(2 m))                                                (quantizer.py:49-51)
//   argmin             first minimal index, NaN counts as minimal                              (quantizer.py:54)
//   z_q                fl(z + fl(e - z))                                                        (quantizer.py:67)
// No matrix cores: one thread owns a code and runs its chains for the tile's eight rows (rows from LDS by broadcast reads, the code's
// row in 16-byte pieces where D % 4 == 0); N K D fused multiply-adds at a fraction of the vector peak -- a correct path for unusual
// widths, not a fast one (262 144 rows, K = 512, D = 48: ~1 ms).
#include "common.h"
#include "vq_track.h"

namespace vqvae {
namespace {

constexpr int kGenRows = 8;                                  // rows per tile

// torch.sum(x, dim=-1) of one row of n floats in ATen's order (aten/native/cpu/SumKernel: vectorized_inner_sum -> row_sum -> multi_row_sum):
// 8-float vectors, 4-way ILP, four cascade levels of 16 steps; elem(c) = the row's c-th element
// rows shorter than one vector take ATen's scalar_inner_sum: four partial sums over elements 4 i + k, leftovers into slot 0, slots 1..3
// into slot 0 (n = 5: ((x0 + x4) + x1 + x2) + x3 -- not the sequential sum)
template <class Elem>
__device__ float aten_row_sum_short(Elem &&elem, int n) {
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int si = n / 4;
    for (int i = 0; i < si; ++i)
        for (int k = 0; k < 4; ++k) p[k] = p[k] + elem(i * 4 + k);
    for (int i = si * 4; i < n; ++i) p[0] = p[0] + elem(i);
    for (int k = 1; k < 4; ++k) p[0] = p[0] + p[k];
    return p[0];
}
template <class Elem>
__device__ float aten_row_sum_generic(Elem &&elem, int n) {
    if (n < 8) return aten_row_sum_short(elem, n);
    const int vec_size = n / 8, size_ilp = vec_size / 4;
    float acc[4][4][8];
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 8; ++t) acc[j][k][t] = 0.0f;
    int level_power = 4;
    {
        int cl2 = 0;
        while ((1 << cl2) < size_ilp) ++cl2;
        if (cl2 / 4 > level_power) level_power = cl2 / 4;
    }
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < 4; ++k)
                for (int t = 0; t < 8; ++t) acc[0][k][t] = acc[0][k][t] + elem((i * 4 + k) * 8 + t);
        for (int j = 1; j < 4; ++j) {
            for (int k = 0; k < 4; ++k)
                for (int t = 0; t < 8; ++t) {
                    acc[j][k][t] = acc[j][k][t] + acc[j - 1][k][t];
                    acc[j - 1][k][t] = 0.0f;
                }
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 8; ++t) acc[0][k][t] = acc[0][k][t] + elem((i * 4 + k) * 8 + t);
    for (int j = 1; j < 4; ++j)
        for (int k = 0; k < 4; ++k)
            for (int t = 0; t < 8; ++t) acc[0][k][t] = acc[0][k][t] + acc[j][k][t];
    for (int v = size_ilp * 4; v < vec_size; ++v)
        for (int t = 0; t < 8; ++t) acc[0][0][t] = acc[0][0][t] + elem(v * 8 + t);
    for (int k = 1; k < 4; ++k)
        for (int t = 0; t < 8; ++t) acc[0][0][t] = acc[0][0][t] + acc[0][k][t];
    float fin = 0.0f;
    for (int c = vec_size * 8; c < n; ++c) fin = fin + elem(c);
    for (int t = 0; t < 8; ++t) fin = fin + acc[0][0][t];
    return fin;
}

// out[r] = sum(x[r] ** 2) in ATen's order: one thread per row.  Row-major rows (stride_c = 1, row r at r * D) or an NCHW map
// (row r = (b, p): element c at (b * D + c) * HW + p)
__global__ __launch_bounds__(256) void vq_generic_sqnorm_kernel(const float *__restrict__ x, long long rows, int D, int HW, bool nchw,
                                                                 float *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *base = nchw ? x + (r / HW) * (long long)D * HW + (r % HW) : x + r * D;
    const long long sc = nchw ? HW : 1;
    out[r] = aten_row_sum_generic([&](int c) { const float v = base[(long long)c * sc]; return v * v; }, D);
}

// The same sum for the kGenRows rows of a tile in LDS (zs: rows of Dp floats), by the whole workgroup: ATen's order is 32 independent
// lanes (k = ILP slot, t = vector element) that only meet at the end, so thread (r, k, t) runs lane (k, t) of row r -- cascade levels
// included -- and one thread per row joins them in ATen's order: leftover vectors into slot 0, slots 1..3 into slot 0, the scalar tail,
// then the eight elements.  part_s: kGenRows x 32 floats; zz_s: kGenRows floats (valid after the closing barrier).
__device__ __forceinline__ void tile_row_sqnorms(const float *zs, int Dp, int D, float *part_s, float *zz_s) {
    const int tid = threadIdx.x, r = tid >> 5, k = (tid >> 3) & 3, t = tid & 7;
    const int vec_size = D / 8, size_ilp = vec_size / 4;
    if (D < 8) {                                             // (shorter than one vector: ATen's scalar path, one thread per row)
        if (tid < kGenRows) zz_s[tid] = aten_row_sum_short([&](int c) { const float v = zs[tid * Dp + c]; return v * v; }, D);
        __syncthreads();
        return;
    }
    auto sq = [&](int c) { const float v = zs[r * Dp + c]; return v * v; };
    float lv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int level_power = 4;
    {
        int cl2 = 0;
        while ((1 << cl2) < size_ilp) ++cl2;
        if (cl2 / 4 > level_power) level_power = cl2 / 4;
    }
    const int level_step = 1 << level_power, level_mask = level_step - 1;
    int i = 0;
    while (i + level_step <= size_ilp) {
        for (int j = 0; j < level_step; ++j, ++i) lv[0] = lv[0] + sq((i * 4 + k) * 8 + t);
        for (int j = 1; j < 4; ++j) {
            lv[j] = lv[j] + lv[j - 1];
            lv[j - 1] = 0.0f;
            if ((i & (level_mask << (j * level_power))) != 0) break;
        }
    }
    for (; i < size_ilp; ++i) lv[0] = lv[0] + sq((i * 4 + k) * 8 + t);
    for (int j = 1; j < 4; ++j) lv[0] = lv[0] + lv[j];
    part_s[tid] = lv[0];                                     // [r][k][t]
    __syncthreads();
    if (tid < kGenRows) {
        const int rr = tid;
        auto sqr = [&](int c) { const float v = zs[rr * Dp + c]; return v * v; };
        float p0[8];
        for (int e = 0; e < 8; ++e) p0[e] = part_s[rr * 32 + e];
        for (int v = size_ilp * 4; v < vec_size; ++v)
            for (int e = 0; e < 8; ++e) p0[e] = p0[e] + sqr(v * 8 + e);
        for (int kk = 1; kk < 4; ++kk)
            for (int e = 0; e < 8; ++e) p0[e] = p0[e] + part_s[rr * 32 + kk * 8 + e];
        float fin = 0.0f;
        for (int c = vec_size * 8; c < D; ++c) fin = fin + sqr(c);
        for (int e = 0; e < 8; ++e) fin = fin + p0[e];
        zz_s[rr] = fin;
    }
    __syncthreads();
}

// the tile version alone: out[r] = sum(x[r] ** 2) for row-major rows (vqvae_debug_row_sqnorm_f32 mode 1: tests compare both forms
// with torch.sum bit for bit)
__global__ __launch_bounds__(256) void vq_generic_sqnorm_tile_kernel(const float *__restrict__ x, long long rows, int D, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float zs[];
    __shared__ float part_s[kGenRows * 32], zz_s[kGenRows];
    const int Dp = (D + 3) & ~3;
    const long long row0 = (long long)blockIdx.x * kGenRows;
    for (int i = threadIdx.x; i < kGenRows * Dp; i += 256) {
        const int r = i / Dp, c = i % Dp;
        zs[i] = (row0 + r < rows && c < D) ? x[(row0 + r) * D + c] : 0.0f;
    }
    __syncthreads();
    tile_row_sqnorms(zs, Dp, D, part_s, zz_s);
    if (threadIdx.x < kGenRows && row0 + threadIdx.x < rows) out[row0 + threadIdx.x] = zz_s[threadIdx.x];
}

// torch.argmin's order on (distance, index) as one key: NaN below everything, then the distance, then the index (trk::dist_key with
// the NaN rule; -0 cannot occur: x - y of x >= +0 is never -0)
__device__ __forceinline__ unsigned long long argmin_key(float d, int k) {
    const unsigned u = __float_as_uint(d);
    const unsigned s = (d != d) ? 0u : ((u >> 31) ? ~u : (u | 0x80000000u));
    return ((unsigned long long)s << 32) | (unsigned)k;
}

template <bool NCHW, bool VEC4>
__global__ __launch_bounds__(256) void vq_generic_kernel(const float *__restrict__ z, const float *__restrict__ cb,
                                                         const float *__restrict__ ee, long long N, int HW, int K, int D,
                                                         float *__restrict__ zq, long long *__restrict__ idx, int *__restrict__ hist,
                                                         double *__restrict__ partials) {
    constexpr int R = kGenRows;
    extern __shared__ __attribute__((aligned(16))) float zs[];            // R rows of Dp floats, then the workgroup's histogram (K ints)
    __shared__ unsigned long long best_s[R];
    __shared__ double red_s[4];
    __shared__ float part_s[R * 32], zz_s[R];
    static_assert(R * 32 == 256, "one thread per (row, ILP slot, vector element) of the tile's ||z||^2");
    const int tid = threadIdx.x, Dp = (D + 3) & ~3;
    int *hist_s = reinterpret_cast<int *>(zs + R * Dp);      // (counts per workgroup first: the rows of a batch crowd on few codes, and
    for (int k = tid; k < K; k += 256) hist_s[k] = 0;        //  262 144 global atomics on a few hundred addresses were 80 % of this kernel)
    double sacc = 0.0;
    const long long ntiles = (N + R - 1) / R;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * R;
        const int nr = (int)(N - row0 < R ? N - row0 : R);
        __syncthreads();                                                  // the previous tile's rows are no longer read
        for (int i = tid; i < R * Dp; i += 256) {
            // row-major: consecutive threads take consecutive channels of a row; NCHW: consecutive rows (pixels) of a channel
            const int r = NCHW ? i % R : i / Dp, c = NCHW ? i / R : i % Dp;
            float v = 0.0f;
            if (r < nr && c < D) {
                const long long row = row0 + r;
                v = NCHW ? z[((row / HW) * D + c) * (long long)HW + row % HW] : z[row * D + c];
            }
            zs[r * Dp + c] = v;
        }
        if (tid < R) best_s[tid] = ~0ull;
        __syncthreads();
        tile_row_sqnorms(zs, Dp, D, part_s, zz_s);           // ||z||^2 of the tile's rows, ATen's order (quantizer.py:49)
        float zzr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) zzr[r] = zz_s[r];
        const float *zg = z + (NCHW ? 0 : row0 * D);          // (row-major: the tile's rows, wave-uniform)
        unsigned long long best[R];
#pragma unroll
        for (int r = 0; r < R; ++r) best[r] = ~0ull;
        // a thread owns codes tid, tid + 256, ... and takes them TWO at a time: the rows' broadcast reads serve both chains, and two
        // code rows are in flight (the second code of a pair is clamped to the first where K runs out; its keys are not used)
        for (int k = tid; k < K; k += 512) {
            const int k2 = k + 256, kb = k2 < K ? k2 : k;
            const float *ea = cb + (size_t)k * D, *eb = cb + (size_t)kb * D;
            float ma[R], mb[R];
#pragma unroll
            for (int r = 0; r < R; ++r) ma[r] = mb[r] = 0.0f;
            int c = 0;
            if constexpr (VEC4) {
                for (; c + 4 <= D; c += 4) {
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(ea + c), vb = *reinterpret_cast<const f32x4 *>(eb + c);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        // the rows' values are the same for every lane: row-major rows come through the SCALAR cache straight from z
                        // (s_load_dwordx4, the value a scalar operand of the fmaf: no LDS traffic in the chain loop -- eight broadcast
                        // reads per 64 fmaf made the LDS the bound); NCHW maps read the transposed tile in LDS
                        const f32x4 zv = NCHW ? *reinterpret_cast<const f32x4 *>(zs + r * Dp + c)
                                              : *reinterpret_cast<const f32x4 *>(zg + (size_t)(r < nr ? r : 0) * D + c);
                        ma[r] = __builtin_fmaf(zv.w, va.w, __builtin_fmaf(zv.z, va.z, __builtin_fmaf(zv.y, va.y, __builtin_fmaf(zv.x, va.x, ma[r]))));
                        mb[r] = __builtin_fmaf(zv.w, vb.w, __builtin_fmaf(zv.z, vb.z, __builtin_fmaf(zv.y, vb.y, __builtin_fmaf(zv.x, vb.x, mb[r]))));
                    }
                }
            }
            for (; c < D; ++c) {
                const float va = ea[c], vb = eb[c];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float zv = zs[r * Dp + c];
                    ma[r] = __builtin_fmaf(zv, va, ma[r]);
                    mb[r] = __builtin_fmaf(zv, vb, mb[r]);
                }
            }
            const float eea = ee[k], eeb = ee[kb];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned long long keya = argmin_key((zzr[r] + eea) - 2.0f * ma[r], k);
                best[r] = keya < best[r] ? keya : best[r];
                const unsigned long long keyb = argmin_key((zzr[r] + eeb) - 2.0f * mb[r], kb);     // (kb == k: the same key again)
                best[r] = keyb < best[r] ? keyb : best[r];
            }
        }
        // the wave's minimum first (256 lanes on eight LDS addresses serialise: this was most of the kernel), then one atomic per wave
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned long long b = best[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)b, o), hi = __shfl_xor((unsigned)(b >> 32), o);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                b = other < b ? other : b;
            }
            if ((tid & 63) == 0 && b != ~0ull) atomicMin(&best_s[r], b);
        }
        __syncthreads();
        // z_q = z + (e - z), the squared error, indices, histogram
        for (int i = tid; i < R * Dp; i += 256) {
            const int r = NCHW ? i % R : i / Dp, c = NCHW ? i / R : i % Dp;
            if (r < nr && c < D) {
                const int kb = (int)(unsigned)best_s[r];
                const float zv = zs[r * Dp + c], diff = cb[(size_t)kb * D + c] - zv;
                sacc += (double)(diff * diff);
                if (zq) {
                    const long long row = row0 + r;
                    const float q = zv + diff;
                    if (NCHW) zq[((row / HW) * D + c) * (long long)HW + row % HW] = q; else zq[row * D + c] = q;
                }
            }
        }
        if (tid < nr) {
            const int kb = (int)(unsigned)best_s[tid];
            idx[row0 + tid] = kb;
            atomicAdd(&hist_s[kb], 1);
        }
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        const int cnt = hist_s[k];
        if (cnt) atomicAdd(&hist[k], cnt);
    }
    // the workgroup's squared-error partial in a fixed order (run-to-run bitwise loss)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if ((tid & 63) == 0) red_s[tid >> 6] = sacc;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = ((red_s[0] + red_s[1]) + red_s[2]) + red_s[3];
}


// =====================================================================================================================
// Round 6: the same contract ON THE MATRIX CORES, for any width.  v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fp32 fmaf chain
// (what vq_exact_kernel rests on for D in {32, 64, 128, 256}) and fmaf(0, 0, acc) = acc, so a row and a code padded
// with ZERO channels to a multiple of eight give the reference's m[n, k] exactly: the N x K x D contraction of an odd width runs
// at the fp32 MFMA rate (~80 TFLOP/s measured) instead of the 16 TFLOP/s of the per-thread chains above.  Everything the padding
// would change stays in the true width: ||z||^2 and ||e||^2 in ATen's order for row length D, distances, keys, z + (e - z).
//   workgroup = 64 rows (two MFMA column tiles) x all codes; the rows sit in LDS as two planes (even / odd channels: lane (row, h)
//   reads four k-steps of its B operand with one 16-byte read), the codebook as an A-operand image in the workspace
//   ([code tile][channel octet][parity][code][4 floats]: a wave's 32 x 2 lanes read 2 x 512 contiguous bytes per 16-byte load),
//   wave w takes the code tiles w, w + 4, ...; per lane a running (distance, index) key -- codes arrive in ascending order -- folded
//   across halves and waves by a 64-bit LDS minimum.  Rows whose ||z||^2 is not < 1e38 (NaN / Inf / overflow: the MFMA's special-value
//   behaviour is not part of the pinned contract) and unusable codebooks take fmaf chains on the vector units, 256 threads per row.
constexpr int kAnyRows = 64;

__global__ __launch_bounds__(64) void vq_anyd_prepare_kernel(const float *__restrict__ cb, int K, int D, int Q, int ntile,
                                                             float *__restrict__ ee, float *__restrict__ eeimg,
                                                             float *__restrict__ aimg, int *__restrict__ flags) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ntile * 32) return;
    const int tile = k >> 5, m = k & 31;
    float n2 = __builtin_inff();                             // padding codes can never win
    if (k < K) {
        const float *row = cb + (size_t)k * D;
        n2 = aten_row_sum_generic([&](int c) { const float v = row[c]; return v * v; }, D);
        ee[k] = n2;
        if (!(n2 < 1.0e38f)) atomicOr(flags, 1);
    }
    // accumulator register r of lane half h holds code (r & 3) + 8 (r >> 2) + 4 h of the tile
    eeimg[(tile * 2 + ((m >> 2) & 1)) * 16 + (m & 3) + 4 * (m >> 3)] = n2;
    for (int c = 0; c < Q * 8; ++c) {
        const float v = (k < K && c < D) ? cb[(size_t)k * D + c] : 0.0f;
        aimg[((((size_t)tile * Q + (c >> 3)) * 2 + (c & 1)) * 32 + m) * 4 + ((c & 7) >> 1)] = v;
    }
}

template <bool NCHW>
__global__ __launch_bounds__(256) void vq_anyd_kernel(const float *__restrict__ z, const float *__restrict__ cb, const float *__restrict__ ee,
                                                      const float *__restrict__ eeimg, const float *__restrict__ aimg,
                                                      const int *__restrict__ flags, long long N, int HW, int K, int D, int Q, int S,
                                                      int ntile, float *__restrict__ zq, long long *__restrict__ idx, int *__restrict__ hist,
                                                      double *__restrict__ partials, int lds_hist, int vec4) {
    constexpr int R = kAnyRows;
    extern __shared__ __attribute__((aligned(16))) float zs[];            // [parity 2][row R][k-step S], zero-padded; then the histogram
    __shared__ unsigned long long best_s[R];
    __shared__ double red_s[4];
    __shared__ float part_s[R * 32], zz_s[R];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    int *hist_s = reinterpret_cast<int *>(zs + 2 * R * S);
    if (lds_hist)
        for (int k = tid; k < K; k += 256) hist_s[k] = 0;
    const int cb_bad = flags[0];
    auto zat = [&](int r, int c) -> float & { return zs[(c & 1) * (R * S) + r * S + (c >> 1)]; };
    double sacc = 0.0;
    const long long ntiles = (N + R - 1) / R;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long row0 = tile * R;
        const int nr = (int)(N - row0 < R ? N - row0 : R);
        __syncthreads();                                                  // the previous tile's rows are no longer read
        const int Dp = Q * 8;
        if (!NCHW && vec4) {
            // row-major rows of a width that is a multiple of four: 16-byte loads, a few per thread and all in flight at once (the scalar
            // form's twelve dependent load -> LDS-store rounds per tile were what bounded the kernel at D = 48); the padding k-steps once
            const int D4 = D >> 2;
#pragma unroll 4
            for (int i = tid; i < R * D4; i += 256) {
                const int r = i / D4, c4 = i - r * D4;
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (r < nr) v = *reinterpret_cast<const f32x4 *>(z + (row0 + r) * D + 4 * c4);
                float *p0 = zs + r * S + 2 * c4, *p1 = p0 + R * S;
                p0[0] = v.x; p0[1] = v.z;
                p1[0] = v.y; p1[1] = v.w;
            }
            for (int i = tid; i < R * (Dp - D); i += 256) {
                const int r = i / (Dp - D), c = D + i % (Dp - D);
                zat(r, c) = 0.0f;
            }
        } else {
#pragma unroll 4
            for (int i = tid; i < R * Dp; i += 256) {
                const int r = NCHW ? i % R : i / Dp, c = NCHW ? i / R : i % Dp;
                float v = 0.0f;
                if (r < nr && c < D) {
                    const long long row = row0 + r;
                    v = NCHW ? z[((row / HW) * D + c) * (long long)HW + row % HW] : z[row * D + c];
                }
                zat(r, c) = v;
            }
        }
        if (tid < R) best_s[tid] = ~0ull;
        __syncthreads();
        // ---- ||z||^2 of the 64 rows in ATen's order for row length D (quantizer.py:49).  ATen's order is 32 independent lanes per row
        // (ILP slot k, vector element t) that only meet at the end: thread (row, k) runs the eight lanes (k, 0..7) of its row -- cascade
        // levels included --, one thread per row joins the 32 partials in ATen's order (leftover vectors into slot 0, slots 1..3 into
        // slot 0, the scalar tail, then the eight elements): two barriers per tile
        {
            const int r = tid >> 2, k = tid & 3;
            const int vec_size = D / 8, size_ilp = vec_size / 4;
            if (D < 8) {
                if (tid < R) zz_s[tid] = aten_row_sum_short([&](int c) { const float v = zat(tid, c); return v * v; }, D);
            } else {
                int level_power = 4;
                {
                    int cl2 = 0;
                    while ((1 << cl2) < size_ilp) ++cl2;
                    if (cl2 / 4 > level_power) level_power = cl2 / 4;
                }
                const int level_step = 1 << level_power, level_mask = level_step - 1;
                auto sq = [&](int c) { const float v = zat(r, c); return v * v; };
                float lv[4][8];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 8; ++t) lv[j][t] = 0.0f;
                int i = 0;
                while (i + level_step <= size_ilp) {
                    for (int j = 0; j < level_step; ++j, ++i)
#pragma unroll
                        for (int t = 0; t < 8; ++t) lv[0][t] = lv[0][t] + sq((i * 4 + k) * 8 + t);
                    bool stop = false;
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        if (!stop) {
#pragma unroll
                            for (int t = 0; t < 8; ++t) {
                                lv[j][t] = lv[j][t] + lv[j - 1][t];
                                lv[j - 1][t] = 0.0f;
                            }
                            if ((i & (level_mask << (j * level_power))) != 0) stop = true;
                        }
                    }
                }
                for (; i < size_ilp; ++i)
#pragma unroll
                    for (int t = 0; t < 8; ++t) lv[0][t] = lv[0][t] + sq((i * 4 + k) * 8 + t);
#pragma unroll
                for (int j = 1; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < 8; ++t) lv[0][t] = lv[0][t] + lv[j][t];
#pragma unroll
                for (int t = 0; t < 8; ++t) part_s[(r * 4 + k) * 8 + t] = lv[0][t];       // [row][k][t]
                __syncthreads();
                if (tid < R) {
                    const int rr = tid;
                    auto sqr = [&](int c) { const float v = zat(rr, c); return v * v; };
                    float p0[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) p0[e] = part_s[rr * 32 + e];
                    for (int v = size_ilp * 4; v < vec_size; ++v)
#pragma unroll
                        for (int e = 0; e < 8; ++e) p0[e] = p0[e] + sqr(v * 8 + e);
#pragma unroll
                    for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                        for (int e = 0; e < 8; ++e) p0[e] = p0[e] + part_s[rr * 32 + kk * 8 + e];
                    float fin = 0.0f;
                    for (int c = vec_size * 8; c < D; ++c) fin = fin + sqr(c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) fin = fin + p0[e];
                    zz_s[rr] = fin;
                }
            }
            __syncthreads();
        }
        // ---- the sweep: wave w the code tiles w, w + 4, ...; both 32-row column tiles against every A operand ------------
        {
            const float zz0 = zz_s[l31], zz1 = zz_s[32 + l31];
            // running (distance, index) per lane and column tile: codes arrive in ASCENDING order, so a strict < keeps torch.argmin's
            // first index; no distance of a row this sweep decides is NaN (||z||^2, ||e||^2 < 1e38: the other rows are redone below)
            float bd0 = __builtin_inff(), bd1 = __builtin_inff();
            int bk0 = 0, bk1 = 0;
            const float *bp = zs + h * (R * S) + l31 * S;
            for (int t = wave; t < ntile; t += 4) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
                const float *ap = aimg + (((size_t)t * Q) * 2 + h) * 128 + l31 * 4;
                f32x4 a = *reinterpret_cast<const f32x4 *>(ap);
                for (int q = 0; q < Q; ++q) {
                    const f32x4 an = *reinterpret_cast<const f32x4 *>(ap + (size_t)(q + 1 < Q ? q + 1 : q) * 256);     // a step ahead
                    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bp + 4 * q);
                    const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bp + 32 * S + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0[i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1[i], acc1, 0, 0, 0);
                    }
                    a = an;
                }
                const float *eet = eeimg + (t * 2 + h) * 16;
                const int code0 = t * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int code = code0 + (r & 3) + 8 * (r >> 2);
                    const float e = eet[r];                               // (+inf for the padding codes of the last tile: never below)
                    const float d0 = __builtin_fmaf(-2.0f, acc0[r], zz0 + e), d1 = __builtin_fmaf(-2.0f, acc1[r], zz1 + e);
                    const bool l0 = d0 < bd0, l1 = d1 < bd1;
                    bd0 = l0 ? d0 : bd0; bk0 = l0 ? code : bk0;
                    bd1 = l1 ? d1 : bd1; bk1 = l1 ? code : bk1;
                }
            }
            // the two halves of a column, then the four waves: (distance, index) keys through the 64-bit LDS minimum
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                unsigned long long b = argmin_key(rt ? bd1 : bd0, rt ? bk1 : bk0);
                const unsigned lo = __shfl_xor((unsigned)b, 32), hi = __shfl_xor((unsigned)(b >> 32), 32);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                b = other < b ? other : b;
                if (h == 0) atomicMin(&best_s[32 * rt + l31], b);
            }
        }
        __syncthreads();
        // ---- rows outside the MFMA's pinned range: fmaf chains on the vector units, the whole workgroup per row --------------
        {
            bool any_bad = cb_bad != 0;
            if (!any_bad)
                for (int r = 0; r < nr; ++r) any_bad = any_bad || !(zz_s[r] < 1.0e38f);      // (LDS broadcast reads; uniform)
            if (any_bad) {
                if (tid < nr && (cb_bad || !(zz_s[tid] < 1.0e38f))) best_s[tid] = ~0ull;
                __syncthreads();
                for (int r = 0; r < nr; ++r) {
                    const float zzr = zz_s[r];
                    if (!cb_bad && zzr < 1.0e38f) continue;
                    unsigned long long b = ~0ull;
                    for (int k = tid; k < K; k += 256) {
                        float m = 0.0f;
                        for (int c = 0; c < D; ++c) m = __builtin_fmaf(zat(r, c), cb[(size_t)k * D + c], m);
                        const unsigned long long key = argmin_key((zzr + ee[k]) - 2.0f * m, k);
                        b = key < b ? key : b;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned lo = __shfl_xor((unsigned)b, o), hi = __shfl_xor((unsigned)(b >> 32), o);
                        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                        b = other < b ? other : b;
                    }
                    if (lane == 0 && b != ~0ull) atomicMin(&best_s[r], b);
                }
                __syncthreads();
            }
        }
        // ---- z_q = z + (e - z), the squared error, indices, histogram ------------------------------------------------------
        if (!NCHW && vec4) {
            const int D4 = D >> 2;
#pragma unroll 4
            for (int i = tid; i < R * D4; i += 256) {
                const int r = i / D4, c4 = i - r * D4;
                if (r < nr) {
                    const int kb = (int)(unsigned)best_s[r];
                    const f32x4 e = *reinterpret_cast<const f32x4 *>(cb + (size_t)kb * D + 4 * c4);
                    const float *p0 = zs + r * S + 2 * c4, *p1 = p0 + R * S;
                    const float z0 = p0[0], z1 = p1[0], z2 = p0[1], z3 = p1[1];
                    const float d0 = e.x - z0, d1 = e.y - z1, d2 = e.z - z2, d3 = e.w - z3;
                    sacc += (double)(d0 * d0);
                    sacc += (double)(d1 * d1);
                    sacc += (double)(d2 * d2);
                    sacc += (double)(d3 * d3);
                    if (zq) *reinterpret_cast<f32x4 *>(zq + (row0 + r) * D + 4 * c4) = f32x4{z0 + d0, z1 + d1, z2 + d2, z3 + d3};
                }
            }
        } else {
#pragma unroll 4
            for (int i = tid; i < R * D; i += 256) {
                const int r = NCHW ? i % R : i / D, c = NCHW ? i / R : i % D;
                if (r < nr) {
                    const int kb = (int)(unsigned)best_s[r];
                    const float zv = zat(r, c), diff = cb[(size_t)kb * D + c] - zv;
                    sacc += (double)(diff * diff);
                    if (zq) {
                        const long long row = row0 + r;
                        const float q = zv + diff;
                        if (NCHW) zq[((row / HW) * D + c) * (long long)HW + row % HW] = q; else zq[row * D + c] = q;
                    }
                }
            }
        }
        if (tid < nr) {
            const int kb = (int)(unsigned)best_s[tid];
            idx[row0 + tid] = kb;
            atomicAdd(lds_hist ? &hist_s[kb] : &hist[kb], 1);
        }
    }
    __syncthreads();
    if (lds_hist)
        for (int k = tid; k < K; k += 256) {
            const int cnt = hist_s[k];
            if (cnt) atomicAdd(&hist[k], cnt);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if ((tid & 63) == 0) red_s[tid >> 6] = sacc;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = ((red_s[0] + red_s[1]) + red_s[2]) + red_s[3];
}

}  // namespace

bool vq_generic_ok(int K, int D) { return K >= 1 && K <= 16384 && D >= 1 && D <= kVqGenericMaxD; }

// workspace: ee (K floats), one loss partial per workgroup; round 6: + the matrix-core kernel's flags, ee image and A-operand image
constexpr int kGenMaxGrid = 4096;
struct AnydPlan {
    int Q, S, ntile;
    size_t off_partials, off_flags, off_eeimg, off_aimg, total;
};
static AnydPlan anyd_plan(int K, int D) {
    AnydPlan p;
    p.Q = (D + 7) / 8;
    const int Dh = p.Q * 4;                                  // k-steps per row (two channels each)
    p.S = (Dh % 8 == 0) ? Dh + 4 : Dh;                       // plane row stride = 4 mod 8 floats: sixteen lanes' 16-byte reads hit sixteen bank groups
    p.ntile = (K + 31) / 32;
    p.off_partials = align_up((size_t)K * 4, 256);
    p.off_flags = p.off_partials + (size_t)kGenMaxGrid * 8;
    p.off_eeimg = p.off_flags + 256;
    p.off_aimg = p.off_eeimg + align_up((size_t)p.ntile * 32 * 4, 256);
    p.total = p.off_aimg + (size_t)p.ntile * 32 * p.Q * 8 * 4;
    return p;
}
size_t vq_generic_workspace_bytes(int K, int D) { return anyd_plan(K, D).total; }

// vector_units: the round-5 kernel (per-thread fmaf chains), kept for A/B runs and as the second implementation the tests compare
// the matrix-core one with (VQVAE_VQ_BF16_FILTER selects it for these widths: the flag has nothing else to select there)
int launch_vq_generic(const float *z, const float *cb, long long N, int HW, int K, int D, float beta, bool rowmajor, float *zq,
                      long long *idx, int *hist, float *loss, float *ppl, char *ws, hipStream_t st, bool hist_zeroed, bool vector_units,
                      bool prepared) {
    if (!vector_units) {
        const AnydPlan p = anyd_plan(K, D);
        float *ee = reinterpret_cast<float *>(ws), *eeimg = reinterpret_cast<float *>(ws + p.off_eeimg), *aimg = reinterpret_cast<float *>(ws + p.off_aimg);
        int *flags = reinterpret_cast<int *>(ws + p.off_flags);
        double *partials = reinterpret_cast<double *>(ws + p.off_partials);
        if (!hist_zeroed && hipMemsetAsync(hist, 0, (size_t)K * sizeof(int), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
        if (!prepared) {
            if (hipMemsetAsync(flags, 0, 256, st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
            hipLaunchKernelGGL(vq_anyd_prepare_kernel, dim3((unsigned)((p.ntile * 32 + 63) / 64)), dim3(64), 0, st, cb, K, D, p.Q, p.ntile, ee, eeimg,
                               aimg, flags);
        }
        const long long ntiles = (N + kAnyRows - 1) / kAnyRows;
        const int lds_hist = K <= 8192 ? 1 : 0;
        const size_t lds = (size_t)2 * kAnyRows * p.S * sizeof(float) + (lds_hist ? (size_t)K * sizeof(int) : 0);
        const long long per_cu = lds > 36 * 1024 ? (lds > 72 * 1024 ? 1 : 2) : 4;
        long long grid = ntiles < per_cu * num_cus() ? ntiles : per_cu * num_cus();
        if (grid > kGenMaxGrid) grid = kGenMaxGrid;
        prof_begin(VQVAE_PROF_VQ_MAIN, st);
#define ANYD_LAUNCH(NCHW_)                                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(vq_anyd_kernel<NCHW_>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 16384); \
    hipLaunchKernelGGL((vq_anyd_kernel<NCHW_>), dim3((unsigned)grid), dim3(256), lds, st, z, cb, ee, eeimg, aimg, flags, N, HW, K, D, p.Q, p.S, \
                       p.ntile, zq, idx, hist, partials, lds_hist, vec4)
        const int vec4 = (D % 4 == 0 && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(cb) | reinterpret_cast<uintptr_t>(zq)) & 15) == 0) ? 1 : 0;
        if (rowmajor) { ANYD_LAUNCH(false); } else { ANYD_LAUNCH(true); }
#undef ANYD_LAUNCH
        prof_end(VQVAE_PROF_VQ_MAIN, st);
        if (hipGetLastError() != hipSuccess) return VQVAE_ERR_UNSUPPORTED;
        return vq_finalize_impl(partials, (int)grid, hist, K, (int64_t)N, D, beta, loss, ppl, st);
    }
    float *ee = reinterpret_cast<float *>(ws);
    double *partials = reinterpret_cast<double *>(ws + align_up((size_t)K * 4, 256));
    if (!hist_zeroed && hipMemsetAsync(hist, 0, (size_t)K * sizeof(int), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
    hipLaunchKernelGGL(vq_generic_sqnorm_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, cb, (long long)K, D, 1, false, ee);
    const long long ntiles = (N + kGenRows - 1) / kGenRows;
    long long grid = ntiles < 8LL * num_cus() ? ntiles : 8LL * num_cus();     // (latency-bound tiles: eight workgroups per CU)
    if (grid > kGenMaxGrid) grid = kGenMaxGrid;
    const size_t lds = (size_t)kGenRows * ((D + 3) & ~3) * sizeof(float) + (size_t)K * sizeof(int);
    const bool vec4 = D % 4 == 0 && (reinterpret_cast<uintptr_t>(cb) & 15) == 0;
    prof_begin(VQVAE_PROF_VQ_MAIN, st);
#define GEN_LAUNCH(NCHW_, VEC4_)                                                                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(vq_generic_kernel<NCHW_, VEC4_>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes - 4096); \
    hipLaunchKernelGGL((vq_generic_kernel<NCHW_, VEC4_>), dim3((unsigned)grid), dim3(256), lds, st, z, cb, ee, N, HW, K, D, zq, idx, hist, \
                       partials)
    if (rowmajor) { if (vec4) { GEN_LAUNCH(false, true); } else { GEN_LAUNCH(false, false); } }
    else { if (vec4) { GEN_LAUNCH(true, true); } else { GEN_LAUNCH(true, false); } }
#undef GEN_LAUNCH
    prof_end(VQVAE_PROF_VQ_MAIN, st);
    return vq_finalize_impl(partials, (int)grid, hist, K, (int64_t)N, D, beta, loss, ppl, st);
}

}  // namespace vqvae

// Test hook: out[r] = torch.sum(x[r] ** 2) for row-major rows of any width D <= 1024, in ATen's order -- mode 0: one thread per row (what
// the codebook's ||e||^2 uses), mode 1: the workgroup-cooperative tile form (what the rows' ||z||^2 uses).  tests/test_vq_generic_gpu.py
// compares both with torch.sum bit for bit, D = 1 .. 1024.
extern "C" VQVAE_API int vqvae_debug_row_sqnorm_f32(const float *x, int64_t rows, int D, int mode, float *out, vqvae_stream_t stream) {
    using namespace vqvae;
    if (!x || !out) return VQVAE_ERR_NULL;
    if (rows < 1 || D < 1) return VQVAE_ERR_SHAPE;
    if (D > kRowSqnormMaxD || (mode != 0 && mode != 1)) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 0)
        hipLaunchKernelGGL(vq_generic_sqnorm_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, (long long)rows, D, 1, false, out);
    else
        hipLaunchKernelGGL(vq_generic_sqnorm_tile_kernel, dim3((unsigned)((rows + kGenRows - 1) / kGenRows)), dim3(256),
                           (size_t)kGenRows * ((D + 3) & ~3) * sizeof(float), st, x, (long long)rows, D, out);
    return (int)hipGetLastError();
}
