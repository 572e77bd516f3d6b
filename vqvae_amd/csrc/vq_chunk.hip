// Fused VectorQuantizer forward for LARGE codebooks and D = 64 / 128 (row-major rows) -- the single-sweep fp16 screen of
// The single-sweep fp16 screen (vq_prepare16.hip has the bound) with the codebook image STREAMED through LDS instead of resident in it.
//
// Same contract and the same bits out as vq_exact.hip (indices and z_q bit-identical to the reference,
// models/quantizer.py:45-74).  vq_sweep.hip needs the whole fp16 image (K x D x 2 B), the ||e||^2 table, a histogram and
// eight wave tiles in 160 KB of LDS: K <= ~600 at D = 64.  BASELINE config 4 (K = 1024, D = 64) and config 5
// (K = 8192, D = 128: a 2 MiB image) do not fit, and the exhaustive fp32 kernel they fell back to is bound by the fp32
// matrix cores (75.7 of config 5's 125 ms step).  Here the work is split into four small kernels per slab of 2^18 rows:
//
//   vq_stream_rows16_kernel   rows -> fp16 B-operand image [tile][D/8][32 rows] x 16 B (so that a wave reads 1 KiB contiguous
//                       per MFMA step) + per row |z^|^2 and the MEASURED |z - z^|^2 (exact differences, fixed order)
//   vq_stream_sweep_kernel    one workgroup per CU, eight waves, each owning two 32-row tiles whose B operands stay in
//                       registers for the whole sweep; the codebook image streams through a double-buffered LDS chunk
//                       (32 KiB: 8 code tiles at D = 64, 4 at D = 128) filled by global_load_lds (no staging registers),
//                       one workgroup barrier per chunk; per lane the top-3 KEYS of vq_sweep.hip (accumulator with its
//                       low 10 bits replaced by [16 - tile-in-epoch : 5][half : 1][r : 4]); every 16 tiles (an "epoch")
//                       the lane's three keys are decoded and merged into a running (value, code) top-3, so the key
//                       field -- and with it the truncation term of the bound -- does not grow with K.
//                       Classification as in vq_sweep.hip: v1 - v2 >= DELTA -> index written; v1 - v3 >= DELTA -> pair
//                       task (row, c1, c2); otherwise / non-finite -> hard task (row)
//   vq_stream_resolve_kernel  pair tasks: one lane per task, both distances EXACTLY (c-ordered fmaf chain, ATen-order ||z||^2,
//                       first-index rule); hard tasks: one wave per row, every code exactly, torch.argmin semantics
//                       (NaN is minimal, first index wins)
//   vq_stream_gather_kernel   z_q = z + (e_k - z), squared error, histogram (once per call, over all rows)
//
// The bound is the one derived in vq_prepare16.hip with g' = (D + 1) 2^-23 and g = D 2^-24 * 1.01.  Extra HBM traffic against
// the fused kernel (fp16 image written and read once, rows read a second time by the gather): 3 D bytes per row -- 0.4 ms
// of config 5's step, against the ~70 ms the fp32 sweep costs.
#include "common.h"
#include "vq_device.h"
#include "vq_track.h"

namespace vqvae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kEpoch = 16;                   // code tiles per key epoch (5-bit field: 1 .. 16)

template <int D> struct ChunkCfg;                        // TC = code tiles per LDS chunk: 32 KiB of image
#ifndef VQS_TC64
#define VQS_TC64 8
#endif
#ifndef VQS_TC128
#define VQS_TC128 4
#endif
template <> struct ChunkCfg<64> { static constexpr int TC = VQS_TC64; };
template <> struct ChunkCfg<128> { static constexpr int TC = VQS_TC128; };

// ---------------------------------------------------------------------------------------------------------------------
// rows -> fp16 B-operand image + row statistics.  One 256-thread block per 32-row tile of the slab.
template <int D>
__global__ __launch_bounds__(256) void vq_stream_rows16_kernel(const float *__restrict__ z, long long nrows,
                                                         u32x4 *__restrict__ img, float2 *__restrict__ stat,
                                                         int *__restrict__ counters, int *__restrict__ batch_done,
                                                         int zero_counters) {
    constexpr int G = D / 8;                              // 8-channel groups per row
    __shared__ float part[32][G][2];
    __shared__ int badrow[32];
    const int tid = threadIdx.x, n = tid & 31, g0 = tid >> 5;
    const long long tile = blockIdx.x;
    const long long row = tile * 32 + n;
    if (zero_counters && blockIdx.x == 0 && tid < 2) counters[tid] = 0;        // first slab of a group (see launch_chunked)
    if (tid == 0) batch_done[blockIdx.x] = 0;             // one possible batch of 32 hard rows per row tile
    if (tid < 32) badrow[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < G / 8; ++pass) {
        const int gg = g0 + 8 * pass;
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, b = a;
        if (row < nrows) {
            const float *src = z + (size_t)row * D + 8 * gg;
            a = *reinterpret_cast<const f32x4 *>(src);
            b = *reinterpret_cast<const f32x4 *>(src + 4);
        }
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned short hbits[8];
        float zh2 = 0.0f, dz2 = 0.0f;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bad = bad || !(__builtin_fabsf(v[j]) <= 65504.0f);          // NaN, Inf, or beyond fp16's range
            const _Float16 hv = (_Float16)v[j];                         // round to nearest even
            const float hf = (float)hv;
            const float d = v[j] - hf;                                  // exact
            zh2 = __builtin_fmaf(hf, hf, zh2);
            dz2 = __builtin_fmaf(d, d, dz2);
            hbits[j] = __builtin_bit_cast(unsigned short, hv);
        }
        u32x4 w;
        w.x = hbits[0] | ((unsigned)hbits[1] << 16);
        w.y = hbits[2] | ((unsigned)hbits[3] << 16);
        w.z = hbits[4] | ((unsigned)hbits[5] << 16);
        w.w = hbits[6] | ((unsigned)hbits[7] << 16);
        img[((size_t)tile * G + gg) * 32 + n] = w;
        part[n][gg][0] = zh2;
        part[n][gg][1] = dz2;
        if (bad) atomicOr(&badrow[n], 1);
    }
    __syncthreads();
    if (tid < 32) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) { s0 += part[tid][gg][0]; s1 += part[tid][gg][1]; }
        if (badrow[tid]) s1 = __builtin_inff();
        stat[tile * 32 + tid] = make_float2(s0, s1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The sweep.  512 threads, one workgroup per CU; dynamic LDS = 2 chunk buffers of TC * (D * 64 + 128) + pad bytes.
template <int D>
__global__ __launch_bounds__(512, 2) void vq_stream_sweep_kernel(
    const u32x4 *__restrict__ rows16, const float2 *__restrict__ stat, const u32x4 *__restrict__ img_g,
    const float *__restrict__ seeds_g, const int *__restrict__ flags, int nrows, long long row0, int K, int ntile,
    long long *__restrict__ idx, uint4 *__restrict__ open_list, unsigned *__restrict__ hard_list,
    unsigned long long *__restrict__ hard_best, int *__restrict__ counters, unsigned rel0) {
    // rel0: this slab's first row relative to its group's (the records of a group of slabs carry group-relative rows and
    // share the counters: one resolve launch per group)
    constexpr int TC = ChunkCfg<D>::TC, NQ = D / 16, G = D / 8;
    constexpr int IMG_UNITS = TC * D * 4;                 // 16-byte units of image per chunk
    constexpr int SEED_PIECES = (TC * 8 + 63) / 64;       // 1 KiB pieces of seeds per chunk (TC * 128 B, rounded up)
    constexpr int BUF_UNITS = IMG_UNITS + 64 * SEED_PIECES;
    static_assert(IMG_UNITS % 512 == 0 && SEED_PIECES <= 8 && kEpoch % TC == 0, "whole 1 KiB pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4 *bufs = reinterpret_cast<u32x4 *>(smem_raw);

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = (ntile + TC - 1) / TC;
    const int nblk = (nrows + 511) / 512;

    const int cb_bad = flags[0];
    const int a_e = flags[5];
    const float A = __builtin_ldexpf(1.0f, a_e);
    const float EEmax = __int_as_float(flags[1]) * 1.0001f;               // max ee_k (unscaled)
    const float Ehat = __builtin_sqrtf(__int_as_float(flags[3])) * 1.0001f;
    const float dE = __builtin_sqrtf(__int_as_float(flags[4])) * 1.0001f;
    const float EmaxS = __builtin_sqrtf(EEmax) * A * 1.0001f;
    const float EEh = 0.5f * EEmax * A, EEa = EEmax * A;
    constexpr float kGp = (float)(D + 1) * 1.1921e-7f * 1.001f;           // (D + 1) 2^-23: fp32 accumulation of the screen
    constexpr float kG = (float)D * 5.9605e-8f * 1.011f;                  // D 2^-24 * 1.01: the reference's fmaf chain
    const float inf = __builtin_inff();

    // chunk j of the codebook stream -> buffer j & 1 (LDS-DMA: every lane's 16 bytes land at piece base + 16 lane)
    auto issue_copy = [&](int chunk, int slot) {
        const int lane = tid & 63;
        u32x4 *dst = bufs + (size_t)slot * BUF_UNITS;
        const u32x4 *src = img_g + (size_t)chunk * IMG_UNITS;
#pragma unroll
        for (int j = 0; j < IMG_UNITS / 512; ++j) {
            const int piece = wave_u * (IMG_UNITS / 512) + j;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + piece * 64 + lane),
                (__attribute__((address_space(3))) void *)(dst + piece * 64), 16, 0, 0);
        }
        if (wave_u < SEED_PIECES) {
            const u32x4 *ssrc = reinterpret_cast<const u32x4 *>(seeds_g) + (size_t)chunk * (TC * 8) + wave_u * 64;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(ssrc + lane),
                (__attribute__((address_space(3))) void *)(dst + IMG_UNITS + wave_u * 64), 16, 0, 0);
        }
    };

    int stream = 0;                                        // chunks consumed so far (buffer parity)
    if (blockIdx.x < nblk) issue_copy(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int lane_v = tid & 63;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v, l31 = lane_v & 31, h = lane_v >> 5;
        const int r0 = blk * 512 + wave_u * 64;           // slab-relative first row of the wave's pair
        const int tl0 = r0 >> 5;

        // B operands of both row tiles: resident for the whole sweep
        f16x8 zb[2][NQ];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                zb[t][q] = __builtin_bit_cast(f16x8, rows16[((size_t)(tl0 + t) * G + 2 * q + h) * 32 + l31]);

        // round 3: the stream tracker of vq_track.h.  Streams S[8] run over the WHOLE codebook; the three largest cell keys
        // (cell = half of a 32-code tile, 6-bit field = cell within the 16-tile epoch) are folded at every epoch end into
        // running (value, global cell) triples G / C, so the key field does not grow with K
        float G1[2], G2[2], G3[2];
        int C1[2], C2[2];
        trk::Lane L[2];
        float pinf = inf, ninf = -inf;                      // opaque: see vq_track.h
        unsigned keymask = trk::kKeyMask;
        asm volatile("" : "+v"(pinf), "+v"(ninf), "+v"(keymask));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            G1[t] = G2[t] = G3[t] = ninf;
            C1[t] = C2[t] = 0;
            trk::init(L[t], ninf);
        }

        const bool more_blocks = blk + (int)gridDim.x < nblk;
        for (int c = 0; c < nchunk; ++c, ++stream) {
            const int slot = stream & 1;
            if (c + 1 < nchunk) issue_copy(c + 1, slot ^ 1);
            else if (more_blocks) issue_copy(0, slot ^ 1);

            const u32x4 *cbuf = bufs + (size_t)slot * BUF_UNITS + h * 32 + l31;
            const float *sbuf = reinterpret_cast<const float *>(bufs + (size_t)slot * BUF_UNITS + IMG_UNITS) + h * 16;
            const int nt_here = (ntile - c * TC) < TC ? (ntile - c * TC) : TC;

            u32x4 a[NQ];
            f32x16 seed;
            auto fetch = [&](int lt) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) a[q] = cbuf[(lt * G + 2 * q) * 32];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sbuf + lt * 32 + 4 * g);
                    seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                }
            };
            // Two accumulator sets: tile lt+1's matrix ops are issued before tile lt's tracker runs, so the tracker never
            // waits for the results it reads and the matrix pipe has work queued while the vector ops issue.
            auto mm = [&](f32x16 (&acc)[2]) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[0][0], seed, 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[1][0], seed, 0, 0, 0);
#pragma unroll
                for (int q = 1; q < NQ; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[0][q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[1][q], acc[1], 0, 0, 0);
                }
            };
            auto track = [&](f32x16 (&acc)[2], int lt) {
                unsigned cell0 = (unsigned)(2 * ((c * TC + lt) & (kEpoch - 1))), cell1 = cell0 + 1u;
                asm volatile("" : "+s"(cell0), "+s"(cell1));
#pragma unroll
                for (int t = 0; t < 2; ++t) trk::tile(L[t], acc[t], cell0, cell1, keymask, ninf, pinf);
            };
            if (nt_here == TC) {                            // every chunk but a ragged last one: straight-line, two sets
                f32x16 acc[2][2];
                fetch(0);
                mm(acc[0]);
#pragma unroll
                for (int lt = 0; lt < TC; ++lt) {
                    if (lt + 1 < TC) { fetch(lt + 1); mm(acc[(lt + 1) & 1]); }
                    track(acc[lt & 1], lt);
                }
            } else {
                fetch(0);
#pragma unroll 1
                for (int lt = 0; lt < nt_here; ++lt) {
                    f32x16 acc[2];
                    mm(acc);
                    if (lt + 1 < nt_here) fetch(lt + 1);
                    track(acc, lt);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) asm volatile("" : "+v"(a[q]));
                }
            }
            // end of a key epoch (16 tiles) or of the codebook: fold the lane's three keys into its running top-3
            if ((((c + 1) * TC) & (kEpoch - 1)) == 0 || c + 1 == nchunk) {
                const int ebase = ((c * TC) / kEpoch) * kEpoch;          // first tile of this epoch
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float mv[3] = {L[t].m1, L[t].m2, L[t].m3};
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const unsigned kb = __float_as_uint(mv[i]);
                        const int code = 2 * ebase + (int)(kb & trk::kCellMask);      // GLOBAL cell id (2 * tile + half-tile)
                        const float v = mv[i];
                        const bool b1 = v > G1[t], b2 = v > G2[t], b3 = v > G3[t];
                        G3[t] = b2 ? G2[t] : (b3 ? v : G3[t]);
                        C2[t] = b1 ? C1[t] : (b2 ? code : C2[t]);
                        G2[t] = b1 ? G1[t] : (b2 ? v : G2[t]);
                        C1[t] = b1 ? code : C1[t];
                        G1[t] = b1 ? v : G1[t];
                    }
                    L[t].m1 = L[t].m2 = L[t].m3 = ninf;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next chunk have landed
            __syncthreads();                                       // everyone is done with this chunk; the next is visible
        }

        // ---- threshold per row, merge of the two lane halves, verdict (vq_track.h) ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float vA = trk::lane_max(L[t], ninf);
            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(vA), __float_as_uint(vA), false, false);
            const float v1 = trk::max3(vA, __uint_as_float(h ? sv[0] : sv[1]), ninf);
            const int rrel = r0 + 32 * t + l31;
            const bool valid = rrel < nrows;
            const float2 st = stat[valid ? rrel : 0];
            const float zs = st.x * 1.0001f;                                       // |z^|^2
            const float errz = __builtin_sqrtf(st.y * 1.0001f) * 1.0001f;          // |z - z^|, measured at conversion
            const float zn = __builtin_sqrtf(zs) * 1.0001f + errz;                 // |z| <= |z^| + |z - z^|
            const float mag = zn * Ehat + EEh;
            const float eps = errz * Ehat + (zn + errz) * dE + kGp * mag;
            const float xi = kG * zn * EmaxS + 1.2e-7f * (A * zn * zn + EEa);
            const float delta = (2.0f * eps + 2.0f * xi) * 1.001f;                 // no truncation term: the streams are exact
            const float thr = v1 - delta, thrB = thr - 8.0e-6f * mag;              // cell keys: 2^6 ulp <= 2^-17 mag below their cell's maximum
            const bool bad = cb_bad || !(zs < 1.0e30f) || !(st.y < 1.0e30f) || !(v1 > -1.0e37f) || !(v1 < 1.0e37f) || !(delta < 1.0e37f) ||
                             (v1 > -1.0e-30f && v1 < 1.0e-30f);
            const trk::Counts CN = trk::counts_of(L[t].S, G1[t], G2[t], G3[t], thr, thrB);
            const trk::Products P = trk::products_of(CN, C1[t], C2[t], h, K);
            // the two halves swap [popA : 4][nB : 2][k11 : 15]
            const unsigned mine = (unsigned)CN.popA | ((unsigned)CN.nB << 4) | ((unsigned)P.k[0] << 6);
            const auto so = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            const unsigned oth = h ? so[0] : so[1];
            const int popO = (int)(oth & 15u), nBO = (int)((oth >> 4) & 3u), k11O = (int)(oth >> 6);
            bool closed = (CN.popA + popO == 1) && (CN.nB + nBO == 1);
            int c1 = CN.popA == 1 ? P.k[0] : k11O;
            bool hard = bad || CN.popA > 2 || CN.nB > 2 || popO > 2 || nBO > 2;
            if (closed && (c1 < 0 || c1 >= K)) { closed = false; hard = true; }
            // open rows with no product at all on either side (cannot happen) -> every code exactly
            const auto sn = __builtin_amdgcn_permlane32_swap((unsigned)P.n, (unsigned)P.n, false, false);
            const int nO = (int)(h ? sn[0] : sn[1]);
            if (!closed && !hard && P.n + nO == 0) hard = true;
            hard = valid && hard;
            const bool open = valid && !hard && !closed;
            if (!closed || c1 < 0 || c1 >= K) c1 = 0;
            const bool writer = h == 0;
            if (valid && writer) idx[row0 + rrel] = c1;
            // open rows: one record per row, both halves' products (two uint4: {row, n0 | n1 << 8, k0 | k1 << 16, k2 | k3 << 16} of
            // half 0 and {k0 | k1 << 16, k2 | k3 << 16, 0, 0} of half 1); the resolve kernel evaluates every code exactly
            const unsigned long long om = __builtin_amdgcn_ballot_w64(open && writer);
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(hard && writer);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (om) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&counters[0], __builtin_popcountll(om));
                base = __builtin_amdgcn_readfirstlane(base);
                int slot = base + __builtin_popcountll(om & below);
                const auto ss = __builtin_amdgcn_permlane32_swap((unsigned)slot, (unsigned)slot, false, false);
                if (h) slot = (int)ss[0];                      // the row's slot, from its half-0 lane
                if (open) {
                    const unsigned k01 = (unsigned)P.k[0] | ((unsigned)P.k[1] << 16), k23 = (unsigned)P.k[2] | ((unsigned)P.k[3] << 16);
                    if (h == 0) open_list[2 * slot] = make_uint4(rel0 + (unsigned)rrel, (unsigned)P.n | ((unsigned)nO << 8), k01, k23);
                    else open_list[2 * slot + 1] = make_uint4(k01, k23, 0u, 0u);
                }
            }
            if (hm) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&counters[1], __builtin_popcountll(hm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (hard && writer) {
                    const int slot = base + __builtin_popcountll(hm & below);
                    hard_list[slot] = rel0 + (unsigned)rrel;
                    hard_best[slot] = ~0ull;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ATen cascade_sum order of sum(z**2) over one row read from memory (models/quantizer.py:50; same order as
// aten_sqsum_full, squares formed on the fly).
template <int D, typename P>
__device__ __forceinline__ float aten_sqsum_stream(P zrow) {
    constexpr int NV = D / 8, NI = NV / 4;
    float part[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 8; ++t) part[q][t] = 0.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float v = zrow[(4 * i + q) * 8 + t];
                part[q][t] = part[q][t] + v * v;
            }
#pragma unroll
    for (int v8 = NI * 4; v8 < NV; ++v8)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float v = zrow[v8 * 8 + t];
            part[0][t] = part[0][t] + v * v;
        }
    float fin = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float a = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
        fin = fin + a;
    }
    return fin;
}

// (d, k) "is better than" under torch.argmin: NaN is minimal, the first index wins among equals
__device__ __forceinline__ bool argmin_better(float da, int ka, float db, int kb) {
    const bool an = da != da, bn = db != db;
    if (an || bn) return an && (!bn || ka < kb);
    return da < db || (da == db && ka < kb);
}

// (d, k) -> 64-bit key whose unsigned order is torch.argmin's order: NaN first, then ascending distance, ties by index
__device__ __forceinline__ unsigned long long argmin_key(float d, int k) {
    unsigned u = 0u;
    if (d == d) {
        u = __float_as_uint(d + 0.0f);                    // -0 -> +0
        u = (u >> 31) ? ~u : (u | 0x80000000u);
    }
    return ((unsigned long long)u << 32) | (unsigned)k;
}

constexpr int kHardRangeTiles = 8;                        // 32-code tiles per work item of the hard-row path (256 codes)

// Pair tasks: one lane each, spread over the whole grid (a task is ~3 D dependent-latency loads: packing them into a few
// waves would leave most CUs idle).  Hard rows: batches of 32 rows are the 32 columns of exact fp32 MFMAs
// (v_mfma_f32_32x32x2_f32 is bit for bit the c-ordered fmaf chain, vq_exact.hip) against the fp32 A-operand image of
// vq_prepare_kernel; a work item is (batch, range of 256 codes); ranges fold their best (distance, index) into a 64-bit
// atomicMin whose order is torch.argmin's; the last range of a batch to finish writes the indices.
template <int D>
__global__ __launch_bounds__(256) void vq_stream_resolve_kernel(const float *__restrict__ z, const float *__restrict__ cb,
                                                                const float *__restrict__ ee, const float *__restrict__ img32,
                                                                int K, int KC, const uint4 *__restrict__ open_list,
                                                                const unsigned *__restrict__ hard_list,
                                                                unsigned long long *__restrict__ hard_best,
                                                                int *__restrict__ batch_done,
                                                                const int *__restrict__ counters, long long *__restrict__ idx) {
    const int npair = counters[0], nhard = counters[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- open rows: one lane per row, every product of its streams and cells exactly (first-index rule among equals) ----
    for (int i = blockIdx.x + gridDim.x * tid; i < npair; i += gridDim.x * 256) {
        const uint4 r0 = open_list[2 * i], r1 = open_list[2 * i + 1];
        const int n0 = (int)(r0.y & 0xffu), n1 = (int)((r0.y >> 8) & 0xffu);
        const unsigned kk[4] = {r0.z, r0.w, r1.x, r1.y};
        const float *zr = z + (size_t)r0.x * D;
        const float zz = aten_sqsum_stream<D>(zr);
        float bd = 0.0f;
        int bk = 0x7fffffff;
        for (int j = 0; j < 8; ++j) {
            const int nj = j < 4 ? n0 : n1;
            if ((j & 3) >= nj) continue;
            const int k = (int)((kk[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            const float *ea = cb + (size_t)k * D;
            float ma = 0.0f;
#pragma unroll 4
            for (int c4 = 0; c4 < D / 4; ++c4) {
                const f32x4 zv = *reinterpret_cast<const f32x4 *>(zr + 4 * c4);
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ea + 4 * c4);
                ma = __builtin_fmaf(zv.x, av.x, ma);
                ma = __builtin_fmaf(zv.y, av.y, ma);
                ma = __builtin_fmaf(zv.z, av.z, ma);
                ma = __builtin_fmaf(zv.w, av.w, ma);
            }
            const float ta = zz + ee[k];
            const float ua = 2.0f * ma;
            const float da = ta - ua;
            if (bk == 0x7fffffff || argmin_better(da, k, bd, bk)) { bd = da; bk = k; }
        }
        if (bk != 0x7fffffff) idx[r0.x] = bk;
    }
    // ---- hard rows ----
    if (nhard == 0) return;
    const int l31 = lane & 31, h = lane >> 5;
    const int ntile = (K + 31) / 32;
    const int G = (ntile + kHardRangeTiles - 1) / kHardRangeTiles;
    const int nb = (nhard + 31) / 32;
    for (int w = blockIdx.x * 4 + wave; w < nb * G; w += gridDim.x * 4) {
        const int b = w / G, g = w - b * G;
        const int slot = b * 32 + l31;
        const bool valid = slot < nhard;
        const unsigned rrel = hard_list[valid ? slot : 0];
        const float *zr_g = z + (size_t)rrel * D;
        // B operand: lane (row l31, k = h) holds channels c = 2 s + h, s = 0 .. D/2-1
        float zr[D / 2];
#pragma unroll
        for (int j = 0; j < D / 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(zr_g + 4 * j);
            zr[2 * j] = h ? v.y : v.x;
            zr[2 * j + 1] = h ? v.w : v.z;
        }
        const float zz = aten_sqsum_stream<D>(zr_g);
        float bd = 0.0f;
        int bk = 0x7fffffff;
        const int t1 = (g + 1) * kHardRangeTiles < ntile ? (g + 1) * kHardRangeTiles : ntile;
        for (int t = g * kHardRangeTiles; t < t1; ++t) {
            const int code0 = t * 32, ch = code0 / KC, kl0 = code0 - ch * KC;
            const float *ap = img32 + (size_t)ch * KC * D + ((size_t)h * KC + kl0 + l31) * 4;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int j = 0; j < D / 8; ++j) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(ap + (size_t)j * 2 * KC * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], zr[4 * j + i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int code = code0 + 8 * (r >> 2) + (r & 3) + 4 * h;
                if (code < K) {
                    const float tt = zz + ee[code];
                    const float d = __builtin_fmaf(-2.0f, acc[r], tt);           // = tt - fl(2 m): 2 m is exact
                    if (bk == 0x7fffffff || argmin_better(d, code, bd, bk)) { bd = d; bk = code; }
                }
            }
        }
        const float od = __shfl_xor(bd, 32);
        const int ok = __shfl_xor(bk, 32);
        if (ok != 0x7fffffff && (bk == 0x7fffffff || argmin_better(od, ok, bd, bk))) { bd = od; bk = ok; }
        if (valid && h == 0 && bk != 0x7fffffff) atomicMin(&hard_best[slot], argmin_key(bd, bk));
        __threadfence();
        int last = 0;
        if (lane == 0) last = atomicAdd(&batch_done[b], 1) == G - 1;
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            __threadfence();
            if (valid && h == 0) {
                const unsigned long long v = __hip_atomic_load(&hard_best[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                idx[rrel] = (long long)(unsigned)(v & 0xffffffffull);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// z_q = z + (e_k - z) (models/quantizer.py:67), squared error partial per block, histogram.  D / 4 lanes per row.
// The histogram is built in LDS and flushed once per block: a trained (or freshly initialised) model sends most rows to
// a few codes, and global atomics on one address serialise in L2 (58 ms for config 5's 4.2 M rows, measured).
template <int D>
__global__ __launch_bounds__(1024) void vq_stream_gather_kernel(const float *__restrict__ z, const float *__restrict__ cb,
                                                                const long long *__restrict__ idx, long long N, int K,
                                                                float *__restrict__ zq, int *__restrict__ hist,
                                                                double *__restrict__ partials, int *__restrict__ zq_amax, int hw) {
    constexpr int LPR = D / 4, RPB = 1024 / LPR;          // lanes per row, rows per block pass
    extern __shared__ int hist_s[];
    __shared__ double red[16];
    const int tid = threadIdx.x, j = tid % LPR, rsub = tid / LPR;
    for (int k = tid; k < K; k += 1024) hist_s[k] = 0;
    __syncthreads();
    double dacc = 0.0;
    // a block takes a CONTIGUOUS run of row passes (RPB rows each), two passes per iteration
    const long long npass = (N + RPB - 1) / RPB;
    const long long per = (npass + gridDim.x - 1) / gridDim.x;
    const long long p0 = (long long)blockIdx.x * per, p1 = p0 + per < npass ? p0 + per : npass;
    // zq_amax: per-image maximum of |z_q| for the decoder's first layer (the two-term fp16 products' scale), images of hw
    // consecutive rows: every lane keeps the running maximum of its current image and publishes it when any lane of the
    // wave moves on to another image (one atomic per wave where the wave was inside one image, per lane otherwise)
    int cur = -1;
    float m = 0.0f;
    long long img0 = 0;
    int rem0 = 0;
    const int q1 = RPB / hw, r1s = RPB % hw, q2 = (2 * RPB) / hw, r2s = (2 * RPB) % hw;
    if (zq_amax && p0 < p1) {
        const long long r = p0 * RPB + rsub;
        img0 = r / hw;
        rem0 = (int)(r - img0 * hw);
    }
    auto publish = [&]() {
        const int c0 = __builtin_amdgcn_readfirstlane(cur);
        if (__all(cur == c0)) {
            float mm = m;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o));
            if ((tid & 63) == 0 && c0 >= 0) atomicMax(zq_amax + c0, __float_as_int(mm));
        } else if (cur >= 0) {
            atomicMax(zq_amax + cur, __float_as_int(m));
        }
    };
    auto track = [&](bool valid, long long img, f32x4 v) {
        const int im = valid ? (int)img : cur;
        if (__any(im != cur)) {
            publish();
            cur = im;
            m = 0.0f;
        }
        if (valid) m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
    };
    for (long long ps = p0; ps < p1; ps += 2) {
        // two rows per iteration: two independent idx -> codebook chains in flight
        const long long r = ps * RPB + rsub, r1 = r + RPB;
        const bool one = r < N, two = ps + 1 < p1 && r1 < N;
        long long k0 = one ? idx[r] : 0, k1 = two ? idx[r1] : 0;
        if (k0 < 0 || k0 >= K) k0 = 0;                     // cannot happen; never read out of bounds
        if (k1 < 0 || k1 >= K) k1 = 0;
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 z0 = one ? *reinterpret_cast<const f32x4 *>(z + (size_t)r * D + 4 * j) : zero4;
        const f32x4 z1 = two ? *reinterpret_cast<const f32x4 *>(z + (size_t)r1 * D + 4 * j) : zero4;
        const f32x4 e0 = *reinterpret_cast<const f32x4 *>(cb + (size_t)k0 * D + 4 * j);
        const f32x4 e1 = *reinterpret_cast<const f32x4 *>(cb + (size_t)k1 * D + 4 * j);
        const f32x4 d0 = e0 - z0, d1 = e1 - z1;
        const f32x4 o0 = z0 + d0, o1 = z1 + d1;
        if (one) {
            float sq = d0.x * d0.x;
            sq = sq + d0.y * d0.y;
            sq = sq + d0.z * d0.z;
            sq = sq + d0.w * d0.w;
            dacc += (double)sq;
            if (zq) *reinterpret_cast<f32x4 *>(zq + (size_t)r * D + 4 * j) = o0;
            if (j == 0) atomicAdd(&hist_s[k0], 1);
        }
        if (two) {
            float sq1 = d1.x * d1.x;
            sq1 = sq1 + d1.y * d1.y;
            sq1 = sq1 + d1.z * d1.z;
            sq1 = sq1 + d1.w * d1.w;
            dacc += (double)sq1;
            if (zq) *reinterpret_cast<f32x4 *>(zq + (size_t)r1 * D + 4 * j) = o1;
            if (j == 0) atomicAdd(&hist_s[k1], 1);
        }
        if (zq_amax) {
            track(one, img0, o0);
            int rem1 = rem0 + r1s;
            const long long img1 = img0 + q1 + (rem1 >= hw ? 1 : 0);
            track(two, img1, o1);
            rem0 += r2s;
            img0 += q2;
            if (rem0 >= hw) { rem0 -= hw; ++img0; }
        }
    }
    if (zq_amax) publish();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
    if ((tid & 63) == 0) red[tid >> 6] = dacc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
    for (int k = tid; k < K; k += 1024) {
        const int c = hist_s[k];
        if (c) atomicAdd(&hist[k], c);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
bool vq_chunk_ok(int K, int D) { return (D == 64 || D == 128) && K >= 1 && K <= 16384; }

size_t vq_chunk_scratch_bytes(int D) {
    // per slab: the rows' fp16 image + statistics; per GROUP of slabs: the open rows' records (32 B per row), the hard rows'
    // list + running best (4 + 8 B per row), one batch counter per 32 rows, the two task counters
    const size_t grows = (size_t)kVqGroupSlabs * kVqSlabRows;
    return align_up((size_t)kVqSlabRows * D * 2, 256) + align_up((size_t)kVqSlabRows * 8, 256) +
           align_up(grows * 32, 256) + align_up(grows * 4, 256) + align_up(grows * 8, 256) + align_up((grows / 32) * 4, 256) + 256;
}

template <int D>
static int launch_chunked(const float *z, const float *cb, long long N, int K, float *zq, long long *idx, int *hist,
                          char *ws, hipStream_t st, int *grid_out, int *zq_amax, int hw) {
    const VqPlan p = vq_plan(K, D);
    constexpr int TC = ChunkCfg<D>::TC;
    char *s = ws + p.off_chunk;
    u32x4 *rows16 = reinterpret_cast<u32x4 *>(s);
    s += align_up((size_t)kVqSlabRows * D * 2, 256);
    float2 *stat = reinterpret_cast<float2 *>(s);
    s += align_up((size_t)kVqSlabRows * 8, 256);
    const size_t grows = (size_t)kVqGroupSlabs * kVqSlabRows;
    uint4 *pairs = reinterpret_cast<uint4 *>(s);                 // one 32-byte record per open row
    s += align_up(grows * 32, 256);
    unsigned *hards = reinterpret_cast<unsigned *>(s);
    s += align_up(grows * 4, 256);
    unsigned long long *hbest = reinterpret_cast<unsigned long long *>(s);
    s += align_up(grows * 8, 256);
    int *bdone = reinterpret_cast<int *>(s);
    s += align_up((grows / 32) * 4, 256);
    int *counters = reinterpret_cast<int *>(s);
    const int *flags = reinterpret_cast<const int *>(ws + p.off_flags);
    const float *ee = reinterpret_cast<const float *>(ws + p.off_ee);
    const u32x4 *img = reinterpret_cast<const u32x4 *>(ws + p.off_imgh);
    const float *seeds = reinterpret_cast<const float *>(ws + p.off_seeds);
    const int ntile = p.K32 / 32;
    const int cus = num_cus();
    const size_t lds = 2 * (size_t)(TC * D * 4 + 64 * ((TC * 8 + 63) / 64)) * 16;
    auto sweep = vq_stream_sweep_kernel<D>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(sweep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // slabs of 2^18 rows (their fp16 image stays cache-sized); the open and hard rows of up to sixteen slabs are resolved by
    // ONE launch -- a slab leaves ~7 000 of them, which a launch of its own turns into 90 us of latency (1.46 ms of config 5's
    // 35, round 3)
    for (long long g0 = 0; g0 < N; g0 += (long long)grows) {
        const long long gn = (N - g0) < (long long)grows ? (N - g0) : (long long)grows;
        for (long long row0 = g0; row0 < g0 + gn; row0 += kVqSlabRows) {
            const int nrows = (int)((g0 + gn - row0) < kVqSlabRows ? (g0 + gn - row0) : kVqSlabRows);
            const int nblk = (nrows + 511) / 512;
            const float *zs = z + (size_t)row0 * D;
            hipLaunchKernelGGL(vq_stream_rows16_kernel<D>, dim3(nblk * 16), dim3(256), 0, st, zs, (long long)nrows, rows16, stat, counters,
                               bdone + (row0 - g0) / 32, row0 == g0 ? 1 : 0);
            hipLaunchKernelGGL(sweep, dim3(nblk < cus ? nblk : cus), dim3(512), lds, st, rows16, stat, img, seeds, flags, nrows,
                               row0, K, ntile, idx, pairs, hards, hbest, counters, (unsigned)(row0 - g0));
        }
        hipLaunchKernelGGL(vq_stream_resolve_kernel<D>, dim3(4 * cus), dim3(256), 0, st, z + (size_t)g0 * D, cb, ee,
                           reinterpret_cast<const float *>(ws + p.off_img), K, p.KC, pairs, hards, hbest, bdone, counters,
                           idx + g0);
    }
    constexpr int kRowsPerPass = 1024 / (D / 4);
    long long g = (N + 2 * kRowsPerPass - 1) / (2 * kRowsPerPass);
    if (g > 2 * cus) g = 2 * cus;
    if (g > kVqMaxGrid) g = kVqMaxGrid;
    hipLaunchKernelGGL(vq_stream_gather_kernel<D>, dim3((unsigned)g), dim3(1024), (size_t)K * 4, st, z, cb, idx, N, K, zq,
                       hist, reinterpret_cast<double *>(ws + p.off_partials), zq ? zq_amax : nullptr, hw > 0 ? hw : 1);
    *grid_out = (int)g;
    return (int)hipGetLastError();
}

int launch_vq_chunked(const float *z, const float *cb, long long N, int K, int D, float *zq, long long *idx, int *hist,
                      char *ws, hipStream_t st, int *grid_out, int *zq_amax, int hw) {
    if (D == 64) return launch_chunked<64>(z, cb, N, K, zq, idx, hist, ws, st, grid_out, zq_amax, hw);
    if (D == 128) return launch_chunked<128>(z, cb, N, K, zq, idx, hist, ws, st, grid_out, zq_amax, hw);
    return VQVAE_ERR_UNSUPPORTED;
}

}  // namespace vqvae
