// Fused VectorQuantizer forward for gfx950, round 3: single-sweep fp16 screen with a STREAM TRACKER, exact refine
// (D = 64, row-major rows, K <= 512 so that the codebook image stays resident in LDS next to eight waves' tiles).
//
// Same contract and the same bits out as vq_exact.hip / vq_sweep.hip (indices and z_q bit-identical to the reference,
// models/quantizer.py:45-74).  What changes against round 2's vq_sweep_kernel_d64 -- which was bound by vector-instruction
// issue (17 M vector instructions per 262 144 rows against 0.5 M MFMAs, profiles/r02_pmc_sq.txt):
//   * the sweep tracks per-lane maxima over two partitions of the accumulator values (8 "streams" by position in the code
//     tile, 2 "cells" per code tile as the three largest keys) instead of the three largest values with their code index:
//     24 vector instructions per 16 screened values instead of 80, and no index bits in the tracked values, so DELTA has
//     no truncation term (vq_track.h has the argument why stream x cell names every code at or above the threshold);
//   * rows are converted to fp16 with two v_cvt_pk_f16_f32 per 16 bytes and nothing else; |z - z^| enters DELTA as the
//     worst-case 2^-11 |z| (measuring it cost more vector instructions than the open rows it saved);
//   * open rows (about 3 % on the reference's own z_e distribution) become exact TASKS (row, code a, code b) straight from
//     the classification -- the products of the streams and cells at or above the threshold -- four tasks per pass, one per
//     16-lane group; rows whose candidates the products cannot cover (three or more streams or cells of one lane at or above
//     the threshold, ~0.01 %) have their row tile screened again against the now-known threshold, as in round 2;
//   * the epilogue has no refine of its own: gather, z + (e_k - z), squared error, stores.
//
// Bound (accumulator units; A = 2^a_e the codebook scale, e' = A e, e^ = fp16(e'), z^ = fp16(z), u = 2^-11):
//   errz := |z - z^| <= u |z| + 2^-22 (the second term covers fp16-subnormal channels, 64 x 2^-25 each at most);
//           |z| <= zn := |z^| + errz
//   eps, xi as in vq_sweep.hip:24-35;  DELTA = 2 eps + 2 xi.  A cell key differs from its cell's maximum by less than
//   2^6 ulp <= 2^-17 (zn Ehat + EEh) =: tB < DELTA / 2, and keys are compared against thr - tB.
#include "common.h"
#include "vq_device.h"
#include "vq_track.h"

namespace vqvae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ void lds_order_wave() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
}  // namespace

// Eight waves per workgroup, one workgroup per CU.  A wave owns UNITS of two 32-row tiles (64 consecutive rows) that share
// every codebook operand and seed read from LDS; its first unit is static, later units come from an LDS ticket.  Rows stay
// in registers in the coalesced load layout (16 lanes x 16 bytes per row) from load to store: HBM traffic is the
// algorithmic 520 B per row.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void vq_track_kernel_d64(
    const float *__restrict__ z, const float *__restrict__ cb, const uint4 *__restrict__ img_g,
    const float *__restrict__ seeds_g, const float *__restrict__ ee_g, const int *__restrict__ flags,
    long long N, int K, int K32, long long nunits, float *__restrict__ zq, long long *__restrict__ idx,
    int *__restrict__ hist, double *__restrict__ partials) {
    constexpr int D = 64, T = 2, RU = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ntile = K32 >> 5;
    uint4 *Eimg = reinterpret_cast<uint4 *>(smem_raw);                                  // [ntile][4][2][32] x 16 B
    float *seeds = reinterpret_cast<float *>(Eimg + (size_t)ntile * 256);               // [ntile][2][16]
    int *hist_s = reinterpret_cast<int *>(seeds + (size_t)ntile * 32);                  // [K]
    double *red = reinterpret_cast<double *>(hist_s + (K + 3) / 4 * 4);                 // [NW]
    int *ticket_s = reinterpret_cast<int *>(red + NW);                                  // next unit of this workgroup (+ pad)
    unsigned char *wave_base = reinterpret_cast<unsigned char *>(red + NW + 2);         // per wave: 8 KiB tile + tables

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int TILEB = 4096 * T, TABB = 1552;
    unsigned char *tile_s = wave_base + (size_t)wave_u * (TILEB + TABB);                // the unit's fp16 rows; later 16 fp32 row slots
    unsigned char *tab_s = tile_s + TILEB;

#ifdef VQ_SWEEP_TIMING
    // debug build (tools/build_variant.py NAME -DVQ_SWEEP_TIMING, tools/vq_phase.py): per-phase wall-clock sums (100 MHz
    // ticks) over the waves of the first 64 workgroups, collected in LDS and written to the spare tail of `partials`
    unsigned *tsum = reinterpret_cast<unsigned *>(red);       // the loss scratch is not used before the loop ends
    if (tid < 8) tsum[tid] = 0;
    unsigned long long tprev = wall_clock64();
    const unsigned long long tstart = tprev;
#define VQ_STAMP(slot)                                                          \
    do {                                                                        \
        const unsigned long long tnow = wall_clock64();                         \
        if ((tid & 63) == 0) atomicAdd(&tsum[slot], (unsigned)(tnow - tprev));  \
        tprev = tnow;                                                           \
    } while (0)
#else
#define VQ_STAMP(slot) do {} while (0)
#endif

    const int cb_bad = flags[0];
    const int a_e = flags[5];
    const float A = __builtin_ldexpf(1.0f, a_e);
    const float EEmax = __int_as_float(flags[1]) * 1.0001f;               // max ee_k (unscaled)
    const float Ehat = __builtin_sqrtf(__int_as_float(flags[3])) * 1.0001f;
    const float dE = __builtin_sqrtf(__int_as_float(flags[4])) * 1.0001f;
    const float EmaxS = __builtin_sqrtf(EEmax) * A * 1.0001f;
    const float EEh = 0.5f * EEmax * A, EEa = EEmax * A;

    // ---- row I/O: F[t][i] = floats 4 j16 .. +3 of row 32 t + 4 i + g4 of the unit (1 KiB contiguous per instruction) ----
    // a buffer descriptor over the unit's 16 KiB clipped at the end of z: rows past the end read zeros (their results are
    // never stored), one 32-bit lane offset serves all 16 loads
    auto load_unit = [&](long long p, f32x4(&F)[T][8], int lane) {
        const long long left = (N - p * RU) * (D * 4);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(z + (size_t)p * RU * D), 0,
                                                          (unsigned)(left < RU * 256 ? left : RU * 256), 0x00020000);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                F[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 16u, (unsigned)(t * 8 + i) * 1024u, 0));
    };

    const long long pstride = (long long)gridDim.x * NW;
    long long p = (long long)blockIdx.x * NW + wave_u;
    f32x4 F[T][8];
    if (p < nunits) load_unit(p, F, tid & 63);

    // codebook image and seeds -> LDS (eight 16-byte requests in flight per thread)
    {
        const u32x4 *src16 = reinterpret_cast<const u32x4 *>(img_g);
        u32x4 *dst16 = reinterpret_cast<u32x4 *>(Eimg);
        const int n16 = ntile * 256;
        // every workgroup reads the same 64 KiB: each starts at its own offset so the CUs do not queue on the same lines
        const int rot = (int)((blockIdx.x * 97u) % (unsigned)ntile) * 256;
        for (int i0 = 0; i0 < n16; i0 += 8 * NW * 64) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * NW * 64 + tid;
                v[j] = src16[i < n16 ? (i + rot) % n16 : 0];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j]));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * NW * 64 + tid;
                if (i < n16) dst16[(i + rot) % n16] = v[j];
            }
        }
    }
    for (int i = tid; i < ntile * 32; i += NW * 64) seeds[i] = seeds_g[i];
    for (int k = tid; k < K; k += NW * 64) hist_s[k] = 0;
    if (tid == 0) ticket_s[0] = NW;                          // units 0 .. NW-1 of the workgroup are taken statically
    __syncthreads();

    const float inf = __builtin_inff();
    VQ_STAMP(0);                                               // codebook image copy + first row requests
    float pinf = inf, ninf = -inf;                           // opaque: see vq_track.h
    unsigned keymask = trk::kKeyMask;
    asm volatile("" : "+v"(pinf), "+v"(ninf), "+v"(keymask));
    double dacc = 0.0;

    while (p < nunits) {
        const long long r0 = p * RU;
        // lane-derived indices are made opaque once per iteration: hipcc otherwise hoists dozens of per-lane address values
        // out of this loop, spills them and reloads them from scratch inside it
        int lane_v = tid & 63;
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v, l31 = lane_v & 31, h = lane_v >> 5, j16 = lane_v & 15, g4 = lane_v >> 4;
        const uint4 *ap0 = Eimg + h * 32 + l31;
        const float *sp0 = seeds + h * 16;

        // ================= fp32 rows -> fp16 B operands through the wave's LDS tile, |z^|^2 ============================
        // tile: row r at 128 r, its 16-byte chunk c at slot c ^ ((r >> 1) & 7) (conflict-free for both access patterns);
        // row 4 i + g4, chunk j16 >> 1: the slot is (j16 >> 1) ^ (g4 >> 1) ^ 2 (i & 3) -- one lane constant, one immediate
        f16x8 zb[T][4];
        float zn2[T];
        {
            const unsigned wbase = (unsigned)g4 * 128u + ((((unsigned)j16 >> 1) ^ ((unsigned)g4 >> 1)) << 4) + (((unsigned)j16 & 1u) << 3);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x2 lo2 = {F[t][i].x, F[t][i].y}, hi2 = {F[t][i].z, F[t][i].w};
                    u32x2 w;
                    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, f16x2));
                    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi2, f16x2));
                    *reinterpret_cast<u32x2 *>(tile_s + t * 4096 + i * 512 + (wbase ^ ((unsigned)(2 * (i & 3)) << 4))) = w;
                }
            lds_order_wave();
            const unsigned rbase = (unsigned)l31 * 128u + ((((unsigned)h ^ ((unsigned)l31 >> 1)) & 7u) << 4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float s = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(tile_s + t * 4096 + (rbase ^ ((unsigned)(2 * q) << 4)));
                    zb[t][q] = __builtin_bit_cast(f16x8, v);
                    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v.x), __builtin_bit_cast(f16x2, v.x), s, false);
                    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v.y), __builtin_bit_cast(f16x2, v.y), s, false);
                    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v.z), __builtin_bit_cast(f16x2, v.z), s, false);
                    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, v.w), __builtin_bit_cast(f16x2, v.w), s, false);
                }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
                zn2[t] = s + __uint_as_float(h ? sw[0] : sw[1]);
            }
        }

        VQ_STAMP(1);                                           // rows landed, fp16 conversion
        // ================= the sweep: 4 MFMAs per (code tile, row tile), stream / cell maxima per lane ====================
        trk::Lane L[T];
#pragma unroll
        for (int t = 0; t < T; ++t) trk::init(L[t], ninf);
        {
            // operands of tile ct+1 are requested right behind the MFMAs of tile ct and land under its vector work; two
            // operand sets ping-pong through a loop unrolled by two, so nothing is copied
            auto fetch = [&](int ct, u32x4(&a)[4], f32x16 &seed) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const u32x4 *>(ap0 + ct * 256 + q * 64);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sp0 + ct * 32 + 4 * g);
                    seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                }
            };
            auto cell = [&](int ct, const u32x4(&a)[4], const f32x16 &seed, u32x4(&an)[4], f32x16 &seedn) {
                f32x16 acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[t][0], seed, 0, 0, 0);
#pragma unroll
                    for (int q = 1; q < 4; ++q)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[t][q], acc[t], 0, 0, 0);
                }
                fetch(ct + 1 < ntile ? ct + 1 : ct, an, seedn);
                unsigned cell0 = (unsigned)(2 * ct), cell1 = cell0 + 1u;        // scalars (opaque: else or3(x & mask, cell0, 1))
                asm volatile("" : "+s"(cell0), "+s"(cell1));
#pragma unroll
                for (int t = 0; t < T; ++t) trk::tile(L[t], acc[t], cell0, cell1, keymask, ninf, pinf);
                // the prefetched operands are first "used" here: their loads cannot sink below, their wait cannot rise above
                asm volatile("" : "+v"(an[0]), "+v"(an[1]), "+v"(an[2]), "+v"(an[3]));
            };
            u32x4 aA[4], aB[4];
            f32x16 sA, sB;
            fetch(0, aA, sA);
            int ct = 0;
            for (; ct + 1 < ntile; ct += 2) {
                cell(ct, aA, sA, aB, sB);
                cell(ct + 1, aB, sB, aA, sA);
            }
            if (ct < ntile) cell(ct, aA, sA, aB, sB);
        }

        VQ_STAMP(2);                                           // sweep
        // ================= threshold, merge of the two lane halves of every row, verdict ===================================
        int kbest[T];
        bool valid[T], bad[T], openf[T], hardf[T];
        float thr[T];
        unsigned *task_s = reinterpret_cast<unsigned *>(tab_s);                  // [64] row | a << 6 | b << 19
        int ncls = 0;                                                            // tasks written by the classification
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            valid[t] = r0 + 32 * t + l31 < N;
            const float vA = trk::lane_max(L[t], ninf);
            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(vA), __float_as_uint(vA), false, false);
            const float v1 = trk::max3(vA, __uint_as_float(h ? sv[0] : sv[1]), ninf);
            // DELTA in accumulator units, every factor rounded up
            const float zs = zn2[t] * 1.0001f;                                         // |z^|^2
            const float zh = __builtin_sqrtf(zs) * 1.0001f;                            // |z^|
            const float errz = zh * 4.89e-4f + 2.5e-7f;                                // |z - z^| <= u |z| + 2^-22, |z| <= |z^| / (1 - u)
            const float zn = zh + errz;                                                // |z| <= |z^| + |z - z^|
            const float mag = zn * Ehat + EEh;                                         // bounds every |acc|
            const float eps = errz * Ehat + (zn + errz) * dE + 7.76e-6f * mag;
            const float xi = 3.86e-6f * zn * EmaxS + 1.2e-7f * (A * zn * zn + EEa);         // g = 64 * 2^-24 * 1.01; 2^-23
            const float delta = (2.0f * eps + 2.0f * xi) * 1.001f;
            const float th = v1 - delta;
            thr[t] = th;
            // |v1| below 1e-30: a key could be a denormal whose cell field a flush would lose -- never on real data
            bad[t] = valid[t] && (cb_bad || !(zs < 1.0e30f) || !(v1 > -1.0e37f) || !(v1 < 1.0e37f) || !(delta < 1.0e37f) ||
                                  (v1 > -1.0e-30f && v1 < 1.0e-30f));
            const trk::Half H = trk::half_of(L[t], th, th - 8.0e-6f * mag, h);
            const unsigned mine = trk::pack(H);
            const auto so = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
            const trk::Verdict V = trk::verdict_of(H, h ? so[0] : so[1], K);
            const bool live = valid[t] && !bad[t];
            openf[t] = live && !V.closed && !V.hard;
            hardf[t] = live && V.hard;
            kbest[t] = V.closed ? V.kbest : 0;
            if (__builtin_amdgcn_ballot_w64(openf[t])) {
                // open rows: this half's exact tasks
                const trk::Cands C = trk::cands_of(L[t], H, h, K);
                const int nt = openf[t] ? C.ntask : 0;
                const unsigned long long b1 = __builtin_amdgcn_ballot_w64(nt >= 1), b2 = __builtin_amdgcn_ballot_w64(nt >= 2);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0)) +
                                  __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
                const int slot = ncls + below;
                const unsigned rowu = (unsigned)(32 * t + l31);
                if (nt >= 1 && slot < 64) task_s[slot] = rowu | ((unsigned)C.ta[0] << 6) | ((unsigned)C.tb[0] << 19);
                if (nt >= 2 && slot + 1 < 64) task_s[slot + 1] = rowu | ((unsigned)C.ta[1] << 6) | ((unsigned)C.tb[1] << 19);
                ncls += __builtin_popcountll(b1) + __builtin_popcountll(b2);
            }
        }

        VQ_STAMP(3);                                           // threshold + verdict
        // ================= exact part (rows the screen left open) =========================================================
        // A TASK is (row, code a, code b); four tasks run per pass, one per 16-lane group, on the row's fp32 data (read again
        // from L2: it was loaded a few microseconds ago): ||z||^2 in ATen's summation order and the two c-ordered fmaf
        // chains, all with DPP row operations, then d = fl(fl(zz + ee_k) - 2 m).  Each row takes the lexicographic (d, k)
        // minimum over its tasks = torch.argmin's first-index rule, folded by a 64-bit LDS atomic minimum.
        //   open rows: the tasks the classification wrote (products of streams and cells at or above the threshold)
        //   hard rows: the row tile's screen is run again with the row's now-known threshold and every code at or above
        //       it becomes a task (same accumulators as in the sweep)
        //   non-finite rows / unusable codebooks / task overflow: scalar torch.argmin semantics, one lane per row
        {
            // lane L of the wave speaks for row L of the unit (tile L >> 5, row L & 31)
            const bool o_open = h ? openf[1] : openf[0];
            const bool o_hard = h ? hardf[1] : hardf[0];
            bool o_bad = h ? bad[1] : bad[0];
            const unsigned long long fm = __builtin_amdgcn_ballot_w64(o_open || o_hard || o_bad);
            if (fm) {
                unsigned long long *best_s = reinterpret_cast<unsigned long long *>(tab_s + 256);   // [64] (distance, index) minimum per row
                float *zz_s = reinterpret_cast<float *>(tab_s + 768);                    // [64] ||z||^2 per row of the unit
                int *cnt_s = reinterpret_cast<int *>(tab_s + 1024);                      // counter of the rescan's tasks
                const unsigned long long lowmask = (1ull << lane) - 1ull;
                best_s[lane] = ~0ull;
                // non-finite rows: one task each, for the row's ||z||^2
                const unsigned long long tmb = __builtin_amdgcn_ballot_w64(o_bad);
                if (o_bad && ncls + __builtin_popcountll(tmb & lowmask) < 64) task_s[ncls + __builtin_popcountll(tmb & lowmask)] = (unsigned)lane;
                const int ndirect = ncls + __builtin_popcountll(tmb);
                int ntasks = ndirect;
                const unsigned long long hmask = __builtin_amdgcn_ballot_w64(o_hard);
                if (hmask && ndirect <= 64) {
                    // rows with candidates the products do not cover: the tile's screen again, hits (acc >= v1 - DELTA) become tasks
                    if (lane == 0) cnt_s[0] = 0;
                    lds_order_wave();
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if ((unsigned)(hmask >> (32 * t))) {
                            // B operands again from the tile (they need not stay in registers through the classification)
                            const unsigned rb = (unsigned)l31 * 128u + ((((unsigned)h ^ ((unsigned)l31 >> 1)) & 7u) << 4);
                            f16x8 zbr[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                zbr[q] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(tile_s + t * 4096 + (rb ^ ((unsigned)(2 * q) << 4))));
                            u32x4 ra[4];
                            f32x16 rs;
                            auto rfetch = [&](int ct) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const u32x4 *>(ap0 + ct * 256 + q * 64);
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sp0 + ct * 32 + 4 * g);
                                    rs[4 * g] = e4.x; rs[4 * g + 1] = e4.y; rs[4 * g + 2] = e4.z; rs[4 * g + 3] = e4.w;
                                }
                            };
                            rfetch(0);
                            const float thr_t = hardf[t] ? thr[t] : inf;             // only the hard rows can hit
                            for (int ct = 0; ct < ntile; ++ct) {
                                f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), zbr[0], rs, 0, 0, 0);
#pragma unroll
                                for (int q = 1; q < 4; ++q)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[q]), zbr[q], acc, 0, 0, 0);
                                rfetch(ct + 1 < ntile ? ct + 1 : ct);
                                const float x0 = trk::max3(trk::max3(acc[0], acc[1], acc[2]), trk::max3(acc[3], acc[4], acc[5]), trk::max3(acc[6], acc[7], acc[8]));
                                const float x1 = trk::max3(trk::max3(acc[9], acc[10], acc[11]), trk::max3(acc[12], acc[13], acc[14]), acc[15]);
                                const float mx = trk::max3(x0, x1, ninf);
                                if (__builtin_amdgcn_ballot_w64(mx >= thr_t)) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) {
                                        const int code = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                                        if (acc[r] >= thr_t && code < K) {
                                            const int sl = ndirect + atomicAdd(&cnt_s[0], 1);
                                            if (sl < 64) task_s[sl] = (unsigned)(32 * t + l31) | ((unsigned)code << 6) | ((unsigned)code << 19);
                                        }
                                    }
                                }
                                asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]));
                            }
                        }
                    }
                    lds_order_wave();
                    ntasks = ndirect + cnt_s[0];
                }
                if (ntasks > 64) {                      // pathological tie counts: every open row takes the scalar path;
                    o_bad = o_bad || o_open || o_hard;  // the tasks only produce its ||z||^2 (<= 64 rows, so they fit)
                    __builtin_amdgcn_wave_barrier();
                    if (o_bad) task_s[__builtin_popcountll(fm & lowmask)] = (unsigned)lane;
                    ntasks = __builtin_popcountll(fm);
                }
                lds_order_wave();
                const long long left = (N - r0) * (D * 4);
                const auto zr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(z + (size_t)r0 * D), 0,
                                                                     (unsigned)(left < RU * 256 ? left : RU * 256), 0x00020000);
                for (int base = 0; base < ntasks; base += 4) {
                    const int jj = base + g4;
                    const unsigned task = task_s[jj < ntasks ? jj : 0];
                    const int rr = (int)(task & 63u), ka = (int)((task >> 6) & 8191u), kb2 = (int)(task >> 19);
                    const f32x4 zv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr_rs, (unsigned)rr * 256u + (unsigned)j16 * 16u, 0, 0));
                    const f32x4 ea = *reinterpret_cast<const f32x4 *>(cb + (size_t)ka * D + 4 * j16);
                    const f32x4 eb = *reinterpret_cast<const f32x4 *>(cb + (size_t)kb2 * D + 4 * j16);
                    const float eea = ee_g[ka], eeb = ee_g[kb2];
                    // ||z||^2 in ATen's order (lane j16 holds elements 4 j16 .. +3): P = v_q + v_{q+4} (lane j + lane j+8),
                    // A = ((P0 + P1) + P2) + P3 (lanes b, b+2, b+4, b+6), then A0..A7 in order (lane 0, then lane 1)
                    float Aq[4];
                    const float sq[4] = {zv.x * zv.x, zv.y * zv.y, zv.z * zv.z, zv.w * zv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float P = sq[e] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq[e]), 0x108, 0xf, 0xf, true));
                        const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P), 0x102, 0xf, 0xf, true));
                        const float p2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P), 0x104, 0xf, 0xf, true));
                        const float p3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P), 0x106, 0xf, 0xf, true));
                        Aq[e] = ((P + p1) + p2) + p3;                                 // valid on lanes 0, 1 of the group
                    }
                    const float fin0 = (((0.0f + Aq[0]) + Aq[1]) + Aq[2]) + Aq[3];   // lane 0: A0..A3
                    const float f0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fin0), 0x111, 0xf, 0xf, true));
                    const float zz = (((f0 + Aq[0]) + Aq[1]) + Aq[2]) + Aq[3];       // lane 1: + A4..A7
                    // c-ordered fmaf chains: lane j continues lane j-1's partial sum (row_shr:1, 0 enters lane 0)
                    float ma = 0.0f, mb = 0.0f;
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) {
                        const float ia = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ma), 0x111, 0xf, 0xf, true));
                        const float ib = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mb), 0x111, 0xf, 0xf, true));
                        ma = __builtin_fmaf(zv.w, ea.w, __builtin_fmaf(zv.z, ea.z, __builtin_fmaf(zv.y, ea.y, __builtin_fmaf(zv.x, ea.x, ia))));
                        mb = __builtin_fmaf(zv.w, eb.w, __builtin_fmaf(zv.z, eb.z, __builtin_fmaf(zv.y, eb.y, __builtin_fmaf(zv.x, eb.x, ib))));
                    }
                    const float ma1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ma), 0x122, 0xf, 0xf, true));   // lane 1 <- lane 15
                    const float mb1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mb), 0x122, 0xf, 0xf, true));
                    const float da = (zz + eea) - 2.0f * ma1, db = (zz + eeb) - 2.0f * mb1;     // valid on lane 1
                    if (j16 == 1 && jj < ntasks) {
                        atomicMin(&best_s[rr], trk::dist_key(da, ka));
                        atomicMin(&best_s[rr], trk::dist_key(db, kb2));
                        zz_s[rr] = zz;
                    }
                }
                lds_order_wave();
                int o_best = 0;
                if ((o_open || o_hard) && !o_bad) {
                    const unsigned long long bk = best_s[lane];
                    if (bk != ~0ull) o_best = (int)(unsigned)bk; else o_bad = true;   // no task came back (cannot happen): scalar path
                }
                if (o_bad) {
                    // torch.argmin semantics (NaN is minimal, first index wins), one lane per row
                    const long long grow = r0 + lane;
                    const float *zr = z + (size_t)(grow < N ? grow : N - 1) * D;
                    const float zz = zz_s[lane];                                      // every flagged row had a task
                    int best = 0;
                    if (zz == zz) {                                                   // NaN ||z||^2: every distance is NaN -> index 0
                        float bd = 0.0f;
                        for (int k = 0; k < K; ++k) {
                            float m = 0.0f;
                            for (int c = 0; c < D; ++c) m = __builtin_fmaf(zr[c], cb[(size_t)k * D + c], m);
                            const float d = (zz + ee_g[k]) - 2.0f * m;
                            const bool dn = d != d, bn = bd != bd;
                            if ((k == 0) || (dn ? !bn : (!bn && d < bd))) { best = k; bd = d; }
                        }
                    }
                    o_best = best;
                }
                const int k0n = __shfl(o_best, l31), k1n = __shfl(o_best, 32 + l31);
                if (openf[0] || hardf[0] || bad[0]) kbest[0] = k0n;
                if (openf[1] || hardf[1] || bad[1]) kbest[1] = k1n;
                __builtin_amdgcn_wave_barrier();
            }
        }

        VQ_STAMP(4);                                           // exact part
        // ================= epilogue: gather, z + (e_k - z), squared error, index, histogram ==================================
        {
            const auto cb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cb), 0, (unsigned)K * (D * 4), 0x00020000);
            f32x4 ev[T][8];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kr = __builtin_amdgcn_ds_bpermute((4 * i + g4) << 2, kbest[t]);
                    ev[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cb_rs, (unsigned)kr * (D * 4) + (unsigned)j16 * 16u, 0, 0));
                }
            const int nleft = (int)(N - r0 < RU ? N - r0 : RU);         // rows of this unit that exist
            // the descriptor covers exactly the unit's existing rows: stores of rows past the end are dropped by the hardware
            const auto zq_rs = __builtin_amdgcn_make_buffer_rsrc(zq ? zq + (size_t)p * RU * D : const_cast<float *>(z), 0,
                                                                 zq ? (unsigned)nleft * (D * 4) : 0u, 0x00020000);
            // Store offsets: four lane bases 4 KiB apart + an immediate, NO scalar offset register.  hipcc (ROCm 7.2) does not
            // guard a 16-byte buffer store whose soffset is an SGPR against the next vector instruction overwriting its data
            // registers (LLVM exempts that form from the store-data hazard); on gfx950 the overwrite corrupted the last dword
            // of lanes 12..15 of each row here.  Without an soffset register the compiler inserts the wait states itself
            // (tools/hazard_scan.py checks the assembly of every source for this pattern; tests/test_build_hazards.py runs it).
            unsigned vo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vo[k] = (unsigned)lane * 16u + 4096u * k;
                asm volatile("" : "+v"(vo[k]));
            }
            float sacc = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 zv = F[t][i], e = ev[t][i];
                    const float d0 = e.x - zv.x, d1 = e.y - zv.y, d2 = e.z - zv.z, d3 = e.w - zv.w;
                    f32x4 o;
                    o.x = zv.x + d0; o.y = zv.y + d1; o.z = zv.z + d2; o.w = zv.w + d3;
                    const float sq = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
                    if (nleft == RU) sacc += sq;                   // fp32 over the unit's 16 groups, one fp64 add per unit
                    else sacc += 32 * t + 4 * i + g4 < nleft ? sq : 0.0f;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), zq_rs, vo[(t * 8 + i) >> 2] + (unsigned)((t * 8 + i) & 3) * 1024u, 0, 0);
                }
            dacc += (double)sacc;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (valid[t] && h == 0) {
                    idx[r0 + 32 * t + l31] = kbest[t];
                    atomicAdd(&hist_s[kbest[t]], 1);
                }
        }
        VQ_STAMP(5);                                           // epilogue
        {
            int q = 0;
            if (lane == 0) q = atomicAdd(ticket_s, 1);
            q = __builtin_amdgcn_readfirstlane(q);
            p = (long long)(q / NW) * pstride + (long long)blockIdx.x * NW + (q % NW);
            if (p < nunits) load_unit(p, F, lane);
        }
    }

#ifdef VQ_SWEEP_TIMING
    VQ_STAMP(6);
    if ((tid & 63) == 0) atomicMax(&tsum[7], (unsigned)(wall_clock64() - tstart));     // slowest wave of the workgroup
    __syncthreads();
    if (tid < 8 && blockIdx.x < 64) reinterpret_cast<unsigned long long *>(partials + 512)[blockIdx.x * 8 + tid] = tsum[tid];
    __syncthreads();
#endif
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
    __syncthreads();
    if ((tid & 63) == 0) red[wave_u] = dacc;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
    for (int k = tid; k < K; k += NW * 64) {
        const int c = hist_s[k];
        if (c) atomicAdd(&hist[k], c);
    }
}

size_t vq_track_lds_bytes(int K) {
    const int K32 = (K + 31) / 32 * 32;
    return (size_t)K32 * 128 + (size_t)K32 * 4 + (size_t)((K + 3) / 4 * 4) * 4 + 8 * 8 + 16 + 8 * (size_t)(8192 + 1552);
}

bool vq_track_ok(int K, int D) { return D == 64 && K <= 1024 && vq_track_lds_bytes(K) <= (size_t)kLdsBytes; }

int launch_vq_track_d64(const float *z, const float *cb, long long N, int K, float *zq, long long *idx, int *hist,
                        char *ws, hipStream_t st, int *grid_out) {
    const VqPlan p = vq_plan(K, 64);
    const int cus = num_cus();
    constexpr int NW = 8;
    const long long nunits = (N + 63) / 64;
    long long grid = (nunits + NW - 1) / NW;
    if (grid > cus) grid = cus;
    if (grid > kVqMaxGrid) grid = kVqMaxGrid;
    *grid_out = (int)grid;
    auto kfn = vq_track_kernel_d64<NW>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NW * 64), vq_track_lds_bytes(K), st, z, cb,
                       reinterpret_cast<const uint4 *>(ws + p.off_imgh), reinterpret_cast<const float *>(ws + p.off_seeds),
                       reinterpret_cast<const float *>(ws + p.off_ee), reinterpret_cast<const int *>(ws + p.off_flags), N, K,
                       p.K32, nunits, zq, idx, hist, reinterpret_cast<double *>(ws + p.off_partials));
    return (int)hipGetLastError();
}

}  // namespace vqvae
