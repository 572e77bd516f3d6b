// Fused VectorQuantizer forward for gfx950, round 3: single-sweep fp16 screen with a STREAM TRACKER, exact refine
// (D = 64; the codebook's fp16 image resident in LDS next to the waves' row tiles: K <= ~600 beside eight or sixteen waves' tiles,
// row-major rows or the module's own NCHW layout; up to K = 1024 beside four waves' tiles, row-major rows).
//
// Same contract and the same bits out as vq_exact.hip (indices and z_q bit-identical to the reference,
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
//   eps, xi as in vq_prepare16.hip (header);  DELTA = 2 eps + 2 xi.  A cell key differs from its cell's maximum by less than
//   2^6 ulp <= 2^-17 (zn Ehat + EEh) =: tB < DELTA / 2, and keys are compared against thr - tB.
#include "common.h"
#include "vq_device.h"
#include "vq_track.h"
#include "vq_unit.h"
#include <type_traits>
#include <hip/hip_ext.h>          // hipExtLaunchKernelGGL: start / stop events on the dispatch itself (prof_dispatch)

namespace vqvae {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

using vqu::lds_order_wave;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Debug builds (tools/build_vq_variant.sh NAME -DVQ_TRACE, tools/ubench/vq_ab.cpp): absolute 100 MHz stamps of every wave --
// [0] wave start, [1] past the prologue's barrier, [2..5] end of its first four units, [6] loop exit, [7] last instruction
#ifdef VQ_TRACE2                 // slots 2..5 = end of the sweep / classification / exact part / epilogue of the wave's FIRST unit
#define VQ_TRACE 1
#endif
#ifdef VQ_TRACE
__device__ unsigned long long g_vq_trace[4096 * 8];
#define VQ_TR(slot) do { if ((tid & 63) == 0) g_vq_trace[((size_t)blockIdx.x * NW + wave_u) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define VQ_TR(slot) do {} while (0)
#endif

// Eight waves per workgroup, one workgroup per CU.  A wave owns UNITS of two 32-row tiles (64 consecutive rows) that share
// every codebook operand and seed read from LDS; its first unit is static, later units come from an LDS ticket.  Rows stay
// in registers in the coalesced load layout (16 lanes x 16 bytes per row) from load to store: HBM traffic is the
// algorithmic 520 B per row.
// NCHW (round 4): z / z_q are (B, 64, HW) with HW % 32 == 0 -- the reference's own module boundary (models/quantizer.py:45-46
// and :74 are its two permute + copy passes).  A unit is 64 (HW % 64 == 0) or 32 consecutive positions of ONE image.  Per row tile the wave reads 8
// channels x 32 positions per instruction (16 bytes per lane = four positions of one channel, whole 128-byte lines), turns the
// 32 x 64 fp32 block around in its 8 KiB LDS tile (conflict-free both ways: 16-byte chunk c >> 2 of row r sits at slot
// (c >> 2) ^ (r >> 2)) and continues in the row-major register layout; z_q takes the same way back.  Everything between is the
// row-major kernel, bit for bit.
// T (round 4): 32-row tiles per unit.  2 = 64-row units on eight waves per CU (the two tiles share every codebook operand read);
// 1 = 32-row units on SIXTEEN waves per CU (<= 128 registers per lane): four waves per SIMD interleave their latency-bound phases
// (classification, exact part, epilogue, row loads) with each other's sweeps -- what pays when a wave has only one or two units.
// Launch forms <NW, NCHW, T> (launch_vq_track_d64 has the rule): <8, ., 2> many rows; <16, false, 1> up to two 32-row units per wave
// (BASELINE config 3); <8, ., 1> up to 8 units per CU (config 2: every CU busy before any wave gets a second unit) and NCHW maps of
// 32 (2 k + 1) pixels; <4, false, 1> codebooks whose image only fits beside four waves' tiles (K up to 1024: config 4).
// NTILE (round 5): 0 = the number of 32-code tiles is a run-time value; 16 = K in 481 .. 512 (the reference's default codebook) with the sweep
// fully unrolled -- every tile index a compile-time constant, so LDS offsets are immediates, cell ids inline constants, and the
// sweep carries no scalar bookkeeping and no branches (~95 instructions per 32-row unit less)
template <int NW, bool NCHW = false, int T = 2, int NTILE = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void vq_track_kernel_d64(
    const float *__restrict__ z, const float *__restrict__ cb, const uint4 *__restrict__ img_g,
    const float *__restrict__ seeds_g, const float *__restrict__ ee_g, const int *__restrict__ flags,
    long long N, int K, int K32, long long nunits, float *__restrict__ zq, long long *__restrict__ idx,
    int *__restrict__ hist, double *__restrict__ partials, int HW, int pool_pct) {
    constexpr int D = 64, RU = 32 * T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int ntile = NTILE ? NTILE : K32 >> 5;
    uint4 *Eimg = reinterpret_cast<uint4 *>(smem_raw);                                  // [ntile][4][2][32] x 16 B
    float *seeds = reinterpret_cast<float *>(Eimg + (size_t)ntile * 256);               // [ntile][2][16]
    int *hist_s = reinterpret_cast<int *>(seeds + (size_t)ntile * 32);                  // [K]
    double *red = reinterpret_cast<double *>(hist_s + (K + 3) / 4 * 4);                 // [NW]
    int *ticket_s = reinterpret_cast<int *>(red + NW);                                  // next unit of this workgroup (+ pad)
    unsigned char *wave_base = reinterpret_cast<unsigned char *>(red + NW + 2);         // per wave: 8 KiB tile + tables

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NCHW: a 32 x 64 fp32 block is turned around in the tile, one row tile at a time -- in the four-wave form (codebooks whose 128 KiB image
    // leaves 4 KiB per wave: K up to 1024, round 5) in two halves of 32 channels through a 32 x 32 tile
    constexpr bool HALF = NCHW && NW == 4;
    constexpr int TILEB = NCHW ? (HALF ? 4096 : 8192) : 4096 * T, TABB = 1552;
    unsigned char *tile_s = wave_base + (size_t)wave_u * (TILEB + TABB);                // the unit's fp16 rows; later 16 fp32 row slots
    unsigned char *tab_s = tile_s + TILEB;
#ifdef VQ_TRACE
    if ((tid & 63) == 0) for (int i = 0; i < 8; ++i) g_vq_trace[((size_t)blockIdx.x * NW + wave_u) * 8 + i] = 0ull;
    int trace_u = 0;
#endif
    VQ_TR(0);

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

    const vqu::BoundP bound = vqu::load_boundp(flags);

    // ---- row I/O: F[t][i] = floats 4 j16 .. +3 of row 32 t + 4 i + g4 of the unit (1 KiB contiguous per instruction) ----
    // a buffer descriptor over the unit's 16 KiB clipped at the end of z: rows past the end read zeros (their results are
    // never stored), one 32-bit lane offset serves all 16 loads
    auto unit_base = [&](long long p, const float *base) -> const float * {      // first element (channel 0) of unit p
        if constexpr (NCHW) {
            const long long b = (p * RU) / HW;
            return base + (size_t)b * D * HW + (p * RU - b * HW);
        } else {
            return base + (size_t)p * RU * D;
        }
    };
    auto load_unit = [&](long long p, f32x4(&F)[T][8], int lane) {
        if constexpr (NCHW) {
            // F[t][i] = positions 32 t + 4 j8 .. +3 of channel 8 i + cl (lane = cl * 8 + j8); convert() turns the block around
            const long long hw0 = (p * RU) % HW;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(unit_base(p, z)), 0,
                                                              (unsigned)(((long long)D * HW - hw0) * 4), 0x00020000);
            const unsigned vo = (unsigned)((lane >> 3) * HW + 4 * (lane & 7)) * 4u;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    F[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)(8 * i * HW + 32 * t) * 4u, 0));
            return;
        }
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

    // ---- prologue (round 4, second session; profiles/r04b_vq_timeline.txt has the per-wave stamps this order comes from) ----
    // The workgroup meets on the codebook image ALONE.  Image and seeds go to LDS by LDS-DMA (1 KiB pieces, no staging registers;
    // inline assembly: for the builtin hipcc waits vmcnt(0) before every later LDS read) and are requested FIRST; the barrier waits
    // for everything but a wave's own row requests (vmcnt counts in issue order).  Only the first-dispatched wave of every SIMD
    // (waves 0..3) asks for its rows in front of the barrier, the others behind it (the last two a little later still): all waves'
    // first units together are half of z at BASELINE config 3 and arrive interleaved, so with every request in flight at once no
    // wave's unit was complete before nearly every wave's was (barrier at 6.0 us after the first wave's start; 2.6 us this way) --
    // and a SIMD serves its oldest wave first anyway.
    const bool early = (wave_u >> 2) == 0;
    {
        const u32x4 *src16 = reinterpret_cast<const u32x4 *>(img_g);
        u32x4 *dst16 = reinterpret_cast<u32x4 *>(Eimg);
        const int npieces = ntile * 4, lane0 = tid & 63;
        // every workgroup reads the same 64 KiB: each starts at its own piece so the CUs do not queue on the same lines
        const int rot = (int)((blockIdx.x * 97u) % (unsigned)npieces);
        for (int pc = wave_u; pc < npieces; pc += NW) {
            const int sp = (pc + rot) % npieces;
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)(dst16 + sp * 64));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src16 + sp * 64 + lane0), "s"(lds) : "memory");
        }
        // the seeds in whole 1 KiB pieces the same way (a register copy would make hipcc wait for EVERY request of the wave before
        // its LDS write); a seed table that is not whole pieces (K % 256 != 0) copies the rest through registers
        const int nsp = (ntile * 32) / 256;
        for (int pc = wave_u; pc < nsp; pc += NW) {
            const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)(seeds + pc * 256));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(reinterpret_cast<const u32x4 *>(seeds_g) + pc * 64 + lane0), "s"(lds) : "memory");
        }
        if ((ntile * 32) % 256)
            for (int i = nsp * 256 + tid; i < ntile * 32; i += NW * 64) seeds[i] = seeds_g[i];
    }
    if (p < nunits && early) load_unit(p, F, tid & 63);
    for (int k = tid; k < K; k += NW * 64) hist_s[k] = 0;
    if (tid == 0) ticket_s[0] = NW;                          // units 0 .. NW-1 of the workgroup are taken statically

    // ---- fp32 rows -> fp16 B operands through the wave's LDS tile, |z^|^2 (of the unit whose rows are in F) ----
    // tile: row r at 128 r, its 16-byte chunk c at slot c ^ ((r >> 1) & 7) (conflict-free for both access patterns);
    // row 4 i + g4, chunk j16 >> 1: the slot is (j16 >> 1) ^ (g4 >> 1) ^ 2 (i & 3) -- one lane constant, one immediate
    f16x8 zb[T][4];
    float zn2[T];
    // the conversion's two lane-dependent tile offsets, computed ONCE (round 5): ~20 vector instructions per unit less for two live
    // registers -- where registers are not the limit (the sixteen-wave form keeps recomputing them: it would spill)
    constexpr bool HOIST = NW != 16;
    unsigned cvt_wbase = 0u, cvt_rbase = 0u;
    if constexpr (HOIST) {
        const unsigned l = (unsigned)tid & 63u, j16h = l & 15u, g4h = l >> 4, l31h = l & 31u, hh = l >> 5;
        cvt_wbase = g4h * 128u + (((j16h >> 1) ^ (g4h >> 1)) << 4) + ((j16h & 1u) << 3);
        cvt_rbase = l31h * 128u + (((hh ^ (l31h >> 1)) & 7u) << 4);
        asm volatile("" : "+v"(cvt_wbase), "+v"(cvt_rbase));
    }
    auto convert = [&]() {
        int lane_c = lane_id();                                 // (not tid & 63: threadIdx would stay live through the unit loop and spill)
        asm volatile("" : "+v"(lane_c));
        const int l31 = lane_c & 31, h = lane_c >> 5, j16 = lane_c & 15, g4 = lane_c >> 4;
        f32x4 Fh[8];                                   // (HALF only: the turned block is collected here -- F is still being read)
        (void)Fh;
        if constexpr (NCHW) {
            // (32 positions x 64 channels) fp32 through the tile, one row tile at a time: in as [channel][4 positions] per lane,
            // out as F[t][i] = floats 4 j16 .. +3 of row 32 t + 4 i + g4 -- the row-major layout everything below works on
            float *tf = reinterpret_cast<float *>(tile_s);
            const int cl = lane_c >> 3, j8 = lane_c & 7;
            if constexpr (HALF) {
                // channels 32 hh .. 32 hh + 31 at a time: rows of 32 floats, the 16-byte chunk c of row r at slot c ^ (r >> 2) (conflict-free
                // both ways as in the 64-channel tile); the lanes whose output chunk j16 lies in the half read, the others keep their registers
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    lds_order_wave();
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            tf[(4 * j8 + e) * 32 + ((((2 * i + (cl >> 2)) ^ j8) & 7) << 2) + (cl & 3)] = F[0][4 * hh + i][e];
                    lds_order_wave();
                    if ((j16 >> 3) == hh) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            Fh[i] = *reinterpret_cast<const f32x4 *>(tf + (4 * i + g4) * 32 + ((((j16 & 7) ^ i) & 7) << 2));
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) F[0][i] = Fh[i];
            } else
#pragma unroll
            for (int t = 0; t < T; ++t) {
                lds_order_wave();
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        tf[(4 * j8 + e) * 64 + ((((2 * i + (cl >> 2)) ^ j8) & 15) << 2) + (cl & 3)] = F[t][i][e];
                lds_order_wave();
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    F[t][i] = *reinterpret_cast<const f32x4 *>(tf + (4 * i + g4) * 64 + (((j16 ^ i) & 15) << 2));
            }
        }
        const unsigned wbase = HOIST ? cvt_wbase : (unsigned)g4 * 128u + ((((unsigned)j16 >> 1) ^ ((unsigned)g4 >> 1)) << 4) + (((unsigned)j16 & 1u) << 3);
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
        const unsigned rbase = HOIST ? cvt_rbase : (unsigned)l31 * 128u + ((((unsigned)h ^ ((unsigned)l31 >> 1)) & 7u) << 4);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float sq = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(tile_s + t * 4096 + (rbase ^ ((unsigned)(2 * q) << 4)));
                zb[t][q] = __builtin_bit_cast(f16x8, v);
                sq = sqsum8_f16(v.x, v.y, v.z, v.w, sq);       // (not four fdot2 builtins: miscompiled, common.h)
            }
            zn2[t] = sq;
        }
        // both halves of a row need its |z^|^2: (lower half's sum) + (upper half's sum) on every lane
        if constexpr (T == 1) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(zn2[0]), __float_as_uint(zn2[0]), false, false);
            zn2[0] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        } else {
            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(zn2[0]), __float_as_uint(zn2[1]), false, false);
            const float sm = __uint_as_float(sv[0]) + __uint_as_float(sv[1]);      // lower lanes: tile 0's rows, upper lanes: tile 1's
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sm), __float_as_uint(sm), false, false);
            zn2[0] = __uint_as_float(sw[0]);
            zn2[1] = __uint_as_float(sw[1]);
        }
    };
    if (p < nunits && early) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * T) : "memory");    // all but the 8 T row requests
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // histogram, ticket (and a ragged seed table's tail)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    VQ_TR(1);
    if ((wave_u >> 2) == 2) __builtin_amdgcn_s_sleep(20);     // (64-cycle units: ~0.55 us and ~1.1 us)
    if ((wave_u >> 2) == 3) __builtin_amdgcn_s_sleep(40);
    if (p < nunits && !early) load_unit(p, F, tid & 63);
    if (p < nunits) convert();

    const float inf = __builtin_inff();
    VQ_STAMP(0);                                               // codebook image copy + first rows + first conversion
    const float pinf = inf, ninf = -inf;                     // (round 5: the tracker no longer pads with them; plain constants)
    unsigned keymask = trk::kKeyMask;                        // opaque: see vq_track.h
    asm volatile("" : "+v"(keymask));
    // the wave's squared-error sum: one fp32 value per lane and unit; fp64 across units.  The sixteen-wave form has no two registers to
    // spare for it through the sweep: there the fp64 sums sit in the spare 512 bytes of the wave's tables and take the fp32 value of
    // eight units at a time
    typedef typename std::conditional<NW == 16, float, double>::type acc_t;
    acc_t dacc = 0;
    int nflush = 0;
    if constexpr (NW == 16) reinterpret_cast<double *>(tab_s + 1032)[tid & 63] = 0.0;

    // Many units per wave (pool_pct > 0: the host sets 25 from four units per wave on): the last quarter of the units is not dealt out
    // to the workgroups.  A wave whose workgroup has used up its share draws from the pool of its group of workgroups (blockIdx % 8 --
    // the XCD under round-robin dispatch; group x owns the pooled units nlocal + x, + 8, ...), one returning atomic per unit, paid by
    // a wave that would otherwise be idle.  A unit's time depends on its rows (how many stay open): with every unit dealt out the
    // slowest workgroup finishes well after the median one (2.1 M rows: 241 -> 231 us with the pool).  With two units per wave the
    // atomic's round trip costs more than the balance gains (262 144 rows: +1-2 us), and one counter for the whole grid serialises
    // (~9 ns per atomic: 110 us at 262 144 rows when every unit came from it).  The counters live behind the workspace's flags and are
    // zero between launches: the last workgroup of a group to leave puts its two words back.
    const int xng = gridDim.x < 8u ? (int)gridDim.x : 8;
    const int xg = (int)(blockIdx.x % (unsigned)xng);
    int *xt = reinterpret_cast<int *>(reinterpret_cast<char *>(const_cast<int *>(flags)) + 256 + xg * 256);
    int *xdone = xt + 512;                                   // (2 KiB behind the ticket slots)
    long long npool = nunits > pstride ? nunits * pool_pct / 100 : 0;
    if (npool > nunits - pstride) npool = nunits - pstride;  // (first units are always dealt out)
    const long long nlocal = nunits - npool;
    // the wave's next unit: its workgroup's LDS ticket, then (many units per wave) the group's pool
    auto next_unit = [&](int lane) -> long long {
        int q = 0;
        if (lane == 0) q = atomicAdd(ticket_s, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        long long pn = (long long)(q / NW) * pstride + (long long)blockIdx.x * NW + (q % NW);
        if (pn >= nlocal && npool > 0) {                    // the workgroup's own share is used up: the group's pool
            int t = 0;
            if (lane == 0) t = atomicAdd(xt, 1);
            pn = nlocal + (long long)xng * __builtin_amdgcn_readfirstlane(t) + xg;
        }
        return pn;
    };
    while (p < nunits) {
        const long long r0 = p * RU;
        // lane-derived indices are made opaque once per iteration: hipcc otherwise hoists dozens of per-lane address values
        // out of this loop, spills them and reloads them from scratch inside it
        int lane_v = lane_id();
        asm volatile("" : "+v"(lane_v));
        const int lane = lane_v, l31 = lane_v & 31, h = lane_v >> 5;
        const uint4 *ap0 = Eimg + h * 32 + l31;
        const float *sp0 = seeds + h * 16;

        // ================= the sweep: 4 MFMAs per (code tile, row tile), stream / cell maxima per lane ====================
        trk::Lane L[T];
#pragma unroll
        for (int t = 0; t < T; ++t) trk::init(L[t], ninf);
        {
            // (VQ_KO_*: timing-only knock-outs of one resource each -- wrong results; tools/build_vq_variant.sh)
            auto fetch = [&](int ct, u32x4(&a)[4], f32x16 &seed) {
#ifdef VQ_KO_AREAD
                if (ct == 0)
#endif
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const u32x4 *>(ap0 + ct * 256 + q * 64);
#ifdef VQ_KO_AREAD
                asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#endif
#ifdef VQ_KO_SEED
                if (ct == 0)
#endif
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sp0 + ct * 32 + 4 * g);
                    seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                }
#ifdef VQ_KO_SEED
                asm volatile("" : "+v"(seed));
#endif
            };
            // Round 4: the sweep as a software pipeline over code tiles.  The eight MFMAs of tile ct are issued INTERLEAVED with the
            // 48 vector instructions that track tile ct - 1's accumulators (two accumulator sets, one operand set: the same 96
            // registers as the operand ping-pong before), six vector instructions behind every MFMA: a wave issues in order, so
            // vector work placed behind a block of MFMAs waits for the block, and the matrix pipe idles while a block of
            // vector work runs -- a 32-cycle MFMA slot has room for about six vector issues (MI355X_MICROARCH.md, "5 fillers
            // per gap"), which is exactly the tracker's budget (24 per 16 values = 6 per MFMA).
            auto mma = [&](f32x16(&acc)[T], const u32x4(&a)[4], const f32x16 &seed) {
#ifdef VQ_KO_MFMA
                for (int t = 0; t < T; ++t) { acc[t] = seed; acc[t][0] += __uint_as_float(a[0].x ^ a[1].y ^ a[2].z ^ a[3].w); }
                return;
#endif
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), zb[t][0], seed, 0, 0, 0);
#pragma unroll
                for (int q = 1; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[q]), zb[t][q], acc[t], 0, 0, 0);
            };
            auto track = [&](int ct, const f32x16(&acc)[T], bool known = false) {
                unsigned cell0 = (unsigned)(2 * ct), cell1 = cell0 + 1u;        // scalars (opaque: else or3(x & mask, cell0, 1))
                if (!known) asm volatile("" : "+s"(cell0), "+s"(cell1));       // (known: compile-time tile index -> inline constants)
#pragma unroll
#ifdef VQ_KO_TRACK
                for (int t = 0; t < T; ++t) { L[t].S[0] = trk::max3(L[t].S[0], acc[t][0], acc[t][15]); L[t].m1 = trk::max3(L[t].m1, acc[t][7], acc[t][8]); }
                (void)cell0; (void)cell1;
#else
                for (int t = 0; t < T; ++t) trk::tile(L[t], acc[t], cell0, cell1, keymask, ninf, pinf);
#endif
            };
            // one pipeline step: operands of tile ct, its MFMAs into `accn`, the tracker of tile ct - 1 on `accp` between them
            auto step = [&](int ct, f32x16(&accn)[T], const f32x16(&accp)[T], bool known = false) {
                u32x4 a[4];
                f32x16 seed;
                fetch(ct, a, seed);
                __builtin_amdgcn_sched_barrier(0);
                mma(accn, a, seed);
                track(ct - 1, accp, known);
#pragma unroll
                for (int i = 0; i < 4 * T; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);          // six vector instructions
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            f32x16 accA[T], accB[T];
            {
                u32x4 a[4];
                f32x16 seed;
                fetch(0, a, seed);
                mma(accA, a, seed);
            }
            if constexpr (NTILE == 16) {
#pragma unroll
                for (int c2 = 0; c2 < 7; ++c2) {
                    step(1 + 2 * c2, accB, accA, true);
                    step(2 + 2 * c2, accA, accB, true);
                }
                step(15, accB, accA, true);
                track(15, accB, true);
            } else {
            int ct = 1;
            for (; ct + 1 < ntile; ct += 2) {
                step(ct, accB, accA);
                step(ct + 1, accA, accB);
            }
            if (ct < ntile) {                              // an even number of tiles: one more step, the last tile is in accB
                step(ct, accB, accA);
                track(ct, accB);
            } else {
                track(ct - 1, accA);
            }
            }
        }

        VQ_STAMP(2);                                           // sweep
#ifdef VQ_TRACE2
        if (trace_u == 0) VQ_TR(2);
#endif
        // ================= threshold, merge of the two lane halves of every row, verdict (vq_unit.h) =========================
        const vqu::Tables tb = vqu::tables(tab_s);
        const int nleft = (int)(N - r0 < RU ? N - r0 : RU);         // rows of this unit that exist
        vqu::RowsSp<T> R;
        vqu::classify_sp<T>(L, zn2, bound, K, lane, nleft, ninf, tb.task_s, R);

        VQ_STAMP(3);                                           // threshold + verdict
#ifdef VQ_TRACE2
        if (trace_u == 0) VQ_TR(3);
#endif
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
            vqu::Flagged FL = vqu::exact_begin_sp<T>(R, lane, tb);
            int ntasks = FL.ndirect;
            if (FL.hmask && FL.ndirect <= 64) {
                int nres = 0;                                       // tasks of the second screen (wave-uniform)
                // rows with candidates the products do not cover: the tile's screen again, hits (acc >= v1 - DELTA) become tasks
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if ((unsigned)(FL.hmask >> (32 * t))) {
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
                        const float thr_t = (((unsigned)(FL.hmask >> (32 * t)) >> l31) & 1u) ? R.thr[t] : inf;   // only the hard rows can hit
                        for (int ct = 0; ct < ntile && FL.ndirect + nres <= 64; ++ct) {      // (past 64: the hard rows go wide anyway)
                            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), zbr[0], rs, 0, 0, 0);
#pragma unroll
                            for (int q = 1; q < 4; ++q)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[q]), zbr[q], acc, 0, 0, 0);
                            rfetch(ct + 1 < ntile ? ct + 1 : ct);
                            vqu::rescan_tile(acc, thr_t, ct, t, lane, K, FL.ndirect, ninf, tb, nres);
                            asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]));
                        }
                    }
                }
                lds_order_wave();
                ntasks = FL.ndirect + nres;
            }
            if constexpr (NCHW) {
                const float *zu = unit_base(p, z);                 // (strided dword reads: ~3 % of the rows, L2-resident)
                vqu::exact_end_sp<T>(R, FL, ntasks, lane, tb, cb, ee_g, K,
                               [&](int rr, int jc) { const float *q = zu + (size_t)(4 * jc) * HW + rr; return f32x4{q[0], q[HW], q[2 * (size_t)HW], q[3 * (size_t)HW]}; },
                               [&](int rr, int c) { return zu[(size_t)c * HW + rr]; });
            } else {
                const long long left = (N - r0) * (D * 4);
                const auto zr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(z + (size_t)r0 * D), 0,
                                                                     (unsigned)(left < RU * 256 ? left : RU * 256), 0x00020000);
                vqu::exact_end_sp<T>(R, FL, ntasks, lane, tb, cb, ee_g, K,
                               [&](int rr, int jc) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr_rs, (unsigned)rr * 256u + (unsigned)jc * 16u, 0, 0)); },
                               [&](int rr, int c) { const long long grow = r0 + rr; return z[(size_t)(grow < N ? grow : N - 1) * D + c]; });
            }
        }

        VQ_STAMP(4);                                           // exact part
#ifdef VQ_TRACE2
        if (trace_u == 0) VQ_TR(4);
#endif
        // ================= epilogue: gather, z + (e_k - z), squared error, index, histogram (vq_unit.h) =======================
        {
            float sacc;
            if constexpr (NCHW)
                sacc = vqu::epilogue_sp<true, T, HALF>(R, lane, cb, K, [&](int t, int i) { return F[t][i]; },
                                           zq ? const_cast<float *>(unit_base(p, zq)) : nullptr, nleft, idx + r0, hist_s,
                                           reinterpret_cast<float *>(tile_s), HW, (unsigned)(((long long)D * HW - (p * RU) % HW) * 4));
            else
                sacc = vqu::epilogue_sp<false, T>(R, lane, cb, K, [&](int t, int i) { return F[t][i]; },
                                     zq ? zq + (size_t)p * RU * D : nullptr, nleft, idx + r0, hist_s);
            dacc += (acc_t)sacc;
            if constexpr (NW == 16) {
                if ((++nflush & 7) == 0) { reinterpret_cast<double *>(tab_s + 1032)[lane] += (double)dacc; dacc = 0; }
            }
        }
#ifdef VQ_DEBUG_VERDICT
        // debug build (tools/build_variant.py dbg -DVQ_DEBUG_VERDICT): the classification of every row next to its index --
        // bits 20..23 = open / hard / bad / valid, high word = the row's threshold (float bits)
        if (R.valid) {
            const float thr_l = T == 2 && h ? R.thr[T - 1] : R.thr[0], zn2_l = T == 2 && h ? zn2[T - 1] : zn2[0];
            idx[r0 + lane] = (long long)R.kbest | ((long long)(R.open ? 1 : 0) << 20) | ((long long)(R.hard ? 1 : 0) << 21) |
                             ((long long)(R.bad ? 1 : 0) << 22) | ((long long)__float_as_uint(VQ_DEBUG_VERDICT == 2 ? zn2_l : thr_l) << 32);
        }
#endif
        VQ_STAMP(5);                                           // epilogue
#ifdef VQ_TRACE2
        if (trace_u == 0) VQ_TR(5);
        ++trace_u;
#elif defined(VQ_TRACE)
        if (trace_u < 4) { VQ_TR(2 + trace_u); }
        ++trace_u;
#endif
        // (round 5 tried two ways of asking for rows a unit ahead; both lost.  (1) The NEXT unit's rows into F as soon as the current rows
        // are fp16 operands, the epilogue reading the current rows again from L2: 37 -> 44 us at 262 144 rows -- the second read misses
        // L2 and waits as long as the first did.  (2) One 4-byte load per 128-byte line of the unit after the next ("touch"), so that
        // the real loads find their lines in L2: 36.9 -> 38.4 us, 216 -> 259 us at 2.1 M rows -- the lines are requested twice and the
        // request path, not the latency, is what the rows wait on.  profiles/r05_vq_notes.txt)
        p = next_unit(lane);
        if (p < nunits) {
            load_unit(p, F, lane);
            convert();                                         // (waits for the rows; the next iteration starts with the sweep)
        }
        VQ_STAMP(1);                                           // next rows landed, fp16 conversion
    }

#ifdef VQ_SWEEP_TIMING
    VQ_STAMP(6);
    if ((tid & 63) == 0) atomicMax(&tsum[7], (unsigned)(wall_clock64() - tstart));     // slowest wave of the workgroup
    __syncthreads();
    if (tid < 8 && blockIdx.x < 64) reinterpret_cast<unsigned long long *>(partials + 512)[blockIdx.x * 8 + tid] = tsum[tid];
    __syncthreads();
#endif
    VQ_TR(6);
    double dsum = (double)dacc;
    if constexpr (NW == 16) dsum += reinterpret_cast<double *>(tab_s + 1032)[tid & 63];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
    __syncthreads();
    if ((tid & 63) == 0) red[wave_u] = dsum;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[w];
        partials[blockIdx.x] = s;
        if (npool > 0) {
            const int ngroup = ((int)gridDim.x - xg + xng - 1) / xng;   // workgroups of this group
            if (atomicAdd(xdone, 1) == ngroup - 1) { atomicExch(xt, 0); atomicExch(xdone, 0); }
        }
    }
    for (int k = tid; k < K; k += NW * 64) {
        const int c = hist_s[k];
        if (c) atomicAdd(&hist[k], c);
    }
    VQ_TR(7);
}

size_t vq_track_lds_bytes(int K, int nw = 8, int T = 2, bool nchw = false) {      // nw waves per CU, T 32-row tiles per unit (4 KiB of fp16 rows each; NCHW: 8 KiB)
    const int K32 = (K + 31) / 32 * 32;
    return (size_t)K32 * 128 + (size_t)K32 * 4 + (size_t)((K + 3) / 4 * 4) * 4 + (size_t)nw * 8 + 16 +
           (size_t)nw * ((nchw ? (nw == 4 ? 4096 : 8192) : 4096 * T) + 1552);
}

// K <= ~600: the image fits beside eight waves' 64-row tiles (every launch form below).  Up to K = 1024 (BASELINE config 4's codebook:
// a 128 KiB image) it still fits beside FOUR waves' 32-row tiles -- one wave per SIMD, row-major rows only: 2.2x the rate of the
// streamed-codebook kernels (vq_chunk.hip), which such codebooks took before (profiles/r04b_vq_timeline.txt section 5).
static bool vq_track_fits8(int K) { return vq_track_lds_bytes(K, 8, 2) <= (size_t)kLdsBytes; }
bool vq_track_ok(int K, int D) { return D == 64 && K <= 1024 && vq_track_lds_bytes(K, 4, 1) <= (size_t)kLdsBytes; }
// NCHW maps: a unit is 64 consecutive positions of ONE image, or 32 (maps whose pixel count is a multiple of 32 only, and few rows)
bool vq_track_nchw_ok(int K, int D, int HW) {
    // (round 5: codebooks whose image does not fit beside eight waves' tiles -- K up to 1024, BASELINE config 4 -- take the four-wave form
    // with the block turned around in halves)
    return D == 64 && K <= 1024 && (vq_track_fits8(K) || vq_track_lds_bytes(K, 4, 1, true) <= (size_t)kLdsBytes) && HW >= 32 && HW % 32 == 0 &&
           (long long)HW * 256 < 0x7FFFFFF0ll;
}

// The launch form of vq_track_kernel_d64<NW, NCHW, T> for a problem (also behind vqvae_vq_launch_form, include/vqvae_hip.h).
bool vq_track_form(long long N, int K, int HW, bool nchw, int form, int cus, VqTrackForm &f) {
    if (nchw ? !vq_track_nchw_ok(K, 64, HW) : !vq_track_ok(K, 64)) return false;     // (NCHW: a unit = 64 or 32 positions of ONE image)
    // Sixteen waves per CU with 32-row units where a wave gets at most two of them (N <= 2 x 16 x CUs x 32 rows: BASELINE
    // config 3) and the codebook image leaves room for sixteen 4 KiB tiles: the kernel is a chain of latency-bound
    // phases per unit, and four waves per SIMD overlap them four-fold.  With more units per wave the sweep's issue slots
    // dominate and the 64-row form (two tiles share every operand read, half the LDS traffic) wins.  form: 0 = this rule,
    // 8 / 16 = forced (A/B: tools/ubench/vq_ab.cpp)
    const bool fits16 = !nchw && vq_track_lds_bytes(K, 16, 1) <= (size_t)kLdsBytes;
    const bool narrow = !vq_track_fits8(K);                                           // four waves, 32-row units (K up to 1024)
    // (round 5: with the speaker-form unit the sixteen-wave form is also the faster one for many units per wave -- 2.1 M rows
    // 207-235 us against 225-233 us -- so the rule takes it whenever its tiles fit)
    const bool wide = form == 16 ? fits16 : ((form == 8 || form == 12 || form == 32) ? false : fits16);
    // Few rows (N <= 8 x CUs x 32: BASELINE config 2): 32-row units on EIGHT waves per CU -- every CU gets a workgroup before any
    // wave gets a second unit, where sixteen waves would leave half the CUs without one (65 536 rows: 21.5 -> 18 us)
    // (beyond that: 27.8 vs 27.4 us at 131 072 rows, 42.9 vs 40.0 at 262 144).  NCHW maps take the same form then -- and always when
    // their pixel count is a multiple of 32 but not of 64
    const bool spread = (form == 0 && (N + 31) / 32 <= 8LL * cus) || (nchw && HW % 64 != 0) || form == 32;
    const bool twelve = form == 12 && !nchw && !narrow && vq_track_lds_bytes(K, 12, 1) <= (size_t)kLdsBytes;
    f.waves = narrow ? 4 : (twelve ? 12 : ((wide && !spread) ? 16 : 8));
    f.unit_rows = (narrow || wide || spread || twelve) ? 32 : 64;
    f.nunits = (N + f.unit_rows - 1) / f.unit_rows;
    long long grid = (f.nunits + f.waves - 1) / f.waves;
    if (grid > cus) grid = cus;
    if (grid > kVqMaxGrid) grid = kVqMaxGrid;
    f.grid = (int)grid;
    f.pool_pct = f.nunits >= 4 * grid * f.waves ? 25 : 0;     // (see the kernel: the pooled tail pays from four units per wave on)
    return true;
}

int launch_vq_track_d64(const float *z, const float *cb, long long N, int K, float *zq, long long *idx, int *hist,
                        char *ws, hipStream_t st, int *grid_out, int HW, bool nchw, int form) {
    const VqPlan p = vq_plan(K, 64);
    VqTrackForm tf;
    if (!vq_track_form(N, K, HW, nchw, form, num_cus(), tf)) return VQVAE_ERR_UNSUPPORTED;
    const int NW = tf.waves, RU = tf.unit_rows, pool_pct = tf.pool_pct;
    const long long nunits = tf.nunits, grid = tf.grid;
    const bool narrow = NW == 4, spread = NW == 8 && RU == 32, wide = NW == 16;
    *grid_out = tf.grid;
    auto launch = [&](auto kfn) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        hipEvent_t e0, e1;
        const uint4 *imgh = reinterpret_cast<const uint4 *>(ws + p.off_imgh);
        const float *seeds = reinterpret_cast<const float *>(ws + p.off_seeds), *ee = reinterpret_cast<const float *>(ws + p.off_ee);
        const int *wflags = reinterpret_cast<const int *>(ws + p.off_flags);
        double *partials = reinterpret_cast<double *>(ws + p.off_partials);
        // the extended launch only while profiling (it carries the dispatch's start / stop events); the plain one otherwise --
        // that is the form a stream capture (vqvae_amd/graph.py) records
        if (prof_dispatch(VQVAE_PROF_VQ_MAIN, &e0, &e1))
            hipExtLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NW * 64), vq_track_lds_bytes(K, NW, RU / 32, nchw), st, e0, e1, 0, z, cb, imgh, seeds, ee,
                                  wflags, N, K, p.K32, nunits, zq, idx, hist, partials, HW, pool_pct);
        else
            hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NW * 64), vq_track_lds_bytes(K, NW, RU / 32, nchw), st, z, cb, imgh, seeds, ee, wflags, N, K,
                               p.K32, nunits, zq, idx, hist, partials, HW, pool_pct);
    };
    const bool k512 = p.K32 == 512;                          // sixteen code tiles: the instances with the unrolled sweep
    if (narrow && nchw) launch(vq_track_kernel_d64<4, true, 1>);
    else if (narrow) launch(vq_track_kernel_d64<4, false, 1>);
    else if (NW == 12) launch(vq_track_kernel_d64<12, false, 1>);
    else if (nchw && spread && k512) launch(vq_track_kernel_d64<8, true, 1, 16>);
    else if (nchw && spread) launch(vq_track_kernel_d64<8, true, 1>);
    else if (nchw && k512) launch(vq_track_kernel_d64<8, true, 2, 16>);
    else if (nchw) launch(vq_track_kernel_d64<8, true, 2>);
    else if (spread && k512) launch(vq_track_kernel_d64<8, false, 1, 16>);
    else if (spread) launch(vq_track_kernel_d64<8, false, 1>);
    else if (wide && k512) launch(vq_track_kernel_d64<16, false, 1, 16>);
    else if (wide) launch(vq_track_kernel_d64<16, false, 1>);
    else launch(vq_track_kernel_d64<8, false, 2>);
    return (int)hipGetLastError();
}

}  // namespace vqvae

#ifdef VQ_TRACE
extern "C" VQVAE_API int vqvae_debug_vq_trace(void *host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vqvae::g_vq_trace), bytes < sizeof(vqvae::g_vq_trace) ? bytes : sizeof(vqvae::g_vq_trace));
}
#endif
