// What a wave does with ONE unit of the stream-tracker quantizer (64 rows = two 32-row tiles) once the sweep has filled its
// trackers -- shared by the standalone kernel (vq_track.hip: codebook image resident in LDS, rows in registers in the load
// layout) and by the encoder's last kernel (conv_fused.hip, conv_res_pair8_h2_kernel<2, true>: z_e straight from the 1x1 conv's
// accumulators, codebook image streamed through the weight stages, rows parked in LDS).  The pieces:
//     classify     threshold per row, merge of the two lane halves, verdict; open rows' exact tasks into the task table
//     exact_begin  flagged rows (open / hard / non-finite), their table entries; hard rows are then screened again by the
//                  CALLER (it owns the codebook image) and appended as tasks
//     exact_end    the exact chains (four tasks per pass, one per 16-lane group), decision by 64-bit LDS minimum, the scalar
//                  torch.argmin path for non-finite rows / overflow
//     epilogue     codebook gather, z + (e_k - z), squared error, z_q stores, index, histogram
// Row data come through functors, so the two callers keep their own layouts.  models/quantizer.py:45-74; the bound is
// derived in vq_track.hip's header.
#pragma once
#include "common.h"
#include "vq_track.h"

#ifndef VQ_ZQ_STORE_AUX
#define VQ_ZQ_STORE_AUX 0          // cache policy of the z_q stores (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1); see vq_track.hip
#endif

namespace vqvae {
namespace vqu {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_order_wave() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

struct Bound {                       // what DELTA needs of the codebook (vq_prepare_kernel / vq_prepare16_kernel -> flags[])
    int cb_bad;
    float A, Ehat, dE, EmaxS, EEh, EEa;
};

__device__ __forceinline__ Bound load_bound(const int *__restrict__ flags) {
    Bound b;
    b.cb_bad = flags[0];
    b.A = __builtin_ldexpf(1.0f, flags[5]);
    const float EEmax = __int_as_float(flags[1]) * 1.0001f;               // max ee_k (unscaled)
    b.Ehat = __builtin_sqrtf(__int_as_float(flags[3])) * 1.0001f;
    b.dE = __builtin_sqrtf(__int_as_float(flags[4])) * 1.0001f;
    b.EmaxS = __builtin_sqrtf(EEmax) * b.A * 1.0001f;
    b.EEh = 0.5f * EEmax * b.A;
    b.EEa = EEmax * b.A;
    return b;
}

// per-wave tables (1552 bytes): [64] tasks  row | a << 6 | b << 19;  [64] (distance, index) minima;  [64] ||z||^2;  counter;
// from byte 1032: 64 fp64 loss sums (vq_track_kernel_d64's sixteen-wave form)
constexpr int kTabBytes = 1552;
struct Tables {
    unsigned *task_s;
    unsigned long long *best_s;
    float *zz_s;
    int *cnt_s;
};
__device__ __forceinline__ Tables tables(unsigned char *tab_s) {
    return Tables{reinterpret_cast<unsigned *>(tab_s), reinterpret_cast<unsigned long long *>(tab_s + 256),
                  reinterpret_cast<float *>(tab_s + 768), reinterpret_cast<int *>(tab_s + 1024)};
}

// ---- exact part, first half: who is flagged; table entries of the non-finite rows -------------------------------------
// lane L of the wave speaks for row L of the unit (tile L >> 5, row L & 31)
struct Flagged {
    bool o_open, o_hard, o_bad;
    unsigned long long fm, hmask;    // flagged rows / rows to be screened again (wave-uniform)
    int ndirect;                     // tasks so far: classification + one per non-finite row
};

// one 32-code tile of a row tile screened again: every code at or above the row's threshold becomes a task.  nres (wave-uniform):
// tasks the second screen has produced so far -- slots by ballot prefix (round 6: it was one LDS atomic per hit, and a row at a
// trained checkpoint's dead-code cluster has 450 hits); once ndirect + nres passes the table's 64 entries the caller stops
// screening: the unit's hard rows then take the wave-wide argmin of exact_end_sp() whatever else turns up.
__device__ __forceinline__ void rescan_tile(const f32x16 &acc, float thr_t, int ct, int t, int lane, int K, int ndirect, float ninf,
                                            const Tables &tb, int &nres) {
    const int l31 = lane & 31, h = lane >> 5;
    const float x0 = trk::max3(trk::max3(acc[0], acc[1], acc[2]), trk::max3(acc[3], acc[4], acc[5]), trk::max3(acc[6], acc[7], acc[8]));
    const float x1 = trk::max3(trk::max3(acc[9], acc[10], acc[11]), trk::max3(acc[12], acc[13], acc[14]), acc[15]);
    const float mx = trk::max3(x0, x1, ninf);
    if (__builtin_amdgcn_ballot_w64(mx >= thr_t)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool hit = acc[r] >= thr_t && code < K;
            const unsigned long long b = __builtin_amdgcn_ballot_w64(hit);
            if (b) {
                const int sl = ndirect + nres + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
                if (hit && sl < 64) tb.task_s[sl] = (unsigned)(32 * t + l31) | ((unsigned)code << 6) | ((unsigned)code << 19);
                nres += __builtin_popcountll(b);
            }
        }
    }
}

// =====================================================================================================================
// Round 5: the same unit in SPEAKER form (vq_track.h, "round 5"): lane L of the wave speaks for row L of the unit -- the
// verdict, the index, the histogram count and the gather address of a row live on ONE lane, so nothing is selected per
// half and nothing is shuffled back.  Same decisions and the same bits out as the functions above (which the encoder's
// fused kernel still uses); per 32-row tile the classification issues ~60 vector instructions instead of ~250, the
// epilogue ~115 instead of ~220 (profiles/r05_vq_notes.txt has the census).
// =====================================================================================================================

// DELTA of classify() as a polynomial in Z = v_sqrt_f32(|z^|^2), every coefficient rounded up:
//   zh <= s Z (s = 1.00016 covers sqrt(1.0001) * 1.0001 and v_sqrt's ulp), errz = c1 zh + c0, zn = zh + errz = q Z + c0 (q = (1 + c1) s)
//   DELTA = 2.002 (eps + xi) = (D2 Z + D1) Z + D0,   key slack tB = 8e-6 (zn Ehat + EEh) = T1 Z + T0
struct BoundP {            // wave-uniform, kept in scalar registers (gfx950 reads ONE scalar register per vector instruction:
    int cb_bad;            // classify_sp() spends a v_mul + v_add + v_fma on DELTA rather than two live vector registers)
    float D2, D1, D0, T1, T0;
};

__device__ __forceinline__ BoundP load_boundp(const int *__restrict__ flags) {
    const Bound b = load_bound(flags);
    const float c1 = 4.89e-4f, c0 = 2.5e-7f, s = 1.00016f, q = (1.0f + c1) * s, up = 1.0002f;
    BoundP p;
    p.cb_bad = b.cb_bad;
    p.D2 = up * 2.002f * (1.2e-7f * b.A * q * q);
    p.D1 = up * 2.002f * (b.Ehat * (c1 * s) + b.dE * ((1.0f + 2.0f * c1) * s) + 7.76e-6f * b.Ehat * q + 3.86e-6f * b.EmaxS * q + 1.2e-7f * b.A * (2.0f * q * c0));
    p.D0 = up * 2.002f * (b.Ehat * c0 + 2.0f * b.dE * c0 + 7.76e-6f * (b.Ehat * c0 + b.EEh) + 3.86e-6f * b.EmaxS * c0 + 1.2e-7f * (b.A * c0 * c0 + b.EEa));
    p.T1 = 1.01f * 8.0e-6f * b.Ehat * q;
    p.T0 = 1.01f * 8.0e-6f * (b.Ehat * c0 + b.EEh);
    p.cb_bad = __builtin_amdgcn_readfirstlane(p.cb_bad);
    // (inline assembly: hipcc folds the builtin away on values it knows to be uniform and then cannot place them in scalar registers)
    int d2, d1, d0, t1, t0;
    asm volatile("v_readfirstlane_b32 %0, %5\n\tv_readfirstlane_b32 %1, %6\n\tv_readfirstlane_b32 %2, %7\n\t"
                 "v_readfirstlane_b32 %3, %8\n\tv_readfirstlane_b32 %4, %9"
                 : "=s"(d2), "=s"(d1), "=s"(d0), "=s"(t1), "=s"(t0) : "v"(p.D2), "v"(p.D1), "v"(p.D0), "v"(p.T1), "v"(p.T0));
    p.D2 = __int_as_float(d2); p.D1 = __int_as_float(d1); p.D0 = __int_as_float(d0); p.T1 = __int_as_float(t1); p.T0 = __int_as_float(t0);
    return p;
}

template <int T>
struct RowsSp {
    int kbest;                           // speaker lane L: the index of row L of the unit (closed rows; the exact part fills in the rest)
    bool valid, bad, open, hard;         // speaker lane L: row L
    unsigned long long openm, hardm;     // the same as wave masks (bit L = row L)
    float thr[T];                        // per accumulator lane: the threshold of row l31 of tile t (the rescan needs it)
    int ncls;                            // tasks the classification wrote (wave-uniform)
};

// nleft: rows of the unit that exist (wave-uniform)
template <int T>
__device__ __forceinline__ void classify_sp(const trk::Lane (&L)[T], const float (&zn2)[T], const BoundP &B, int K, int lane, int nleft,
                                            float ninf, unsigned *task_s, RowsSp<T> &R) {
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned gemask = trk::kGeBits | (((unsigned)lane & 32u) << 9);      // | h << 14: see trk::word_of
    // ---- the row maxima on every lane ----
    float v1[T];
    if constexpr (T == 1) {
        const float a = trk::lane_max(L[0], ninf);
        const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        v1[0] = trk::max2(__uint_as_float(sv[0]), __uint_as_float(sv[1]));
    } else {
        const float a0 = trk::lane_max(L[0], ninf), a1 = trk::lane_max(L[1], ninf);
        const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
        const float m = trk::max2(__uint_as_float(sv[0]), __uint_as_float(sv[1]));          // lower lanes: tile 0's rows, upper: tile 1's
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        v1[0] = __uint_as_float(sw[0]);
        v1[1] = __uint_as_float(sw[1]);
    }
    // ---- thresholds, words ----
    unsigned ge[T], w[T];
    bool badl[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float Z = __builtin_amdgcn_sqrtf(zn2[t]);
        const float delta = __builtin_fmaf(Z * B.D2 + B.D1, Z, B.D0);
        const float th = v1[t] - delta;
        const float thB = __builtin_fmaf(Z, -B.T1, th - B.T0);
        R.thr[t] = th;
        // NaN / Inf anywhere in the row, an overflowing bound, or |v1| below 1e-30 (a key could be a denormal whose cell field a
        // flush would lose -- never on real data): the scalar path
        const float av = __builtin_fabsf(v1[t]);
        badl[t] = !(av < 1.0e37f) || !(delta < 1.0e37f) || av < 1.0e-30f;
        ge[t] = ~trk::lt_of(L[t], th, thB, ninf) & gemask;
        w[t] = trk::word_of(L[t], ge[t]);
    }
    // ---- the speakers get their partners' words: lanes 0..31 speak for tile 0, lanes 32..63 for tile 1 ----
    unsigned Wlo, Whi;
    bool bad_sp;                          // (both halves of a row computed the same v1 and delta)
    if constexpr (T == 1) {
        unsigned junk;
        asm volatile("" : "=v"(junk));
        const auto sv = __builtin_amdgcn_permlane32_swap(w[0], junk, false, false);
        Wlo = sv[0]; Whi = sv[1];
        bad_sp = badl[0];
    } else {
        const auto sv = __builtin_amdgcn_permlane32_swap(w[0], w[1], false, false);
        Wlo = sv[0]; Whi = sv[1];
        bad_sp = h ? badl[T - 1] : badl[0];
    }
    const trk::Spoken V = trk::spoken_of(Wlo, Whi);
    const bool closed = trk::spoken_closed(V, K);
    R.kbest = V.kbest;
    R.valid = lane < nleft;
    R.bad = R.valid && (bad_sp || B.cb_bad);
    R.open = false; R.hard = false; R.openm = 0ull; R.hardm = 0ull; R.ncls = 0;
    const bool nonclosed = R.valid && !R.bad && !closed;
    if (__builtin_amdgcn_ballot_w64(nonclosed)) {
        R.hard = nonclosed && trk::spoken_hard(V, K);
        R.open = nonclosed && !R.hard;
        R.openm = __builtin_amdgcn_ballot_w64(R.open);
        R.hardm = __builtin_amdgcn_ballot_w64(R.hard);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const unsigned om = (unsigned)(R.openm >> (32 * t));
            if (om) {
                // open rows: this half's exact tasks
                const trk::Cands C = trk::cands2_of(L[t], ge[t], h, K);
                const int nt = ((om >> l31) & 1u) ? C.ntask : 0;
                const unsigned long long b1 = __builtin_amdgcn_ballot_w64(nt >= 1), b2 = __builtin_amdgcn_ballot_w64(nt >= 2);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0)) +
                                  __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
                const int slot = R.ncls + below;
                const unsigned rowu = (unsigned)(32 * t + l31);
                if (nt >= 1 && slot < 64) task_s[slot] = rowu | ((unsigned)C.ta[0] << 6) | ((unsigned)C.tb[0] << 19);
                if (nt >= 2 && slot + 1 < 64) task_s[slot + 1] = rowu | ((unsigned)C.ta[1] << 6) | ((unsigned)C.tb[1] << 19);
                R.ncls += __builtin_popcountll(b1) + __builtin_popcountll(b2);
            }
        }
    }
}

template <int T>
__device__ __forceinline__ Flagged exact_begin_sp(const RowsSp<T> &R, int lane, const Tables &tb) {
    Flagged F;
    F.o_open = R.open; F.o_hard = R.hard; F.o_bad = R.bad;
    F.fm = __builtin_amdgcn_ballot_w64(R.open || R.hard || R.bad);
    F.hmask = 0ull;
    F.ndirect = R.ncls;
    if (F.fm) {
        const unsigned long long lowmask = (1ull << lane) - 1ull;
        unsigned ones;                       // (made here: hipcc hoists a plain ~0ull pair out of the unit loop and spills it)
        asm volatile("v_mov_b32 %0, -1" : "=v"(ones));
        reinterpret_cast<unsigned *>(tb.best_s)[2 * lane] = ones;
        reinterpret_cast<unsigned *>(tb.best_s)[2 * lane + 1] = ones;
        // non-finite rows: one task each, for the row's ||z||^2
        const unsigned long long tmb = __builtin_amdgcn_ballot_w64(F.o_bad);
        if (F.o_bad && R.ncls + __builtin_popcountll(tmb & lowmask) < 64) tb.task_s[R.ncls + __builtin_popcountll(tmb & lowmask)] = (unsigned)lane;
        F.ndirect = R.ncls + __builtin_popcountll(tmb);
        F.hmask = R.hardm;
        if (F.hmask && lane == 0) tb.cnt_s[0] = 0;
        lds_order_wave();
    }
    return F;
}

// the chains and the decision of exact_end(); R.kbest of the flagged rows final afterwards
template <int T, class ZRow, class ZScalar>
__device__ __forceinline__ void exact_end_sp(RowsSp<T> &R, Flagged &F, int ntasks, int lane, const Tables &tb, const float *__restrict__ cb,
                                             const float *__restrict__ ee_g, int K, ZRow &&zrow, ZScalar &&zscalar) {
    constexpr int D = 64;
    if (!F.fm) return;
    const int j8 = lane & 7, g8 = lane >> 3;
    // Rows that take torch.argmin over ALL codes with the whole wave (below): non-finite rows -- and, when a unit's candidates overflow
    // the 64-entry task table, its HARD rows (the rows whose second screen found the many candidates); the open rows keep the tasks
    // the classification wrote.  The tasks only produce a wide row's ||z||^2.
    bool o_wide = F.o_bad;
    if (ntasks > 64) {
        const unsigned long long lowmask = (1ull << lane) - 1ull;
        const int nh = __builtin_popcountll(F.hmask);
        __builtin_amdgcn_wave_barrier();
        if (F.hmask && F.ndirect + nh <= 64) {
            if (F.o_hard) tb.task_s[F.ndirect + __builtin_popcountll(F.hmask & lowmask)] = (unsigned)lane;   // (over the second screen's tasks)
            o_wide = o_wide || F.o_hard;
            ntasks = F.ndirect + nh;
        } else {                            // more tasks than entries from the open rows alone: every flagged row goes wide
            o_wide = o_wide || F.o_open || F.o_hard;
            if (o_wide) tb.task_s[__builtin_popcountll(F.fm & lowmask)] = (unsigned)lane;
            ntasks = __builtin_popcountll(F.fm);
        }
    }
    lds_order_wave();
    const auto cb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cb), 0, (unsigned)K * (D * 4), 0x00020000);
    // Four tasks per pass as before, but ONE code per group of EIGHT lanes (group g: task g >> 1, code a or b), lane j of a group
    // holding channels 8 j .. 8 j + 7 of the row and of the code: the c-ordered fmaf chain passes through the group in 8 steps of
    // 8 fmaf instead of through 16 lanes in 16 steps of 2 x 4 -- the same 64 sequential fmaf per (row, code), half the instructions
    // per pass (72 against 160: what counts is the step count, every lane issues every step).
    for (int base = 0; base < ntasks; base += 4) {
        const int jj = base + (g8 >> 1);
        const unsigned task = tb.task_s[jj < ntasks ? jj : 0];
        const int rr = (int)(task & 63u), ka = (int)((task >> 6) & 8191u), kb2 = (int)(task >> 19);
        const int kc = (g8 & 1) ? kb2 : ka;
        const f32x4 z0 = zrow(rr, 2 * j8), z1 = zrow(rr, 2 * j8 + 1);
        const f32x4 e0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cb_rs, (unsigned)kc * 256u + (unsigned)j8 * 32u, 0, 0));
        const f32x4 e1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cb_rs, (unsigned)kc * 256u + (unsigned)j8 * 32u + 16u, 0, 0));
        const float eek = ee_g[kc];
        // ||z||^2 in ATen's order (vq_device.h, aten_sqsum_full<64>): lane j holds vector j (elements 8 j + t): part[q][t] = v_q[t] +
        // v_{q+4}[t] (lane q + lane q + 4), a_t = ((part_0 + part_1) + part_2) + part_3 (lanes 0..3 of the group), then a_0..a_7 in order
        const float sq[8] = {z0.x * z0.x, z0.y * z0.y, z0.z * z0.z, z0.w * z0.w, z1.x * z1.x, z1.y * z1.y, z1.z * z1.z, z1.w * z1.w};
        // (the eight t-chains level by level, not one after the other: a DPP read needs two instructions between it and the write
        // of its source -- back to back, hipcc pads every level of every chain with an s_nop)
        float P[8], A[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) P[t] = sq[t] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sq[t]), 0x104, 0xf, 0xf, true));
#pragma unroll
        for (int t = 0; t < 8; ++t) A[t] = P[t] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P[t]), 0x101, 0xf, 0xf, true));
#pragma unroll
        for (int t = 0; t < 8; ++t) A[t] = A[t] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P[t]), 0x102, 0xf, 0xf, true));
#pragma unroll
        for (int t = 0; t < 8; ++t) A[t] = A[t] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P[t]), 0x103, 0xf, 0xf, true));
        float zz = 0.0f;
#pragma unroll
        for (int t = 0; t < 8; ++t) zz = zz + A[t];                       // valid on lane 0 of the group
        // the chain: lane j continues lane j - 1's partial sum (row_shr:1; every partial sum is 0 before the first step, and lane s's
        // value after step s only depends on lanes below it in its own group)
        float m = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < 8; ++sidx) {
            const float im = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x111, 0xf, 0xf, true));
            m = __builtin_fmaf(z1.w, e1.w, __builtin_fmaf(z1.z, e1.z, __builtin_fmaf(z1.y, e1.y, __builtin_fmaf(z1.x, e1.x,
                __builtin_fmaf(z0.w, e0.w, __builtin_fmaf(z0.z, e0.z, __builtin_fmaf(z0.y, e0.y, __builtin_fmaf(z0.x, e0.x, im))))))));
        }
        const float zz7 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zz), 0x117, 0xf, 0xf, true));   // lane 7 <- lane 0
        const float d = (zz7 + eek) - 2.0f * m;                                       // valid on lane 7 of the group
        if (j8 == 7 && jj < ntasks) {
            atomicMin(&tb.best_s[rr], trk::dist_key(d, kc));
            tb.zz_s[rr] = zz7;
        }
    }
    lds_order_wave();
    int o_best = 0;
    if ((F.o_open || F.o_hard) && !o_wide) {
        const unsigned long long bk = tb.best_s[lane];
        if (bk != ~0ull) o_best = (int)(unsigned)bk; else o_wide = true;   // no task came back (cannot happen): the wide path
    }
    // torch.argmin over ALL codes (NaN is minimal, first index wins) for the rows the tasks cannot decide.  The overflow kind is what
    // a TRAINED checkpoint produces -- its hundreds of dead codes still sit within 1/K of the origin, nearly one point at the scale of
    // |z| ~ 10, so a row near them finds hundreds of codes above its threshold.  Round 6: the WAVE takes such a row -- lane l the codes
    // l, l + 64, ... four at a time, each the same c-ordered fmaf chain on the row's fp32 data (broadcast from the 256 bytes of the task
    // table, which nobody reads any more), (distance, index) keys folded by the 64-bit LDS minimum -- instead of one lane running K x D
    // serial fmaf: 33 such rows among 262 144 made the kernel 14 times slower on the trained checkpoints (profiles/r06_vq_trained.txt).
    unsigned long long wm = __builtin_amdgcn_ballot_w64(o_wide);
    if (wm) {
        float *zrow_s = reinterpret_cast<float *>(tb.task_s);
        do {
            const int r = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(wm));
            wm &= wm - 1ull;
            __builtin_amdgcn_wave_barrier();
            zrow_s[lane] = zscalar(r, lane);
            if (lane == 0) tb.best_s[r] = ~0ull;
            lds_order_wave();
            const float zz = tb.zz_s[r];                                  // every wide row had a task
            int best = 0;
            if (zz == zz) {                                               // NaN ||z||^2: every distance is NaN -> index 0
                unsigned long long key = ~0ull;
                for (int kb = lane; kb < K; kb += 256) {                  // codes kb, kb + 64, kb + 128, kb + 192 (past K: zeros, dropped)
                    float m[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
                    for (int c4 = 0; c4 < D / 4; ++c4) {
                        const f32x4 zv = *reinterpret_cast<const f32x4 *>(zrow_s + 4 * c4);
                        f32x4 e[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            e[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cb_rs, (unsigned)(kb + 64 * i) * (D * 4) + (unsigned)c4 * 16u, 0, 0));
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            m[i] = __builtin_fmaf(zv.w, e[i].w, __builtin_fmaf(zv.z, e[i].z, __builtin_fmaf(zv.y, e[i].y, __builtin_fmaf(zv.x, e[i].x, m[i]))));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = kb + 64 * i;
                        if (k < K) {
                            const float d = (zz + ee_g[k]) - 2.0f * m[i];
                            const unsigned long long kk = d != d ? (unsigned long long)(unsigned)k : trk::dist_key(d, k);   // NaN below every distance
                            key = kk < key ? kk : key;
                        }
                    }
                }
                atomicMin(&tb.best_s[r], key);
                lds_order_wave();
                best = (int)(unsigned)tb.best_s[r];
            }
            if (lane == r) o_best = best;
        } while (wm);
    }
    if (R.open || R.hard || R.bad) R.kbest = o_best;
    __builtin_amdgcn_wave_barrier();
}

// epilogue() with the rows' indices on their speaker lanes: frow / zq_unit / nleft / idx_unit / hist_s / NCHW arguments as there
// HALF (NCHW, four-wave form): `tile_f` holds 32 x 32 floats only -- z_q leaves in two halves of 32 channels
template <bool NCHW = false, int T = 2, bool HALF = false, class FRow>
__device__ __forceinline__ float epilogue_sp(const RowsSp<T> &R, int lane, const float *__restrict__ cb, int K, FRow &&frow,
                                             float *__restrict__ zq_unit, int nleft, long long *__restrict__ idx_unit,
                                             int *__restrict__ hist_s, float *tile_f = nullptr, int HW = 0, unsigned zq_bytes = 0u) {
    constexpr int D = 64, RU = 32 * T;
    const int j16 = lane & 15, g4 = lane >> 4;
    const auto cb_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cb), 0, (unsigned)K * (D * 4), 0x00020000);
    f32x4 ev[T][8];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kr = __builtin_amdgcn_ds_bpermute((32 * t + 4 * i + g4) << 2, R.kbest);
            ev[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cb_rs, (unsigned)kr * (D * 4) + (unsigned)j16 * 16u, 0, 0));
        }
    // index and histogram count of row L from lane L while the gathers are on their way
    if (R.valid) {
        idx_unit[lane] = R.kbest;
        atomicAdd(&hist_s[R.kbest], 1);
    }
    const auto zq_rs = __builtin_amdgcn_make_buffer_rsrc(zq_unit ? zq_unit : const_cast<float *>(cb), 0,
                                                         zq_unit ? (NCHW ? zq_bytes : (unsigned)nleft * (D * 4)) : 0u, 0x00020000);
    // (store offsets without a scalar offset register: the hazard note in epilogue())
    unsigned vo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vo[k] = (unsigned)lane * 16u + 4096u * k;
        asm volatile("" : "+v"(vo[k]));
    }
    float sqv[T][8];
    f32x4 oh[8];
    (void)oh;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 zv = frow(t, i), e = ev[t][i];
            const float d0 = e.x - zv.x, d1 = e.y - zv.y, d2 = e.z - zv.z, d3 = e.w - zv.w;
            f32x4 o;
            o.x = zv.x + d0; o.y = zv.y + d1; o.z = zv.z + d2; o.w = zv.w + d3;
            sqv[t][i] = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
            if constexpr (!NCHW) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), zq_rs, vo[(t * 8 + i) >> 2] + (unsigned)((t * 8 + i) & 3) * 1024u, 0, VQ_ZQ_STORE_AUX);
            } else if constexpr (HALF) {
                oh[i] = o;                                  // (collected: the tile takes them in two halves below)
            } else {
                if (i == 0) lds_order_wave();
                *reinterpret_cast<f32x4 *>(tile_f + (4 * i + g4) * 64 + (((j16 ^ i) & 15) << 2)) = o;
                if (i == 7) {
                    lds_order_wave();
                    const int cl = lane >> 3, j8 = lane & 7;
                    unsigned so = (unsigned)(cl * HW + 4 * j8 + 32 * t) * 4u;
#pragma unroll
                    for (int c8 = 0; c8 < 8; ++c8) {
                        f32x4 wv;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) wv[e2] = tile_f[(4 * j8 + e2) * 64 + ((((2 * c8 + (cl >> 2)) ^ j8) & 15) << 2) + (cl & 3)];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, wv), zq_rs, so, 0, VQ_ZQ_STORE_AUX);
                        so += (unsigned)(8 * HW) * 4u;
                    }
                }
            }
        }
    if constexpr (NCHW && HALF) {
        // rows in (the lanes whose 16-byte chunk j16 lies in the half), [channel][four positions] out: 32-float rows, chunk c of row r
        // at slot c ^ (r >> 2) as in the kernel's convert()
        static_assert(T == 1, "the half-tile form has one row tile per unit");
        const int cl = lane >> 3, j8 = lane & 7;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            lds_order_wave();
            if ((j16 >> 3) == hh) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4 *>(tile_f + (4 * i + g4) * 32 + ((((j16 & 7) ^ i) & 7) << 2)) = oh[i];
            }
            lds_order_wave();
            unsigned so = (unsigned)((32 * hh + cl) * HW + 4 * j8) * 4u;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                f32x4 wv;
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) wv[e2] = tile_f[(4 * j8 + e2) * 32 + ((((2 * c8 + (cl >> 2)) ^ j8) & 7) << 2) + (cl & 3)];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, wv), zq_rs, so, 0, VQ_ZQ_STORE_AUX);
                so += (unsigned)(8 * HW) * 4u;
            }
        }
    }
    // fp32 over the unit's 16 groups in the order of epilogue() (one fp64 add per unit in the caller); only a ragged last unit masks
    float sacc = 0.0f;
    if (nleft == RU) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc += sqv[t][i];
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) sacc += 32 * t + 4 * i + g4 < nleft ? sqv[t][i] : 0.0f;
    }
    return sacc;
}

}  // namespace vqu
}  // namespace vqvae
