// Per-lane logic of the stream tracker of vq_track.hip (round 3), written so that the SAME source compiles for gfx950
// (hipcc, device) and for the host (g++: tests/test_vq_track_host.py builds a harness that emulates a wave lane by lane
// and checks the tracker's verdicts against a brute-force scan of the same accumulator matrix).
//
// What is tracked.  The screen of models/quantizer.py:49-54 is acc[k] = z^ . e^_k - A ||e_k||^2 / 2 for the K codes of a
// row; an accumulator lane (row n, half h) sees 16 of every 32-code tile's values, acc[r] <-> code 32 T + (r & 3) +
// 8 (r >> 2) + 4 h.  Round 2 kept the three largest values of a lane as index-carrying keys (5 vector instructions per
// value).  Here a lane keeps maxima over TWO partitions of its values instead:
//     streams   S[a]      = max over all tiles of max(acc[a], acc[a + 8]),  a = 0..7          (position in the tile)
//     cells     (T, s)    = max of acc[8 s .. 8 s + 7] of tile T,  as the top three KEYS       (which tile, which half-tile)
// Every value of the lane lies in exactly one stream and one cell, and a (stream, cell) pair holds exactly ONE value.
// Hence, for a threshold thr = v1 - DELTA below the row's largest value v1:
//   * every code with acc >= thr lies in a stream whose maximum is >= thr AND in a cell whose maximum is >= thr;
//   * if exactly one stream and exactly one cell of the row (both halves together) reach thr, exactly one code does, and
//     its index is (stream, cell) -- no index bits were carried through the sweep;
//   * otherwise the candidates are the (stream, cell) products of each half: a superset of the codes at or above thr,
//     each evaluated EXACTLY afterwards (phantom products merely lose).
// Cost in the sweep: 8 v_max3 for the streams + 8 v_max3 for the two cell maxima + 2 x (and_or, med3, med3, med3) for the
// keys = 24 vector instructions per 16 values (1.5 per value instead of 5), and DELTA loses the 10-bit truncation term
// (only the cell keys are truncated, by 6 bits, and only the test of a key against the threshold pays for it).
#pragma once

#include <cmath>
#include <cstring>
#if defined(__HIPCC__)
#define VQT_FN __host__ __device__ __forceinline__      // (both passes of hipcc see the same declarations)
#else
#define VQT_FN inline
#endif

namespace vqvae {
namespace trk {

VQT_FN unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
VQT_FN float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// three-operand forms only on the device: hipcc lowers a two-operand fmaxf to v_max_f32 plus a canonicalising v_max_f32 x, x
// per input
VQT_FN float max3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
#else
    return std::fmax(std::fmax(a, b), c);
#endif
}
// two operands, ONE instruction and no -inf pad register (the builtin form of a two-operand maximum is canonicalised, see above)
VQT_FN float max2(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return std::fmax(a, b);
#endif
}
VQT_FN float med3(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c));
#endif
}
VQT_FN unsigned shl1_in(unsigned bits, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(bits, f2u(d), 31);
#else
    return (bits << 1) | (f2u(d) >> 31);
#endif
}
VQT_FN int popc(unsigned x) { return __builtin_popcount(x); }
VQT_FN int clz(unsigned x) { return __builtin_clz(x); }

constexpr unsigned kCellBits = 6;                 // cell id = 2 * tile + s < 64  (K <= 1024)
constexpr unsigned kCellMask = (1u << kCellBits) - 1u;
constexpr unsigned kKeyMask = ~kCellMask;

struct Lane {                                     // tracker state of one accumulator lane and one 32-row tile
    float S[8];
    float m1, m2, m3;
};

VQT_FN void init(Lane &L, float ninf) {
#pragma unroll
    for (int a = 0; a < 8; ++a) L.S[a] = ninf;
    L.m1 = ninf; L.m2 = ninf; L.m3 = ninf;
}

// One 32-code tile.  acc: the lane's 16 accumulator values (anything indexable); cell0 = 2 * tile, cell1 = cell0 + 1 (wave-uniform);
// ninf / pinf: -inf / +inf in registers the compiler cannot see through (so max(x, y) stays ONE v_max3 / v_med3);
// keymask: kKeyMask, likewise in a register (gfx950's v_and_or_b32 takes no literal: given the constant, hipcc emits v_and + v_or).
template <class ACC>
VQT_FN void tile(Lane &L, const ACC &acc, unsigned cell0, unsigned cell1, unsigned keymask, float ninf, float pinf) {
#pragma unroll
    for (int a = 0; a < 8; ++a) L.S[a] = max3(L.S[a], acc[a], acc[a + 8]);
    // (round 5: the fourth v_max3 of eight values repeats one of them instead of padding with a -inf register, and the largest
    // key's v_max3 repeats a smaller key: no +-inf registers live through the sweep, and still no two-operand maximum)
    const float x0 = max3(max3(max3(acc[0], acc[1], acc[2]), max3(acc[3], acc[4], acc[5]), acc[6]), acc[7], acc[0]);
    const float x1 = max3(max3(max3(acc[8], acc[9], acc[10]), max3(acc[11], acc[12], acc[13]), acc[14]), acc[15], acc[8]);
    const float k0 = u2f((f2u(x0) & keymask) | cell0);
    const float k1 = u2f((f2u(x1) & keymask) | cell1);
    L.m3 = med3(L.m2, L.m3, k0);
    L.m2 = med3(L.m1, L.m2, k0);
    L.m1 = max3(L.m1, k0, L.m3);                  // (m3 <= m1: the maximum of m1 and k0)
    L.m3 = med3(L.m2, L.m3, k1);
    L.m2 = med3(L.m1, L.m2, k1);
    L.m1 = max3(L.m1, k1, L.m3);
    (void)pinf; (void)ninf;
}

// largest value the lane has seen (exact accumulator bits)
VQT_FN float lane_max(const Lane &L, float ninf) {
    (void)ninf;
    return max3(max3(max3(L.S[0], L.S[1], L.S[2]), max3(L.S[3], L.S[4], L.S[5]), L.S[6]), L.S[7], L.S[0]);
}

VQT_FN int code_of(int a, int cell, int h) { return 32 * (cell >> 1) + (a & 3) + 8 * (a >> 2) + 16 * (cell & 1) + 4 * h; }

// Stage 1 (every row): what one half of a row knows once the row's threshold is known.
struct Half {
    unsigned ge;           // bit 7 - a: stream a of this half is at or above thr
    int popA;              // streams of this half at or above thr
    int nB;                // cell keys of this half at or above thrB = thr - (key truncation), 0..3 (3 = "three or more")
    int k11;               // code of (largest stream at or above thr, largest cell key)
};

VQT_FN Half half_of(const Lane &L, float thr, float thrB, int h) {
    unsigned lt = 0u;                                     // bit 7 - a = (S[a] < thr)
#pragma unroll
    for (int a = 0; a < 8; ++a) lt = shl1_in(lt, L.S[a] - thr);
    Half H;
    H.ge = ~lt & 0xffu;
    H.popA = popc(H.ge);
    H.nB = (L.m1 >= thrB ? 1 : 0) + (L.m2 >= thrB ? 1 : 0) + (L.m3 >= thrB ? 1 : 0);
    const int p1 = 31 - clz(H.ge | 1u);                   // highest set bit (ge == 0: p1 = 0, and popA = 0 says it is unused)
    H.k11 = code_of((7 - p1) & 7, (int)(f2u(L.m1) & kCellMask), h);
    return H;
}

// the word the two halves of a row swap: [popA : 4][nB : 2][k11 : 14]
VQT_FN unsigned pack(const Half &H) { return (unsigned)H.popA | ((unsigned)H.nB << 4) | ((unsigned)H.k11 << 6); }

struct Verdict {           // the same on both halves of a row
    bool closed;           // exactly one code reaches the threshold: kbest is the reference's argmin
    bool hard;             // more candidates than two streams x two cells per half cover: the row tile is screened again
    int kbest;             // closed rows
};

VQT_FN Verdict verdict_of(const Half &H, unsigned other, int K) {
    const int popO = (int)(other & 15u), nBO = (int)((other >> 4) & 3u), k11O = (int)(other >> 6);
    Verdict V;
    V.kbest = H.popA == 1 ? H.k11 : k11O;
    V.closed = (H.popA + popO == 1) && (H.nB + nBO == 1);
    V.hard = H.popA > 2 || H.nB > 2 || popO > 2 || nBO > 2;
    // a padding code (k >= K) at or above the threshold can only come from a broken screen: scan the row again
    if (V.closed && V.kbest >= K) { V.closed = false; V.hard = true; }
    if (V.hard) V.closed = false;
    return V;
}

// Stage 2 (rows that are neither closed nor hard): the exact tasks THIS half contributes -- the products of its (at most
// two) streams and (at most two) cells at or above the threshold, as pairs of codes.
struct Cands {
    int ntask;             // 0..2
    int ta[2], tb[2];
};

VQT_FN Cands cands_of(const Lane &L, const Half &H, int h, int K) {
    const int p1 = 31 - clz(H.ge | 1u);
    const unsigned ge2 = H.ge & ~(1u << p1);
    const int p2 = ge2 ? 31 - clz(ge2) : p1;
    const int a1 = (7 - p1) & 7, a2 = (7 - p2) & 7;
    const int c1 = (int)(f2u(L.m1) & kCellMask);
    const int c2 = H.nB >= 2 ? (int)(f2u(L.m2) & kCellMask) : c1;
    int k11 = code_of(a1, c1, h), k12 = code_of(a1, c2, h), k21 = code_of(a2, c1, h), k22 = code_of(a2, c2, h);
    // a product that is a padding code (k >= K; the last tile of a codebook with K % 32 != 0) is a phantom -- padding
    // scores sit at -3e38 -- and is replaced by a product that is a real code (one exists: the half's value at or above thr)
    const int kv = k11 < K ? k11 : (k12 < K ? k12 : (k21 < K ? k21 : k22));
    if (k11 >= K) k11 = kv;
    if (k12 >= K) k12 = kv;
    if (k21 >= K) k21 = kv;
    if (k22 >= K) k22 = kv;
    const bool two = H.popA == 2 && H.nB == 2;
    const bool any = H.popA >= 1 && H.nB >= 1 && kv < K;
    Cands C;
    C.ntask = any ? (two ? 2 : 1) : 0;
    C.ta[0] = k11; C.tb[0] = k22;
    C.ta[1] = k12; C.tb[1] = k21;
    return C;
}

// ---- round 5: what a half knows as ONE word, the verdict on the row's SPEAKER lane ---------------------------------------
// Same decisions as half_of / verdict_of / cands_of above (tests/host/trk_harness.cpp checks both against the brute-force scan
// and against each other), in a form that costs the kernel ~35 vector instructions per row tile instead of ~110:
//   * the eight stream tests and the three key tests go into ONE shift register, stream a at bit pos(a) = (a & 3) + 8 (a >> 2)
//     -- the code's offset inside its cell, code = 16 cell + pos + 4 h, so the highest set stream bit IS the low part of k11 --
//     the keys m1, m2, m3 at bits 4, 5, 6 (bit 7: a filler that is never set);
//   * a half's word  w = ge | pos << 12 | h << 14 | cell(m1) << 16  (bits 12..21 = k11) is all the other half ever needs, and only
//     ONE lane of the row needs both words: the row's speaker (the lane that later owns the row's index, histogram count and
//     gather).  One v_permlane32_swap hands the speakers of both row tiles their partners' words.
constexpr unsigned kGeStream = 0xF0Fu, kGeCell = 0x070u, kGeBits = 0xF7Fu;

VQT_FN unsigned ffbh(unsigned x) {                   // v_ffbh_u32: leading zeros, ~0 for 0
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;                                       // (the C form costs a compare and a select around the same instruction)
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x ? (unsigned)__builtin_clz(x) : ~0u;
#endif
}

// bit set = BELOW the threshold (the sign bits of the differences); the caller's ge = ~lt & gemask, gemask = kGeBits | h << 14
VQT_FN unsigned lt_of(const Lane &L, float thr, float thrB, float ninf) {
    unsigned lt = f2u(L.S[7] - thr) >> 31;
    lt = shl1_in(lt, L.S[6] - thr);
    lt = shl1_in(lt, L.S[5] - thr);
    lt = shl1_in(lt, L.S[4] - thr);
    lt = (lt << 1) | 1u;                              // bit 7: never "at or above"
    (void)ninf;
    lt = shl1_in(lt, L.m3 - thrB);
    lt = shl1_in(lt, L.m2 - thrB);
    lt = shl1_in(lt, L.m1 - thrB);
    lt = shl1_in(lt, L.S[3] - thr);
    lt = shl1_in(lt, L.S[2] - thr);
    lt = shl1_in(lt, L.S[1] - thr);
    lt = shl1_in(lt, L.S[0] - thr);
    return lt;
}

// (a half without a stream at or above thr: pos is garbage and spills into the cell field -- such a word's k11 is never selected)
VQT_FN unsigned word_of(const Lane &L, unsigned ge) {
    const unsigned p = 31u ^ ffbh(ge & kGeStream);
    return ge | (p << 12) | (f2u(L.m1) << 16);
}

struct Spoken {            // on the row's speaker lane: Wlo / Whi = the words of the row's lower / upper accumulator lane
    unsigned g;            // the halves' ge bits side by side: lower half at 0..11, upper at 12..23 (bits 24.. garbage)
    int pT, nT;            // streams / cell keys of the row at or above their thresholds
    int kbest;             // k11 of the half that has a stream at or above thr (the lower one if both)
};

VQT_FN Spoken spoken_of(unsigned Wlo, unsigned Whi) {
    Spoken V;
    V.g = (Wlo & 0xFFFu) | (Whi << 12);
    V.pT = popc(V.g & (kGeStream | kGeStream << 12));
    V.nT = popc(V.g & (kGeCell | kGeCell << 12));
    const unsigned sel = (Wlo & kGeStream) ? Wlo : Whi;
    V.kbest = (int)((sel >> 12) & 1023u);
    return V;
}
VQT_FN bool spoken_closed(const Spoken &V, int K) { return V.pT == 1 && V.nT == 1 && V.kbest < K; }
// (a row closed by its counts whose code is a padding code can only come from a broken screen: it is screened again, as above)
VQT_FN bool spoken_hard(const Spoken &V, int K) {
    const int a = popc(V.g & kGeStream), b = popc(V.g & (kGeStream << 12)), c = popc(V.g & kGeCell), d = popc(V.g & (kGeCell << 12));
    const int m = a > b ? a : b, n = c > d ? c : d;
    return (m > n ? m : n) > 2 || (V.pT == 1 && V.nT == 1 && V.kbest >= K);
}

// stage 2 from a half's own ge bits: the products of its (at most two) streams and cells at or above the thresholds
VQT_FN Cands cands2_of(const Lane &L, unsigned ge, int h, int K) {
    const unsigned x = ge & kGeStream;
    const int popA = popc(x), nB = popc(ge & kGeCell);
    const int p1 = 31 - clz(x | 1u);
    const unsigned x2 = x & ~(1u << p1);
    const int p2 = x2 ? 31 - clz(x2) : p1;
    const int c1 = (int)(f2u(L.m1) & kCellMask);
    const int c2 = nB >= 2 ? (int)(f2u(L.m2) & kCellMask) : c1;
    int k11 = 16 * c1 + p1 + 4 * h, k12 = 16 * c2 + p1 + 4 * h, k21 = 16 * c1 + p2 + 4 * h, k22 = 16 * c2 + p2 + 4 * h;
    const int kv = k11 < K ? k11 : (k12 < K ? k12 : (k21 < K ? k21 : k22));
    if (k11 >= K) k11 = kv;
    if (k12 >= K) k12 = kv;
    if (k21 >= K) k21 = kv;
    if (k22 >= K) k22 = kv;
    const bool two = popA == 2 && nB == 2;
    const bool any = popA >= 1 && nB >= 1 && kv < K;
    Cands C;
    C.ntask = any ? (two ? 2 : 1) : 0;
    C.ta[0] = k11; C.tb[0] = k22;
    C.ta[1] = k12; C.tb[1] = k21;
    return C;
}

// ---- the same stage 1 / stage 2 for trackers that keep their three largest cell maxima OUTSIDE the keys' 6-bit field
// (vq_chunk.hip: codebooks of up to 512 tiles; cell keys are folded every 16 tiles into running (value, global cell) triples)
struct Counts {
    unsigned ge;           // bit 7 - a: stream a at or above thr
    int popA, nB;
};

VQT_FN Counts counts_of(const float (&S)[8], float k1, float k2, float k3, float thr, float thrB) {
    unsigned lt = 0u;
#pragma unroll
    for (int a = 0; a < 8; ++a) lt = shl1_in(lt, S[a] - thr);
    Counts C;
    C.ge = ~lt & 0xffu;
    C.popA = popc(C.ge);
    C.nB = (k1 >= thrB ? 1 : 0) + (k2 >= thrB ? 1 : 0) + (k3 >= thrB ? 1 : 0);
    return C;
}

// products of the (at most two) streams and cells at or above the threshold: n = popA * nB codes (0 when either is 0);
// c1 / c2: GLOBAL cell ids (2 * tile + s) of the two largest cell maxima
struct Products {
    int n;
    int k[4];
};

VQT_FN Products products_of(const Counts &C, int c1, int c2, int h, int K) {
    const int p1 = 31 - clz(C.ge | 1u);
    const unsigned ge2 = C.ge & ~(1u << p1);
    const int p2 = ge2 ? 31 - clz(ge2) : p1;
    const int a1 = (7 - p1) & 7, a2 = (7 - p2) & 7;
    if (C.nB < 2) c2 = c1;
    Products P;
    P.k[0] = code_of(a1, c1, h); P.k[1] = code_of(a2, c2, h);      // the diagonal first: (1,1) and (2,2)
    P.k[2] = code_of(a1, c2, h); P.k[3] = code_of(a2, c1, h);
    const bool two = C.popA == 2 && C.nB == 2;
    P.n = (C.popA >= 1 && C.nB >= 1) ? (two ? 4 : (C.popA == 2 || C.nB == 2 ? 2 : 1)) : 0;
    // padding codes (k >= K) are phantoms: replace them by a product that is a real code
    const int kv = P.k[0] < K ? P.k[0] : (P.k[1] < K ? P.k[1] : (P.k[2] < K ? P.k[2] : P.k[3]));
    if (kv >= K) P.n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (P.k[i] >= K) P.k[i] = kv;
    return P;
}

// torch.argmin's order on (distance, index) as one unsigned 64-bit key (finite distances; -0 cannot occur: x - y of
// x >= +0 is never -0): smaller distance first, then smaller index
VQT_FN unsigned long long dist_key(float d, int k) {
    const unsigned u = f2u(d);
    const unsigned s = (u >> 31) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)s << 32) | (unsigned)k;
}

}  // namespace trk
}  // namespace vqvae
