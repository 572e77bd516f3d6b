// The fused kernels of the 32x32 path (8x8 latent maps): 3x3 conv(-transpose) + residual stack (+ 1x1 conv + quantizer), the
// encoder's first two layers, the decoder's last two layers; split out of conv.hip in round 4 (shared device code: conv_device.h).
#include "conv_host.h"

namespace vqvae {
// ---------------------------------------------------------------------------
// A 3x3 conv / 3x3 conv-transpose (stride 1) IN FRONT of a residual pair, all in one kernel (8x8 maps, two-term fp16
// products): models/encoder.py:35-38 (conv 3x3 + ReLU -> ResidualStack) and models/decoder.py:28-30 (conv-transpose 3x3 ->
// ResidualStack).  One wave owns one image.  The front conv accumulates straight into the registers that hold the
// residual layers' map (Y[m-tile][n-tile], accumulator layout: 128 channels x 64 pixels = 128 registers per lane); BOTH
// residual layers then take their 3x3 operands from Y through the in-LDS transposition of res_pair8_h2_kernel's second
// layer and their skip from Y itself.  The conv's output map and the first layer's output map never exist in memory.
// Operand order per accumulator = the separate kernels' (chunk, tap, k-step for the front conv as in
// conv_tile8_bf3_kernel; slice, tap for the residual 3x3): results are bitwise those of the separate launches.
// NT3 > 0: the 1x1 conv behind the pair as in res_pair8_h2_kernel.
// Everything is computed TRANSPOSED (weights = A operand, pixels = B operand): an accumulator lane then owns one pixel and
// its registers run over channels, which is the B-operand layout of the next GEMM up to half-wave swaps (acc_to_ksteps) --
// the 1x1 GEMMs take their inputs straight from registers and the 3x3 slices go registers -> fp16 planes in LDS without
// the accumulator -> LDS -> transposed read round trip of res_pair8_h2_kernel.
#ifndef CRP_NW
#define CRP_NW 4        // waves (= images) per workgroup of conv_res_pair8_h2_kernel.  8 (one workgroup per CU, weights shared by eight
                        // images, every stage barrier spanning all waves of the CU; tools/build_variant.py nw8 -DCRP_NW=8) is 17 us per
                        // step SLOWER: 0.589 vs 0.572 ms for the two launches
#endif
#ifndef CRP_MINW
#define CRP_MINW 2      // waves per SIMD the register allocation must allow (tools/build_variant.py crp1 -DCRP_MINW=1: 388 registers, no
                        // scratch, one workgroup per CU)
#endif
template <int NT3, bool VQ = false, bool GATHER = false, bool ZEOUT = false>
__global__ __launch_bounds__(CRP_NW * 64, CRP_MINW) void conv_res_pair8_h2_kernel(const float *__restrict__ in, FrontConv fc,
                                                                   const u32x4 *__restrict__ w1img, const u32x4 *__restrict__ w2img,
                                                                   float *__restrict__ out, int B, int flags,
                                                                   const int *__restrict__ hdr1, const int *__restrict__ hdr2,
                                                                   const int *__restrict__ in_amax, int *__restrict__ out_amax,
                                                                   const u32x4 *__restrict__ w3img, const int *__restrict__ hdr3,
                                                                   const float *__restrict__ bias3, float *__restrict__ out3,
                                                                   int *__restrict__ zero_buf, int zero_n, VqFuse vq) {
    static_assert(NT3 == 0 || NT3 == 1 || NT3 == 2 || NT3 == 4, "the 1x1 post conv streams through NT3 weight stages of 16 KiB");
    static_assert(!VQ || (NT3 == 2 && CRP_NW == 4), "the fused quantizer takes the 64-channel z_e of four images per workgroup");
    constexpr int NT2 = 4, C = 128, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // front conv: [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int RBUF = 4 * HP;                           // residual slice: [term 2][half 2][pixel + zero]; two buffers = TILE4
    __shared__ u32x4 As_all[CRP_NW * TILE4];
    // Weights stream through two 18 KiB LDS buffers shared by the workgroup's four images, filled by LDS-DMA (no staging
    // registers) one stage ahead; one workgroup barrier per stage.  Stages: one (tap, chunk) of the front conv (16 pieces
    // of 1 KiB); then per residual layer the nine taps of each 16-channel slice of the 3x3 (18 pieces) x 8 and the 1x1
    // (16 pieces); then the 1x1 post conv in NT3 parts of 4 / NT3 channel tiles (16 pieces each).  Per-wave loads straight from L2 cost 87 + 51 us
    // per step in exposed latency (knock-outs, profiles/r02_vq_stream.txt).
    constexpr int WBUF = 18 * 64;
    // stages after the front conv: 18 (two residual layers) + NT3 (post conv) + (VQ) one per four 32-code tiles of the codebook
    const int nvq = VQ ? (vq.K32 >> 7) : 0;
    const int NSTAGE = 18 + NT3 + nvq;
    __shared__ u32x4 Wb_all[2 * WBUF];
    // fused quantizer: per-wave tables (vq_unit.h), the workgroup's histogram and loss partials
    __shared__ __attribute__((aligned(16))) unsigned char vq_tab_all[VQ ? CRP_NW * 1040 : 16];
    __shared__ int vq_hist_s[VQ ? 1024 : 1];             // (K <= 1024: 32 code tiles = the tracker's 6-bit cell field; 2 x 80 384 B of LDS per CU)
    // the four weight tensors' per-output-channel scales 2^-kw[c]: front conv [0, 128), residual 3x3 [128, 160), residual 1x1
    // [160, 288), post conv [288, 288 + 32 NT3) (a stage barrier precedes every use)
    __shared__ __attribute__((aligned(16))) float dw_s[288 + 32 * (NT3 > 0 ? NT3 : 1)];
    for (int i = threadIdx.x; i < 288 + 32 * NT3; i += CRP_NW * 64)
        dw_s[i] = i < 128 ? h2_dw(fc.hdr)[i] : (i < 160 ? h2_dw(hdr1)[i - 128] : (i < 288 ? h2_dw(hdr2)[i - 160] : h2_dw(hdr3)[i - 288]));
    __shared__ double vq_red_s[VQ ? CRP_NW : 1];
    if constexpr (VQ) {
        for (int i = threadIdx.x; i < vq.K; i += CRP_NW * 64) vq_hist_s[i] = 0;        // (a stage barrier precedes every use)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    float *Hs = reinterpret_cast<float *>(As);
    const bool relu_out = flags & kFlagReluOut;            // of the SECOND residual layer (the stack's final ReLU)
    constexpr int cpt = C >> 5;

    const long long img = (long long)blockIdx.x * CRP_NW + wave;
    const bool img_ok = img < B;
    // a buffer the NEXT kernel of the stream wants zeroed (the quantizer's histogram: saves a fill launch per step)
    if (zero_buf && blockIdx.x == 0)
        for (int i = tid; i < zero_n; i += CRP_NW * 64) zero_buf[i] = 0;

    // pixel bookkeeping: the residual 3x3 (taps t/3-1, t%3-1) and the front conv (taps from the geometry masks)
    int spx[MT];
    unsigned tapok[MT], tapok0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, x = spx[mt] & 7;
        unsigned m = 0, m0 = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << t;
            const int y0 = y + (int)((fc.dym >> (4 * t)) & 15) - 8, x0 = x + (int)((fc.dxm >> (4 * t)) & 15) - 8;
            if (y0 >= 0 && y0 < 8 && x0 >= 0 && x0 < 8) m0 |= 1u << t;
        }
        tapok[mt] = m;
        tapok0[mt] = m0;
    }

    // Y[mt][nt][r]: channel 32 nt + (r & 3) + 8 (r >> 2) + 4 h of pixel 32 mt + l31
    f32x16 Y[MT][NT2];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // one 1 KiB piece global -> LDS: every lane's 16 bytes land at dst + 16 lane
    // Issued as inline assembly: for the builtin hipcc puts s_waitcnt vmcnt(0) in front of every later LDS read that it
    // cannot prove disjoint from the destination -- i.e. it waits for the NEXT stage's pieces before reading this stage's.
    // The waits are explicit here (dma_wait_sync); the compiler's own vmcnt waits stay correct (loads return in order and an
    // uncounted outstanding load only makes a counted wait longer).
    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // the nine taps of slice sl of the residual 3x3 -> buffer `buf`: piece p = tap * 2 + term
    auto dma_slice = [&](int sl, int buf) {
        const u32x4 *base = w1img + (size_t)(sl >> 1) * 256 + (sl & 1) * 64;
        for (int p = wave_u; p < 18; p += CRP_NW)
            dma(base + (size_t)(p >> 1) * cpt * 256 + (p & 1) * 128, Wb_all + buf * WBUF + p * 64);
    };
    // 16 KiB of an image as it lies (the 1x1 GEMMs): four pieces per wave
    auto dma_linear = [&](const u32x4 *src, int buf) {
#pragma unroll
        for (int j = 0; j < 16 / CRP_NW; ++j) dma(src + (wave_u * (16 / CRP_NW) + j) * 64, Wb_all + buf * WBUF + (wave_u * (16 / CRP_NW) + j) * 64);
    };
    // stage k after the front conv: 9 LI + slice (3x3 of layer LI), 9 LI + 8 (its 1x1), 18 + j (part j of the post conv)
    auto dma_stage = [&](int k, int buf) {
        if (VQ && k >= 18 + NT3) {
            // four 32-code tiles of the codebook's fp16 image (16 pieces) + their seeds -A ee / 2 (512 bytes of piece 16)
            const int j = k - (18 + NT3);
            dma_linear(reinterpret_cast<const u32x4 *>(vq.imgf) + (size_t)j * 1024, buf);
            if (wave_u == 0) dma(reinterpret_cast<const u32x4 *>(vq.seeds) + (size_t)j * 32, Wb_all + buf * WBUF + 16 * 64);
        }
        else if (k >= 18) dma_linear(w3img + (size_t)(k - 18) * 1024, buf);
        else if (k % 9 == 8) dma_linear(w2img, buf);
        else dma_slice(k % 9, buf);
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    float ymax = 0.0f;                                     // largest |Y| (the next consumer's scale)
    int wstage = 0;                                        // weight stages of the front conv (LDS buffer parity)
    // =========================================== front conv ===========================================
    {
        const int cpt0 = fc.Cin >> 5;
        const float *src = in + ((size_t)(img_ok ? img : 0) * PX + lane) * fc.Cin;     // this lane's pixel row
        // decode-from-indices (vqvae_decode_f32; the GATHER instance): `in` is the codebook and a pixel's row is its code's row --
        // z_q never exists in memory (visualization.ipynb:358-365).  An index outside [0, K) never reads the codebook: its
        // pixel becomes NaN, as in vqvae_vq_decode_indices_f32
        bool gbad = false;
        if constexpr (GATHER) {
            const long long k = vq.idx[(size_t)(img_ok ? img : 0) * PX + lane];
            gbad = k < 0 || k >= vq.K;
            src = in + (size_t)(gbad ? 0 : k) * fc.Cin;
        }
        const bool gany = GATHER && __builtin_amdgcn_ballot_w64(gbad) != 0;
        f32x4 raw[8];
        auto load_raw0 = [&](int cc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4 *>(src + 32 * cc + 4 * j);
            if (gany) {
                const float qn = __builtin_nanf("");
#pragma unroll
                for (int j = 0; j < 8; ++j) if (gbad) raw[j] = f32x4{qn, qn, qn, qn};
            }
        };
        float m = 0.0f;
        const int given = (in_amax && img_ok) ? in_amax[img] : -1;
        if (given >= 0) m = __int_as_float(given);
        else for (int cc = 0; cc < cpt0; ++cc) {
            load_raw0(cc);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(raw[j].x), __builtin_fabsf(raw[j].y)), fmaxf(__builtin_fabsf(raw[j].z), __builtin_fabsf(raw[j].w))));
        }
        const int kx = wave_scale_exp(img_ok ? m : 0.0f);
        const float xs = __builtin_ldexpf(1.0f, kx), d0 = __builtin_ldexpf(1.0f, -kx);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[mt][nt][r] = 0.0f;
        // padding pixels of the eight planes (the residual slices' two plane buffers have theirs at the same units)
        if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};
        // stage (cc, tap) = 16 pieces: piece p = nt * 4 + t * 2 + term, four per wave
        auto dma_front = [&](int cc, int tap, int buf) {
            const u32x4 *base = fc.wimg + (size_t)(tap * cpt0 + cc) * (NT2 * 256);
#pragma unroll
            for (int j = 0; j < 16 / CRP_NW; ++j) {
                const int p = wave_u * (16 / CRP_NW) + j;
                dma(base + (p >> 2) * 256 + ((p >> 1) & 1) * 64 + (p & 1) * 128, Wb_all + buf * WBUF + p * 64);
            }
        };
        load_raw0(0);
        dma_front(0, 0, 0);
        for (int cc = 0; cc < cpt0; ++cc) {
            // park the chunk: k-step t, half hh hold channels 16 hh + 8 t + [0, 8) (conv_tile8_bf3_kernel's stage())
            __builtin_amdgcn_wave_barrier();
            u32x4 *dst = As + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    u32x4 t1, t2;
                    split8_h(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], xs, t1, t2);
                    dst[(t * 2 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 2 + 1) * PLANE + hh * HP] = t2;
                }
            lds_order_wave();
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap, ++wstage) {
                // the pixel operands do not depend on the stage buffer: read them before the barrier
                const int shift = ((int)((fc.dym >> (4 * tap)) & 15) - 8) * 8 + ((int)((fc.dxm >> (4 * tap)) & 15) - 8);
                u32x4 X[2][MT][2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int p = ((tapok0[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                        const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                        X[t][mt][0] = ap[0];
                        X[t][mt][1] = ap[PLANE];
                    }
                // this stage's weights are in; everyone is done with the other buffer.  The next chunk's eight activation loads
                // go out BEHIND tap 1's weights and may stay in flight across tap 1's wait (they are its youngest requests):
                // in front of tap 0's wait, as before, every chunk sat out their whole latency at that barrier
                if (tap == 1 && cc + 1 < cpt0) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __syncthreads();
                } else dma_wait_sync();
                if (tap + 1 < 9) dma_front(cc, tap + 1, (wstage + 1) & 1);
                else if (cc + 1 < cpt0) dma_front(cc + 1, 0, (wstage + 1) & 1);
                else dma_stage(0, (wstage + 1) & 1);           // the first slice of the first residual layer
                if (tap == 0 && cc + 1 < cpt0) load_raw0(cc + 1);
                const u32x4 *wt = Wb_all + (wstage & 1) * WBUF + lane;
                // group g = (t, nt): weights one group ahead of the matrix instructions
                u32x4 Wc0 = wt[0], Wc1 = wt[64];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int t = g >> 2, nt = g & 3;
                    u32x4 Wn0 = Wc0, Wn1 = Wc1;
                    if (g + 1 < 8) {
                        const u32x4 *bp = wt + (((g + 1) & 3) * 4 + ((g + 1) >> 2) * 2) * 64;
                        Wn0 = bp[0];
                        Wn1 = bp[64];
                    }
                    __builtin_amdgcn_sched_barrier(0);         // hipcc otherwise sinks the reads to just before their use
                    prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, Y[0][nt], Y[1][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    Wc0 = Wn0;
                    Wc1 = Wn1;
                }
            }
        }
        // bias + ReLU (encoder.py:36 / the stack's first in-place ReLU applied by the producer, decoder.py:29-30)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (fc.bias) bv = *reinterpret_cast<const f32x4 *>(fc.bias + nt * 32 + 8 * g + 4 * h);
                const f32x4 dv = h2_dw4(dw_s, nt * 32, g, h, d0);          // 2^-(kx + kw[c]) of registers 4 g .. 4 g + 3
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; q += 2)
                        SCALE2_BIAS_RELU2(Y[mt][nt][4 * g + q], Y[mt][nt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], ymax);
            }
    }
    lds_order_wave();

    // =========================================== residual layers from Y ===========================================
    // stage k: wait for its weights, start the next stage's, return this lane's column of its buffer
    auto stage_sync = [&](int k) -> const u32x4 * {
        dma_wait_sync();
        if (k + 1 < NSTAGE) dma_stage(k + 1, (wstage + k + 1) & 1);
        return Wb_all + ((wstage + k) & 1) * WBUF + lane;
    };
    f32x16 acc1[MT];
    // nine taps of one slice: wb = this lane's column of the stage buffer ([tap * 2 + term] x 64 units), pl = this half-wave's
    // planes of the slice; operands of tap + 1 are read while tap's products run
    auto taps = [&](const u32x4 *wb, const u32x4 *pl) {
        u32x4 Xc[MT][2], Wc[2];
        auto ld = [&](int tap, u32x4(&X)[MT][2], u32x4(&W)[2]) {
            const int shift = (tap / 3 - 1) * 8 + (tap % 3 - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = ((tapok[mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                X[mt][0] = pl[p];
                X[mt][1] = pl[p + HP * 2];
            }
            W[0] = wb[tap * 128];
            W[1] = wb[tap * 128 + 64];
        };
        ld(0, Xc, Wc);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            u32x4 Xn[MT][2], Wn[2];
            if (tap + 1 < 9) ld(tap + 1, Xn, Wn);
            __builtin_amdgcn_sched_barrier(0);
            prod3x2t(Xc[0][0], Xc[0][1], Xc[1][0], Xc[1][1], Wc[0], Wc[1], acc1[0], acc1[1]);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { Xc[mt][0] = Xn[mt][0]; Xc[mt][1] = Xn[mt][1]; }
                Wc[0] = Wn[0]; Wc[1] = Wn[1];
            }
        }
    };
    // one k-step's operands of both pixel tiles -> plane buffer `buf`
    auto put_planes = [&](int buf, const u32x4(&T1)[MT][2], const u32x4(&T2)[MT][2], int t) {
        u32x4 *pb = As + buf * RBUF + h * HP + l31;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            pb[32 * mt] = T1[mt][t];
            pb[32 * mt + 2 * HP] = T2[mt][t];
        }
    };
    // Y <- [relu](Y + W2 relu(W1 (*) Y)); ymax in: largest Y, out: largest new Y.  always_inline: hipcc does not inline a
    // lambda this size twice by itself, and Y (captured by reference) then lives in scratch memory -- 5 ms per launch,
    // measured; a two-iteration loop around the body instead spills 275 registers
    auto layer = [&](auto LT, bool relu_after) __attribute__((always_inline)) {
        constexpr int LI = decltype(LT)::value;                // 0 or 1: stages 9 LI ..
        const int kx = wave_scale_exp(img_ok ? ymax : 0.0f);
        const float xscale = __builtin_ldexpf(1.0f, kx), d1 = __builtin_ldexpf(1.0f, -kx);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mt][r] = 0.0f;
        u32x4 T1[MT][2], T2[MT][2];                            // [pixel tile][k-step] of the current 32-channel tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][0], xscale, T1[mt], T2[mt]);
        __builtin_amdgcn_wave_barrier();
        put_planes(0, T1, T2, 0);
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            // slice (c, 0) from plane buffer 0; (c, 1)'s planes go to buffer 1 (last read by slice (c - 1, 1))
            __builtin_amdgcn_wave_barrier();
            put_planes(1, T1, T2, 1);
            lds_order_wave();
            taps(stage_sync(9 * LI + 2 * c), As + h * HP);
            // slice (c, 1); the next tile's operands are made now and its first k-step goes to buffer 0
            if (c + 1 < NT2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][c + 1], xscale, T1[mt], T2[mt]);
                __builtin_amdgcn_wave_barrier();
                put_planes(0, T1, T2, 0);
            }
            lds_order_wave();
            taps(stage_sync(9 * LI + 2 * c + 1), As + RBUF + h * HP);
        }
        // hidden tile -> B operands of the 1x1 GEMM, in registers
        float m = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 dv = h2_dw4(dw_s + 128, 0, g, h, d1);            // 2^-(kx + kw1[j]) of the hidden channels 8 g + 4 h + [0, 4)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; q += 2)
                    SCALE2_BIAS_RELU2(acc1[mt][4 * g + q], acc1[mt][4 * g + q + 1], dv[q], dv[q + 1], 0.0f, 0.0f, m);
        }
        const int kh = wave_scale_exp(m);
        const float hscale = __builtin_ldexpf(1.0f, kh), d2 = __builtin_ldexpf(1.0f, -kh);
        u32x4 H1[MT][2], Hb[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(acc1[mt], hscale, H1[mt], Hb[mt]);
        const u32x4 *wb = stage_sync(9 * LI + 8);              // [nt][term][k-step] x 64 units
        float nmax = 0.0f;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            // (no read-ahead here: four short GEMMs per layer, and 32 more live registers spill next to Y, H and acc2)
            u32x4 Wc[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) { Wc[t][0] = wb[nt * 256 + t * 64]; Wc[t][1] = wb[nt * 256 + t * 64 + 128]; }
            f32x16 acc2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                prod3x2t(H1[0][t], Hb[0][t], H1[1][t], Hb[1][t], Wc[t][0], Wc[t][1], acc2[0], acc2[1]);
            // Y <- [relu](Y + acc2 * 2^-k), nmax: FMA (exact product), single-instruction max; the ReLU flag is wave-uniform and
            // decided once per tile, not per value
            // (the 1x1 rows' own scales: 2^-(kh + kw2[c]) of registers 4 g .. 4 g + 3, four at a time)
            if (relu_after) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dv = h2_dw4(dw_s + 160, nt * 32, g, h, d2);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const int r = 4 * g + q;
                            const float y0 = relu1(__builtin_fmaf(acc2[mt][r], dv[q], Y[mt][nt][r]));
                            const float y1 = relu1(__builtin_fmaf(acc2[mt][r + 1], dv[q + 1], Y[mt][nt][r + 1]));
                            Y[mt][nt][r] = y0;
                            Y[mt][nt][r + 1] = y1;
                            vmax3(nmax, y0, y1);
                        }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 dv = h2_dw4(dw_s + 160, nt * 32, g, h, d2);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const int r = 4 * g + q;
                            const float y0 = __builtin_fmaf(acc2[mt][r], dv[q], Y[mt][nt][r]), y1 = __builtin_fmaf(acc2[mt][r + 1], dv[q + 1], Y[mt][nt][r + 1]);
                            Y[mt][nt][r] = y0;
                            Y[mt][nt][r + 1] = y1;
                            vmax3_abs(nmax, y0, y1);
                        }
                }
            }
        }
        ymax = nmax;
    };
    layer(std::integral_constant<int, 0>{}, true);         // the second layer's in-place ReLU is applied by its producer
    layer(std::integral_constant<int, 1>{}, relu_out);
    if (out_amax && img_ok) publish_amax_exclusive(out_amax, img, ymax, lane);

    // one transposed 32-pixel x 32-channel tile -> rows of `ld` floats at dst (pixel-major), whole 128-byte lines per
    // eight lanes: registers -> wave-private LDS tile [pixel][36] -> linear 16-byte reads
    auto store_tile = [&](const float(&v)[16], float *dst, int ld) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(Hs + l31 * 36 + 8 * g + 4 * h) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
        lds_order_wave();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = (lane >> 3) + 8 * k;
            *reinterpret_cast<f32x4 *>(dst + (size_t)p * ld + 4 * (lane & 7)) = *reinterpret_cast<const f32x4 *>(Hs + p * 36 + 4 * (lane & 7));
        }
    };
    const long long wbase = img * PX;
    if constexpr (NT3 > 0) {
        const int kx3 = wave_scale_exp(img_ok ? ymax : 0.0f);
        const float xs3 = __builtin_ldexpf(1.0f, kx3), d3 = __builtin_ldexpf(1.0f, -kx3);
        f32x16 acc3[MT][NT3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[mt][n3][r] = 0.0f;
        const u32x4 *wb = nullptr;
#pragma unroll
        for (int c = 0; c < NT2; ++c) {
            u32x4 T1[MT][2], T2[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(Y[mt][c], xs3, T1[mt], T2[mt]);
            constexpr int CPS = 4 / NT3;                       // channel tiles per stage: [c % CPS][n3][term][k-step] x 64 units
            if (c % CPS == 0) wb = stage_sync(18 + c / CPS);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) {
                    const u32x4 *bp = wb + ((c % CPS) * NT3 + n3) * 256 + t * 64;
                    prod3x2t(T1[0][t], T2[0][t], T1[1][t], T2[1][t], bp[0], bp[128], acc3[0][n3], acc3[1][n3]);
                }
        }
        if constexpr (VQ) {
            // ================= the quantizer on z_e = this image's 64 rows, straight from the accumulators =================
            // (models/vqvae.py:33-34: z_e is never written.)  acc3 <- z_e: lane = row 32 mt + l31, register = channel
            // 32 n3 + (r & 3) + 8 (r >> 2) + 4 h.  The codebook's fp16 image streams through the weight stages (four 32-code
            // tiles + their seeds per stage, the image in THIS kernel's channel order: vq_prepare16_kernel's `imgf`); the sweep,
            // the trackers and everything behind them are vq_track.hip's (vq_track.h / vq_unit.h).
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias3) bv = *reinterpret_cast<const f32x4 *>(bias3 + n3 * 32 + 8 * g + 4 * h);
                        const f32x4 dv = h2_dw4(dw_s + 288, n3 * 32, g, h, d3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc3[mt][n3][4 * g + q] = acc3[mt][n3][4 * g + q] * dv[q] + bv[q];
                    }
            // fp16 B operands of the screen: k-step ks = 2 n3 + t, this half's channels 32 n3 + 16 h + 8 t + [0, 8)
            u32x4 zb[MT][4];
            float zn2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float sq = 0.0f;
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        float P[4], Q[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            P[q] = acc3[mt][n3][4 * t + q];
                            Q[q] = acc3[mt][n3][8 + 4 * t + q];
                            swap_halves(P[q], Q[q]);
                        }
                        u32x4 v;
                        v.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{P[0], P[1]}), f16x2));
                        v.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{P[2], P[3]}), f16x2));
                        v.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{Q[0], Q[1]}), f16x2));
                        v.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v{Q[2], Q[3]}), f16x2));
                        zb[mt][2 * n3 + t] = v;
                        sq = sqsum8_f16(v.x, v.y, v.z, v.w, sq);       // (not four fdot2 builtins: miscompiled, common.h)
                    }
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
                zn2[mt] = sq + __uint_as_float(h ? sw[0] : sw[1]);
            }
            const float inf = __builtin_inff();
            const float pinf = inf, ninf = -inf;             // (round 5: the tracker pads with neither; plain constants)
            unsigned keymask = trk::kKeyMask;
            asm volatile("" : "+v"(keymask));
            trk::Lane L[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) trk::init(L[mt], ninf);
            // one stage = four code tiles: [tile][k-step 4][half 2][code 32] x 16 bytes, seeds [tile][half][16] floats in piece 16
            auto sweep_stage = [&](const u32x4 *wb, const float *sd, int j, auto &&use) {
#pragma unroll
                for (int ctl = 0; ctl < 4; ++ctl) {
                    f32x16 seed;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 e4 = *reinterpret_cast<const f32x4 *>(sd + ctl * 32 + h * 16 + 4 * g);
                        seed[4 * g] = e4.x; seed[4 * g + 1] = e4.y; seed[4 * g + 2] = e4.z; seed[4 * g + 3] = e4.w;
                    }
                    u32x4 a[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) a[ks] = wb[ctl * 256 + ks * 64];
                    use(4 * j + ctl, a, seed);
                }
            };
            for (int j = 0; j < nvq; ++j) {
                const u32x4 *wb = stage_sync(18 + NT3 + j);
                const float *sd = reinterpret_cast<const float *>(wb - lane + 16 * 64);
                sweep_stage(wb, sd, j, [&](int ct, const u32x4(&a)[4], const f32x16 &seed) {
                    f32x16 acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, zb[mt][0]), seed, 0, 0, 0);
#pragma unroll
                        for (int ks = 1; ks < 4; ++ks)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks]), __builtin_bit_cast(f16x8, zb[mt][ks]), acc[mt], 0, 0, 0);
                    }
                    unsigned cell0 = (unsigned)(2 * ct), cell1 = cell0 + 1u;
                    asm volatile("" : "+s"(cell0), "+s"(cell1));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) trk::tile(L[mt], acc[mt], cell0, cell1, keymask, ninf, pinf);
                });
            }
            // ---- verdicts; exact tasks
            // (round 5: the unit in speaker form, vq_unit.h -- lane L speaks for row L of the image; the same decisions and bits)
            vqu::Tables tb = vqu::tables(vq_tab_all + wave_u * 1040);
            const vqu::BoundP bound = vqu::load_boundp(vq.flags);
            vqu::RowsSp<MT> R;
            vqu::classify_sp<MT>(L, zn2, bound, vq.K, lane, img_ok ? PX : 0, ninf, tb.task_s, R);
            vqu::Flagged FL = vqu::exact_begin_sp<MT>(R, lane, tb);
            int ntasks = FL.ndirect;
            // rows whose candidates the stream x cell products do not cover (~0.01 %) need the codebook image once more: the
            // workgroup votes, and if any of its waves has one, all four stream the stages again (the others only keep the barriers)
            const bool rescan_me = FL.hmask && FL.ndirect <= 64;
            int nres = 0;                               // tasks of the second screen (wave-uniform)
            if (__syncthreads_or(rescan_me ? 1 : 0)) {
                dma_stage(18 + NT3, 0);
                for (int j = 0; j < nvq; ++j) {
                    dma_wait_sync();
                    if (j + 1 < nvq) dma_stage(18 + NT3 + j + 1, (j + 1) & 1);
                    if (rescan_me && FL.ndirect + nres <= 64) {         // (past 64: the hard rows go wide anyway; the stages still stream)
                        const u32x4 *wb = Wb_all + (j & 1) * WBUF + lane;
                        const float *sd = reinterpret_cast<const float *>(Wb_all + (j & 1) * WBUF + 16 * 64);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            if ((unsigned)(FL.hmask >> (32 * mt))) {
                                const float thr_t = (((unsigned)(FL.hmask >> (32 * mt)) >> l31) & 1u) ? R.thr[mt] : inf;
                                sweep_stage(wb, sd, j, [&](int ct, const u32x4(&a)[4], const f32x16 &seed) {
                                    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, zb[mt][0]), seed, 0, 0, 0);
#pragma unroll
                                    for (int ks = 1; ks < 4; ++ks)
                                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ks]), __builtin_bit_cast(f16x8, zb[mt][ks]), acc, 0, 0, 0);
                                    vqu::rescan_tile(acc, thr_t, ct, mt, lane, vq.K, FL.ndirect, ninf, tb, nres);
                                });
                            }
                    }
                }
                if (rescan_me) {
                    lds_order_wave();
                    ntasks = FL.ndirect + nres;
                }
            }
            __syncthreads();                        // every wave is done with the weight buffers: they hold half of the rows now
            // ---- z_e rows (fp32) -> LDS: rows 0..31 in this wave's plane region, 32..63 in its quarter of the weight buffers;
            // 256 bytes per row, the 16-byte chunk c of row r at slot c ^ (r & 15)
            unsigned char *zlo = reinterpret_cast<unsigned char *>(As);
            unsigned char *zhi = reinterpret_cast<unsigned char *>(Wb_all) + (size_t)wave_u * 9216;
            auto zchunk = [&](int row, int c16) -> f32x4 * {
                unsigned char *b = row < 32 ? zlo + row * 256 : zhi + (row - 32) * 256;
                return reinterpret_cast<f32x4 *>(b + (((unsigned)c16 ^ ((unsigned)row & 15u)) << 4));
            };
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *zchunk(32 * mt + l31, 8 * n3 + 2 * g + h) = f32x4{acc3[mt][n3][4 * g], acc3[mt][n3][4 * g + 1], acc3[mt][n3][4 * g + 2], acc3[mt][n3][4 * g + 3]};
            lds_order_wave();
            if constexpr (ZEOUT) {
                // debug instance (VQVAE_FWD_DEBUG_ZE): the z_e rows the quantizer works on, bit for bit, for the oracle (tests only)
                if (img_ok)
#pragma unroll
                    for (int c16 = 0; c16 < 16; ++c16)
                        *reinterpret_cast<f32x4 *>(out3 + ((size_t)img * PX + lane) * 64 + 4 * c16) = *zchunk(lane, c16);
            }
            vqu::exact_end_sp<MT>(R, FL, ntasks, lane, tb, vq.cb, vq.ee, vq.K,
                           [&](int rr, int jc) { return *zchunk(rr, jc); },
                           [&](int rr, int c) { return reinterpret_cast<const float *>(zchunk(rr, c >> 2))[c & 3]; });
            const int j16 = lane & 15, g4 = lane >> 4;
            const float sacc = vqu::epilogue_sp<false, MT>(R, lane, vq.cb, vq.K, [&](int t, int i) { return *zchunk(32 * t + 4 * i + g4, j16); },
                                             (img_ok && vq.zq) ? vq.zq + (size_t)img * PX * 64 : nullptr, img_ok ? PX : 0,
                                             vq.idx + (size_t)(img_ok ? img : 0) * PX, vq_hist_s);
            // loss partial and histogram of the workgroup (fixed order: run-to-run bitwise loss / perplexity)
            double dacc = img_ok ? (double)sacc : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o);
            if (lane == 0) vq_red_s[wave_u] = dacc;
            __syncthreads();
            if (tid == 0) {
                double sum = 0.0;
                for (int w = 0; w < CRP_NW; ++w) sum += vq_red_s[w];
                vq.partials[blockIdx.x] = sum;
            }
            for (int k = tid; k < vq.K; k += CRP_NW * 64) {
                const int c = vq_hist_s[k];
                if (c) atomicAdd(&vq.hist[k], c);
            }
        } else
        if (img_ok) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int n3 = 0; n3 < NT3; ++n3) {
                    float v[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (bias3) bv = *reinterpret_cast<const f32x4 *>(bias3 + n3 * 32 + 8 * g + 4 * h);
                        const f32x4 dv = h2_dw4(dw_s + 288, n3 * 32, g, h, d3);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[4 * g + q] = acc3[mt][n3][4 * g + q] * dv[q] + bv[q];
                    }
                    store_tile(v, out3 + (wbase + mt * 32) * (32 * NT3) + n3 * 32, 32 * NT3);
                }
        }
    } else if (img_ok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = Y[mt][nt][r];
                store_tile(v, out + (wbase + mt * 32) * C + nt * 32, C);
            }
    }
}

// ---------------------------------------------------------------------------
// Encoder front in one launch (models/encoder.py:29-34): Conv2d(3 -> 64, 4x4 s2 p1) + ReLU + Conv2d(64 -> 128, 4x4 s2 p1)
// + ReLU on 32x32 images; the 16x16x64 map between them (64 KiB per image, written and read back by the separate
// kernels: 536 MB per 4096 images) never exists.  One wave owns one image; everything is computed transposed as in
// conv_res_pair8_h2_kernel.  The second conv runs as in conv_tile8_bf3_kernel<.., S2D>: a conv over the 8x8 grid of 2x2
// blocks of the 16x16 map, chunk = (block sub-position s, 32-channel slice), four block offsets (taps) per chunk.  Per
// sub-position the wave builds ITS OWN operand slice: the 64 block pixels' 4x4x3 input patches are gathered from the NCHW
// image (two 16-byte loads per channel and lane half: rows ky = 2h, 2h+1), split into fp16 terms and multiplied with the
// first layer's weights on the matrix cores (3 k-steps of 16 = the 48 taps; +9 % matrix work), + bias, ReLU, and the
// accumulator becomes the second conv's B operands by half-wave swaps (acc_to_ksteps).
// Scales: the image's largest |x| is measured; the first layer's outputs are bounded by L1 * max|x| + max|b| (L1 = the
// largest absolute row sum of its weights, in the header) -- a power of two up to ~8x above the true maximum, which costs
// the second term's range three bits at the very bottom and nothing where it matters (see split8_h).
#ifndef EF_MINW
#define EF_MINW 2
#endif
template <int CIN>
__global__ __launch_bounds__(256, EF_MINW) void enc_front8_h2_kernel(const float *__restrict__ x, const u32x4 *__restrict__ w0img,
                                                               const int *__restrict__ hdr0, const float *__restrict__ bias0,
                                                               const u32x4 *__restrict__ w2img, const int *__restrict__ hdr2,
                                                               const float *__restrict__ bias2, float *__restrict__ out, int B,
                                                               int *__restrict__ out_amax, int *__restrict__ zero_buf, int zero_n) {
    constexpr int NT = 4, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2, C0 = 64, C = 128;
    // ints a LATER kernel of the stream wants zeroed (the quantizer's histogram when the encoder's last kernel quantizes)
    if (zero_buf && blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_buf[i] = 0;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int WBUF = 16 * 64, NSTAGE = 32;             // a stage = one (chunk, tap) of the second conv: 16 pieces of 1 KiB
    __shared__ u32x4 As_all[4 * TILE4];
    __shared__ u32x4 Wb_all[2 * WBUF];
    __shared__ u32x4 W0s[2 * CIN * 2 * 64];                // first layer: [slice 2][ci][term 2] x 64 lanes
    __shared__ __attribute__((aligned(16))) float dw_s[64 + 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    float *Hs = reinterpret_cast<float *>(As);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;

    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // stage k = chunk * 4 + tap, chunk = 2 s + slice: 16 KiB as it lies in the space-to-depth image
    auto dma_stage = [&](int k, int buf) {
        const u32x4 *src = w2img + (size_t)k * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(src + (wave_u * 4 + j) * 64, Wb_all + buf * WBUF + (wave_u * 4 + j) * 64);
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    dma_stage(0, 0);
    for (int i = tid; i < 2 * CIN * 2 * 64; i += 256) W0s[i] = w0img[i];
    // both layers' per-output-channel weight scales 2^-kw[c]: first layer [0, 64), second [64, 192) (behind the W0s barrier)
    if (tid < C0 + C) dw_s[tid] = tid < C0 ? h2_dw(hdr0)[tid] : h2_dw(hdr2)[tid - C0];
    if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};       // padding pixels of the four planes

    // block-pixel bookkeeping: bit sub * 4 + tap of tapok = block offset ((tap >> 1) - (sub >> 1), (tap & 1) - (sub & 1)) is inside
    int spx[MT];
    unsigned tapok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        spx[mt] = 32 * mt + l31;
        const int y = spx[mt] >> 3, xx0 = spx[mt] & 7;
        unsigned m = 0;
        for (int q = 0; q < 16; ++q) {
            const int yy = y + ((q >> 1) & 1) - (q >> 3), xx = xx0 + (q & 1) - ((q >> 2) & 1);
            if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) m |= 1u << q;
        }
        tapok[mt] = m;
    }

    // scales: the image's largest |x| -> the first layer's operand scale; the bound on its outputs -> the second layer's
    const float *ximg = x + (size_t)(img_ok ? img : 0) * (CIN * 1024);
    float xm = 0.0f;
#pragma unroll
    for (int j = 0; j < CIN * 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ximg + 4 * lane + 256 * j);
        xm = fmaxf(xm, fmaxf(fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xm = fmaxf(xm, __shfl_xor(xm, o));
    const int kx0 = wave_scale_exp(img_ok ? xm : 0.0f);
    const float xs0 = __builtin_ldexpf(1.0f, kx0), d0 = __builtin_ldexpf(1.0f, -kx0);       // (x the weight rows' 2^-kw[c] at the use)
    float bm = bias0 ? __builtin_fabsf(bias0[lane]) : 0.0f;                                 // C0 = 64 channels
    const float bound = (__int_as_float(hdr0[1]) * xm + bm) * 1.0001f;
    const int k1 = wave_scale_exp(img_ok ? bound : 0.0f);                                     // (reduces bm over the wave)
    const float xs1 = __builtin_ldexpf(1.0f, k1), d2 = __builtin_ldexpf(1.0f, -k1);

    f32x16 Y[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[mt][nt][r] = 0.0f;
    // (built from the wave-uniform image index: with a lane-derived one hipcc wraps every gather in a waterfall loop)
    const long long img_u = (long long)blockIdx.x * 4 + wave_u;
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)(img_u < B ? img_u : 0) * (CIN * 1024)), 0,
                                                       (unsigned)(CIN * 4096), 0x00020000);
    __syncthreads();                                       // W0s

    // the four taps of chunk cc (its operand planes are in the wave's tile): stages 4 cc .. 4 cc + 3
    auto taps = [&](int cc) {
        const int sub = cc >> 1;
#pragma unroll 1
        for (int tap = 0; tap < 4; ++tap) {
            const int k = cc * 4 + tap;
            const int shift = ((tap >> 1) - (sub >> 1)) * 8 + ((tap & 1) - (sub & 1)), okbit = sub * 4 + tap;
            u32x4 X[2][MT][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int p = ((tapok[mt] >> okbit) & 1u) ? spx[mt] + shift : PX;
                    const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                    X[t][mt][0] = ap[0];
                    X[t][mt][1] = ap[PLANE];
                }
            dma_wait_sync();                               // this stage's weights are in; everyone is done with the other buffer
            if (k + 1 < NSTAGE) dma_stage(k + 1, (k + 1) & 1);
            const u32x4 *wt = Wb_all + (k & 1) * WBUF + lane;      // [nt][term][k-step] x 64 units
            u32x4 Wc0 = wt[0], Wc1 = wt[128];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int t = g >> 2, nt = g & 3;
                u32x4 Wn0 = Wc0, Wn1 = Wc1;
                if (g + 1 < 8) {
                    const u32x4 *bp = wt + ((g + 1) & 3) * 256 + ((g + 1) >> 2) * 64;
                    Wn0 = bp[0];
                    Wn1 = bp[128];
                }
                __builtin_amdgcn_sched_barrier(0);         // hipcc otherwise sinks the reads to just before their use
                prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, Y[0][nt], Y[1][nt]);
                __builtin_amdgcn_sched_barrier(0);
                Wc0 = Wn0;
                Wc1 = Wn1;
            }
        }
    };
    auto put_planes = [&](const u32x4(&T1)[MT][2], const u32x4(&T2)[MT][2]) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                As[(t * 2 + 0) * PLANE + h * HP + 32 * mt + l31] = T1[mt][t];
                As[(t * 2 + 1) * PLANE + h * HP + 32 * mt + l31] = T2[mt][t];
            }
        lds_order_wave();
    };
    // first layer on the patches XB for output channels 32 sl .. +31: + bias, ReLU, -> the second layer's operands
    auto first_layer = [&](int sl, const u32x4(&XB)[MT][CIN][2], u32x4(&T1)[MT][2], u32x4(&T2)[MT][2]) {
        f32x16 acc0[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[mt][r] = 0.0f;
        const u32x4 *wp = W0s + (sl * CIN) * 128 + lane;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
            prod3x2t(XB[0][ci][0], XB[0][ci][1], XB[1][ci][0], XB[1][ci][1], wp[ci * 128], wp[ci * 128 + 64], acc0[0], acc0[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias0) bv = *reinterpret_cast<const f32x4 *>(bias0 + sl * 32 + 8 * g + 4 * h);
            const f32x4 dv = h2_dw4(dw_s, sl * 32, g, h, d0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    float unused = 0.0f;
                    SCALE2_BIAS_RELU2(acc0[mt][4 * g + q], acc0[mt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], unused);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc_to_ksteps(acc0[mt], xs1, T1[mt], T2[mt]);
    };

#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const int sy = s >> 1, sx = s & 1;
        // patches of the 64 block pixels at sub-position s: lane half h holds rows ky = 2h, 2h + 1 (4 columns each) per channel
        u32x4 XB[MT][CIN][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int by = spx[mt] >> 3, bx = spx[mt] & 7;
            const int col0 = 4 * bx + 2 * sx - 1;
            const int adj = col0 < 0 ? 1 : (col0 + 3 > 31 ? -1 : 0);       // edge lanes load one column off and shift
            f32x4 pv[CIN][2];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int row = 4 * by + 2 * sy - 1 + 2 * h + rr;
                    const unsigned off = (row >= 0 && row < 32) ? (unsigned)(((ci * 32 + row) * 32 + col0 + adj) * 4) : kOobOffset;
                    pv[ci][rr] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
                }
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const f32x4 v = pv[ci][rr];
                    f32x4 o;
                    o.x = adj > 0 ? 0.0f : (adj < 0 ? v.y : v.x);
                    o.y = adj > 0 ? v.x : (adj < 0 ? v.z : v.y);
                    o.z = adj > 0 ? v.y : (adj < 0 ? v.w : v.z);
                    o.w = adj > 0 ? v.z : (adj < 0 ? 0.0f : v.w);
                    pv[ci][rr] = o;
                }
                split8_h(pv[ci][0], pv[ci][1], xs0, XB[mt][ci][0], XB[mt][ci][1]);
            }
        }
        u32x4 T1[MT][2], T2[MT][2], U1[MT][2], U2[MT][2];
        first_layer(0, XB, T1, T2);
        put_planes(T1, T2);
        first_layer(1, XB, U1, U2);
        taps(2 * s);
        put_planes(U1, U2);
        taps(2 * s + 1);
    }

    // bias + ReLU (encoder.py:32-34), the image's maximum for the next layer, whole-line stores
    float ymax = 0.0f;
    const long long wbase = img * PX;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (bias2) bv = *reinterpret_cast<const f32x4 *>(bias2 + nt * 32 + 8 * g + 4 * h);
                const f32x4 dv = h2_dw4(dw_s + C0, nt * 32, g, h, d2);
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    v[4 * g + q] = Y[mt][nt][4 * g + q];
                    v[4 * g + q + 1] = Y[mt][nt][4 * g + q + 1];
                    SCALE2_BIAS_RELU2(v[4 * g + q], v[4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], ymax);
                }
            }
            if (img_ok) {
                float *dst = out + (wbase + mt * 32) * C + nt * 32;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4 *>(Hs + l31 * 36 + 8 * g + 4 * h) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                lds_order_wave();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p = (lane >> 3) + 8 * k;
                    *reinterpret_cast<f32x4 *>(dst + (size_t)p * C + 4 * (lane & 7)) = *reinterpret_cast<const f32x4 *>(Hs + p * 36 + 4 * (lane & 7));
                }
            }
        }
    if (out_amax && img_ok) publish_amax_exclusive(out_amax, img, ymax, lane);
    (void)C0;
}

// Decoder tail in one launch (models/decoder.py:31-35): ConvTranspose2d(128 -> 64, 4x4 s2 p1) + ReLU +
// ConvTranspose2d(64 -> 3, 4x4 s2 p1) on 8x8 maps -> 32x32 NCHW images; the 16x16x64 map between them (64 KiB per image)
// never exists.  One wave owns one image and walks the four output phases (py, px) of the first layer, two per pass (the
// phases (py, 0) and (py, 1) share the parked input planes); everything is computed transposed as in conv_res_pair8_h2_kernel.
//   layer 1, phase (py, px): u[2y + py][2x + px][c] = relu(b + sum over 2x2 taps and 128 channels), as in
//       conv_tile8_bf3_kernel (4 chunks of 32 channels parked as fp16 planes, tap = shifted plane read), weights streamed
//       by LDS-DMA two taps per stage; the accumulator (lane = block pixel, registers = channels) becomes the second
//       layer's B operands by half-wave swaps (acc_to_ksteps), scaled by the phase tile's own maximum;
//   layer 2 in its GEMM + col2im form: T[co * 16 + tap][pixel] = sum_c w4[c][co][tap] u[pixel][c] (48 rows = two A tiles),
//       and out[co][4y + 2py - 1 + ky][4x + 2px - 1 + kx] += T: every output element receives exactly ONE term per phase.
//       The two phases of a pass are combined in REGISTERS: a lane holds its pixel's terms of both phases for the output
//       rows 4y + 2py - 1 + h and + 2; its own 16-byte quad of such a row (columns 4x .. 4x + 3) is its six inner terms plus one
//       term each of its left and right pixel, fetched by DPP row shifts -- eight lanes then store one whole 128-byte output
//       row straight from registers: out = bias + quad in pass 0, out += quad in pass 1 with 16-byte read-modify-writes of
//       the wave's own 12 KiB image (L2-resident; plain accesses by the wave that wrote them, its write-through L1 keeps no
//       stale copy, and a pass's stores are complete -- sixteen s_waitcnt vmcnt(0) later -- before the next pass's loads are
//       issued): a fixed summation order, no atomics, no accumulation tile.  (Round 2 scattered the terms into a zeroed
//       wave-private 34 x 42 LDS tile per channel and read it back: 420 LDS operations per image, 28 % of a wave's time;
//       scattered 4-byte read-modify-writes straight from the accumulator layout were measured first: 570 us instead of 330
//       for the two separate kernels -- L2 request bound.)
struct TailGeom {
    unsigned long long dym[4], dxm[4];             // 4 bits per tap: dy + 8, dx + 8 (ConvGeom) of each phase
};

#ifndef DT_MINW
#define DT_MINW 2
#endif
__global__ __launch_bounds__(256, DT_MINW) void dec_tail8_h2_kernel(const float *__restrict__ in, const u32x4 *__restrict__ w2img,
                                                              const int *__restrict__ hdr2, const float *__restrict__ bias2,
                                                              TailGeom tg, const u32x4 *__restrict__ w4img,
                                                              const int *__restrict__ hdr4, const float *__restrict__ bias4,
                                                              float *__restrict__ out, int B, const int *__restrict__ in_amax) {
    constexpr int NT = 2, MT = 2, PX = 64, HP = PX + 1, PLANE = HP * 2, CIN = 128, CPT = CIN / 32, CO = 3;
    constexpr int TILE4 = 2 * 2 * PLANE;                   // [k-step 2][term 2][half 2][pixel + zero] = 520 units
    constexpr int WBUF = 16 * 64, NSTAGE = 34;             // per pass: 4 chunks x 4 tap pairs (16 pieces each) + the second layer's image
    __shared__ u32x4 As_all[4 * TILE4];
    __shared__ u32x4 Wb_all[2 * WBUF];
    __shared__ __attribute__((aligned(16))) float dw_s[64];      // the first layer's per-output-channel weight scales 2^-kw[c]
    if (threadIdx.x < 64) dw_s[threadIdx.x] = h2_dw(hdr2)[threadIdx.x];      // (stage barriers precede every use)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    u32x4 *As = As_all + wave * TILE4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const long long img = (long long)blockIdx.x * 4 + wave;
    const bool img_ok = img < B;

    // (scalar source base + this lane's constant byte offset: no vector instruction and no address register per piece)
    const unsigned dma_lane = (unsigned)lane * 16u;
    auto dma = [&](const u32x4 *src_uniform, u32x4 *dst_piece) {
        const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)dst_piece);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(dma_lane), "s"(src_uniform), "s"(lds) : "memory");
    };
    // A PASS covers the two phases (py, 0) and (py, 1): they share the parked planes, and where their taps read the same
    // input offset (dx = 0) also the operand reads.  stage k = 17 py + i: i < 16: chunk i >> 2, tap pair i & 3 = (ty, kind) of
    // the first layer -- kind 0: the two phases' dx = 0 taps (tx = 0 of px = 0, tx = 1 of px = 1), kind 1: the other two;
    // pieces 0..7 = phase (py, 0)'s tap, 8..15 = phase (py, 1)'s ([nt][term][k-step] each); i = 16: the second layer's A image
    auto dma_stage = [&](int k, int buf) {
        const int py = k / 17, i = k - 17 * py;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave_u * 4 + j, half = p >> 3;
            const int ty = (i >> 1) & 1, kind = i & 1, tap = ty * 2 + (half ? 1 - kind : kind);
            const u32x4 *src = i == 16 ? w4img + p * 64
                                       : w2img + (size_t)((2 * py + half) * 16 + tap * CPT + (i >> 2)) * 512 + (p & 7) * 64;
            dma(src, Wb_all + buf * WBUF + p * 64);
        }
    };
    auto dma_wait_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    dma_stage(0, 0);
    if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};       // padding pixels of the four planes

    // (built from the wave-uniform image index: with a lane-derived one hipcc wraps every access in a waterfall loop)
    const long long img_u = (long long)blockIdx.x * 4 + wave_u;
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(img_u < B ? img_u : 0) * (CO * 1024), 0, img_u < B ? (unsigned)(CO * 4096) : 0u, 0x00020000);
    int spx[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) spx[mt] = 32 * mt + l31;
    const float *src = in + ((size_t)(img_ok ? img : 0) * PX + lane) * CIN;                 // this lane's pixel row
    f32x4 raw[8];
    auto load_raw = [&](int cc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const f32x4 *>(src + 32 * cc + 4 * j);
    };
    float m = 0.0f;
    const int given = (in_amax && img_ok) ? in_amax[img] : -1;
    if (given >= 0) m = __int_as_float(given);
    else for (int cc = 0; cc < CPT; ++cc) {
        load_raw(cc);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(raw[j].x), __builtin_fabsf(raw[j].y)), fmaxf(__builtin_fabsf(raw[j].z), __builtin_fabsf(raw[j].w))));
    }
    const int kx = wave_scale_exp(img_ok ? m : 0.0f);
    const float xs = __builtin_ldexpf(1.0f, kx), d1 = __builtin_ldexpf(1.0f, -kx);      // (x the weight rows' 2^-kw[c] at the use)
    load_raw(0);

#pragma unroll 1
    for (int py = 0; py < 2; ++py) {
        unsigned long long dym[2], dxm[2];
        unsigned tapok[2][MT];
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            dym[px] = tg.dym[2 * py + px];
            dxm[px] = tg.dxm[2 * py + px];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int y = spx[mt] >> 3, x = spx[mt] & 7;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int yy = y + (int)((dym[px] >> (4 * t)) & 15) - 8, xx = x + (int)((dxm[px] >> (4 * t)) & 15) - 8;
                    if (yy >= 0 && yy < 8 && xx >= 0 && xx < 8) mk |= 1u << t;
                }
                tapok[px][mt] = mk;
            }
        }
        f32x16 acc[2][MT][NT];                              // [px][pixel tile][channel tile]
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[px][mt][nt][r] = 0.0f;
        // ----------------------------- layer 1, both phases of the pass -----------------------------
#pragma unroll 1
        for (int cc = 0; cc < CPT; ++cc) {
            __builtin_amdgcn_wave_barrier();
            u32x4 *dst = As + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    u32x4 t1, t2;
                    split8_h(raw[4 * hh + 2 * t], raw[4 * hh + 2 * t + 1], xs, t1, t2);
                    dst[(t * 2 + 0) * PLANE + hh * HP] = t1;
                    dst[(t * 2 + 1) * PLANE + hh * HP] = t2;
                }
            lds_order_wave();
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int k = py * 17 + cc * 4 + i;
                const int ty = i >> 1, kind = i & 1;
                const int tapA = ty * 2 + kind, tapB = ty * 2 + 1 - kind;     // of phase px = 0 / px = 1
                u32x4 X[2][MT][2];                          // [k-step][pixel tile][term] of the half-stage in flight
                // k-step t of tap `tap` of phase px (a k-step's registers are reloaded for the second half as soon as the first
                // half's groups that read them have been issued)
                auto ldX = [&](int t, int px, int tap) {
                    const int shift = ((int)((dym[px] >> (4 * tap)) & 15) - 8) * 8 + ((int)((dxm[px] >> (4 * tap)) & 15) - 8);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int p = ((tapok[px][mt] >> tap) & 1u) ? spx[mt] + shift : PX;
                        const u32x4 *ap = As + (t * 2) * PLANE + h * HP + p;
                        X[t][mt][0] = ap[0];
                        X[t][mt][1] = ap[PLANE];
                    }
                };
                ldX(0, 0, tapA);
                ldX(1, 0, tapA);
                // this stage's weights are in; everyone is done with the other buffer.  (The next chunk's eight activation loads
                // go out behind stage i = 1's weights and may stay in flight across its wait: see conv_res_pair8_h2_kernel.)
                if (i == 1) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __syncthreads();
                } else dma_wait_sync();
                dma_stage(k + 1, (k + 1) & 1);
                if (i == 0) load_raw(cc + 1 < CPT ? cc + 1 : 0);       // (the next pass starts over at chunk 0)
                const u32x4 *wt = Wb_all + (k & 1) * WBUF + lane;      // [phase of the pair][nt][term][k-step] x 64 units
                u32x4 Wc0 = wt[0], Wc1 = wt[128];
#pragma unroll
                for (int g = 0; g < 8; ++g) {              // group g = (phase of the pair, k-step, nt)
                    const int px = g >> 2, t = (g >> 1) & 1, nt = g & 1;
                    u32x4 Wn0 = Wc0, Wn1 = Wc1;
                    if (g + 1 < 8) {
                        const int g1 = g + 1;
                        const u32x4 *bp = wt + (g1 >> 2) * 512 + (g1 & 1) * 256 + ((g1 >> 1) & 1) * 64;
                        Wn0 = bp[0];
                        Wn1 = bp[128];
                    }
                    if (g == 2) ldX(0, 1, tapB);           // groups 0, 1 (the readers of k-step 0) are behind us
                    if (g == 4) ldX(1, 1, tapB);           // groups 2, 3 likewise
                    __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise sinks the reads to just before their use
                    prod3x2t(X[t][0][0], X[t][0][1], X[t][1][0], X[t][1][1], Wc0, Wc1, acc[px][0][nt], acc[px][1][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    Wc0 = Wn0;
                    Wc1 = Wn1;
                }
            }
        }
        // ----------------------------- per phase: bias + ReLU, its scale, T = W4 u -----------------------------
        // row rho = 32 m + (r & 3) + 8 (r >> 2) + 4 h of T is (co = rho >> 4, tap = rho & 15 = ky * 4 + kx): register r of A tile m
        // holds co = 2 m + (r >> 3), ky = h + 2 ((r >> 2) & 1), kx = r & 3; rows >= 48 (m = 1, r >= 8) are padding
        f32x16 T[2][2][MT];                                 // [px][A tile][pixel tile]
        float d4[2];
        const u32x4 *wt4 = nullptr;
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            float um = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 bv = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (bias2) bv = *reinterpret_cast<const f32x4 *>(bias2 + nt * 32 + 8 * g + 4 * h);
                    const f32x4 dv = h2_dw4(dw_s, nt * 32, g, h, d1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int q = 0; q < 4; q += 2)
                            SCALE2_BIAS_RELU2(acc[px][mt][nt][4 * g + q], acc[px][mt][nt][4 * g + q + 1], dv[q], dv[q + 1], bv[q], bv[q + 1], um);
                }
            const int ku = wave_scale_exp(img_ok ? um : 0.0f);
            const float us = __builtin_ldexpf(1.0f, ku);
            d4[px] = __builtin_ldexpf(1.0f, -ku);
            u32x4 U1[MT][NT][2], U2[MT][NT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc_to_ksteps(acc[px][mt][nt], us, U1[mt][nt], U2[mt][nt]);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) T[px][m2][mt][r] = 0.0f;
            if (px == 0) {
                const int k = py * 17 + 16;
                dma_wait_sync();
                if (k + 1 < NSTAGE) dma_stage(k + 1, (k + 1) & 1);
                wt4 = Wb_all + (k & 1) * WBUF + lane;       // [m][k-step 4][term] x 64 units
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const u32x4 *bp = wt4 + (m2 * 4 + kk) * 128;
                    prod3x2t(U1[0][kk >> 1][kk & 1], U2[0][kk >> 1][kk & 1], U1[1][kk >> 1][kk & 1], U2[1][kk >> 1][kk & 1], bp[0], bp[64],
                             T[px][m2][0], T[px][m2][1]);
                }
        }
        // ----------------------------- col2im of the pass, in registers -----------------------------
        // Lane (pixel (y, x), half h) holds for every output channel the eight terms of its pixel's two phases for the two
        // output rows oy = 4y + 2py - 1 + h (kernel row ky = h) and oy + 2 (ky = h + 2): a[kx] of phase (py, 0) lands at column
        // 4x - 1 + kx, b[kx] of phase (py, 1) at 4x + 1 + kx.  The lane's own 16-byte quad of a row, columns 4x .. 4x + 3, is
        //     { a1 + b3 of the LEFT pixel,  a2 + b0,  a3 + b1,  b2 + a0 of the RIGHT pixel }
        // -- the two neighbour terms come by DPP row shifts inside the 8-lane pixel row (nothing at the image's left / right
        // edge: those taps fall outside) -- so the eight lanes of a pixel row write one whole 128-byte output row straight
        // from registers: pass 0 stores bias + quad, pass 1 adds to what pass 0 stored (row 31 gets its only term in pass 1).
        // Same terms in the same order as the LDS-tile form this replaces (phase 0's term first, pass 0 first): same bits; no
        // LDS tile to zero, scatter into and read back (420 LDS operations per image), 28 % of a wave's time before.
        const bool xl = (lane & 7) != 0, xr = (lane & 7) != 7;
        unsigned roff[MT][2];                              // byte offset of the lane's quad in rows oy / oy + 2 of channel 0, or out of range
        bool only1[MT][2];                                 // pass 1: the row got nothing in pass 0 (row 31)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int row = 4 * (spx[mt] >> 3) + 2 * py - 1 + h + 2 * g;
                roff[mt][g] = (row >= 0 && row < 32) ? (unsigned)((row * 32 + 4 * (lane & 7)) * 4) : kOobOffset;
                only1[mt][g] = row == 31;
            }
        // pass 1 reads what pass 0 stored: one channel's four quads a channel ahead (round 6: all three channels' twelve quads at once
        // were 48 registers on top of T's 96 and the next pass's parked activations -- most of the kernel's 56 spilled registers)
        f32x4 ov[2][MT][2];
        auto ov_load = [&](int co, f32x4(&o)[MT][2]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    o[mt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, roff[mt][g], (unsigned)co * 4096u, 0));
        };
        if (py > 0) ov_load(0, ov[0]);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            if (py > 0 && co + 1 < CO) ov_load(co + 1, ov[(co + 1) & 1]);
            const float bv = bias4 ? bias4[co] : 0.0f;
            const float w4d = h2_dw(hdr4)[co];                  // the output channel's own weight scale 2^-kw4[co] (wave-uniform)
            const float d40 = d4[0] * w4d, d41 = d4[1] * w4d;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x16 &T0 = T[0][co >> 1][mt], &T1 = T[1][co >> 1][mt];
                    const int r0 = 8 * (co & 1) + 4 * g;
                    const float a0 = T0[r0] * d40, b0 = T1[r0] * d41, b1 = T1[r0 + 1] * d41, b2 = T1[r0 + 2] * d41, b3 = T1[r0 + 3] * d41;
                    // neighbours: row_shr:1 hands lane i the value of lane i - 1, row_shl:1 that of lane i + 1 (16-lane rows = two pixel rows)
                    float lb3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b3), 0x111, 0xf, 0xf, true));
                    float ra0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x101, 0xf, 0xf, true));
                    lb3 = xl ? lb3 : 0.0f;
                    ra0 = xr ? ra0 : 0.0f;
                    f32x4 e;
                    e.x = __builtin_fmaf(T0[r0 + 1], d40, lb3);            // (the products by 2^-k are exact: one rounding, as mul + add)
                    e.y = __builtin_fmaf(T0[r0 + 2], d40, b0);
                    e.z = __builtin_fmaf(T0[r0 + 3], d40, b1);
                    e.w = ra0 + b2;
                    f32x4 base = {bv, bv, bv, bv};
                    if (py > 0 && !only1[mt][g]) base = ov[co & 1][mt][g];
                    const f32x4 v = base + e;
                    // (the channel's offset in the VECTOR offset: a scalar-offset store followed by an overwrite of its data registers is the
                    // hazard hipcc leaves unguarded, tools/hazard_scan.py)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors, roff[mt][g] == kOobOffset ? kOobOffset : roff[mt][g] + (unsigned)co * 4096u, 0, 0);
                }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 8) As[(lane >> 1) * PLANE + (lane & 1) * HP + PX] = u32x4{0, 0, 0, 0};   // the planes' padding pixels again
    }
}

}  // namespace vqvae

using namespace vqvae;

// A 3x3 conv / conv-transpose (stride 1, Cin -> 128, + bias + ReLU) and the two residual layers behind it in one launch
// (conv_res_pair8_h2_kernel); post as in res_pair_forward_impl.  x == y is not allowed (x has Cin channels).
bool vqvae::conv_res_pair_supported(int kind, int H, int W, int Cin, int C, int Rh) {
    return (kind == VQVAE_CONV_3x3_S1 || kind == VQVAE_CONVT_3x3_S1) && H == 8 && W == 8 && C == 128 && Cin >= 32 && Cin % 32 == 0 &&
           Cin <= 256 && Rh >= 1 && Rh <= 32;
}

int vqvae::conv_res_pair_forward_impl(int kind, const float *x, const float *packed_front, const float *bias_front, int Cin,
                                      const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C, int Rh,
                                      int flags, float *y, hipStream_t st, const int *in_amax, int *out_amax,
                                      const ResPairPost *post, const int64_t *gather_idx, int gather_K) {
    if (gather_idx && (post || gather_K < 1)) return VQVAE_ERR_UNSUPPORTED;      // (the gather rides in the instance without a post conv)
    if (!x || !packed_front || !packed_w1 || !packed_w2 || (!y && !post)) return VQVAE_ERR_NULL;
    if (B < 1 || !conv_res_pair_supported(kind, H, W, Cin, C, Rh)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(post ? post->out : nullptr)) & 15)
        return VQVAE_ERR_UNSUPPORTED;
    ConvGeom g;
    if (make_geom(kind, B, H, W, Cin, C, 0, g) != VQVAE_OK || g.nphase != 1 || g.ntaps != 9) return VQVAE_ERR_UNSUPPORTED;
    const char *hf = reinterpret_cast<const char *>(packed_front) + packed_h2_offset(g, kind);
    FrontConv fc;
    fc.wimg = reinterpret_cast<const u32x4 *>(hf + h2_header_bytes(g.ntile));
    fc.hdr = reinterpret_cast<const int *>(hf);
    fc.bias = bias_front;
    fc.dym = g.dymask[0];
    fc.dxm = g.dxmask[0];
    fc.Cin = Cin;
    const int cpt = C / 32;
    const char *h1 = reinterpret_cast<const char *>(packed_w1) + (size_t)9 * cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const char *h2 = reinterpret_cast<const char *>(packed_w2) + (size_t)cpt * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    const u32x4 *w1h = reinterpret_cast<const u32x4 *>(h1 + h2_header_bytes(1)), *w2h = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes((C + 31) / 32));
    const int *hd1 = reinterpret_cast<const int *>(h1), *hd2 = reinterpret_cast<const int *>(h2);
    const unsigned gtc = (unsigned)((B + CRP_NW - 1) / CRP_NW);
    // the checks that can refuse come BEFORE prof_begin: an early return behind it would leave an unmatched begin event
    ConvGeom g3;
    if (post && (!post->packed || !post->out || !res_pair_post_supported(C, post->Cout) ||
                 make_geom(VQVAE_CONV_1x1, 1, 8, 8, C, post->Cout, 0, g3) != VQVAE_OK)) return VQVAE_ERR_UNSUPPORTED;
    if (post && post->vq && (post->Cout != 64 || CRP_NW != 4 || post->vq->K32 % 128 || post->vq->K32 > 1024 || !post->vq->partials))
        return VQVAE_ERR_UNSUPPORTED;
    prof_begin(VQVAE_PROF_RES_LAYER, st);
    if (post) {
        const char *h3 = reinterpret_cast<const char *>(post->packed) + packed_h2_offset(g3, VQVAE_CONV_1x1);
        const u32x4 *w3h = reinterpret_cast<const u32x4 *>(h3 + h2_header_bytes(g3.ntile));
        const int *hd3 = reinterpret_cast<const int *>(h3);
#define CRP_POST(NT3_)                                                                                                          \
    hipLaunchKernelGGL((conv_res_pair8_h2_kernel<NT3_>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags, hd1, hd2, \
                       in_amax, out_amax, w3h, hd3, post->bias, post->out, post->zero, post->zero_n, VqFuse{})
        if (post->vq && post->debug_ze) {
            hipLaunchKernelGGL((conv_res_pair8_h2_kernel<2, true, false, true>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags,
                               hd1, hd2, in_amax, out_amax, w3h, hd3, post->bias, post->out, post->zero, post->zero_n, *post->vq);
        } else if (post->vq) {
            // the quantizer rides behind the 1x1 conv: z_e is never written (post->out unused)
            hipLaunchKernelGGL((conv_res_pair8_h2_kernel<2, true>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags,
                               hd1, hd2, in_amax, out_amax, w3h, hd3, post->bias, post->out, post->zero, post->zero_n, *post->vq);
        } else
        switch (post->Cout / 32) {
            case 1: CRP_POST(1); break;
            case 2: CRP_POST(2); break;
            case 4: CRP_POST(4); break;
        }
#undef CRP_POST
    } else {
        if (gather_idx) {
            VqFuse gf{};
            gf.idx = reinterpret_cast<long long *>(const_cast<int64_t *>(gather_idx));       // (read only; x = the codebook)
            gf.K = gather_K;
            hipLaunchKernelGGL((conv_res_pair8_h2_kernel<0, false, true>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags,
                               hd1, hd2, in_amax, out_amax, nullptr, nullptr, nullptr, nullptr, nullptr, 0, gf);
        } else
        hipLaunchKernelGGL((conv_res_pair8_h2_kernel<0>), dim3(gtc), dim3(CRP_NW * 64), 0, st, x, fc, w1h, w2h, y, (int)B, flags, hd1, hd2,
                           in_amax, out_amax, nullptr, nullptr, nullptr, nullptr, nullptr, 0, VqFuse{});
    }
    prof_end(VQVAE_PROF_RES_LAYER, st);
    return (int)hipGetLastError();
}

// The encoder's first two layers in one launch (enc_front8_h2_kernel): 32x32 images, 3 input channels, 64 -> 128 channels.
bool vqvae::enc_front_supported(int H, int W, int Cin, int C1, int C2) { return H == 32 && W == 32 && Cin == 3 && C1 == 64 && C2 == 128; }

int vqvae::enc_front_forward_impl(const float *x_nchw, const float *packed_in, const float *bias_in, const float *packed2,
                                  const float *bias2, int64_t B, int H, int W, int Cin, int C1, int C2, float *y, hipStream_t st,
                                  int *out_amax, int *zero_buf, int zero_n) {
    if (!x_nchw || !packed_in || !packed2 || !y) return VQVAE_ERR_NULL;
    if (B < 1 || !enc_front_supported(H, W, Cin, C1, C2)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_nchw) | reinterpret_cast<uintptr_t>(y)) & 15) return VQVAE_ERR_UNSUPPORTED;
    const int ntile0 = (C1 + 31) / 32;
    const char *h0 = reinterpret_cast<const char *>(packed_in) +
                     (size_t)ntile0 * ((size_t)((Cin * 8 + 3) / 4) * 256 + (size_t)Cin * 768) * sizeof(float);
    ConvGeom g;
    if (make_geom(VQVAE_CONV_4x4_S2, B, H / 2, W / 2, C1, C2, 0, g) != VQVAE_OK) return VQVAE_ERR_UNSUPPORTED;
    const char *h2 = reinterpret_cast<const char *>(packed2) + packed_h2_offset(g, VQVAE_CONV_4x4_S2);
    const u32x4 *w2s2d = reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes(g.ntile) + packed_h2_bytes(g));     // space-to-depth chunk order
    prof_begin(VQVAE_PROF_CONV_IGEMM, st);
    hipLaunchKernelGGL((enc_front8_h2_kernel<3>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, x_nchw,
                       reinterpret_cast<const u32x4 *>(h0 + h2_header_bytes(ntile0)), reinterpret_cast<const int *>(h0), bias_in, w2s2d,
                       reinterpret_cast<const int *>(h2), bias2, y, (int)B, out_amax, zero_buf, zero_n);
    prof_end(VQVAE_PROF_CONV_IGEMM, st);
    return (int)hipGetLastError();
}

// The decoder's last two layers in one launch (dec_tail8_h2_kernel): 8x8 maps, 128 -> 64 -> 3 channels.
bool vqvae::dec_tail_supported(int h4, int w4, int C, int C1, int Cout) { return h4 == 8 && w4 == 8 && C == 128 && C1 == 64 && Cout == 3; }

int vqvae::dec_tail_forward_impl(const float *x, const float *packed2, const float *bias2, const float *packed4, const float *bias4,
                                 int64_t B, int h4, int w4, int C, int C1, int Cout, float *y_nchw, hipStream_t st, const int *in_amax) {
    if (!x || !packed2 || !packed4 || !y_nchw) return VQVAE_ERR_NULL;
    if (B < 1 || !dec_tail_supported(h4, w4, C, C1, Cout)) return VQVAE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y_nchw)) & 15) return VQVAE_ERR_UNSUPPORTED;
    ConvGeom g;
    if (make_geom(VQVAE_CONVT_4x4_S2, B, h4, w4, C, C1, 0, g) != VQVAE_OK) return VQVAE_ERR_UNSUPPORTED;
    const char *h2 = reinterpret_cast<const char *>(packed2) + packed_h2_offset(g, VQVAE_CONVT_4x4_S2);
    TailGeom tg;
    for (int ph = 0; ph < 4; ++ph) { tg.dym[ph] = g.dymask[ph]; tg.dxm[ph] = g.dxmask[ph]; }
    const int ntile4 = (16 * Cout + 31) / 32, cpt4 = (C1 + 31) / 32;
    const size_t cells = (size_t)cpt4 * ntile4;
    const char *h4p = reinterpret_cast<const char *>(packed4) + cells * (1024 * sizeof(float) + 3072 * sizeof(unsigned short));
    prof_begin(VQVAE_PROF_CONV_OUT, st);
    hipLaunchKernelGGL(dec_tail8_h2_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, x,
                       reinterpret_cast<const u32x4 *>(h2 + h2_header_bytes(g.ntile)), reinterpret_cast<const int *>(h2), bias2, tg,
                       reinterpret_cast<const u32x4 *>(h4p + h2_header_bytes(1) + cells * 2048 * sizeof(unsigned short)),
                       reinterpret_cast<const int *>(h4p), bias4, y_nchw, (int)B, in_amax);
    prof_end(VQVAE_PROF_CONV_OUT, st);
    return (int)hipGetLastError();
}
