// Weight / bias gradients and ReLU masks for the conv layers of the path, gfx950.
// The DATA gradients need no kernels of their own: the gradient of a conv w.r.t. its input is the transposed
// conv with the same weight tensor and vice versa, i.e. the forward kernels of conv.hip / conv_ends.hip with the other `kind`
// (vqvae_amd/autograd_conv.py does that mapping).
//
//   vqvae_conv_wgrad_f32   dW[ca][cb][ky][kx] = sum over pixels of  A[pixel][ca] * Bt[pixel*s + (ky,kx) - p][cb]
//        nn.Conv2d:           A = grad_y (B,Ho,Wo,Cout), Bt = x      -> dW in the (Cout,Cin,kh,kw) layout
//        nn.ConvTranspose2d:  A = x (B,H,W,Cin),         Bt = grad_y -> dW in the (Cin,Cout,kh,kw) layout
//      (the same index formula serves both: y_Bt = y_A * stride + ky - pad).  Exact fp32 products on the fp32
//      matrix cores (v_mfma_f32_32x32x2_f32: the reduction index is the PIXEL, two pixels per MFMA step, and
//      with row-major activations both operands are plain coalesced 128-byte reads -- no transposition);
//      the pixel (or image) range is split over workgroups, partial sums are combined in a fixed order
//      (no floating-point atomics: gradients are bit-reproducible run to run).  Three kernels: 8x8 maps with both
//      maps of an image resident in LDS and all taps from one staging (conv_wgrad_map8_kernel: the path's layers at
//      32x32 images, 80-90 % of the fp32 matrix peak); any map, per tap (conv_wgrad_kernel); image operand with the
//      taps folded into the MFMA N dimension (conv_wgrad_img_kernel: first / last layer).
//   vqvae_bias_grad_f32    db[c] = sum over pixels of grad_y[pixel][c]   (fixed-order two-stage reduction, fp64)
//   vqvae_relu_backward_f32  g_in = g_out * (y > 0)
#include "common.h"

namespace vqvae {

constexpr int kWgMaxSplit = 64;       // pixel-range splits across workgroups
constexpr int kWgMapSplit = 512;      // workgroups of the map-resident kernel (two per CU): image ranges = 512 / (ca, cb) tiles
constexpr int kWgImgSplit = 512;      // workgroups (= partials) of the image-operand kernel

struct WgradGeom {
    int B, HA, WA, CA, HB, WB, CB;
    int k, stride, pad;
    int bt_nchw;                      // Bt is an NCHW image tensor (first / last layer), else row-major
    int nsplit;
    long long rows_per_split;         // 32-pixel blocks of A per workgroup
};

// Workgroup = 4 waves; output tile = 64 ca x 64 cb for one tap.  The reduction runs over blocks of 32 A-pixels:
// the block's A rows and the tap-shifted Bt rows (zero outside the map) are staged in LDS with coalesced 16-byte
// loads (double-buffered, one barrier per block); wave w multiplies pixels [8w, 8w+8) of the block -- four MFMA
// k-steps whose operands are plain ds_read_b32 of one channel per lane, so the fp32 matrix pipe (which shares its
// issue slots with the VALU) sees almost no address arithmetic.  The four waves' tiles are then added in wave
// order through LDS and wave 0 writes the split's partial.
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                            float *__restrict__ partial, WgradGeom g) {
    constexpr int MT = 2, NT = 2, PB = 32, LD = 68;     // LD: padded row (64 channels + 4) -> conflict-free reads
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * PB * LD];     // [buf][A | B][pixel][LD]; reused as `red`
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_b = (g.CB + 63) / 64, tiles_a = (g.CA + 63) / 64;
    int t = blockIdx.x;
    const int tb = t % tiles_b; t /= tiles_b;
    const int ta = t % tiles_a; t /= tiles_a;
    const int tap = t;                                  // ky * k + kx
    const int ky = tap / g.k, kx = tap - ky * g.k;
    const int split = blockIdx.y;
    const unsigned npix = (unsigned)g.B * g.HA * g.WA, img_px = (unsigned)g.HA * g.WA;
    const unsigned nblk = (npix + PB - 1) / PB;
    const unsigned blk_lo = (unsigned)(split * g.rows_per_split);            // rows_per_split counts pixel blocks here
    unsigned blk_hi = blk_lo + (unsigned)g.rows_per_split;
    if (blk_hi > nblk) blk_hi = nblk;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // staging role of this thread: pixel pi of the block, channels [8*cg, 8*cg + 8) of the 64-channel tile
    const int pi = tid >> 3, cg = tid & 7;
    const int ca0 = ta * 64 + 8 * cg, cb0 = tb * 64 + 8 * cg;
    const bool va = (g.CA & 3) == 0, vb = !g.bt_nchw && (g.CB & 3) == 0;      // 16-byte loads allowed
    f32x4 ra[2], rb[2];
    auto fetch = [&](unsigned blk) {
        const unsigned p = blk * PB + pi;
        const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        ra[0] = ra[1] = rb[0] = rb[1] = z4;
        if (p >= npix) return;
        const unsigned b = p / img_px, rem = p - b * img_px;
        const int yA = (int)(rem / (unsigned)g.WA), xA = (int)(rem - (unsigned)yA * g.WA);
        const float *ap = A + (size_t)p * g.CA;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = ca0 + 4 * q;
            if (va && c + 3 < g.CA) ra[q] = *reinterpret_cast<const f32x4 *>(ap + c);
            else {
                if (c < g.CA) ra[q].x = ap[c];
                if (c + 1 < g.CA) ra[q].y = ap[c + 1];
                if (c + 2 < g.CA) ra[q].z = ap[c + 2];
                if (c + 3 < g.CA) ra[q].w = ap[c + 3];
            }
        }
        const int yB = yA * g.stride + ky - g.pad, xB = xA * g.stride + kx - g.pad;
        if (yB < 0 || yB >= g.HB || xB < 0 || xB >= g.WB) return;
        if (g.bt_nchw) {
            const float *bp = Bt + ((size_t)b * g.CB * g.HB + yB) * g.WB + xB;    // + c * HB * WB
            const size_t cs = (size_t)g.HB * g.WB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = cb0 + 4 * q;
                if (c < g.CB) rb[q].x = bp[c * cs];
                if (c + 1 < g.CB) rb[q].y = bp[(c + 1) * cs];
                if (c + 2 < g.CB) rb[q].z = bp[(c + 2) * cs];
                if (c + 3 < g.CB) rb[q].w = bp[(c + 3) * cs];
            }
        } else {
            const float *bp = Bt + (((size_t)b * g.HB + yB) * g.WB + xB) * g.CB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = cb0 + 4 * q;
                if (vb && c + 3 < g.CB) rb[q] = *reinterpret_cast<const f32x4 *>(bp + c);
                else {
                    if (c < g.CB) rb[q].x = bp[c];
                    if (c + 1 < g.CB) rb[q].y = bp[c + 1];
                    if (c + 2 < g.CB) rb[q].z = bp[c + 2];
                    if (c + 3 < g.CB) rb[q].w = bp[c + 3];
                }
            }
        }
    };
    auto park = [&](int buf) {
        float *as = smem + (buf * 2 + 0) * PB * LD + pi * LD + 8 * cg;
        float *bs = smem + (buf * 2 + 1) * PB * LD + pi * LD + 8 * cg;
        *reinterpret_cast<f32x4 *>(as) = ra[0]; *reinterpret_cast<f32x4 *>(as + 4) = ra[1];
        *reinterpret_cast<f32x4 *>(bs) = rb[0]; *reinterpret_cast<f32x4 *>(bs + 4) = rb[1];
    };

    if (blk_lo < blk_hi) {
        fetch(blk_lo);
        park(0);
    }
    __syncthreads();
    for (unsigned blk = blk_lo; blk < blk_hi; ++blk) {
        const int buf = (blk - blk_lo) & 1;
        if (blk + 1 < blk_hi) fetch(blk + 1);
        const float *as = smem + (buf * 2 + 0) * PB * LD + (8 * wave + h) * LD + l31;
        const float *bs = smem + (buf * 2 + 1) * PB * LD + (8 * wave + h) * LD + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // k-step q: pixels 8*wave + 2q + h
            float av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = as[2 * q * LD + 32 * mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bs[2 * q * LD + 32 * nt];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
        if (blk + 1 < blk_hi) park(buf ^ 1);                // its last readers passed the barrier of the previous block
        __syncthreads();
    }
    // the four waves' tiles are added in wave order (0 + 1 + 2 + 3) through one LDS tile; wave 0 writes
    float *red = smem;                                       // 4096 floats <= 2*2*32*68
    for (int w = 3; w >= 1; --w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((mt * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] += red[((mt * NT + nt) * 16 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        float *dst = partial + ((size_t)split * g.k * g.k + tap) * g.CA * g.CB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int a = ta * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;     // accumulator row = ca
                    const int c = tb * 64 + nt * 32 + l31;
                    if (a < g.CA && c < g.CB) dst[(size_t)a * g.CB + c] = acc[mt][nt][r];
                }
    }
}

// Weight gradient with both maps resident in LDS, for 8x8 A maps (the body of the path at 32x32 images: every conv,
// conv-transpose and residual layer between the first and the last one).  The kernel above reads A once per tap and
// per 64 columns of Bt (16 flops per byte: it runs at the L2's pace, not the matrix pipe's); here a workgroup
// parks one image's A tile (64 pixels x 32 WA channels) and the zero-framed Bt tile (PH x PH pixels x 32 WB channels)
// in LDS -- LDS-DMA, 1 KiB pieces of 8 pixels x 32 channels, the frame is zeroed once and never written again --
// and every tap reads its operand from there at a constant offset: one ds_read_b32 per 64-cycle MFMA, no address
// arithmetic.  A wave owns a 32 x 32 tile of (ca, cb) for NTW taps (K = 4: the 16 taps are split over two waves),
// so its accumulators are final for its split: no reduction across waves.  Two workgroups per CU: one loads while the
// other multiplies.  Partials [split][tap][ca][cb] as above, combined in a fixed order by conv_wgrad_reduce_kernel.
template <int K, int S, int WA, int WB>
__global__ __launch_bounds__(256, 2) void conv_wgrad_map8_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                 float *__restrict__ partial, WgradGeom g, int imgs_per_split) {
    constexpr int TG = 4 / (WA * WB), NTW = K * K / TG, PH = 7 * S + K, HB = 8 * S, PAD = K == 1 ? 0 : 1;
    static_assert(WA * WB * TG == 4 && NTW * TG == K * K && NTW % K == 0, "wave layout");
    constexpr int ASZ = WA * 64 * 32, BSZ = WB * PH * PH * 32;                  // floats
    constexpr int NPA = WA * 8, NPB = WB * HB * S, NPIECE = NPA + NPB;         // 1 KiB pieces per image
    __shared__ __attribute__((aligned(128))) float smem[ASZ + BSZ];
    float *As = smem, *Bs = smem + ASZ;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qa = wave_u % WA, qb = (wave_u / WA) % WB, tg = wave_u / (WA * WB);
    const int tiles_b = g.CB / (32 * WB);
    const int tb = blockIdx.x % tiles_b, ta = blockIdx.x / tiles_b;
    const int ca0 = ta * 32 * WA, cb0 = tb * 32 * WB;
    const int split = blockIdx.y;
    const long long b_lo = (long long)split * imgs_per_split;
    long long b_hi = b_lo + imgs_per_split;
    if (b_hi > g.B) b_hi = g.B;

    for (int i = tid; i < BSZ / 4; i += 256) reinterpret_cast<f32x4 *>(Bs)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // a piece = 8 pixels x 32 channels: lane -> pixel lane / 8, channels 4 (lane % 8) ..+3 (scalar base + lane offset)
    const unsigned la = (unsigned)((lane >> 3) * g.CA + (lane & 7) * 4) * 4u;
    const unsigned lb = (unsigned)((lane >> 3) * g.CB + (lane & 7) * 4) * 4u;
    const unsigned as_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)As);
    const unsigned bs_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)Bs);
    auto dma = [&](unsigned lane_off, const float *src_uniform, unsigned lds) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(src_uniform), "s"(lds) : "memory");
    };
    // operand addresses of this lane: pixel pair q = pixels 2q + h = (q >> 2, 2 (q & 3) + h) of the 8x8 map
    const float *ar = As + (qa * 64 + h) * 32 + l31;                                       // + q * 64
    const float *br = Bs + (qb * PH * PH + h * S + tg * (NTW / K) * PH) * 32 + l31;       // + (pixel, tap) offset

    for (long long b = b_lo; b < b_hi; ++b) {
        __syncthreads();                                    // the previous image's operands have been read (first: the zero fill)
#pragma unroll
        for (int j = 0; j < (NPIECE + 3) / 4; ++j) {
            const int p = wave_u + 4 * j;
            if (p < NPA) {
                const int ct = p >> 3, px = (p & 7) * 8;
                dma(la, A + ((size_t)(b * 64 + px) * g.CA + ca0 + 32 * ct), as_lds + (unsigned)((ct * 64 + px) * 128));
            } else if (p < NPIECE) {
                const int r = p - NPA, ct = r / (HB * S), rr = r - ct * (HB * S), y = rr / S, u = rr - y * S;
                dma(lb, Bt + ((size_t)((b * HB + y) * HB + 8 * u) * g.CB + cb0 + 32 * ct),
                    bs_lds + (unsigned)((ct * PH * PH + (y + PAD) * PH + PAD + 8 * u) * 128));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float av = ar[q * 64];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int ky = t / K, kx = t % K;
                const float bv = br[(((q >> 2) * S + ky) * PH + 2 * (q & 3) * S + kx) * 32];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    float *dst = partial + (size_t)split * K * K * g.CA * g.CB;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tap = tg * NTW + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = ca0 + qa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[((size_t)tap * g.CA + a) * g.CB + cb0 + qb * 32 + l31] = acc[t][r];
        }
    }
}

// The map-resident weight gradient on two-term fp16 products (round 4): the same decomposition as conv_wgrad_map8_kernel -- a
// workgroup owns (WA x 32) x (WB x 32) channels of (ca, cb) and a range of images, a wave a 32 x 32 tile for NTW taps -- with
// v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_32x32x2_f32: three term products per 16 pixels instead of eight fp32 steps,
// about three times the fp32 pipe's rate at the clock the chip holds under this load.
//   * The reduction index of an MFMA is the pixel, so each lane needs EIGHT CONSECUTIVE PIXELS of one channel, while the
//     activations are [pixel][channel].  The LDS images stay [pixel][32 channels] (64-byte rows of fp16, written with 8-byte
//     stores straight from the converted float4s), and ds_read_b64_tr_b16 turns 4 pixels x 16 channels around on the way to the
//     registers: lane (c = lane & 31, h = lane >> 5) receives pixels 8 h + 4 rd .. + 3 of channel c from the address
//     row(8 h + 4 rd + ((lane & 15) >> 2)) * 64 + (16 ((lane >> 4) & 1) + 4 (lane & 3)) * 2 (tools/ubench/tr_probe.hip checks
//     exactly this on the device).  A tap is a different ROW offset into the zero-framed Bt image -- a compile-time immediate,
//     no alignment case -- and for the stride-2 layer the image keeps even and odd columns in separate planes so that the four
//     pixels of a read stay consecutive.  Rows of 64 bytes: the four rows of a 32-lane half cover all 64 banks once.
//   * Scales: x = (h1 + h2) 2^-k with ONE power of two per image and operand tile, measured by the workgroup while the image
//     sits in its registers (no pass over the tensors, no maxima handed in).  Images of one range carry different scales, so the
//     fp32 accumulators live on the CURRENT image's scale 2^(ka + kb) and are multiplied by the exact power of two between two
//     images; the partial sums leave through v_ldexp.  An image more than 2^60 below the LARGEST one of the range so far keeps a
//     coarser scale, so that the accumulators cannot overflow however the magnitudes are ordered.  Relative error per product
//     <= 2^-21 as in the forward kernels (representation 2^-23 per operand + the dropped h2 h2 term); elements 2^17 below their image tile's maximum keep fewer bits
//     -- absolute error <= 2^-39 of the tile maxima's product -- which is the forward scheme's statement for activations.
//   * Fixed summation order as before (image order inside a range, ranges by conv_wgrad_reduce_kernel): bit-reproducible.
typedef short s16x4v __attribute__((__vector_size__(8)));
typedef short s16x8v __attribute__((__vector_size__(16)));
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u32x2 __attribute__((ext_vector_type(2)));

// two fp32 values times sc -> packed fp16 leading terms (p1) and packed fp16 remainders (p2)
__device__ __forceinline__ void wg_split2(float a, float b, float sc, unsigned &p1, unsigned &p2) {
    unsigned hh;
    float ra, rb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(a), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(b), "v"(sc));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(sc), "v"(hh));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(sc), "v"(hh));
    const wg_f16x2 r = {(_Float16)ra, (_Float16)rb};
    p1 = hh;
    p2 = __builtin_bit_cast(unsigned, r);
}

template <int K, int S, int WA, int WB>
__global__ __launch_bounds__(256, 2) void conv_wgrad_map8_h2_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                    float *__restrict__ partial, WgradGeom g, int imgs_per_split) {
    constexpr int TG = 4 / (WA * WB), NTW = K * K / TG, PH = 7 * S + K, HB = 8 * S, PAD = K == 1 ? 0 : 1;
    static_assert(WA * WB * TG == 4 && NTW * TG == K * K && NTW % K == 0, "wave layout");
    static_assert(S == 1 || (S == 2 && PH % 2 == 0), "column planes of the stride-2 image");
    constexpr int NPIXB = PH * PH;                                             // pixels of the framed Bt image (both column planes)
    constexpr int APL = WA * 64 * 64, BPL = WB * NPIXB * 64;                   // bytes of one term's A / Bt planes
    constexpr int NPA = WA * 8, NPB = WB * HB * S, NPIECE = NPA + NPB, PPW = (NPIECE + 3) / 4;   // 8-pixel x 32-channel pieces
    static_assert(2 * (APL + BPL) + 64 <= 64 * 1024, "static LDS");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (APL + BPL)];
    __shared__ float red[8];
    unsigned char *As = smem, *Bs = smem + 2 * APL;                            // [term][tile][pixel][32 ch] fp16 each
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qa = wave_u % WA, qb = (wave_u / WA) % WB, tg = wave_u / (WA * WB);
    const int tiles_b = g.CB / (32 * WB);
    const int tb = blockIdx.x % tiles_b, ta = blockIdx.x / tiles_b;
    const int ca0 = ta * 32 * WA, cb0 = tb * 32 * WB;
    const int split = blockIdx.y;
    const long long b_lo = (long long)split * imgs_per_split;
    long long b_hi = b_lo + imgs_per_split;
    if (b_hi > g.B) b_hi = g.B;

    for (int i = tid; i < 2 * BPL / 16; i += 256) reinterpret_cast<f32x4 *>(Bs)[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // the frame stays zero
    f32x16 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // piece p = wave + 4 j of an image: lane -> pixel lane / 8 of the piece, channels 4 (lane % 8) .. + 3
    f32x4 raw[PPW];
    auto load_image = [&](long long b) {
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave_u + 4 * j;
            if (p < NPA) {
                const int ct = p >> 3, px = (p & 7) * 8 + (lane >> 3);
                raw[j] = *reinterpret_cast<const f32x4 *>(A + (size_t)(b * 64 + px) * g.CA + ca0 + 32 * ct + 4 * (lane & 7));
            } else if (p < NPIECE) {
                const int r = p - NPA, ct = r / (HB * S), rr = r - ct * (HB * S), y = rr / S, u = rr - y * S;
                raw[j] = *reinterpret_cast<const f32x4 *>(Bt + (size_t)((b * HB + y) * HB + 8 * u + (lane >> 3)) * g.CB + cb0 + 32 * ct + 4 * (lane & 7));
            } else raw[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    // operand addresses of this lane (see the header): A pixel 8 h + r, Bt frame row h * S / column r * S of the wave's first tap row
    const int r4 = (lane & 15) >> 2, chb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    typedef __attribute__((address_space(3))) s16x4v *lds_tr_t;
    const unsigned char *aop = As + ((qa * 64 + 8 * h + r4) * 64 + chb);
    const int trow = tg * (NTW / K);                                            // first kernel row of this wave's taps
    const unsigned char *bop = Bs + ((qb * NPIXB + (S == 1 ? (h + trow) * PH + r4 : ((h * 2 + trow) * 2) * (PH / 2) + r4)) * 64 + chb);
    auto tr8 = [&](const unsigned char *base, int off) {                       // eight pixels of this lane's channel: two transposing reads
        const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + off));
        const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_t)(base + off + 4 * 64));
        return __builtin_bit_cast(wg_f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    int e_acc = 0, e_min = 0;                                                   // the accumulators hold sum * 2^e_acc; e_min: the largest image's scale
    bool first = true;
    if (b_lo < b_hi) load_image(b_lo);
    for (long long b = b_lo; b < b_hi; ++b) {
        // ---- this image's two scales: maxima of the A tile and of the Bt tile over the workgroup ----
        float ma = 0.0f, mb = 0.0f;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const float m = fmaxf(fmaxf(__builtin_fabsf(raw[j].x), __builtin_fabsf(raw[j].y)), fmaxf(__builtin_fabsf(raw[j].z), __builtin_fabsf(raw[j].w)));
            if (wave_u + 4 * j < NPA) ma = fmaxf(ma, m); else mb = fmaxf(mb, m);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o)); mb = fmaxf(mb, __shfl_xor(mb, o)); }
        if (lane == 0) { red[2 * wave_u] = ma; red[2 * wave_u + 1] = mb; }
        __syncthreads();                                    // + every wave is done with the previous image's operands
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        auto scale_exp = [](float m) {                      // -> the power of two that puts m into [2^14, 2^15); 0 for 0 / Inf / NaN
            int e = 15;
            // (round 5, ADVICE r4: every FINITE maximum is scaled, also one above 3e38 -- it used to fall through to "unscaled", become
            // Inf in fp16 and NaN in the residual term where the fp32 kernel stays finite; the clamp below is +-113, fp32's own range)
            if (m > 0.0f && m <= 3.4028234e38f) (void)__builtin_frexpf(m, &e);
            e = 15 - e;
            return e > 113 ? 113 : (e < -113 ? -113 : e);
        };
        const int ka = scale_exp(ma);
        int kb = scale_exp(mb);
        // an image far SMALLER than the largest one summed so far would push the accumulators towards overflow when they follow
        // its scale (sums at the largest image's own scale stay below 2^48: 2^30 per product, 64 pixels, <= 2^12 images): such
        // an image keeps a coarser scale, at most 2^60 finer than the LARGEST image's -- not than the previous image's, or a run
        // of ever smaller images would climb 60 binades at a time (tests/test_wgrad_scheme_cpu.py found that) -- and its whole
        // contribution is below 2^-60 of that image's
        e_min = first ? ka + kb : (ka + kb < e_min ? ka + kb : e_min);
        if (ka + kb > e_min + 60) kb = e_min + 60 - ka;
        const int e_img = __builtin_amdgcn_readfirstlane(ka + kb);
        const float sa = __builtin_ldexpf(1.0f, ka), sb = __builtin_ldexpf(1.0f, kb);
        // ---- convert and park: [term][tile][pixel][32 ch], 8 bytes per lane and term ----
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = wave_u + 4 * j;
            unsigned p1a, p2a, p1b, p2b;
            unsigned char *dst;
            if (p < NPA) {
                const int ct = p >> 3, px = (p & 7) * 8 + (lane >> 3);
                wg_split2(raw[j].x, raw[j].y, sa, p1a, p2a);
                wg_split2(raw[j].z, raw[j].w, sa, p1b, p2b);
                dst = As + ((ct * 64 + px) * 64 + (lane & 7) * 8);
                *reinterpret_cast<wg_u32x2 *>(dst) = wg_u32x2{p1a, p1b};
                *reinterpret_cast<wg_u32x2 *>(dst + APL) = wg_u32x2{p2a, p2b};
            } else if (p < NPIECE) {
                const int r = p - NPA, ct = r / (HB * S), rr = r - ct * (HB * S), y = rr / S, u = rr - y * S;
                const int Y = y + PAD, X = 8 * u + (lane >> 3) + PAD;
                const int idx = S == 1 ? Y * PH + X : (Y * 2 + (X & 1)) * (PH / 2) + (X >> 1);
                wg_split2(raw[j].x, raw[j].y, sb, p1a, p2a);
                wg_split2(raw[j].z, raw[j].w, sb, p1b, p2b);
                dst = Bs + ((ct * NPIXB + idx) * 64 + (lane & 7) * 8);
                *reinterpret_cast<wg_u32x2 *>(dst) = wg_u32x2{p1a, p1b};
                *reinterpret_cast<wg_u32x2 *>(dst + BPL) = wg_u32x2{p2a, p2b};
            }
        }
        __syncthreads();
        if (b + 1 < b_hi) load_image(b + 1);                // in flight under this image's matrix work
        // ---- the accumulators follow the image's scale ----
        if (first) { e_acc = e_img; first = false; }
        else if (e_img != e_acc) {
            const int d = e_img - e_acc;
            const float f = d < -120 ? 0.0f : __builtin_ldexpf(1.0f, d);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= f;
            e_acc = e_img;
        }
        // ---- 64 pixels = four 16-pixel steps (A rows 2 s, 2 s + 1), every tap of this wave per step ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const wg_f16x8 a1 = tr8(aop, (16 * s4) * 64), a2 = tr8(aop, (16 * s4) * 64 + APL);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int ky = t / K, kx = t % K;
                const int off = S == 1 ? ((2 * s4 + ky) * PH + kx) * 64
                                       : ((((2 * s4) * 2 + ky) * 2 + (kx & 1)) * (PH / 2) + (kx >> 1)) * 64;
                const wg_f16x8 b1 = tr8(bop, off), b2 = tr8(bop, off + BPL);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[t], 0, 0, 0);
            }
        }
    }
    const int l31 = lane & 31;
    float *dst = partial + (size_t)split * K * K * g.CA * g.CB;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tap = tg * NTW + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int a = ca0 + qa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[((size_t)tap * g.CA + a) * g.CB + cb0 + qb * 32 + l31] = __builtin_ldexpf(acc[t][r], -e_acc);
        }
    }
}

// Weight gradient when Bt is a small NCHW image tensor (first / last layer: CB <= 4 image channels): the k*k taps
// are folded into the MFMA N dimension (column n = tap*CB + cb, k*k*CB <= 64), so one pass over the pixels serves
// every tap instead of k*k passes that each fill 3 of 64 columns.  One wave owns one image at a time: the image's
// CB planes are staged in LDS (wave-private), the B operand is gathered from them at the tap-shifted position,
// the A operand (row-major, CA <= 64) is one coalesced channel per lane straight from global memory.
__global__ __launch_bounds__(256) void conv_wgrad_img_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                             float *__restrict__ partial, WgradGeom g, int imgs_per_wg) {
    constexpr int MT = 2, NT = 2;
    extern __shared__ __attribute__((aligned(16))) float smem_img[];         // [4 waves][CB*HB*WB], later `red`
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int plane = g.HB * g.WB, img_f = g.CB * plane;
    float *img = smem_img + (size_t)wave * img_f;
    const int ntap = g.k * g.k, ncol = ntap * g.CB;

    // this lane's B columns: n = nt*32 + l31 -> (tap, cb)
    int ky[NT], kx[NT], kyp[NT], kxp[NT], cbo[NT];
    bool nok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + l31;
        nok[nt] = n < ncol;
        const int tap = nok[nt] ? n / g.CB : 0;
        cbo[nt] = (nok[nt] ? n - tap * g.CB : 0) * plane;
        ky[nt] = tap / g.k; kx[nt] = tap - ky[nt] * g.k;
        kyp[nt] = ky[nt] - g.pad; kxp[nt] = kx[nt] - g.pad;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const long long b_lo = (long long)blockIdx.x * imgs_per_wg;
    long long b_hi = b_lo + imgs_per_wg;
    if (b_hi > g.B) b_hi = g.B;
    const int npx = g.HA * g.WA;
    for (long long b = b_lo + wave; b < b_hi; b += 4) {
        __builtin_amdgcn_wave_barrier();
        const float *src = Bt + (size_t)b * img_f;
        for (int i = lane; i < img_f; i += 64) img[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const float *ab = A + (size_t)b * npx * g.CA;
        int yA = h / g.WA, xA = h - yA * g.WA;               // pixel h of the image
        // eight pixel pairs per round: their A values are requested together (one memory latency per round, not per pair)
        for (int p0 = 0; p0 < npx; p0 += 16) {
            float av[8][MT];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = p0 + 2 * j + h;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int c = mt * 32 + l31;
                    av[j][mt] = (p < npx && c < g.CA) ? ab[(size_t)p * g.CA + c] : 0.0f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = p0 + 2 * j + h;
                const bool pok = p < npx;
                // (yA, xA) of pixel p, carried along instead of divided out: two pixels further per step
                const int ys = yA * g.stride, xs = xA * g.stride;
                float bv[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int yB = ys + kyp[nt], xB = xs + kxp[nt];
                    const bool ok = pok && nok[nt] && (unsigned)yB < (unsigned)g.HB && (unsigned)xB < (unsigned)g.WB;
                    bv[nt] = ok ? img[cbo[nt] + yB * g.WB + xB] : 0.0f;
                }
                xA += 2;
                while (xA >= g.WA) { xA -= g.WA; ++yA; }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    float *red = smem_img;                                   // 4096 floats (the launch sizes LDS for it)
    for (int w = 3; w >= 1; --w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((mt * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
        }
        __syncthreads();
        if (wave == w - 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] += red[((mt * NT + nt) * 16 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        // partial layout of the generic kernel: [split = workgroup][tap][ca][cb]
        float *dst = partial + (size_t)blockIdx.x * ntap * g.CA * g.CB;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int a = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int n = nt * 32 + l31;
                    if (a < g.CA && n < ncol) {
                        const int tap = n / g.CB, cb = n - tap * g.CB;
                        dst[((size_t)tap * g.CA + a) * g.CB + cb] = acc[mt][nt][r];
                    }
                }
    }
}

// dW[ca][cb][tap] = sum_split partial[split][tap][ca][cb]   (fixed order)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ partial, int nsplit, int ntap,
                                                                int CA, int CB, float *__restrict__ dw) {
    const long long total = (long long)ntap * CA * CB;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        // eight interleaved running sums (loads in flight instead of one dependent add per load), combined in a fixed order
        float s8[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s8[j] += partial[(size_t)(sp + j) * total + e];
        for (int j = 0; sp < nsplit; ++sp, ++j) s8[j] += partial[(size_t)sp * total + e];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        const int tap = (int)(e / ((long long)CA * CB));
        const long long rem = e - (long long)tap * CA * CB;          // ca * CB + cb
        dw[rem * ntap + tap] = s;
    }
}

// per-channel sums of a (P, C) row-major tensor [or an NCHW one: (B, C, HW)] -- stage 1: one partial per block.
// Row-major with C % 4 == 0: a thread owns four consecutive channels (16-byte loads), 1024 / C rows in flight per
// block iteration, fp64 accumulators; otherwise one channel per thread.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float *__restrict__ g, long long P, int C,
                                                                long long HW, int nchw, long long rows_per_block,
                                                                double *__restrict__ partial) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const long long lo = (long long)blockIdx.x * rows_per_block;
    long long hi = lo + rows_per_block;
    if (hi > P) hi = P;
    if (!nchw && (C & 3) == 0 && C <= 1024) {
        const int c4 = C >> 2;                              // threads per row
        const int G = 256 / c4 > 0 ? 256 / c4 : 1;          // rows per block iteration (C <= 1024)
        const int grp = tid / c4, q = tid - grp * c4;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (grp < G) {
            long long p = lo + grp;
            for (; p + 3 * G < hi; p += 4 * G) {             // four rows in flight; the sums keep their row order
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4 *>(g + (size_t)(p + j * G) * C + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a0 += (double)v[j].x; a1 += (double)v[j].y; a2 += (double)v[j].z; a3 += (double)v[j].w; }
            }
            for (; p < hi; p += G) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(g + (size_t)p * C + 4 * q);
                a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
            }
        }
        // red[(grp * C) + channel]
        if (grp < G) {
            red[grp * C + 4 * q] = a0; red[grp * C + 4 * q + 1] = a1; red[grp * C + 4 * q + 2] = a2; red[grp * C + 4 * q + 3] = a3;
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            double s = 0.0;
            for (int w = 0; w < G; ++w) s += red[w * C + c];
            partial[(size_t)blockIdx.x * C + c] = s;
        }
        return;
    }
    const int G = 256 / C > 0 ? 256 / C : 1;            // C <= 256
    const int grp = tid / C, c = tid - grp * C;
    double acc = 0.0;
    if (grp < G) {
        for (long long p = lo + grp; p < hi; p += G) {
            float v;
            if (nchw) {
                const long long b = p / HW;
                v = g[((size_t)b * C + c) * HW + (p - b * HW)];
            } else {
                v = g[(size_t)p * C + c];
            }
            acc += (double)v;
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < C) {
        double s = 0.0;
        for (int q = 0; q < G; ++q) s += red[q * C + tid];
        partial[(size_t)blockIdx.x * C + tid] = s;
    }
}

// stage 2: one workgroup per channel, fixed-order tree over the block partials
__global__ __launch_bounds__(256) void bias_grad_final_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                              float *__restrict__ db) {
    __shared__ double red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int b = tid; b < nblocks; b += 256) s += partial[(size_t)b * C + c];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) db[c] = (float)red[0];
}

__global__ __launch_bounds__(256) void relu_backward_kernel(const float *__restrict__ gout, const float *__restrict__ y,
                                                            long long n, float *__restrict__ gin) {
    const long long n4 = n >> 2;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gout), *y4 = reinterpret_cast<const f32x4 *>(y);
    f32x4 *o4 = reinterpret_cast<f32x4 *>(gin);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 gv = g4[i], yv = y4[i];
        f32x4 o;
        o.x = yv.x > 0.0f ? gv.x : 0.0f; o.y = yv.y > 0.0f ? gv.y : 0.0f;
        o.z = yv.z > 0.0f ? gv.z : 0.0f; o.w = yv.w > 0.0f ? gv.w : 0.0f;
        o4[i] = o;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) gin[i] = y[i] > 0.0f ? gout[i] : 0.0f;
}

static int wgrad_plan(int64_t B, int HA, int WA, int CA, int CB, int k, WgradGeom &g) {
    const long long nblk = ((long long)B * HA * WA + 31) / 32;        // 32-pixel blocks
    // enough workgroups to fill the chip a few times over: tiles * taps * splits >= ~8 per CU
    const long long tiles = (long long)((CA + 63) / 64) * ((CB + 63) / 64) * k * k;
    long long ns = (8LL * num_cus() + tiles - 1) / tiles;
    if (ns > (nblk + 7) / 8) ns = (nblk + 7) / 8;         // at least 8 blocks per workgroup
    if (ns > kWgMaxSplit) ns = kWgMaxSplit;
    if (ns < 1) ns = 1;
    g.nsplit = (int)ns;
    g.rows_per_split = (nblk + ns - 1) / ns;
    return VQVAE_OK;
}

}  // namespace vqvae

using namespace vqvae;

extern "C" {

size_t vqvae_conv_wgrad_workspace_bytes(int CA, int CB, int k) {
    if (CA < 1 || CB < 1 || k < 1 || k > 4) return 0;
    // image-operand kernel: kWgImgSplit partials; map-resident kernel: kWgMapSplit workgroups over its (ca, cb) tiles
    // (a tile is 64 x 64 or 32 x 128 channels, 64 x 32 for k = 4); generic kernel: kWgMaxSplit
    size_t splits = (k * k * CB <= 64 && CA <= 64) ? kWgImgSplit : kWgMaxSplit;
    if (CA % 32 == 0 && CB % 32 == 0) {
        const size_t tiles = k == 4 ? (size_t)((CA + 63) / 64) * (CB / 32) : ((size_t)CA * CB + 4095) / 4096;
        const size_t ns = ((size_t)kWgMapSplit + tiles - 1) / tiles;
        if (ns > splits) splits = ns;
    }
    return splits * k * k * CA * CB * sizeof(float);
}

int vqvae_conv_wgrad_f32(const float *a, const float *bt, int64_t B, int HA, int WA, int CA, int HB, int WB, int CB,
                         int k, int stride, int pad, int bt_nchw, float *grad_w, void *workspace,
                         size_t workspace_bytes, vqvae_stream_t stream) {
    return vqvae_conv_wgrad_ex_f32(a, bt, B, HA, WA, CA, HB, WB, CB, k, stride, pad, bt_nchw, VQVAE_CONV_EXACT_FP32, grad_w, workspace,
                                   workspace_bytes, stream);
}

int vqvae_conv_wgrad_ex_f32(const float *a, const float *bt, int64_t B, int HA, int WA, int CA, int HB, int WB, int CB,
                            int k, int stride, int pad, int bt_nchw, int flags, float *grad_w, void *workspace,
                            size_t workspace_bytes, vqvae_stream_t stream) {
    if (!a || !bt || !grad_w) return VQVAE_ERR_NULL;
    if (flags & ~VQVAE_CONV_EXACT_FP32) return VQVAE_ERR_UNSUPPORTED;
    // two-term fp16 products where the map-resident kernel applies and multiplies enough to pay for its conversion pass (the
    // 1x1 layers are operand-bound: their fp32 form is the faster one, 38 against 53 us at B = 4096)
    const bool h2 = !(flags & VQVAE_CONV_EXACT_FP32) && k >= 3;
    if (B < 1 || HA < 1 || WA < 1 || CA < 1 || HB < 1 || WB < 1 || CB < 1 || stride < 1 || pad < 0) return VQVAE_ERR_SHAPE;
    if (k < 1 || k > 4) return VQVAE_ERR_UNSUPPORTED;
    if (B * (int64_t)HA * WA > INT32_MAX || B * (int64_t)HB * WB * CB > ((int64_t)1 << 40)) return VQVAE_ERR_OVERFLOW;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(bt)) & 15) return VQVAE_ERR_UNSUPPORTED;   // 16-byte loads
    if (!workspace || workspace_bytes < vqvae_conv_wgrad_workspace_bytes(CA, CB, k)) return VQVAE_ERR_WORKSPACE;
    WgradGeom g;
    g.B = (int)B; g.HA = HA; g.WA = WA; g.CA = CA; g.HB = HB; g.WB = WB; g.CB = CB;
    g.k = k; g.stride = stride; g.pad = pad; g.bt_nchw = bt_nchw ? 1 : 0;
    wgrad_plan(B, HA, WA, CA, CB, k, g);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *partial = static_cast<float *>(workspace);
    const size_t img_lds = (size_t)4 * CB * HB * WB * sizeof(float);
    if (bt_nchw && CA <= 64 && k * k * CB <= 64 && img_lds <= 96 * 1024) {
        // small NCHW image operand: taps folded into the MFMA N dimension, one image per wave
        long long nwg = (B + 3) / 4;
        if (nwg > kWgImgSplit) nwg = kWgImgSplit;
        const int ipw = (int)((B + nwg - 1) / nwg);
        nwg = (B + ipw - 1) / ipw;
        const size_t lds = img_lds > 16384 ? img_lds : 16384;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_img_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL(conv_wgrad_img_kernel, dim3((unsigned)nwg), dim3(256), lds, st, a, bt, partial, g, ipw);
        const long long tot = (long long)k * k * CA * CB;
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, partial,
                           (int)nwg, k * k, CA, CB, grad_w);
        return (int)hipGetLastError();
    }
    const long long total = (long long)k * k * CA * CB;
    long long rgrid = (total + 255) / 256;
    if (rgrid > 4096) rgrid = 4096;
    // 8x8 A maps with whole 32-channel tiles: both maps resident in LDS, all taps from one staging
    if (!bt_nchw && HA == 8 && WA == 8 && HB == 8 * stride && WB == 8 * stride && pad == (k == 1 ? 0 : 1) &&
        ((k == 4 && stride == 2) || ((k == 3 || k == 1) && stride == 1))) {
        int wa = 0, wb = 0;
        if (k == 4) { if (CA % 64 == 0 && CB % 32 == 0) wa = 2, wb = 1; }
        else if (CA % 64 == 0 && CB % 64 == 0) wa = 2, wb = 2;
        else if (CA % 32 == 0 && CB % 128 == 0) wa = 1, wb = 4;
        else if (CA % 128 == 0 && CB % 32 == 0) wa = 4, wb = 1;
        if (wa) {
            const long long tiles = (long long)(CA / (32 * wa)) * (CB / (32 * wb));
            long long ns = (kWgMapSplit + tiles - 1) / tiles;              // two workgroups per CU of the 256
            if (ns > B) ns = B;
            const int ips = (int)((B + ns - 1) / ns);
            ns = (B + ips - 1) / ips;
            const dim3 grid((unsigned)tiles, (unsigned)ns);
#define MAP8_LAUNCH(K_, S_, WA_, WB_)                                                                                         \
    do {                                                                                                                      \
        if (h2) hipLaunchKernelGGL((conv_wgrad_map8_h2_kernel<K_, S_, WA_, WB_>), grid, dim3(256), 0, st, a, bt, partial, g, ips); \
        else hipLaunchKernelGGL((conv_wgrad_map8_kernel<K_, S_, WA_, WB_>), grid, dim3(256), 0, st, a, bt, partial, g, ips);  \
    } while (0)
            if (k == 4) MAP8_LAUNCH(4, 2, 2, 1);
            else if (k == 3 && wa == 2) MAP8_LAUNCH(3, 1, 2, 2);
            else if (k == 3 && wa == 1) MAP8_LAUNCH(3, 1, 1, 4);
            else if (k == 3) MAP8_LAUNCH(3, 1, 4, 1);
            else if (wa == 2) MAP8_LAUNCH(1, 1, 2, 2);
            else if (wa == 1) MAP8_LAUNCH(1, 1, 1, 4);
            else MAP8_LAUNCH(1, 1, 4, 1);
#undef MAP8_LAUNCH
            hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, st, partial, (int)ns, k * k, CA, CB, grad_w);
            return (int)hipGetLastError();
        }
    }
    const unsigned gx = (unsigned)(((CA + 63) / 64) * ((CB + 63) / 64) * k * k);
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(gx, (unsigned)g.nsplit), dim3(256), 0, st, a, bt, partial, g);
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)rgrid), dim3(256), 0, st, partial, g.nsplit, k * k, CA,
                       CB, grad_w);
    return (int)hipGetLastError();
}

size_t vqvae_bias_grad_workspace_bytes(int C) { return C < 1 || C > 256 ? 0 : (size_t)1024 * C * sizeof(double); }

int vqvae_bias_grad_f32(const float *grad_y, int64_t B, int HW, int C, int nchw, float *grad_b, void *workspace,
                        size_t workspace_bytes, vqvae_stream_t stream) {
    if (!grad_y || !grad_b) return VQVAE_ERR_NULL;
    if (B < 1 || HW < 1 || C < 1) return VQVAE_ERR_SHAPE;
    if (C > 256) return VQVAE_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < vqvae_bias_grad_workspace_bytes(C)) return VQVAE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long P = (long long)B * HW;
    long long nb = (P + 1023) / 1024;
    if (nb > 512) nb = 512;
    const long long rpb = (P + nb - 1) / nb;
    nb = (P + rpb - 1) / rpb;
    double *partial = static_cast<double *>(workspace);
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, grad_y, P, C, (long long)HW,
                       nchw ? 1 : 0, rpb, partial);
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3((unsigned)C), dim3(256), 0, st, partial, (int)nb, C, grad_b);
    return (int)hipGetLastError();
}

int vqvae_relu_backward_f32(const float *grad_out, const float *y, int64_t n, float *grad_in, vqvae_stream_t stream) {
    if (!grad_out || !y || !grad_in) return VQVAE_ERR_NULL;
    if (n < 1) return VQVAE_ERR_SHAPE;
    if ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(grad_in)) & 15)
        return VQVAE_ERR_UNSUPPORTED;                    // 16-byte accesses
    long long grid = ((n >> 2) + 255) / 256;
    if (grid > 65536) grid = 65536;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       grad_out, y, (long long)n, grad_in);
    return (int)hipGetLastError();
}

}  // extern "C"
