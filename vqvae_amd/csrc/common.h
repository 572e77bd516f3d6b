// Shared declarations for libvqvae_hip.so (gfx950 only; compiled with
// -ffp-contract=off so every fused multiply-add in this library is explicit).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/vqvae_hip.h"

namespace vqvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// |v|^2 of the eight fp16 values in a 16-byte vector (four packed pairs), added to s.  Inline assembly on purpose: hipcc (ROCm 7.2,
// clang 22) lowers four consecutive __builtin_amdgcn_fdot2 calls on the components .x .y .z .w of one loaded vector to four
// v_dot2c_f32_f16 that ALL read the FIRST component's register -- the sum came out as 4 (x0^2 + x1^2), right in expectation on
// i.i.d. channels and silently far too small on rows whose energy sits in other channels.  Round 3's screen bound took its row
// norm |z^| from that sum (vq_track.hip, round 2's vq_sweep.hip, the fused quantizer in conv_fused.hip): found in round 4 by the
// heterogeneous-channel parity test (tests/golden/vq_hetero_unit.npz is the row that came back with the wrong index).
// tools/hazard_scan.py now also rejects the miscompiled pattern in every source's assembly.
__device__ __forceinline__ float sqsum8_f16(unsigned vx, unsigned vy, unsigned vz, unsigned vw, float s) {
    asm("v_dot2c_f32_f16 %0, %1, %1\n\tv_dot2c_f32_f16 %0, %2, %2\n\tv_dot2c_f32_f16 %0, %3, %3\n\tv_dot2c_f32_f16 %0, %4, %4"
        : "+v"(s) : "v"(vx), "v"(vy), "v"(vz), "v"(vw));
    return s;
}

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kLdsBytes = 160 * 1024;  // per CU on gfx950

// CU count of the CURRENT device (cached per device id: one process may drive several GPUs)
inline int num_cus() {
    static int n[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        hipDeviceProp_t p;
        int c = 0;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) c = p.multiProcessorCount;
        n[dev] = c > 0 ? c : 256;          // benign race: every thread computes the same value
    }
    return n[dev];
}

// profiling hooks (capi.hip); no-ops unless vqvae_profile_enable(1)
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);
bool prof_dispatch(int id, hipEvent_t *start, hipEvent_t *stop);    // events for hipExtLaunchKernelGGL (null when profiling is off)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- quantizer workspace layout (host + device agree on it) -----------------
struct VqPlan {
    int KC;           // codes per LDS chunk (multiple of 32)
    int nchunks;      // ceil(K / KC)
    int K_pad;        // nchunks * KC
    size_t off_flags, off_ee, off_img, off_partials, off_img16, off_neh, off_imgh, off_seeds, off_imgf, off_chunk, total;
    size_t lds_bytes;
    // filter-and-refine kernel (bf16 screening): usable when the whole bf16 image fits LDS
    bool filter_ok;
    int K32;                 // K rounded up to 32
    size_t filter_lds_bytes;
};
constexpr int kVqCandCap = 8;   // per lane half (16 per row), unsigned short entries   // candidate list capacity per row in the filter kernel
constexpr int kVqTilesPerWave = 2;   // 32-row tiles a wave of the filter kernel walks per iteration
constexpr int kVqTicketBytes = 4096;  // unit counters behind the workspace's flags (see vq_plan)
constexpr int kVqMaxGrid = 1024;     // persistent grid never exceeds this many workgroups
constexpr int kVqSlabRows = 1 << 18; // rows per pass of the streamed-codebook kernels (vq_chunk.hip): bounds their scratch
constexpr int kVqGroupSlabs = 16;    // slabs whose open / hard rows are resolved by ONE launch (their records' scratch: 44.1 B per row of a group)

bool vq_track_ok(int K, int D);          // vq_track.hip: the codebook's fp16 image fits LDS next to four waves' 32-row tiles (D = 64, K <= 1024; row-major rows)
bool vq_chunk_ok(int K, int D);
size_t vq_chunk_scratch_bytes(int D);

inline VqPlan vq_plan(int K, int D) {
    VqPlan p;
    // LDS: chunk image KC*D*4 + ee KC*4 + histogram K*4 + 256 B reduction scratch
    const long budget = kLdsBytes - 256 - (long)K * 4;
    long kc = budget / ((long)D * 4 + 4);
    kc = kc / 32 * 32;
    const long kneed = ((long)K + 31) / 32 * 32;
    if (kc > kneed) kc = kneed;
    if (kc < 32) kc = 32;
    p.KC = (int)kc;
    p.nchunks = (K + p.KC - 1) / p.KC;
    p.K_pad = p.nchunks * p.KC;
    p.lds_bytes = (size_t)p.KC * D * 4 + (size_t)p.KC * 4 + (size_t)K * 4 + 256;
    p.off_flags = 0;
    // behind the flags: kVqTicketBytes of unit counters of the stream-tracker kernel (vq_track.hip: eight groups of workgroups, one
    // 256-byte slot each for the group's next pooled unit and for the workgroups that have left; zero between launches)
    p.off_ee = 256 + kVqTicketBytes;
    p.off_img = align_up(p.off_ee + (size_t)p.K_pad * 4, 256);
    p.off_partials = align_up(p.off_img + (size_t)p.K_pad * D * 4, 256);
    p.K32 = (K + 31) / 32 * 32;
    p.off_img16 = align_up(p.off_partials + sizeof(double) * kVqMaxGrid, 256);
    p.off_neh = align_up(p.off_img16 + (size_t)p.K32 * D * 2, 256);
    p.off_imgh = align_up(p.off_neh + (size_t)p.K32 * 4, 256);        // fp16 image + seeds of the single-sweep kernel
    // (+ 512 codes / 4 KiB of padding: the streamed-codebook kernel copies whole chunks)
    p.off_seeds = align_up(p.off_imgh + (size_t)(p.K32 + 512) * D * 2, 256);
    // the same fp16 image in the fused conv kernels' channel order (the quantizer inside the encoder's last kernel, D = 64)
    p.off_imgf = align_up(p.off_seeds + (size_t)p.K32 * 4 + 4096, 256);
    p.off_chunk = align_up(p.off_imgf + (size_t)(p.K32 + 512) * D * 2, 256);
    // row scratch of the streamed-codebook kernels: only where they are the default path
    p.total = p.off_chunk + ((vq_chunk_ok(K, D) && !vq_track_ok(K, D)) ? vq_chunk_scratch_bytes(D) : 0);
    // LDS of the filter kernel: bf16 image + (-||e||^2/2) + histogram + per-wave candidate lists + scratch
    p.filter_lds_bytes = (size_t)p.K32 * D * 2 + (size_t)p.K32 * 4 + (size_t)K * 4 +
                         kVqTilesPerWave * (8 * 32 * 2 * kVqCandCap * 2 + 8 * 96 * 4) + 256 + 8;
    p.filter_ok = (D == 64) && p.filter_lds_bytes <= (size_t)kLdsBytes;
    return p;
}

// vq_filter.hip: bf16-screened, exactly-refined VectorQuantizer kernel (same outputs as the exact one)
int launch_vq_filter_d64(const float *z, const float *cb, long long N, int HW, int K, bool rowmajor,
                         float *zq, long long *idx, int *hist, char *ws, hipStream_t st, int *grid_out);

// vq_track.hip: single-sweep fp16-screened, exactly-refined VectorQuantizer kernel with the stream tracker (D = 64; row-major rows or NCHW)
bool vq_track_nchw_ok(int K, int D, int HW);       // NCHW input: maps whose pixel count is a multiple of 64
// launch form forced by the caller's flags: 0 = the rule, 8 / 16 = the two public forms, 12 / 32 = experimental (both bits: 32-row units on
// twelve waves; VQVAE_VQ_UNITS32_8WAVES: 32-row units on eight waves)
inline int vq_form_of_flags(int flags) {
    const bool a = flags & VQVAE_VQ_UNITS64_8WAVES, b = flags & VQVAE_VQ_UNITS32_16WAVES;
    if (flags & VQVAE_VQ_UNITS32_8WAVES) return 32;
    return a && b ? 12 : (b ? 16 : (a ? 8 : 0));
}
struct VqTrackForm { int waves, unit_rows, grid, pool_pct; long long nunits; };     // waves per CU, rows per unit, workgroups, pooled tail units (%)
bool vq_track_form(long long N, int K, int HW, bool nchw, int form, int cus, VqTrackForm &f);      // false: the kernel does not take this problem
int launch_vq_track_d64(const float *z, const float *cb, long long N, int K, float *zq, long long *idx, int *hist,
                        char *ws, hipStream_t st, int *grid_out, int HW = 0, bool nchw = false, int form = 0);
void launch_vq_prepare16(const float *cb, int K, int D, char *ws, hipStream_t st);
// vq_chunk.hip: the same screen with the codebook image streamed through LDS (D = 64 / 128, any K <= 16384)
// zq_amax: NULL, or an array of N / hw ints (images of hw consecutive rows) that receives max |z_q| per image (atomicMax on the
// bits of a non-negative float: the caller has filled it with -1)
int launch_vq_chunked(const float *z, const float *cb, long long N, int K, int D, float *zq, long long *idx, int *hist,
                      char *ws, hipStream_t st, int *grid_out, int *zq_amax = nullptr, int hw = 1);

// conv.hip: the per-layer entry points with the per-image activation maxima of the two-term fp16 product path
// (arrays of B ints, -1 = not provided; NULL = none): written by a producing layer, read by the consuming one.
int conv_forward_impl(int kind, const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin,
                      int Cout, int flags, float *y, hipStream_t stream, const int *in_amax, int *out_amax,
                      const float *ep_add = nullptr, const float *ep_mask = nullptr);   // vqvae_conv_forward_ep_f32
int res_layer_forward_impl(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C,
                           int Rh, int flags, float *y, hipStream_t stream, const int *in_amax, int *out_amax,
                           float *hidden = nullptr, float *hid_scratch = nullptr);
// widths the fused residual kernels cover (else: 3x3 conv -> 1x1 conv -> combine through hid_scratch, B*H*W*Rh floats)
bool res_layer_fused_ok(int C, int Rh);      // hidden: relu(W1 (*) r(x)) as (B,8,8,32), kept for backward (8x8 maps, Rh = 32)
bool res_pair_supported(int H, int W, int C, int Rh, int flags);
// a 1x1 conv (+ bias) fused behind a residual pair: packed = vqvae_conv_pack_f32(VQVAE_CONV_1x1, ...), out (B,8,8,Cout) row-major
// zero / zero_n (conv_res_pair_forward_impl only): ints the kernel clears for the next kernel of the stream
// The quantizer inside the encoder's last kernel (conv_res_pair8_h2_kernel<2, true>, models/vqvae.py:33-34 in one launch):
// the codebook's prepared images (vq_prepare_impl: fp16 image in that kernel's channel order, seeds, ||e||^2, bound
// statistics), the codebook itself and where the quantizer's outputs go.  K32 must be a multiple of 128, K <= 1024, D = 64.
struct VqFuse {
    const uint4 *imgf; const float *seeds; const float *ee; const int *flags; const float *cb;
    int K, K32;
    float *zq; long long *idx; int *hist; double *partials;
};
struct ResPairPost { const float *packed; const float *bias; int Cout; float *out; int *zero = nullptr; int zero_n = 0;
                     const VqFuse *vq = nullptr; bool debug_ze = false; };      // debug_ze: the fused quantizer also writes its z_e rows to `out`
bool res_pair_post_supported(int C, int Cout);
int res_pair_forward_impl(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C,
                          int Rh, int flags, float *y, hipStream_t stream, const int *in_amax, int *out_amax,
                          const ResPairPost *post = nullptr);
bool conv_res_pair_supported(int kind, int H, int W, int Cin, int C, int Rh);
int conv_res_pair_forward_impl(int kind, const float *x, const float *packed_front, const float *bias_front, int Cin,
                               const float *packed_w1, const float *packed_w2, int64_t B, int H, int W, int C, int Rh, int flags,
                               float *y, hipStream_t stream, const int *in_amax, int *out_amax, const ResPairPost *post = nullptr,
                               const int64_t *gather_idx = nullptr, int gather_K = 0);     // gather_idx: x = a (K, Cin) table, pixel p takes row gather_idx[p]
int convt_out_forward_impl(const float *x, const float *packed, const float *bias, int64_t B, int H, int W, int Cin, int Cout,
                           int flags, float *y_nchw, hipStream_t stream, const int *in_amax);
// log2 of the spread of a conv weight's INPUT channels after per-output-channel normalisation -> out[0] (device); w == NULL: 0
void weight_spread_impl(const float *w, int Cout, int Cin, int taps, bool transposed, float *out, hipStream_t st);
void act_absmax_impl(const float *x, int64_t B, long long elems_per_image, int *amax, hipStream_t st);   // amax[b] = max(amax[b], max |x_b|)
// the pieces of vq_forward_impl around its main kernel, for the whole-path entry that quantizes inside the encoder's last kernel
// any embedding width (vq_generic.hip): the exact fp32 path for D outside {32, 64, 128, 256}
constexpr int kVqGenericMaxD = 256;      // (beyond 383 the reference's matmul is no longer one fmaf chain: MKL blocks the reduction)
constexpr int kRowSqnormMaxD = 1024;     // the row-sum test hook alone
bool vq_generic_ok(int K, int D);
size_t vq_generic_workspace_bytes(int K, int D);
int launch_vq_generic(const float *z, const float *cb, long long N, int HW, int K, int D, float beta, bool rowmajor, float *zq,
                      long long *idx, int *hist, float *loss, float *ppl, char *ws, hipStream_t st, bool hist_zeroed,
                      bool vector_units = false, bool prepared = false);
bool vq_fuse_ok(int K, int D, int64_t B, int flags);
int vq_prepare_impl(const float *codebook, int K, int D, int flags, void *workspace, size_t workspace_bytes, hipStream_t st);
VqFuse vq_fuse_args(const float *codebook, int K, void *workspace, float *z_q, int64_t *idx, int32_t *hist);
int vq_finalize_impl(const double *partials, int grid, int32_t *hist, int K, int64_t n_rows, int D, float beta, float *loss,
                     float *perplexity, hipStream_t st);
int vq_forward_impl(const float *z_e, const float *codebook, int64_t B, int D, int H, int W, int K, float beta, int flags,
                    float *z_q, int64_t *idx, int32_t *hist, float *loss, float *perplexity, void *workspace,
                    size_t workspace_bytes, vqvae_stream_t stream, bool hist_zeroed, int *zq_amax = nullptr,
                    bool *zq_amax_done = nullptr);     // zq_amax: see launch_vq_chunked; *zq_amax_done = a kernel that publishes it ran
bool enc_front_supported(int H, int W, int Cin, int C1, int C2);
int enc_front_forward_impl(const float *x_nchw, const float *packed_in, const float *bias_in, const float *packed2,
                           const float *bias2, int64_t B, int H, int W, int Cin, int C1, int C2, float *y, hipStream_t st,
                           int *out_amax, int *zero_buf = nullptr, int zero_n = 0);
bool dec_tail_supported(int h4, int w4, int C, int C1, int Cout);
int dec_tail_forward_impl(const float *x, const float *packed2, const float *bias2, const float *packed4, const float *bias4,
                          int64_t B, int h4, int w4, int C, int C1, int Cout, float *y_nchw, hipStream_t st, const int *in_amax);
int conv_in_forward_impl(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H, int W, int Cin,
                         int Cout, int flags, float *y, hipStream_t stream, int *out_amax, const float *ep_mask = nullptr);

}  // namespace vqvae
