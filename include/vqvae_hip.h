/*
 * vqvae_hip.h -- C ABI of libvqvae_hip.so: the MI355X (gfx950) implementation of
 * the VQ-VAE forward hot path of MishaLaskin/vqvae.
 *
 * The reference has no FFI or plugin interface; its seam for this path is the
 * nn.Module.forward boundary of five Python classes.  Each entry
 * point below names the reference interface it replaces (file:line under the
 * reference tree).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     its name says host; all tensors are dense, contiguous fp32 unless noted.
 *   - the caller owns and allocates every buffer, including the workspace
 *     (size from the matching *_workspace_bytes()); the library never allocates,
 *     frees or synchronises; all work is enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the null stream).
 *   - inputs are const and never written.
 *   - return value: 0 on success, a negative VQVAE_ERR_* on argument errors
 *     (nothing is launched), a positive hipError_t if a launch failed.
 *     No exceptions cross this boundary.  vqvae_strerror() names any of them.
 */
#ifndef VQVAE_HIP_H
#define VQVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQVAE_HIP_ABI_VERSION 9   /* 2: round 2's packed-weight layout, whole-path entries; 3: quantizer flag 0x10 = VQVAE_VQ_TOP3_KEYS;
                                    4: forward in parts (begin / part / end), residual layer with hidden output, larger
                                       weight-gradient and streamed-quantizer workspaces (always ask the *_bytes functions);
                                    5-6: round 4's headers in front of the two-term images, whole-path product flags, removed
                                       quantizer flags; 7: data-gradient epilogues (vqvae_conv_forward_ep_f32);
                                    8: vqvae_vq_launch_form, unit counters in the quantizer workspace (its size changed: ask
                                       vqvae_vq_workspace_bytes), K up to 1024 on the resident-image and the fused quantizer;
                                    9: round 5 -- vqvae_encode_f32 / vqvae_decode_f32 (the index wire format as fused entry points),
                                       VQVAE_VQ_UNITS32_8WAVES, the sixteen-wave quantizer form at every size */

#define VQVAE_OK               0
#define VQVAE_ERR_NULL        -1   /* a required pointer is NULL                       */
#define VQVAE_ERR_SHAPE       -2   /* non-positive or inconsistent dimension           */
#define VQVAE_ERR_UNSUPPORTED -3   /* shape outside what the kernels are built for     */
#define VQVAE_ERR_WORKSPACE   -4   /* workspace missing or smaller than required       */
#define VQVAE_ERR_OVERFLOW    -5   /* an element count does not fit the index type     */

typedef void *vqvae_stream_t;      /* hipStream_t */

#if defined(__GNUC__)
#define VQVAE_API __attribute__((visibility("default")))
#else
#define VQVAE_API
#endif

VQVAE_API int vqvae_abi_version(void);
/* sha256 (16 hex digits) over the names and bytes of the kernel sources (vqvae_amd/csrc) this library was built from; the same string
 * follows the marker "VQVAE_SRC_FP=" inside the library FILE, so a loader can tell a stale build without loading it (vqvae_amd/_lib.py
 * rebuilds or refuses; bench.py stamps its profiles with it).  Round 5; no ABI change for existing callers.                          */
VQVAE_API const char *vqvae_source_fingerprint(void);
VQVAE_API const char *vqvae_strerror(int code);

/* ---------------------------------------------------------------- profiling
 * Measurement aid for bench.py (never used on the product path): while enabled,
 * every launch of an instrumented kernel is bracketed by a pair of hipEvents
 * recorded on the launch stream.  vqvae_profile_collect() is the only call in
 * this library that synchronises: it waits for the recorded events of one
 * kernel id, returns the summed kernel time and launch count, and resets them.
 * At most 256 launches per id are recorded between collects.                  */
#define VQVAE_PROF_VQ_MAIN     0   /* fused VectorQuantizer kernel                     */
#define VQVAE_PROF_CONV_IGEMM  1   /* implicit-GEMM conv / conv-transpose kernels      */
#define VQVAE_PROF_RES_LAYER   2   /* fused residual-layer kernel                      */
#define VQVAE_PROF_CONV_IN     3   /* first conv (NCHW image in)                       */
#define VQVAE_PROF_CONV_OUT    4   /* last conv-transpose (NCHW image out)             */
#define VQVAE_PROF_NUM_IDS     5
VQVAE_API int vqvae_profile_enable(int on);
VQVAE_API int vqvae_profile_collect(int kernel_id, double *total_ms, int *launches);

/* Box calibration (bench.py): one launch of a bare fp16 MFMA stream on random operands (512 workgroups x 4 waves, `iters` x 4
 * v_mfma_f32_32x32x16_f16 per wave = vqvae_calibration_flops(iters) flop); the caller times it with events.  scratch (at least
 * vqvae_calibration_scratch_bytes()): afterwards 512 pairs of uint64 {shader cycles, 100 MHz ticks} of each workgroup's loop
 * -- cycles / (ticks * 10 ns) = the clock the chip held under this load -- followed by the kernel's (meaningless) sums.       */
VQVAE_API size_t vqvae_calibration_scratch_bytes(void);
VQVAE_API double vqvae_calibration_flops(int iters);
VQVAE_API int vqvae_calibration_mfma_f16(int iters, void *scratch, size_t scratch_bytes, vqvae_stream_t stream);

/* ---------------------------------------------------------------- quantizer */

/* flags for vqvae_vq_forward_f32 */
#define VQVAE_VQ_NCHW          0x0  /* z_e / z_q are (B,D,H,W): the module boundary   */
#define VQVAE_VQ_ROWMAJOR      0x1  /* z_e / z_q are (B,H,W,D) = (N,D) rows (internal) */
#define VQVAE_VQ_CODEBOOK_PREPARED 0x2 /* workspace already holds this codebook's image
                                        (a previous call with the same codebook, K, D
                                        and workspace): skip the prepare kernel        */

#define VQVAE_VQ_EXACT_SWEEP    0x4  /* force the exhaustive exact-fp32 MFMA sweep instead of the
                                        16-bit-screened + exactly-refined kernels (identical outputs) */
#define VQVAE_VQ_BF16_FILTER    0x8  /* use round 1's two-sweep bf16 filter kernel where the default would be the
                                        single-sweep fp16 kernel (identical outputs; A/B timing and tests) */

#define VQVAE_VQ_REMOVED_FLAGS  0x30 /* 0x10 / 0x20 selected round 2's tracker kernel (vq_sweep_kernel_d64: index-carrying top-3 keys; 32-row units
                                        on sixteen waves) for A/B runs.  Round 4 removed that kernel: VQVAE_ERR_UNSUPPORTED */

#define VQVAE_VQ_UNITS64_8WAVES  0x100 /* vq_track_kernel_d64's launch form, forced (identical outputs; tests and A/B timing): 64-row units on */
#define VQVAE_VQ_UNITS32_16WAVES 0x200 /* eight waves per CU / 32-row units on sixteen waves per CU (row-major rows, K <= 512).  Default: the
                                        * 32-row units on eight waves up to 8 units per CU (every CU busy before any wave gets a second unit),
                                        * the sixteen-wave form where a wave gets at most two units (N <= 2 x 16 x CUs x 32 rows), else 64-row
                                        * units on eight waves */

#define VQVAE_VQ_UNITS32_8WAVES  0x400 /* 32-row units on eight waves per CU, forced (the rule takes this form up to 8 units per CU) */

#define VQVAE_VQ_UNFUSED        0x40 /* vqvae_forward_f32 only: run the quantizer as its own launch even where the encoder's last
                                        kernel would quantize its z_e in place (32x32 images, h_dim 128, K = 128 k <= 1024, D = 64: z_e is
                                        then never written); identical outputs, A/B timing and tests */

/* Which kernel vqvae_vq_forward_f32 launches for this shape / flags ("vq_track_kernel_d64" (codebook image resident in LDS: D = 64,
 * K <= 1024 -- up to ~600 beside eight or sixteen waves' tiles, beyond that with four waves per CU; row-major rows or, since round 5 for
 * every such K, NCHW maps whose pixel count is a multiple of 32),
 * "vq_stream_sweep_kernel" (image streamed through LDS: D = 64 / 128, K <= 16384),
 * "vq_filter_kernel_d64", "vq_exact_kernel"), and how many times that kernel sweeps the codebook on the 16-bit matrix
 * cores per row (0 for the exact-fp32 kernel).  For reporting (bench.py). */
VQVAE_API const char *vqvae_vq_kernel_name(int K, int D, int flags);
VQVAE_API int vqvae_vq_screen_sweeps(int K, int D, int flags);
/* Where vqvae_vq_kernel_name says "vq_track_kernel_d64": the launch form that kernel takes for n_rows rows (HW = pixels per image,
 * which matters for NCHW maps only) on the CURRENT device -- waves per CU (4 / 8 / 16), rows per unit (32 / 64) and the share of
 * the units (per cent) that waves draw from per-group pools at the end; VQVAE_ERR_UNSUPPORTED where another kernel runs.
 * For reporting and tests (bench.py names the template instance that ran). */
VQVAE_API int vqvae_vq_launch_form(int64_t n_rows, int K, int D, int HW, int flags, int *waves, int *unit_rows, int *pool_pct);

/* Bytes of workspace vqvae_vq_forward_f32 needs (any number of rows: the streamed-codebook kernels work through the rows
 * in slabs of 2^18 and resolve the open rows of up to sixteen slabs per launch, so their scratch -- 221 MB at D = 64,
 * 254 MB at D = 128: a slab's fp16 rows + 44 bytes of records per row of a sixteen-slab group -- does not grow with n_rows). */
VQVAE_API size_t vqvae_vq_workspace_bytes(int64_t n_rows, int K, int D);

/*
 * Fused VectorQuantizer forward.  Replaces VectorQuantizer.forward
 * (models/quantizer.py:29-76): NCHW->rows (:45-46), the distance matrix
 * ||z||^2 + ||e||^2 - 2 z.E^T (:49-51), argmin (:54), the one-hot @ E gather
 * (:55-60), the loss (:63-64), the straight-through value z + (z_q - z) (:67),
 * the perplexity (:70-71) and rows->NCHW (:74) -- one pass over z_e.
 *
 *   z_e        (B,D,H,W) fp32 [or (B,H,W,D) with VQVAE_VQ_ROWMAJOR]        in
 *   codebook   (K,D) fp32 row-major = embedding.weight (:26)               in
 *   z_q        same shape/layout as z_e; may be NULL (index-only encode)   out
 *   idx        (N) int64, N = B*H*W, row order (b,h,w) = min_encoding_indices (N,1)  out
 *   hist       (K) int32: how many rows chose each code (column sums of
 *              min_encodings, :55-57)                                      out
 *   loss       1 fp32  = mean((z_q-z)^2) + beta*mean((z_q-z)^2)  (:63-64)  out
 *   perplexity 1 fp32  = exp(-sum p log(p + 1e-10)), p = hist/N  (:70-71)  out
 *
 * Bit-exactness contract (tests/test_vq_gpu.py): idx and z_q are bit-identical
 * to the reference's for identical z_e bits, including first-index tie-breaking
 * and NaN-counts-as-minimum; loss / perplexity agree to rtol 1e-6.
 * Supported: 1 <= K <= 16384; D in {32, 64, 128, 256} on the kernels named by vqvae_vq_kernel_name; any other 1 <= D <= 256
 * (main.py:21 leaves --embedding_dim free and the reference "just runs") with the same contract on the fp32 matrix cores (round 6,
 * vq_anyd_kernel: rows and codes zero-padded to a multiple of eight channels -- fmaf(0, 0, acc) = acc, so v_mfma_f32_32x32x2_f32 stays
 * the reference's k-ordered chain; ~60-80 TFLOP/s of distance arithmetic); VQVAE_VQ_BF16_FILTER selects round 5's per-thread vector
 * chains there (vq_generic_kernel, 16 TFLOP/s; kept for A/B runs), the other kernel-selection flags select nothing for these widths.
 * D > 256 stays VQVAE_ERR_UNSUPPORTED: from D = 384 on the reference's own z @ E^T is no longer one k-ordered fmaf chain (its sgemm
 * blocks the reduction), so there are no pinned bits to be exact against.
 */
VQVAE_API int vqvae_vq_forward_f32(const float *z_e, const float *codebook,
                         int64_t B, int D, int H, int W, int K, float beta, int flags,
                         float *z_q, int64_t *idx, int32_t *hist,
                         float *loss, float *perplexity,
                         void *workspace, size_t workspace_bytes, vqvae_stream_t stream);

/* min_encodings, the (N,K) fp32 one-hot (models/quantizer.py:55-57).  Optional:
 * VQVAE.forward discards it (models/vqvae.py:34); N*K must fit in int64.       */
/* Test hook (round 5): out[r] = torch.sum(x[r] ** 2) of `rows` row-major rows of any width D <= 1024, in the order ATen's CPU kernel
 * uses (models/quantizer.py:49-50 delegates it; the tests' CPU oracle restates it) -- mode 0: one thread per row (the codebook term of
 * vq_generic_kernel), mode 1: the workgroup-cooperative form (its row term).  tests/test_vq_generic_gpu.py compares both with
 * torch.sum bit for bit.                                                                                                               */
VQVAE_API int vqvae_debug_row_sqnorm_f32(const float *x, int64_t rows, int D, int mode, float *out, vqvae_stream_t stream);

VQVAE_API int vqvae_vq_onehot_f32(const int64_t *idx, int64_t N, int K, float *onehot,
                        vqvae_stream_t stream);

/* indices -> z_q (B,D,H,W): the one-hot @ embedding.weight -> view -> permute
 * sequence of the notebook's generate_samples (visualization.ipynb:358-365).
 * An index outside [0, K) never reads the codebook: its D elements become NaN
 * (the reference raises; a launch cannot -- validate on the host if the indices
 * are not your own, as the Python front end does).                             */
VQVAE_API int vqvae_vq_decode_indices_f32(const int64_t *idx, const float *codebook,
                                int64_t B, int D, int H, int W, int K,
                                float *z_q, vqvae_stream_t stream);

/* ------------------------------------------------------- conv / residual stacks
 * Activations between layers are ROW-MAJOR (B,H,W,C): one contiguous C-vector per
 * pixel.  NCHW appears only at the image boundaries (vqvae_conv_in_*, vqvae_convt_out_*)
 * and, for callers that use sub-modules directly, through vqvae_transpose_f32.
 * Weights are passed PACKED: vqvae_*_pack_f32 rewrites a torch-layout weight into the
 * MFMA B-operand image once per weight version (bytes from vqvae_*_packed_bytes).
 * fp32 in, fp32 out, fp32 accumulation.  Products are formed from 16-bit splits of both fp32 operands on the
 * 16-bit matrix cores: two fp16 terms per operand and three term products on 8x8 maps (operands carry exact
 * power-of-two scales: per layer for the weights, per image for the activations, measured by the kernel itself) --
 * relative error per product <= 2^-21 (representation 2^-23 per operand, dropped term pair 2^-22; fp32's own is 2^-24);
 * three bf16 terms (an exact split) and six term products on other map sizes and everywhere with VQVAE_CONV_BF16_SPLIT --
 * per product <= 3*2^-24.  The whole-path entry points (vqvae_forward_f32 ...) hand the per-image maxima from layer
 * to layer and use the fp16 scheme on every map size.  VQVAE_CONV_EXACT_FP32 selects the exact-fp32 MFMA kernels.  Parity with the reference is
 * tolerance-level either way (oneDNN's summation order is opaque): |y - y_ref| <= 1e-5 + 1e-4|y_ref|.
 * All activation pointers must be 16-byte aligned (VQVAE_ERR_UNSUPPORTED otherwise).
 * Shapes the reference uses at 32x32 images (8x8 maps, and the 4x4 s2 conv on 16x16 maps) take
 * tile-resident kernels (one image per wavefront); every other shape takes the generic implicit-GEMM ones. */
#define VQVAE_CONV_4x4_S2   0   /* nn.Conv2d(k=4,s=2,p=1), weight (Cout,Cin,4,4)   encoder.py:29-33 */
#define VQVAE_CONV_3x3_S1   1   /* nn.Conv2d(k=3,s=1,p=1), weight (Cout,Cin,3,3)   encoder.py:35, residual.py:20 */
#define VQVAE_CONV_1x1      2   /* nn.Conv2d(k=1),         weight (Cout,Cin,1,1)   vqvae.py:16, residual.py:23 */
#define VQVAE_CONVT_3x3_S1  3   /* nn.ConvTranspose2d(k=3,s=1,p=1), weight (Cin,Cout,3,3)  decoder.py:28 */
#define VQVAE_CONVT_4x4_S2  4   /* nn.ConvTranspose2d(k=4,s=2,p=1), weight (Cin,Cout,4,4)  decoder.py:31 */
#define VQVAE_CONV_TAPS     6   /* stride-1 conv over an explicit tap list: only through the vqvae_conv_taps_* entry points below */
#define VQVAE_CONVT_1x1     5   /* nn.ConvTranspose2d(k=1), weight (Cin,Cout,1,1): the data gradient of a 1x1 nn.Conv2d */

#define VQVAE_CONV_RELU_IN  0x1 /* apply ReLU to the input as it is read (the in-place nn.ReLU(True)
                                   in front of a conv, residual.py:19,22)                          */
#define VQVAE_CONV_RELU_OUT 0x2 /* ReLU on the result (encoder.py:31,34, decoder.py:33)           */
#define VQVAE_CONV_BF16_SPLIT 0x8 /* use round 1's three-term bf16 products (6 per fp32 product) where the default on 8x8
                                    maps is the two-term fp16 scheme (3 per product, same error bound; A/B and tests) */
#define VQVAE_CONV_EXACT_FP32 0x4 /* use the exact-fp32 MFMA kernels (157 TF peak) instead of the default 16-bit
                                   split products (see above)                                                  */

VQVAE_API size_t vqvae_conv_packed_bytes(int kind, int Cin, int Cout);
VQVAE_API int vqvae_conv_pack_f32(int kind, const float *w, int Cin, int Cout, float *packed,
                                  vqvae_stream_t stream);
/* y = conv(x) + bias [ReLU]; x (B,H,W,Cin) row-major, y (B,Hout,Wout,Cout) row-major; bias may be
 * NULL.  Cin must be a multiple of 4.  Replaces one nn.Conv2d / nn.ConvTranspose2d call.
 * packed: [fp32 image][three-term bf16 image][4x4 s2 only: the same in space-to-depth chunk order]
 *         [header {weight scale exponent}][two-term fp16 image][4x4 s2 only: the same in space-to-depth chunk order]. */
/* How many 16-bit MFMA term products vqvae_conv_forward_f32 issues per fp32 multiply-add for this layer shape: 3 (two-term
 * fp16, 8x8 maps), 6 (three-term bf16), 1 (VQVAE_CONV_EXACT_FP32), 0 = unsupported shape.  With
 * VQVAE_CONV_QUERY_WHOLE_PATH in flags: what the whole-path entry points (vqvae_forward_f32 ...) launch for the layer --
 * they hand the per-image maxima over, so every map size runs the two-term fp16 products (3).  For reporting (bench.py). */
#define VQVAE_CONV_QUERY_WHOLE_PATH 0x100
VQVAE_API int vqvae_conv_term_products(int kind, int H, int W, int Cin, int Cout, int flags);

VQVAE_API int vqvae_conv_forward_f32(int kind, const float *x, const float *packed, const float *bias,
                                     int64_t B, int H, int W, int Cin, int Cout, int flags,
                                     float *y, vqvae_stream_t stream);

/* One ResidualLayer.forward (models/residual.py:27-29) as a single kernel:
 *   y = r(x) + W2 (*) relu(W1 (*) r(x)),  r = ReLU if VQVAE_CONV_RELU_IN else identity,
 *   followed by ReLU if VQVAE_CONV_RELU_OUT (the stack's final F.relu, residual.py:50, or the
 *   next layer's in-place ReLU hoisted into this one).
 * packed_w1 = pack(VQVAE_CONV_3x3_S1, res_block[1].weight (Rh,C,3,3)), packed_w2 =
 * pack(VQVAE_CONV_1x1, res_block[3].weight (C,Rh,1,1)).  C in {32,64,128}, Rh <= 32, x != y (other widths:
 * vqvae_res_layer_forward_ws_f32 below; here VQVAE_ERR_UNSUPPORTED).                                */
VQVAE_API int vqvae_res_layer_forward_f32(const float *x, const float *packed_w1,
                                          const float *packed_w2, int64_t B, int H, int W, int C,
                                          int Rh, int flags, float *y, vqvae_stream_t stream);

/* The same for ANY width with C % 4 == 0 and Rh % 4 == 0 (round 4: main.py's --n_hiddens / --n_residual_hiddens are free):
 * widths outside the fused kernels (C not in {32,64,128} or Rh > 32) run as 3x3 conv -> 1x1 conv -> skip + ReLU through the conv
 * kernels, the hidden map in `scratch` (at least B*H*W*Rh floats; unused and may be NULL for the fused widths).              */
VQVAE_API int vqvae_res_layer_forward_ws_f32(const float *x, const float *packed_w1, const float *packed_w2, int64_t B, int H,
                                             int W, int C, int Rh, int flags, float *y, float *scratch, size_t scratch_bytes,
                                             vqvae_stream_t stream);
/* The same layer for a training step (main.py:74-78 through models/residual.py:27-29): also writes the hidden
 * activation relu(W1 (*) r(x)) as hidden (B,H,W,Rh) row-major, which the backward pass needs for the ReLU mask and the
 * 1x1 weight gradient -- instead of recomputing the 3x3 conv there.  Only where a wave owns whole images:
 * H = W = 8, Rh = 32 (VQVAE_ERR_UNSUPPORTED otherwise: the caller recomputes with vqvae_conv_forward_f32).     */
VQVAE_API int vqvae_res_layer_forward_hidden_f32(const float *x, const float *packed_w1,
                                                 const float *packed_w2, int64_t B, int H, int W, int C,
                                                 int Rh, int flags, float *y, float *hidden,
                                                 vqvae_stream_t stream);

/* First encoder conv, nn.Conv2d(Cin,Cout,k=4,s=2,p=1) (models/encoder.py:29-31), reading the NCHW
 * image x (B,Cin,H,W) and writing row-major (B,H/2,W/2,Cout).  Cin in {1,3,4}, Cout <= 128.
 * flags: VQVAE_CONV_RELU_OUT, VQVAE_CONV_EXACT_FP32 (fp32 MFMA instead of the split-bf16 products).
 * (The packed image also carries the two-term fp16 operand image of the fused encoder front, see vqvae_encoder_f32.)  */
VQVAE_API size_t vqvae_conv_in_packed_bytes(int Cin, int Cout);
VQVAE_API int vqvae_conv_in_pack_f32(const float *w, int Cin, int Cout, float *packed,
                                     vqvae_stream_t stream);
VQVAE_API int vqvae_conv_in_forward_f32(const float *x_nchw, const float *packed, const float *bias,
                                        int64_t B, int H, int W, int Cin, int Cout, int flags,
                                        float *y, vqvae_stream_t stream);

/* Last decoder layer, nn.ConvTranspose2d(Cin,Cout,k=4,s=2,p=1) (models/decoder.py:34-35), reading
 * row-major (B,H,W,Cin) and writing the NCHW image (B,Cout,2H,2W).  Cout <= 4, Cin % 4 == 0,
 * Cin <= 256.  Runs as GEMM (pixels x Cin x 16*Cout on the MFMA) + in-LDS col2im.  flags: 0, or
 * VQVAE_CONV_EXACT_FP32 for the fp32 MFMA instead of the split-bf16 products.                      */
VQVAE_API size_t vqvae_convt_out_packed_bytes(int Cin, int Cout);
VQVAE_API int vqvae_convt_out_pack_f32(const float *w, int Cin, int Cout, float *packed,
                                       vqvae_stream_t stream);
VQVAE_API int vqvae_convt_out_forward_f32(const float *x, const float *packed, const float *bias,
                                          int64_t B, int H, int W, int Cin, int Cout, int flags,
                                          float *y_nchw, vqvae_stream_t stream);

/* Batched transpose x[batch][R][C] -> y[batch][C][R]: (B,C,HW) <-> (B,HW,C) layout changes for
 * callers that enter or leave the path at a sub-module boundary (visualization.ipynb:84-90).     */
VQVAE_API int vqvae_transpose_f32(const float *x, int64_t batch, int R, int C, float *y,
                                  vqvae_stream_t stream);

/* ------------------------------------------------------- training-step companions
 * what main.py:74-83 needs around the forward path.
 *
 * vqvae_vq_backward_f32 -- the gradients autograd derives from models/quantizer.py:63-67:
 *     grad_z        = grad_zq + g * 2 (z - e_idx) / (N D)             (:63 first term, :67 straight-through)
 *     grad_codebook = g * 2 beta * sum_{i: idx_i=k} (e_k - z_i) / (N D)   (:63-64; beta sits on the codebook
 *                     term in the reference, the reverse of the paper -- reproduced as written)
 *   g = *grad_loss (device scalar; NULL = 1), grad_zq = upstream gradient of the returned z_q (NULL = 0).
 *   z_e / grad_zq / grad_z share the layout selected by VQVAE_VQ_ROWMAJOR.  Either output may be NULL.
 *   The codebook gradient uses no floating-point atomics: rows are stably sorted by code and summed in a
 *   fixed order in fp64, so it is bit-reproducible run to run.  Parity with torch autograd: rtol 1e-5.   */
VQVAE_API size_t vqvae_vq_backward_workspace_bytes(int64_t N, int K, int D);
VQVAE_API int vqvae_vq_backward_f32(const float *z_e, const float *codebook, const int64_t *idx,
                                    const float *grad_zq, const float *grad_loss,
                                    int64_t B, int D, int H, int W, int K, float beta, int flags,
                                    float *grad_z, float *grad_codebook,
                                    void *workspace, size_t workspace_bytes, vqvae_stream_t stream);

/* recon_loss = mean((x_hat - x)^2) / x_train_var; loss = recon_loss + embedding_loss (main.py:75-76).
 * out3 = {recon_loss, loss, perplexity}: the three values main.py:81-83 copies to the host one by one,
 * packed so that a step needs one D2H copy.  embedding_loss / perplexity are device scalars (NULL = 0);
 * x_hat and x must be 16-byte aligned.  inv_var = 1 / x_train_var.                                    */
VQVAE_API size_t vqvae_recon_loss_workspace_bytes(void);
VQVAE_API int vqvae_recon_loss_f32(const float *x_hat, const float *x, int64_t n, float inv_var,
                                   const float *embedding_loss, const float *perplexity, float *out3,
                                   void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
/* grad_x_hat = g * 2 (x_hat - x) inv_var / n,  g = *grad_loss (NULL = 1) */
VQVAE_API int vqvae_recon_loss_backward_f32(const float *x_hat, const float *x, int64_t n, float inv_var,
                                            const float *grad_loss, float *grad_x_hat,
                                            vqvae_stream_t stream);

/* ------------------------------------------------------- conv backward
 * Data gradients reuse the forward kernels: d/dx of nn.Conv2d(k,s,p) is the ConvTranspose2d with the SAME weight
 * tensor and vice versa (kinds 0<->4, 1<->3, 2<->5; ReLU masks via vqvae_relu_backward_f32); d/dx of the last
 * layer (convT 4x4 s2, Cout<=4) is vqvae_conv_in_forward_f32 on the NCHW gradient with that layer's weight.
 *
 * vqvae_conv_wgrad_f32 -- weight gradient of one conv / conv-transpose layer:
 *     grad_w[ca][cb][ky][kx] = sum_{b,y,x} a[b,y,x,ca] * bt[b, y*stride + ky - pad, x*stride + kx - pad, cb]
 *   nn.Conv2d:          a = grad_y (B,Ho,Wo,Cout), bt = x (B,H,W,Cin)        -> grad_w is (Cout,Cin,k,k)
 *   nn.ConvTranspose2d: a = x (B,H,W,Cin),         bt = grad_y (B,Ho,Wo,Cout) -> grad_w is (Cin,Cout,k,k)
 *   a is row-major; bt is row-major, or an NCHW image tensor when bt_nchw != 0 (first / last layer).
 *   Exact fp32 products (fp32 MFMA), fixed-order reduction: bit-reproducible.  k <= 4.
 *   8x8 a maps with 32-channel multiples (every layer of the path between the first and the last at 32x32 images):
 *   both maps of an image resident in LDS, all taps from one staging (conv_wgrad_map8_kernel); else pixel blocks
 *   per tap.  The workspace holds the per-range partial sums ([range][tap][ca][cb]).                          */
VQVAE_API size_t vqvae_conv_wgrad_workspace_bytes(int CA, int CB, int k);
VQVAE_API int vqvae_conv_wgrad_f32(const float *a, const float *bt, int64_t B, int HA, int WA, int CA,
                                   int HB, int WB, int CB, int k, int stride, int pad, int bt_nchw,
                                   float *grad_w, void *workspace, size_t workspace_bytes,
                                   vqvae_stream_t stream);
/* The same with a product-scheme flag (round 4).  flags = 0: on the map-resident shapes (8x8 a maps, 32-channel multiples) with
 * k >= 3 the products are formed from two fp16 terms per operand on the fp16 matrix cores (conv_wgrad_map8_h2_kernel: <= 2^-21 relative per
 * product, one power-of-two scale per image and operand tile measured inside the kernel, accumulators rescaled exactly between
 * images, still a fixed summation order) -- about three times the fp32 pipe's rate; every other shape runs the fp32 kernels
 * either way.  flags = VQVAE_CONV_EXACT_FP32: vqvae_conv_wgrad_f32 (which is this call with that flag).  Same workspace. */
VQVAE_API int vqvae_conv_wgrad_ex_f32(const float *a, const float *bt, int64_t B, int HA, int WA, int CA,
                                      int HB, int WB, int CB, int k, int stride, int pad, int bt_nchw, int flags,
                                      float *grad_w, void *workspace, size_t workspace_bytes,
                                      vqvae_stream_t stream);
/* grad_b[c] = sum over pixels of grad_y[.., c]; grad_y (B*HW, C) row-major, or (B,C,HW) when nchw != 0.  C <= 256. */
VQVAE_API size_t vqvae_bias_grad_workspace_bytes(int C);
VQVAE_API int vqvae_bias_grad_f32(const float *grad_y, int64_t B, int HW, int C, int nchw, float *grad_b,
                                  void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
/* grad_in = grad_out * (y > 0): backward of y = relu(.)                                                      */
VQVAE_API int vqvae_relu_backward_f32(const float *grad_out, const float *y, int64_t n, float *grad_in,
                                      vqvae_stream_t stream);

/* Data-gradient launches with the element-wise tail of the backward chain in the conv's epilogue (round 4):
 *     y = (mask > 0) ? conv(x) + addend : 0
 *   addend (or NULL): the gradient arriving over a residual layer's skip (models/residual.py:28: d(x + block(x)));
 *   mask   (or NULL): the activation whose ReLU sits below this layer in the forward order -- the conv's own INPUT in the
 *                     forward pass is relu(.) of the layer below, so its data gradient times (input > 0) is that ReLU's
 *                     backward (what vqvae_relu_backward_f32 did in a pass of its own);
 *   both row-major with y's shape, 16-byte aligned, distinct from y.  Split-product kernels only: VQVAE_ERR_UNSUPPORTED with
 *   VQVAE_CONV_EXACT_FP32, and for the first-layer kernel outside its row-band form (the caller then uses the separate passes). */
VQVAE_API int vqvae_conv_forward_ep_f32(int kind, const float *x, const float *packed, const float *bias, int64_t B, int H,
                                        int W, int Cin, int Cout, int flags, const float *addend, const float *mask,
                                        float *y, vqvae_stream_t stream);
VQVAE_API int vqvae_conv_in_forward_ep_f32(const float *x_nchw, const float *packed, const float *bias, int64_t B, int H,
                                           int W, int Cin, int Cout, int flags, const float *mask, float *y,
                                           vqvae_stream_t stream);

/* ------------------------------------------------------- GatedPixelCNN prior
 * pixelcnn/models.py: the masked convolutions run as im2col over their causal tap list followed by the 1x1 conv
 * (vqvae_conv_forward_f32, kind VQVAE_CONV_1x1); these are the memory-bound pieces around it, all on row-major
 * (B,H,W,C) activations, C % 4 == 0, pointers 16-byte aligned.
 *   gather_rows       out[i][:] = table[idx[i]][:]            nn.Embedding (:119-121; class_cond_embedding :68)
 *   im2col_rows       out[b,y,x,t*C+c] = x[b,y+dy[t],x+dx[t],c], 0 outside the map; dy/dx are HOST arrays, ntaps <= 32
 *   gated_activation  out[.., c] = tanh(a) * sigmoid(g), a = t1[.., c] (+ t2[.., c]) (+ cond[b][c]),
 *                     g = the same at channel dim + c (GatedActivation :21-27 with the sums of :71 / :77);
 *                     t1, t2 (B,HW,2*dim), cond (B,2*dim), out (B,HW,dim); t2 and cond may be NULL
 *   add               out = a + b                              the horizontal residual (:79)                 */
VQVAE_API int vqvae_gather_rows_f32(const int64_t *idx, const float *table, int64_t n, int C, int rows,
                                    float *out, vqvae_stream_t stream);
VQVAE_API int vqvae_im2col_rows_f32(const float *x, int64_t B, int H, int W, int C, int ntaps,
                                    const int8_t *dy, const int8_t *dx, float *out, vqvae_stream_t stream);
/* A masked convolution WITHOUT the im2col pass (round 4): a stride-1 conv over an explicit list of ntaps <= 16 taps, tap t reading
 * the input at (y + dy[t], x + dx[t]) (zero outside the map; |dy|, |dx| <= 7).  w is (Cout, Cin, ntaps) contiguous -- a masked
 * conv's own (Cout, Cin, kh, kw) tensor when the list enumerates (ky, kx) in row-major order (GatedMaskedConv2d's vertical
 * (k//2+1) x k and horizontal 1 x (k//2+1) stacks, pixelcnn/models.py:45-58).  dy / dx are HOST arrays.  Same kernels, packed
 * image and flags as vqvae_conv_forward_f32 (on 8x8 maps with 32-channel multiples: the tile-resident kernel, every tap from
 * one parked image).                                                                                                   */
VQVAE_API size_t vqvae_conv_taps_packed_bytes(int ntaps, int Cin, int Cout);
VQVAE_API int vqvae_conv_taps_pack_f32(const float *w, int ntaps, const int8_t *dy, const int8_t *dx, int Cin, int Cout,
                                       float *packed, vqvae_stream_t stream);
VQVAE_API int vqvae_conv_taps_forward_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                                          int Cin, int Cout, int ntaps, const int8_t *dy, const int8_t *dx, int flags,
                                          float *y, vqvae_stream_t stream);
/* the same with vqvae_conv_forward_ep_f32's epilogue, y = (mask > 0) ? conv + addend : 0: a tap list longer than 16 runs as a chain of
 * launches over slices of it, each adding to the previous one's result (the first GatedMaskedConv2d's 4 x 7 vertical stack)      */
VQVAE_API int vqvae_conv_taps_forward_ep_f32(const float *x, const float *packed, const float *bias, int64_t B, int H, int W,
                                             int Cin, int Cout, int ntaps, const int8_t *dy, const int8_t *dx, int flags,
                                             const float *addend, const float *mask, float *y, vqvae_stream_t stream);
VQVAE_API int vqvae_gated_activation_f32(const float *t1, const float *t2, const float *cond, int64_t B,
                                         int HW, int dim, float *out, vqvae_stream_t stream);
VQVAE_API int vqvae_add_f32(const float *a, const float *b, int64_t n, float *out, vqvae_stream_t stream);

/* ------------------------------------------------------------------- whole path
 * models/vqvae.py:29-44 as ONE call: Encoder (models/encoder.py:28-43) -> pre_quantization_conv (models/vqvae.py:33)
 * -> VectorQuantizer (models/quantizer.py:45-76) -> Decoder (models/decoder.py:27-39).  The caller owns every buffer:
 * the packed weights (built once per weight version with vqvae_weights_pack_f32), the activation workspace
 * (vqvae_workspace_bytes) and, optionally, a persistent quantizer workspace that keeps the prepared codebook images
 * across calls.  Nothing is allocated or synchronised; every kernel goes to `stream`.                                 */
typedef struct VqvaeDims {          /* VQVAE(h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, beta), vqvae.py:11-12 */
    int h_dim, res_h_dim, n_res_layers, n_embeddings, embedding_dim, in_ch;
    float beta;
} VqvaeDims;

typedef struct VqvaeRawWeights {    /* device pointers to the parameters in torch's layouts; names = state_dict keys */
    const float *enc0_w, *enc0_b;           /* encoder.conv_stack.0  (h/2, in_ch, 4, 4), (h/2)                      */
    const float *enc2_w, *enc2_b;           /* encoder.conv_stack.2  (h, h/2, 4, 4), (h)                            */
    const float *enc4_w, *enc4_b;           /* encoder.conv_stack.4  (h, h, 3, 3), (h)                              */
    const float *enc_res_w1, *enc_res_w2;   /* encoder.conv_stack.5.stack.0.res_block.{1,3}  (Rh, h, 3, 3), (h, Rh, 1, 1) */
    const float *pre_w, *pre_b;             /* pre_quantization_conv (D, h, 1, 1), (D)                              */
    const float *codebook;                  /* vector_quantization.embedding.weight (K, D)                          */
    const float *dec0_w, *dec0_b;           /* decoder.inverse_conv_stack.0  (D, h, 3, 3), (h)   [ConvTranspose2d]  */
    const float *dec_res_w1, *dec_res_w2;   /* decoder.inverse_conv_stack.1.stack.0.res_block.{1,3}                 */
    const float *dec2_w, *dec2_b;           /* decoder.inverse_conv_stack.2  (h, h/2, 4, 4), (h/2)                  */
    const float *dec4_w, *dec4_b;           /* decoder.inverse_conv_stack.4  (h/2, in_ch, 4, 4), (in_ch)            */
} VqvaeRawWeights;

typedef struct VqvaeWeights {       /* what the whole-path entry points consume: packed images + biases + codebook */
    VqvaeDims dims;
    const float *enc0, *enc0_b, *enc2, *enc2_b, *enc4, *enc4_b, *enc_res_w1, *enc_res_w2, *pre, *pre_b, *codebook;
    const float *dec0, *dec0_b, *dec_res_w1, *dec_res_w2, *dec2, *dec2_b, *dec4, *dec4_b;
} VqvaeWeights;

/* Which product scheme do these WEIGHTS call for?  (round 5)  The default two-term fp16 scheme carries one power-of-two scale per output
 * channel of every weight tensor and one per image of every activation map; a layer whose INPUT channels still differ by many binades
 * after that normalisation pairs its largest weights with activations far below their image's maximum, whose second fp16 term
 * underflows -- the regime in which the default path is 100x further from fp64 than the reference's fp32.  That is a
 * property of the checkpoint: this call measures, per layer, log2(max_c r[c] / min_c r[c]) with r[c] = max over (o, taps) of
 * |w[o, c]| / max |w[o, ., .]| (spread_log2_host: 11 floats in VqvaeRawWeights' order of the weight tensors, may be NULL) and sets
 * *recommended_flags to VQVAE_FWD_CONV_BF16_SPLIT when any layer behind the first exceeds VQVAE_RANGE_SPREAD_LIMIT_LOG2 binades, else 0.
 * OR it into vqvae_forward_f32's vq_flags / the _ex_ entries' flags (the Python layer and integration/vqvae_hip_stub.py do, once per
 * weight version).  Launches 11 one-workgroup kernels on `stream` and SYNCHRONISES it; scratch_device: >= 64 bytes of device memory. */
#define VQVAE_RANGE_SPREAD_LIMIT_LOG2 10.0f
VQVAE_API int vqvae_weights_range_check_f32(const VqvaeDims *dims, const VqvaeRawWeights *raw, float *spread_log2_host,
                                            int *recommended_flags, void *scratch_device, size_t scratch_bytes, vqvae_stream_t stream);

/* Bytes of the one buffer that holds every packed weight image (0: unsupported dims). */
VQVAE_API size_t vqvae_weights_packed_bytes(const VqvaeDims *dims);
/* Packs every layer into `packed` and fills `out` (its bias / codebook pointers alias `raw`'s: keep those alive). */
VQVAE_API int vqvae_weights_pack_f32(const VqvaeDims *dims, const VqvaeRawWeights *raw, void *packed, size_t packed_bytes,
                                     VqvaeWeights *out, vqvae_stream_t stream);
/* Bytes of activation workspace for a (B, in_ch, H, W) batch (H, W multiples of 4; 0: unsupported). */
VQVAE_API size_t vqvae_workspace_bytes(const VqvaeDims *dims, int64_t B, int H, int W);
VQVAE_API size_t vqvae_workspace_ze_offset(const VqvaeDims *dims, int64_t B, int H, int W);   /* bytes from the workspace's start to z_e (0: unsupported shape) */

/* ResidualStack.forward (models/residual.py:47-51): n_layers applications of the SAME layer, then F.relu if
 * VQVAE_CONV_RELU_OUT; VQVAE_CONV_RELU_IN applies the first layer's in-place ReLU on read.  x, y, tmp row-major
 * (B,H,W,C); the result is in y; tmp (same size) is needed when n_layers > 1; x is not modified.                    */
VQVAE_API int vqvae_resstack_f32(const float *packed_w1, const float *packed_w2, const float *x, int64_t B, int H, int W,
                                 int C, int Rh, int n_layers, int flags, float *y, float *tmp, vqvae_stream_t stream);

/* Encoder + pre_quantization_conv: x (B,in_ch,H,W) NCHW -> z_e (B,H/4,W/4,D) ROW-MAJOR (the layout vqvae_vq_forward_f32
 * takes with VQVAE_VQ_ROWMAJOR).  workspace: vqvae_workspace_bytes(dims, B, H, W) is always enough; the entry itself needs the two
 * activation buffers, plus -- where h_dim is not 32 / 64 / 128 or res_h_dim > 32 (the generic residual path) -- one hidden map of
 * B * H/4 * W/4 * res_h_dim floats, and uses room beyond that for the per-image maxima (without them every layer measures its own).
 * On 32x32 RGB images with h_dim 128 and two residual layers (the reference's defaults) the encoder is two launches
 * (models/encoder.py:29-34, then :35-38 + models/vqvae.py:33) and the decoder two (models/decoder.py:28-30, :31-35): no
 * intermediate map is written inside a launch; other shapes run layer by layer through the kernels of the per-layer
 * entry points above.                                                                                                  */
VQVAE_API int vqvae_encoder_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, float *z_e,
                                void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
/* Decoder: z_q (B,h,w,D) row-major -> x_hat (B,in_ch,4h,4w) NCHW. */
VQVAE_API int vqvae_decoder_f32(const VqvaeWeights *w, const float *z_q, int64_t B, int h, int w_, float *x_hat,
                                void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
/* The same two with a product-scheme selector (flags: 0, VQVAE_FWD_CONV_BF16_SPLIT or VQVAE_FWD_CONV_EXACT_FP32, below). */
VQVAE_API int vqvae_encoder_ex_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int flags, float *z_e,
                                   void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
VQVAE_API int vqvae_decoder_ex_f32(const VqvaeWeights *w, const float *z_q, int64_t B, int h, int w_, int flags, float *x_hat,
                                   void *workspace, size_t workspace_bytes, vqvae_stream_t stream);

/* Whole-path product scheme (round 4), OR-ed into vqvae_forward_f32's vq_flags / the flags of the _ex_ entries above:
 *   default (neither bit)       two-term fp16 products (3 MFMAs per fp32 product, <= 2^-21 relative per product) with one
 *                               power-of-two scale per OUTPUT CHANNEL of every weight tensor and one per image for the
 *                               activations; on the reference's default shapes the four fused kernels;
 *   VQVAE_FWD_CONV_BF16_SPLIT   every layer through the per-layer kernels with three-term bf16 products (6 MFMAs per fp32
 *                               product, <= 3 * 2^-24 relative per product, fp32's exponent range: no operand scales at all);
 *   VQVAE_FWD_CONV_EXACT_FP32   every layer through the exact-fp32 MFMA kernels (v_mfma_f32_32x32x2_f32: fp32 products,
 *                               fp32 accumulation -- the reference's own arithmetic up to summation order).
 * The two selectors are the escape hatch for callers whose data defeats the fp16 scheme's range assumptions (DESIGN.md section 5,
 * "error bound per output channel") and the honest side-by-side legs of bench.py.  Both at once: VQVAE_ERR_UNSUPPORTED.   */
#define VQVAE_FWD_CONV_BF16_SPLIT 0x1000
#define VQVAE_FWD_CONV_EXACT_FP32 0x2000
/* Test hook (round 5): on the default shapes' fused path -- where the encoder's last kernel quantizes a z_e that never leaves the chip --
 * vqvae_forward_f32 ALSO writes the very z_e rows that kernel quantizes, row-major (B*H/4*W/4, D), at vqvae_workspace_ze_offset() of the
 * workspace (a separate instance of the kernel; idx must be given).  tests/test_model_gpu.py feeds those bits to the CPU oracle.
 * Ignored where the quantizer runs as its own launch (z_e is in the workspace there anyway).                                          */
#define VQVAE_FWD_DEBUG_ZE        0x4000

/* VQVAE.forward: x -> (embedding_loss, x_hat, perplexity) (models/vqvae.py:44); idx (B*H/4*W/4 int64) is optional.
 * vq_flags: VQVAE_VQ_CODEBOOK_PREPARED (only meaningful with a persistent vq_workspace of vqvae_vq_workspace_bytes),
 * VQVAE_VQ_EXACT_SWEEP / _BF16_FILTER, VQVAE_FWD_CONV_BF16_SPLIT / _EXACT_FP32.  vq_workspace may
 * be NULL (then the codebook images are rebuilt inside `workspace` on every call).                                                                       */
VQVAE_API int vqvae_forward_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int vq_flags,
                                float *x_hat, float *loss, float *perplexity, int64_t *idx, void *workspace,
                                size_t workspace_bytes, void *vq_workspace, size_t vq_workspace_bytes,
                                vqvae_stream_t stream);

/* The index wire format as single entry points -- what the PixelCNN prior is trained on and sampled into:
 *   vqvae_encode_f32   x (B,C,H,W) -> min_encoding_indices (B*H/4*W/4 int64), README.md:56 / visualization.ipynb:84-90 (encode_data);
 *   vqvae_decode_f32   indices -> x_hat (B,C,4h,4w), visualization.ipynb:358-365 / 642-650 (one-hot @ embedding -> decoder).
 * On the default shapes' fused path (32x32 images, h_dim 128, two residual layers, K a multiple of 128 up to 1024, D = 64) neither
 * touches a latent map: the encoder's last kernel quantizes its own z_e and writes 512 bytes of indices per image (no z_e, no z_q:
 * 12 KiB in, 0.5 KiB out), and the decoder's first kernel takes each latent pixel's row straight from the codebook (0.5 KiB in, 12 KiB
 * out).  Other shapes run encoder -> stand-alone quantizer / row gather -> decoder through the workspace.  Indices equal
 * vqvae_forward_f32's bit for bit; x_hat equals the decoder's on the same z_q bit for bit.  An index outside [0, K) never reads the
 * codebook: that latent pixel enters the first conv as NaN on EVERY path (fused gather and workspace gather alike) and -- the fused
 * ReLUs keep a NaN as nn.ReLU does -- shows as NaN pixels of its image's x_hat; no other image is affected.  The Python layer
 * validates foreign indices on the host and raises, as the reference's scatter does.  workspace: vqvae_workspace_bytes(dims, B, H, W) (decode: H = 4h, W = 4w); vq_workspace /
 * vq_flags as for vqvae_forward_f32; decode's flags: VQVAE_FWD_CONV_BF16_SPLIT / _EXACT_FP32 or 0.                                   */
VQVAE_API int vqvae_encode_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int vq_flags, int64_t *idx,
                               void *workspace, size_t workspace_bytes, void *vq_workspace, size_t vq_workspace_bytes,
                               vqvae_stream_t stream);
VQVAE_API int vqvae_decode_f32(const VqvaeWeights *w, const int64_t *idx, int64_t B, int h, int w_, int flags, float *x_hat,
                               void *workspace, size_t workspace_bytes, vqvae_stream_t stream);

/* The same step in PARTS, for callers that spread a large batch over several streams (the default shapes' path only: 32x32
 * images, h_dim 128, two residual layers, K a multiple of 128 up to 1024, D = 64; VQVAE_ERR_UNSUPPORTED otherwise -- use vqvae_forward_f32):
 *   vqvae_forward_begin_f32   codebook images + cleared histogram, on `stream`;
 *   vqvae_forward_part_f32    images [b0, b0 + Bc) of the batch of B, on ANY stream ordered behind the begin (b0 and every Bc
 *                             but the last multiples of 64); x, x_hat, idx and both workspaces are the WHOLE batch's, as
 *                             vqvae_forward_f32 would get them; parts must not overlap;
 *   vqvae_forward_end_f32     loss and perplexity of the whole batch, on a stream ordered behind every part.
 * Results are bit-identical to one vqvae_forward_f32 call.  Why: with the four kernels of the whole batch one after the other,
 * 10-15 % of every kernel's workgroup slots stand empty while it ramps up and while its last workgroups finish; kernels of
 * different parts on different streams can fill those ends -- measured between -7 % and +9 % per step depending on the box
 * (profiles/r03_notes.txt section 11), which is why the Python host layer does not do it by default.
 * The library creates no streams or events: ordering between the streams is the caller's (hipStreamWaitEvent / torch
 * wait_stream).
 * Checked on the host since round 5 (a record per workspace pointer, opened by begin and closed by end; VQVAE_ERR_SHAPE on violation):
 * every part repeats begin's B, H, W, flags and quantizer workspace; parts do not overlap; end finds [0, B) covered exactly once (a part
 * leaves one loss partial per four images and adds to the histogram atomically: a gap or a repeat would give wrong loss / perplexity).
 * What stays the caller's, because no launch-time check can see it: begin / parts / end are ORDERED on their streams as described, and
 * every buffer stays alive, untouched by other work, until the stream of `end` has passed it.  With vq_workspace == NULL the codebook
 * images live inside `workspace`, prepared by begin -- parts never prepare them. */
VQVAE_API int vqvae_forward_begin_f32(const VqvaeWeights *w, int64_t B, int H, int W, int vq_flags, void *workspace,
                                      size_t workspace_bytes, void *vq_workspace, size_t vq_workspace_bytes,
                                      vqvae_stream_t stream);
VQVAE_API int vqvae_forward_part_f32(const VqvaeWeights *w, const float *x, int64_t B, int64_t b0, int64_t Bc, int H, int W,
                                     int vq_flags, float *x_hat, int64_t *idx, void *workspace, size_t workspace_bytes,
                                     void *vq_workspace, size_t vq_workspace_bytes, vqvae_stream_t stream);
VQVAE_API int vqvae_forward_end_f32(const VqvaeWeights *w, int64_t B, int H, int W, float *loss, float *perplexity,
                                    void *workspace, size_t workspace_bytes, vqvae_stream_t stream);
/* Host-side state of the step in parts (the only state the library keeps): one record per `workspace` pointer in a process-wide table,
 * opened by begin, closed by end.  A part whose launches fail gives its claim back (it may be retried); a step that will not be
 * finished is dropped with vqvae_forward_abort_f32(workspace) -- otherwise its record stays until the next begin on the same pointer
 * replaces it (a freed and reused pointer would inherit it only until that begin).  Always VQVAE_OK. */
VQVAE_API int vqvae_forward_abort_f32(void *workspace);

#ifdef __cplusplus
}
#endif
#endif /* VQVAE_HIP_H */
