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

#define VQVAE_HIP_ABI_VERSION 1

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

/* ---------------------------------------------------------------- quantizer */

/* flags for vqvae_vq_forward_f32 */
#define VQVAE_VQ_NCHW          0x0  /* z_e / z_q are (B,D,H,W): the module boundary   */
#define VQVAE_VQ_ROWMAJOR      0x1  /* z_e / z_q are (B,H,W,D) = (N,D) rows (internal) */
#define VQVAE_VQ_CODEBOOK_PREPARED 0x2 /* workspace already holds this codebook's image
                                        (a previous call with the same codebook, K, D
                                        and workspace): skip the prepare kernel        */

/* Bytes of workspace vqvae_vq_forward_f32 needs for n_rows = B*H*W latent rows. */
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
 * Supported: D in {32, 64, 128, 256}, 1 <= K <= 16384.
 */
VQVAE_API int vqvae_vq_forward_f32(const float *z_e, const float *codebook,
                         int64_t B, int D, int H, int W, int K, float beta, int flags,
                         float *z_q, int64_t *idx, int32_t *hist,
                         float *loss, float *perplexity,
                         void *workspace, size_t workspace_bytes, vqvae_stream_t stream);

/* min_encodings, the (N,K) fp32 one-hot (models/quantizer.py:55-57).  Optional:
 * VQVAE.forward discards it (models/vqvae.py:34); N*K must fit in int64.       */
VQVAE_API int vqvae_vq_onehot_f32(const int64_t *idx, int64_t N, int K, float *onehot,
                        vqvae_stream_t stream);

/* indices -> z_q (B,D,H,W): the one-hot @ embedding.weight -> view -> permute
 * sequence of the notebook's generate_samples (visualization.ipynb:358-365).  */
VQVAE_API int vqvae_vq_decode_indices_f32(const int64_t *idx, const float *codebook,
                                int64_t B, int D, int H, int W, int K,
                                float *z_q, vqvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQVAE_HIP_H */
