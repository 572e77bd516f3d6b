// Whole-path entry points of the C ABI (include/vqvae_hip.h): models/vqvae.py:29-44 = Encoder (models/encoder.py:28-43)
// -> pre_quantization_conv (models/vqvae.py:33) -> VectorQuantizer (models/quantizer.py:45-76) -> Decoder
// (models/decoder.py:27-39) as ONE call over caller-owned buffers.  These functions only sequence the per-layer entry
// points of this library on the caller's stream and carve the caller's workspace; they allocate nothing and never
// synchronise.  A C / C++ caller needs nothing from the Python package.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace vqvae {
namespace {

// Host-side bookkeeping of a step in PARTS (vqvae_forward_begin / part / end), keyed by the caller's workspace pointer (round 5; ADVICE
// r3 item 5): begin opens a record, every part must repeat begin's B / H / W / flags / quantizer workspace and
// claim images no other part of the step has claimed, end must find [0, B) covered exactly and closes the record.  What stays the
// caller's: the ORDER of the streams and the lifetime of the buffers -- nothing a launch-time check can see.
struct PartsRecord {
    int64_t B;
    int H, W, flags;
    const void *vqws;
    std::vector<std::pair<int64_t, int64_t>> claimed;
};
std::mutex g_parts_mu;
std::map<const void *, PartsRecord> g_parts;

constexpr int kVqFormFlags = VQVAE_VQ_UNITS64_8WAVES | VQVAE_VQ_UNITS32_16WAVES | VQVAE_VQ_UNITS32_8WAVES;

struct Carve {
    char *p;
    size_t left;
    bool ok = true;
    float *f32(size_t n) { return reinterpret_cast<float *>(raw(n * sizeof(float))); }
    void *raw(size_t bytes) {
        bytes = align_up(bytes, 256);
        if (bytes > left) { ok = false; return nullptr; }
        void *r = p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

bool dims_ok(const VqvaeDims *d) {
    return d && d->h_dim >= 8 && d->h_dim % 8 == 0 && d->res_h_dim >= 1 && d->n_res_layers >= 0 && d->n_embeddings >= 1 &&
           d->embedding_dim >= 1 && (d->in_ch == 1 || d->in_ch == 3 || d->in_ch == 4);
}

// elements of the largest row-major activation between two layers, and of one latent map
size_t act_elems(const VqvaeDims *d, int64_t B, int H, int W) {
    const size_t half = (size_t)B * (H / 2) * (W / 2) * (d->h_dim / 2);
    const size_t quarter = (size_t)B * (H / 4) * (W / 4) * d->h_dim;
    return half > quarter ? half : quarter;
}

// amax: NULL, or (n_layers + 1) arrays of B ints (-1 = not provided): [0] belongs to x, [i + 1] to layer i's output
int res_stack(const float *w1, const float *w2, const float *x, int64_t B, int H, int W, int C, int Rh, int n_layers,
              bool first_relu_in, bool final_relu, float *y, float *tmp, hipStream_t st, const float **out, int *amax = nullptr,
              const ResPairPost *post = nullptr, bool *post_done = nullptr, int conv_flags = 0, float *hid_scratch = nullptr) {
    // every layer's output feeds the next layer's in-place ReLU (residual.py:19) or the stack's final F.relu (:50), so
    // the producer applies it.  Layers run in PAIRS where the fused two-layer kernel applies (8x8 maps, two-term fp16
    // products; the intermediate map stays on chip), a trailing odd layer alone; buffers alternate so that the result of
    // the LAST step lands in y.  A pair may write over its own input (tmp == x in the encoder / decoder), a single layer
    // never has to.
    const float *cur = x;
    *out = x;
    const bool generic = !res_layer_fused_ok(C, Rh);      // widths outside the fused kernels: conv + conv + combine, no maxima
    if (generic && !hid_scratch) return VQVAE_ERR_UNSUPPORTED;
    if (generic) amax = nullptr;
    const bool pairs = !generic && n_layers >= 2 && res_pair_supported(H, W, C, Rh, conv_flags);
    const int nsteps = pairs ? n_layers / 2 + (n_layers & 1) : n_layers;
    int i = 0;
    for (int j = 0; j < nsteps; ++j) {
        const bool pair = pairs && i + 1 < n_layers;
        const int last = pair ? i + 1 : i;
        int flags = ((i == 0 && first_relu_in) ? VQVAE_CONV_RELU_IN : 0) | conv_flags;
        if (last < n_layers - 1 || final_relu) flags |= VQVAE_CONV_RELU_OUT;
        float *dst = ((nsteps - 1 - j) % 2 == 0) ? y : tmp;
        const int *ain = amax ? amax + (size_t)i * B : nullptr;
        int *aout = amax ? amax + (size_t)(last + 1) * B : nullptr;
        // the 1x1 conv behind the stack rides in the last step when that step is a fused pair
#ifdef VQVAE_NO_PAIR_POST        // A/B builds (tools/build_variant.py): keep the 1x1 conv as its own launch
        const bool with_post = false;
#else
        const bool with_post = pair && post && j == nsteps - 1 && res_pair_post_supported(C, post->Cout);
#endif
        const int rc = pair ? res_pair_forward_impl(cur, w1, w2, B, H, W, C, Rh, flags, dst, st, ain, aout, with_post ? post : nullptr)
                            : res_layer_forward_impl(cur, w1, w2, B, H, W, C, Rh, flags, dst, st, ain, aout, nullptr, hid_scratch);
        if (with_post && post_done) *post_done = true;
        if (rc != 0) return rc;
        cur = dst;
        i = last + 1;
    }
    *out = cur;
    return 0;
}

// hidden map of a residual layer outside the fused kernels' widths (generic path), else nothing
size_t res_hidden_bytes(const VqvaeDims *d, int64_t B, int H, int W) {
    if (d->n_res_layers < 1 || res_layer_fused_ok(d->h_dim, d->res_h_dim)) return 0;
    return align_up((size_t)B * (H / 4) * (W / 4) * d->res_h_dim * sizeof(float), 256);
}

// per-image activation maxima handed from layer to layer (two-term fp16 product path): one array of B ints per
// inter-layer activation of the encoder (conv_in, enc2, enc4, n residual layers) or the decoder (dec0, n residual layers,
// z_q where the decoder's first layer is a generic kernel, dec2)
size_t amax_bytes(const VqvaeDims *d, int64_t B) { return align_up((size_t)(4 + d->n_res_layers) * B * sizeof(int), 256); }

// vqvae_decode_f32 on shapes without the gathering decoder kernel: z_q[i][:] = codebook[idx[i]][:] for ANY row length, and -- the
// contract of the fused gather, include/vqvae_hip.h -- an index outside [0, K) never reads the codebook: its latent pixel becomes NaN
__global__ __launch_bounds__(256) void decode_gather_kernel(const long long *__restrict__ idx, const float *__restrict__ codebook,
                                                            long long rows, int D, int K, float *__restrict__ z_q) {
    const long long total = rows * D;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / D;
        const int c = (int)(e - i * D);
        const long long k = idx[i];
        z_q[e] = (k >= 0 && k < K) ? codebook[(size_t)k * D + c] : __builtin_nanf("");
    }
}

}  // namespace
}  // namespace vqvae

using namespace vqvae;

extern "C" {

size_t vqvae_weights_packed_bytes(const VqvaeDims *d) {
    if (!dims_ok(d)) return 0;
    const int h = d->h_dim, D = d->embedding_dim, Rh = d->res_h_dim;
    size_t n = 0;
    const size_t parts[] = {
        vqvae_conv_in_packed_bytes(d->in_ch, h / 2), vqvae_conv_packed_bytes(VQVAE_CONV_4x4_S2, h / 2, h),
        vqvae_conv_packed_bytes(VQVAE_CONV_3x3_S1, h, h), vqvae_conv_packed_bytes(VQVAE_CONV_3x3_S1, h, Rh),
        vqvae_conv_packed_bytes(VQVAE_CONV_1x1, Rh, h), vqvae_conv_packed_bytes(VQVAE_CONV_1x1, h, D),
        vqvae_conv_packed_bytes(VQVAE_CONVT_3x3_S1, D, h), vqvae_conv_packed_bytes(VQVAE_CONV_3x3_S1, h, Rh),
        vqvae_conv_packed_bytes(VQVAE_CONV_1x1, Rh, h), vqvae_conv_packed_bytes(VQVAE_CONVT_4x4_S2, h, h / 2),
        vqvae_convt_out_packed_bytes(h / 2, d->in_ch)};
    for (size_t b : parts) {
        if (b == 0) return 0;
        n += align_up(b, 256);
    }
    return n;
}

int vqvae_weights_pack_f32(const VqvaeDims *d, const VqvaeRawWeights *raw, void *packed, size_t packed_bytes,
                           VqvaeWeights *out, vqvae_stream_t stream) {
    if (!d || !raw || !packed || !out) return VQVAE_ERR_NULL;
    const size_t need = vqvae_weights_packed_bytes(d);
    if (need == 0) return VQVAE_ERR_UNSUPPORTED;
    if (packed_bytes < need) return VQVAE_ERR_WORKSPACE;
    const float *req[] = {raw->enc0_w, raw->enc0_b, raw->enc2_w, raw->enc2_b, raw->enc4_w, raw->enc4_b, raw->enc_res_w1,
                          raw->enc_res_w2, raw->pre_w, raw->pre_b, raw->codebook, raw->dec0_w, raw->dec0_b, raw->dec_res_w1,
                          raw->dec_res_w2, raw->dec2_w, raw->dec2_b, raw->dec4_w, raw->dec4_b};
    for (const float *q : req)
        if (!q) return VQVAE_ERR_NULL;
    const int h = d->h_dim, D = d->embedding_dim, Rh = d->res_h_dim;
    Carve c{static_cast<char *>(packed), packed_bytes};
    int rc;
#define PACK_CONV(dst, kind, w, Cin, Cout)                                                          \
    do {                                                                                            \
        float *buf = static_cast<float *>(c.raw(vqvae_conv_packed_bytes(kind, Cin, Cout)));         \
        if (!c.ok) return VQVAE_ERR_WORKSPACE;                                                      \
        if ((rc = vqvae_conv_pack_f32(kind, w, Cin, Cout, buf, stream)) != 0) return rc;            \
        out->dst = buf;                                                                             \
    } while (0)
    {
        float *buf = static_cast<float *>(c.raw(vqvae_conv_in_packed_bytes(d->in_ch, h / 2)));
        if (!c.ok) return VQVAE_ERR_WORKSPACE;
        if ((rc = vqvae_conv_in_pack_f32(raw->enc0_w, d->in_ch, h / 2, buf, stream)) != 0) return rc;
        out->enc0 = buf;
    }
    PACK_CONV(enc2, VQVAE_CONV_4x4_S2, raw->enc2_w, h / 2, h);
    PACK_CONV(enc4, VQVAE_CONV_3x3_S1, raw->enc4_w, h, h);
    PACK_CONV(enc_res_w1, VQVAE_CONV_3x3_S1, raw->enc_res_w1, h, Rh);
    PACK_CONV(enc_res_w2, VQVAE_CONV_1x1, raw->enc_res_w2, Rh, h);
    PACK_CONV(pre, VQVAE_CONV_1x1, raw->pre_w, h, D);
    PACK_CONV(dec0, VQVAE_CONVT_3x3_S1, raw->dec0_w, D, h);
    PACK_CONV(dec_res_w1, VQVAE_CONV_3x3_S1, raw->dec_res_w1, h, Rh);
    PACK_CONV(dec_res_w2, VQVAE_CONV_1x1, raw->dec_res_w2, Rh, h);
    PACK_CONV(dec2, VQVAE_CONVT_4x4_S2, raw->dec2_w, h, h / 2);
#undef PACK_CONV
    {
        float *buf = static_cast<float *>(c.raw(vqvae_convt_out_packed_bytes(h / 2, d->in_ch)));
        if (!c.ok) return VQVAE_ERR_WORKSPACE;
        if ((rc = vqvae_convt_out_pack_f32(raw->dec4_w, h / 2, d->in_ch, buf, stream)) != 0) return rc;
        out->dec4 = buf;
    }
    out->dims = *d;
    out->enc0_b = raw->enc0_b; out->enc2_b = raw->enc2_b; out->enc4_b = raw->enc4_b; out->pre_b = raw->pre_b;
    out->dec0_b = raw->dec0_b; out->dec2_b = raw->dec2_b; out->dec4_b = raw->dec4_b;
    out->codebook = raw->codebook;
    return VQVAE_OK;
}

// ---- round 5: which product scheme the WEIGHTS call for ("nothing at runtime tells a caller their data is in that regime")
// The default two-term fp16 scheme scales every weight tensor per OUTPUT channel and every activation map per image.  What it cannot
// absorb is a layer whose INPUT channels differ by many binades after that normalisation: the activations that meet the large
// weights are then tiny beside their image's maximum, their second fp16 term drops below fp16's window, and the terms that carry
// the result arrive with 11 bits (tests/hetero.py builds such checkpoints on purpose: 20 binades).  The spread is a property of the
// checkpoint, so it is measured ONCE per weight version: per layer, r[c] = max over (o, taps) of |w[o, c]| / max|w[o, .]|, spread =
// log2(max r / min r).  Default-initialised and ordinarily trained layers sit at 0-3 binades.
int vqvae_weights_range_check_f32(const VqvaeDims *d, const VqvaeRawWeights *raw, float *spread_log2_host, int *recommended_flags,
                                  void *scratch_device, size_t scratch_bytes, vqvae_stream_t stream) {
    if (!d || !raw || !recommended_flags || !scratch_device) return VQVAE_ERR_NULL;
    if (!dims_ok(d)) return VQVAE_ERR_UNSUPPORTED;
    if (scratch_bytes < 16 * sizeof(float)) return VQVAE_ERR_WORKSPACE;
    const int h = d->h_dim, D = d->embedding_dim, Rh = d->res_h_dim;
    struct Lay { const float *w; int Cout, Cin, taps; bool transposed; };
    const Lay lay[11] = {{raw->enc0_w, h / 2, d->in_ch, 16, false}, {raw->enc2_w, h, h / 2, 16, false}, {raw->enc4_w, h, h, 9, false},
                         {raw->enc_res_w1, Rh, h, 9, false},         {raw->enc_res_w2, h, Rh, 1, false},  {raw->pre_w, D, h, 1, false},
                         {raw->dec0_w, h, D, 9, true},               {raw->dec_res_w1, Rh, h, 9, false},  {raw->dec_res_w2, h, Rh, 1, false},
                         {raw->dec2_w, h / 2, h, 16, true},          {raw->dec4_w, d->in_ch, h / 2, 16, true}};
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *out = static_cast<float *>(scratch_device);
    for (int i = 0; i < 11; ++i) {
        if (!lay[i].w) return VQVAE_ERR_NULL;
        if (d->n_res_layers == 0 && (i == 3 || i == 4 || i == 7 || i == 8)) { weight_spread_impl(nullptr, 0, 0, 0, false, out + i, st); continue; }
        if (lay[i].Cout > 1024 || lay[i].Cin > 1024) return VQVAE_ERR_UNSUPPORTED;
        weight_spread_impl(lay[i].w, lay[i].Cout, lay[i].Cin, lay[i].taps, lay[i].transposed, out + i, st);
    }
    float host[11];
    hipError_t e = hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                  // (once per weight version: the decision is the host's)
    if (e != hipSuccess) return (int)e;
    float worst = 0.0f;
    for (int i = 1; i < 11; ++i) worst = host[i] > worst ? host[i] : worst;     // (layer 0 reads the image itself: its channels are the caller's)
    if (spread_log2_host) for (int i = 0; i < 11; ++i) spread_log2_host[i] = host[i];
    *recommended_flags = worst > VQVAE_RANGE_SPREAD_LIMIT_LOG2 ? VQVAE_FWD_CONV_BF16_SPLIT : 0;
    return VQVAE_OK;
}

size_t vqvae_workspace_ze_offset(const VqvaeDims *d, int64_t B, int H, int W) {
    if (vqvae_workspace_bytes(d, B, H, W) == 0) return 0;
    return 2 * align_up(act_elems(d, B, H, W) * sizeof(float), 256) + 2 * amax_bytes(d, B);      // (carve_forward's order)
}

size_t vqvae_workspace_bytes(const VqvaeDims *d, int64_t B, int H, int W) {
    if (!dims_ok(d) || B < 1 || H < 4 || W < 4 || H % 4 || W % 4) return 0;
    const size_t vq = vqvae_vq_workspace_bytes(B * (int64_t)(H / 4) * (W / 4), d->n_embeddings, d->embedding_dim);
    if (vq == 0) return 0;
    const size_t act = align_up(act_elems(d, B, H, W) * sizeof(float), 256);
    const size_t lat = align_up((size_t)B * (H / 4) * (W / 4) * d->embedding_dim * sizeof(float), 256);
    const size_t rows = (size_t)B * (H / 4) * (W / 4);
    return 2 * act + 2 * amax_bytes(d, B) + 2 * lat + align_up(rows * sizeof(int64_t), 256) +
           align_up((size_t)d->n_embeddings * sizeof(int32_t), 256) + align_up(vq, 256) + 256 + res_hidden_bytes(d, B, H, W);
}

int vqvae_resstack_f32(const float *packed_w1, const float *packed_w2, const float *x, int64_t B, int H, int W, int C,
                       int Rh, int n_layers, int flags, float *y, float *tmp, vqvae_stream_t stream) {
    if (!packed_w1 || !packed_w2 || !x || !y) return VQVAE_ERR_NULL;
    if (n_layers > 1 && !tmp) return VQVAE_ERR_NULL;
    if (n_layers < 1) return VQVAE_ERR_SHAPE;
    const float *res = nullptr;
    return res_stack(packed_w1, packed_w2, x, B, H, W, C, Rh, n_layers, flags & VQVAE_CONV_RELU_IN, flags & VQVAE_CONV_RELU_OUT,
                     y, tmp, static_cast<hipStream_t>(stream), &res);
}

// am_given: a maxima region (amax_bytes) the caller has already set to -1, or NULL to carve and initialise one here
// zero_buf / zero_n / zeroed: ints the last encoder kernel should clear for the quantizer behind it (its histogram);
// *zeroed tells the caller whether a kernel that can do so ran
// vq: quantize inside the last kernel (fused 32x32 path only; z_e is then NOT written and zero_buf is cleared by the FIRST kernel)
static int encoder_run(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, float *z_e, void *workspace,
                       size_t workspace_bytes, hipStream_t st, int *am_given, int *zero_buf = nullptr, int zero_n = 0,
                       bool *zeroed = nullptr, bool am_exclusive = false, const VqFuse *vq = nullptr, int cf = 0,
                       float *hid_given = nullptr, bool debug_ze = false) {
    // cf: 0 = the default product scheme (two-term fp16 where the kernels have it), VQVAE_CONV_BF16_SPLIT / VQVAE_CONV_EXACT_FP32 =
    // every layer through the per-layer kernels of that scheme (no fused kernels: they exist for the fp16 products only)
    if (!w || !x || !z_e || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    if (!dims_ok(d) || B < 1 || H < 4 || W < 4 || H % 4 || W % 4) return VQVAE_ERR_SHAPE;
    Carve c{static_cast<char *>(workspace), workspace_bytes};
    const size_t act = act_elems(d, B, H, W);
    float *a = c.f32(act), *b = c.f32(act);
    if (!c.ok) return VQVAE_ERR_WORKSPACE;
    // residual widths outside the fused kernels: the hidden map's scratch (given by the forward entry, else carved here -- BEFORE the
    // optional maxima, so that a short workspace loses only those: ADVICE r4)
    float *hid = hid_given;
    if (!hid && res_hidden_bytes(d, B, H, W)) {
        hid = static_cast<float *>(c.raw(res_hidden_bytes(d, B, H, W)));
        if (!c.ok) return VQVAE_ERR_WORKSPACE;
    }
    // optional: the per-image maxima (a workspace of the documented size has room; without them every consumer measures
    // its own image)
    int *am = am_given;
    if (!am) {
        am = static_cast<int *>(c.raw(amax_bytes(d, B)));
        // (no fill where every array that is read has a one-wave-per-image producer with plain stores: fused_c3_path)
        if (am && !am_exclusive && hipMemsetAsync(am, 0xFF, amax_bytes(d, B), st) != hipSuccess) am = nullptr;
    }
    int *am0 = am, *am1 = am ? am + B : nullptr, *am2 = am ? am + 2 * B : nullptr;     // conv_in, enc2, enc4 (+ residual layers)
    const int h = d->h_dim;
    int rc;
    // encoder.py:29-31, :32-34, :35-36 (+ the residual stack's first in-place ReLU, residual.py:19)
#ifndef VQVAE_NO_ENC_FRONT_FUSION    // A/B builds (tools/build_variant.py)
    // encoder.py:29-34 in ONE launch on 32x32 RGB images: the 16x16 x h/2 map between the two stride-2 convs is never written
    if (!cf && enc_front_supported(H, W, d->in_ch, h / 2, h)) {
        if ((rc = enc_front_forward_impl(x, w->enc0, w->enc0_b, w->enc2, w->enc2_b, B, H, W, d->in_ch, h / 2, h, b, st, am1,
                                         vq ? zero_buf : nullptr, vq ? zero_n : 0)) != 0) return rc;
    } else
#endif
    {
        if ((rc = conv_in_forward_impl(x, w->enc0, w->enc0_b, B, H, W, d->in_ch, h / 2, VQVAE_CONV_RELU_OUT | cf, a, st, am0)) != 0) return rc;
        if ((rc = conv_forward_impl(VQVAE_CONV_4x4_S2, a, w->enc2, w->enc2_b, B, H / 2, W / 2, h / 2, h, VQVAE_CONV_RELU_OUT | cf, b, st, am0, am1)) != 0) return rc;
    }
#ifndef VQVAE_NO_FRONT_FUSION    // A/B builds (tools/build_variant.py)
    // encoder.py:35-38 + vqvae.py:33 in ONE launch where the shapes allow (8x8 latent maps, h_dim 128, two residual layers):
    // 3x3 conv + ReLU, both residual layers and the pre-quantisation conv; none of the three intermediate maps is written
    if (!cf && d->n_res_layers == 2 && conv_res_pair_supported(VQVAE_CONV_3x3_S1, H / 4, W / 4, h, h, d->res_h_dim) &&
        res_pair_post_supported(h, d->embedding_dim)) {
        const ResPairPost post{w->pre, w->pre_b, d->embedding_dim, z_e, vq ? nullptr : zero_buf, vq ? 0 : zero_n, vq, vq && debug_ze};
        if (zeroed) *zeroed = zero_buf != nullptr;
        return conv_res_pair_forward_impl(VQVAE_CONV_3x3_S1, b, w->enc4, w->enc4_b, h, w->enc_res_w1, w->enc_res_w2, B, H / 4, W / 4, h,
                                          d->res_h_dim, VQVAE_CONV_RELU_OUT, nullptr, st, am1, nullptr, &post);
    }
#endif
    if ((rc = conv_forward_impl(VQVAE_CONV_3x3_S1, b, w->enc4, w->enc4_b, B, H / 4, W / 4, h, h, VQVAE_CONV_RELU_OUT | cf, a, st, am1, am2)) != 0) return rc;
    const float *t = a;
    const int *amt = am2;
    if (d->n_res_layers > 0) {                                                                  // encoder.py:37-38
        // the first layer must not write into its own input (a): with an even layer count the result comes back to a
        float *y = (d->n_res_layers & 1) ? b : a, *tmp = (d->n_res_layers & 1) ? a : b;
        // vqvae.py:33: the pre-quantisation 1x1 conv consumes the stack's output inside its last kernel where that is a
        // fused pair (8x8 maps, h_dim 128): the stack's output map is never written
        const ResPairPost post{w->pre, w->pre_b, d->embedding_dim, z_e};
        bool post_done = false;
        if ((rc = res_stack(w->enc_res_w1, w->enc_res_w2, a, B, H / 4, W / 4, h, d->res_h_dim, d->n_res_layers, false, true, y, tmp, st, &t, am2,
                            &post, &post_done, cf, hid)) != 0)
            return rc;
        if (post_done) return 0;
        if (am2) amt = am2 + (size_t)d->n_res_layers * B;
        if (!res_layer_fused_ok(h, d->res_h_dim)) amt = nullptr;          // the generic residual path publishes no maxima
    }
    // n_res_layers == 0: F.relu of an already ReLU'd tensor is the identity
    return conv_forward_impl(VQVAE_CONV_1x1, t, w->pre, w->pre_b, B, H / 4, W / 4, h, d->embedding_dim, cf, z_e, st, amt, nullptr);   // vqvae.py:33
}

// 32x32 RGB images, h_dim 128, two residual layers: the step is four fused conv kernels (+ the quantizer), every per-image
// maximum that is read has ONE producing wave that stores it plainly -- the maxima arrays need no fill
static bool fused_c3_path(const VqvaeDims *d, int H, int W, int cf = 0) {
    if (cf) return false;
#if defined(VQVAE_NO_FRONT_FUSION) || defined(VQVAE_NO_ENC_FRONT_FUSION) || defined(VQVAE_NO_DEC_TAIL_FUSION)
    (void)d; (void)H; (void)W;
    return false;
#else
    const int h = d->h_dim;
    return d->n_res_layers == 2 && enc_front_supported(H, W, d->in_ch, h / 2, h) &&
           conv_res_pair_supported(VQVAE_CONV_3x3_S1, H / 4, W / 4, h, h, d->res_h_dim) && res_pair_post_supported(h, d->embedding_dim) &&
           conv_res_pair_supported(VQVAE_CONVT_3x3_S1, H / 4, W / 4, d->embedding_dim, h, d->res_h_dim) &&
           dec_tail_supported(H / 4, W / 4, h, h / 2, d->in_ch);
#endif
}

// VQVAE_FWD_CONV_* -> the per-layer product-scheme flag; -1: both at once / unknown bits
static int conv_scheme(int flags) {
    const int sel = flags & (VQVAE_FWD_CONV_BF16_SPLIT | VQVAE_FWD_CONV_EXACT_FP32);
    if (sel == (VQVAE_FWD_CONV_BF16_SPLIT | VQVAE_FWD_CONV_EXACT_FP32)) return -1;
    return sel == VQVAE_FWD_CONV_BF16_SPLIT ? VQVAE_CONV_BF16_SPLIT : (sel == VQVAE_FWD_CONV_EXACT_FP32 ? VQVAE_CONV_EXACT_FP32 : 0);
}

int vqvae_encoder_ex_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int flags, float *z_e, void *workspace,
                         size_t workspace_bytes, vqvae_stream_t stream) {
    const int cf = conv_scheme(flags);
    if (cf < 0 || (flags & ~(VQVAE_FWD_CONV_BF16_SPLIT | VQVAE_FWD_CONV_EXACT_FP32))) return VQVAE_ERR_UNSUPPORTED;
    return encoder_run(w, x, B, H, W, z_e, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr, 0, nullptr,
                       w && dims_ok(&w->dims) && fused_c3_path(&w->dims, H, W, cf), nullptr, cf);
}

int vqvae_encoder_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, float *z_e, void *workspace,
                      size_t workspace_bytes, vqvae_stream_t stream) {
    return vqvae_encoder_ex_f32(w, x, B, H, W, 0, z_e, workspace, workspace_bytes, stream);
}

// where z_q's per-image maxima go inside a maxima region (amax_bytes) for the decoder's first layer
static int *zq_amax_slot(const VqvaeDims *d, int64_t B, int *am) { return am + (size_t)(2 + d->n_res_layers) * B; }
// does the decoder's first layer want them (the generic / halo kernels do; the 8x8-map kernel measures its image itself)
static bool zq_amax_wanted(const VqvaeDims *d, int h4, int w4) {
    return vqvae_conv_term_products(VQVAE_CONVT_3x3_S1, h4, w4, d->embedding_dim, d->h_dim, 0) != 3;
}

// zq_amax_given: the quantizer has already published z_q's maxima into the region's slot
// gather_idx: z_q = the codebook and the first kernel takes pixel p's row from code gather_idx[p] (decoder_gather_ok shapes only)
static bool decoder_gather_ok(const VqvaeDims *d, int h4, int w4, int cf) {
#ifndef VQVAE_NO_FRONT_FUSION
    return !cf && d->n_res_layers == 2 && conv_res_pair_supported(VQVAE_CONVT_3x3_S1, h4, w4, d->embedding_dim, d->h_dim, d->res_h_dim);
#else
    (void)d; (void)h4; (void)w4; (void)cf;
    return false;
#endif
}
static int decoder_run(const VqvaeWeights *w, const float *z_q, int64_t B, int h4, int w4, float *x_hat, void *workspace,
                       size_t workspace_bytes, hipStream_t st, int *am_given, bool am_exclusive = false, bool zq_amax_given = false,
                       int cf = 0, float *hid_given = nullptr, const int64_t *gather_idx = nullptr) {
    if (!w || !z_q || !x_hat || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    if (!dims_ok(d) || B < 1 || h4 < 1 || w4 < 1) return VQVAE_ERR_SHAPE;
    Carve c{static_cast<char *>(workspace), workspace_bytes};
    const size_t act = act_elems(d, B, 4 * h4, 4 * w4);
    float *a = c.f32(act), *b = c.f32(act);
    if (!c.ok) return VQVAE_ERR_WORKSPACE;
    float *hid = hid_given;
    if (!hid && res_hidden_bytes(d, B, 4 * h4, 4 * w4)) {
        hid = static_cast<float *>(c.raw(res_hidden_bytes(d, B, 4 * h4, 4 * w4)));
        if (!c.ok) return VQVAE_ERR_WORKSPACE;
    }
    int *am = am_given;                                                       // optional, see encoder_run
    if (!am) {
        am = static_cast<int *>(c.raw(amax_bytes(d, B)));
        if (am && !am_exclusive && hipMemsetAsync(am, 0xFF, amax_bytes(d, B), st) != hipSuccess) am = nullptr;
    }
    const int h = d->h_dim;
    int rc;
    // decoder.py:28-29 (+ the stack's first in-place ReLU), :30, :31-33, :34-35
    // z_q has no producer that publishes maxima: the 8x8-map kernel measures its image itself, the generic one gets them
    // from one more pass over z_q (array [2 + n_res_layers] of the region)
    if (gather_idx && !decoder_gather_ok(d, h4, w4, cf)) return VQVAE_ERR_UNSUPPORTED;
    int *amz = nullptr;
    if (am && !gather_idx && zq_amax_wanted(d, h4, w4)) {
        amz = zq_amax_slot(d, B, am);
        if (!zq_amax_given) act_absmax_impl(z_q, B, (long long)h4 * w4 * d->embedding_dim, amz, st);
    }
    const float *t = a;
    const int *amt = am;
#ifndef VQVAE_NO_FRONT_FUSION
    // decoder.py:28-30 in one launch where the shapes allow: conv-transpose 3x3 (+ the stack's first ReLU) and both residual layers
    const bool front = !cf && d->n_res_layers == 2 && conv_res_pair_supported(VQVAE_CONVT_3x3_S1, h4, w4, d->embedding_dim, h, d->res_h_dim);
#else
    const bool front = false;
#endif
    if (front) {
        int *aout = am ? am + (size_t)2 * B : nullptr;
        if ((rc = conv_res_pair_forward_impl(VQVAE_CONVT_3x3_S1, z_q, w->dec0, w->dec0_b, d->embedding_dim, w->dec_res_w1, w->dec_res_w2, B, h4,
                                             w4, h, d->res_h_dim, VQVAE_CONV_RELU_OUT, a, st, amz, aout, nullptr, gather_idx,
                                             d->n_embeddings)) != 0) return rc;
        amt = aout;
    } else if ((rc = conv_forward_impl(VQVAE_CONVT_3x3_S1, z_q, w->dec0, w->dec0_b, B, h4, w4, d->embedding_dim, h, VQVAE_CONV_RELU_OUT | cf, a, st, amz, am)) != 0) return rc;
    if (!front && d->n_res_layers > 0) {
        float *y = (d->n_res_layers & 1) ? b : a, *tmp = (d->n_res_layers & 1) ? a : b;
        if ((rc = res_stack(w->dec_res_w1, w->dec_res_w2, a, B, h4, w4, h, d->res_h_dim, d->n_res_layers, false, true, y, tmp, st, &t, am,
                            nullptr, nullptr, cf, hid)) != 0) return rc;
        if (am) amt = am + (size_t)d->n_res_layers * B;
        if (!res_layer_fused_ok(h, d->res_h_dim)) amt = nullptr;
    }
#ifndef VQVAE_NO_DEC_TAIL_FUSION    // A/B builds (tools/build_variant.py)
    // decoder.py:31-35 in ONE launch on 8x8 latent maps: the 16x16 x h/2 map between the two stride-2 transposed convs is never written
    if (!cf && dec_tail_supported(h4, w4, h, h / 2, d->in_ch))
        return dec_tail_forward_impl(t, w->dec2, w->dec2_b, w->dec4, w->dec4_b, B, h4, w4, h, h / 2, d->in_ch, x_hat, st, amt);
#endif
    float *u = (t == a) ? b : a;
    int *am_u = am ? am + (size_t)(3 + d->n_res_layers) * B : nullptr;       // dec2's output maxima for the last layer
    if ((rc = conv_forward_impl(VQVAE_CONVT_4x4_S2, t, w->dec2, w->dec2_b, B, h4, w4, h, h / 2, VQVAE_CONV_RELU_OUT | cf, u, st, amt, am_u)) != 0) return rc;
    return convt_out_forward_impl(u, w->dec4, w->dec4_b, B, 2 * h4, 2 * w4, h / 2, d->in_ch, cf, x_hat, st, am_u);
}

int vqvae_decoder_ex_f32(const VqvaeWeights *w, const float *z_q, int64_t B, int h4, int w4, int flags, float *x_hat, void *workspace,
                         size_t workspace_bytes, vqvae_stream_t stream) {
    const int cf = conv_scheme(flags);
    if (cf < 0 || (flags & ~(VQVAE_FWD_CONV_BF16_SPLIT | VQVAE_FWD_CONV_EXACT_FP32))) return VQVAE_ERR_UNSUPPORTED;
    return decoder_run(w, z_q, B, h4, w4, x_hat, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr,
                       w && dims_ok(&w->dims) && fused_c3_path(&w->dims, 4 * h4, 4 * w4, cf), false, cf);
}

int vqvae_decoder_f32(const VqvaeWeights *w, const float *z_q, int64_t B, int h4, int w4, float *x_hat, void *workspace,
                      size_t workspace_bytes, vqvae_stream_t stream) {
    return vqvae_decoder_ex_f32(w, z_q, B, h4, w4, 0, x_hat, workspace, workspace_bytes, stream);
}

}  // extern "C"

namespace {
// The whole-batch workspace of the forward entry points (vqvae_workspace_bytes), carved the same way by all of them
struct FwdWs {
    void *acts; size_t act, rows; int *am2; float *z_e, *z_q; int64_t *idx_ws; int32_t *hist; void *vqws; size_t vqws_bytes;
    float *hid;          // hidden map of the generic residual path (NULL where the fused kernels cover the widths)
};
// with_vq = false: the caller does not touch the quantizer's workspace (vqvae_forward_end_f32); everything in front of it
// lies where the other entry points put it
int carve_forward(const VqvaeDims *d, int64_t B, int H, int W, void *workspace, size_t workspace_bytes, void *vq_workspace,
                  size_t vq_workspace_bytes, int &vq_flags, FwdWs &f, bool with_vq = true) {
    const size_t need = vqvae_workspace_bytes(d, B, H, W);
    if (need == 0) return VQVAE_ERR_UNSUPPORTED;
    if (workspace_bytes < need) return VQVAE_ERR_WORKSPACE;
    Carve c{static_cast<char *>(workspace), workspace_bytes};
    f.act = act_elems(d, B, H, W);
    f.rows = (size_t)B * (H / 4) * (W / 4);
    f.acts = c.raw(2 * align_up(f.act * sizeof(float), 256));
    f.am2 = static_cast<int *>(c.raw(2 * amax_bytes(d, B)));          // encoder's and decoder's maxima: one fill for both
    f.z_e = c.f32(f.rows * d->embedding_dim);
    f.z_q = c.f32(f.rows * d->embedding_dim);
    f.idx_ws = static_cast<int64_t *>(c.raw(f.rows * sizeof(int64_t)));
    f.hist = static_cast<int32_t *>(c.raw((size_t)d->n_embeddings * sizeof(int32_t)));
    const size_t vqb = vqvae_vq_workspace_bytes((int64_t)f.rows, d->n_embeddings, d->embedding_dim);
    f.vqws = vq_workspace;
    f.vqws_bytes = vq_workspace_bytes;
    if (!f.vqws && with_vq) {                          // no persistent codebook workspace: use (and re-prepare) ours
        f.vqws = c.raw(vqb);
        f.vqws_bytes = vqb;
        vq_flags &= ~VQVAE_VQ_CODEBOOK_PREPARED;
    }
    f.hid = res_hidden_bytes(d, B, H, W) ? static_cast<float *>(c.raw(res_hidden_bytes(d, B, H, W))) : nullptr;
    return c.ok ? VQVAE_OK : VQVAE_ERR_WORKSPACE;
}
}  // namespace

extern "C" {

int vqvae_forward_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int vq_flags, float *x_hat,
                      float *loss, float *perplexity, int64_t *idx, void *workspace, size_t workspace_bytes,
                      void *vq_workspace, size_t vq_workspace_bytes, vqvae_stream_t stream) {
    if (!w || !x || !x_hat || !loss || !perplexity || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    FwdWs f;
    {
        const int crc = carve_forward(d, B, H, W, workspace, workspace_bytes, vq_workspace, vq_workspace_bytes, vq_flags, f);
        if (crc != VQVAE_OK) return crc;
    }
    const size_t act = f.act, rows = f.rows;
    void *acts = f.acts;
    int *am2 = f.am2;
    float *z_e = f.z_e, *z_q = f.z_q;
    int64_t *idx_ws = f.idx_ws;
    int32_t *hist = f.hist;
    void *vqws = f.vqws;
    const size_t vqws_bytes = f.vqws_bytes;
    const size_t acts_bytes = 2 * align_up(act * sizeof(float), 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // two fill launches per step saved on the fused 32x32 path: the maxima need none there, and the quantizer's histogram
    // is cleared by the encoder's last kernel
    const int cf = conv_scheme(vq_flags);
    if (cf < 0) return VQVAE_ERR_UNSUPPORTED;
    const bool fused = fused_c3_path(d, H, W, cf);
    if (!fused && hipMemsetAsync(am2, 0xFF, 2 * amax_bytes(d, B), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
    int *am_dec = reinterpret_cast<int *>(reinterpret_cast<char *>(am2) + amax_bytes(d, B));
    int rc;
    bool hist_zeroed = false;
    // (a forced launch form of the stand-alone quantizer -- VQVAE_VQ_UNITS* -- means the stand-alone quantizer: ADVICE r4)
    if (fused && !(vq_flags & kVqFormFlags) && vq_fuse_ok(d->n_embeddings, d->embedding_dim, B, vq_flags)) {
        // vqvae.py:31-34 in TWO launches: the encoder's last kernel quantizes its own z_e (never written); the codebook's
        // images are prepared first, the histogram is cleared by the encoder's first kernel, loss / perplexity by the finalize
        if ((rc = vq_prepare_impl(w->codebook, d->n_embeddings, d->embedding_dim, vq_flags, vqws, vqws_bytes, st)) != 0) return rc;
        VqFuse vf = vq_fuse_args(w->codebook, d->n_embeddings, vqws, z_q, idx ? idx : idx_ws, hist);
        // one loss partial per workgroup of four images: they go where z_e would have gone (any batch size)
        vf.partials = reinterpret_cast<double *>(z_e);
        // VQVAE_FWD_DEBUG_ZE (tests): the fused kernel also writes the z_e rows it quantizes (vqvae_workspace_ze_offset); the
        // partials move to the workspace's index region, which the caller's own idx buffer leaves free
        const bool debug_ze = vq_flags & VQVAE_FWD_DEBUG_ZE;
        if (debug_ze && !idx) return VQVAE_ERR_NULL;
        if (debug_ze) vf.partials = reinterpret_cast<double *>(idx_ws);
        if ((rc = encoder_run(w, x, B, H, W, z_e, acts, acts_bytes, st, am2, hist, d->n_embeddings, &hist_zeroed, false, &vf, 0, nullptr,
                              debug_ze)) != 0) return rc;
        if ((rc = vq_finalize_impl(vf.partials, (int)((B + 3) / 4), hist, d->n_embeddings, (int64_t)rows, d->embedding_dim, d->beta, loss,
                                   perplexity, st)) != 0) return rc;
        return decoder_run(w, z_q, B, H / 4, W / 4, x_hat, acts, acts_bytes, st, am_dec);                         // :36
    }
    bool zq_amax_done = false;
    if ((rc = encoder_run(w, x, B, H, W, z_e, acts, acts_bytes, st, am2, fused ? hist : nullptr, d->n_embeddings, &hist_zeroed, false,
                          nullptr, cf, f.hid)) != 0)
        return rc;                                                                                              // vqvae.py:31-33
    if ((rc = vq_forward_impl(z_e, w->codebook, B, d->embedding_dim, H / 4, W / 4, d->n_embeddings, d->beta,
                              (vq_flags & (VQVAE_VQ_CODEBOOK_PREPARED | VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER |
                                           VQVAE_VQ_REMOVED_FLAGS | kVqFormFlags)) | VQVAE_VQ_ROWMAJOR,
                              z_q, idx ? idx : idx_ws, hist, loss, perplexity, vqws, vqws_bytes, stream, hist_zeroed,
                              zq_amax_wanted(d, H / 4, W / 4) ? zq_amax_slot(d, B, am_dec) : nullptr, &zq_amax_done)) != 0) return rc;   // :34
    return decoder_run(w, z_q, B, H / 4, W / 4, x_hat, acts, acts_bytes, st, am_dec, false, zq_amax_done, cf, f.hid);    // :36
}


// ---- the index wire format: x -> indices, indices -> x_hat ---------------------------------------------
// README.md:56 / visualization.ipynb:84-90 (encode_data: the (N, 1) int64 indices are what the PixelCNN prior trains on) and
// visualization.ipynb:358-365 (generate_samples: one-hot @ embedding -> decoder).  On the default shapes' fused path neither
// direction touches a latent map: the encoder's last kernel quantizes its own z_e and writes 512 bytes of indices per image
// (no z_e, no z_q), the decoder's first kernel gathers the codebook rows itself.  Workspace: vqvae_workspace_bytes, as for the forward.
int vqvae_encode_f32(const VqvaeWeights *w, const float *x, int64_t B, int H, int W, int vq_flags, int64_t *idx, void *workspace,
                     size_t workspace_bytes, void *vq_workspace, size_t vq_workspace_bytes, vqvae_stream_t stream) {
    if (!w || !x || !idx || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    FwdWs f;
    {
        const int crc = carve_forward(d, B, H, W, workspace, workspace_bytes, vq_workspace, vq_workspace_bytes, vq_flags, f);
        if (crc != VQVAE_OK) return crc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cf = conv_scheme(vq_flags);
    if (cf < 0) return VQVAE_ERR_UNSUPPORTED;
    const size_t acts_bytes = 2 * align_up(f.act * sizeof(float), 256);
    const bool fused = fused_c3_path(d, H, W, cf);
    if (!fused && hipMemsetAsync(f.am2, 0xFF, 2 * amax_bytes(d, B), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
    int rc;
    bool hist_zeroed = false;
    if (fused && !(vq_flags & kVqFormFlags) && vq_fuse_ok(d->n_embeddings, d->embedding_dim, B, vq_flags)) {
        if ((rc = vq_prepare_impl(w->codebook, d->n_embeddings, d->embedding_dim, vq_flags, f.vqws, f.vqws_bytes, st)) != 0) return rc;
        VqFuse vf = vq_fuse_args(w->codebook, d->n_embeddings, f.vqws, nullptr, idx, f.hist);      // no z_q: its stores are dropped
        vf.partials = reinterpret_cast<double *>(f.z_e);                // (the loss partials and the histogram are by-products nobody reads)
        return encoder_run(w, x, B, H, W, f.z_e, f.acts, acts_bytes, st, f.am2, f.hist, d->n_embeddings, &hist_zeroed, false, &vf);
    }
    // other shapes: encoder -> z_e (workspace) -> the stand-alone quantizer (z_q, loss and perplexity land in the workspace, unread)
    if ((rc = encoder_run(w, x, B, H, W, f.z_e, f.acts, acts_bytes, st, f.am2, fused ? f.hist : nullptr, d->n_embeddings, &hist_zeroed, false,
                          nullptr, cf, f.hid)) != 0) return rc;
    float *scal = reinterpret_cast<float *>(f.idx_ws);                  // (the workspace's index region is free: idx is the caller's)
    return vq_forward_impl(f.z_e, w->codebook, B, d->embedding_dim, H / 4, W / 4, d->n_embeddings, d->beta,
                           (vq_flags & (VQVAE_VQ_CODEBOOK_PREPARED | VQVAE_VQ_EXACT_SWEEP | VQVAE_VQ_BF16_FILTER | VQVAE_VQ_REMOVED_FLAGS |
                                        kVqFormFlags)) | VQVAE_VQ_ROWMAJOR,
                           f.z_q, idx, f.hist, scal, scal + 1, f.vqws, f.vqws_bytes, stream, hist_zeroed, nullptr, nullptr);
}

int vqvae_decode_f32(const VqvaeWeights *w, const int64_t *idx, int64_t B, int h4, int w4, int flags, float *x_hat, void *workspace,
                     size_t workspace_bytes, vqvae_stream_t stream) {
    if (!w || !idx || !x_hat || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    const int cf = conv_scheme(flags);
    if (cf < 0 || (flags & ~(VQVAE_FWD_CONV_BF16_SPLIT | VQVAE_FWD_CONV_EXACT_FP32))) return VQVAE_ERR_UNSUPPORTED;
    if (!dims_ok(d) || B < 1 || h4 < 1 || w4 < 1) return VQVAE_ERR_SHAPE;
    FwdWs f;
    int vq_flags = VQVAE_VQ_CODEBOOK_PREPARED;                          // (no quantizer workspace is touched here)
    {
        const int crc = carve_forward(d, B, 4 * h4, 4 * w4, workspace, workspace_bytes, workspace, 0, vq_flags, f, false);
        if (crc != VQVAE_OK) return crc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t acts_bytes = 2 * align_up(f.act * sizeof(float), 256);
    int *am_dec = reinterpret_cast<int *>(reinterpret_cast<char *>(f.am2) + amax_bytes(d, B));
    const bool fused = fused_c3_path(d, 4 * h4, 4 * w4, cf);
    if (!fused && hipMemsetAsync(am_dec, 0xFF, amax_bytes(d, B), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
    if (decoder_gather_ok(d, h4, w4, cf))
        return decoder_run(w, w->codebook, B, h4, w4, x_hat, f.acts, acts_bytes, st, am_dec, false, false, cf, f.hid, idx);
    // other shapes: the codebook rows of every position into the workspace's z_q (row-major), then the decoder
    {
        const long long total = (long long)f.rows * d->embedding_dim;
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(decode_gather_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                           reinterpret_cast<const long long *>(idx), w->codebook, (long long)f.rows, d->embedding_dim, d->n_embeddings,
                           f.z_q);
        const int lrc = (int)hipGetLastError();
        if (lrc != 0) return lrc;
    }
    return decoder_run(w, f.z_q, B, h4, w4, x_hat, f.acts, acts_bytes, st, am_dec, false, false, cf, f.hid);
}

// ---- the same step in PARTS (the default shapes' fused path only) -----------------------------------------------------------
// vqvae_forward_begin_f32 (codebook images, histogram cleared) -> vqvae_forward_part_f32 for disjoint image ranges, each on
// ANY stream ordered behind the begin -> vqvae_forward_end_f32 ordered behind all parts (loss, perplexity of the whole batch).
// Buffers and workspaces are the whole batch's, exactly vqvae_forward_f32's; results are bit-identical to it (a workgroup
// quantizes the same four images and leaves the same loss partial; the histogram is integer).  Why: run on several streams,
// the kernels of different parts fill each other's ramp-up and tail -- 10-15 % of a kernel's workgroup slots stand empty there
// when the four kernels of the whole batch run one after the other (profiles/r03_notes.txt sections 9, 11).
int vqvae_forward_begin_f32(const VqvaeWeights *w, int64_t B, int H, int W, int vq_flags, void *workspace, size_t workspace_bytes,
                            void *vq_workspace, size_t vq_workspace_bytes, vqvae_stream_t stream) {
    if (!w || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    FwdWs f;
    int rc = carve_forward(d, B, H, W, workspace, workspace_bytes, vq_workspace, vq_workspace_bytes, vq_flags, f);
    if (rc != VQVAE_OK) return rc;
    if (!fused_c3_path(d, H, W, conv_scheme(vq_flags)) || !vq_fuse_ok(d->n_embeddings, d->embedding_dim, B, vq_flags)) return VQVAE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = vq_prepare_impl(w->codebook, d->n_embeddings, d->embedding_dim, vq_flags, f.vqws, f.vqws_bytes, st)) != 0) return rc;
    if (hipMemsetAsync(f.hist, 0, (size_t)d->n_embeddings * sizeof(int32_t), st) != hipSuccess) return VQVAE_ERR_WORKSPACE;
    {
        std::lock_guard<std::mutex> lk(g_parts_mu);
        g_parts[workspace] = PartsRecord{B, H, W, vq_flags, vq_workspace, {}};      // (a begin without an end is simply replaced)
    }
    return VQVAE_OK;
}

int vqvae_forward_part_f32(const VqvaeWeights *w, const float *x, int64_t B, int64_t b0, int64_t Bc, int H, int W, int vq_flags,
                           float *x_hat, int64_t *idx, void *workspace, size_t workspace_bytes, void *vq_workspace,
                           size_t vq_workspace_bytes, vqvae_stream_t stream) {
    if (!w || !x || !x_hat || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    // whole 64-image blocks per part (the last part takes what is left): the parts' slices of the maxima region stay aligned
    if (B < 1 || b0 < 0 || Bc < 1 || b0 + Bc > B || b0 % 64 || (b0 + Bc < B && Bc % 64)) return VQVAE_ERR_SHAPE;
    FwdWs f;
    int rc = carve_forward(d, B, H, W, workspace, workspace_bytes, vq_workspace, vq_workspace_bytes, vq_flags, f);
    if (rc != VQVAE_OK) return rc;
    if (!fused_c3_path(d, H, W, conv_scheme(vq_flags)) || !vq_fuse_ok(d->n_embeddings, d->embedding_dim, B, vq_flags)) return VQVAE_ERR_UNSUPPORTED;
    {
        // this part against the step's record: same shapes / flags / quantizer workspace as begin, images nobody has claimed
        std::lock_guard<std::mutex> lk(g_parts_mu);
        auto it = g_parts.find(workspace);
        if (it == g_parts.end()) return VQVAE_ERR_SHAPE;                                   // no vqvae_forward_begin_f32 on this workspace
        PartsRecord &r = it->second;
        if (r.B != B || r.H != H || r.W != W || r.flags != vq_flags || r.vqws != vq_workspace) return VQVAE_ERR_SHAPE;
        for (const auto &c : r.claimed)
            if (b0 < c.second && c.first < b0 + Bc) return VQVAE_ERR_SHAPE;                // overlaps another part of this step
        r.claimed.emplace_back(b0, b0 + Bc);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t per_img = f.act / (size_t)B, lat = (size_t)(H / 4) * (W / 4);
    // this part's own two activation buffers and maxima regions inside the whole batch's
    char *acts_p = static_cast<char *>(f.acts) + 2 * (size_t)b0 * per_img * sizeof(float);
    const size_t acts_p_bytes = 2 * align_up((size_t)Bc * per_img * sizeof(float), 256);
    const size_t am_ints = (size_t)(4 + d->n_res_layers);
    int *am_p = f.am2 + 2 * am_ints * (size_t)b0;
    int *am_dec_p = reinterpret_cast<int *>(reinterpret_cast<char *>(am_p) + amax_bytes(d, Bc));
    float *z_q_p = f.z_q + (size_t)b0 * lat * d->embedding_dim;
    int64_t *idx_p = (idx ? idx : f.idx_ws) + (size_t)b0 * lat;
    VqFuse vf = vq_fuse_args(w->codebook, d->n_embeddings, f.vqws, z_q_p, idx_p, f.hist);
    vf.partials = reinterpret_cast<double *>(f.z_e) + b0 / 4;
    const float *x_p = x + (size_t)b0 * d->in_ch * H * W;
    rc = encoder_run(w, x_p, Bc, H, W, f.z_e, acts_p, acts_p_bytes, st, am_p, nullptr, 0, nullptr, false, &vf);
    if (rc == 0) rc = decoder_run(w, z_q_p, Bc, H / 4, W / 4, x_hat + (size_t)b0 * d->in_ch * H * W, acts_p, acts_p_bytes, st, am_dec_p);
    if (rc != 0) {
        // a part whose launches failed gives its claim back: the caller may retry [b0, b0 + Bc)
        std::lock_guard<std::mutex> lk(g_parts_mu);
        auto it = g_parts.find(workspace);
        if (it != g_parts.end()) {
            auto &cl = it->second.claimed;
            for (size_t i = 0; i < cl.size(); ++i)
                if (cl[i].first == b0 && cl[i].second == b0 + Bc) { cl.erase(cl.begin() + (long)i); break; }
        }
    }
    return rc;
}

// Drops the host-side record of a step in parts that will not be finished (a begin without an end otherwise stays in the process-wide
// table, keyed by the workspace pointer, until the next begin on that pointer replaces it).  Always VQVAE_OK.
int vqvae_forward_abort_f32(void *workspace) {
    std::lock_guard<std::mutex> lk(g_parts_mu);
    g_parts.erase(workspace);
    return VQVAE_OK;
}

int vqvae_forward_end_f32(const VqvaeWeights *w, int64_t B, int H, int W, float *loss, float *perplexity, void *workspace,
                          size_t workspace_bytes, vqvae_stream_t stream) {
    if (!w || !loss || !perplexity || !workspace) return VQVAE_ERR_NULL;
    const VqvaeDims *d = &w->dims;
    FwdWs f;
    int flags = 0;
    int rc = carve_forward(d, B, H, W, workspace, workspace_bytes, nullptr, 0, flags, f, false);
    if (rc != VQVAE_OK) return rc;
    {
        // the step's record: opened by begin with these shapes, [0, B) claimed exactly once -- else loss / perplexity would be wrong
        std::lock_guard<std::mutex> lk(g_parts_mu);
        auto it = g_parts.find(workspace);
        if (it == g_parts.end() || it->second.B != B || it->second.H != H || it->second.W != W) return VQVAE_ERR_SHAPE;
        int64_t covered = 0;
        for (const auto &c : it->second.claimed) covered += c.second - c.first;            // (parts never overlap: checked when claimed)
        if (covered != B) return VQVAE_ERR_SHAPE;
        g_parts.erase(it);
    }
    return vq_finalize_impl(reinterpret_cast<const double *>(f.z_e), (int)((B + 3) / 4), f.hist, d->n_embeddings, (int64_t)f.rows,
                            d->embedding_dim, d->beta, loss, perplexity, static_cast<hipStream_t>(stream));
}

}  // extern "C"
