// Torch-free A/B of the whole-path conv kernels on LARGER maps (BASELINE config 4's 224x224 images: the halo-tile kernels) over
// several builds of libvqvae_hip.so: random weights of the reference's shapes (uniform +-1/sqrt(fan_in)), random images; every
// build's z_e and x_hat are compared BITWISE with the first build's (a scheduling change must not move a bit), kernel groups are
// timed through the library's profile hooks (ids 1 conv / 2 residual / 3 first layer / 4 last layer), builds interleaved.
//   enc_ab B H ITERS lib1.so [lib2.so ...]
// hipcc -O2 enc_ab.cpp -o enc_ab -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/vqvae_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)

struct Lib {
    std::string name; void *h;
    size_t (*packed_bytes)(const VqvaeDims *);
    int (*pack)(const VqvaeDims *, const VqvaeRawWeights *, void *, size_t, VqvaeWeights *, void *);
    size_t (*ws_bytes)(const VqvaeDims *, int64_t, int, int);
    int (*enc)(const VqvaeWeights *, const float *, int64_t, int, int, float *, void *, size_t, void *);
    int (*dec)(const VqvaeWeights *, const float *, int64_t, int, int, float *, void *, size_t, void *);
    int (*pen)(int); int (*pcol)(int, double *, int *);
    void *packed; VqvaeWeights W;
};

static unsigned long long rng = 88172645463325252ull;
static float urand() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }
static float *dev_rand(size_t n, float scale) {
    std::vector<float> h(n); for (auto &v : h) v = urand() * scale;
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: enc_ab B H ITERS lib...\n"); return 1; }
    const int64_t B = atoll(argv[1]); const int H = atoi(argv[2]), W = H, iters = atoi(argv[3]);
    VqvaeDims d; d.h_dim = 128; d.res_h_dim = 32; d.n_res_layers = 2; d.n_embeddings = 1024; d.embedding_dim = 64; d.in_ch = 3; d.beta = 0.25f;
    const int h = d.h_dim, Rh = d.res_h_dim, D = d.embedding_dim, K = d.n_embeddings;
    VqvaeRawWeights raw;
    raw.enc0_w = dev_rand((size_t)(h / 2) * 3 * 16, 1.0f / sqrtf(3 * 16)); raw.enc0_b = dev_rand(h / 2, 0.1f);
    raw.enc2_w = dev_rand((size_t)h * (h / 2) * 16, 1.0f / sqrtf(h / 2 * 16)); raw.enc2_b = dev_rand(h, 0.1f);
    raw.enc4_w = dev_rand((size_t)h * h * 9, 1.0f / sqrtf(h * 9)); raw.enc4_b = dev_rand(h, 0.1f);
    raw.enc_res_w1 = dev_rand((size_t)Rh * h * 9, 1.0f / sqrtf(h * 9)); raw.enc_res_w2 = dev_rand((size_t)h * Rh, 1.0f / sqrtf(Rh));
    raw.pre_w = dev_rand((size_t)D * h, 1.0f / sqrtf(h)); raw.pre_b = dev_rand(D, 0.1f);
    raw.codebook = dev_rand((size_t)K * D, 1.0f / K);
    raw.dec0_w = dev_rand((size_t)D * h * 9, 1.0f / sqrtf(D * 9)); raw.dec0_b = dev_rand(h, 0.1f);
    raw.dec_res_w1 = dev_rand((size_t)Rh * h * 9, 1.0f / sqrtf(h * 9)); raw.dec_res_w2 = dev_rand((size_t)h * Rh, 1.0f / sqrtf(Rh));
    raw.dec2_w = dev_rand((size_t)h * (h / 2) * 16, 1.0f / sqrtf(h * 4)); raw.dec2_b = dev_rand(h / 2, 0.1f);
    raw.dec4_w = dev_rand((size_t)(h / 2) * 3 * 16, 1.0f / sqrtf(h / 2 * 4)); raw.dec4_b = dev_rand(3, 0.1f);
    float *x = dev_rand((size_t)B * 3 * H * W, 1.0f);
    const size_t nz = (size_t)B * (H / 4) * (W / 4) * D, nx = (size_t)B * 3 * H * W;
    float *ze, *xh; CK(hipMalloc(&ze, nz * 4)); CK(hipMalloc(&xh, nx * 4));
    std::vector<Lib> libs;
    for (int i = 4; i < argc; ++i) {
        Lib L; L.name = argv[i]; const size_t sl = L.name.rfind('/'); if (sl != std::string::npos) L.name = L.name.substr(sl + 1);
        L.h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
#define SYM(f, n) *(void **)(&L.f) = dlsym(L.h, n); if (!L.f) { fprintf(stderr, "%s: no %s\n", argv[i], n); return 1; }
        SYM(packed_bytes, "vqvae_weights_packed_bytes") SYM(pack, "vqvae_weights_pack_f32") SYM(ws_bytes, "vqvae_workspace_bytes")
        SYM(enc, "vqvae_encoder_f32") SYM(dec, "vqvae_decoder_f32") SYM(pen, "vqvae_profile_enable") SYM(pcol, "vqvae_profile_collect")
        const size_t pb = L.packed_bytes(&d);
        CK(hipMalloc(&L.packed, pb));
        int rc = L.pack(&d, &raw, L.packed, pb, &L.W, nullptr);
        if (rc) { fprintf(stderr, "%s: pack rc %d\n", argv[i], rc); return 1; }
        libs.push_back(L);
    }
    const size_t wsb = libs[0].ws_bytes(&d, B, H, W);
    void *ws; CK(hipMalloc(&ws, wsb));
    std::vector<float> ref_ze(nz), ref_xh(nx), got_ze(nz), got_xh(nx);
    std::vector<int> ok(libs.size(), 1);
    for (size_t l = 0; l < libs.size(); ++l) {
        CK(hipMemset(ze, 0xff, nz * 4)); CK(hipMemset(xh, 0xff, nx * 4));
        int rc = libs[l].enc(&libs[l].W, x, B, H, W, ze, ws, wsb, nullptr);
        if (!rc) rc = libs[l].dec(&libs[l].W, ze, B, H / 4, W / 4, xh, ws, wsb, nullptr);
        CK(hipDeviceSynchronize());
        if (rc) { fprintf(stderr, "%s: rc %d\n", libs[l].name.c_str(), rc); ok[l] = 0; continue; }
        CK(hipMemcpy(l ? got_ze.data() : ref_ze.data(), ze, nz * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(l ? got_xh.data() : ref_xh.data(), xh, nx * 4, hipMemcpyDeviceToHost));
        if (l) ok[l] = !memcmp(got_ze.data(), ref_ze.data(), nz * 4) && !memcmp(got_xh.data(), ref_xh.data(), nx * 4);
        else { double s = 0, m = 0; for (size_t i = 0; i < nz; ++i) { s += fabs(ref_ze[i]); m = std::max(m, (double)fabs(ref_ze[i])); } printf("z_e mean|.| %.4g max %.4g (finite: %d)\n", s / nz, m, (int)std::isfinite(m)); }
    }
    std::vector<std::vector<double>> t(libs.size(), std::vector<double>(6, 0.0));
    std::vector<std::vector<float>> wall(libs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < iters; ++it)
        for (size_t l = 0; l < libs.size(); ++l) {
            libs[l].pen(1);
            CK(hipEventRecord(e0, nullptr));
            libs[l].enc(&libs[l].W, x, B, H, W, ze, ws, wsb, nullptr);
            libs[l].dec(&libs[l].W, ze, B, H / 4, W / 4, xh, ws, wsb, nullptr);
            CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); wall[l].push_back(ms);
            for (int id = 1; id <= 4; ++id) { double m = 0; int n = 0; libs[l].pcol(id, &m, &n); t[l][id] += m; }
            libs[l].pen(0);
        }
    for (size_t l = 0; l < libs.size(); ++l) {
        std::sort(wall[l].begin(), wall[l].end());
        printf("%-28s B=%lld %dx%d: encoder+decoder (events inflate) best %.3f ms median %.3f ms | per call: conv %.3f  residual %.3f  first %.3f  last %.3f ms | bitwise == first build: %s\n",
               libs[l].name.c_str(), (long long)B, H, W, wall[l][0], wall[l][wall[l].size() / 2], t[l][1] / iters, t[l][2] / iters, t[l][3] / iters, t[l][4] / iters, ok[l] ? "yes" : "NO");
    }
    return 0;
}
