// Torch-free A/B of the stand-alone quantizer over several builds of libvqvae_hip.so (tools/build_variant.py), on the z_e rows of
// the reference-initialised model (tools/vq_cdata.py): every build is checked bit for bit (indices against the reference's own,
// z_q against fl(z + fl(e - z)) computed here) and timed by the dispatch's events (vqvae_profile_*), builds interleaved.
//   vq_ab DATA.bin ITERS lib1.so [lib2.so ...]         -> one JSON line per (rows, launch form)
// A build with -DVQ_TRACE also exports vqvae_debug_vq_trace: its per-wave stamps are summarised (and dumped to TRACE_OUT if set).
// hipcc -O2 vq_ab.cpp -o vq_ab -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef size_t (*ws_fn)(int64_t, int, int);
typedef int (*fwd_fn)(const float *, const float *, int64_t, int, int, int, int, float, int, float *, int64_t *, int32_t *, float *, float *, void *, size_t, void *);
typedef int (*pen_fn)(int);
typedef int (*pcol_fn)(int, double *, int *);
typedef int (*trace_fn)(void *, size_t);

struct Lib { std::string name; void *h; ws_fn ws; fwd_fn fwd; pen_fn pen; pcol_fn pcol; trace_fn trace; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: vq_ab DATA ITERS lib...\n"); return 1; }
    const int iters = atoi(argv[2]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("data"); return 1; }
    int64_t hdr[3];
    if (fread(hdr, 8, 3, f) != 3) return 1;
    const int64_t N0 = hdr[0]; const int K = (int)hdr[1], D = (int)hdr[2];
    std::vector<float> z0((size_t)N0 * D), cb((size_t)K * D);
    std::vector<int32_t> idx0(N0);
    if (fread(z0.data(), 4, z0.size(), f) != z0.size() || fread(cb.data(), 4, cb.size(), f) != cb.size() || fread(idx0.data(), 4, N0, f) != (size_t)N0) return 1;
    fclose(f);
    // z_q the reference computes: fl(z + fl(e_idx - z))  (models/quantizer.py:67)
    std::vector<float> zq0((size_t)N0 * D);
    for (int64_t r = 0; r < N0; ++r)
        for (int c = 0; c < D; ++c) {
            const float zz = z0[r * D + c]; volatile float d = cb[(size_t)idx0[r] * D + c] - zz; zq0[r * D + c] = zz + d;
        }
    // VQ_AB_NCHW=1: the module's own (B, D, 8, 8) layout in and out (flags without VQVAE_VQ_ROWMAJOR): images of 64 consecutive rows
    const bool nchw = getenv("VQ_AB_NCHW") && atoi(getenv("VQ_AB_NCHW"));
    if (nchw) {
        auto to_nchw = [&](std::vector<float> &v) {
            std::vector<float> t(v.size());
            for (int64_t b = 0; b < N0 / 64; ++b)
                for (int pos = 0; pos < 64; ++pos)
                    for (int c = 0; c < D; ++c) t[(size_t)(b * D + c) * 64 + pos] = v[(size_t)(b * 64 + pos) * D + c];
            v.swap(t);
        };
        to_nchw(z0); to_nchw(zq0);
    }
    std::vector<Lib> libs;
    for (int i = 3; i < argc; ++i) {
        Lib L; L.name = argv[i];
        const size_t sl = L.name.rfind('/'); if (sl != std::string::npos) L.name = L.name.substr(sl + 1);
        L.h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
        L.ws = (ws_fn)dlsym(L.h, "vqvae_vq_workspace_bytes"); L.fwd = (fwd_fn)dlsym(L.h, "vqvae_vq_forward_f32");
        L.pen = (pen_fn)dlsym(L.h, "vqvae_profile_enable"); L.pcol = (pcol_fn)dlsym(L.h, "vqvae_profile_collect");
        L.trace = (trace_fn)dlsym(L.h, "vqvae_debug_vq_trace");
        if (!L.ws || !L.fwd || !L.pen || !L.pcol) { fprintf(stderr, "%s: missing symbols\n", argv[i]); return 1; }
        libs.push_back(L);
    }
    const char *mults_env = getenv("VQ_AB_MULTS");
    std::vector<int> mults;
    { std::string m = mults_env ? mults_env : "1,4,32"; size_t p = 0; while (p < m.size()) { mults.push_back(atoi(m.c_str() + p)); p = m.find(',', p); if (p == std::string::npos) break; ++p; } }
    const char *forms_env = getenv("VQ_AB_FORMS");
    std::vector<int> forms;
    { std::string m = forms_env ? forms_env : "8,16"; size_t p = 0; while (p < m.size()) { forms.push_back(atoi(m.c_str() + p)); p = m.find(',', p); if (p == std::string::npos) break; ++p; } }

    for (int mult : mults) {
        const int64_t N = N0 * mult, B = N / 64;
        float *dz, *dzq, *dcb, *dloss; int64_t *didx; int32_t *dhist;
        CK(hipMalloc(&dz, N * D * 4)); CK(hipMalloc(&dzq, N * D * 4)); CK(hipMalloc(&dcb, (size_t)K * D * 4)); CK(hipMalloc(&dloss, 64));
        CK(hipMalloc(&didx, N * 8)); CK(hipMalloc(&dhist, K * 4));
        for (int m = 0; m < mult; ++m) CK(hipMemcpy(dz + (size_t)m * N0 * D, z0.data(), (size_t)N0 * D * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dcb, cb.data(), (size_t)K * D * 4, hipMemcpyHostToDevice));
        std::vector<void *> wss(libs.size()); std::vector<size_t> wsb(libs.size());
        for (size_t l = 0; l < libs.size(); ++l) { wsb[l] = libs[l].ws(N, K, D); CK(hipMalloc(&wss[l], wsb[l])); }
        std::vector<float> hzq((size_t)N0 * D); std::vector<int64_t> hidx(N0);
        for (int form : forms) {
            const int fl = (nchw ? 0x0 : 0x1) | (form == 8 ? 0x100 : form == 16 ? 0x200 : form == 12 ? 0x300 : form == 32 ? 0x400 : 0);
            std::vector<std::vector<float>> ts(libs.size());
            std::vector<int> ok(libs.size(), 1);
            // correctness + warm-up
            for (size_t l = 0; l < libs.size(); ++l) {
                CK(hipMemset(dzq, 0xff, N * D * 4)); CK(hipMemset(didx, 0xff, N * 8));
                int rc = libs[l].fwd(dz, dcb, B, D, 8, 8, K, 0.25f, fl, dzq, didx, dhist, dloss, dloss + 1, wss[l], wsb[l], nullptr);
                if (rc) { fprintf(stderr, "%s: rc %d\n", libs[l].name.c_str(), rc); ok[l] = 0; continue; }
                for (int w = 0; w < 3; ++w) libs[l].fwd(dz, dcb, B, D, 8, 8, K, 0.25f, fl | 0x2, dzq, didx, dhist, dloss, dloss + 1, wss[l], wsb[l], nullptr);
                CK(hipDeviceSynchronize());
                for (int m = 0; m < mult; m += (mult > 4 ? mult - 1 : 1)) {           // first and last copy (every copy when few)
                    CK(hipMemcpy(hzq.data(), dzq + (size_t)m * N0 * D, (size_t)N0 * D * 4, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hidx.data(), didx + (size_t)m * N0, (size_t)N0 * 8, hipMemcpyDeviceToHost));
                    long bad_i = 0, bad_q = 0;
                    for (int64_t r = 0; r < N0; ++r) bad_i += hidx[r] != idx0[r];
                    bad_q = memcmp(hzq.data(), zq0.data(), (size_t)N0 * D * 4) != 0;
                    if (bad_i || bad_q) { ok[l] = 0; fprintf(stderr, "%s rows=%lld form=%d copy %d: %ld wrong indices, z_q %s\n", libs[l].name.c_str(), (long long)N, form, m, bad_i, bad_q ? "DIFFERS" : "ok"); }
                }
            }
            for (int it = 0; it < iters; ++it)
                for (size_t l = 0; l < libs.size(); ++l) {
                    libs[l].pen(1);
                    libs[l].fwd(dz, dcb, B, D, 8, 8, K, 0.25f, fl | 0x2, dzq, didx, dhist, dloss, dloss + 1, wss[l], wsb[l], nullptr);
                    double ms = 0; int n = 0; libs[l].pcol(0, &ms, &n);
                    libs[l].pen(0);
                    ts[l].push_back((float)(ms / (n > 0 ? n : 1) * 1e3));
                }
            printf("{\"rows\": %lld, \"form\": %d", (long long)N, form);
            for (size_t l = 0; l < libs.size(); ++l) {
                std::sort(ts[l].begin(), ts[l].end());
                printf(", \"%s\": {\"best_us\": %.2f, \"median_us\": %.2f, \"frac_best\": %.4f, \"bit_exact\": %s}", libs[l].name.c_str(), ts[l][0], ts[l][ts[l].size() / 2],
                       N * 520.0 / ts[l][0] / 8e6, ok[l] ? "true" : "false");
            }
            printf("}\n"); fflush(stdout);
            // per-wave stamps of a trace build (of the LAST launch)
            for (size_t l = 0; l < libs.size(); ++l) if (libs[l].trace) {
                std::vector<unsigned long long> tr(4096 * 8);
                libs[l].fwd(dz, dcb, B, D, 8, 8, K, 0.25f, fl | 0x2, dzq, didx, dhist, dloss, dloss + 1, wss[l], wsb[l], nullptr);
                CK(hipDeviceSynchronize());
                if (libs[l].trace(tr.data(), tr.size() * 8)) continue;
                int NW = form == 32 ? 8 : form;
                if (NW == 0) { NW = 8; for (int w = 0; w < 256; ++w) if (tr[(w * 16 + 8) * 8 + 7]) NW = 16; }   // (default rule: which form ran?)
                if (NW == 16) { bool odd = false; for (int w = 0; w < 256; ++w) if (tr[(w * 16 + 9) * 8 + 7]) odd = true; if (!odd) NW = 8; }
                const int nwaves = 256 * NW;
                unsigned long long t0 = ~0ull, tend = 0;
                for (int w = 0; w < nwaves; ++w) { if (tr[w * 8] && tr[w * 8] < t0) t0 = tr[w * 8]; if (tr[w * 8 + 7] > tend) tend = tr[w * 8 + 7]; }
                const char *nm[8] = {"start", "barrier", "unit1", "unit2", "unit3", "unit4", "loop_exit", "end"};
                printf("  trace %s rows=%lld form=%d: first start -> last end %.2f us; per stamp (us after the first wave's start) min / median / max over waves that have it [count]\n",
                       libs[l].name.c_str(), (long long)N, form, (tend - t0) * 0.01);
                for (int s = 0; s < 8; ++s) {
                    std::vector<double> v;
                    for (int w = 0; w < nwaves; ++w) if (tr[w * 8 + s]) v.push_back((tr[w * 8 + s] - t0) * 0.01);
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    printf("    %-9s %6.2f / %6.2f / %6.2f  [%zu]\n", nm[s], v[0], v[v.size() / 2], v.back(), v.size());
                }
                // by position on the SIMD (wave >> 2): median of each stamp
                for (int q = 0; q < NW / 4; ++q) {
                    printf("    waves %d..%d of a workgroup (SIMD slot %d): ", 4 * q, 4 * q + 3, q);
                    for (int s = 0; s < 8; ++s) {
                        std::vector<double> v;
                        for (int w = 0; w < nwaves; ++w) if (((w % NW) >> 2) == q && tr[w * 8 + s]) v.push_back((tr[w * 8 + s] - t0) * 0.01);
                        if (v.empty()) { printf("%s -  ", nm[s]); continue; }
                        std::sort(v.begin(), v.end());
                        printf("%s %.1f[%zu]  ", nm[s], v[v.size() / 2], v.size());
                    }
                    printf("\n");
                }
                const char *out = getenv("TRACE_OUT");
                if (out) {
                    char path[512]; snprintf(path, sizeof path, "%s_%s_%lld_%d.bin", out, libs[l].name.c_str(), (long long)N, form);
                    FILE *o = fopen(path, "wb"); if (o) { fwrite(tr.data(), 8, (size_t)nwaves * 8, o); fclose(o); }
                }
            }
        }
        for (void *w : wss) CK(hipFree(w));
        CK(hipFree(dz)); CK(hipFree(dzq)); CK(hipFree(dcb)); CK(hipFree(dloss)); CK(hipFree(didx)); CK(hipFree(dhist));
    }
    return 0;
}
