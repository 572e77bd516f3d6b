// Library-level entry points of the C ABI (include/vqvae_hip.h).
#include "common.h"

namespace vqvae {
namespace {
constexpr int kMaxRec = 256;
struct ProfState {
    bool on = false;
    hipEvent_t start[VQVAE_PROF_NUM_IDS][kMaxRec];
    hipEvent_t stop[VQVAE_PROF_NUM_IDS][kMaxRec];
    bool created[VQVAE_PROF_NUM_IDS][kMaxRec] = {};
    int n[VQVAE_PROF_NUM_IDS] = {};
    bool open_[VQVAE_PROF_NUM_IDS] = {};
} g_prof;
}  // namespace

void prof_begin(int id, hipStream_t st) {
    if (!g_prof.on || id < 0 || id >= VQVAE_PROF_NUM_IDS || g_prof.n[id] >= kMaxRec) return;
    const int i = g_prof.n[id];
    if (!g_prof.created[id][i]) {
        if (hipEventCreate(&g_prof.start[id][i]) != hipSuccess) return;
        if (hipEventCreate(&g_prof.stop[id][i]) != hipSuccess) return;
        g_prof.created[id][i] = true;
    }
    (void)hipEventRecord(g_prof.start[id][i], st);
    g_prof.open_[id] = true;
}

void prof_end(int id, hipStream_t st) {
    if (!g_prof.on || id < 0 || id >= VQVAE_PROF_NUM_IDS || !g_prof.open_[id]) return;
    (void)hipEventRecord(g_prof.stop[id][g_prof.n[id]], st);
    g_prof.n[id] += 1;
    g_prof.open_[id] = false;
}
}  // namespace vqvae

extern "C" {

int vqvae_profile_enable(int on) {
    vqvae::g_prof.on = on != 0;
    return VQVAE_OK;
}

int vqvae_profile_collect(int kernel_id, double *total_ms, int *launches) {
    using vqvae::g_prof;
    if (!total_ms || !launches) return VQVAE_ERR_NULL;
    if (kernel_id < 0 || kernel_id >= VQVAE_PROF_NUM_IDS) return VQVAE_ERR_SHAPE;
    double tot = 0.0;
    for (int i = 0; i < g_prof.n[kernel_id]; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.stop[kernel_id][i]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, g_prof.start[kernel_id][i], g_prof.stop[kernel_id][i]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
    }
    *total_ms = tot;
    *launches = g_prof.n[kernel_id];
    g_prof.n[kernel_id] = 0;
    return VQVAE_OK;
}

int vqvae_abi_version(void) { return VQVAE_HIP_ABI_VERSION; }

const char *vqvae_strerror(int code) {
    switch (code) {
        case VQVAE_OK: return "ok";
        case VQVAE_ERR_NULL: return "vqvae: a required pointer argument is NULL";
        case VQVAE_ERR_SHAPE: return "vqvae: non-positive or inconsistent dimension";
        case VQVAE_ERR_UNSUPPORTED: return "vqvae: shape not supported by the gfx950 kernels";
        case VQVAE_ERR_WORKSPACE: return "vqvae: workspace missing or too small";
        case VQVAE_ERR_OVERFLOW: return "vqvae: element count overflows the index type";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "vqvae: unknown error code";
}

}  // extern "C"
