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

// one launch = kCalibBlocks x 4 waves (two per SIMD on 256 CUs), `iters` x four independent MFMAs per wave; operands from a
// fixed pseudo-random sequence (the power draw, and with it the clock, depends on the operand bits: zeros run 40 % faster);
// clk[2 b] = shader cycles, clk[2 b + 1] = 100 MHz ticks of block b's first wave around its loop
constexpr int kCalibBlocks = 512;
typedef _Float16 calib_f16x8 __attribute__((ext_vector_type(8)));
typedef float calib_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void calib_mfma_kernel(float *__restrict__ out, unsigned long long *__restrict__ clk, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    calib_f16x8 a0, a1, b0, b1;
    unsigned s = 0x9E3779B9u * (unsigned)(lane + 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        auto next = [&]() { s = s * 1664525u + 1013904223u; return (_Float16)((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f); };
        a0[q] = next(); a1[q] = next(); b0[q] = next(); b1[q] = next();
    }
    calib_f32x16 y[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        y[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, y[0], 0, 0, 0);
        y[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, y[1], 0, 0, 0);
        y[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, y[2], 0, 0, 0);
        y[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, y[3], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float acc = 0.0f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc += y[j][r];
    out[(size_t)blockIdx.x * 256 + tid] = acc;
    if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

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

// the kernel's OWN begin / end (hipExtLaunchKernelGGL's start / stop events carry the dispatch's timestamps, what rocprofv3's
// kernel trace reports) instead of two marker packets around it, which add their own 2-3 us to a 40 us kernel
bool prof_dispatch(int id, hipEvent_t *start, hipEvent_t *stop) {
    *start = *stop = nullptr;
    if (!g_prof.on || id < 0 || id >= VQVAE_PROF_NUM_IDS || g_prof.n[id] >= kMaxRec) return false;
    const int i = g_prof.n[id];
    if (!g_prof.created[id][i]) {
        if (hipEventCreate(&g_prof.start[id][i]) != hipSuccess) return false;
        if (hipEventCreate(&g_prof.stop[id][i]) != hipSuccess) return false;
        g_prof.created[id][i] = true;
    }
    *start = g_prof.start[id][i];
    *stop = g_prof.stop[id][i];
    g_prof.n[id] += 1;
    return true;
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

}  // extern "C"

namespace vqvae {
// one workgroup per weight tensor (at most 256 x 256 x 16 elements, once per weight version): a[o] = max |w[o, ., .]|, then
// r[c] = max over (o, taps) of |w[o, c, .]| / a[o], then log2(max r / min r) over the channels that are not all zero
__global__ __launch_bounds__(1024) void weight_spread_kernel(const float *__restrict__ w, int Cout, int Cin, int taps, int transposed,
                                                             float *__restrict__ out) {
    __shared__ float a_s[1024], r_s[1024];
    __shared__ float red_hi[16], red_lo[16];
    const int tid = threadIdx.x;
    if (!w) { if (tid == 0) out[0] = 0.0f; return; }
    auto at = [&](int o, int c, int t) { return transposed ? w[((size_t)c * Cout + o) * taps + t] : w[((size_t)o * Cin + c) * taps + t]; };
    for (int o = tid; o < Cout; o += 1024) {
        float m = 0.0f;
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < taps; ++t) m = fmaxf(m, __builtin_fabsf(at(o, c, t)));
        a_s[o] = m;
    }
    __syncthreads();
    float hi = 0.0f, lo = 3.0e38f;
    for (int c = tid; c < Cin; c += 1024) {
        float r = 0.0f;
        for (int o = 0; o < Cout; ++o) {
            const float a = a_s[o];
            if (!(a > 0.0f) || !(a < 3.0e38f)) continue;
            for (int t = 0; t < taps; ++t) r = fmaxf(r, __builtin_fabsf(at(o, c, t)) / a);
        }
        r_s[c] = r;
        if (r > 0.0f) { hi = fmaxf(hi, r); lo = fminf(lo, r); }
    }
    for (int o = 32; o > 0; o >>= 1) { hi = fmaxf(hi, __shfl_xor(hi, o)); lo = fminf(lo, __shfl_xor(lo, o)); }
    if ((tid & 63) == 0) { red_hi[tid >> 6] = hi; red_lo[tid >> 6] = lo; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) { hi = fmaxf(hi, red_hi[i]); lo = fminf(lo, red_lo[i]); }
        hi = fmaxf(hi, red_hi[0]); lo = fminf(lo, red_lo[0]);
        out[0] = (hi > 0.0f && lo < 3.0e38f) ? __builtin_log2f(hi / lo) : 0.0f;
    }
}

void weight_spread_impl(const float *w, int Cout, int Cin, int taps, bool transposed, float *out, hipStream_t st) {
    hipLaunchKernelGGL(weight_spread_kernel, dim3(1), dim3(1024), 0, st, w, Cout, Cin, taps, transposed ? 1 : 0, out);
}
}  // namespace vqvae

extern "C" {

// ---- box calibration (round 4): a bare stream of v_mfma_f32_32x32x16_f16 on RANDOM operands, two waves
// per SIMD, what tools/ubench/mfma_power.hip measures: the chip's sustained matrix rate on this data is set by the clock it
// holds at its power limit, and that differs from box to box (profiles/r03_notes.txt section 10).  bench.py runs it for a
// fraction of a second before the timed region and prints TFLOP/s and the in-kernel shader clock next to the step time.
int vqvae_calibration_mfma_f16(int iters, void *scratch, size_t scratch_bytes, vqvae_stream_t stream) {
    if (!scratch) return VQVAE_ERR_NULL;
    if (iters < 1) return VQVAE_ERR_SHAPE;
    if (scratch_bytes < vqvae_calibration_scratch_bytes()) return VQVAE_ERR_WORKSPACE;
    char *p = static_cast<char *>(scratch);
    hipLaunchKernelGGL(vqvae::calib_mfma_kernel, dim3(vqvae::kCalibBlocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float *>(p + 16 * vqvae::kCalibBlocks), reinterpret_cast<unsigned long long *>(p), iters);
    return (int)hipGetLastError();
}

size_t vqvae_calibration_scratch_bytes(void) { return (size_t)vqvae::kCalibBlocks * (16 + 256 * sizeof(float)); }

double vqvae_calibration_flops(int iters) { return (double)vqvae::kCalibBlocks * 4 * (double)iters * 4 * 2.0 * 32 * 32 * 16; }

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
