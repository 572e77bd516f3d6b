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

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kLdsBytes = 160 * 1024;  // per CU on gfx950

inline int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// profiling hooks (capi.hip); no-ops unless vqvae_profile_enable(1)
void prof_begin(int id, hipStream_t st);
void prof_end(int id, hipStream_t st);

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- quantizer workspace layout (host + device agree on it) -----------------
struct VqPlan {
    int KC;           // codes per LDS chunk (multiple of 32)
    int nchunks;      // ceil(K / KC)
    int K_pad;        // nchunks * KC
    size_t off_flags, off_ee, off_img, off_partials, total;
    size_t lds_bytes;
};
constexpr int kVqMaxGrid = 1024;     // persistent grid never exceeds this many workgroups

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
    p.off_ee = 256;
    p.off_img = align_up(p.off_ee + (size_t)p.K_pad * 4, 256);
    p.off_partials = align_up(p.off_img + (size_t)p.K_pad * D * 4, 256);
    p.total = align_up(p.off_partials + sizeof(double) * kVqMaxGrid, 256);
    return p;
}

}  // namespace vqvae
