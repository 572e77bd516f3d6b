// Round 5: what shares a SIMD's issue slots on gfx950?  W waves per SIMD each run a loop whose body is A independent v_max3_f32
// + S s_add_u32 (+ L ds_read_b32 of a fixed address), interleaved; cycles per loop iteration per SIMD from s_memtime.
// If scalar / LDS instructions of OTHER waves issue beside a wave's vector instructions, t(A, S) = max(t(A, 0), t(0, S)); if every
// instruction of every wave takes its own slot, t(A, S) = t(A, 0) + t(0, S).
//   hipcc --offload-arch=gfx950 -O2 issue_mix.hip -o issue_mix && ./issue_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int A, int S, int L>
__global__ __launch_bounds__(1024) void mix(float *out, unsigned long long *cyc, int iters) {
    __shared__ float lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    float lacc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < A / 4) {        // four independent vector instructions
                asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %3, %3, %1, %2\n\tv_max3_f32 %4, %4, %1, %2\n\tv_max3_f32 %5, %5, %1, %2"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) ::);
            }
            if (k < S / 4) {
                asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
            }
            if (k < L) {
                float x;
                asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((threadIdx.x & 63) * 4) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                lacc += x;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = lacc + (float)(s0 + s1 + s2 + s3);
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int A, int S, int L>
void run(const char *name, int waves_per_simd) {
    const int threads = waves_per_simd * 4 * 64, blocks = 256, iters = 2000;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, (size_t)blocks * (threads / 64) * 8);
    mix<A, S, L><<<blocks, threads>>>(out, cyc, iters);
    mix<A, S, L><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[h.size() / 2] / iters;           // s_memtime ticks (100 MHz constant clock) per iteration per wave
    printf("%-28s waves/SIMD %d  A=%2d S=%2d L=%2d : %.3f ticks/iter (median wave)  = %.1f ns ; per instruction of the SIMD %.2f ns\n", name, waves_per_simd, A, S, L,
           ticks, ticks * 10.0, ticks * 10.0 / ((A + S + L * 2) * waves_per_simd));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<64, 0, 0>("valu only", 1); run<0, 64, 0>("salu only", 1); run<64, 64, 0>("valu + salu", 1); run<64, 32, 0>("valu + salu/2", 1); run<64, 0, 8>("valu + 8 lds", 1); run<32, 32, 0>("valu32 + salu32", 1); }
        if (w == 2) { run<64, 0, 0>("valu only", 2); run<0, 64, 0>("salu only", 2); run<64, 64, 0>("valu + salu", 2); run<64, 32, 0>("valu + salu/2", 2); run<64, 0, 8>("valu + 8 lds", 2); run<32, 32, 0>("valu32 + salu32", 2); }
        if (w == 4) { run<64, 0, 0>("valu only", 4); run<0, 64, 0>("salu only", 4); run<64, 64, 0>("valu + salu", 4); run<64, 32, 0>("valu + salu/2", 4); run<64, 0, 8>("valu + 8 lds", 4); run<32, 32, 0>("valu32 + salu32", 4); }
    }
    return 0;
}
