// Access-pattern probe for 256-B rows (N x 64 fp32): copy in -> out with
//   A  row per lane pair: lane (n = l&31, h = l>>5) moves bytes [128h, 128h+128) of row n in 8 x 16 B   (vq_filter today)
//   B  fully coalesced: each wave instruction moves 1 KiB contiguous (4 rows)
//   C  lane (n, h) moves 32-B pieces: bytes [64q + 32h, +32) of row n, q = 0..3
//   D  16 rows x 4 lanes: lane (i = l&15, g = l>>4) moves bytes [32g + 128s, +32), s = 0,1 (16x16x32 MFMA fragment shape)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const float *__restrict__ in, float *__restrict__ out, long long nblk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        const size_t r0 = (size_t)b * 256 + wave * 32;      // 32 rows per wave
        f32x4 v[8];
        size_t off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) off[q] = (r0 + l31) * 64 + 32 * h + 4 * q;
            if (MODE == 1) off[q] = r0 * 64 + (size_t)q * 256 + lane * 4;
            if (MODE == 2) off[q] = (r0 + l31) * 64 + 16 * (q >> 1) + 8 * h + 4 * (q & 1);
            if (MODE == 3) off[q] = (r0 + (lane & 15) + 16 * (q >> 2)) * 64 + 8 * (lane >> 4) + 32 * ((q >> 1) & 1) + 4 * (q & 1);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4 *>(in + off[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4 *>(out + off[q]) = v[q] + 1.0f;
    }
}

template <int MODE>
void run(const float *in, float *out, long long N, const char *name) {
    const long long nblk = N / 256;
    const int grid = nblk < 512 ? (int)nblk : 512;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, in, out, nblk);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, in, out, nblk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("N=%lld %-28s %.1f us  %.0f GB/s\n", N, name, ms * 1e3, N * 512.0 / ms / 1e6);
}

int main() {
    for (long long N : {262144LL, 2097152LL}) {
        float *in, *out;
        (void)hipMalloc(&in, N * 256); (void)hipMalloc(&out, N * 256);
        (void)hipMemset(in, 0, N * 256);
        run<0>(in, out, N, "A row-per-lane-pair 8x16B");
        run<1>(in, out, N, "B coalesced 1KiB/instr");
        run<2>(in, out, N, "C 32B pieces");
        run<3>(in, out, N, "D 16x4 fragment");
        (void)hipFree(in); (void)hipFree(out);
    }
    return 0;
}
