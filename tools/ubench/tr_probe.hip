// ds_read_b64_tr_b16 addressing probe (gfx950): an LDS image [row][32] of fp16 (64 B per row, value = row * 32 + col); every
// lane reads the MFMA 32x32x16 operand the weight-gradient kernel wants -- lane (c = lane & 31, h = lane >> 5), read rd = 0 / 1:
// elements j = 0..3 = image[8 h + 4 rd + j][c] -- with the lane address  row(8 h + 4 rd + ((lane & 15) >> 2)) * 64 +
// (16 ((lane >> 4) & 1) + 4 (lane & 3)) * 2.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short *out) {
    __shared__ __attribute__((aligned(16))) _Float16 img[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) img[i] = (_Float16)(float)i;     // exact up to 2048
    __syncthreads();
    const int lane = threadIdx.x, h = lane >> 5;
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char *)(char *)img;
    for (int rd = 0; rd < 2; ++rd) {
        const unsigned addr = base + (unsigned)((8 * h + 4 * rd + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
        u32x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        out[(lane * 2 + rd) * 4 + 0] = (unsigned short)(v.x & 0xffff);
        out[(lane * 2 + rd) * 4 + 1] = (unsigned short)(v.x >> 16);
        out[(lane * 2 + rd) * 4 + 2] = (unsigned short)(v.y & 0xffff);
        out[(lane * 2 + rd) * 4 + 3] = (unsigned short)(v.y >> 16);
    }
}
static float h2f(unsigned short x) {
    _Float16 v;
    __builtin_memcpy(&v, &x, 2);
    return (float)v;
}
int main() {
    unsigned short *d, hbuf[64 * 8];
    hipMalloc(&d, sizeof(hbuf));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int rd = 0; rd < 2; ++rd)
            for (int j = 0; j < 4; ++j) {
                const int c = lane & 31, hh = lane >> 5;
                const float want = (float)((8 * hh + 4 * rd + j) * 32 + c), got = h2f(hbuf[(lane * 2 + rd) * 4 + j]);
                if (want != got) { if (bad < 12) printf("lane %d rd %d j %d: got %g want %g\n", lane, rd, j, got, want); ++bad; }
            }
    printf("tr_probe: %d mismatches of 512\n", bad);
    return bad != 0;
}
