#!/usr/bin/env python3
"""Adds s_memtime phase stamps to conv_halo8_h2_kernel IN PLACE (debug builds only, never committed):

    cp vqvae_amd/csrc/conv.hip /tmp/conv_clean.hip
    python tools/conv_stamp_patch.py && python tools/build_variant.py stamp
    cp /tmp/conv_clean.hip vqvae_amd/csrc/conv.hip
    VQVAE_BENCH_LIB=vqvae_amd/build/variants/libvqvae_stamp.so python tools/halo_phase.py 256 256

Every wave sums the cycles between the stamps into eight slots and adds them to a global table at its end
(vqvae_debug_conv_stamps reads / clears it); profiles/r03_notes.txt section 8 holds the figures."""

p='vqvae_amd/csrc/conv.hip'; s=open(p).read()
anchor='template <int NT, int TPS, bool S2D = false, int NPH = 1, int HALO = 1>\n__global__ __launch_bounds__(256, 2) void conv_halo8_h2_kernel('
assert s.count(anchor)==1
s=s.replace(anchor,'''__device__ unsigned long long g_conv_stamp[64];
#define CST(slot) do { const unsigned long long n_ = __builtin_readcyclecounter(); tacc[slot] += (unsigned)(n_ - tprev); tprev = n_; } while (0)
'''+anchor)
i=s.index(anchor)
def rep(old,new):
    global s
    j=s.index(old,i); s=s[:j]+new+s[j+len(old):]
rep('''    u32x4 *As = As_all + wave * TILE;
''','''    u32x4 *As = As_all + wave * TILE;
    unsigned tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
''')
rep('''    load_raw(0);
    int sl = 0, grp = 0;
''','''    load_raw(0);
    int sl = 0, grp = 0;
    CST(0);
''')
rep('''        if (grp == 0) stage();                         // (wave-private tile, LDS operations of a wave execute in order)
''','''        if (grp == 0) { stage(); CST(1); }
''')
rep('''        __syncthreads();
        if (s + 1 < nstage) dma_stage(grp + 1 == ngrp ? sl + 1 : sl, grp + 1 == ngrp ? 0 : grp + 1, (s + 1) & 1);
        if (grp == 0 && sl + 1 < nslice) load_raw(sl + 1);
''','''        CST(2);
        __syncthreads();
        CST(3);
        if (s + 1 < nstage) dma_stage(grp + 1 == ngrp ? sl + 1 : sl, grp + 1 == ngrp ? 0 : grp + 1, (s + 1) & 1);
        CST(4);
        if (grp == 0 && sl + 1 < nslice) load_raw(sl + 1);
        CST(5);
''')
rep('''        if (++grp == ngrp) { grp = 0; ++sl; }
    }
''','''        if (++grp == ngrp) { grp = 0; ++sl; }
        CST(6);
    }
''')
rep('''    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
}
''','''    if (out_amax && img_ok) publish_amax(out_amax, img, omax, lane);
    CST(7);
    if (lane == 0) {
        const int var = S2D ? 0 : (HALO == 0 ? 2 : (NT == 2 ? 3 : 1));
        for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_conv_stamp[16 * var + i_], (unsigned long long)tacc[i_]);
        atomicAdd(&g_conv_stamp[16 * var + 8], 1ull);
    }
}
extern "C" __attribute__((visibility("default"))) int vqvae_debug_conv_stamps(unsigned long long *dst_host, int reset) {
    unsigned long long z[64] = {0};
    int rc = (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_conv_stamp), sizeof(z));
    if (reset) rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamp), z, sizeof(z));
    return rc;
}
''')
open(p,'w').write(s)
