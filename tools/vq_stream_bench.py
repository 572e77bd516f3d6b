#!/usr/bin/env python3
"""Time the quantizer alone at BASELINE config 4 / 5 sizes (streamed-codebook kernels, vq_chunk.hip) through the C ABI.
VQVAE_HIP_LIB_OVERRIDE selects a variant library (tools/build_variant.py).  Prints ms per call (median of 5)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import functional as F, _lib

dev = torch.device("cuda:0")
for K, D, rows in ((8192, 128, 1 << 20), (1024, 64, 1605632)):
    g = torch.Generator().manual_seed(K)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
    z = (torch.randn(rows // 64, 8, 8, D, generator=g) * 0.066).to(dev)
    for _ in range(2):
        out = F.vq_forward(z, cb, 0.25, rowmajor=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = F.vq_forward(z, cb, 0.25, rowmajor=True); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{os.environ.get('VQVAE_HIP_LIB_OVERRIDE', 'default').split('/')[-1]:28s} {_lib.vq_kernel_name(K, D)} K={K} D={D} rows={rows}: "
          f"{ts[2]:.3f} ms  ({rows / ts[2] / 1e6:.2f} G rows/s) perplexity {out[2].item():.2f}", flush=True)
