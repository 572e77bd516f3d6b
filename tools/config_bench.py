#!/usr/bin/env python3
"""images/s of VQVAE.forward at the other BASELINE configurations (single GPU; not bench.py lines -- those
configs are parity cases): config 4 (224x224, K=1024, D=64, B=512) and config 5 per GPU (256x256, K=8192, D=128,
B=1024).  Maps larger than 8x8 / 16x16 take the generic (not tile-resident) conv kernels; K=8192, D=128 takes the
exhaustive fp32 quantizer kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
for name, (K, D, B, H) in {"config 4": (1024, 64, 512, 224), "config 5 (per GPU)": (8192, 128, 1024, 256),
                           "config 3": (512, 64, 4096, 32)}.items():
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    torch.manual_seed(0)
    m = VQVAE(128, 32, 2, K, D, 0.25).to(dev).eval()
    x = torch.randn(B, 3, H, H, device=dev)
    with torch.no_grad():
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            out = m(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: B={B} {H}x{H} K={K} D={D}: {dt * 1e3:9.2f} ms/step  {B / dt:10.0f} img/s  "
          f"({torch.cuda.max_memory_allocated() / 2**30:.1f} GiB peak)", flush=True)
    del m, x, out
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
