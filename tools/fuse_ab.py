#!/usr/bin/env python3
"""Whole forward at BASELINE config 3 with the quantizer inside the encoder's last kernel (default) against the separate
quantizer launch (VQVAE_VQ_UNFUSED), interleaved in one process: ms per step, median of `reps` repeats of `steps` steps."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import functional as F
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).eval().to(dev)
x = torch.randn(4096, 3, 32, 32, device=dev)
steps, reps = 30, 15
res = {"fused": [], "unfused": []}
with torch.no_grad():
    for _ in range(5):
        m._forward_c(x); m._forward_c(x, vq_flags=F.VQ_UNFUSED)
    for r in range(reps):
        for name, fl in (("fused", 0), ("unfused", F.VQ_UNFUSED)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                m._forward_c(x, vq_flags=fl)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
for k, v in res.items():
    print(f"{k:8s} median {statistics.median(v):.4f} ms/step  min {min(v):.4f}  ({4096 / statistics.median(v) / 1e3:.3f} M images/s)")
