#!/usr/bin/env python3
"""Run the fused VQ kernel alone on a working set far beyond the 256 MiB Infinity Cache (so the
fabric-side counters see real HBM traffic) -- meant to be wrapped by rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import functional as F
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
K, D, N = 512, 64, 1 << 22                      # 4.19 M rows: z_e 1 GiB, z_q 1 GiB
cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
z = (torch.randn(N // 64, 8, 8, D, generator=g) * 0.066).to(dev)
ws = F.vq_workspace(K, D, dev)
for _ in range(3):
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
torch.cuda.synchronize()
print("rows", N, "algorithmic bytes per launch", N * (8 * D + 8))
