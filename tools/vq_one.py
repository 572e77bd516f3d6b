#!/usr/bin/env python3
"""Run the fused VQ kernel a few times at BASELINE config-3 size (for rocprofv3 --pmc wrappers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import functional as F
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
K, D, N = 512, 64, 262144
cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
z = (torch.randn(N // 64, 8, 8, D, generator=g) * 0.066).to(dev)
ws = F.vq_workspace(K, D, dev)
for _ in range(5):
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
torch.cuda.synchronize()
