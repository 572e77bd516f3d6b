"""rocprofv3 target: the stand-alone quantizer at K = 512, D = 128, 262 144 rows (the streamed-codebook path), 10 calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
K, D, N = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 128, 262144
cb = torch.empty(K, D, device=dev).uniform_(-1 / K, 1 / K)
z = torch.randn(N // 64, 8, 8, D, device=dev) * 0.07
ws = F.vq_workspace(K, D, dev)
F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
for _ in range(10):
    F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True)
torch.cuda.synchronize()
