"""Timing of the streamed-codebook quantizer (K = 8192, D = 128, one slab of 262 144 rows = config 5's shape) for A/B and knock-out builds
of vq_chunk.hip (VQVAE_HIP_LIB_OVERRIDE; knock-outs give WRONG results, only their time means something)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd import _lib, functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
K, D, N = 8192, 128, 262144
cb = torch.empty(K, D, device=dev).uniform_(-1 / K, 1 / K)
z = torch.randn(N // 64, 8, 8, D, device=dev) * 0.07
ws = F.vq_workspace(K, D, dev)
F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws)
torch.cuda.synchronize()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); F.vq_forward(z, cb, 0.25, rowmajor=True, workspace=ws, prepared=True); b.record(); b.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print(f"{os.path.basename(_lib.LIB_PATH):28s} whole call (rows16 + sweep + resolve + gather + finalize): median {ts[4]:8.1f} us  min {ts[0]:8.1f} us", flush=True)
