#!/usr/bin/env python3
"""Small-batch latency of VQVAE.forward: eager launches vs one hipGraph replay (vqvae_amd/graph.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd.graph import GraphedForward
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).eval()
for B in (1, 32, 256, 1024):
    x = torch.randn(B, 3, 32, 32, device=dev)
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
    g = GraphedForward(m, x)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / n
    print(f"B={B:5d}  eager {eager * 1e6:8.1f} us/forward ({B / eager:10.0f} img/s)   "
          f"graph {graph * 1e6:8.1f} us/forward ({B / graph:10.0f} img/s)   x{eager / graph:.2f}")
