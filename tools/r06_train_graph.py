#!/usr/bin/env python3
"""Is the training step (main.py:74-78 on the HIP kernels, ~90 launches) launch-bound?  The whole step -- forward, fused losses, backward --
captured in ONE hipGraph (torch.cuda.graph: every launch of the library goes to the capturing stream, nothing in the step synchronises)
and replayed, against the eager loop.  Measured: 5.85 ms eager, 5.91 ms replayed at B = 4096 -- it is not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import conv, training as T
from vqvae_amd.modules import VQVAE
dev = torch.device("cuda:0")
B = 4096
torch.manual_seed(0)
conv.set_conv_backend("hip")
model = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).train()
x = torch.randn(B, 3, 32, 32, device=dev)
def step():
    model.zero_grad(set_to_none=False)
    el, xh, pp = model(x)
    stats = T.step_losses(el, xh, pp, x, 0.06)
    stats[1].backward()
    return stats
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        st = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", st[1].item())
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
