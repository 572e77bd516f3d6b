#!/usr/bin/env python3
"""Does running micro-batches of one forward on separate HIP streams help?  (Layer kernels of different
micro-batches are in different phases, so one's HBM-heavy epilogue can overlap another's MFMA-heavy reduction.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd.modules import VQVAE

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VQVAE(128, 32, 2, 512, 64, 0.25).to(dev).eval()
B = 4096
x = torch.randn(B, 3, 32, 32, device=dev)


def run(nchunk, nstream, steps=20):
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    chunks = x.chunk(nchunk)

    def step():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        outs = []
        for i, c in enumerate(chunks):
            with torch.cuda.stream(streams[i % nstream]):
                outs.append(m(c))
        for s in streams:
            cur.wait_stream(s)
        return outs
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"chunks={nchunk} streams={nstream}: {dt * 1e3:.3f} ms/step  {B / dt / 1e6:.3f} M img/s", flush=True)


with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
print(f"single batch, one stream: {dt * 1e3:.3f} ms/step  {B / dt / 1e6:.3f} M img/s", flush=True)
for nchunk, nstream in ((2, 1), (2, 2), (4, 2), (4, 4), (8, 4), (8, 8)):
    run(nchunk, nstream)
