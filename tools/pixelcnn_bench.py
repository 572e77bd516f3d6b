#!/usr/bin/env python3
"""GatedPixelCNN prior: forward latency and generate() time on the HIP kernels (eager and hipGraph replay) next to
the same module's ops issued through torch (MIOpen / rocBLAS) on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vqvae_amd.pixelcnn import GatedPixelCNN

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GatedPixelCNN(512, 64, 15, 10).to(dev).eval()


def torch_forward(x, label):                                  # pixelcnn/models.py:64-84, 118-127 with torch ops
    t = m.embedding(x.view(-1)).view(x.size() + (-1,)).permute(0, 3, 1, 2)
    xv, xh = t, t
    for i, L in enumerate(m.layers):
        h = L.class_cond_embedding(label)
        hv = L.vert_stack(xv)[:, :, :xv.size(-1), :]
        a, b = (hv + h[:, :, None, None]).chunk(2, dim=1)
        ov = torch.tanh(a) * torch.sigmoid(b)
        hh = L.horiz_stack(xh)[:, :, :, :xh.size(-2)]
        a, b = (L.vert_to_horiz(hv) + hh + h[:, :, None, None]).chunk(2, dim=1)
        o = L.horiz_resid(torch.tanh(a) * torch.sigmoid(b))
        xv, xh = ov, (o + xh if L.residual else o)
    return m.output_conv(xh)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    for B in (64, 1024):
        x = torch.randint(0, 512, (B, 8, 8), device=dev)
        label = torch.randint(0, 10, (B,), device=dev)
        m(x, label)
        th = timeit(lambda: m(x, label))
        tt = timeit(lambda: torch_forward(x, label))
        print(f"forward B={B:5d}: HIP {th * 1e3:7.2f} ms   torch {tt * 1e3:7.2f} ms", flush=True)
    label = torch.arange(10, device=dev).repeat(10)[:64]
    for g in (False, True):
        t0 = time.perf_counter()
        s = m.generate(label, (8, 8), 64, use_graph=g)
        torch.cuda.synchronize()
        print(f"generate(64 samples, 8x8) {'hipGraph' if g else 'eager   '}: {(time.perf_counter() - t0) * 1e3:8.1f} ms", flush=True)
    t0 = time.perf_counter()
    x = torch.zeros((64, 8, 8), dtype=torch.int64, device=dev)
    for i in range(8):
        for j in range(8):
            p = torch.softmax(torch_forward(x, label)[:, :, i, j], -1)
            x[:, i, j].copy_(p.multinomial(1).squeeze(-1))
    torch.cuda.synchronize()
    print(f"generate(64 samples, 8x8) torch ops: {(time.perf_counter() - t0) * 1e3:8.1f} ms", flush=True)
