#!/usr/bin/env python3
"""Micro-benchmark of the fused VQ kernel alone (HIP events on torch's current stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqvae_amd import functional as F, _lib

def run(K, D, B, H, W, rowmajor=False, iters=20, exact=False, trained=False, bf16=False):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K).to(dev)
    z = (torch.randn(B, D, H, W, generator=g) * 0.066).to(dev)
    if trained:   # trained-like regime: E ~ N(0,1), z = E[j] + 0.3 N(0,1)
        cb = torch.randn(K, D, generator=g).to(dev)
        j = torch.randint(0, K, (B * H * W,), generator=g).to(dev)
        z = (cb[j] + 0.3 * torch.randn(B * H * W, D, generator=g).to(dev)).view(B, H, W, D).permute(0, 3, 1, 2).contiguous()
    if rowmajor: z = z.permute(0, 2, 3, 1).contiguous()
    ws = F.vq_workspace(K, D, dev)
    F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, exact_sweep=exact, bf16_filter=bf16)
    for _ in range(3): F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, prepared=True, exact_sweep=exact, bf16_filter=bf16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, prepared=True, exact_sweep=exact, bf16_filter=bf16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    _lib.profile_enable(True)
    for _ in range(iters): F.vq_forward(z, cb, 0.25, rowmajor=rowmajor, workspace=ws, prepared=True, exact_sweep=exact, bf16_filter=bf16)
    kms, kn = _lib.profile_collect('vq_main')
    _lib.profile_enable(False)
    N = B * H * W
    print(json.dumps(dict(K=K, D=D, N=N, rowmajor=rowmajor, kernel='exact' if exact else ('bf16_filter' if bf16 else 'auto'), data='trained-like' if trained else 'init', us=round(ms * 1e3, 2), kernel_us=round(kms / max(kn, 1) * 1e3, 2), Grows_s=round(N / ms / 1e6, 3),
                          alg_GBps=round(N * (8 * D + 8) / ms / 1e6, 1), TFLOPs=round(2.0 * N * K * D / ms / 1e9, 1))))

if __name__ == "__main__":
    for rm in (False, True):
        for ex in (False, True):
            run(512, 64, 1024, 8, 8, rm, exact=ex)
            run(512, 64, 4096, 8, 8, rm, exact=ex)
            run(512, 64, 32768, 8, 8, rm, exact=ex)
    run(512, 64, 32768, 8, 8, True, trained=True)
    run(512, 64, 32768, 8, 8, True, trained=True, exact=True)
    run(1024, 64, 512, 56, 56)
    run(8192, 128, 64, 64, 64)
