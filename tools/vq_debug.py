#!/usr/bin/env python3
"""Find rows where the default VQ kernel disagrees with the C oracle and print their exact distances (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from vqvae_amd import functional as F

def check(K, D, B, H, W, scale, seed=None):
    g = torch.Generator().manual_seed(K * 7 + D + B if seed is None else seed)
    cb = ((torch.rand(K, D, generator=g) * 2 - 1) / K) if scale < 1 else torch.randn(K, D, generator=g)
    z = torch.randn(B, D, H, W, generator=g) * scale
    ref = c_oracle.vq_forward(z.numpy(), cb.numpy(), 0.25)
    dev = torch.device("cuda:0")
    zr = z.permute(0, 2, 3, 1).contiguous().to(dev)
    loss, zq, ppl, idx, hist = F.vq_forward(zr, cb.to(dev), 0.25, rowmajor=True)
    idx = idx.cpu().numpy().reshape(-1); ri = ref["idx"].reshape(-1)
    bad = np.nonzero(idx != ri)[0]
    print(f"K={K} D={D} N={len(ri)} scale={scale}: {len(bad)} mismatches", bad[:10])
    zf = zr.cpu().numpy().reshape(-1, D).astype(np.float64); e = cb.numpy().astype(np.float64)
    for n in bad[:5]:
        d = (zf[n] ** 2).sum() + (e ** 2).sum(1) - 2 * e @ zf[n]
        o = np.argsort(d)[:4]
        print("  row", n, "row%64", n % 64, "got", idx[n], "want", ri[n], "top4", o, d[o] - d[o[0]], "d[got]-dmin", d[idx[n]] - d.min())
        s = e @ zf[n] - 0.5 * (e ** 2).sum(1)
        print("     scores: want", s[ri[n]], "got", s[idx[n]], "|z|", np.linalg.norm(zf[n]), "Emax", np.linalg.norm(e, axis=1).max())

if __name__ == "__main__":
    check(512, 64, 33, 8, 8, 1.0)
    for s in range(5):
        check(512, 64, 64, 8, 8, 1.0, seed=100 + s)
        check(512, 64, 64, 8, 8, 0.066, seed=200 + s)
