#!/usr/bin/env python3
"""Repro for the one index the quantizer gets wrong on trained-like (heterogeneous channel scale) z_e (tests/test_parity_hetero_gpu.py,
case coupled2-mixed): every kernel family on the oracle's z_e of that case against the C oracle; mismatching rows are printed and
dumped (row, codebook) for offline analysis."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import c_oracle, torch_port
from tests import hetero
from vqvae_amd import functional as F

dev = torch.device("cuda:0")
sd0 = torch_port.init_state_dict()
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04c")
os.makedirs(out_dir, exist_ok=True)
for kind, seed, img in (("coupled", 2, "mixed"), ("coupled", 1, "normal"), ("coupled", 3, "normal"), ("coupled", 4, "mixed")):
    sd = hetero.rescale_coupled(sd0, seed)
    x = hetero.outlier_images(64, 78, img) if img != "normal" else torch.randn(64, 3, 32, 32, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        ze = torch_port.encode(sd, x.clone(), 2).contiguous()
    cb = sd["vector_quantization.embedding.weight"].contiguous()
    ref = c_oracle.vq_forward(ze.numpy(), cb.numpy(), 0.25)["idx"].reshape(-1)
    zd, cbd = ze.to(dev), cb.to(dev)
    zr = zd.permute(0, 2, 3, 1).contiguous()
    for name, z, kw in (("track rows", zr, dict(rowmajor=True)), ("track nchw", zd, dict(rowmajor=False)),
                        ("filter", zr, dict(rowmajor=True, bf16_filter=True)),
                        ("exact", zr, dict(rowmajor=True, exact_sweep=True))):
        idx = F.vq_forward(z, cbd, 0.25, **kw)[3].cpu().numpy().reshape(-1)
        bad = np.nonzero(idx != ref)[0]
        print(f"{kind}{seed}/{img:6s} {name:11s}: {len(bad)} mismatches {bad[:6].tolist()}", flush=True)
        if len(bad) and name == "track rows":
            r = int(bad[0])
            zrow = ze.permute(0, 2, 3, 1).reshape(-1, 64)[r].numpy()
            one = c_oracle.vq_forward(np.ascontiguousarray(zrow.reshape(1, 64, 1, 1)), cb.numpy(), 0.25, want_dist=True)["dist"][0]
            order = np.argsort(one)[:6]
            print(f"   row {r}: device {idx[r]} oracle {ref[r]}; best six (k, d): {[(int(k), float(one[k])) for k in order]}; d[device] = {float(one[idx[r]])}; "
                  f"|z|^2 = {float((zrow.astype(np.float64) ** 2).sum()):.6g}, max|z| = {float(np.abs(zrow).max()):.4g}, min|z| = {float(np.abs(zrow).min()):.4g}")
            np.savez(os.path.join(out_dir, f"repro_{kind}{seed}_{img}.npz"), z=ze.permute(0, 2, 3, 1).reshape(-1, 64).numpy(), cb=cb.numpy(), bad=bad, idx=idx, ref=ref)
